// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's Stage 1 (query terms -> candidate selection -> BM25+ top-k).
//
// Follows (paths relative to /root/reference/src/Infidex):
//   Tokenization/Tokenizer.cs:144-196     EnumerateShinglesForSearch (words first, then n-grams of padded text)
//   Indexing/VectorModel.cs:376-602       SearchWithMaxScore (<=128 raw tokens, sort by (termId,text), RLE dedupe,
//                                         fuzzy expansion, idf / maxScore)
//   Indexing/VectorModel.cs:643-743       ExpandMissingTerm (LD1 union -> virtual term, tf==1, LRU(1000))
//   Scoring/TieredCandidateSelector.cs:53-237,243-322,328-437,455-532   candidate tiers (quirk Q11)
//   Indexing/ArrayPostingsEnum.cs:38-135, RoaringPostingsEnum.cs:26-107  NextDoc / Advance
//   Indexing/Bm25Scorer.cs:56-193,195-330,332-445,524-533,643-670        chunked scoring, two BM25 formulas (Q9),
//                                         pruning heap (BCL PriorityQueue, Q10)
//   Core/TopKHeap.cs, Core/ScoreEntry.cs:25-36, Scoring/SegmentProcessor.cs:15-37
#pragma once
#include <atomic>
#include <chrono>
#include "index.hpp"
#include <list>
#include <memory>
#include <limits>

namespace orc {

// Where a query's time goes (bench.py's cpu_baseline.stage_ms): nanoseconds summed over all threads — [0] planning (tokens, term lookup, fuzzy expansion, idf),
// [1] candidate selection (TieredCandidateSelector), [2] BM25+ scoring and the top-k heap (Bm25Scorer), [3] WordMatcher, [4] coverage + fusion (Stage 2).
inline std::atomic<unsigned long long>& stage_ns(int i) { static std::atomic<unsigned long long> a[5]; return a[i]; }
struct StageClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int i) { const auto n = std::chrono::steady_clock::now(); stage_ns(i).fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(), std::memory_order_relaxed); t = n; }
};

constexpr int NO_MORE_DOCS = std::numeric_limits<int>::max();

struct ScoreEntry {
    float score; int64_t key; uint8_t tie;
    // ScoreEntry.CompareTo
    static int compare(const ScoreEntry& a, const ScoreEntry& b) {
        int c = dotnet::cmp_float(a.score, b.score); if (c) return c;
        if (a.tie != b.tie) return a.tie < b.tie ? -1 : 1;
        return b.key < a.key ? -1 : (b.key > a.key ? 1 : 0);   // other.DocumentId.CompareTo(DocumentId)
    }
};

struct TopKHeap {   // Core/TopKHeap.cs
    struct Cmp { int operator()(const ScoreEntry& a, const ScoreEntry& b) const { return ScoreEntry::compare(a, b); } };
    dotnet::PriorityQueue<ScoreEntry, ScoreEntry, Cmp> pq{Cmp{}};
    int limit;
    explicit TopKHeap(int l) : limit(l) {}
    void add(const ScoreEntry& e) {
        if (pq.count() < limit) pq.enqueue(e, e);
        else if (ScoreEntry::compare(e, pq.peek().e) > 0) { pq.dequeue(); pq.enqueue(e, e); }
    }
    std::vector<ScoreEntry> get_topk() {
        std::vector<ScoreEntry> r(pq.count());
        for (int i = 0, n = (int)r.size(); i < n; i++) r[n - 1 - i] = pq.dequeue().e;
        return r;
    }
};

// SegmentProcessor.ConsolidateSegments: best entry per key, descending total order
inline std::vector<ScoreEntry> consolidate(const std::vector<ScoreEntry>& in) {
    std::unordered_map<int64_t, size_t> pos;
    std::vector<ScoreEntry> best;
    for (auto& e : in) {
        auto it = pos.find(e.key);
        if (it == pos.end()) { pos[e.key] = best.size(); best.push_back(e); }
        else if (ScoreEntry::compare(e, best[it->second]) > 0) best[it->second] = e;
    }
    dotnet::sort(best, [](const ScoreEntry& a, const ScoreEntry& b) { return ScoreEntry::compare(b, a); });
    return best;
}

// A posting source: either an index term (docIds + byte weights) or a fuzzy virtual term (doc set, tf == 1)
struct PostingSrc { const int32_t* docs; const uint8_t* w; int n; };

struct PostingsEnum {   // ArrayPostingsEnum / RoaringPostingsEnum (same observable behaviour)
    PostingSrc s; int pos = -1; int doc = -1;
    explicit PostingsEnum(PostingSrc src) : s(src) {}
    float freq() const { return (pos >= 0 && pos < s.n) ? (s.w ? (float)s.w[pos] : 1.f) : 0.f; }
    int next_doc() { pos++; if (pos >= s.n) { doc = NO_MORE_DOCS; return doc; } doc = s.docs[pos]; return doc; }
    int advance(int target) {
        if (doc == NO_MORE_DOCS) return doc;
        if (target <= doc) return doc;
        int start = pos + 1, count = s.n;
        if (start >= count) { pos = count; doc = NO_MORE_DOCS; return doc; }
        int limit = count - 1, jump = 1, high = start;
        while (high <= limit && s.docs[high] < target) { start = high + 1; high += jump; jump *= 2; }
        if (high > limit) high = limit;
        if (start <= high) pos = (int)(std::lower_bound(s.docs + start, s.docs + high + 1, target) - s.docs);
        else pos = start;
        if (pos >= count) { doc = NO_MORE_DOCS; return doc; }
        doc = s.docs[pos]; return doc;
    }
    long cost() const { return s.n; }
};

struct TermScoreInfo { PostingSrc src; int df; float idf; float maxScore; int termId; };

struct FuzzyTerm { std::vector<int32_t> docs; };

struct Stage1Stats { long candidates = 0; long postingsTouched = 0; int mode = 0; };

struct Stage1 {
    const Index& ix;
    // LruCache<string, Term>(1000)  (VectorModel.cs:42,745-802)
    std::list<std::pair<ustr, std::shared_ptr<FuzzyTerm>>> lruList;
    std::unordered_map<ustr, decltype(lruList)::iterator, UHash> lruMap;
    std::vector<float> upperBounds;   // rented float[totalDocs], cleared per query (Bm25Scorer.cs:79-80)
    Stage1Stats stats;

    explicit Stage1(const Index& i) : ix(i) {}

    std::shared_ptr<FuzzyTerm> lru_get(const ustr& k) {
        auto it = lruMap.find(k);
        if (it == lruMap.end()) return nullptr;
        lruList.splice(lruList.begin(), lruList, it->second);
        return it->second->second;
    }
    void lru_put(const ustr& k, std::shared_ptr<FuzzyTerm> v) {
        if (lruMap.size() >= 1000) { lruMap.erase(lruList.back().first); lruList.pop_back(); }
        lruList.emplace_front(k, v); lruMap[k] = lruList.begin();
    }

    struct RawToken { int termId; ustr text; };

    // Tokenizer.EnumerateShinglesForSearch + the visitor of VectorModel.cs:403-431
    void raw_tokens(uview queryText, std::vector<RawToken>& raw) const {
        raw.clear();
        ustr text = default_normalizer().normalize(queryText);
        int n = ix.cfg.ngram;
        auto visit = [&](uview span) {
            if (raw.size() >= 128) return;
            int id = ix.get_term(span);
            if (id >= 0) raw.push_back({id, ustr()});
            else raw.push_back({-1, ustr(span)});
        };
        std::vector<Slice> words; split_words(text, words);
        for (auto& w : words) if (w.len >= n) visit(uview(text.data() + w.off, w.len));
        ustr padded(ix.cfg.startPad, START_PAD); padded += text; padded.append(ix.cfg.stopPad, STOP_PAD);
        if ((int)padded.size() >= n)
            for (int i = 0; i + n <= (int)padded.size(); i++) {
                uview g(padded.data() + i, n);
                bool allpad = true; for (u16 c : g) if (c != START_PAD && c != STOP_PAD) { allpad = false; break; }
                if (allpad) continue;
                visit(g);
            }
    }

    // ---- TieredCandidateSelector ---------------------------------------------------------------
    struct TermInfo { const TermScoreInfo* t; float idf; float maxScore; };

    static void sort_idf_desc(std::vector<TermInfo>& v) {
        dotnet::sort(v, [](const TermInfo& a, const TermInfo& b) { return dotnet::cmp_float(b.idf, a.idf); });
    }
    std::vector<int32_t> intersect_terms(const std::vector<TermInfo>& terms, float tierUB) {
        std::vector<int32_t> out;
        if (terms.empty()) return out;
        std::vector<PostingsEnum> en;
        for (auto& t : terms) en.emplace_back(t.t->src);
        dotnet::sort(en, [](const PostingsEnum& a, const PostingsEnum& b) { return a.cost() < b.cost() ? -1 : (a.cost() > b.cost() ? 1 : 0); });
        PostingsEnum& driver = en[0];
        int doc = driver.next_doc();
        while (doc != NO_MORE_DOCS) {
            bool match = true;
            for (size_t i = 1; i < en.size(); i++) {
                int target = en[i].advance(doc);
                if (target > doc) {
                    match = false;
                    doc = target;
                    if (doc == NO_MORE_DOCS) return out;
                    doc = driver.advance(doc);
                    if (doc == NO_MORE_DOCS) return out;
                    break;
                }
            }
            if (match) {
                out.push_back(doc);
                if (upperBounds[doc] == 0) upperBounds[doc] = tierUB;
                doc = driver.next_doc();
            }
        }
        return out;
    }
    std::vector<int32_t> select_disjunctive(std::vector<TermInfo>& terms, int topK) {
        float maxIdf = 0.f;
        for (auto& t : terms) if (t.idf > maxIdf) maxIdf = t.idf;
        sort_idf_desc(terms);
        bool hasSelective = false; int localCount = 0;
        std::vector<int32_t> result;
        for (auto& ti : terms) {
            bool lowq = ti.idf < (maxIdf * 0.2f);
            if (terms.size() > 1 && lowq && hasSelective) continue;
            PostingsEnum p(ti.t->src);
            std::vector<int32_t> buf;
            while (true) {
                int d = p.next_doc();
                if (d == NO_MORE_DOCS) break;
                float ub = upperBounds[d];
                if (ub == 0) { upperBounds[d] = ti.maxScore; localCount++; }
                else upperBounds[d] = ub + ti.maxScore;
                buf.push_back(d);
            }
            stats.postingsTouched += (long)buf.size();
            union_sorted(result, buf);
            if (!lowq && localCount > 0) hasSelective = true;
            if (localCount >= topK * 100) break;
        }
        return result;
    }
    static void union_sorted(std::vector<int32_t>& acc, const std::vector<int32_t>& add) {
        if (add.empty()) return;
        if (acc.empty()) { acc = add; return; }
        std::vector<int32_t> r; r.reserve(acc.size() + add.size());
        std::set_union(acc.begin(), acc.end(), add.begin(), add.end(), std::back_inserter(r));
        acc.swap(r);
    }
    // TrySelectPrefixCandidates: returns pointer to the accepted DocSet (or nullptr)
    const std::vector<int32_t>* try_prefix(uview originalQuery, int topK, int queryTermCount, float& ub) const {
        ub = 0.f;
        ustr ql = to_lower_inv(originalQuery);
        int maxLen = std::min((int)ql.size(), 3);
        for (int len = maxLen; len >= 1; len--) {
            const std::vector<int32_t>* ds = ix.prefix_docset(uview(ql.data(), len));
            if (!ds || ds->empty()) continue;
            long pop = (long)ds->size();
            if (pop > (long)topK * 20) continue;
            if (pop > 0 && pop <= (long)topK * 10) { ub = queryTermCount * 10.f; return ds; }
        }
        return nullptr;
    }
    std::vector<int32_t> select_candidates(const std::vector<TermScoreInfo>& q, int topK, uview originalQuery) {
        std::vector<int32_t> empty;
        if (q.empty()) return empty;
        if (!originalQuery.empty()) {
            float pub;
            const std::vector<int32_t>* pc = try_prefix(originalQuery, topK, (int)q.size(), pub);
            if (pc && !pc->empty()) {
                for (int d : *pc) upperBounds[d] = pub;
                if ((long)pc->size() >= std::min(topK * 2, 100)) { stats.mode = 1; return *pc; }
            }
        }
        std::vector<TermInfo> terms;
        int missing = 0;
        for (auto& t : q) { if (t.df <= 0) { missing++; continue; } terms.push_back({&t, t.idf, t.maxScore}); }
        if (terms.empty()) return empty;
        bool typo = false; float maxIdf = 0.f;
        for (auto& t : terms) { if (t.t->df < 10) typo = true; if (t.idf > maxIdf) maxIdf = t.idf; }
        if (typo || missing > 0 || q.size() == 1) { stats.mode = 2; return select_disjunctive(terms, topK); }
        stats.mode = 3;
        sort_idf_desc(terms);
        std::vector<int32_t> global;
        if (terms.size() >= 2) {
            float ub0 = 0.f; for (auto& t : terms) ub0 += t.maxScore;
            auto t0 = intersect_terms(terms, ub0);
            union_sorted(global, t0);
            if ((long)global.size() >= (long)topK * 2) return global;
        }
        if (terms.size() >= 3 && (long)global.size() < (long)topK * 3) {
            std::vector<TermInfo> t1(terms.begin(), terms.end() - 1);
            float ub1 = 0.f; for (auto& t : t1) ub1 += t.maxScore;
            auto r1 = intersect_terms(t1, ub1);
            union_sorted(global, r1);
        }
        if ((long)global.size() < (long)topK * 5) {
            std::vector<TermInfo> sel;
            size_t cap = std::min<size_t>(2, terms.size());
            float cutoff = maxIdf * 0.3f;
            for (auto& t : terms) {
                if (t.idf <= 0.f) continue;
                if (t.idf < cutoff) continue;
                sel.push_back(t);
                if (sel.size() == cap) break;
            }
            for (auto& ti : sel) {
                PostingsEnum p(ti.t->src);
                std::vector<int32_t> buf;
                while (true) {
                    int d = p.next_doc();
                    if (d == NO_MORE_DOCS) break;
                    if (upperBounds[d] == 0) upperBounds[d] = ti.maxScore;
                    buf.push_back(d);
                }
                stats.postingsTouched += (long)buf.size();
                union_sorted(global, buf);
                if ((long)global.size() >= (long)topK * 10) break;
            }
        }
        return global;
    }

    // ---- Bm25Scorer ------------------------------------------------------------------------------
    static float term_score_scalar(float tf, float dl, float avgdl, float idf) {   // ComputeTermScore :643-652
        const float K1 = 1.2f, B = 0.75f, Delta = 1.0f;
        float normFactor = K1 * (1.f - B + B * (dl / avgdl));
        float denom = tf + normFactor;
        if (denom <= 0.f) return 0.f;
        float core = (tf * (K1 + 1.f)) / denom;
        return idf * (core + Delta);
    }
    struct PruneCmp { int operator()(float a, float b) const { return dotnet::cmp_float(a, b); } };
    using PruneHeap = dotnet::PriorityQueue<int, float, PruneCmp>;

    static void update_topk(int id, float score, int topK, PruneHeap& h, float& threshold) {
        if (topK >= std::numeric_limits<int>::max()) return;
        if (h.count() < topK) { h.enqueue(id, score); if (h.count() == topK) threshold = h.peek().p; }
        else if (score > threshold) { h.enqueue_dequeue(id, score); threshold = h.peek().p; }
    }

    void score_block(PostingsEnum& p, const TermScoreInfo& info, float remainingMax, int topK, float avgdl,
                     std::vector<float>& scoreBlock, const int32_t* docBlock, int count,
                     std::vector<float>& tfBlock, std::vector<int>& indexBlock, float threshold) {
        int mc = 0;
        for (int j = 0; j < count; j++) {
            float cur = scoreBlock[j];
            if (topK < std::numeric_limits<int>::max() && cur + info.maxScore + remainingMax <= threshold) continue;
            int target = docBlock[j];
            int d = p.advance(target);
            if (d == target) { indexBlock[mc] = j; tfBlock[mc] = p.freq(); mc++; }
        }
        if (mc == 0) return;
        const float k1 = 1.2f, b = 0.75f, delta = 1.0f, idf = info.idf;
        const float k1p1 = k1 + 1.0f, minDlNorm = 1.f - b, bDivAvg = b / avgdl;
        int i = 0;
        for (; i <= mc - 8; i += 8) {        // Vector256 lanes: K1*((1-B) + (B/avgdl)*dl)
            for (int l = 0; l < 8; l++) {
                float tf = tfBlock[i + l];
                float dl = ix.docLen[docBlock[indexBlock[i + l]]];
                float t1 = bDivAvg * dl;
                float t2 = minDlNorm + t1;
                float norm = k1 * t2;
                float denom = tf + norm;
                float core = (tf * k1p1) / denom;
                float sc = idf * (core + delta);
                scoreBlock[indexBlock[i + l]] += sc;
            }
        }
        for (; i < mc; i++) {
            float tf = tfBlock[i]; int idx = indexBlock[i];
            float dl = ix.docLen[docBlock[idx]];
            if (dl <= 0.f) dl = 1.f;
            scoreBlock[idx] += term_score_scalar(tf, dl, avgdl, info.idf);
        }
    }

    // Bm25Scorer.Search -> TopKHeap of (DocumentKey, score)
    std::vector<ScoreEntry> bm25_search(const std::vector<TermScoreInfo>& terms, int topK, uview originalQuery) {
        TopKHeap result(topK);
        if (terms.empty() || ix.N == 0) return {};
        float avgdl = ix.avgdl > 0.f ? ix.avgdl : 1.f;
        upperBounds.assign((size_t)ix.N, 0.f);
        StageClock clk_;
        std::vector<int32_t> cand = select_candidates(terms, topK, originalQuery);
        clk_.lap(1);
        struct ScoreLap { StageClock& c; ~ScoreLap() { c.lap(2); } } scoreLap_{clk_};      // everything from here to the return is scoring
        stats.candidates = (long)cand.size();
        int T = (int)terms.size();
        std::vector<float> suffix(T + 1, 0.f);
        for (int i = T - 1; i >= 0; i--) suffix[i] = suffix[i + 1] + terms[i].maxScore;
        PruneHeap heap{PruneCmp{}};
        float threshold = 0.f;
        if (!cand.empty()) {
            std::vector<PostingsEnum> en;
            for (auto& t : terms) en.emplace_back(t.src);
            std::vector<float> scoreBlock(4096), tfBlock(4096); std::vector<int> indexBlock(4096);
            size_t i = 0;
            while (i < cand.size()) {
                // one roaring container = one 65536-id range; chunks of <= 4096 inside it
                int hi = cand[i] >> 16;
                size_t j = i;
                while (j < cand.size() && (cand[j] >> 16) == hi) j++;
                size_t processed = i;
                while (processed < j) {
                    int chunk = (int)std::min<size_t>(4096, j - processed);
                    std::fill(scoreBlock.begin(), scoreBlock.begin() + chunk, 0.f);
                    const int32_t* docBlock = cand.data() + processed;
                    for (int t = 0; t < T; t++) {
                        if (terms[t].idf <= 0.f) continue;
                        score_block(en[t], terms[t], suffix[t + 1], topK, avgdl, scoreBlock, docBlock, chunk, tfBlock, indexBlock, threshold);
                    }
                    for (int c = 0; c < chunk; c++) {
                        float s = scoreBlock[c];
                        // Bm25Scorer.cs:318-327: a deleted document is scored with its chunk but never offered to the pruning heap
                        if (s > 0.f && !ix.is_deleted(docBlock[c])) update_topk(docBlock[c], s, topK, heap, threshold);
                    }
                    processed += chunk;
                }
                i = j;
            }
        } else {
            // full scan (Bm25Scorer.cs:153-176, 589-641)
            std::vector<float> docScores((size_t)ix.N, 0.f);
            for (int t = 0; t < T; t++) {
                if (terms[t].idf <= 0.f) continue;
                PostingsEnum p(terms[t].src);
                while (true) {
                    int d = p.next_doc();
                    if (d == NO_MORE_DOCS) break;
                    if ((unsigned)d >= (unsigned)ix.N) continue;
                    float cur = docScores[d];
                    if (topK < std::numeric_limits<int>::max() && heap.count() >= topK) {
                        if (cur + terms[t].maxScore + suffix[t + 1] <= threshold) continue;
                    }
                    if (ix.is_deleted(d)) continue;                 // Bm25Scorer.cs:622-624
                    float tf = p.freq(); if (tf <= 0.f) continue;
                    float dl = ix.docLen[d]; if (dl <= 0.f) dl = 1.f;
                    float ns = cur + term_score_scalar(tf, dl, avgdl, terms[t].idf);
                    docScores[d] = ns;
                    update_topk(d, ns, topK, heap, threshold);
                }
            }
        }
        while (heap.count() > 0) { auto nd = heap.dequeue(); result.add(ScoreEntry{nd.p, ix.docKey[nd.e], 0}); }
        return result.get_topk();
    }

    // VectorModel.SearchWithMaxScore
    std::vector<ScoreEntry> search_with_maxscore(uview queryText, int topK) {
        StageClock planClk_;
        stats = Stage1Stats();
        std::vector<RawToken> raw; raw_tokens(queryText, raw);
        dotnet::sort(raw, [](const RawToken& a, const RawToken& b) {
            if (a.termId != b.termId) return a.termId < b.termId ? -1 : 1;
            return a.text < b.text ? -1 : (a.text > b.text ? 1 : 0);
        });
        struct Stat { int termId; ustr text; int df; std::shared_ptr<FuzzyTerm> fuzzy; };
        std::vector<Stat> st;
        for (size_t i = 0; i < raw.size(); i++) {
            if (!st.empty()) {
                bool same = raw[i].termId >= 0 ? raw[i].termId == st.back().termId
                                               : (st.back().termId < 0 && raw[i].text == st.back().text);
                if (same) continue;
            }
            Stat s{raw[i].termId, raw[i].text, 0, nullptr};
            if (s.termId >= 0) s.df = ix.termDf[s.termId];   // GatherTermInfo (GetTerm(text) of an unknown text is null)
            st.push_back(std::move(s));
        }
        for (auto& s : st) {
            if (s.df == 0 && s.termId < 0 && s.text.size() >= 4) {
                auto c = lru_get(s.text);
                if (!c) {
                    std::vector<int> matches;
                    ix.match_ld1(s.text, matches, 1024);
                    std::vector<int32_t> all;
                    for (int id : matches) if (ix.termDf[id] > 0 && ix.plen(id) > 0) all.insert(all.end(), ix.pdoc(id), ix.pdoc(id) + ix.plen(id));
                    if (!all.empty()) {
                        std::sort(all.begin(), all.end());
                        all.erase(std::unique(all.begin(), all.end()), all.end());
                        c = std::make_shared<FuzzyTerm>(); c->docs.swap(all);
                        lru_put(s.text, c);
                    }
                }
                if (c) { s.fuzzy = c; s.df = (int)c->docs.size(); }
            }
        }
        float avgdl = ix.avgdl > 0.f ? ix.avgdl : 1.f;
        std::vector<TermScoreInfo> infos;
        std::vector<std::shared_ptr<FuzzyTerm>> keep;
        for (auto& s : st) {
            int df = s.df;
            if (df <= 0 || df > ix.cfg.stopTermLimit) continue;
            float idf = compute_idf(ix.N, df);
            const float maxTf = 255.f, k1 = 1.2f, b = 0.75f, delta = 1.0f;
            float minDlNorm = 1.f - b + b * (1.f / avgdl);
            float maxCore = (maxTf * (k1 + 1.f)) / (maxTf + k1 * minDlNorm);
            float maxScore = idf * (maxCore + delta);
            if (s.fuzzy) { keep.push_back(s.fuzzy); infos.push_back({PostingSrc{s.fuzzy->docs.data(), nullptr, (int)s.fuzzy->docs.size()}, df, idf, maxScore, -1}); }
            else if (s.termId >= 0) infos.push_back({PostingSrc{ix.pdoc(s.termId), ix.pw(s.termId), ix.plen(s.termId)}, df, idf, maxScore, s.termId});
        }
        lastTerms = infos;
        planClk_.lap(0);
        return bm25_search(infos, topK, queryText);
    }
    std::vector<TermScoreInfo> lastTerms;   // exposed for tests / parity harness
};

} // namespace orc
