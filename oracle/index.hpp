// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's index builder (the inputs of the hot path).
//
// Follows (paths relative to /root/reference/src/Infidex):
//   Api/DocumentFields.cs:124-172      GetSearchableTexts ('§' join, field boundaries)
//   Indexing/VectorModel.cs:73-128     IndexDocument, DetermineFieldWeight
//   Tokenization/Tokenizer.cs:89-139   EnumerateTokensForIndexing (n-grams of padded text, then words)
//   Core/TermCollection.cs:75-139      CountTermUsage  (df++ per occurrence)
//   Core/Term.cs:71-146                FirstCycleAdd / IncrementTermUsageCounter (byte tf, stop terms)
//   Indexing/VectorModel.cs:130-220    BuildInvertedLists (docLengths, fp32 sequential avgdl)
//   Indexing/VectorModel.cs:864-908    BuildWordIdfCache
//   Indexing/Fst/FstIndex.cs:49-351    GetExact / GetByPrefix / GetBySuffix / MatchWithinEditDistance1
//        (the "FST" is an uncompressed trie with label-sorted arcs; its pre-order DFS equals ordinal
//         sorted order of the term strings, which is what this file iterates)
//   Indexing/ShortQuery/PositionalPrefixIndex.cs:55-120, PrefixPosting.cs:109-137  (DocSet per 1-3 char prefix)
#pragma once
#include "text.hpp"
#include "dotnet.hpp"
#include "synonyms.hpp"
#include <unordered_map>
#include <algorithm>
#include <cstring>
#include <cmath>

namespace orc {

constexpr u16 START_PAD = 0xFFFF;
constexpr u16 STOP_PAD = 0xFFFE;

struct UHash {
    size_t operator()(const ustr& s) const noexcept {
        uint64_t h = 1469598103934665603ull;
        for (u16 c : s) { h ^= c; h *= 1099511628211ull; }
        return (size_t)(h ^ (h >> 29));
    }
};

struct Config {               // ConfigurationParameters.cs:101-124 (config 400) + SearchEngine.cs:44-55
    int ngram = 3;
    int startPad = 2;
    int stopPad = 0;
    int stopTermLimit = 1250000;
    float fieldWeights[3] = {1.5f, 1.25f, 1.0f};
    bool enableCoverage = true;
    bool wordMatcher = true;  // WordMatcherSetup(exact 2..8, LD1 3..8, affix)
    int wmMinExact = 2, wmMaxExact = 8, wmMinLD1 = 3, wmMaxLD1 = 8;
};

struct FieldIn { ustr text; int weight; };   // weight: 0 High, 1 Med, 2 Low (Api/Weight.cs:7-26)

inline float compute_idf(int totalDocs, int df) {   // Bm25Scorer.cs:686-695
    if (df <= 0 || totalDocs <= 0) return 0.f;
    float d = (float)df, N = (float)totalDocs;
    float ratio = (N - d + 0.5f) / (d + 0.5f);
    return ratio <= 0.f ? 0.f : logf(ratio + 1.f);
}

struct Index {
    Config cfg;
    SynonymMap syn;                      // SearchEngine(..., synonymMap): empty unless a test adds pairs before indexing
    int N = 0;
    // documents
    std::vector<int64_t> docKey;
    std::vector<u16> textArena;          // raw IndexedText (concatenated fields), not normalised
    std::vector<uint64_t> textOff;       // N+1
    std::unordered_map<int64_t, int> keyToFirstId;
    // terms (id = order of first appearance, TermCollection._termList)
    std::vector<ustr> termText;
    std::vector<int> termDf;             // -1 = stop term
    std::vector<std::vector<int32_t>> postDoc;   // build-time
    std::vector<std::vector<uint8_t>> postW;
    // flattened CSR (after finalize)
    std::vector<uint64_t> postOff;
    std::vector<int32_t> postDocFlat;
    std::vector<uint8_t> postWFlat;
    std::unordered_map<ustr, int, UHash> termDict;
    std::unordered_map<uint64_t, int> shortDict;     // len 2..3 fast path (TermCollection.cs:88-118)
    std::vector<float> docLen;
    float avgdl = 0.f;
    // Document.Deleted (Core/Document.cs): set after indexing.  Postings, df, doc lengths and avgdl are NOT touched (the reference only rebuilds
    // them on the next full re-index); the query path skips deleted documents at Bm25Scorer.cs:323,456,623 and SearchPipeline.cs:405,464,535.
    std::vector<uint8_t> deleted;
    bool is_deleted(int d) const { return !deleted.empty() && deleted[(size_t)d]; }
    std::vector<int> sortedTerms;        // term ids in ordinal string order (trie pre-order)
    std::unordered_map<ustr, float, UHash> wordIdf;   // keys folded with to_upper_inv (OrdinalIgnoreCase dictionary)
    // prefix DocSets: key = packed (len, c0,c1,c2)
    std::unordered_map<uint64_t, std::vector<int32_t>> prefixDocs;

    uview raw_text(int d) const { return uview(textArena.data() + textOff[d], (size_t)(textOff[d + 1] - textOff[d])); }
    const int32_t* pdoc(int t) const { return postDocFlat.data() + postOff[t]; }
    const uint8_t* pw(int t) const { return postWFlat.data() + postOff[t]; }
    int plen(int t) const { return (int)(postOff[t + 1] - postOff[t]); }

    static uint64_t pack_short(uview s) {
        uint64_t k = (uint64_t)s.size();
        for (u16 c : s) k = (k << 16) | c;
        return k;
    }
    int get_term(uview s) const {     // TermCollection.GetTerm / FstIndex.GetExact
        if (s.size() >= 2 && s.size() <= 3) {
            auto it = shortDict.find(pack_short(s));
            return it == shortDict.end() ? -1 : it->second;
        }
        auto it = termDict.find(ustr(s));
        return it == termDict.end() ? -1 : it->second;
    }
    int count_term_usage(uview s) {
        int id;
        if (s.size() >= 2 && s.size() <= 3) {
            uint64_t k = pack_short(s);
            auto it = shortDict.find(k);
            if (it != shortDict.end()) id = it->second;
            else { id = new_term(s); shortDict.emplace(k, id); termDf[id] = 1; return id; }
        } else {
            auto it = termDict.find(ustr(s));
            if (it != termDict.end()) id = it->second;
            else { id = new_term(s); termDict.emplace(ustr(s), id); termDf[id] = 1; return id; }
        }
        // IncrementTermUsageCounter (Term.cs:134-146)
        if (termDf[id] != -1) { termDf[id]++; if (termDf[id] > cfg.stopTermLimit) termDf[id] = -1; }
        return id;
    }
    int new_term(uview s) {
        termText.emplace_back(s); termDf.push_back(0); postDoc.emplace_back(); postW.emplace_back();
        return (int)termText.size() - 1;
    }
    // Term.FirstCycleAdd (Term.cs:71-122), removeDuplicates=false
    void first_cycle_add(int id, int doc, float fw) {
        if (termDf[id] < 0) return;
        auto& D = postDoc[id]; auto& W = postW[id];
        if ((int)W.size() < cfg.stopTermLimit) {
            if (D.empty() || D.back() != doc) {
                double r = dotnet::round_even((double)fw);
                W.push_back((uint8_t)std::min(r, 255.0));
                D.push_back(doc);
            } else {
                float nw = (float)W.back() + fw;
                if (nw <= 255.f) { W.back() = (uint8_t)dotnet::round_even((double)nw); termDf[id]--; }
            }
            return;
        }
        termDf[id] = -1; D.clear(); W.clear();
    }

    // VectorModel.IndexDocument (VectorModel.cs:73-112)
    void add_document(int64_t key, const std::vector<FieldIn>& fieldsIn) {
        int doc = N++;
        docKey.push_back(key);
        if (!keyToFirstId.count(key)) keyToFirstId[key] = doc;
        // GetSearchAbleFieldList: OrderBy(weight) stable
        std::vector<const FieldIn*> fs;
        for (auto& f : fieldsIn) fs.push_back(&f);
        std::stable_sort(fs.begin(), fs.end(), [](const FieldIn* a, const FieldIn* b) { return a->weight < b->weight; });
        ustr concat;
        std::vector<std::pair<int, int>> bounds;   // (position, weightIdx)
        for (size_t i = 0; i < fs.size(); i++) {
            bounds.push_back({(int)(uint16_t)concat.size(), fs[i]->weight});
            concat += fs[i]->text;
            if (i + 1 < fs.size()) concat.push_back(u'§');
        }
        std::stable_sort(bounds.begin(), bounds.end(), [](auto& a, auto& b) { return a.first < b.first; });
        if (textOff.empty()) textOff.push_back(0);
        textArena.insert(textArena.end(), concat.begin(), concat.end());
        textOff.push_back(textArena.size());

        ustr indexText = to_lower_inv(default_normalizer().normalize(concat));
        if (syn.has()) indexText = syn.canonicalize(indexText);          // VectorModel.cs:90-93
        if (indexText.empty()) return;   // Tokenizer.cs:91-92
        // Tokenizer normalises again (idempotent on already-normalised text? not in general: lower-casing can
        // create new mappable chars) — restate literally:
        ustr t2 = default_normalizer().normalize(indexText);
        auto field_weight = [&](int pos) -> float {
            if (bounds.empty()) return 1.0f;
            int wi = 0;
            for (auto& b : bounds) { if (b.first <= pos) wi = b.second; else break; }
            return wi < 3 ? cfg.fieldWeights[wi] : 1.0f;
        };
        ustr padded(cfg.startPad, START_PAD); padded += t2; padded.append(cfg.stopPad, STOP_PAD);
        int n = cfg.ngram;
        if ((int)padded.size() >= n) {
            for (int i = 0; i + n <= (int)padded.size(); i++) {
                uview g(padded.data() + i, n);
                bool allpad = true;
                for (u16 c : g) if (c != START_PAD && c != STOP_PAD) { allpad = false; break; }
                if (allpad) continue;
                int id = count_term_usage(g);
                first_cycle_add(id, doc, field_weight(i));
            }
        }
        std::vector<Slice> words; split_words(t2, words);
        for (auto& w : words) {
            if (w.len >= n) {
                int id = count_term_usage(uview(t2.data() + w.off, w.len));
                first_cycle_add(id, doc, field_weight(cfg.startPad + w.off));
            }
        }
        // PositionalPrefixIndex.IndexDocument(indexText, doc.Id)  (VectorModel.cs:109) — note: indexText, not t2
        std::vector<Slice> toks; split_words(indexText, toks);
        for (auto& w : toks) {
            int mx = std::min(w.len, 3);
            for (int L = 1; L <= mx; L++) {
                uint64_t k = pack_short(uview(indexText.data() + w.off, L));
                auto& v = prefixDocs[k];
                if (v.empty() || v.back() != doc) v.push_back(doc);
            }
        }
    }

    // BuildInvertedLists + BuildWordIdfCache + BuildOptimizedIndexes
    void finalize() {
        int T = (int)termText.size();
        docLen.assign(N, 0.f);
        postOff.assign(T + 1, 0);
        for (int t = 0; t < T; t++) {
            size_t len = termDf[t] > 0 ? postDoc[t].size() : 0;
            postOff[t + 1] = postOff[t] + len;
        }
        postDocFlat.resize(postOff[T]); postWFlat.resize(postOff[T]);
        for (int t = 0; t < T; t++) {
            if (termDf[t] > 0) {
                std::memcpy(postDocFlat.data() + postOff[t], postDoc[t].data(), postDoc[t].size() * 4);
                std::memcpy(postWFlat.data() + postOff[t], postW[t].data(), postW[t].size());
                for (size_t i = 0; i < postDoc[t].size(); i++) docLen[postDoc[t][i]] += (float)postW[t][i];
            }
            std::vector<int32_t>().swap(postDoc[t]); std::vector<uint8_t>().swap(postW[t]);
        }
        float total = 0.f;
        for (int d = 0; d < N; d++) total += docLen[d];
        avgdl = N > 0 ? total / (float)N : 0.f;
        build_sorted_terms();
        build_word_idf();
    }
    void build_sorted_terms() {
        int T = (int)termText.size();
        sortedTerms.resize(T);
        for (int i = 0; i < T; i++) sortedTerms[i] = i;
        std::sort(sortedTerms.begin(), sortedTerms.end(), [&](int a, int b) { return termText[a] < termText[b]; });
    }
    void build_word_idf() {
        std::unordered_map<ustr, int, UHash> wdf;
        std::vector<Slice> words;
        std::vector<ustr> uniq;
        for (int d = 0; d < N; d++) {
            uview raw = raw_text(d);
            if (raw.empty()) continue;
            ustr norm = default_normalizer().normalize(to_lower_inv(raw));
            split_words(norm, words);
            uniq.clear();
            for (auto& w : words) { ustr k(norm.data() + w.off, w.len); for (auto& c : k) c = to_upper_inv(c); uniq.push_back(std::move(k)); }
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            for (auto& k : uniq) wdf[k]++;
        }
        wordIdf.clear();
        wordIdf.reserve(wdf.size());
        for (auto& kv : wdf) if (kv.second > 0 && kv.second <= N) wordIdf[kv.first] = compute_idf(N, kv.second);
    }
    bool word_idf(uview token, float& out) const {
        ustr k(token); for (auto& c : k) c = to_upper_inv(c);
        auto it = wordIdf.find(k);
        if (it == wordIdf.end()) return false;
        out = it->second; return true;
    }
    const std::vector<int32_t>* prefix_docset(uview p) const {
        auto it = prefixDocs.find(pack_short(p));
        return it == prefixDocs.end() ? nullptr : &it->second;
    }

    // FstIndex.MatchWithinEditDistance1 (FstIndex.cs:202-351): Myers bit-vector state carried down the trie,
    // *search* variant (no |1 on the horizontal delta => score = best match of the query against a SUFFIX of
    // the path), every node to depth m+1 is visited in label order; a final node with score<=1 is reported.
    // Returns total match count; first `cap` outputs stored.
    int match_ld1(uview q, std::vector<int>& out, int cap = 1024) const {
        out.clear();
        int m = (int)q.size();
        if (termText.empty()) return 0;
        int count = 0;
        if (m == 0) return 0;   // no empty / 1-char terms can be final below n-gram size 3; kept trivial
        if (m > 64) return match_ld1_slow(q, out, cap);
        struct St { uint64_t vp, vn; int score; };
        std::vector<St> st(m + 2);
        st[0] = {~0ull, 0ull, m};
        uint64_t maskM = 1ull << (m - 1);
        const ustr* prev = nullptr;
        for (int id : sortedTerms) {
            const ustr& s = termText[id];
            int L = (int)s.size();
            if (L > m + 1) continue;     // deeper nodes are never expanded (Depth >= m+1 -> continue)
            int lcp = 0;
            if (prev) { int mx = std::min((int)prev->size(), L); while (lcp < mx && (*prev)[lcp] == s[lcp]) lcp++; }
            for (int dpt = lcp; dpt < L; dpt++) {
                u16 c = s[dpt];
                uint64_t pm = 0;
                for (int i = 0; i < m; i++) if (q[i] == c) pm |= 1ull << i;
                const St& f = st[dpt];
                uint64_t x = pm | f.vn;
                uint64_t d0 = ((f.vp + (x & f.vp)) ^ f.vp) | x;
                uint64_t hn = f.vp & d0;
                uint64_t hp = f.vn | ~(f.vp | d0);
                uint64_t nvp = (hn << 1) | ~(d0 | (hp << 1));
                uint64_t nvn = d0 & (hp << 1);
                int ns = f.score;
                if (hp & maskM) ns++;
                if (hn & maskM) ns--;
                st[dpt + 1] = {nvp, nvn, ns};
            }
            if (st[L].score <= 1) { if (count < cap) out.push_back(id); count++; }
            prev = &s;
        }
        return count;
    }
    // MatchEditDistance1Slow (FstIndex.cs:363-440): true global Levenshtein row DP, stops once `cap` outputs.
    int match_ld1_slow(uview q, std::vector<int>& out, int cap) const {
        int m = (int)q.size();
        int count = 0;
        std::vector<std::vector<int>> rows(1, std::vector<int>(m + 1));
        for (int i = 0; i <= m; i++) rows[0][i] = i;
        // stack-based DFS pops arcs in DESCENDING label order here (pushes ascending, pops last first)
        std::vector<int> order(sortedTerms.rbegin(), sortedTerms.rend());
        // Descending ordinal order is not exactly reverse pre-order (parents come after children in a plain
        // reverse); emulate: children descending, parent before children => sort with custom comparator.
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            const ustr& x = termText[a]; const ustr& y = termText[b];
            size_t n = std::min(x.size(), y.size());
            for (size_t i = 0; i < n; i++) if (x[i] != y[i]) return x[i] > y[i];
            return x.size() < y.size();
        });
        for (int id : order) {
            const ustr& s = termText[id];
            std::vector<int> row(m + 1), prevRow(m + 1);
            for (int i = 0; i <= m; i++) prevRow[i] = i;
            bool pruned = false;
            for (size_t d = 0; d < s.size(); d++) {
                int mn = prevRow[0]; for (int i = 1; i <= m; i++) mn = std::min(mn, prevRow[i]);
                if (mn > 1) { pruned = true; break; }
                row[0] = prevRow[0] + 1;
                for (int i = 1; i <= m; i++) {
                    int cost = (q[i - 1] == s[d]) ? 0 : 1;
                    row[i] = std::min(std::min(row[i - 1] + 1, prevRow[i] + 1), prevRow[i - 1] + cost);
                }
                std::swap(row, prevRow);
            }
            if (pruned) continue;
            if (prevRow[m] <= 1) { if (count < cap) { out.push_back(id); count++; } if (count >= cap) return count; }
        }
        return count;
    }
};

} // namespace orc
