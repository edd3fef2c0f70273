// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of SearchEngine.Search -> SearchPipeline.Execute (Stage 1 -> Stage 2 -> truncation).
//
// Follows (paths relative to /root/reference/src/Infidex):
//   SearchEngine.cs:256-319               Search (trim, normalise, lower-case, Take(max))
//   Scoring/SearchPipeline.cs:49-206      Execute (short-query gates, coverage gate, fallback)
//   Scoring/SearchPipeline.cs:208-296     ExecuteRelevancyStage  (n-gram path only; QueryAnalyzer.cs:10-54)
//   Scoring/SearchPipeline.cs:298-447     ExecuteCoverageStage   (candidate order Q12, LCS scratch Q7)
//   Scoring/SearchPipeline.cs:449-576     ProcessCandidate, BuildDocumentKeyIndex, PartitionWordMatcherCandidates
//   Scoring/ResultProcessor.cs:146-178    CalculateTruncationIndex
// Out of scope (returns unsupported=true): queries with no word >= n-gram size (ShortQueryProcessor path).
#pragma once
#include "stage1.hpp"
#include "coverage.hpp"
#include "wordmatcher.hpp"
#include <unordered_set>

namespace orc {

struct QueryParams {       // Api/Query.cs defaults
    int maxResults = 10;
    int coverageDepth = 500;
    bool enableCoverage = true;
};

struct CandidateTrace {    // per Stage-2 evaluation, for parity checks of the integer features
    int internalId; float baseScore; float score; uint8_t tie; CoverageFeatures f; int lcs;
};

struct SearchOutput {
    std::vector<ScoreEntry> records;
    std::vector<ScoreEntry> stage1;           // consolidated Stage-1 results (<= depth)
    std::vector<CandidateTrace> trace;        // filled when Engine::keepTrace
    bool unsupported = false;
    bool usedCoverage = false;
    int totalCandidates = 0;
};

struct Engine {
    Index ix;
    WordMatcher wm;
    CoverageEngine cov;
    std::unique_ptr<Stage1> s1;
    bool keepTrace = false;
    bool indexed = false;

    explicit Engine(const Config& cfg = Config()) { ix.cfg = cfg; wm.cfg = &ix.cfg; }

    void add_document(int64_t key, const std::vector<FieldIn>& fields) {
        ix.add_document(key, fields);
        if (ix.cfg.wordMatcher) wm.load(ix.raw_text(ix.N - 1), ix.N - 1);
    }
    void add_document(int64_t key, const ustr& text) { add_document(key, std::vector<FieldIn>{{text, 1}}); }
    void finalize() {
        ix.finalize();
        if (ix.cfg.wordMatcher) wm.finalize_index();
        cov.ix = &ix;
        s1.reset(new Stage1(ix));
        indexed = true;
    }

    // QueryAnalyzer.Analyze
    static void analyze(uview text, int minIndexSize, bool& canUseNGrams, bool& mixed, ustr& longWords) {
        canUseNGrams = false; mixed = false; longWords = ustr(text);
        std::vector<Slice> words; split_words(text, words);
        if (words.empty()) { canUseNGrams = (int)text.size() >= minIndexSize; return; }
        int shortCnt = 0; ustr joined; int longCnt = 0;
        for (auto& w : words) {
            if (w.len >= minIndexSize) { if (longCnt++) joined.push_back(u' '); joined.append(text.substr(w.off, w.len)); }
            else shortCnt++;
        }
        if (longCnt > 0) { canUseNGrams = true; longWords = joined; }
        if (shortCnt > 0 && longCnt > 0) mixed = true;
    }

    SearchOutput search(uview rawQuery, const QueryParams& qp) {
        SearchOutput out;
        if (!indexed) return out;
        // SearchEngine.Search: Trim, Normalize, ToLowerInvariant
        size_t b = 0, e = rawQuery.size();
        while (b < e && is_whitespace(rawQuery[b])) b++;
        while (e > b && is_whitespace(rawQuery[e - 1])) e--;
        ustr q = to_lower_inv(default_normalizer().normalize(rawQuery.substr(b, e - b)));
        if (ix.syn.has()) q = ix.syn.canonicalize(q);                      // SearchEngine.cs:276-286
        return search_with(*s1, q, qp);
    }
    // q = trimmed, normalised, lower-cased query text; st = per-thread Stage-1 scratch (fuzzy LRU, upperBounds)
    SearchOutput search_with(Stage1& st, const ustr& q, const QueryParams& qp) {
        SearchOutput out;
        if (!indexed) return out;
        bool allws = true; for (u16 c : q) if (!is_whitespace(c)) { allws = false; break; }
        if (allws) return out;
        std::vector<ScoreEntry> res = execute(q, qp, out, st);
        out.totalCandidates = (int)res.size();
        if ((int)res.size() > qp.maxResults) res.resize(qp.maxResults);
        out.records = res;
        return out;
    }

    std::vector<ScoreEntry> execute(const ustr& searchTextIn, const QueryParams& qp, SearchOutput& out, Stage1& st) {
        ustr searchText = default_normalizer().normalize(searchTextIn);
        int n = ix.cfg.ngram;
        bool canUse, mixed; ustr longWords;
        analyze(searchText, n, canUse, mixed, longWords);
        if (!canUse) { out.unsupported = true; return {}; }
        ustr tfidfQuery = mixed ? longWords : searchText;
        { bool ws = true; for (u16 c : tfidfQuery) if (!is_whitespace(c)) { ws = false; break; } if (ws) tfidfQuery = searchText; }
        std::vector<ScoreEntry> stage1 = consolidate(st.search_with_maxscore(tfidfQuery, qp.coverageDepth));
        out.stage1 = stage1;

        bool isShort = !searchText.empty() && searchText.size() <= 3;
        if (isShort) for (u16 c : searchText) if (default_delims().is(c)) { isShort = false; break; }
        if (isShort && (int)stage1.size() >= qp.maxResults && qp.maxResults < std::numeric_limits<int>::max()) {
            if ((int)stage1.size() > qp.maxResults) stage1.resize(qp.maxResults);
            return stage1;
        }
        int shortCount = 0; bool shortKnown = false;
        if (isShort) { auto* ds = ix.prefix_docset(searchText); shortCount = ds ? (int)ds->size() : 0; shortKnown = true; }
        bool allowShortCov = isShort && shortKnown && shortCount > 0 && shortCount <= 500;
        bool skipCov = isShort && shortKnown && shortCount > 500;
        if (!ix.cfg.enableCoverage || !qp.enableCoverage || (!canUse && !allowShortCov) || skipCov) return stage1;

        out.usedCoverage = true;
        std::vector<ScoreEntry> cr = coverage_stage(searchText, qp, stage1, out);
        if (cr.empty() && !stage1.empty()) return stage1;
        return cr;
    }

    std::vector<ScoreEntry> coverage_stage(const ustr& searchText, const QueryParams& qp, std::vector<ScoreEntry> top, SearchOutput& out) {
        const CoverageSetup& cs = cov.setup;
        int depth = qp.coverageDepth;
        if ((int)top.size() > depth) top.resize(depth);
        std::vector<int32_t> wmIds;
        StageClock clk_;
        if (ix.cfg.wordMatcher) wm.execute(searchText, cs.CoverPrefixSuffix, wmIds);
        clk_.lap(3);
        struct CovLap { StageClock& c; ~CovLap() { c.lap(4); } } covLap_{clk_};      // the rest of the coverage stage
        // BuildDocumentKeyIndex: insertion-ordered unique keys: top candidates first, then WM ids ascending
        std::unordered_map<int64_t, int> keyToIndex;
        int next = 0;
        for (auto& c : top) if (keyToIndex.emplace(c.key, next).second) next++;
        for (int id : wmIds) if (!ix.is_deleted(id) && keyToIndex.emplace(ix.docKey[id], next).second) next++;     // SearchPipeline.cs:532-537
        int nDocs = next;
        uint8_t lcsRow[2] = {0, 0}, hitsRow[2] = {0, 0};   // only docIndex < Height(=2) is ever touched (quirk Q7)
        TopKHeap finalScores(depth);
        int maxWordHits = 0;
        QueryContext ctx = cov.prepare_query(searchText);
        std::unordered_set<int> tfidfIds;
        for (auto& c : top) { auto it = ix.keyToFirstId.find(c.key); if (it != ix.keyToFirstId.end()) tfidfIds.insert(it->second); }
        std::vector<int> overlap, uniq;
        for (int id : wmIds) (tfidfIds.count(id) ? overlap : uniq).push_back(id);
        int wmLimit = std::max(0, depth - (int)overlap.size());

        auto process = [&](int internalId, float baseScore) {
            if (ix.is_deleted(internalId)) return;                 // SearchPipeline.cs:463-465 (a deleted WordMatcher-only id still counts against wmLimit)
            auto kit = keyToIndex.find(ix.docKey[internalId]);
            if (kit == keyToIndex.end()) return;
            int docIndex = kit->second;
            ustr docText = default_normalizer().normalize(ix.raw_text(internalId));
            if (ix.syn.has()) docText = ix.syn.canonicalize(docText);       // SearchPipeline.cs:482-489
            int lcs = 0;
            if (docIndex < 2 && nDocs > 0) {
                lcs = lcsRow[docIndex];
                if (lcs == 0) {
                    int tol = 0;
                    if ((int)ctx.query.size() >= cs.CoverageQLimitForErrorTolerance) tol = (int)((double)ctx.query.size() * cs.CoverageLcsErrorToleranceRelativeq);
                    lcs = lcs_metric(to_lower_inv(ctx.query), to_lower_inv(docText), tol);
                    lcsRow[docIndex] = (uint8_t)std::min(lcs, 255);
                }
            }
            CoverageFeatures f = cov.calculate_features(ctx, docText, (double)lcs);
            auto sc = fusion_calculate(ctx.query, docText, f, baseScore);
            if (docIndex < 2 && hitsRow[docIndex] == 0) hitsRow[docIndex] = (uint8_t)std::min(f.WordHits, 255);
            maxWordHits = std::max(maxWordHits, f.WordHits);
            finalScores.add(ScoreEntry{sc.first, ix.docKey[internalId], sc.second});
            if (keepTrace) out.trace.push_back(CandidateTrace{internalId, baseScore, sc.first, sc.second, f, lcs});
        };
        for (int id : overlap) process(id, 0.f);
        int done = 0;
        for (int id : uniq) { if (done >= wmLimit) break; process(id, 0.f); done++; }
        for (auto& c : top) {
            auto it = ix.keyToFirstId.find(c.key);
            if (it == ix.keyToFirstId.end() || ix.is_deleted(it->second)) continue;      // SearchPipeline.cs:404-406
            float maxT = !top.empty() ? top[0].score : 1.f;
            float norm = maxT > 0 ? c.score / maxT : 0.f;
            process(it->second, norm);
        }
        if (maxWordHits == 0 && wmIds.empty()) return {};
        std::vector<ScoreEntry> fin = consolidate(finalScores.get_topk());
        int truncIdx = -1;
        if (cs.Truncate && !fin.empty()) {
            int minHits = std::max(cs.CoverageMinWordHitsAbs, maxWordHits - cs.CoverageMinWordHitsRelative);
            for (int i = (int)fin.size() - 1; i >= 0; i--) {
                auto it = keyToIndex.find(fin[i].key);
                if (it == keyToIndex.end()) continue;
                int di = it->second;
                if (di >= nDocs) continue;
                uint8_t wh = di < 2 ? hitsRow[di] : 0, lc = di < 2 ? lcsRow[di] : 0;
                if (wh >= minHits || lc > 0 || fin[i].score >= (float)cs.TruncationScore) { truncIdx = i; break; }
            }
        }
        int resultCount = (truncIdx == -1 || !cs.Truncate) ? qp.maxResults : std::min(std::max(0, truncIdx) + 1, qp.maxResults);
        if ((int)fin.size() > resultCount) fin.resize(resultCount);
        return fin;
    }
};

} // namespace orc
