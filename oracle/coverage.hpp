// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's Stage 2: lexical coverage features + fusion score.
//
// Follows (paths relative to /root/reference/src/Infidex):
//   Metrics/LevenshteinDistance.cs:181-257 (Calculate), :281-341 (CalculateDamerau)
//   Metrics/StringMetrics.cs:12-36         (Lcs = containment or common-prefix + tolerance)
//   Coverage/CoverageTokenizer.cs:7-108    (TokenizeToSpan, Deduplicate*)
//   Coverage/CoverageEngine.cs:61-126      (PrepareQuery), :222-382 (CalculateCoverageInternal), :388-427 (ComputeTermIdf)
//   Coverage/WholeWordMatcher.cs:5-68, JoinedWordMatcher.cs:5-135, PrefixSuffixMatcher.cs:8-214, FuzzyWordMatcher.cs:14-144
//   Coverage/CoverageScorer.cs:5-198       (CalculateFinalScore)
//   Coverage/FusionSignalComputer.cs:53-568
//   Scoring/FusionScorer.cs:19-236, 313-396
//   Coverage/CoverageSetup.cs:6-103        (defaults)
#pragma once
#include "index.hpp"

namespace orc {

struct CoverageSetup {
    int MinWordSize = 2, LevenshteinMaxWordSize = 20, NumTypos = 2, MinLengthOneTypo = 3, MinLengthTwoTypos = 7;
    int CoverageMinWordHitsAbs = 1, CoverageMinWordHitsRelative = 0, CoverageQLimitForErrorTolerance = 5;
    double CoverageLcsErrorToleranceRelativeq = 0.2;
    bool CoverWholeQuery = true, CoverWholeWords = true, CoverFuzzyWords = true, CoverJoinedWords = true, CoverPrefixSuffix = true;
    bool Truncate = true;
    uint8_t TruncationScore = 254;
};

// ---- LevenshteinDistance -------------------------------------------------------------------
inline int lev_calculate(uview pattern, uview text, int maxErrors, bool ignoreCase) {
    if (pattern.empty()) return (int)text.size();
    if (text.empty()) return (int)pattern.size();
    if (pattern.size() > text.size()) std::swap(pattern, text);
    int m = (int)pattern.size(), n = (int)text.size();
    std::vector<int> costs(m + 1);
    for (int i = 0; i <= m; i++) costs[i] = i;
    for (int j = 0; j < n; j++) {
        u16 tv = text[j]; if (ignoreCase) tv = to_upper_inv(tv);
        int diagonal = costs[0];
        costs[0] = j + 1;
        int minCost = costs[0];
        for (int i = 0; i < m; i++) {
            int left = costs[i + 1], up = costs[i];
            u16 pv = pattern[i]; if (ignoreCase) pv = to_upper_inv(pv);
            int cost;
            if (tv == pv) cost = diagonal;
            else { cost = up + 1; if (left + 1 < cost) cost = left + 1; if (diagonal + 1 < cost) cost = diagonal + 1; }
            diagonal = left;
            costs[i + 1] = cost;
            if (cost < minCost) minCost = cost;
        }
        if (minCost > maxErrors) return maxErrors + 1;
    }
    return costs[m];
}
inline int lev_damerau(uview source, uview target, int maxDistance, bool ignoreCase) {
    int lenDiff = std::abs((int)source.size() - (int)target.size());
    if (lenDiff > maxDistance) return maxDistance + 1;
    int dist = lev_calculate(source, target, maxDistance + 1, ignoreCase);
    if (dist <= maxDistance) return dist;
    if (dist <= maxDistance + 1) {
        int len = (int)source.size();
        for (int i = 0; i < len - 1; i++) {
            if (i >= (int)target.size()) break;
            u16 s1 = ignoreCase ? to_lower_inv(source[i]) : source[i];
            u16 t1 = ignoreCase ? to_lower_inv(target[i]) : target[i];
            if (s1 != t1) {
                if (i + 1 >= (int)target.size()) break;
                u16 s2 = ignoreCase ? to_lower_inv(source[i + 1]) : source[i + 1];
                u16 t2 = ignoreCase ? to_lower_inv(target[i + 1]) : target[i + 1];
                if (s1 == t2 && s2 == t1) {
                    int budget = maxDistance - 1;
                    if (budget < 0) return maxDistance + 1;
                    uview sRest = (i + 2 < len) ? source.substr(i + 2) : uview();
                    uview tRest = (i + 2 < (int)target.size()) ? target.substr(i + 2) : uview();
                    int rest = lev_calculate(sRest, tRest, budget, ignoreCase);
                    if (rest <= budget) return 1 + rest;
                }
                break;
            }
        }
    }
    return dist;
}
// StringMetrics.Lcs (ordinal, inputs already lower-cased by SegmentProcessor.CalculateLcs)
inline int lcs_metric(uview q, uview r, int tol) {
    if (q.empty() || r.empty()) return 0;
    if (q == r) return (int)q.size();
    if (r.find(q) != uview::npos) return (int)q.size();
    int prefix = 0, len = (int)std::min(q.size(), r.size());
    for (int i = 0; i < len; i++) { if (q[i] != r[i]) break; prefix++; }
    return prefix == 0 ? 0 : std::min(prefix + tol, len);
}

// ---- tokens -----------------------------------------------------------------------------------
struct Tok { int off, len, pos; };
inline int cov_tokenize(uview text, std::vector<Tok>& out, int minWordSize, int maxTokens) {
    const Delims& D = default_delims();
    out.clear();
    int n = (int)text.size(), i = 0;
    while (i < n) {
        while (i < n && D.is(text[i])) i++;
        if (i >= n) break;
        int st = i;
        while (i < n && !D.is(text[i])) i++;
        int L = i - st;
        if (L >= minWordSize && (int)out.size() < maxTokens) out.push_back({st, L, st});
    }
    return (int)out.size();
}
inline void dedup_tokens(const std::vector<Tok>& raw, std::vector<Tok>& uniq, uview text) {
    uniq.clear();
    for (auto& c : raw) {
        bool dup = false;
        uview cs = text.substr(c.off, c.len);
        for (auto& e : uniq) if (e.len == c.len && eq_ic(text.substr(e.off, e.len), cs)) { dup = true; break; }
        if (!dup) uniq.push_back(c);
    }
}

struct FusionSignals {
    int UnfilteredQueryTokenCount = 0;
    bool LexicalPrefixLast = false, AllPrecedingExact = false, IsPerfectDocLexical = false, HasStemEvidence = false, HasAnchorStem = false;
    uint8_t TrailingMatchDensity = 0, SingleTermLexicalSim = 0;
    int SingleCharLastTokenBoost = 0;
};

struct CoverageFeatures {
    uint8_t CoverageScore = 0;
    int TermsCount = 0, TermsWithAnyMatch = 0, TermsFullyMatched = 0, TermsStrictMatched = 0, TermsPrefixMatched = 0;
    int FirstMatchIndex = -1;
    float SumCi = 0.f;
    int WordHits = 0, DocTokenCount = 0, LongestPrefixRun = 0, SuffixPrefixRun = 0, PhraseSpan = 0, PrecedingStrictCount = 0;
    bool LastTokenHasPrefix = false;
    float LastTermCi = 0.f, WeightedCoverage = 0.f;
    bool LastTermIsTypeAhead = false;
    float IdfCoverage = 0.f, TotalIdf = 0.f, MissingIdf = 0.f;
    bool hasTermArrays = false;
    std::vector<float> TermIdf, TermCi;     // word-level IDF per token / per-token Ci
    FusionSignals Fusion;
};

struct QueryContext {      // CoverageQueryContext
    ustr query;
    std::vector<Tok> tokens;         // deduplicated, MinWordSize filtered
    std::vector<float> termIdf;      // n-gram averaged IDF
    std::vector<int> termMaxChars;
    std::vector<float> wordLevelIdf; // from WordIdfCache (0 when missing)
    bool hasWordLevelIdf = false;
    std::vector<Tok> fusionTokens;   // unfiltered (minWordSize 0)
};

struct CoverageEngine {
    const Index* ix = nullptr;      // corpus statistics (null => fallback IDF = log2(len+1))
    CoverageSetup setup;
    const std::unordered_map<ustr, float, UHash>* fixedWordIdf = nullptr;   // SetWordIdfCache without a corpus (BugReproductionTests.cs:24-31); keys upper-folded

    float compute_term_idf(uview term) const {   // CoverageEngine.cs:388-427
        if (!ix || ix->N == 0) return log2f((float)(term.size() + 1));
        int n = ix->cfg.ngram;
        float sum = 0.f; int cnt = 0;
        if ((int)term.size() >= n)
            for (int i = 0; i + n <= (int)term.size(); i++) {
                int id = ix->get_term(term.substr(i, n));
                if (id >= 0 && ix->termDf[id] > 0) { sum += compute_idf(ix->N, ix->termDf[id]); cnt++; }
            }
        return cnt > 0 ? sum / (float)cnt : log2f((float)(term.size() + 1));
    }
    QueryContext prepare_query(uview query) const {
        QueryContext c; c.query = ustr(query);
        if (query.empty()) return c;
        int maxQ = (int)query.size() / 2 + 1;
        std::vector<Tok> raw;
        if (cov_tokenize(query, raw, setup.MinWordSize, maxQ) == 0) return c;
        dedup_tokens(raw, c.tokens, query);
        int q = (int)c.tokens.size();
        c.termIdf.resize(q); c.termMaxChars.resize(q);
        for (int i = 0; i < q; i++) {
            c.termMaxChars[i] = c.tokens[i].len;
            c.termIdf[i] = (ix && ix->N > 0) ? compute_term_idf(query.substr(c.tokens[i].off, c.tokens[i].len))
                                             : log2f((float)(c.termMaxChars[i] + 1));
        }
        if (ix) {
            c.hasWordLevelIdf = true; c.wordLevelIdf.resize(q);
            for (int i = 0; i < q; i++) { float v; c.wordLevelIdf[i] = ix->word_idf(query.substr(c.tokens[i].off, c.tokens[i].len), v) ? v : 0.f; }
        }
        else if (fixedWordIdf) {
            c.hasWordLevelIdf = true; c.wordLevelIdf.resize(q);
            for (int i = 0; i < q; i++) {
                ustr k(query.substr(c.tokens[i].off, c.tokens[i].len)); for (auto& ch : k) ch = to_upper_inv(ch);
                auto it = fixedWordIdf->find(k);
                c.wordLevelIdf[i] = it == fixedWordIdf->end() ? 0.f : it->second;
            }
        }
        cov_tokenize(query, c.fusionTokens, 0, (int)query.size() / 2 + 1);
        return c;
    }

    // ---- FusionSignalComputer ----
    static FusionSignals compute_signals(uview Q, uview D, const std::vector<Tok>& qt, const std::vector<Tok>& dt, int minStemLength) {
        FusionSignals s;
        int qCount = (int)qt.size(), dCount = (int)dt.size();
        s.UnfilteredQueryTokenCount = qCount;
        if (qCount == 0 || dCount == 0) return s;
        auto qs = [&](int i) { return Q.substr(qt[i].off, qt[i].len); };
        auto ds = [&](int i) { return D.substr(dt[i].off, dt[i].len); };
        // 1. CheckPrefixLastMatch
        {
            bool pl = false, ape = false;
            if (qCount == 1) {
                uview q = qs(0);
                for (int i = 0; i < dCount; i++) if (starts_with_ic(ds(i), q)) { pl = true; ape = eq_ic(ds(i), q); break; }
            } else {
                bool allExact = true;
                for (int i = 0; i < qCount - 1; i++) {
                    uview q = qs(i);
                    if (q.empty()) continue;
                    bool found = false;
                    for (int j = 0; j < dCount; j++) if (eq_ic(ds(j), q)) { found = true; break; }
                    if (!found) { allExact = false; break; }
                }
                if (allExact) {
                    uview last = qs(qCount - 1);
                    if (last.empty()) { pl = true; ape = true; }
                    else for (int i = 0; i < dCount; i++) if (starts_with_ic(ds(i), last)) { pl = true; ape = true; break; }
                }
            }
            s.LexicalPrefixLast = pl; s.AllPrecedingExact = ape;
        }
        // 2. ComputePerfectDoc
        {
            bool perfect = true;
            for (int j = 0; j < dCount && perfect; j++) {
                bool explained = false;
                for (int i = 0; i < qCount; i++) if (starts_with_ic(ds(j), qs(i)) || starts_with_ic(qs(i), ds(j))) { explained = true; break; }
                if (!explained) perfect = false;
            }
            s.IsPerfectDocLexical = perfect;
        }
        // 3. CheckStemEvidence
        if (qCount >= 2) {
            int unmatched = 0, evidence = 0;
            for (int qi = 0; qi < qCount; qi++) {
                uview q = qs(qi);
                if ((int)q.size() < minStemLength) continue;
                bool wordMatch = false;
                for (int di = 0; di < dCount; di++) { uview d = ds(di); if (d.empty()) continue; if (eq_ic(d, q) || starts_with_ic(d, q)) { wordMatch = true; break; } }
                if (wordMatch) continue;
                unmatched++;
                for (int di = 0; di < dCount; di++) {
                    uview d = ds(di);
                    if ((int)d.size() < minStemLength) continue;
                    if (starts_with_ic(q, d)) { evidence++; break; }
                    int maxCheck = (int)std::min(q.size(), d.size());
                    if (maxCheck >= minStemLength) {
                        int pl = 0;
                        for (int i = 0; i < maxCheck; i++) { if (to_lower_inv(q[i]) == to_lower_inv(d[i])) pl++; else break; }
                        if (pl >= minStemLength) { evidence++; break; }
                    }
                }
            }
            s.HasStemEvidence = unmatched > 0 && evidence == unmatched;
        }
        // 4. HasAnchorStem — DocumentMetadataCache is null on a freshly indexed engine (SearchEngine.cs:176-185
        //    wires it BEFORE BuildOptimizedIndexes creates it), so the "no precomputed metadata" loop runs.
        if (qt[0].len >= 3) {
            uview stem = qs(0).substr(0, 3);
            for (int i = 0; i < dCount; i++) { uview d = ds(i); if (d.size() >= stem.size() && starts_with_ic(d, stem)) { s.HasAnchorStem = true; break; } }
        }
        // 5. TrailingMatchDensity
        if (qCount >= 2) {
            const Tok& lt = qt[qCount - 1];
            if (lt.len >= 1 && lt.len <= 2) {
                uview lq = qs(qCount - 1);
                int matchable = 0;
                for (int i = 0; i < dCount; i++) { uview d = ds(i); if (starts_with_ic(d, lq) || (d.size() > lq.size() && contains_ic(d, lq))) matchable++; }
                if (matchable > 0) {
                    float dens = (float)matchable / (float)dCount;
                    float v = dens * 255.f; if (v < 0.f) v = 0.f; if (v > 255.f) v = 255.f;
                    s.TrailingMatchDensity = (uint8_t)v;
                }
            }
        }
        // 6. SingleTermLexicalSim
        if (qCount == 1) {
            float sim = single_term_sim(qs(0), D, dt);
            float v = sim * 255.f; if (v < 0.f) v = 0.f; if (v > 255.f) v = 255.f;
            s.SingleTermLexicalSim = (uint8_t)v;
        }
        // 7. SingleCharLastTokenBoost
        if (qCount >= 2) s.SingleCharLastTokenBoost = single_char_last(Q, D, qt, dt);
        return s;
    }
    static int single_char_last(uview Q, uview D, const std::vector<Tok>& qt, const std::vector<Tok>& dt) {
        int qCount = (int)qt.size(), dCount = (int)dt.size();
        const Tok& lastQ = qt[qCount - 1];
        if (lastQ.len != 1) return 0;
        u16 target = to_lower_inv(Q[lastQ.off]);
        if (!is_letter(target)) return 0;
        int dIndex = 0, firstMatch = -1;
        for (int i = 0; i < qCount - 1; i++) {
            uview qTerm = Q.substr(qt[i].off, qt[i].len);
            bool found = false;
            while (dIndex < dCount) {
                uview dTerm = D.substr(dt[dIndex].off, dt[dIndex].len);
                if (index_of_ic(dTerm, qTerm) >= 0) { found = true; if (firstMatch == -1) firstMatch = dIndex; break; }
                dIndex++;
            }
            if (!found) return 0;
        }
        if (dIndex + 1 < dCount) {
            const Tok& nx = dt[dIndex + 1];
            uview nextTerm = D.substr(nx.off, nx.len);
            if (!nextTerm.empty() && to_lower_inv(nextTerm[0]) == target) {
                int endOfLast = dt[dIndex].off + dt[dIndex].len;
                bool broken = false;
                for (int p = endOfLast; p < nx.off; p++) if (!is_whitespace(D[p])) { broken = true; break; }
                if (!broken) {
                    int boost = 8 + std::max(0, 16 - firstMatch);
                    if (nextTerm.size() == 1) boost += 4;
                    return boost;
                }
            }
        }
        return 0;
    }
    static float single_term_sim(uview query, uview D, const std::vector<Tok>& dt) {
        int qLen = (int)query.size();
        if (qLen < 3) return 0.f;
        ustr ql = to_lower_inv(query);
        float best = 0.f;
        for (auto& t : dt) {
            if (t.len < 2) continue;
            ustr tl = to_lower_inv(D.substr(t.off, t.len));
            size_t idx = ql.find(tl);
            if (idx != ustr::npos) {
                float lenFrac = (float)tl.size() / (float)qLen;
                float posF = 1.f - (float)idx / (float)qLen;
                float sc = lenFrac * posF;
                if (sc > best) best = sc;
                continue;
            }
            int maxK = std::min(qLen, (int)tl.size()), bestK = 0;
            for (int len = maxK; len >= 2; len--) if (uview(ql).substr(qLen - len) == uview(tl).substr(0, len)) { bestK = len; break; }
            float ps = bestK > 0 ? (float)bestK / (float)qLen : 0.f;
            float fz = 0.f;
            if (tl.size() <= 32) {
                int dist = lev_damerau(ql, tl, 2, false);
                if (dist <= 2) fz = (float)(qLen - dist) / (float)qLen;
            }
            float comb = std::max(ps, fz);
            if (comb > best) best = comb;
        }
        if (qLen >= 6) {
            int segLen = std::min(6, qLen / 2);
            uview pf = uview(ql).substr(0, segLen), sf = uview(ql).substr(qLen - segLen, segLen);
            int pi = -1, si = -1;
            for (int i = 0; i < (int)dt.size(); i++) {
                if (dt[i].len < 3) continue;
                ustr tl = to_lower_inv(D.substr(dt[i].off, dt[i].len));
                if (pi == -1 && (starts_with(tl, pf) || starts_with(pf, tl))) pi = i;
                if (si == -1 && (ends_with(tl, sf) || ends_with(sf, tl))) si = i;
                if (pi != -1 && si != -1) break;
            }
            if (pi != -1 && si != -1 && pi != si) {
                float two = std::min(1.f, (float)(pf.size() + sf.size()) / (float)qLen);
                if (two > best) best = two;
            }
        }
        return best;
    }

    // ---- CalculateFeatures -----
    CoverageFeatures calculate_features(const QueryContext& ctx, uview docText, double lcsSum) const {
        CoverageFeatures F;
        int qCount = (int)ctx.tokens.size();
        if (qCount == 0) { F.FirstMatchIndex = -1; return F; }   // CoverageResult(0,0,-1,0); fusionSignals default
        uview Q = ctx.query, D = docText;
        std::vector<Tok> rawDoc, dtok;
        int dCountRaw = cov_tokenize(D, rawDoc, setup.MinWordSize, (int)D.size() / 2 + 1);
        F.DocTokenCount = dCountRaw;
        dedup_tokens(rawDoc, dtok, D);
        int dCount = (int)dtok.size();
        std::vector<char> qActive(qCount, 1), dActive(dCount, 1), hasWhole(qCount, 0), hasJoined(qCount, 0), hasPrefix(qCount, 0);
        std::vector<float> matched(qCount, 0.f);
        std::vector<int> firstPos(qCount, -1);
        const std::vector<int>& maxChars = ctx.termMaxChars;
        int wordHits = 0; double numWhole = 0, numJoined = 0, numFuzzy = 0, numPS = 0; uint8_t penalty = 0;
        auto qs = [&](int i) { return Q.substr(ctx.tokens[i].off, ctx.tokens[i].len); };
        auto ds = [&](int j) { return D.substr(dtok[j].off, dtok[j].len); };
        auto setpos = [&](int i, int pos) { if (firstPos[i] == -1 || pos < firstPos[i]) firstPos[i] = pos; };

        if (setup.CoverWholeWords) {   // WholeWordMatcher.Match
            int pInc = qCount > 1 ? 1 : 0;
            for (int i = 0; i < qCount; i++) {
                int mi = -1;
                for (int j = 0; j < dCount; j++) if (dActive[j] && dtok[j].len == ctx.tokens[i].len && eq_ic(qs(i), ds(j))) { mi = j; break; }
                if (mi != -1) {
                    wordHits++; numWhole += ctx.tokens[i].len;
                    matched[i] += (float)ctx.tokens[i].len; hasWhole[i] = 1; hasPrefix[i] = 1;
                    setpos(i, dtok[mi].pos);
                    if (dCount > i) { if (dtok[i].len != ctx.tokens[i].len || !eq_ic(qs(i), ds(i))) penalty++; }
                    else penalty++;
                    if (i < qCount - 1) numWhole += pInc;
                    qActive[i] = 0; dActive[mi] = 0;
                }
            }
        }
        if (setup.CoverJoinedWords && qCount > 0) {   // JoinedWordMatcher.Match
            for (int i = 0; i < qCount - 1; i++) {
                if (!qActive[i] || !qActive[i + 1]) continue;
                int next = -1;
                for (int k = i + 1; k < qCount; k++) if (qActive[k]) { next = k; break; }
                if (next == -1) break;
                int jl = ctx.tokens[i].len + ctx.tokens[next].len;
                int mi = -1;
                for (int j = 0; j < dCount; j++) if (dActive[j] && dtok[j].len == jl && starts_with_ic(ds(j), qs(i)) && ends_with_ic(ds(j), qs(next))) { mi = j; break; }
                if (mi != -1) {
                    numJoined += jl; wordHits += 2;
                    matched[i] += (float)ctx.tokens[i].len; hasJoined[i] = 1; hasPrefix[i] = 1;
                    int pos = dtok[mi].pos; setpos(i, pos);
                    matched[next] += (float)ctx.tokens[next].len; hasJoined[next] = 1; setpos(next, pos);
                    qActive[i] = 0; qActive[next] = 0; dActive[mi] = 0;
                }
            }
            for (int i = 0; i < dCount - 1; i++) {
                if (!dActive[i]) continue;
                int next = -1;
                for (int k = i + 1; k < dCount; k++) if (dActive[k]) { next = k; break; }
                if (next == -1) break;
                int jl = dtok[i].len + dtok[next].len;
                int mi = -1;
                for (int j = 0; j < qCount; j++) if (qActive[j] && ctx.tokens[j].len == jl && starts_with_ic(qs(j), ds(i)) && ends_with_ic(qs(j), ds(next))) { mi = j; break; }
                if (mi != -1) {
                    numJoined += jl; wordHits += 1;
                    matched[mi] += (float)jl; hasJoined[mi] = 1; hasPrefix[mi] = 1;
                    setpos(mi, dtok[i].pos);
                    qActive[mi] = 0; dActive[i] = 0; dActive[next] = 0;
                }
            }
        }
        if (setup.CoverPrefixSuffix && qCount > 0) {   // PrefixSuffixMatcher.Match
            std::vector<int> qi, di;
            for (int i = 0; i < qCount; i++) if (qActive[i]) qi.push_back(i);
            for (int j = 0; j < dCount; j++) if (dActive[j]) di.push_back(j);
            auto sort_len_desc = [](std::vector<int>& idx, auto lenOf) {
                for (size_t i = 1; i < idx.size(); i++) {
                    int cur = idx[i], cl = lenOf(cur); int j = (int)i - 1;
                    while (j >= 0 && lenOf(idx[j]) < cl) { idx[j + 1] = idx[j]; j--; }
                    idx[j + 1] = cur;
                }
            };
            sort_len_desc(qi, [&](int i) { return ctx.tokens[i].len; });
            sort_len_desc(di, [&](int j) { return dtok[j].len; });
            for (int i : qi) {     // MatchExact
                if (!qActive[i]) continue;
                int ql = ctx.tokens[i].len; uview qt = qs(i);
                for (int j : di) {
                    if (!dActive[j]) continue;
                    int dl = dtok[j].len;
                    if (ql == dl) continue;
                    uview dtx = ds(j);
                    bool isMatch = false, isPrefix = false; double ms = 0;
                    if (ql < dl) {
                        if (starts_with_ic(dtx, qt)) { ms = ql; isMatch = true; isPrefix = true; }
                        else if (ends_with_ic(dtx, qt)) { ms = std::max(1, ql / 2); isMatch = true; }
                        else if (ql >= 4 && contains_ic(dtx, qt)) { ms = ql * 0.6; isMatch = true; }
                    } else {
                        if (ends_with_ic(qt, dtx)) { ms = dl; isMatch = true; }
                    }
                    if (isMatch) {
                        numPS += ms; wordHits++;
                        matched[i] += (float)ms; if (isPrefix) hasPrefix[i] = 1;
                        setpos(i, dtok[j].pos);
                        qActive[i] = 0; dActive[j] = 0;
                        break;
                    }
                }
            }
            for (int i : qi) {     // MatchFuzzyPrefix
                if (!qActive[i]) continue;
                int ql = ctx.tokens[i].len; uview qt = qs(i);
                if (!(ql >= 4 || (i == qCount - 1 && ql >= 2))) continue;
                for (int j : di) {
                    if (!dActive[j]) continue;
                    int dl = dtok[j].len;
                    if (ql >= dl) continue;
                    uview dtx = ds(j);
                    bool isMatch = false; double ms = 0;
                    int dist = lev_damerau(qt, dtx.substr(0, ql), 1, true);
                    if (dist <= 1) { ms = ql - dist; if (ms < 0.1) ms = 0.1; isMatch = true; }
                    else if (dl > ql) {
                        dist = lev_damerau(qt, dtx.substr(0, ql + 1), 1, true);
                        if (dist <= 1) { ms = ql - dist; if (ms < 0.1) ms = 0.1; isMatch = true; }
                        else if (ql > 1) {
                            dist = lev_damerau(qt, dtx.substr(0, ql - 1), 1, true);
                            if (dist <= 1) { ms = ql - 1 - dist; if (ms < 0.1) ms = 0.1; isMatch = true; }
                        }
                    }
                    if (isMatch) {
                        numPS += ms; wordHits++;
                        matched[i] += (float)ms;
                        setpos(i, dtok[j].pos);
                        qActive[i] = 0; dActive[j] = 0;
                        break;
                    }
                }
            }
        }
        bool allFull = true;
        for (int i = 0; i < qCount; i++) if (maxChars[i] > 0 && matched[i] < (float)maxChars[i]) { allFull = false; break; }
        if (setup.CoverFuzzyWords && qCount > 0 && !allFull) {   // FuzzyWordMatcher.Match
            int maxQL = 0;
            for (int i = 0; i < qCount; i++) if (qActive[i] && ctx.tokens[i].len > maxQL) maxQL = ctx.tokens[i].len;
            if (maxQL != 0) {
                int maxEd = maxQL >= setup.MinLengthTwoTypos ? 2 : (maxQL >= setup.MinLengthOneTypo ? 1 : 0);
                if (maxQL == 2 && maxEd == 0 && setup.NumTypos >= 1) maxEd = 1;
                if (maxEd > setup.NumTypos) maxEd = setup.NumTypos;
                for (int ed = 1; ed <= maxEd; ed++) {
                    bool anyQ = false; for (int i = 0; i < qCount; i++) if (qActive[i]) anyQ = true;
                    if (!anyQ) break;
                    for (int i = 0; i < qCount; i++) {
                        if (!qActive[i]) continue;
                        int ql = ctx.tokens[i].len;
                        if (ql < setup.MinWordSize) continue;
                        int tme = ql >= setup.MinLengthTwoTypos ? 2 : (ql >= setup.MinLengthOneTypo ? 1 : 0);
                        bool special = false;
                        if (ql == 2 && tme == 0 && setup.NumTypos >= 1) { tme = 1; special = true; }
                        if (tme > setup.NumTypos) tme = setup.NumTypos;
                        if (ed > tme) continue;
                        if (special && ed != 1) continue;
                        int minLen = std::max(setup.MinWordSize, ql - ed);
                        int maxLen = std::min(setup.LevenshteinMaxWordSize, ql + ed);
                        if (maxLen > 63) maxLen = 63;
                        uview qt = qs(i);
                        for (int j = 0; j < dCount; j++) {
                            if (!dActive[j]) continue;
                            int dl = dtok[j].len;
                            if (dl > maxLen || dl < minLen) continue;
                            uview dtx = ds(j);
                            if (special && (dtx.empty() || to_lower_inv(dtx[0]) != to_lower_inv(qt[0]))) continue;
                            int dist = lev_damerau(qt, dtx, ed, true);
                            if (dist <= ed) {
                                wordHits++; numFuzzy += (ql - dist);
                                matched[i] += (float)(ql - dist);
                                setpos(i, dtok[j].pos);
                                qActive[i] = 0; dActive[j] = 0;
                                break;
                            }
                        }
                    }
                }
            }
        }
        F.WordHits = wordHits;
        // ---- CoverageScorer.CalculateFinalScore ----
        int queryLen = (int)Q.size();
        if (!setup.CoverWholeQuery) lcsSum = 0.0;
        double num11 = numJoined + numWhole + numFuzzy + numPS - (double)penalty;
        if (num11 == 0.0 && lcsSum > 2.0) num11 = lcsSum - 2.0;
        {
            double v = std::min(num11 / (double)queryLen * 255.0, 255.0);
            // C# (byte)double: truncation; negative values are undefined-ish in unchecked context -> emulate x64 cvttsd2si & 0xFF
            F.CoverageScore = (uint8_t)(int64_t)v;
        }
        float sumCi = 0.f, wsum = 0.f, totalW = 0.f, idfW = 0.f, totalIdf = 0.f, missingIdf = 0.f, lastCi = 0.f, lastIdf = 0.f;
        int firstMatchIndex = -1, minPos = std::numeric_limits<int>::max(), maxPos = -1;
        bool haveCi = ctx.hasWordLevelIdf && qCount > 0;
        if (haveCi) F.TermCi.assign(qCount, 0.f);
        for (int i = 0; i < qCount; i++) {
            if (maxChars[i] <= 0) continue;
            float ci = std::min(1.0f, matched[i] / (float)maxChars[i]);
            sumCi += ci;
            if (haveCi) F.TermCi[i] = ci;
            if (ci > 0) F.TermsWithAnyMatch++;
            float tw = (float)maxChars[i];
            totalW += tw; wsum += ci * tw;
            float idf = ctx.termIdf[i];
            totalIdf += idf; idfW += ci * idf;
            if (ci < 1.0f) missingIdf += (1.0f - ci) * idf;
            if (i == qCount - 1) { lastCi = ci; lastIdf = idf; }
            bool full = matched[i] >= ((float)maxChars[i] - 0.01f);
            if (full) F.TermsFullyMatched++;
            if ((hasWhole[i] || hasJoined[i]) && full) F.TermsStrictMatched++;
            if (hasPrefix[i]) F.TermsPrefixMatched++;
            if (firstPos[i] >= 0) {
                if (firstMatchIndex == -1 || firstPos[i] < firstMatchIndex) firstMatchIndex = firstPos[i];
                if (firstPos[i] < minPos) minPos = firstPos[i];
                if (firstPos[i] > maxPos) maxPos = firstPos[i];
            }
        }
        F.WeightedCoverage = totalW > 0.f ? wsum / totalW : 0.f;
        F.IdfCoverage = totalIdf > 0.f ? idfW / totalIdf : 0.f;
        if (qCount > 0 && totalIdf > 0.f) { float share = lastIdf / totalIdf; float thr = 1.f / (float)(qCount + 1); F.LastTermIsTypeAhead = share <= thr; }
        if (qCount == 1 && queryLen > 0 && lcsSum > 0.0) { float ciL = (float)std::min(1.0, lcsSum / (double)queryLen); if (ciL > sumCi) sumCi = ciL; }
        int run = 0;
        for (int i = 0; i < qCount; i++) {
            bool ph = hasPrefix[i] && maxChars[i] > 0 && matched[i] > 0;
            if (ph) { run++; if (run > F.LongestPrefixRun) F.LongestPrefixRun = run; } else run = 0;
        }
        int srun = 0;
        for (int i = qCount - 1; i >= 0; i--) { bool ph = hasPrefix[i] && maxChars[i] > 0 && matched[i] > 0; if (ph) srun++; else break; }
        F.SuffixPrefixRun = srun;
        if (minPos != std::numeric_limits<int>::max() && maxPos >= minPos && F.TermsWithAnyMatch >= 2) F.PhraseSpan = (maxPos - minPos) + 1;
        if (qCount >= 1) {
            int li = qCount - 1;
            F.LastTokenHasPrefix = hasPrefix[li] && matched[li] > 0;
            if (qCount >= 2) for (int i = 0; i < qCount - 1; i++) if ((hasWhole[i] || hasJoined[i]) && matched[i] >= ((float)maxChars[i] - 0.01f)) F.PrecedingStrictCount++;
        }
        F.TermsCount = qCount; F.FirstMatchIndex = firstMatchIndex; F.SumCi = sumCi; F.LastTermCi = lastCi;
        F.TotalIdf = totalIdf; F.MissingIdf = missingIdf;
        if (ctx.hasWordLevelIdf) { F.hasTermArrays = true; F.TermIdf = ctx.wordLevelIdf; }
        // ---- fusion signals on unfiltered tokens ----
        std::vector<Tok> fd;
        cov_tokenize(D, fd, 0, (int)D.size() / 2 + 1);
        F.Fusion = compute_signals(Q, D, ctx.fusionTokens, fd, setup.MinWordSize);
        return F;
    }
};

// ---- FusionScorer.Calculate --------------------------------------------------------------------
inline std::pair<float, uint8_t> fusion_calculate(uview queryText, uview documentText, const CoverageFeatures& f, float bm25Score) {
    const FusionSignals& S = f.Fusion;
    int n = S.UnfilteredQueryTokenCount > 0 ? S.UnfilteredQueryTokenCount : f.TermsCount;
    bool single = n <= 1;
    bool isComplete = f.TermsCount > 0 && f.TermsWithAnyMatch == f.TermsCount;
    bool isClean = f.TermsCount > 0 && f.TermsPrefixMatched == f.TermsCount;
    bool isExact = f.TermsCount > 0 && f.TermsStrictMatched == f.TermsCount;
    bool startsAtBeginning = f.FirstMatchIndex == 0;
    bool lexicalPrefixLast = S.LexicalPrefixLast;
    int precedingTerms = std::max(0, f.TermsCount - 1);
    bool coveragePrefixLast = f.TermsCount >= 1 && f.PrecedingStrictCount == precedingTerms && f.LastTokenHasPrefix;
    bool isPrefixLastStrong = lexicalPrefixLast && coveragePrefixLast;
    bool isPerfectDoc = S.IsPerfectDocLexical;
    int precedence = 0;
    int tier = 0;
    if (!single && f.TermsCount > 0) {
        int matched = f.TermsWithAnyMatch, total = f.TermsCount;
        if (matched >= total) tier = 3; else if (matched == total - 1) tier = 2; else if (matched * 2 >= total) tier = 1; else tier = 0;
    }
    if (!single && tier > 0) precedence |= (tier & 3) << 16;
    bool isExactPrefix = !single && isClean && startsAtBeginning && lexicalPrefixLast && isComplete;
    bool isSubset = !single && f.DocTokenCount > 0 && f.WordHits == f.DocTokenCount;
    if (isExactPrefix) precedence |= (1 << 15);
    if (isSubset) precedence |= (1 << 14);
    float avgIdf = 0.f;
    if (!single && f.TermsCount >= 2) {
        bool hasDominant = false;
        bool arrays = f.hasTermArrays && (int)f.TermIdf.size() == f.TermsCount && (int)f.TermCi.size() == f.TermsCount;
        if (arrays) {
            avgIdf = (f.TotalIdf > 0.f && f.TermsCount > 0) ? f.TotalIdf / (float)f.TermsCount : 0.f;
            for (int c = 0; c < f.TermsCount; c++) {
                float power = f.TermIdf[c] * f.TermCi[c];
                if (f.TermCi[c] <= 0.1f || f.TermIdf[c] <= 0.f || f.TermIdf[c] < avgIdf) continue;
                float other = 0.f;
                for (int i = 0; i < f.TermsCount; i++) if (i != c) other += f.TermIdf[i] * f.TermCi[i];
                if (power >= other) { hasDominant = true; break; }
            }
        }
        bool strongAnchor = S.HasAnchorStem && f.hasTermArrays && f.TermIdf.size() >= 1 && f.TermIdf[0] >= avgIdf;
        if (hasDominant || strongAnchor) precedence |= (1 << 13);
        int unmatched = f.TermsCount - f.TermsWithAnyMatch;
        if (hasDominant && unmatched == 1) precedence |= 8;
    }
    if (single) {
        if (isComplete) precedence |= (1 << 17);
        if (isClean && f.TermsCount > 0) precedence |= (1 << 16);
        int t = 0;
        if (isComplete) { if (startsAtBeginning) { if (isExact) t = 4; else if (isClean) t = 3; } else { if (isExact) t = 2; else if (isClean) t = 1; } }
        precedence |= t << 3;
    } else {
        bool anchorRun = S.HasAnchorStem && f.LongestPrefixRun >= 2;
        int mt = isPrefixLastStrong ? 3 : (lexicalPrefixLast ? 2 : ((isPerfectDoc || anchorRun) ? 1 : 0));
        if (S.UnfilteredQueryTokenCount > f.TermsCount) mt += S.SingleCharLastTokenBoost;
        precedence |= mt;
    }
    float coverageRatio = f.TermsCount > 0 ? (float)f.TermsWithAnyMatch / (float)f.TermsCount : 0.f;
    bool partial = coverageRatio > 0.f && coverageRatio < 1.f;
    if (partial && n >= 2) {
        if (S.HasStemEvidence) precedence |= 8;
        else {
            int unmatched = f.TermsCount - f.TermsWithAnyMatch;
            bool lastMatched = f.LastTokenHasPrefix || (f.TermsCount > 0 && f.TermsWithAnyMatch == f.TermsCount);
            bool canBoost = (lastMatched || !f.LastTermIsTypeAhead) && f.TotalIdf > 0.f;
            if (unmatched == 1 && canBoost) {
                float missRatio = f.MissingIdf / f.TotalIdf;
                float gap = 1.f - coverageRatio;
                if (missRatio < gap) precedence |= 8;
            }
        }
    }
    // ComputeSemanticScore
    float avgCi = f.TermsCount > 0 ? f.SumCi / (float)f.TermsCount : 0.f;
    float semantic;
    if (single) { float ls = (float)S.SingleTermLexicalSim / 255.f; semantic = (avgCi + ls) / 2.f; }
    else if (f.DocTokenCount == 0) semantic = avgCi;
    else {
        int unmatched = f.TermsCount - f.TermsWithAnyMatch;
        bool lastMatched = f.LastTokenHasPrefix || (f.TermsCount > 0 && f.TermsWithAnyMatch == f.TermsCount);
        bool canUseIdf = (lastMatched || !f.LastTermIsTypeAhead) && f.TotalIdf > 0.f;
        bool useIdf = partial && unmatched == 1 && canUseIdf && f.IdfCoverage > coverageRatio;
        float base = useIdf ? f.IdfCoverage : avgCi;
        float density = (float)f.WordHits / (float)f.DocTokenCount;
        semantic = base * density;
        if (f.TermsCount >= 3) {   // ApplyIntentBonus
            int sc = (S.HasAnchorStem ? 1 : 0) + (f.SuffixPrefixRun >= 2 ? 1 : 0);
            if (sc > 0) { float bonus = 0.15f * (float)sc; semantic = std::min(1.f, semantic + bonus); }
        }
        if (f.TermsCount >= 2) {   // ApplyTrailingTermBonus
            float md = (float)S.TrailingMatchDensity / 255.f;
            if (md > 0.f) { float head = 1.f - semantic; semantic += head * md; }
        }
    }
    float gap = 1.f - coverageRatio;
    if (partial && bm25Score >= gap) semantic = coverageRatio * semantic + gap * bm25Score;
    if (semantic < 0.f) semantic = 0.f; if (semantic > 0.999f) semantic = 0.999f;
    uint8_t tie = 0;
    if (n >= 2 && !documentText.empty()) {
        float focus = std::min(1.f, (float)queryText.size() / (float)documentText.size());
        tie = (uint8_t)(focus * 255.f);
    }
    return {(float)precedence + semantic, tie};
}

} // namespace orc
