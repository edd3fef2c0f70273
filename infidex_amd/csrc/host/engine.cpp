// Host engine of the product: the C++ stand-in for the reference's C# SearchEngine (no .NET toolchain in the image).
// Same surface for the hot path — CreateDefault/CreateMinimal, IndexDocuments, Search(Query) — with the two accelerated
// seams delegated to the HIP kernels through the C ABI of include/infidex_hip.h. There is NO CPU scoring path here: without
// a GPU Search fails with INFX_EHIP.
//   SearchEngine.Search / IndexDocuments     src/Infidex/SearchEngine.cs:96-192, 256-319
//   SearchPipeline.Execute                   src/Infidex/Scoring/SearchPipeline.cs:49-206 (gates), :298-447 (coverage stage)
//   ResultProcessor.CalculateTruncationIndex src/Infidex/Scoring/ResultProcessor.cs:146-178
//   TopKHeap / ScoreEntry / ConsolidateSegments  Core/TopKHeap.cs, Core/ScoreEntry.cs:25-36, Scoring/SegmentProcessor.cs:15-37
#include "query.h"
#include "../../../include/infidex_engine.h"
#include <chrono>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>

using namespace infx;

namespace {
thread_local std::string g_eerr;
int32_t efail(int32_t code, const std::string& m) { g_eerr = m; return code; }
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Entry { float score; int64_t key; uint8_t tie; };
inline int cmp_entry(const Entry& a, const Entry& b) {    // ScoreEntry.CompareTo
    if (a.score != b.score) return a.score < b.score ? -1 : 1;
    if (a.tie != b.tie) return a.tie < b.tie ? -1 : 1;
    return b.key < a.key ? -1 : (b.key > a.key ? 1 : 0);
}
}

// One in-flight batch: its own HIP stream + scratch. Several sessions on one engine let the host preparation of one batch
// overlap the GPU stages of another (SearchEngine.Search is callable from many threads concurrently, SearchEngine.cs:258).
struct infx_session {
    infx_engine* e = nullptr;
    infx_stream* stream = nullptr;
    double tPrep1 = 0, tStage1 = 0, tPrep2 = 0, tStage2 = 0, tPost = 0;
    float msAcc = 0, msSel = 0, msCov = 0; uint64_t algBytes = 0; uint64_t s2Candidates = 0, s2TextBytes = 0, streamedBytes = 0, s1Candidates = 0;
    // last-batch introspection for parity tests
    std::vector<QueryPlan> lastPlans;
    std::vector<infx_hit> lastHits; std::vector<uint32_t> lastHitCount; int lastStride = 0;
    std::vector<infx_cov_cand> lastCands; std::vector<infx_cov_out> lastOuts; std::vector<int32_t> lastFeat;
};

struct infx_engine {
    HostIndex ix;
    infx_engine_config cfg{};
    infx_index* dev = nullptr;
    bool indexed = false;
    FuzzyCache fuzzy;
    std::unordered_map<int64_t, int32_t> keyToFirst;
    bool keysAreIds = false;
    int threads = 1;
    infx_session* def = nullptr;      // default session (single-caller API)
};

extern "C" {

const char* infx_engine_last_error(void) { return g_eerr.empty() ? infx_last_error() : g_eerr.c_str(); }

int32_t infx_engine_create(const infx_engine_config* cfg, infx_engine** out) {
    if (!cfg || !out) return efail(INFX_EINVAL, "null argument");
    infx_engine* e = new infx_engine();
    e->cfg = *cfg;
    HostConfig& h = e->ix.cfg;
    h.enableCoverage = cfg->enable_coverage != 0; h.wordMatcher = cfg->word_matcher != 0;
    if (cfg->stop_term_limit > 0) h.stopTermLimit = cfg->stop_term_limit;
    h.maxDepth = cfg->max_depth > 0 ? cfg->max_depth : 500;
    h.threads = cfg->threads;
    e->threads = cfg->threads > 0 ? cfg->threads : (int)std::max(1u, std::thread::hardware_concurrency());
    if (cfg->device >= 0) {
        infx_config dc{}; dc.device = cfg->device; dc.range_docs = cfg->range_docs; dc.max_depth = h.maxDepth;
        int32_t rc = infx_create(&dc, &e->dev);
        if (rc) { g_eerr = infx_last_error(); delete e; return rc; }
    }
    e->def = new infx_session(); e->def->e = e;
    *out = e; return INFX_OK;
}

void infx_engine_destroy(infx_engine* e) {
    if (!e) return;
    infx_session* S = e->def;
    if (S && S->stream) infx_stream_destroy(S->stream);
    delete S;
    if (e->dev) infx_destroy(e->dev);
    delete e;
}

// SearchEngine.IndexDocuments: n documents x field_count fields (UTF-16 arena + offsets), keys may be null (key = index)
int32_t infx_engine_index_documents(infx_engine* e, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs,
                                    int32_t field_count, const int32_t* field_weights) {
    if (!e || n < 0 || (n && (!arena || !offs)) || field_count < 1 || !field_weights) return efail(INFX_EINVAL, "bad arguments");
    if (e->indexed) return efail(INFX_EINVAL, "this engine instance is already indexed (re-indexing: create a new engine)");
    if (n > 0x7FFFFFF0ll) return efail(INFX_EINVAL, "too many documents");
    DocSource src{n, field_count, field_weights, keys, (const u16*)arena, offs};
    build_index(src, e->ix);
    e->keysAreIds = (keys == nullptr);
    if (keys) { e->keyToFirst.reserve((size_t)n * 2); for (int64_t d = 0; d < n; d++) e->keyToFirst.emplace(keys[d], (int32_t)d); }
    if (e->dev) {
        HostIndex& ix = e->ix;
        int32_t rc = infx_upload_docs(e->dev, (uint32_t)ix.N, ix.docLen.data(), ix.avgdl, ix.docKey.data(), ix.textOff.data(), (const uint16_t*)ix.text.data());
        if (!rc) rc = infx_upload_postings(e->dev, (uint32_t)ix.terms.K(), ix.terms.off.data(), ix.terms.doc.data(), ix.terms.w.data(), ix.df.data());
        if (!rc) rc = infx_upload_prefix_docsets(e->dev, (uint32_t)(ix.psOff.size() - 1), ix.psOff.data(), ix.psDocs.data());
        if (!rc) rc = infx_stream_create(e->dev, &e->def->stream);
        if (rc) { g_eerr = infx_last_error(); return rc; }
    }
    e->indexed = true;
    return INFX_OK;
}

static int32_t key_to_id(infx_engine* e, int64_t key) {
    if (e->keysAreIds) return (key >= 0 && key < e->ix.N) ? (int32_t)key : -1;
    auto it = e->keyToFirst.find(key); return it == e->keyToFirst.end() ? -1 : it->second;
}

// SearchEngine.Search for a batch of queries (each = one reference Search call; results are independent of batching).
// out_* are nq x max_results; out_flags: bit0 unsupported (short-query path), bit1 coverage stage ran, bit2 fell back to Stage 1
static int32_t search_batch_impl(infx_engine* e, infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                 int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                 uint32_t* out_counts, uint32_t* out_flags) {
    if (!e || (nq && (!q_arena || !q_offs || !out_keys || !out_scores || !out_counts)) || max_results < 1) return efail(INFX_EINVAL, "bad arguments");
    if (!e->indexed) { for (uint32_t i = 0; i < nq; i++) out_counts[i] = 0; return INFX_OK; }   // Result.MakeEmptyResult(), SearchEngine.cs:261-262
    if (!e->dev || !S || !S->stream) return efail(INFX_EHIP, "no GPU: the scoring hot path has no CPU fallback");
    if (depth <= 0 || depth > e->ix.cfg.maxDepth) return efail(INFX_EINVAL, "CoverageDepth exceeds the engine's max_depth");
    const HostIndex& ix = e->ix;
    const int threads = e->threads;
    g_eerr.clear();
    double t0 = now_ms();
    // ---------------- Stage-1 planning (host, parallel over queries) ----------------
    std::vector<QueryPlan>& plans = S->lastPlans; plans.assign(nq, QueryPlan());
    parallel_dyn(nq, threads, 4, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; i++) plan_stage1(ix, e->fuzzy, uview((const u16*)q_arena + q_offs[i], (size_t)(q_offs[i + 1] - q_offs[i])), depth, plans[i]);
    });
    double tPlanPar = now_ms() - t0;
    std::vector<infx_query> dq; std::vector<infx_term> dterms; std::vector<int32_t> extra; std::vector<uint32_t> qmap;   // device batch -> query index
    {
        size_t nterm = 0, nextra = 0;
        std::vector<size_t> termBase(nq), extraBase(nq);
        for (uint32_t i = 0; i < nq; i++) {
            QueryPlan& P = plans[i];
            termBase[i] = nterm; extraBase[i] = nextra;
            if (P.blank || P.unsupported || P.noTerms) continue;
            P.q.term_off = (uint32_t)nterm; nterm += P.terms.size();
            for (size_t k = 0; k < P.terms.size(); k++) if (P.terms[k].term_id < 0) nextra += P.fuzzy[k]->docs.size();
            dq.push_back(P.q); qmap.push_back(i);
        }
        if (nextra > 0xFFFFFFF0ull) return efail(INFX_ECAPACITY, "fuzzy unions of this batch exceed 2^32 postings; split the batch");
        dterms.resize(nterm); extra.resize(nextra);
        parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) {
            for (int64_t i = b; i < en; i++) {
                QueryPlan& P = plans[i];
                if (P.blank || P.unsupported || P.noTerms) continue;
                size_t xo = extraBase[i];
                for (size_t k = 0; k < P.terms.size(); k++) {
                    infx_term t = P.terms[k];
                    if (t.term_id < 0) { const auto& d = P.fuzzy[k]->docs; t.extra_off = (uint32_t)xo; std::memcpy(extra.data() + xo, d.data(), d.size() * 4); xo += d.size(); }
                    dterms[termBase[i] + k] = t;
                }
            }
        });
    }
    double t1 = now_ms();
    // ---------------- Stage 1 on the GPU ----------------
    const uint32_t nd = (uint32_t)dq.size();
    std::vector<infx_hit>& hits = S->lastHits; std::vector<uint32_t>& hitCount = S->lastHitCount;
    hits.assign((size_t)nd * depth, infx_hit{0, 0.f}); hitCount.assign(nd, 0); S->lastStride = depth;
    S->msAcc = S->msSel = S->msCov = 0; S->algBytes = 0;
    if (nd) {
        int32_t rc = infx_stage1_batch(S->stream, nd, dq.data(), (uint32_t)dterms.size(), dterms.data(), (uint32_t)extra.size(), extra.data(), hits.data(), hitCount.data());
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_timings(S->stream, &S->msAcc, &S->msSel, nullptr);
        infx_last_alg_bytes(S->stream, &S->streamedBytes);
        infx_last_candidates(S->stream, &S->s1Candidates);
        // B_alg(q) = sum_t df_t * 5 B (4 B for fuzzy virtual terms) + card(C_q) * 4 B + depth * 12 B
        uint64_t ab = 0;
        for (auto& t : dterms) ab += t.term_id >= 0 ? (uint64_t)ix.terms.len((uint32_t)t.term_id) * 5ull : (uint64_t)t.extra_len * 4ull;
        uint64_t nh = 0; for (uint32_t c : hitCount) nh += c;
        S->algBytes = ab + S->s1Candidates * 4ull + nh * 12ull;
    }
    double t2 = now_ms();
    // ---------------- Stage-2 preparation (host) ----------------
    struct PerQ {
        std::vector<Entry> stage1;          // consolidated Stage-1 (score desc, key asc)
        std::vector<int32_t> stage1Doc;
        bool runCov = false, wmAny = false, done = false;
        uint32_t candOff = 0, candCount = 0; int covIndex = -1;
        int32_t idx0 = -1, idx1 = -1;       // docs with docIndex 0 / 1
    };
    std::vector<PerQ> pq(nq);
    std::vector<int> devOf(nq, -1);
    for (uint32_t j = 0; j < nd; j++) devOf[qmap[j]] = (int)j;
    std::vector<std::vector<infx_cov_cand>> candLocal(nq);
    std::vector<infx_cov_query> covQ(nq);
    std::vector<int32_t> covErr(nq, 0);
    const bool covEnabled = ix.cfg.enableCoverage && enable_coverage;
    parallel_dyn(nq, threads, 4, [&](int64_t b, int64_t en, int) {
        WmResult wm; std::vector<int32_t> sortedTop, overlap, uniq;
        for (int64_t i = b; i < en; i++) {
            QueryPlan& P = plans[i]; PerQ& S = pq[i];
            if (P.blank || P.unsupported) { S.done = true; continue; }
            int j = devOf[i];
            if (j >= 0) {
                uint32_t c = hitCount[j];
                S.stage1.resize(c); S.stage1Doc.resize(c);
                // TopKHeap -> ConsolidateSegments: (score desc, key asc). Device order is (score desc, internal id asc).
                std::vector<uint32_t> o(c); for (uint32_t k = 0; k < c; k++) o[k] = k;
                const infx_hit* H = hits.data() + (size_t)j * depth;
                std::sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) { if (H[x].score != H[y].score) return H[x].score > H[y].score; return ix.docKey[H[x].doc] < ix.docKey[H[y].doc]; });
                for (uint32_t k = 0; k < c; k++) { S.stage1[k] = Entry{H[o[k]].score, ix.docKey[H[o[k]].doc], 0}; S.stage1Doc[k] = H[o[k]].doc; }
            }
            const ustr& st = P.searchText;
            bool isShort = !st.empty() && st.size() <= 3;
            if (isShort) for (u16 ch : st) if (is_delim(ch)) { isShort = false; break; }
            if (isShort && (int)S.stage1.size() >= max_results) { S.done = true; continue; }     // SearchPipeline.cs:114-120
            int shortCount = 0;
            if (isShort) { int64_t pk = ix.prefixKeys.find(st); shortCount = pk >= 0 ? (int)ix.prefixPop[pk] : 0; }
            bool skipCov = isShort && shortCount > 500;
            if (!covEnabled || skipCov) { S.done = true; continue; }
            S.runCov = true;
            // ---- ExecuteCoverageStage preparation ----
            wm_collect(ix, st, true, wm);
            S.wmAny = wm.any;
            size_t ntop = std::min<size_t>(S.stage1.size(), (size_t)depth);
            sortedTop.assign(S.stage1Doc.begin(), S.stage1Doc.begin() + ntop);
            std::sort(sortedTop.begin(), sortedTop.end());
            overlap.clear();
            if (wm.any) for (int32_t d : sortedTop) if (wm_contains(wm, d)) overlap.push_back(d);   // ascending
            size_t wmLimit = (size_t)std::max(0, depth - (int)overlap.size());
            size_t need = std::max<size_t>(wmLimit, 2);
            wm_first_unique(wm, sortedTop, need, uniq);
            // docIndex 0/1 = first two keys in insertion order: Stage-1 docs, then WordMatcher-only ids ascending
            int32_t first2[2] = {-1, -1}; int nf = 0;
            for (size_t k = 0; k < ntop && nf < 2; k++) first2[nf++] = S.stage1Doc[k];
            for (size_t k = 0; k < uniq.size() && nf < 2; k++) first2[nf++] = uniq[k];
            S.idx0 = first2[0]; S.idx1 = first2[1];
            covErr[i] = prepare_cov_query(ix, st, covQ[i]);
            if (covErr[i]) continue;
            auto& CL = candLocal[i];
            auto push = [&](int32_t doc, float base) { infx_cov_cand c{}; c.query = 0; c.doc = doc; c.base_score = base; c.want_lcs = (doc == first2[0] || doc == first2[1]) ? 1 : 0; CL.push_back(c); };
            for (int32_t d : overlap) push(d, 0.f);
            for (size_t k = 0; k < uniq.size() && k < wmLimit; k++) push(uniq[k], 0.f);
            float maxT = ntop ? S.stage1[0].score : 1.f;
            for (size_t k = 0; k < ntop; k++) push(S.stage1Doc[k], maxT > 0 ? S.stage1[k].score / maxT : 0.f);
        }
    });
    for (uint32_t i = 0; i < nq; i++) if (covErr[i]) return efail(covErr[i], "query exceeds the Stage-2 envelope (INFX_MAX_QUERY_TOKENS / INFX_MAX_QUERY_CHARS / token length)");
    std::vector<infx_cov_query> covBatch; std::vector<infx_cov_cand>& cands = S->lastCands; cands.clear();
    for (uint32_t i = 0; i < nq; i++) {
        PerQ& S = pq[i];
        if (!S.runCov) continue;
        S.covIndex = (int)covBatch.size(); covBatch.push_back(covQ[i]);
        S.candOff = (uint32_t)cands.size(); S.candCount = (uint32_t)candLocal[i].size();
        for (auto c : candLocal[i]) { c.query = (uint32_t)S.covIndex; cands.push_back(c); }
    }
    double t3 = now_ms();
    // ---------------- Stage 2 on the GPU ----------------
    std::vector<infx_cov_out>& outs = S->lastOuts; outs.assign(cands.size(), infx_cov_out{});
    S->s2Candidates = cands.size(); S->s2TextBytes = 0;
    for (auto& c : cands) S->s2TextBytes += 2 * (ix.textOff[c.doc + 1] - ix.textOff[c.doc]);
    if (e->cfg.want_features) S->lastFeat.assign(cands.size() * INFX_NFEAT, 0);
    if (!cands.empty()) {
        int32_t rc = infx_stage2_batch(S->stream, (uint32_t)covBatch.size(), covBatch.data(), (uint32_t)cands.size(), cands.data(), outs.data(), e->cfg.want_features ? S->lastFeat.data() : nullptr);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_timings(S->stream, nullptr, nullptr, &S->msCov);
        for (auto& o : outs) if (o.status) return efail(INFX_EUNSUPPORTED, "a candidate document exceeds the Stage-2 envelope (INFX_MAX_DOC_TOKENS)");
    }
    double t4 = now_ms();
    // ---------------- final ordering / truncation (host) ----------------
    parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) {
        std::vector<Entry> fin, cons;
        for (int64_t i = b; i < en; i++) {
            PerQ& S = pq[i]; const QueryPlan& P = plans[i];
            uint32_t flags = 0; const std::vector<Entry>* res = &S.stage1;
            if (P.unsupported) flags |= 1;
            if (S.runCov) {
                flags |= 2;
                int maxWordHits = 0; uint8_t hits01[2] = {0, 0}, lcs01[2] = {0, 0};
                fin.clear();
                for (uint32_t k = 0; k < S.candCount; k++) {
                    const infx_cov_cand& c = cands[S.candOff + k]; const infx_cov_out& o = outs[S.candOff + k];
                    maxWordHits = std::max(maxWordHits, o.word_hits_full);
                    for (int z = 0; z < 2; z++) { int32_t dz = z == 0 ? S.idx0 : S.idx1; if (dz >= 0 && c.doc == dz) { if (hits01[z] == 0) hits01[z] = o.word_hits; if (lcs01[z] == 0) lcs01[z] = o.lcs; } }
                    fin.push_back(Entry{o.score, ix.docKey[c.doc], o.tiebreaker});
                }
                if (maxWordHits == 0 && !S.wmAny) { cons.clear(); }
                else {
                    // TopKHeap(depth): best `depth` by the total order, then ConsolidateSegments (best per key, descending)
                    std::sort(fin.begin(), fin.end(), [](const Entry& a, const Entry& c) { return cmp_entry(a, c) > 0; });
                    if ((int)fin.size() > depth) fin.resize(depth);
                    cons.clear();
                    std::unordered_map<int64_t, char> seen; seen.reserve(fin.size() * 2);
                    for (auto& x : fin) if (seen.emplace(x.key, 1).second) cons.push_back(x);
                    int truncIdx = -1;
                    int minHits = std::max(1, maxWordHits - 0);
                    int64_t k0 = S.idx0 >= 0 ? ix.docKey[S.idx0] : INT64_MIN, k1 = S.idx1 >= 0 ? ix.docKey[S.idx1] : INT64_MIN;
                    for (int r = (int)cons.size() - 1; r >= 0; r--) {
                        uint8_t wh = 0, lc = 0;
                        if (S.idx0 >= 0 && cons[r].key == k0) { wh = hits01[0]; lc = lcs01[0]; }
                        else if (S.idx1 >= 0 && cons[r].key == k1) { wh = hits01[1]; lc = lcs01[1]; }
                        if (wh >= minHits || lc > 0 || cons[r].score >= 254.f) { truncIdx = r; break; }
                    }
                    int resultCount = truncIdx == -1 ? max_results : std::min(std::max(0, truncIdx) + 1, max_results);
                    if ((int)cons.size() > resultCount) cons.resize(resultCount);
                }
                if (cons.empty() && !S.stage1.empty()) { flags |= 4; res = &S.stage1; } else res = &cons;
            }
            uint32_t cnt = (uint32_t)std::min<size_t>(res->size(), (size_t)max_results);
            out_counts[i] = cnt;
            for (uint32_t k = 0; k < cnt; k++) {
                out_keys[(size_t)i * max_results + k] = (*res)[k].key; out_scores[(size_t)i * max_results + k] = (*res)[k].score;
                if (out_ties) out_ties[(size_t)i * max_results + k] = (*res)[k].tie;
            }
            if (out_flags) out_flags[i] = flags;
        }
    });
    double t5 = now_ms();
    if (getenv("INFX_DEBUG")) {
        int nm[4] = {0, 0, 0, 0}; unsigned long long dfsum[4] = {0, 0, 0, 0}; size_t maxT = 0;
        for (auto& P : plans) { if (P.blank || P.unsupported || P.noTerms) continue; nm[P.q.mode]++; maxT = std::max(maxT, P.terms.size());
            for (auto& t : P.terms) dfsum[P.q.mode] += t.term_id >= 0 ? (unsigned long long)ix.terms.len((uint32_t)t.term_id) : t.extra_len; }
        fprintf(stderr, "[infx] modes: prefix=%d disj=%d and=%d | postings per mode: %llu %llu %llu | maxT=%zu\n", nm[1], nm[2], nm[3], dfsum[1], dfsum[2], dfsum[3], maxT);
        fprintf(stderr, "[infx] nq=%u dev=%u terms=%zu extra=%zu cands=%zu | plan %.1f (build %.1f) s1 %.1f prep2 %.1f s2 %.1f post %.1f ms | fuzzy calls=%lld %.1f ms-cpu (ld1 %.1f) docs=%lld\n",
                nq, nd, dterms.size(), extra.size(), cands.size(), t1 - t0, tPlanPar, t2 - t1, t3 - t2, t4 - t3, t5 - t4,
                (long long)e->fuzzy.fuzzyCalls.exchange(0), e->fuzzy.fuzzyNs.exchange(0) / 1e6, e->fuzzy.ld1Ns.exchange(0) / 1e6, (long long)e->fuzzy.fuzzyDocs.exchange(0));
    }
    S->tPrep1 = t1 - t0; S->tStage1 = t2 - t1; S->tPrep2 = t3 - t2; S->tStage2 = t4 - t3; S->tPost = t5 - t4;
    return INFX_OK;
}

int32_t infx_engine_search_batch(infx_engine* e, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                 int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                 uint32_t* out_counts, uint32_t* out_flags) {
    if (!e) return efail(INFX_EINVAL, "null engine");
    return search_batch_impl(e, e->def, nq, q_arena, q_offs, max_results, depth, enable_coverage, out_keys, out_scores, out_ties, out_counts, out_flags);
}
int32_t infx_engine_session_create(infx_engine* e, infx_session** out) {
    if (!e || !out) return efail(INFX_EINVAL, "null");
    if (!e->indexed || !e->dev) return efail(INFX_EINVAL, "sessions need an indexed engine with a GPU");
    infx_session* S = new infx_session(); S->e = e;
    int32_t rc = infx_stream_create(e->dev, &S->stream);
    if (rc) { g_eerr = infx_last_error(); delete S; return rc; }
    *out = S; return INFX_OK;
}
void infx_engine_session_destroy(infx_session* S) { if (!S) return; if (S->stream) infx_stream_destroy(S->stream); delete S; }
int32_t infx_engine_session_search_batch(infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                         int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                         uint32_t* out_counts, uint32_t* out_flags) {
    if (!S) return efail(INFX_EINVAL, "null session");
    return search_batch_impl(S->e, S, nq, q_arena, q_offs, max_results, depth, enable_coverage, out_keys, out_scores, out_ties, out_counts, out_flags);
}
int32_t infx_engine_session_last_timings(infx_session* S, double* host_ms5, float* kernel_ms3, uint64_t* alg_bytes3) {
    if (!S) return efail(INFX_EINVAL, "null");
    if (host_ms5) { host_ms5[0] = S->tPrep1; host_ms5[1] = S->tStage1; host_ms5[2] = S->tPrep2; host_ms5[3] = S->tStage2; host_ms5[4] = S->tPost; }
    if (kernel_ms3) { kernel_ms3[0] = S->msAcc; kernel_ms3[1] = S->msSel; kernel_ms3[2] = S->msCov; }
    if (alg_bytes3) { alg_bytes3[0] = S->algBytes; alg_bytes3[1] = S->s2Candidates; alg_bytes3[2] = S->s2TextBytes; alg_bytes3[3] = S->streamedBytes; alg_bytes3[4] = S->s1Candidates; }
    return INFX_OK;
}

int32_t infx_engine_last_timings(infx_engine* e, double* host_ms5, float* kernel_ms3, uint64_t* alg_bytes3) {
    if (!e) return efail(INFX_EINVAL, "null");
    infx_session* S = e->def;
    if (host_ms5) { host_ms5[0] = S->tPrep1; host_ms5[1] = S->tStage1; host_ms5[2] = S->tPrep2; host_ms5[3] = S->tStage2; host_ms5[4] = S->tPost; }
    if (kernel_ms3) { kernel_ms3[0] = S->msAcc; kernel_ms3[1] = S->msSel; kernel_ms3[2] = S->msCov; }
    if (alg_bytes3) { alg_bytes3[0] = S->algBytes; alg_bytes3[1] = S->s2Candidates; alg_bytes3[2] = S->s2TextBytes; alg_bytes3[3] = S->streamedBytes; alg_bytes3[4] = S->s1Candidates; }
    return INFX_OK;
}

// ---- introspection for parity tests (host logic is testable without a GPU) ------------------------------------------------
int32_t infx_engine_index_stats(infx_engine* e, int64_t* n_docs, int64_t* n_terms, int64_t* n_postings, float* avgdl) {
    if (!e) return efail(INFX_EINVAL, "null");
    if (n_docs) *n_docs = e->ix.N; if (n_terms) *n_terms = (int64_t)e->ix.terms.K();
    if (n_postings) *n_postings = (int64_t)e->ix.terms.doc.size(); if (avgdl) *avgdl = e->ix.avgdl;
    return INFX_OK;
}
int32_t infx_engine_export_index(infx_engine* e, int32_t* df, uint64_t* post_off, int32_t* post_doc, uint8_t* post_w, float* doc_len) {
    if (!e) return efail(INFX_EINVAL, "null");
    const HostIndex& ix = e->ix; size_t T = ix.terms.K();
    if (df) std::memcpy(df, ix.df.data(), T * 4);
    if (post_off) std::memcpy(post_off, ix.terms.off.data(), (T + 1) * 8);
    if (post_doc) std::memcpy(post_doc, ix.terms.doc.data(), ix.terms.doc.size() * 4);
    if (post_w) std::memcpy(post_w, ix.terms.w.data(), ix.terms.w.size());
    if (doc_len) std::memcpy(doc_len, ix.docLen.data(), (size_t)ix.N * 4);
    return INFX_OK;
}
int32_t infx_engine_term_text(infx_engine* e, int32_t t, uint16_t* out, int32_t cap) {
    if (!e || t < 0 || t >= (int32_t)e->ix.terms.K()) return -1;
    uview s = e->ix.terms.keys.key((uint32_t)t);
    std::memcpy(out, s.data(), (size_t)std::min<int>(cap, (int)s.size()) * 2);
    return (int32_t)s.size();
}
int32_t infx_engine_match_ld1(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap) {
    std::vector<int> m; int c = match_ld1(e->ix, uview((const u16*)q, len), m, cap);
    for (size_t i = 0; i < m.size(); i++) out[i] = m[i];
    return c;
}
// Stage-1 plan of one query (no GPU needed): returns number of terms; mode/prefix_set/n_and/df_s1/df_s2 in meta[5]
int32_t infx_engine_plan(infx_engine* e, const uint16_t* q, int32_t len, int32_t depth, int32_t* term_ids, int32_t* dfs, float* idfs,
                         uint8_t* roles, uint8_t* ranks, int32_t cap, int32_t* meta, int32_t* flags) {
    if (!e) return -1;
    QueryPlan P; plan_stage1(e->ix, e->fuzzy, uview((const u16*)q, len), depth, P);
    if (flags) *flags = (P.blank ? 1 : 0) | (P.unsupported ? 2 : 0) | (P.noTerms ? 4 : 0);
    int n = (int)P.terms.size();
    for (int i = 0; i < n && i < cap; i++) {
        term_ids[i] = P.terms[i].term_id; idfs[i] = P.terms[i].idf; roles[i] = P.terms[i].role; ranks[i] = P.terms[i].rank;
        dfs[i] = P.terms[i].term_id >= 0 ? e->ix.df[P.terms[i].term_id] : (int32_t)P.terms[i].extra_len;
    }
    if (meta) { meta[0] = P.q.mode; meta[1] = P.q.prefix_set; meta[2] = P.q.n_and; meta[3] = P.q.df_s1; meta[4] = P.q.df_s2; }
    return n;
}
// WordMatcherLookup.Execute, fully enumerated (tests only): sorted unique ids
int64_t infx_engine_wordmatcher(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int64_t cap) {
    WmResult wm; ustr t = normalize(uview((const u16*)q, len)); lower_inplace(t);
    wm_collect(e->ix, t, true, wm);
    std::vector<int32_t> all;
    for (auto& l : wm.lists) all.insert(all.end(), l.p, l.p + l.n);
    std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
    for (size_t i = 0; i < all.size() && (int64_t)i < cap; i++) out[i] = all[i];
    return (int64_t)all.size();
}
int32_t infx_engine_prefix_pop(infx_engine* e, const uint16_t* p, int32_t len) {
    int64_t k = e->ix.prefixKeys.find(uview((const u16*)p, len)); return k < 0 ? 0 : (int32_t)e->ix.prefixPop[k];
}
// last batch: Stage-1 hits of query i (device order re-sorted to the reference's) and the Stage-2 records
int32_t infx_engine_last_stage1(infx_engine* e, uint32_t qi, int64_t* keys, float* scores, int32_t cap) {
    if (!e) return -1;
    infx_session* S = e->def;
    if (qi >= S->lastPlans.size()) return -1;
    // recompute the device index of query qi
    uint32_t j = 0; bool found = false;
    for (uint32_t i = 0; i < S->lastPlans.size(); i++) { const QueryPlan& P = S->lastPlans[i]; if (P.blank || P.unsupported || P.noTerms) { if (i == qi) break; continue; } if (i == qi) { found = true; break; } j++; }
    if (!found) return 0;
    uint32_t c = S->lastHitCount[j]; const infx_hit* H = S->lastHits.data() + (size_t)j * S->lastStride;
    std::vector<uint32_t> o(c); for (uint32_t k = 0; k < c; k++) o[k] = k;
    std::sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) { if (H[x].score != H[y].score) return H[x].score > H[y].score; return e->ix.docKey[H[x].doc] < e->ix.docKey[H[y].doc]; });
    for (uint32_t k = 0; k < c && (int32_t)k < cap; k++) { keys[k] = e->ix.docKey[H[o[k]].doc]; scores[k] = H[o[k]].score; }
    return (int32_t)c;
}
int64_t infx_engine_last_stage2(infx_engine* e, uint32_t* query_of, int32_t* docs, float* base, float* scores, uint8_t* ties, int32_t* feat, int64_t cap) {
    if (!e) return -1;
    infx_session* S = e->def;
    int64_t n = (int64_t)S->lastCands.size();
    // map cov index -> query index
    for (int64_t i = 0; i < n && i < cap; i++) {
        const infx_cov_cand& c = S->lastCands[i]; const infx_cov_out& o = S->lastOuts[i];
        if (query_of) query_of[i] = c.query; if (docs) docs[i] = c.doc; if (base) base[i] = c.base_score;
        if (scores) scores[i] = o.score; if (ties) ties[i] = o.tiebreaker;
        if (feat) { if (S->lastFeat.size() >= (size_t)(i + 1) * INFX_NFEAT) std::memcpy(feat + (size_t)i * INFX_NFEAT, S->lastFeat.data() + (size_t)i * INFX_NFEAT, INFX_NFEAT * 4); else std::memset(feat + (size_t)i * INFX_NFEAT, 0, INFX_NFEAT * 4); }
    }
    return n;
}
int32_t infx_engine_normalize(const uint16_t* s, int32_t len, int32_t lower, uint16_t* out, int32_t cap) {
    ustr r = normalize(uview((const u16*)s, len)); if (lower) lower_inplace(r);
    std::memcpy(out, r.data(), (size_t)std::min<int>(cap, (int)r.size()) * 2);
    return (int32_t)r.size();
}
// raw access for bench.py (HBM-resident inputs are the engine's; these expose the flat host arrays for oracle adoption)
int32_t infx_engine_device_handles(infx_engine* e, infx_index** idx, infx_stream** st) { if (!e) return INFX_EINVAL; if (idx) *idx = e->dev; if (st) *st = e->def->stream; return INFX_OK; }

} // extern "C"
