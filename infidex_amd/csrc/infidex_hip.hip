// libinfidex_hip.so — device side + C ABI (include/infidex_hip.h). gfx950 (MI355X) only.
//
// Replaces, for a BATCH of queries on a device-resident index shard:
//   Bm25Scorer.Search + TieredCandidateSelector.SelectCandidates   (Indexing/Bm25Scorer.cs:56-193,
//        Scoring/TieredCandidateSelector.cs:53-237)        -> k_accumulate + k_select
//   CoverageEngine.CalculateFeatures + FusionScorer.Calculate      (Coverage/CoverageEngine.cs:174-382,
//        Scoring/FusionScorer.cs:19-236)                    -> k_stage2   (stage2.hip.inc)
//
// Stage-1 design (HBM-bound integer/byte streaming, no MFMA): the doc-id space is cut into ranges of R docs. One
// wave (= one 64-thread workgroup) owns one (query, range): the fp32 partial scores and the tier flags of the R docs
// live in LDS; every posting list of the query is read exactly once, coalesced, from the slice that falls in the
// range (located through a per-term range skip table); a wave executes its LDS read-modify-writes in program order, so
// the per-document accumulation order is the reference's term order with no barriers and no atomics. Candidate tiers
// (quirk Q11) are evaluated from per-doc flags; a range that holds no candidate-generating posting exits before it
// streams anything else (range-granular WAND skip). Survivors are compacted with wave ballots into a batch arena and
// k_select picks the top-`depth` per query with an LDS radix select + bitonic sort.
#include <type_traits>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <mutex>
#include <chrono>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: the functions are resolved with dlopen (rccl_api)
#include "../../include/infidex_hip.h"
#include "unicode_tables.h"        // generated: BMP simple case mappings, letter set (tools/gen_unicode_tables.py)

#define WAVE 64
#define FILT_MAXCOL 64

static thread_local std::string g_err;
static int32_t fail(int32_t code, const char* fmt, const char* a = "") {
    char buf[512]; snprintf(buf, sizeof buf, fmt, a); g_err = buf; return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(INFX_EHIP, #x ": %s", hipGetErrorString(e_)); } while (0)
// Start of an API call: select the index's device and drop whatever error an EARLIER runtime call of this thread left behind — the library checks
// hipGetLastError() after its launches, and a stale error of someone else's call (torch next door leaves "invalid device ordinal" / "peer access already
// enabled" behind as a matter of course) would otherwise be reported as the failure of the first launch that follows.
static inline hipError_t enter_device(int device) { const hipError_t e = hipSetDevice(device); (void)hipGetLastError(); return e; }

// ---------------------------------------------------------------------------------------------------------------
struct DevIndex {
    int32_t N, T, R, rshift, nRanges, docBase, totalDocs;
    const uint64_t* postOff; const int32_t* postDoc; const uint8_t* postW;
    const float* docNorm;      // K1*((1-B) + (B/avgdl)*dl) — the 8-lane formula of Bm25Scorer.cs:413-416
    const float* docLen;       // VectorModel._docLengths (the scalar-tail formula of ComputeTermScore needs dl / avgdl, k_exact1)
    const uint8_t* deleted;    // Document.Deleted per GLOBAL internal id (nullptr = nothing deleted)
    const int64_t* docKey;
    const uint64_t* textOff; const uint16_t* text;
    const uint32_t* skipIdx;   // per term: first entry in skipTbl, or 0xFFFFFFFF
    const uint32_t* skipTbl;   // nRanges+1 offsets (relative to postOff[t]) per skipped term
    const uint32_t* psSkip;    // per prefix DocSet: first entry in psSkipTbl, or 0xFFFFFFFF
    const uint32_t* psSkipTbl; // nRanges+1 offsets (relative to psOff[k]) per skipped set
    const int32_t* wmExact;    // WordMatcher exact-word doc lists (infx_upload_wordmatcher), global ids
    const int32_t* wmLd1;      // WordMatcher symmetric-delete doc lists
    const int64_t* docKeyAll;  // DocumentKey by GLOBAL internal id (== docKey when unsharded)
    int packed;                // 1: postDoc entries are (doc << 8) | tf (shards below 2^24 - 1 documents), 0: plain doc ids + postW
    const uint64_t* psOff; const int32_t* psDocs; uint32_t nSets;
};

struct infx_index {
    infx_config cfg;
    DevIndex d{};
    std::vector<void*> allocs;
    bool havePostings = false, haveDocs = false, haveWm = false;
    uint64_t nWmExact = 0, nWmLd1 = 0;
    float avgdl = 0.f;
    std::vector<uint64_t> hPostOff;   // host copy of the (padded) list starts
    std::vector<uint64_t> hPostLen;   // true list lengths
    std::vector<int32_t> hDf;
    std::vector<uint32_t> hSkipIdx;   // host copy of DevIndex::skipIdx (filled by infx_upload_postings)
    uint8_t* dDeleted = nullptr;      // device copy of the global Document.Deleted flags (infx_set_deleted)
    const uint32_t* colCodes[FILT_MAXCOL] = {}; uint32_t colValues[FILT_MAXCOL] = {}; uint32_t colDocs[FILT_MAXCOL] = {}; uint32_t colCap[FILT_MAXCOL] = {};   // device-resident columns (infx_upload_column)
    std::vector<uint64_t> hPsOff;
    int rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;         // infx_set_shard_comm
    struct DevLookup* lk = nullptr;    // dictionaries / term trie of the device-side planning lookups (lookup.hip.inc); owned, freed by infx_destroy
    // Turnstile of the full-width phase (k_accumulate .. k_select) between the streams of an index: see infx_search_fused
    std::mutex turnMu; hipEvent_t turnEvent = nullptr;
    bool haveDict = false, haveTrie = false;
    bool hasAlias = false;            // the corpus text holds one of the 22 OrdinalIgnoreCase alias characters (infx_upload_docs counts them): Stage 2 runs its ALIAS instantiation
};

template <class Tp> static hipError_t dalloc(infx_index* ix, Tp** p, size_t n) {
    void* v = nullptr; hipError_t e = hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(Tp));
    if (e == hipSuccess) { ix->allocs.push_back(v); *p = (Tp*)v; }
    return e;
}

template <class Tp> static int32_t dcopy(infx_index* ix, const Tp** out, const Tp* src, size_t n) {
    Tp* d = nullptr;
    HIPCHK(dalloc(ix, &d, n));
    if (n) HIPCHK(hipMemcpy(d, src, n * sizeof(Tp), hipMemcpyHostToDevice));
    *out = d; return INFX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Device posting layout: every list starts on a 16-byte boundary and is padded to a multiple of 4 postings with the sentinel doc id
// 0x7F7F7F7F (beyond any real id).  k_accumulate streams lists with 16-byte loads from an aligned-down address; with this layout the
// extra elements of a group are either postings of the SAME list outside the block's doc range or sentinels, so a plain range check
// replaces the per-posting index check.  One thread per posting: find its list (upper bound in the unpadded offsets), copy.
__global__ void k_pad_lists(const uint64_t* __restrict__ offs, const uint64_t* __restrict__ offs2, uint32_t T, uint64_t P,
                            const int32_t* __restrict__ inDoc, const uint8_t* __restrict__ inW, int32_t* __restrict__ outDoc, uint8_t* __restrict__ outW, int packed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint64_t lo = 0, hi = (uint64_t)T + 1;                      // first index with offs[idx] > i
    while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (offs[m] <= i) lo = m + 1; else hi = m; }
    const uint64_t t = lo - 1;
    const uint64_t dst = offs2[t] + (i - offs[t]);
    outDoc[dst] = packed ? (int32_t)(((uint32_t)inDoc[i] << 8) | inW[i]) : inDoc[i]; outW[dst] = inW[i];
}

// skip table build: one thread per (skipped term, range boundary)
__global__ void k_build_skip(const uint64_t* postOff, const int32_t* postDoc, const uint32_t* skipTerms, uint32_t nSkip,
                             uint32_t* skipTbl, int nRanges, int rshift, int packed) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t per = (uint64_t)nRanges + 1;
    if (gid >= (uint64_t)nSkip * per) return;
    uint32_t s = (uint32_t)(gid / per); int r = (int)(gid % per);
    uint32_t t = skipTerms[s];
    uint64_t lo = postOff[t], hi = postOff[t + 1];
    int64_t target = (int64_t)r << rshift;
    uint64_t a = lo, b = hi;
    while (a < b) { uint64_t m = (a + b) >> 1; if ((int64_t)(packed ? (int32_t)((uint32_t)postDoc[m] >> 8) : postDoc[m]) < target) a = m + 1; else b = m; }
    skipTbl[(uint64_t)s * per + r] = (uint32_t)(a - lo);
}

__global__ void k_doc_norm(const float* docLen, float* norm, int n, float bDivAvg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float k1 = 1.2f, minDl = 1.f - 0.75f;
    float t1 = bDivAvg * docLen[i];
    float t2 = minDl + t1;
    norm[i] = k1 * t2;
}

// ---------------------------------------------------------------------------------------------------------------
struct DevTerm {           // per (query term), device layout
    uint64_t begin, end;   // absolute posting slice in postDoc/postW, or in extraDocs for virtual terms
    float idf;
    uint32_t skip;         // skipTbl base or 0xFFFFFFFF
    uint8_t role, rank, isVirtual, pad;   // pad = fuzzy-union dedupe group (member-list terms), 0 = none
    uint16_t refIdx, pad2; // position of the term in the query's Bm25Scorer order (member lists share their virtual term's position)
};
struct DevQuery {
    uint32_t termOff, numTerms;
    int32_t mode, prefixSet, depth, nAnd;
    uint32_t refOff, numRef;   // the query's terms as the caller passed them (Bm25Scorer order): DevRefTerm[refOff .. +numRef)
};
struct SelRule {           // decided on the host from the (global) class counts
    int32_t mode;          // INFX_MODE_*
    int32_t cutoffRank;    // DISJ: include class&0x7F <= cutoff
    uint32_t classMask;    // AND: include if (class & classMask) != 0
    int32_t depth;
    uint32_t total;        // upper bound on the number of entries the rule lets through
    uint32_t pad;
};

template <class Tp> __device__ __forceinline__ Tp rfl(Tp v) { return v; }
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t rfl_u64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// posting lists of a packed index hold (doc << 8) | tf; sentinels are 0xFFFFFFFF
__device__ __forceinline__ int32_t post_doc(int32_t v, int packed) { return packed ? (int32_t)((uint32_t)v >> 8) : v; }
__device__ __forceinline__ uint64_t lower_bound_post(const int32_t* a, uint64_t lo, uint64_t hi, int32_t target, int packed) {
    while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (post_doc(a[m], packed) < target) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ uint64_t lower_bound_i32(const int32_t* a, uint64_t lo, uint64_t hi, int32_t target) {
    while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (a[m] < target) lo = m + 1; else hi = m; }
    return lo;
}

#include "stage1.hip.inc"
#include "stage1_sparse.hip.inc"
#include "exact1.hip.inc"
#include "exact3.hip.inc"
#include "exactsh.hip.inc"
#include "stage2.hip.inc"

// TieredCandidateSelector tier rules evaluated from class counts (see header of this file / DESIGN.md)
__host__ __device__ static inline SelRule make_rule0(const infx_query& Q, const uint32_t* c) {
    SelRule r{}; r.mode = Q.mode; r.depth = Q.depth; r.cutoffRank = 127; r.classMask = 0xFFFFFFFFu;
    const long k = Q.depth;
    if (Q.mode == INFX_MODE_AND) {
        unsigned long long t0 = 0, t1 = 0;
        for (int i = 0; i < 16; i++) { if (i & 1) t0 += c[i]; if (i & 2) t1 += c[i]; }
        if ((long)t0 >= k * 2) { r.classMask = 1; return r; }                       // Tier 0 alone (:160-161)
        unsigned long long g = t0; uint32_t gmask = 1;
        if (Q.n_and >= 3 && (long)t0 < k * 3) { g = t1; gmask = 2 | 1; }            // Tier 1 (:165-171); T1 contains T0
        if ((long)g < k * 5) {                                                      // Tier 2 (:174-234)
            if (Q.df_s1 <= 0) { r.classMask = gmask; return r; }
            if ((long)Q.df_s1 >= k * 10 || Q.df_s2 <= 0) { r.classMask = 4 | gmask; return r; }   // G u S1 == S1
            r.classMask = 4 | 8 | gmask; return r;
        }
        r.classMask = gmask; return r;
    }
    if (Q.mode == INFX_MODE_DISJ) {
        // SelectCandidatesDisjunctive (:260-319): ranks [0, n_and) are not low-quality, the rest are
        int nElig = Q.n_and; bool hasSel = false; long local = 0; int cutoff = -1;
        int nRanks = 0; for (int i = 127; i >= 0; i--) if (c[i]) { nRanks = i + 1; break; }
        int totalRanks = nRanks > Q.df_s1 ? nRanks : Q.df_s1;   // df_s1 carries the number of ranks in DISJ mode
        for (int i = 0; i < totalRanks; i++) {
            bool lowq = i >= nElig;
            if (totalRanks > 1 && lowq && hasSel) continue;
            cutoff = i; local += c[i];
            if (!lowq && local > 0) hasSel = true;
            if (local >= k * 100) break;
        }
        r.cutoffRank = cutoff; return r;
    }
    return r;
}

__host__ __device__ static inline SelRule make_rule(const infx_query& Q, const uint32_t* c) {
    SelRule r = make_rule0(Q, c);
    unsigned long long tot = 0;
    if (Q.mode == INFX_MODE_AND) { for (int i = 0; i < 16; i++) if (i & r.classMask) tot += c[i]; }
    else if (Q.mode == INFX_MODE_DISJ) { for (int i = 0; i <= r.cutoffRank && i < 128; i++) tot += c[i]; tot += 128; /* pre-seen docs (< 100) are not in the histogram */ }
    else tot = c[1];
    r.total = (uint32_t)(tot < 0xFFFFFFFFull ? tot : 0xFFFFFFFFull); r.pad = 0;
    return r;
}

#include "fused.hip.inc"
#include "lookup.hip.inc"
#include "filter.hip.inc"

// ---------------------------------------------------------------------------------------------------------------
struct infx_filter {
    infx_index* ix; DevFilter d{}; uint32_t nops = 0, nleaves = 0; void *dOps = nullptr, *dLeaves = nullptr, *dTables = nullptr;
};

struct infx_stream {
    infx_index* ix;
    // post-filter / facets (infx_stream_set_postfilter): applied to the rows of every fused / sharded finalize on this stream
    infx_filter* postFilter = nullptr; uint32_t nFacet = 0; uint32_t facetCols[INFX_MAX_FACET_COLS] = {};
    void *dFDocs = nullptr, *dFacetCols = nullptr, *dFacCodes = nullptr, *dFacCounts = nullptr, *dFacN = nullptr;
    size_t capFDocs = 0, capFacCodes = 0, capFacCounts = 0, capFacN = 0;
    std::vector<uint32_t> hFacCodes, hFacCounts, hFacN; uint32_t facetNq = 0;
    hipStream_t st = nullptr;
    // Planning kernels (k_ld1, k_union count pass) are tiny and the host WAITS for their results (idf needs the union cardinalities): queued behind the
    // streaming kernels of the other batches in flight they came back after 10-15 ms (measured: plan_ms 14.9 per batch of which ~2 ms host work).  They
    // run on a stream of their own with the highest priority the device offers, so their few hundred waves are placed as soon as any CU has room.
    hipStream_t stPlan = nullptr, stMain = nullptr; hipEvent_t evPlan = nullptr;
    hipEvent_t evTurn = nullptr;                                   // end of this stream's k_select: what the next batch's k_accumulate (another stream) waits for
    hipStream_t stAux = nullptr; hipEvent_t evJoin = nullptr;      // the replay's two k_ex_chunk launches run side by side (both are tail-bound: one wave per chunk)
    hipEvent_t evA0, evA1, evS0, evS1, evC0, evC1, evP0, evP1, evF0, evF1, evX0, evX1, evSync;
    hipEvent_t evXa, evXb, evXc; bool timedReplayParts = false; float msReplayParts[4] = {0, 0, 0, 0};      // inside the replay: after the scan (k_ex_walk x2, k_ex_prefix, k_ex_theta), after the k_ex_chunk launches, after k_ex_heap
    bool timedReplay = false; float msReplay = 0.f; uint32_t lastFlagWhy[4] = {0, 0, 0, 0};     // exact replay of the last batch: kernel time, why its queries were flagged
    // fused pipeline workspaces
    void *dFQ = nullptr, *dFLists = nullptr, *dFOwned = nullptr, *dFS1 = nullptr, *dFMeta = nullptr, *dFQueries = nullptr, *dFKeys = nullptr, *dFScores = nullptr, *dFTies = nullptr, *dFCounts = nullptr, *dFFlags = nullptr, *dFErr = nullptr, *dFHitsAll = nullptr, *dFHcAll = nullptr, *dFPairs = nullptr;
    size_t capFQ = 0, capFLists = 0, capFOwned = 0, capFS1 = 0, capFMeta = 0, capFQueries = 0, capFKeys = 0, capFScores = 0, capFTies = 0, capFCounts = 0, capFFlags = 0, capFErr = 0, capFHitsAll = 0, capFHcAll = 0, capFPairs = 0;
    uint64_t fusedS1 = 0, fusedCands = 0, fusedTextBytes = 0;
    uint32_t fusedNq = 0; int fusedDepth = 0; bool fusedDebug = false, timedFused = false; float msFused[5] = {0, 0, 0, 0, 0};
    // device workspaces (grown on demand)
    void* dQueries = nullptr; size_t capQueries = 0;
    void* dTerms = nullptr; size_t capTerms = 0;
    void* dExtra = nullptr; size_t capExtra = 0;
    void* dRules = nullptr; size_t capRules = 0;
    void* dHits = nullptr; size_t capHits = 0;
    void* dHitCount = nullptr; size_t capHitCount = 0;
    void* dBlockOut = nullptr; size_t capBlockOut = 0;
    void* dBlockOutHi = nullptr; size_t capBlockOutHi = 0;
    void* dQBytes = nullptr; size_t capQBytes = 0; uint32_t lastNqAlloc = 0;
    void* dUOffs = nullptr; size_t capUOffs = 0; void* dUMem = nullptr; size_t capUMem = 0; void* dUCnt = nullptr; size_t capUCnt = 0;
    void* dURange = nullptr; size_t capURange = 0; void* dUBase = nullptr; size_t capUBase = 0; void* dUDocs = nullptr; size_t capUDocs = 0;
    struct PinChunk { char* base; size_t cap, off; }; std::vector<PinChunk> pins;      // pinned staging arena
    struct PendingOut { void* dst; const void* src; size_t bytes; }; std::vector<PendingOut> pendingOut; bool unsynced = false;
    std::vector<uint32_t> unionCount; std::vector<unsigned long long> unionBase{0};   // device-resident unions of the last infx_union_build
    void* dCounts = nullptr; size_t capCounts = 0;
    void* dDense = nullptr; size_t capDense = 0;      // k_accumulate_sparse -> k_accumulate hand-over flags, one byte per (query, stripe)
    void* dCovQ = nullptr; size_t capCovQ = 0;
    void* dCovQL = nullptr; size_t capCovQL = 0; uint32_t nLongQ = 0;      // infx_stage2_long_queries: the long-query table of the next Stage-2 call
    void* dCovC = nullptr; size_t capCovC = 0;
    void* dCovO = nullptr; size_t capCovO = 0;
    void* dCovF = nullptr; size_t capCovF = 0;
    int32_t* arDoc = nullptr; float* arScore = nullptr; uint8_t* arCls = nullptr; size_t arCap = 0;
    bool accLayoutBad = false;
    bool batchAlias = false, longAlias = false;      // ... or the coverage queries of the batch / its long-query table do
    unsigned long long* arMask = nullptr; size_t arMaskCap = 0; int maskWords = 0;     // per-row hit masks of the last accumulate launch
    void* dWideQ = nullptr; size_t capWideQ = 0; uint32_t nWide = 0;                     // queries of the batch with more than 64 reference terms: beyond the masks, never replayed
    uint32_t* arExc = nullptr; uint32_t* exCand = nullptr; infx_hit* exOut = nullptr; size_t exCap = 0;       // tf exception records, candidate lists, replay rows (arena-sized)
    void* exChunks = nullptr; size_t capExChunks = 0; void* exQueries = nullptr; size_t capExQueries = 0; void* exTasks = nullptr; size_t capExTasks = 0; uint32_t* exCounters = nullptr;
    void* dSelOrder = nullptr; size_t capSelOrder = 0;        // k_select_order: the batch's queries by row count, descending
    void* dAccOrder = nullptr; size_t capAccOrder = 0; uint32_t nAccHeavy = 0xFFFFFFFFu;      // k_accumulate: the batch's queries by row bound, descending, and how many of them go first (~0: query order)
    uint32_t nBoundGiants = 0;                               // queries of the batch whose row bound reaches k_select's multi-workgroup threshold
    void* dSelG = nullptr; size_t capSelG = 0;                // k_selg_hist / k_selg_gather: global histograms, key lists and flags of the batch's largest queries
    void* exContEnd = nullptr; size_t capExContEnd = 0;      // k_ex_cand: candidates up to the end of every (query, container)
    uint32_t exChunkCap = 0; size_t arBound = 0; unsigned long long maxQueryBound = 0;
    void* dDir = nullptr; size_t capDir = 0;
    void* dRefTerms = nullptr; size_t capRefTerms = 0; void* dExactFlag = nullptr; size_t capExactFlag = 0; uint32_t* dExactStat = nullptr;   // k_exact1 inputs
    uint32_t lastExact2[2] = {0, 0};                                                     // last batch: queries replayed exactly, of them by the sequential fallback                                            // (query, range) chunk directory
    unsigned long long* dCursor = nullptr;   // [0]=cursor [1]=algBytes
    uint32_t* dOverflow = nullptr;
    std::vector<infx_query> lastQ;            // kept between accumulate and select
    uint32_t lastNq = 0;
    float msAcc = 0, msSel = 0, msCov = 0;
    uint64_t lastAlgBytes = 0, lastCandTotal = 0;
    bool timedAcc = false, timedSel = false, timedCov = false;
    void* dStats = nullptr;      // k_accumulate profiling counters (INFX_ACC_SKIP=8)
    void* dExProf = nullptr; size_t capExProf = 0;      // INFX_EXACT_PROF: per-query cycle counters of the replay kernels (EXP_WORDS u64 per query)
    // exact replay across document shards (exactsh.hip.inc)
    void *dNext = nullptr, *dPrior = nullptr, *shBlob = nullptr, *dAllBlobs = nullptr, *dAllNext = nullptr, *dChainState = nullptr, *dChainNeed = nullptr;
    size_t capNext = 0, capPrior = 0, capShBlob = 0, capAllBlobs = 0, capAllNext = 0, capChainState = 0, capChainNeed = 0;
    uint32_t shHead[4] = {0, 0, 0, 0}; uint32_t shNd = 0; int shDepth = 0; bool shSelected = false;
    void* scratch[16] = {}; size_t capScratch[16] = {};      // infx_stream_scratch
    std::vector<void*> parked;                                // outgrown workspaces of a sharded stream (see grow)
    void *dHugeWs = nullptr, *dHugeCnt = nullptr; size_t capHugeWs = 0, capHugeCnt = 0;      // k_stage2's global-workspace pass
    ncclComm_t comm = nullptr;                                 // infx_stream_comm: this stream's own communicator (several batches in flight per rank)
    void *dLWordOff = nullptr, *dLChars = nullptr, *dLMembers = nullptr, *dLCount = nullptr; size_t capLWordOff = 0, capLChars = 0, capLMembers = 0, capLCount = 0;   // infx_ld1_expand
};

#define S2_HUGE_POOL_U16 (32u << 20)      // token-table pool of k_stage2's over-long-document pass: 64 MB per stream (a 700-word row takes 7.1 KB, a 32 768-token one 332 KB)
static int32_t grow(infx_stream* s, void** p, size_t* cap, size_t need);
static int32_t s2_huge_ready(infx_stream* s) {      // pool + bump counter of the over-long-document pass (allocated at the stream's first Stage-2 launch)
    int32_t rc = grow(s, &s->dHugeWs, &s->capHugeWs, (size_t)S2_HUGE_POOL_U16 * 2); if (rc) return rc;
    rc = grow(s, &s->dHugeCnt, &s->capHugeCnt, 16); if (rc) return rc;
    if (hipMemsetAsync(s->dHugeCnt, 0, 4, s->st) != hipSuccess) return fail(INFX_EHIP, "hipMemsetAsync failed%s");
    return INFX_OK;
}
// hipFree synchronises the whole device: it returns when EVERY stream has drained.  With several sessions in flight the others keep refilling the device, so a
// session that outgrows a workspace in the middle of a stream waited in hipFree until the run ended (measured, round 4: one batch of a 20-batch run 12 % larger
// than its session's earlier batches sat 290 ms in grow() — 310 ms instead of 250 ms for the run); and on a document shard another session's collective may be
// in flight at that moment, spinning until the peer rank's matching collective runs, while on the peer the roles are swapped: neither rank can enqueue what the
// other waits for.  Workspaces are therefore never freed while their stream lives: the outgrown buffer is parked on the stream (freed by infx_stream_destroy);
// growth is geometric, so the parked buffers add up to less than the live one.
static void ws_release(infx_stream* s, void* p);
static int32_t grow(infx_stream* s, void** p, size_t* cap, size_t need) {
    if (need <= *cap) return INFX_OK;
    if (*p) s->parked.push_back(*p);
    // a quarter of headroom: batches of one workload differ by a few per cent, and a reallocation (hipFree + hipMalloc synchronise the device) in a
    // stream's second batch would stall every other stream's batch in flight
    size_t n = std::max(need + need / 4 + 4096, *cap * 2);
    static const bool dbgGrow = getenv("INFX_DEBUG_GROW") != nullptr;
    const auto tg0 = std::chrono::steady_clock::now();
    if (hipMalloc(p, n) != hipSuccess) { *p = nullptr; *cap = 0; return fail(INFX_ENOMEM, "hipMalloc workspace failed%s"); }
    if (dbgGrow) fprintf(stderr, "[infx] grow: %zu -> %zu bytes (need %zu), hipMalloc took %.2f ms\n", *cap, n, need, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count());
    *cap = n; return INFX_OK;
}
static void ws_release(infx_stream* s, void* p) { if (p) s->parked.push_back(p); }
#define GROW(p, cap, need) do { int32_t rc_ = grow(s, (void**)&(p), &(cap), (need)); if (rc_) return rc_; } while (0)      /* `s`: the stream of the enclosing call */

// Stage-2 launches: the register budget of the fast variant is selectable for tuning (INFX_S2_WAVES = 2, 4 or 6 waves per SIMD)
static int s2_waves() { static const int w = [] { const char* e = getenv("INFX_S2_WAVES"); int v = e ? atoi(e) : 0; return (v == 2 || v == 4 || v == 6 || v == 8) ? v : S2_MIN_WAVES; }(); return w; }
// LDS pool of the fast launch: UTF-16 units of document text per 64-candidate workgroup (INFX_S2_POOL overrides; 0 = texts stay in global memory)
static int s2_pool() { static const int w = [] { const char* e = getenv("INFX_S2_POOL"); int v = e ? atoi(e) : S2_POOL_CHARS; return (v < 0 || v > 32768) ? S2_POOL_CHARS : v; }(); return w; }
#define S2_GRID(lds) (ncand + S2_THREADS - 1) / S2_THREADS, S2_THREADS, (lds), s->st
#define S2_FAST_TAIL pool_, (uint16_t*)nullptr, (uint32_t*)nullptr, 0u, (const infx_cov_query_long*)nullptr, s->nLongQ      /* the first launch tells long-query rows apart (and checks their table index) */
// AL: the ALIAS instantiation (stage2.hip.inc: OrdinalIgnoreCase sites compare class representatives) — whenever the corpus or the batch's queries hold one of the 22 alias
// characters; else the plain one (every comparison `==`, exact without them)
#define S2_AL (s->ix->hasAlias || s->batchAlias)
#define S2_LAUNCH_FAST(...) do { const int pool_ = s2_pool(); if (S2_AL) { k_stage2<S2_FASTD, 6, false, false, true><<<S2_GRID(pool_ * 2)>>>(__VA_ARGS__, S2_FAST_TAIL); break; } \
                                 switch (s2_waves()) { case 2: k_stage2<S2_FASTD, 2><<<S2_GRID(pool_ * 2)>>>(__VA_ARGS__, S2_FAST_TAIL); break; case 6: k_stage2<S2_FASTD, 6><<<S2_GRID(pool_ * 2)>>>(__VA_ARGS__, S2_FAST_TAIL); break; \
                                                     case 8: k_stage2<S2_FASTD, 8><<<S2_GRID(pool_ * 2)>>>(__VA_ARGS__, S2_FAST_TAIL); break; default: k_stage2<S2_FASTD, 4><<<S2_GRID(pool_ * 2)>>>(__VA_ARGS__, S2_FAST_TAIL); break; } } while (0)
#define S2_LAUNCH_SLOW(...) do { if (S2_AL) k_stage2<S2_MAXD, 4, false, false, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0); else k_stage2<S2_MAXD, 4><<<S2_GRID(0)>>>(__VA_ARGS__, 0); } while (0)
#define S2_LAUNCH_HUGE(...) do { if (S2_AL) k_stage2<S2_HUGE_TOKENS, 1, true, false, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)s->dHugeWs, (uint32_t*)s->dHugeCnt, (uint32_t)S2_HUGE_POOL_U16); \
                                 else k_stage2<S2_HUGE_TOKENS, 1, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)s->dHugeWs, (uint32_t*)s->dHugeCnt, (uint32_t)S2_HUGE_POOL_U16); } while (0)
// rows of long queries (marked by the first launch): documents up to S2_MAXD words, then the rest through the global-workspace pass; nothing is launched for a batch without long queries.
// The table is consumed: the next Stage-2 call starts without one.
#define S2_LAUNCH_LONGQ(...) do { if (s->nLongQ) { \
    if (S2_AL) k_stage2<S2_MAXD, 2, false, true, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)nullptr, (uint32_t*)nullptr, 0u, (const infx_cov_query_long*)s->dCovQL, s->nLongQ); \
    else k_stage2<S2_MAXD, 2, false, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)nullptr, (uint32_t*)nullptr, 0u, (const infx_cov_query_long*)s->dCovQL, s->nLongQ); \
    (void)hipMemsetAsync(s->dHugeCnt, 0, 4, s->st);      /* the long-query rows get the whole token-table pool, not what the ordinary rows' pool pass left of it */ \
    if (S2_AL) k_stage2<S2_HUGE_TOKENS, 1, true, true, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)s->dHugeWs, (uint32_t*)s->dHugeCnt, (uint32_t)S2_HUGE_POOL_U16, (const infx_cov_query_long*)s->dCovQL, s->nLongQ); \
    else k_stage2<S2_HUGE_TOKENS, 1, true, true><<<S2_GRID(0)>>>(__VA_ARGS__, 0, (uint16_t*)s->dHugeWs, (uint32_t*)s->dHugeCnt, (uint32_t)S2_HUGE_POOL_U16, (const infx_cov_query_long*)s->dCovQL, s->nLongQ); \
    s->nLongQ = 0; s->longAlias = false; } } while (0)
static int32_t s2_huge_ready(infx_stream* s);

// ---- host <-> device transfers through pinned staging -------------------------------------------------------------------
// The C ABI takes plain (pageable) host pointers.  Handing those to hipMemcpyAsync makes the runtime pin/unpin the caller's pages
// per copy, which takes the process-wide address-space lock and stalls every host thread that page-faults meanwhile (observed:
// 50-70 ms stalls of unrelated planning threads with three sessions in flight).  Each stream therefore owns a pinned arena:
// inputs are copied into it and DMA'd from there; outputs are DMA'd into it and copied out after the stream synchronises.
static void* pin_take(infx_stream* s, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& c : s->pins) if (c.cap - c.off >= bytes) { void* p = c.base + c.off; c.off += bytes; return p; }
    size_t cap = std::max<size_t>(bytes, s->pins.empty() ? (size_t)8 << 20 : s->pins.back().cap * 2);
    void* b = nullptr;
    if (hipHostMalloc(&b, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    s->pins.push_back({(char*)b, cap, bytes});
    return b;
}
static int comm_timeout_s() { static const int v = [] { const char* e = getenv("INFX_COMM_TIMEOUT_S"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 120; }(); return v; }
static int32_t stream_sync(infx_stream* s) {
    // blocking wait (interrupt-driven) instead of hipStreamSynchronize's busy poll: a waiting host thread must not burn a core of a
    // CPU-quota-limited container while the planner pool of another session (or another rank's process) needs it
    HIPCHK(hipEventRecord(s->evSync, s->st));
    static const bool pollAlways = [] { const char* e = getenv("INFX_SYNC_POLL"); return e && e[0] == '1'; }();
    if (pollAlways || (s->ix->nranks > 1 && (s->comm || s->ix->comm))) {
        // A stream that carries RCCL collectives waits with a deadline: a peer that died or fell out of step leaves the collective kernel spinning for ever,
        // and an unbounded wait would turn that into a hung job.  INFX_COMM_TIMEOUT_S (default 120) -> INFX_ENCCL.
        const auto t0 = std::chrono::steady_clock::now(); const double limit = (double)comm_timeout_s();
        for (unsigned spin = 0;; spin++) {
            const hipError_t q = hipEventQuery(s->evSync);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(INFX_EHIP, "hipEventQuery: %s", hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                s->pendingOut.clear();
                return fail(INFX_ENCCL, "a collective (or the kernels behind it) did not complete within INFX_COMM_TIMEOUT_S seconds: a peer rank is gone or out of step%s");
            }
            if (spin < 200) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(spin < 2000 ? 20 : 200));
        }
    } else
    HIPCHK(hipEventSynchronize(s->evSync));
    for (auto& d : s->pendingOut) std::memcpy(d.dst, d.src, d.bytes);
    s->pendingOut.clear(); s->unsynced = false;
    return INFX_OK;
}
static int32_t pin_reset(infx_stream* s) {     // start of an API call: the staging of the previous call must have been consumed
    // Outputs still pending here belong to a call that returned before its own synchronisation (an error path): their destinations
    // (often that call's stack variables) are gone, so they are dropped, never copied.
    s->pendingOut.clear();
    if (s->unsynced) { int32_t rc = stream_sync(s); if (rc) return rc; }
    for (auto& c : s->pins) c.off = 0;
    return INFX_OK;
}
static int32_t up(infx_stream* s, void* dst, const void* src, size_t bytes) {
    if (!bytes) return INFX_OK;
    void* p = pin_take(s, bytes); if (!p) return fail(INFX_ENOMEM, "hipHostMalloc staging failed%s");
    std::memcpy(p, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s->st)); s->unsynced = true;
    return INFX_OK;
}
static int32_t down(infx_stream* s, void* dstHost, const void* srcDev, size_t bytes) {   // lands in dstHost at the next stream_sync
    if (!bytes) return INFX_OK;
    void* p = pin_take(s, bytes); if (!p) return fail(INFX_ENOMEM, "hipHostMalloc staging failed%s");
    HIPCHK(hipMemcpyAsync(p, srcDev, bytes, hipMemcpyDeviceToHost, s->st)); s->unsynced = true;
    s->pendingOut.push_back({dstHost, p, bytes});
    return INFX_OK;
}
// Exchange buffers of the sharded stage API may live on the device (e.g. torch CUDA tensors handed to RCCL): those are copied
// device-to-device on the stream instead of being staged through the pinned arena.
static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // plain pageable host memory
    return a.type == hipMemoryTypeDevice;
}
static int32_t upx(infx_stream* s, void* dst, const void* src, size_t bytes) {
    if (!bytes) return INFX_OK;
    if (!is_device_ptr(src)) return up(s, dst, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s->st)); s->unsynced = true;
    return INFX_OK;
}
static int32_t downx(infx_stream* s, void* dst, const void* srcDev, size_t bytes) {
    if (!bytes) return INFX_OK;
    if (!is_device_ptr(dst)) return down(s, dst, srcDev, bytes);
    HIPCHK(hipMemcpyAsync(dst, srcDev, bytes, hipMemcpyDeviceToDevice, s->st)); s->unsynced = true;
    return INFX_OK;
}
// Planning calls (infx_ld1_expand, infx_union_build) run on the stream's high-priority companion: every helper above works on s->st, which is swapped
// for the duration of the call; leave() makes the main stream wait for what the planning stream still has queued (the union write pass).
struct PlanStream {
    infx_stream* s; bool on;
    explicit PlanStream(infx_stream* x) : s(x), on(x->stPlan != nullptr) { if (on) s->st = s->stPlan; }
    int32_t leave() {
        if (!on) return INFX_OK;
        on = false; s->st = s->stMain;
        HIPCHK(hipEventRecord(s->evPlan, s->stPlan));
        HIPCHK(hipStreamWaitEvent(s->stMain, s->evPlan, 0));
        return INFX_OK;
    }
    ~PlanStream() { if (on) s->st = s->stMain; }
};
#define UPX(dst, src, n) do { int32_t rc_ = upx(s, (dst), (src), (n)); if (rc_) return rc_; } while (0)
#define DOWNX(dst, src, n) do { int32_t rc_ = downx(s, (dst), (src), (n)); if (rc_) return rc_; } while (0)
#define UP(dst, src, n) do { int32_t rc_ = up(s, (dst), (src), (n)); if (rc_) return rc_; } while (0)
#define DOWN(dst, src, n) do { int32_t rc_ = down(s, (dst), (src), (n)); if (rc_) return rc_; } while (0)
#define SYNC() do { int32_t rc_ = stream_sync(s); if (rc_) return rc_; } while (0)

template <int R> static void launch_union(infx_stream* s, uint32_t nv, const uint32_t* dBeg, const uint32_t* dEnd, const int32_t* dMembers, uint32_t* dRangeCount,
                                            const unsigned long long* dBase, int32_t* outDocs) {
    uint64_t blocks = (uint64_t)nv * s->ix->d.nRanges;
    k_union<R><<<dim3((unsigned)blocks), dim3(WAVE), 0, s->st>>>(s->ix->d, dBeg, dEnd, dMembers, nv, dRangeCount, dBase, outDocs);
}
static void launch_union_any(infx_stream* s, uint32_t nv, const uint32_t* dBeg, const uint32_t* dEnd, const int32_t* dMembers, uint32_t* dRangeCount,
                             const unsigned long long* dBase, int32_t* outDocs) {
    switch (s->ix->d.R) {
        case 512: launch_union<512>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
        case 1024: launch_union<1024>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
        case 2048: launch_union<2048>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
        case 4096: launch_union<4096>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
        case 8192: launch_union<8192>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
        default: launch_union<16384>(s, nv, dBeg, dEnd, dMembers, dRangeCount, dBase, outDocs); break;
    }
}
// exact replay of the reference's Stage-1 order effects (k_exact1) — on unless INFX_EXACT=0
static bool exact_enabled(const infx_index* ix) { static const bool v = [] { const char* e = getenv("INFX_EXACT"); return !(e && e[0] == '0'); }(); return v && !(ix->cfg.flags & INFX_CFG_NO_EXACT_REPLAY); }
static Arena make_arena(infx_stream* s) {
    return Arena{s->arDoc, s->arScore, s->arCls, s->arMask, s->arExc, (const unsigned long long*)s->dBlockOut, (uint32_t*)s->dBlockOutHi,
                 (uint32_t*)s->dCounts, s->dOverflow, (unsigned long long*)s->dQBytes, (uint32_t*)((unsigned long long*)s->dQBytes + s->lastNqAlloc), (uint2*)s->dDir, s->maskWords};
}
static int acc_stripe() { static const int v = [] { const char* e = getenv("INFX_ACC_STRIPE"); int x = e ? atoi(e) : 0; return (x >= 1 && x <= 64) ? x : 4; }(); return v; }
// LDS8 (stage1.hip.inc) addresses the tf array by raw LDS offset: true only while k_accumulate owns no static __shared__ data, i.e. its dynamic
// LDS starts at address 0.  Checked once per instantiation against the code object; a violation fails the search loudly.
template <int R> static bool acc_lds_layout_ok() {
    static const bool ok = [] {
        hipFuncAttributes a1{}, a2{}, a4{};
        if (hipFuncGetAttributes(&a1, (const void*)k_accumulate<R, 1>) != hipSuccess || hipFuncGetAttributes(&a2, (const void*)k_accumulate<R, 2>) != hipSuccess ||
            hipFuncGetAttributes(&a4, (const void*)k_accumulate<R, 4>) != hipSuccess) return false;
        return a1.sharedSizeBytes == 0 && a2.sharedSizeBytes == 0 && a4.sharedSizeBytes == 0;
    }();
    return ok;
}
// Two kernels share the (query, stripe) pairs of a batch by candidate density (INFX_ACC_SPARSE_T candidates per stripe, default 96 — measured 64: 6.27, 96: 6.18, 128: 6.20, 192: 6.31, 256: 6.41, 512: 6.85 ms for the pair; 0: one kernel, 7.21 ms):
//   k_accumulate_sparse (stage1_sparse.hip.inc)  stripes of <= T candidates: the candidates look their postings up (binary search of the stripe slice, one per lane)
//   k_accumulate        (stage1.hip.inc)         the rest: byte scatter + probe per (list, range) — cheapest per candidate once a range holds dozens of them
static int acc_sparse_t() { static const int v = [] { const char* e = getenv("INFX_ACC_SPARSE_T"); const int x = e ? atoi(e) : 96; return std::max(0, std::min(4096, x)); }(); return v; }
template <int R> static void launch_acc(infx_stream* s, uint32_t nq, Arena ar, int maxT, int useGrp, int maxRef) {
    (void)maxRef;
    if (!acc_lds_layout_ok<R>()) { s->accLayoutBad = true; return; }
    static const int dbgSkip = [] { const char* e = getenv("INFX_ACC_SKIP"); return e ? atoi(e) : 0; }();     // kernel ablation for profiling only
    const int sparseT = s->ix->hPostOff.empty() || s->ix->hPostOff.back() + 4 <= 0xFFFFFFF0ull ? acc_sparse_t() : 0;      // (k_accumulate_sparse indexes a shard's postings with 32 bits)
    const int stripe = sparseT > 0 ? std::max(1, std::min(4, 65536 / R)) : acc_stripe();      // (the two kernels split the same stripes: a power of two)
    const int nStripes = (s->ix->d.nRanges + stripe - 1) / stripe;
    const uint64_t blocks = (uint64_t)nq * 8u * ((nStripes + 7) / 8);     // stripes rounded up to whole groups of 8 (one per XCD)
    const uint8_t* dense = nullptr;
    if (sparseT > 0) {
        if (grow(s, &s->dDense, &s->capDense, (size_t)nq * nStripes)) { s->accLayoutBad = true; return; }
        hipMemsetAsync(s->dDense, 0, (size_t)nq * nStripes, s->st);
        const size_t sw = (size_t)stripe * (R / 32);
        const int maxTl = (std::min(64, std::max(1, maxT)) + 3) & ~3;
        static const int dbgS = [] { const char* e = getenv("INFX_ACCS_SKIP"); return e ? atoi(e) : 0; }();      // kernel ablation for profiling only
        const size_t ldsS = (sw + 64 + 4) * 4 + INFX_NCLASS * 4 + WAVE * 2 + WAVE * 12 + (size_t)WAVE * maxTl + (size_t)64 * maxTl;      // padded bitmap (one pad word per lane) | class histogram | slot table | slice table | hit matrix | slice samples
#define ACCS_LAUNCH(MW_) k_accumulate_sparse<R, MW_><<<dim3((unsigned)blocks), dim3(WAVE), ldsS, s->st>>>(s->ix->d, (const DevQuery*)s->dQueries, (const DevTerm*)s->dTerms, (const int32_t*)s->dExtra, \
            (const int32_t*)s->dUDocs, (const uint32_t*)s->dURange, nq, ar, stripe, useGrp, (uint8_t*)s->dDense, (uint32_t)sparseT, nStripes, maxTl, dbgS)
        if (ar.maskWords == 4) ACCS_LAUNCH(4); else if (ar.maskWords == 2) ACCS_LAUNCH(2); else ACCS_LAUNCH(1);
#undef ACCS_LAUNCH
        dense = (const uint8_t*)s->dDense;
    }
    const size_t lds = (size_t)R + 128 + ((size_t)(R / 32) + 2) * 4 + INFX_NCLASS * 4 + ACC_CAP_DEFAULT * 2;
#define ACC_LAUNCH(MW_) k_accumulate<R, MW_><<<dim3((unsigned)blocks), dim3(WAVE), lds, s->st>>>(s->ix->d, (const DevQuery*)s->dQueries, (const DevTerm*)s->dTerms, (const int32_t*)s->dExtra, \
        (const int32_t*)s->dUDocs, (const uint32_t*)s->dURange, nq, ar, maxT, stripe, useGrp, dbgSkip, (unsigned long long*)s->dStats, dense, nStripes, \
        s->nAccHeavy != 0xFFFFFFFFu ? (const uint32_t*)s->dAccOrder : nullptr, s->nAccHeavy != 0xFFFFFFFFu ? s->nAccHeavy : 0u)
    if (ar.maskWords == 4) ACC_LAUNCH(4); else if (ar.maskWords == 2) ACC_LAUNCH(2); else ACC_LAUNCH(1);
#undef ACC_LAUNCH
    if (dbgSkip & 8) { unsigned long long h[4] = {0, 0, 0, 0}; hipStreamSynchronize(s->st); hipMemcpy(h, s->dStats, 32, hipMemcpyDeviceToHost); hipMemset(s->dStats, 0, 32);
        fprintf(stderr, "[infx] k_accumulate stats: %llu ranges with candidates (%llu blocks), %.2f rounds/range, %.1f candidates/range\n", h[0], (unsigned long long)blocks, h[0] ? (double)h[1] / h[0] : 0.0, h[0] ? (double)h[2] / h[0] : 0.0); }
}

// Longest-queries-first order for k_select's workgroups (k_select_order, stage1.hip.inc); nullptr: query order (batches beyond SEL_ORDER_MAX, INFX_SELECT_LPT=0)
static const uint32_t* select_order(infx_stream* s, uint32_t nq) {
    static const bool off = [] { const char* e = getenv("INFX_SELECT_LPT"); return e && e[0] == '0'; }();
    if (off || nq < 64 || nq > SEL_ORDER_MAX) return nullptr;
    if (grow(s, &s->dSelOrder, &s->capSelOrder, (size_t)nq * 4)) return nullptr;
    k_select_order<<<1, 1024, 0, s->st>>>((const uint32_t*)s->dBlockOutHi, nq, (uint32_t*)s->dSelOrder);
    return (const uint32_t*)s->dSelOrder;
}
#ifdef SEL_PROF
// profiling build only: per-workgroup wall_clock64() stamps (8 words per workgroup: t0 .. t5, two free words) -> span of the launch, the longest workgroups, percentiles
static void wgprof_dump(const char* name, const unsigned long long* dProf, uint32_t n, const char* const ph[5], hipStream_t st) {
    hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)n * 8); hipMemcpy(h.data(), dProf, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t5 = 0; for (uint32_t q = 0; q < n; q++) if (h[q * 8]) { t0 = std::min(t0, h[q * 8]); for (int k = 1; k < 6; k++) t5 = std::max(t5, h[q * 8 + k]); }
    fprintf(stderr, "[%s] span %.1f us\n", name, (t5 - t0) / 100.0);
    auto endOf = [&](uint32_t q) { unsigned long long e = h[q * 8]; for (int k = 1; k < 6; k++) e = std::max(e, h[q * 8 + k]); return e; };
    std::vector<uint32_t> ord(n); for (uint32_t q = 0; q < n; q++) ord[q] = q;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return endOf(a) - h[a * 8] > endOf(b) - h[b * 8]; });
    auto us = [&](uint32_t q, int a, int b) { const unsigned long long x = h[q * 8 + a], y = h[q * 8 + b]; return (x && y) ? (double)(y - x) / 100.0 : -1.0; };
    auto line = [&](const char* tag, uint32_t q) { fprintf(stderr, "[%s] %s wg %u w6 %llu w7 %llu: start %.1f total %.1f | %s %.1f %s %.1f %s %.1f %s %.1f %s %.1f\n", name, tag, q, h[q * 8 + 6], h[q * 8 + 7], (h[q * 8] - t0) / 100.0,
                                                        (endOf(q) - h[q * 8]) / 100.0, ph[0], us(q, 0, 1), ph[1], us(q, 1, 2), ph[2], us(q, 2, 3), ph[3], us(q, 3, 4), ph[4], us(q, 4, 5)); };
    for (int i = 0; i < 10 && i < (int)n; i++) line("top", ord[i]);
    line("p50", ord[n / 2]); line("p90", ord[n / 10]); line("p99", ord[n / 100]);
    std::vector<unsigned long long> en(n); for (uint32_t q = 0; q < n; q++) en[q] = endOf(q) - t0; std::sort(en.begin(), en.end());
    fprintf(stderr, "[%s] ends: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us\n", name, en[n / 10] / 100.0, en[n / 2] / 100.0, en[n * 9 / 10] / 100.0, en[n * 99 / 100] / 100.0, en[n - 1] / 100.0);
}
#endif
// The largest queries of the batch swept by many workgroups in front of k_select (k_selg_hist / k_selg_gather, stage1.hip.inc).  INFX_SEL_GIANT_MIN: rows from which a
// query is one (default 65536; 0: off — k_select sweeps every query itself).  Needs the longest-first order (its first SELG_MAX entries are the candidates).
static uint32_t giant_min_rows() { static const uint32_t v = [] { const char* e = getenv("INFX_SEL_GIANT_MIN"); return e ? (uint32_t)std::max(0, atoi(e)) : 65536u; }(); return v; }
static SelGiant select_giants(infx_stream* s, Arena ar, uint32_t nq, const uint32_t* order) {
    const uint32_t minRows = giant_min_rows();
    SelGiant G{}; if (!order || !minRows || !s->nBoundGiants || s->maxQueryBound < 4ull * minRows) return G;      // (three launches + two memsets cost ~50 us: not for a batch whose largest query k_select sweeps in that time)
    const size_t head = (size_t)SELG_MAX * 2 * 4096 * 4 + (size_t)SELG_MAX * 4 * 5, total = head + (size_t)SELG_MAX * SEL_CAP * 8;
    if (grow(s, &s->dSelG, &s->capSelG, total)) return G;
    const uint32_t nSlots = std::min<uint32_t>(SELG_MAX, std::min(nq, s->nBoundGiants));
    hipMemsetAsync(s->dSelG, 0, (size_t)nSlots * 2 * 4096 * 4, s->st);                                             // the histograms of the slots in use
    hipMemsetAsync((char*)s->dSelG + (size_t)SELG_MAX * 2 * 4096 * 4, 0, (size_t)SELG_MAX * 4 * 5, s->st);       // count | left | mode | cut | shift
    G.hist = (uint32_t*)s->dSelG; G.count = G.hist + (size_t)SELG_MAX * 2 * 4096; G.left = G.count + SELG_MAX; G.mode = G.left + SELG_MAX; G.cut = G.mode + SELG_MAX; G.shift = G.cut + SELG_MAX;
    G.keys = (unsigned long long*)((char*)s->dSelG + head); G.minRows = minRows;
    // (a query holds at most its row bound: the slots and partitions that can be needed are known on the host — a batch of small queries launches a handful of workgroups)
    const dim3 grid((unsigned)std::min<uint64_t>(SELG_PARTS, (s->maxQueryBound + SELG_PART_MIN - 1) / SELG_PART_MIN), nSlots);
    k_selg_hist<<<grid, SEL_THREADS, 0, s->st>>>(ar, (const SelRule*)s->dRules, order, nq, G);
    k_selg_cut<<<grid.y, 256, 0, s->st>>>(ar, (const SelRule*)s->dRules, order, nq, G);
    k_selg_gather<<<grid, SEL_THREADS, 0, s->st>>>(ar, (const SelRule*)s->dRules, order, nq, G);
    return G;
}
// k_exact1 behind k_select: unsharded indexes only (the reference's chunking follows GLOBAL 65 536-id containers and its heap is sequential
// over the whole corpus; document shards keep k_select's deterministic (score, doc id) cut)
static bool exact_slow_only() { static const bool v = [] { const char* e = getenv("INFX_EXACT_SLOW"); return e && e[0] == '1'; }(); return v; }
// the full heap of the replay lives in registers (RHeap, exact3.hip.inc) when its entries fit the packing: internal ids below 2^30, depth <= 512;
// INFX_EX_HEAP_LDS=1 keeps the LDS heap (A/B measurements, parity tooling)
static int ex_heap_in_regs(const infx_index* ix, int depth) {
    static const bool off = [] { const char* e = getenv("INFX_EX_HEAP_LDS"); return e && e[0] == '1'; }();
    const long long total = std::max<long long>(ix->d.totalDocs, (long long)ix->d.docBase + ix->d.N);
    return (!off && depth <= EXS_MAXDEPTH && total < (1ll << 30)) ? 1 : 0;
}
// After the batch's k_select (with replay flags): a flagged query beyond two mask words (65 .. 128 reference terms) goes to the sequential replay (k_exact1 reads
// its four-word masks from the arena; the parallel replay's kernels keep two words in registers) — flag 1 -> 2, what k_ex_theta / k_ex_heap write for a query they hand over.
__global__ void k_mark_wide(uint32_t* __restrict__ flags, const uint32_t* __restrict__ list, uint32_t n) { for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) if (flags[list[i]] == 1u) flags[list[i]] = 2u; }
static int32_t mark_wide_queries(infx_stream* s, int depth) {
    if (!s->nWide || exact_slow_only() || depth > EXS_MAXDEPTH) return INFX_OK;      // (no parallel replay: k_exact1 takes every flagged query as it is)
    k_mark_wide<<<1, 64, 0, s->st>>>((uint32_t*)s->dExactFlag, (const uint32_t*)s->dWideQ, s->nWide);
    HIPCHK(hipGetLastError());
    return INFX_OK;
}
static bool exact_possible(infx_stream* s) { return exact_enabled(s->ix) && s->maskWords > 0 && s->ix->nranks == 1 && s->ix->d.docBase == 0; }
// chunk table of the parallel replay: every query needs at most (containers + reserved rows / 4096 + 2) entries
static int32_t exact_chunk_tables(infx_stream* s, uint32_t nq, ExBufs& xb) {
    infx_index* ix = s->ix;
    const int rpc = 65536 / ix->d.R, nCont = (ix->d.nRanges + rpc - 1) / rpc;
    const size_t cap = (size_t)nq * (nCont + 2) + s->arBound / EX_CHUNK + 16;
    if (cap > 0x7FFFFFF0ull) return fail(INFX_ECAPACITY, "exact-replay chunk table too large; split the batch%s");
    GROW(s->exChunks, s->capExChunks, cap * sizeof(ExChunk));
    GROW(s->exTasks, s->capExTasks, cap * 3 * 4);
    GROW(s->exQueries, s->capExQueries, (size_t)nq * sizeof(ExQuery));
    GROW(s->exContEnd, s->capExContEnd, (size_t)nq * nCont * 4);
    s->exChunkCap = (uint32_t)cap;
    HIPCHK(hipMemsetAsync(s->exCounters, 0, 32, s->st));
    xb = ExBufs{s->exCand, s->exOut, s->arExc, (ExChunk*)s->exChunks, s->exChunkCap, (ExQuery*)s->exQueries, s->exCounters, (uint32_t*)s->exTasks, (uint32_t*)s->exTasks + cap,
                (uint32_t*)s->exTasks + 2 * cap, (uint32_t*)s->exContEnd, nullptr, nullptr};
    return INFX_OK;
}
// The scan and the chunk pass of the parallel replay on stream `st` (k_ex_cand, k_ex_theta, k_ex_chunk<1|4|16>).  With an auxiliary stream the two
// workgroup variants of k_ex_chunk run beside the one-wave variant (disjoint chunks; each launch ends in a tail of a few long tasks).
static int32_t launch_scan_and_chunks(infx_stream* s, uint32_t nq, Arena ar, ExBufs xb, const float* prior, bool useAux) {
    infx_index* ix = s->ix;
    const int rpc = 65536 / ix->d.R, nCont = (ix->d.nRanges + rpc - 1) / rpc;
    if (nCont > 65535) return fail(INFX_ECAPACITY, "exact replay: more than 65535 containers of 65536 documents in one shard%s");
    k_ex_walk<false><<<dim3(nq, nCont), WAVE, 0, s->st>>>(ix->d, ar, (const SelRule*)s->dRules, (const uint32_t*)s->dExactFlag, xb, nCont);
    k_ex_prefix<<<nq, EXS_THREADS, 0, s->st>>>((const uint32_t*)s->dExactFlag, xb, nCont);
    k_ex_walk<true><<<dim3(nq, nCont), WAVE, 0, s->st>>>(ix->d, ar, (const SelRule*)s->dRules, (const uint32_t*)s->dExactFlag, xb, nCont);
    k_ex_theta<<<nq, WAVE, 0, s->st>>>(ar, (const SelRule*)s->dRules, (uint32_t*)s->dExactFlag, xb, prior, nCont);
    HIPCHK(hipEventRecord(s->evXa, s->st));
    hipStream_t bigSt = s->st;
    if (useAux && s->stAux && s->st == s->stMain) { HIPCHK(hipStreamWaitEvent(s->stAux, s->evXa, 0)); bigSt = s->stAux; }
    k_ex_chunk<16><<<EXP_GRID16, 16 * WAVE, 0, bigSt>>>(ix->d, (const DevQuery*)s->dQueries, (const DevRefTerm*)s->dRefTerms, ar, xb, xb.tasksBig, 4, ix->avgdl);
    k_ex_chunk<4><<<EXP_GRID4, 4 * WAVE, 0, bigSt>>>(ix->d, (const DevQuery*)s->dQueries, (const DevRefTerm*)s->dRefTerms, ar, xb, xb.tasksMid, 2, ix->avgdl);
    if (bigSt != s->st) HIPCHK(hipEventRecord(s->evJoin, bigSt));
    k_ex_chunk<1><<<EXP_GRID1, WAVE, 0, s->st>>>(ix->d, (const DevQuery*)s->dQueries, (const DevRefTerm*)s->dRefTerms, ar, xb, xb.tasksSmall, 1, ix->avgdl);
    if (bigSt != s->st) HIPCHK(hipStreamWaitEvent(s->st, s->evJoin, 0));
    HIPCHK(hipGetLastError());
    return INFX_OK;
}
static int32_t exact1_lds_ready(infx_index* ix, int MW, size_t* ldsOut) {
    const int depthCap = ix->cfg.max_depth;
    const size_t lds1 = (size_t)(EX_CHUNK + EX_THREADS) * (MW <= 2 ? MW : 1) * 8 + (size_t)(EX_CHUNK + EX_THREADS) * 4 + (size_t)EX_CHUNK * 4 + (size_t)depthCap * 8 +
                        (INFX_MAX_QUERY_TERMS + 1) * 4 + 130 * 4 + 129 * 4 + 8 * 4 + (size_t)EX_CHUNK * 2 + 64;
    static std::mutex mu; static size_t attr1 = 0;
    { std::lock_guard<std::mutex> lk(mu);
      if (lds1 > attr1) { HIPCHK(hipFuncSetAttribute((const void*)k_exact1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)); attr1 = lds1; } }
    *ldsOut = lds1; return INFX_OK;
}
static int32_t enqueue_exact(infx_stream* s, uint32_t nq, int stride) {
    infx_index* ix = s->ix;
    const int MW = s->maskWords, depthCap = ix->cfg.max_depth;
    static const bool slowOnly = exact_slow_only();      // parity tooling: k_exact1 for every flagged query
    size_t lds1 = 0; { int32_t rc_ = exact1_lds_ready(ix, MW, &lds1); if (rc_) return rc_; }
    Arena ar = make_arena(s);
    HIPCHK(hipMemsetAsync(s->dExactStat, 0, 16, s->st));
    HIPCHK(hipEventRecord(s->evX0, s->st));
    const bool fast = !slowOnly && depthCap <= EXS_MAXDEPTH;
    if (fast) {
        ExBufs xb; { int32_t rc_ = exact_chunk_tables(s, nq, xb); if (rc_) return rc_; }
        static const bool exProf = getenv("INFX_EXACT_PROF") != nullptr;     // per-query cycle counters of the replay kernels (profiling only)
        if (exProf) { GROW(s->dExProf, s->capExProf, ((size_t)nq * EXP_WORDS + (EXP_GRID1 + EXP_GRID4 + EXP_GRID16) * 4) * 8); HIPCHK(hipMemsetAsync(s->dExProf, 0, ((size_t)nq * EXP_WORDS + (EXP_GRID1 + EXP_GRID4 + EXP_GRID16) * 4) * 8, s->st)); }
        xb.prof = exProf ? (unsigned long long*)s->dExProf : nullptr; xb.profChunk = exProf ? (unsigned long long*)s->dExProf + (size_t)nq * EXP_WORDS : nullptr;
        { int32_t rc_ = launch_scan_and_chunks(s, nq, ar, xb, nullptr, true); if (rc_) return rc_; }
        HIPCHK(hipEventRecord(s->evXb, s->st));
        k_ex_heap<<<nq, WAVE, 0, s->st>>>(ar, (const SelRule*)s->dRules, (uint32_t*)s->dExactFlag, xb, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, stride, s->dExactStat,
                                          exProf ? (unsigned long long*)s->dExProf : nullptr, ex_heap_in_regs(ix, depthCap));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->evXc, s->st)); s->timedReplayParts = true;
        if (exProf) {       // profiling only: per-query counters of the replay kernels -> mean / p50 / p90 / max and the slowest queries
            hipStreamSynchronize(s->st);
            std::vector<unsigned long long> h((size_t)nq * EXP_WORDS); hipMemcpy(h.data(), s->dExProf, h.size() * 8, hipMemcpyDeviceToHost);
            uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpy(cnt, s->exCounters, 32, hipMemcpyDeviceToHost);
            static const char* names[EXP_WORDS] = {"heap.cycles", "heap.heap_cycles", "heap.rows", "heap.ops", "heap.chunks", "scan.cycles", "theta.insert_cycles", "theta.inserts", "cand.batches", "cand.rows", "cand.cands",
                                                   "theta.cycles", "-", "-", "-", "-"};
            std::vector<uint32_t> qs; for (uint32_t q = 0; q < nq; q++) if (h[(size_t)q * EXP_WORDS + 0] || h[(size_t)q * EXP_WORDS + 5]) qs.push_back(q);
            fprintf(stderr, "[infx] exact replay profile: %zu replayed queries of %u; chunk tasks 1-wave %u 4-wave %u 16-wave %u (chunk slots reserved %u)\n", qs.size(), nq, cnt[1], cnt[2], cnt[4], cnt[0]);
            if (!qs.empty()) for (int w = 0; w < 12; w++) {
                std::vector<unsigned long long> v; for (uint32_t q : qs) v.push_back(h[(size_t)q * EXP_WORDS + w]);
                std::sort(v.begin(), v.end()); double sum = 0; for (auto x : v) sum += (double)x;
                fprintf(stderr, "[infx]   %-20s mean %12.0f  p50 %12llu  p90 %12llu  p99 %12llu  max %12llu\n", names[w], sum / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
            }
            for (int which : {0, 5}) {
                std::sort(qs.begin(), qs.end(), [&](uint32_t a, uint32_t b) { return h[(size_t)a * EXP_WORDS + which] > h[(size_t)b * EXP_WORDS + which]; });
                for (size_t i = 0; i < std::min<size_t>(3, qs.size()); i++) { fprintf(stderr, "[infx]   slowest by %s: q %u:", names[which], qs[i]); for (int w = 0; w < 13; w++) fprintf(stderr, " %llu", h[(size_t)qs[i] * EXP_WORDS + w]); fprintf(stderr, "\n"); }
            }
            std::vector<unsigned long long> ck((EXP_GRID1 + EXP_GRID4 + EXP_GRID16) * 4); hipMemcpy(ck.data(), (char*)s->dExProf + (size_t)nq * EXP_WORDS * 8, ck.size() * 8, hipMemcpyDeviceToHost);
            const int g0[4] = {0, EXP_GRID1, EXP_GRID1 + EXP_GRID4, EXP_GRID1 + EXP_GRID4 + EXP_GRID16};
            for (int w = 0; w < 3; w++) {
                unsigned long long n = 0, cy = 0, mx = 0, tl = 0, wgMax = 0;
                for (int b = g0[w]; b < g0[w + 1]; b++) { n += ck[b * 4]; cy += ck[b * 4 + 1]; mx = std::max(mx, ck[b * 4 + 2]); tl += ck[b * 4 + 3]; wgMax = std::max(wgMax, ck[b * 4 + 1]); }
                fprintf(stderr, "[infx]   k_ex_chunk<%d>: %llu tasks, cycles per task mean %.0f max %llu, busiest workgroup %llu cycles, tiles mean %.1f\n", w == 0 ? 1 : (w == 1 ? 4 : 16), n, n ? (double)cy / n : 0.0, mx, wgMax, n ? (double)tl / n : 0.0);
            }
        }
    }
    k_exact1<<<nq, EX_THREADS, lds1, s->st>>>(ix->d, (const DevQuery*)s->dQueries, (const DevRefTerm*)s->dRefTerms, ar, (const SelRule*)s->dRules, (const uint32_t*)s->dExactFlag,
                                             fast ? 2u : 1u, ix->avgdl, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, stride, depthCap, s->dExactStat, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evX1, s->st)); s->timedReplay = true;
    DOWN(s->lastExact2, s->dExactStat, 8);          // [0] replayed queries, [1] of them through the sequential k_exact1 fallback; lands at the caller's synchronisation
    DOWN(s->lastFlagWhy, s->dExactStat + 4, 16);
    return INFX_OK;
}

// An all-gather of one packed block per rank ([part 0 | part 1 | ...], `stride` bytes apart) unpacked into per-part arrays of nranks consecutive pieces:
// three exchanges of a batch (first-pass lists, their counts, the best scores left out) travel as ONE collective.
struct UnpackArgs { const uint32_t* src; uint64_t strideW; int32_t nranks, nparts; uint64_t offW[4], lenW[4]; uint32_t* dst[4]; };
__global__ void k_unpack(UnpackArgs a) {
    uint64_t per = 0; for (int p = 0; p < a.nparts; p++) per += a.lenW[p];
    const uint64_t total = per * (uint64_t)a.nranks;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / per; uint64_t w = i % per; int p = 0;
        while (w >= a.lenW[p]) { w -= a.lenW[p]; p++; }
        a.dst[p][r * a.lenW[p] + w] = a.src[r * a.strideW + a.offW[p] + w];
    }
}
// ---- RCCL (dlopen) ---------------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    bool ok = false; const char* why = "";
    ncclResult_t (*getId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*init)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*allreduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allgather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
    const char* (*errstr)(ncclResult_t) = nullptr;
};
static RcclApi& rccl_api() {
    static RcclApi a = [] {
        RcclApi r;
        // a process that already holds an RCCL (e.g. torch's) keeps using that copy
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { r.why = "librccl.so not found"; return r; }
        r.getId = (decltype(r.getId))dlsym(h, "ncclGetUniqueId"); r.init = (decltype(r.init))dlsym(h, "ncclCommInitRank");
        r.allreduce = (decltype(r.allreduce))dlsym(h, "ncclAllReduce"); r.allgather = (decltype(r.allgather))dlsym(h, "ncclAllGather");
        r.destroy = (decltype(r.destroy))dlsym(h, "ncclCommDestroy"); r.errstr = (decltype(r.errstr))dlsym(h, "ncclGetErrorString");
        r.ok = r.getId && r.init && r.allreduce && r.allgather && r.destroy && r.errstr;
        if (!r.ok) r.why = "librccl.so lacks an expected symbol";
        return r;
    }();
    return a;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(INFX_ENCCL, #x ": %s", rccl_api().errstr(r_)); } while (0)

extern "C" {

const char* infx_last_error(void) { return g_err.c_str(); }

int32_t infx_create(const infx_config* cfg, infx_index** out) {
    if (!cfg || !out) return fail(INFX_EINVAL, "null argument%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(INFX_EHIP, "no HIP device available (%s) — the GPU path is mandatory, there is no CPU fallback", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(INFX_EINVAL, "bad device ordinal%s");
    HIPCHK(enter_device(cfg->device));
    infx_index* ix = new infx_index();
    ix->cfg = *cfg;
    int R = cfg->range_docs ? cfg->range_docs : 1024;
    if (R != 512 && R != 1024 && R != 2048 && R != 4096 && R != 8192 && R != 16384) { delete ix; return fail(INFX_EINVAL, "range_docs must be a power of two in [512, 16384]%s"); }
    ix->d.R = R; ix->d.rshift = __builtin_ctz(R);
    if (ix->cfg.max_depth <= 0) ix->cfg.max_depth = 500;
    if (ix->cfg.max_depth > SEL_CAP / 2) { delete ix; return fail(INFX_EINVAL, "max_depth too large%s"); }
    *out = ix;
    return INFX_OK;
}

void infx_destroy(infx_index* ix) {
    if (!ix) return;
    hipSetDevice(ix->cfg.device);
    if (ix->comm && rccl_api().ok) rccl_api().destroy(ix->comm);
    for (void* p : ix->allocs) hipFree(p);
    delete ix->lk;
    delete ix;
}

int32_t infx_set_deleted(infx_index* ix, uint32_t total, const uint8_t* deleted) {
    if (!ix) return fail(INFX_EINVAL, "null argument%s");
    if (!ix->haveDocs) return fail(INFX_EINVAL, "infx_set_deleted before infx_upload_docs%s");
    if (deleted && total != (uint32_t)ix->d.totalDocs) return fail(INFX_EINVAL, "infx_set_deleted: one flag per global internal id (total_docs) is required%s");
    HIPCHK(enter_device(ix->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    if (!deleted) { ix->d.deleted = nullptr; return INFX_OK; }
    bool any = false; for (uint32_t i = 0; i < total && !any; i++) any = deleted[i] != 0;
    if (!any) { ix->d.deleted = nullptr; return INFX_OK; }           // the kernels test the pointer: no per-row gather while nothing is deleted
    if (!ix->dDeleted) HIPCHK(dalloc(ix, &ix->dDeleted, (size_t)total));
    HIPCHK(hipMemcpy(ix->dDeleted, deleted, (size_t)total, hipMemcpyHostToDevice));
    ix->d.deleted = ix->dDeleted;
    return INFX_OK;
}

int32_t infx_upload_docs(infx_index* ix, uint32_t N, const float* doc_len, float avgdl, const int64_t* doc_key, const uint8_t* deleted,
                         const uint64_t* text_offs, const uint16_t* text) {
    if (!ix || !doc_len || !doc_key) return fail(INFX_EINVAL, "null argument%s");
    if (deleted && (ix->d.docBase != 0 || (ix->d.totalDocs != 0 && ix->d.totalDocs != (int32_t)N)))
        return fail(INFX_EINVAL, "infx_upload_docs: deleted[] is for unsharded indexes; a shard takes the global flags through infx_set_deleted%s");
    HIPCHK(enter_device(ix->cfg.device));
    float *dLen = nullptr, *dNorm = nullptr; int64_t* dKey = nullptr; uint64_t* dTO = nullptr; uint16_t* dTx = nullptr;
    HIPCHK(dalloc(ix, &dNorm, N)); HIPCHK(dalloc(ix, &dKey, N)); HIPCHK(dalloc(ix, &dLen, N));
    HIPCHK(hipMemcpy(dLen, doc_len, (size_t)N * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dKey, doc_key, (size_t)N * 8, hipMemcpyHostToDevice));
    float a = avgdl > 0.f ? avgdl : 1.f;
    float bDivAvg = 0.75f / a;     // Vector256.Create(b / avgdl), Bm25Scorer.cs:390
    if (N) k_doc_norm<<<(N + 255) / 256, 256>>>(dLen, dNorm, (int)N, bDivAvg);
    HIPCHK(hipDeviceSynchronize());
    if (text_offs && text) {
        uint64_t tot = text_offs[N];
        HIPCHK(dalloc(ix, &dTO, (size_t)N + 1)); HIPCHK(dalloc(ix, &dTx, (size_t)tot));
        HIPCHK(hipMemcpy(dTO, text_offs, ((size_t)N + 1) * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dTx, text, (size_t)tot * 2, hipMemcpyHostToDevice));
        if (tot) {      // OrdinalIgnoreCase alias characters of the corpus (stage2.hip.inc): none in most corpora — then every Stage-2 comparison is `==`
            unsigned long long* dCnt = nullptr; unsigned long long hCnt = 0;
            HIPCHK(hipMalloc((void**)&dCnt, 8)); HIPCHK(hipMemset(dCnt, 0, 8));
            k_count_alias<<<(unsigned)std::min<uint64_t>((tot + 255) / 256, 65536), 256>>>(dTx, (unsigned long long)tot, dCnt);
            HIPCHK(hipGetLastError()); HIPCHK(hipMemcpy(&hCnt, dCnt, 8, hipMemcpyDeviceToHost)); hipFree(dCnt);
            ix->hasAlias = hCnt != 0;
        }
    }
    if (ix->cfg.range_docs == 0) {   // default: wide ranges for large shards (fewer, longer per-range posting slices; measured best at 10M docs)
        int R = N >= (4u << 20) ? 8192 : N >= (1u << 20) ? 4096 : N >= (1u << 18) ? 2048 : 1024;
        ix->d.R = R; ix->d.rshift = __builtin_ctz(R);
    }
    ix->d.N = (int32_t)N; ix->d.docNorm = dNorm; ix->d.docLen = dLen; ix->d.docKey = dKey; if (!ix->d.docKeyAll) ix->d.docKeyAll = dKey; ix->d.textOff = dTO; ix->d.text = dTx;
    ix->d.nRanges = (int32_t)(((uint64_t)N + ix->d.R - 1) >> ix->d.rshift);
    if (ix->d.nRanges == 0) ix->d.nRanges = 1;
    if (ix->d.totalDocs == 0) ix->d.totalDocs = (int32_t)N;
    ix->avgdl = avgdl; ix->haveDocs = true;
    if (deleted) return infx_set_deleted(ix, N, deleted);
    return INFX_OK;
}

int32_t infx_upload_postings(infx_index* ix, uint32_t T, const uint64_t* offs, const int32_t* doc_ids, const uint8_t* tf, const int32_t* df) {
    if (!ix || !offs || !df) return fail(INFX_EINVAL, "null argument%s");
    if (!ix->haveDocs) return fail(INFX_EINVAL, "infx_upload_docs must precede infx_upload_postings (range count)%s");
    HIPCHK(enter_device(ix->cfg.device));
    uint64_t P = offs[T];
    uint64_t* dOff = nullptr; int32_t* dDoc = nullptr; uint8_t* dW = nullptr;
    // padded layout (k_pad_lists): list t lives at [off2[t], off2[t] + len_t), the rest of its 4-aligned slot holds sentinels
    std::vector<uint64_t> off2((size_t)T + 1); off2[0] = 0;
    for (uint32_t t = 0; t < T; t++) off2[t + 1] = off2[t] + ((offs[t + 1] - offs[t] + 3) & ~(uint64_t)3);
    const uint64_t P2 = off2[T];
    HIPCHK(dalloc(ix, &dOff, (size_t)T + 1)); HIPCHK(dalloc(ix, &dDoc, (size_t)P2 + 4)); HIPCHK(dalloc(ix, &dW, (size_t)P2 + 4));
    HIPCHK(hipMemcpy(dOff, off2.data(), ((size_t)T + 1) * 8, hipMemcpyHostToDevice));
    // packed postings: (doc << 8) | tf in one word — the tf of a candidate posting needs no second (gather) access and a posting costs 4 B
    // instead of 5 B of traffic; possible while shard-local doc ids stay below 2^24 - 1 (sentinel 0xFFFFFFFF decodes to doc 2^24 - 1)
    // (every doc id a range boundary can name must stay below the sentinel's doc field, so the padded range count is what matters)
    ix->d.packed = ((uint64_t)ix->d.nRanges * (uint64_t)ix->d.R <= 0xFFFFFFull && !getenv("INFX_UNPACKED")) ? 1 : 0;
    HIPCHK(hipMemset(dDoc, ix->d.packed ? 0xFF : 0x7F, ((size_t)P2 + 4) * 4)); HIPCHK(hipMemset(dW, 0, (size_t)P2 + 4));
    if (P) {
        uint64_t* tOff = nullptr; int32_t* tDoc = nullptr; uint8_t* tW = nullptr;
        HIPCHK(hipMalloc((void**)&tOff, ((size_t)T + 1) * 8)); HIPCHK(hipMalloc((void**)&tDoc, (size_t)P * 4)); HIPCHK(hipMalloc((void**)&tW, (size_t)P));
        HIPCHK(hipMemcpy(tOff, offs, ((size_t)T + 1) * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(tDoc, doc_ids, (size_t)P * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(tW, tf, (size_t)P, hipMemcpyHostToDevice));
        k_pad_lists<<<(unsigned)((P + 255) / 256), 256>>>(tOff, dOff, T, P, tDoc, tW, dDoc, dW, ix->d.packed);
        HIPCHK(hipDeviceSynchronize());
        hipFree(tOff); hipFree(tDoc); hipFree(tW);
    }
    ix->hPostOff = off2; ix->hDf.assign(df, df + T);
    ix->hPostLen.resize(T); for (uint32_t t = 0; t < T; t++) ix->hPostLen[t] = offs[t + 1] - offs[t];
    // Range skip tables (nRanges+1 offsets per list): a workgroup finds a list's slice inside its doc range with one lookup
    // instead of two ~log2(df)-step binary searches whose dependent L2 round trips every (query, range) workgroup would pay.
    // HBM is plentiful, so every list of >= 64 postings gets one as long as the tables stay below 2^31 entries (8 GiB);
    // otherwise the threshold doubles until they fit.
    int nR = ix->d.nRanges;
    std::vector<uint32_t> skipIdx(T, 0xFFFFFFFFu), skipTerms;
    uint64_t per = (uint64_t)nR + 1;
    uint64_t thr = 64;
    for (;; thr *= 2) {
        uint64_t n = 0; for (uint32_t t = 0; t < T; t++) if (offs[t + 1] - offs[t] >= thr) n++;
        if (n * per <= 0x7FFFFFFFull) break;
    }
    for (uint32_t t = 0; t < T; t++) {
        uint64_t len = offs[t + 1] - offs[t];
        if (len >= thr) { skipIdx[t] = (uint32_t)(skipTerms.size() * per); skipTerms.push_back(t); }
    }
    uint32_t *dSkipIdx = nullptr, *dSkipTbl = nullptr, *dSkipTerms = nullptr;
    HIPCHK(dalloc(ix, &dSkipIdx, T)); HIPCHK(dalloc(ix, &dSkipTbl, skipTerms.size() * per));
    HIPCHK(hipMemcpy(dSkipIdx, skipIdx.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    if (!skipTerms.empty()) {
        if (skipTerms.size() * per > 0xFFFFFFF0ull) return fail(INFX_EINVAL, "skip table exceeds 32-bit indexing%s");
        HIPCHK(hipMalloc((void**)&dSkipTerms, skipTerms.size() * 4));
        HIPCHK(hipMemcpy(dSkipTerms, skipTerms.data(), skipTerms.size() * 4, hipMemcpyHostToDevice));
        uint64_t tot = skipTerms.size() * per;
        k_build_skip<<<(unsigned)((tot + 255) / 256), 256>>>(dOff, dDoc, dSkipTerms, (uint32_t)skipTerms.size(), dSkipTbl, nR, ix->d.rshift, ix->d.packed);
        HIPCHK(hipDeviceSynchronize());
        hipFree(dSkipTerms);
    }
    ix->hSkipIdx = std::move(skipIdx);
    ix->d.T = (int32_t)T; ix->d.postOff = dOff; ix->d.postDoc = dDoc; ix->d.postW = dW; ix->d.skipIdx = dSkipIdx; ix->d.skipTbl = dSkipTbl;
    ix->havePostings = true;
    return INFX_OK;
}

int32_t infx_upload_prefix_docsets(infx_index* ix, uint32_t nsets, const uint64_t* offs, const int32_t* docs) {
    if (!ix || (nsets && !offs)) return fail(INFX_EINVAL, "null argument%s");
    HIPCHK(enter_device(ix->cfg.device));
    uint64_t* dOff = nullptr; int32_t* dDocs = nullptr;
    uint64_t tot = nsets ? offs[nsets] : 0;
    HIPCHK(dalloc(ix, &dOff, (size_t)nsets + 1)); HIPCHK(dalloc(ix, &dDocs, (size_t)tot));
    if (nsets) { HIPCHK(hipMemcpy(dOff, offs, ((size_t)nsets + 1) * 8, hipMemcpyHostToDevice)); }
    else { uint64_t z = 0; HIPCHK(hipMemcpy(dOff, &z, 8, hipMemcpyHostToDevice)); }
    if (tot) HIPCHK(hipMemcpy(dDocs, docs, (size_t)tot * 4, hipMemcpyHostToDevice));
    {   // range skip tables for the DocSets too (same layout as the posting lists')
        if (!ix->haveDocs) return fail(INFX_EINVAL, "infx_upload_docs must precede infx_upload_prefix_docsets (range count)%s");
        const int nR = ix->d.nRanges; const uint64_t per = (uint64_t)nR + 1;
        std::vector<uint32_t> psSkip(std::max<uint32_t>(nsets, 1), 0xFFFFFFFFu), sets;
        for (uint32_t k = 0; k < nsets; k++) if (offs[k + 1] - offs[k] >= 64 && (sets.size() + 1) * per <= 0x7FFFFFFFull) { psSkip[k] = (uint32_t)(sets.size() * per); sets.push_back(k); }
        uint32_t *dPsSkip = nullptr, *dPsTbl = nullptr, *dSets = nullptr;
        HIPCHK(dalloc(ix, &dPsSkip, psSkip.size())); HIPCHK(dalloc(ix, &dPsTbl, sets.size() * per));
        HIPCHK(hipMemcpy(dPsSkip, psSkip.data(), psSkip.size() * 4, hipMemcpyHostToDevice));
        if (!sets.empty()) {
            HIPCHK(hipMalloc((void**)&dSets, sets.size() * 4));
            HIPCHK(hipMemcpy(dSets, sets.data(), sets.size() * 4, hipMemcpyHostToDevice));
            uint64_t n = sets.size() * per;
            k_build_skip<<<(unsigned)((n + 255) / 256), 256>>>(dOff, dDocs, dSets, (uint32_t)sets.size(), dPsTbl, nR, ix->d.rshift, 0);
            HIPCHK(hipDeviceSynchronize());
            hipFree(dSets);
        }
        ix->d.psSkip = dPsSkip; ix->d.psSkipTbl = dPsTbl;
    }
    ix->d.psOff = dOff; ix->d.psDocs = dDocs; ix->d.nSets = nsets;
    if (nsets) ix->hPsOff.assign(offs, offs + nsets + 1); else ix->hPsOff.assign(1, 0);
    return INFX_OK;
}

int32_t infx_set_shard(infx_index* ix, int32_t rank, int32_t nranks, int32_t doc_base, int32_t total_docs) {
    if (!ix || nranks < 1 || rank < 0 || rank >= nranks) return fail(INFX_EINVAL, "bad shard arguments%s");
    ix->rank = rank; ix->nranks = nranks; ix->d.docBase = doc_base; ix->d.totalDocs = total_docs;
    return INFX_OK;
}

int32_t infx_rccl_unique_id(void* id128) {
    if (!id128) return fail(INFX_EINVAL, "null argument%s");
    if (!rccl_api().ok) return fail(INFX_ENCCL, "RCCL unavailable: %s", rccl_api().why);
    ncclUniqueId id; NCCLCHK(rccl_api().getId(&id));
    static_assert(sizeof(ncclUniqueId) == INFX_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof id); return INFX_OK;
}
int32_t infx_set_shard_comm(infx_index* ix, const void* id128) {
    if (!ix || !id128) return fail(INFX_EINVAL, "null argument%s");
    if (!rccl_api().ok) return fail(INFX_ENCCL, "RCCL unavailable: %s", rccl_api().why);
    if (ix->comm) return fail(INFX_EINVAL, "this index already joined a communicator%s");
    HIPCHK(enter_device(ix->cfg.device));
    ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
    NCCLCHK(rccl_api().init(&ix->comm, ix->nranks, id, ix->rank));
    return INFX_OK;
}
int32_t infx_stream_comm(infx_stream* s, const void* id128) {
    if (!s || !id128) return fail(INFX_EINVAL, "null argument%s");
    if (!rccl_api().ok) return fail(INFX_ENCCL, "RCCL unavailable: %s", rccl_api().why);
    if (s->comm) return fail(INFX_EINVAL, "this stream already joined a communicator%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
    NCCLCHK(rccl_api().init(&s->comm, s->ix->nranks, id, s->ix->rank));
    return INFX_OK;
}
int32_t infx_comm_allreduce_sum_u32(infx_stream* s, void* buf, uint64_t count) {
    if (!s || (count && !buf)) return fail(INFX_EINVAL, "null argument%s");
    ncclComm_t c = s->comm ? s->comm : s->ix->comm;
    if (!c) return fail(INFX_EINVAL, "infx_set_shard_comm / infx_stream_comm has not been called%s");
    if (!count) return INFX_OK;
    NCCLCHK(rccl_api().allreduce(buf, buf, (size_t)count, ncclUint32, ncclSum, c, s->st)); s->unsynced = true;
    return INFX_OK;
}
int32_t infx_comm_allgather(infx_stream* s, const void* send, void* recv, uint64_t bytes) {
    if (!s || (bytes && (!send || !recv))) return fail(INFX_EINVAL, "null argument%s");
    ncclComm_t c = s->comm ? s->comm : s->ix->comm;
    if (!c) return fail(INFX_EINVAL, "infx_set_shard_comm / infx_stream_comm has not been called%s");
    if (!bytes) return INFX_OK;
    NCCLCHK(rccl_api().allgather(send, recv, (size_t)bytes, ncclUint8, c, s->st)); s->unsynced = true;
    return INFX_OK;
}
int32_t infx_stream_scratch(infx_stream* s, int32_t slot, uint64_t bytes, void** out) {
    if (!s || !out || slot < 0 || slot >= 16) return fail(INFX_EINVAL, "bad scratch arguments%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    if (bytes > s->capScratch[slot]) {
        if (s->unsynced) { int32_t rc_ = stream_sync(s); if (rc_) return rc_; }      // work queued on the old buffer
        GROW(s->scratch[slot], s->capScratch[slot], (size_t)bytes);
    }
    *out = s->scratch[slot]; return INFX_OK;
}
int32_t infx_stream_copy(infx_stream* s, void* dst, const void* src, uint64_t bytes) {
    if (!s || (bytes && (!dst || !src))) return fail(INFX_EINVAL, "null argument%s");
    if (!bytes) return INFX_OK;
    HIPCHK(enter_device(s->ix->cfg.device));
    const bool dd = is_device_ptr(dst), sd = is_device_ptr(src);
    if (dd && sd) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s->st)); s->unsynced = true; return INFX_OK; }
    if (dd) return up(s, dst, src, bytes);
    if (sd) return down(s, dst, src, bytes);
    if (s->unsynced) { int32_t rc_ = stream_sync(s); if (rc_) return rc_; }
    std::memcpy(dst, src, bytes); return INFX_OK;
}
int32_t infx_stream_unpack(infx_stream* s, const void* src, uint64_t stride, int32_t nranks, int32_t nparts, const uint64_t* part_bytes, void* const* dsts) {
    if (!s || !src || nranks < 1 || nparts < 1 || nparts > 4 || !part_bytes || !dsts) return fail(INFX_EINVAL, "bad unpack arguments%s");
    uint64_t tot = 0; for (int p = 0; p < nparts; p++) { if (!dsts[p] || (part_bytes[p] & 3)) return fail(INFX_EINVAL, "unpack parts are whole 32-bit words%s"); tot += part_bytes[p]; }
    if (tot > stride || (stride & 3)) return fail(INFX_EINVAL, "unpack parts exceed the block%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    if (is_device_ptr(src)) {
        UnpackArgs a{}; a.src = (const uint32_t*)src; a.strideW = stride >> 2; a.nranks = nranks; a.nparts = nparts;
        uint64_t off = 0; for (int p = 0; p < nparts; p++) { a.offW[p] = off >> 2; a.lenW[p] = part_bytes[p] >> 2; a.dst[p] = (uint32_t*)dsts[p]; off += part_bytes[p]; }
        const uint64_t words = (tot >> 2) * (uint64_t)nranks;
        const unsigned blocks = (unsigned)std::min<uint64_t>(4096, (words + 255) / 256);
        if (blocks) k_unpack<<<blocks, 256, 0, s->st>>>(a);
        HIPCHK(hipGetLastError()); s->unsynced = true;
    } else {
        uint64_t off = 0;
        for (int p = 0; p < nparts; p++) { for (int r = 0; r < nranks; r++) std::memcpy((char*)dsts[p] + (uint64_t)r * part_bytes[p], (const char*)src + (uint64_t)r * stride + off, part_bytes[p]); off += part_bytes[p]; }
    }
    return INFX_OK;
}
int32_t infx_stream_fill0(infx_stream* s, void* dev, uint64_t bytes) {
    if (!s || (bytes && !dev)) return fail(INFX_EINVAL, "null argument%s");
    if (bytes) { HIPCHK(enter_device(s->ix->cfg.device)); HIPCHK(hipMemsetAsync(dev, 0, bytes, s->st)); s->unsynced = true; }
    return INFX_OK;
}
int32_t infx_stream_wait(infx_stream* s) {
    if (!s) return fail(INFX_EINVAL, "null argument%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    return stream_sync(s);
}

int32_t infx_stream_native(infx_stream* s, void** hip_stream) {
    if (!s || !hip_stream) return fail(INFX_EINVAL, "null argument%s");
    *hip_stream = (void*)s->st; return INFX_OK;
}

int32_t infx_stream_create(infx_index* ix, infx_stream** out) {
    if (!ix || !out) return fail(INFX_EINVAL, "null argument%s");
    HIPCHK(enter_device(ix->cfg.device));
    infx_stream* s = new infx_stream(); s->ix = ix;
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    s->stMain = s->st;
    {
        static const bool noPrio = [] { const char* e = getenv("INFX_PLAN_PRIORITY"); return e && e[0] == '0'; }();
        int least = 0, greatest = 0;
        if (!noPrio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
            hipStreamCreateWithPriority(&s->stPlan, hipStreamNonBlocking, greatest) == hipSuccess) HIPCHK(hipEventCreateWithFlags(&s->evPlan, hipEventDisableTiming));
        else { (void)hipGetLastError(); s->stPlan = nullptr; }
    }
    {
        static const bool noAux = [] { const char* e = getenv("INFX_REPLAY_AUX"); return e && e[0] == '0'; }();
        if (!noAux && hipStreamCreateWithFlags(&s->stAux, hipStreamNonBlocking) == hipSuccess) HIPCHK(hipEventCreateWithFlags(&s->evJoin, hipEventDisableTiming));
        else { (void)hipGetLastError(); s->stAux = nullptr; }
    }
    HIPCHK(hipEventCreateWithFlags(&s->evTurn, hipEventDisableTiming));
    hipEvent_t* ev[] = {&s->evA0, &s->evA1, &s->evS0, &s->evS1, &s->evC0, &s->evC1, &s->evP0, &s->evP1, &s->evF0, &s->evF1, &s->evX0, &s->evX1, &s->evXa, &s->evXb, &s->evXc};
    for (auto e : ev) HIPCHK(hipEventCreate(e));
    HIPCHK(hipEventCreateWithFlags(&s->evSync, hipEventBlockingSync | hipEventDisableTiming));
    HIPCHK(hipMalloc((void**)&s->dCursor, 16)); HIPCHK(hipMalloc((void**)&s->dOverflow, 4));
    HIPCHK(hipMalloc((void**)&s->dStats, 64)); HIPCHK(hipMemset(s->dStats, 0, 64));
    HIPCHK(hipMalloc((void**)&s->dExactStat, 32)); HIPCHK(hipMemset(s->dExactStat, 0, 32));      // [0..3] replay outcome counters, [4..7] k_select flag reasons
    HIPCHK(hipMalloc((void**)&s->exCounters, 32)); HIPCHK(hipMemset(s->exCounters, 0, 32));
    *out = s; return INFX_OK;
}
void infx_stream_destroy(infx_stream* s) {
    if (!s) return;
    hipSetDevice(s->ix->cfg.device);
    void* ps[] = {s->dQueries, s->dTerms, s->dExtra, s->dRules, s->dHits, s->dHitCount, s->dBlockOut, s->dBlockOutHi, s->dQBytes, s->dUOffs, s->dUMem, s->dUCnt, s->dURange, s->dUBase, s->dUDocs, s->dCounts,
                  s->dCovQ, s->dCovC, s->dCovO, s->dCovF, s->arDoc, s->arScore, s->arCls, s->dCursor, s->dOverflow,
                  s->dFQ, s->dFLists, s->dFOwned, s->dFS1, s->dFMeta, s->dFQueries, s->dFKeys, s->dFScores, s->dFTies, s->dFCounts, s->dFFlags, s->dFErr, s->dFHitsAll, s->dFHcAll, s->dFPairs, s->arMask, s->dDir, s->dFDocs, s->dFacetCols, s->dFacCodes, s->dFacCounts, s->dFacN, s->dRefTerms, s->dExactFlag, s->dExactStat, s->arExc, s->exCand, s->exOut, s->exChunks, s->exQueries, s->exTasks, s->exCounters, s->exContEnd, s->dExProf, s->dSelOrder,
                  s->dNext, s->dPrior, s->shBlob, s->dAllBlobs, s->dAllNext, s->dChainState, s->dChainNeed, s->dHugeWs, s->dHugeCnt, s->dLWordOff, s->dLChars, s->dLMembers, s->dLCount, s->dDense, s->dSelG, s->dAccOrder};
    for (void* p : ps) if (p) hipFree(p);
    for (void* p : s->scratch) if (p) hipFree(p);
    for (void* p : s->parked) hipFree(p);
    if (s->comm && rccl_api().ok) rccl_api().destroy(s->comm);
    if (s->st) hipStreamSynchronize(s->st);
    for (auto& c : s->pins) hipHostFree(c.base);
    hipEvent_t ev[] = {s->evA0, s->evA1, s->evS0, s->evS1, s->evC0, s->evC1, s->evP0, s->evP1, s->evF0, s->evF1, s->evX0, s->evX1, s->evXa, s->evXb, s->evXc};
    for (auto e : ev) hipEventDestroy(e);
    hipEventDestroy(s->evSync);
    if (s->evPlan) hipEventDestroy(s->evPlan);
    if (s->evJoin) hipEventDestroy(s->evJoin);
    { std::lock_guard<std::mutex> lk(s->ix->turnMu); if (s->ix->turnEvent == s->evTurn) s->ix->turnEvent = nullptr; }
    if (s->evTurn) hipEventDestroy(s->evTurn);
    if (s->stAux) { hipStreamSynchronize(s->stAux); hipStreamDestroy(s->stAux); }
    if (s->stPlan) { hipStreamSynchronize(s->stPlan); hipStreamDestroy(s->stPlan); }
    if (s->st) hipStreamDestroy(s->st);
    delete s;
}

// Everything of a Stage-1 accumulate launch up to (and including) the kernel: translation of the query terms into device
// entries, arena bounds, uploads, launch.  No synchronisation.
static int32_t acc_enqueue(infx_stream* s, uint32_t nq, const infx_query* q, uint32_t nterms, const infx_term* terms,
                           uint32_t extra_n, const int32_t* extra_docs) {
    infx_index* ix = s->ix;
    if ((uint64_t)nq * ((uint64_t)ix->d.nRanges + 8) > 0x7FFFFFFFull) return fail(INFX_ECAPACITY, "nq * nRanges exceeds the grid limit; split the batch%s");
    // translate + capacity bound.  A virtual term given as a MEMBER LIST (infx_term.reserved == 1: extra_docs[extra_off..+len) are index
    // term ids) is expanded into one device entry per member, all sharing the term's idf / role / rank and a dedupe group.
    std::vector<DevQuery> dq(nq); std::vector<DevTerm> dt; dt.reserve(nterms + 64);
    std::vector<int32_t> termOfEntry; termOfEntry.reserve(nterms + 64);
    std::vector<DevRefTerm> refT(std::max<uint32_t>(1, nterms));
    for (uint32_t k = 0; k < nterms; k++) refT[k] = DevRefTerm{terms[k].idf, terms[k].max_score, terms[k].term_id, terms[k].term_id < 0 ? 1u : 0u};
    std::vector<unsigned long long> qbase((size_t)nq + 1); unsigned long long maxQb = 0;
    unsigned long long bound = 0; int maxT = 1, useGrp = 0, maxRef = 0, maxRefNarrow = 0; std::vector<uint32_t> wideQ;
    for (uint32_t i = 0; i < nq; i++) {
        const infx_query& Q = q[i];
        if (Q.num_terms > INFX_MAX_QUERY_TERMS || (uint64_t)Q.term_off + Q.num_terms > nterms) return fail(INFX_EINVAL, "bad term range%s");
        if (Q.depth <= 0 || Q.depth > ix->cfg.max_depth) return fail(INFX_EINVAL, "query depth exceeds infx_config.max_depth%s");
        if (Q.mode < INFX_MODE_PREFIX || Q.mode > INFX_MODE_AND) return fail(INFX_EINVAL, "bad query mode%s");
        if (Q.prefix_set >= (int32_t)ix->d.nSets || (Q.mode == INFX_MODE_PREFIX && Q.prefix_set < 0)) return fail(INFX_EINVAL, "bad prefix set%s");
        const uint32_t entryOff = (uint32_t)dt.size();
        unsigned long long qb = 0; int group = 0;
        for (uint32_t k = 0; k < Q.num_terms; k++) {
            const infx_term& tm = terms[Q.term_off + k];
            const bool gen = (Q.mode == INFX_MODE_AND && (tm.role & (INFX_ROLE_S1 | INFX_ROLE_S2))) ||
                             (Q.mode == INFX_MODE_DISJ && (tm.role & (INFX_ROLE_ELIGIBLE | INFX_ROLE_LOWQ)));
            DevTerm D{}; D.idf = tm.idf; D.role = tm.role; D.rank = tm.rank; D.pad = 0; D.refIdx = (uint16_t)k; D.pad2 = 0;
            if (tm.term_id >= 0) {
                if (tm.term_id >= ix->d.T) return fail(INFX_EINVAL, "term id out of range%s");
                D.begin = ix->hPostOff[tm.term_id]; D.end = D.begin + ix->hPostLen[tm.term_id]; D.isVirtual = 0;
                dt.push_back(D); termOfEntry.push_back(tm.term_id);
                if (gen) qb += D.end - D.begin;
            } else if (tm.reserved == 2) {     // union built on the device by the last infx_union_build: extra_off = its index
                if (tm.extra_off >= s->unionCount.size()) return fail(INFX_EINVAL, "virtual term refers to a union that was not built%s");
                D.begin = s->unionBase[tm.extra_off]; D.end = D.begin + s->unionCount[tm.extra_off]; D.isVirtual = 4 | 2;
                D.skip = (uint32_t)((uint64_t)tm.extra_off * (ix->d.nRanges + 1));     // k_union's per-range offsets are its skip table
                dt.push_back(D); termOfEntry.push_back(-1);
                if (gen) qb += D.end - D.begin;
            } else if (tm.reserved == 1) {
                if ((uint64_t)tm.extra_off + tm.extra_len > extra_n) return fail(INFX_EINVAL, "virtual term member list out of range%s");
                if (++group > 255) return fail(INFX_ECAPACITY, "more than 255 fuzzy virtual terms in one query%s");
                useGrp = 1;
                for (uint32_t m = 0; m < tm.extra_len; m++) {
                    int32_t mt = extra_docs[tm.extra_off + m];
                    if (mt < 0 || mt >= ix->d.T) return fail(INFX_EINVAL, "member term id out of range%s");
                    DevTerm M = D; M.begin = ix->hPostOff[mt]; M.end = M.begin + ix->hPostLen[mt]; M.isVirtual = 2; M.pad = (uint8_t)group;
                    dt.push_back(M); termOfEntry.push_back(mt);
                    if (gen) qb += M.end - M.begin;
                }
            } else {
                if ((uint64_t)tm.extra_off + tm.extra_len > extra_n) return fail(INFX_EINVAL, "virtual term slice out of range%s");
                D.begin = tm.extra_off; D.end = (uint64_t)tm.extra_off + tm.extra_len; D.isVirtual = 1; D.skip = 0xFFFFFFFFu;
                dt.push_back(D); termOfEntry.push_back(-1);
                if (gen) qb += D.end - D.begin;
            }
        }
        const uint32_t nEntries = (uint32_t)dt.size() - entryOff;
        if (nEntries > 2048) return fail(INFX_ECAPACITY, "query expands to more than 2048 posting lists (fuzzy member lists); materialise the union on the host%s");
        dq[i] = DevQuery{entryOff, nEntries, Q.mode, Q.prefix_set, Q.depth, Q.n_and, Q.term_off, Q.num_terms};
        maxT = std::max(maxT, (int)nEntries); maxRef = std::max(maxRef, (int)Q.num_terms);
        if (Q.num_terms > 64) wideQ.push_back(i); else maxRefNarrow = std::max(maxRefNarrow, (int)Q.num_terms);
        if (Q.mode == INFX_MODE_PREFIX) qb = ix->hPsOff[Q.prefix_set + 1] - ix->hPsOff[Q.prefix_set];
        qbase[i] = bound;
        bound += std::min<unsigned long long>(qb, (unsigned long long)ix->d.N);
        maxQb = std::max(maxQb, std::min<unsigned long long>(qb, (unsigned long long)ix->d.N));
    }
    qbase[nq] = bound;
    {   // k_accumulate's block order: queries by row bound, descending; "heavy" = at least INFX_ACC_HEAVY_ROWS (default 393216) candidate rows possible, at most 64 queries
        static const long long heavyRows = [] { const char* e = getenv("INFX_ACC_HEAVY_ROWS"); return e ? atoll(e) : 393216ll; }();      // 0: query order
        s->nAccHeavy = 0xFFFFFFFFu;
        if (heavyRows > 0 && nq > 1) {
            std::vector<uint32_t> ord(nq); for (uint32_t i = 0; i < nq; i++) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return qbase[a + 1] - qbase[a] > qbase[b + 1] - qbase[b]; });
            uint32_t h = 0; while (h < nq - 1 && h < 64u && qbase[ord[h] + 1] - qbase[ord[h]] >= (unsigned long long)heavyRows) h++;
            GROW(s->dAccOrder, s->capAccOrder, (size_t)nq * 4); UP(s->dAccOrder, ord.data(), (size_t)nq * 4);
            s->nAccHeavy = h;
        }
    }
    { const unsigned long long gm = giant_min_rows(); uint32_t c = 0; if (gm) for (uint32_t i = 0; i < nq; i++) if (qbase[i + 1] - qbase[i] >= gm) c++; s->nBoundGiants = c; }
    s->maxQueryBound = maxQb;      // no query of the batch can hold more rows than this (select_giants: nothing to sweep outside k_select below its threshold)
    {   // INFX_ACC_SHARE_STATS=1 (profiling): how many posting bytes do the queries of the batch share?  total = sum over (query, list) of the list length;
        // distinct = every list once; by XCD = every list once per XCD under the current block -> XCD assignment (q % 8) and under a chunked assignment of
        // the queries sorted by their longest list
        static const bool want = [] { const char* e = getenv("INFX_ACC_SHARE_STATS"); return e && e[0] == '1'; }();
        if (want) {
            unsigned long long total = 0, distinct = 0, byMod = 0, byChunk = 0;
            std::vector<std::pair<unsigned long long, uint32_t>> order(nq);       // (begin of the longest list, query)
            for (uint32_t i = 0; i < nq; i++) {
                unsigned long long bestLen = 0, bestBeg = 0;
                for (uint32_t k = 0; k < dq[i].numTerms; k++) { const DevTerm& D = dt[dq[i].termOff + k]; const unsigned long long L = D.end - D.begin; total += L; if (!D.isVirtual && L > bestLen) { bestLen = L; bestBeg = D.begin; } }
                order[i] = {bestBeg, i};
            }
            std::sort(order.begin(), order.end());
            std::vector<uint32_t> chunkOf(nq); for (uint32_t j = 0; j < nq; j++) chunkOf[order[j].second] = (uint32_t)((uint64_t)j * 8 / nq);
            std::unordered_map<unsigned long long, uint32_t> seenAll, seenMod, seenChunk;      // begin -> bitmask of XCDs that already fetch it
            for (uint32_t i = 0; i < nq; i++)
                for (uint32_t k = 0; k < dq[i].numTerms; k++) {
                    const DevTerm& D = dt[dq[i].termOff + k]; const unsigned long long L = D.end - D.begin;
                    if (D.isVirtual & 5) { distinct += L; byMod += L; byChunk += L; continue; }
                    if (!seenAll[D.begin]++) distinct += L;
                    uint32_t& m = seenMod[D.begin]; if (!(m & (1u << (i & 7)))) { m |= 1u << (i & 7); byMod += L; }
                    uint32_t& c = seenChunk[D.begin]; if (!(c & (1u << chunkOf[i]))) { c |= 1u << chunkOf[i]; byChunk += L; }
                }
            fprintf(stderr, "[infx] posting sharing of the batch (%u queries): total %.2f G postings, distinct %.2f G (x%.2f), once per XCD with q %% 8: %.2f G (x%.2f), with sorted chunks: %.2f G (x%.2f)\n",
                    nq, total * 1e-9, distinct * 1e-9, (double)total / std::max(1ull, distinct), byMod * 1e-9, (double)total / std::max(1ull, byMod), byChunk * 1e-9, (double)total / std::max(1ull, byChunk));
        }
    }
    // arena
    size_t need = (size_t)bound + 64;
    if (need > s->arCap) {
        if (need > ((size_t)1 << 31)) return fail(INFX_ECAPACITY, "candidate superset bound exceeds 2^31 entries; split the batch%s");
        if (s->arDoc) { ws_release(s, s->arDoc); ws_release(s, s->arScore); ws_release(s, s->arCls); s->arDoc = nullptr; }
        // The bound (sum of the candidate-generating lists' lengths) varies by +-30 % between 1000-query batches of one workload, and a reallocation in the
        // middle of a stream is not free even without hipFree (a multi-GB hipMalloc can wait for the device): twice the first batch's need, doubling after that.
        // ~40 B per row: 5 GB per session at 10 M documents, of 288.
        size_t n = std::max(need * 2, s->arCap * 2);
        if (getenv("INFX_DEBUG_GROW")) fprintf(stderr, "[infx] arena grow: %zu -> %zu rows (need %zu)\n", s->arCap, n, need);
        if (hipMalloc((void**)&s->arDoc, n * 4) != hipSuccess || hipMalloc((void**)&s->arScore, n * 4) != hipSuccess || hipMalloc((void**)&s->arCls, n) != hipSuccess)
            return fail(INFX_ENOMEM, "arena allocation failed%s");
        s->arCap = n;
    }
    // per-row hit masks (2 bits per reference term) for the exact Stage-1 replay, as wide as the batch's queries need: one or two 64-bit words for queries of <= 64
    // terms.  A query with more terms (a very long query: words + n-grams, up to 128 = VectorModel.cs:381's cap) takes FOUR words for the rows of its whole batch
    // (round 6; rare: the masks of such a batch cost twice the bytes) and is replayed by the sequential kernel, which reads the masks from the arena (k_mark_wide,
    // k_exact1).  Document shards keep two words and the first-pass cut for such a query (k_gflag passes it over): the sharded replay's kernels hold two words.
    const bool wideExact = !wideQ.empty() && exact_enabled(ix) && ix->nranks == 1 && ix->d.docBase == 0;
    s->maskWords = !exact_enabled(ix) || (!wideExact && wideQ.size() == nq) ? 0 : (wideExact ? 4 : (maxRefNarrow <= 32 ? 1 : 2));
    s->nWide = (uint32_t)wideQ.size();
    if (s->nWide) { GROW(s->dWideQ, s->capWideQ, wideQ.size() * 4); UP(s->dWideQ, wideQ.data(), wideQ.size() * 4); }
    if (s->maskWords && s->arCap * (size_t)s->maskWords > s->arMaskCap) {
        if (s->arMask) { ws_release(s, s->arMask); s->arMask = nullptr; s->arMaskCap = 0; }
        const size_t n = s->arCap * (size_t)s->maskWords;
        if (hipMalloc((void**)&s->arMask, n * 8) != hipSuccess) return fail(INFX_ENOMEM, "arena mask allocation failed%s");
        s->arMaskCap = n;
    }
    if (s->maskWords && s->arCap > s->exCap) {
        if (s->arExc) { ws_release(s, s->arExc); ws_release(s, s->exCand); ws_release(s, s->exOut); s->arExc = nullptr; s->exCand = nullptr; s->exOut = nullptr; s->exCap = 0; }
        if (hipMalloc((void**)&s->arExc, s->arCap * 4) != hipSuccess || hipMalloc((void**)&s->exCand, s->arCap * 4) != hipSuccess || hipMalloc((void**)&s->exOut, s->arCap * sizeof(infx_hit)) != hipSuccess)
            return fail(INFX_ENOMEM, "exact-replay workspace allocation failed%s");
        s->exCap = s->arCap;
    }
    s->arBound = (size_t)bound;
    GROW(s->dDir, s->capDir, (size_t)nq * ix->d.nRanges * sizeof(uint2));
    GROW(s->dRefTerms, s->capRefTerms, refT.size() * sizeof(DevRefTerm));
    GROW(s->dExactFlag, s->capExactFlag, (size_t)nq * 4);
    GROW(s->dQueries, s->capQueries, nq * sizeof(DevQuery));
    GROW(s->dTerms, s->capTerms, std::max<size_t>(1, dt.size()) * sizeof(DevTerm));
    GROW(s->dExtra, s->capExtra, ((size_t)extra_n + 4) * 4);
    GROW(s->dUDocs, s->capUDocs, 16); GROW(s->dURange, s->capURange, 4);
    GROW(s->dBlockOut, s->capBlockOut, ((size_t)nq + 1) * 8);      // qBase
    GROW(s->dBlockOutHi, s->capBlockOutHi, (size_t)nq * 4);        // qCursor
    GROW(s->dQBytes, s->capQBytes, (size_t)nq * 12); s->lastNqAlloc = nq;      // per query: algorithmic bytes (u64) | best emitted score bits (u32, behind the nq u64s)
    GROW(s->dCounts, s->capCounts, (size_t)nq * INFX_NCLASS * 4);
    for (size_t i = 0; i < dt.size(); i++) if (termOfEntry[i] >= 0) dt[i].skip = ix->hSkipIdx[termOfEntry[i]];

    UP(s->dQueries, dq.data(), nq * sizeof(DevQuery));
    UP(s->dTerms, dt.data(), dt.size() * sizeof(DevTerm));
    UP(s->dRefTerms, refT.data(), refT.size() * sizeof(DevRefTerm));
    HIPCHK(hipMemsetAsync(s->dExactFlag, 0, (size_t)nq * 4, s->st));
    UP(s->dExtra, extra_docs, (size_t)extra_n * 4);
    UP(s->dBlockOut, qbase.data(), ((size_t)nq + 1) * 8);
    HIPCHK(hipMemsetAsync(s->dQBytes, 0, (size_t)nq * 12, s->st));
    HIPCHK(hipMemsetAsync(s->dOverflow, 0, 4, s->st));
    HIPCHK(hipMemsetAsync(s->dCounts, 0, (size_t)nq * INFX_NCLASS * 4, s->st));
    HIPCHK(hipMemsetAsync(s->dBlockOutHi, 0, (size_t)nq * 4, s->st));
    HIPCHK(hipMemsetAsync(s->dDir, 0, (size_t)nq * ix->d.nRanges * sizeof(uint2), s->st));
    Arena ar = make_arena(s);
    HIPCHK(hipEventRecord(s->evA0, s->st));
    switch (ix->d.R) {
        case 512: launch_acc<512>(s, nq, ar, maxT, useGrp, maxRef); break;
        case 1024: launch_acc<1024>(s, nq, ar, maxT, useGrp, maxRef); break;
        case 2048: launch_acc<2048>(s, nq, ar, maxT, useGrp, maxRef); break;
        case 4096: launch_acc<4096>(s, nq, ar, maxT, useGrp, maxRef); break;
        case 8192: launch_acc<8192>(s, nq, ar, maxT, useGrp, maxRef); break;
        default: launch_acc<16384>(s, nq, ar, maxT, useGrp, maxRef); break;
    }
    if (s->accLayoutBad) return fail(INFX_EHIP, "k_accumulate was built with static LDS: its byte addressing (LDS8) is invalid%s");
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evA1, s->st));
    s->timedAcc = true;
    s->lastQ.assign(q, q + nq); s->lastNq = nq; s->lastExact2[0] = s->lastExact2[1] = 0; s->msReplay = 0.f; s->timedReplay = false; s->lastFlagWhy[0] = s->lastFlagWhy[1] = s->lastFlagWhy[2] = 0;
    return INFX_OK;
}

int32_t infx_stage1_accumulate(infx_stream* s, uint32_t nq, const infx_query* q, uint32_t nterms, const infx_term* terms,
                               uint32_t extra_n, const int32_t* extra_docs, infx_counts* counts_out) {
    if (!s || !q || (nterms && !terms)) return fail(INFX_EINVAL, "null argument%s");
    infx_index* ix = s->ix;
    if (!ix->havePostings || !ix->haveDocs) return fail(INFX_EINVAL, "index not uploaded%s");
    if (nq == 0) return INFX_OK;
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    {   // the turnstile of infx_search_fused for the staged (document-sharded) pipeline: a batch's accumulation waits, stream-side, for the accumulation of the batch
        // submitted before it on this index — the pipeline sessions of a rank take the GPU-filling kernel one after the other instead of against each other
        static const bool turnstile = [] { const char* e = getenv("INFX_TURNSTILE"); return !(e && e[0] == '0'); }();
        std::unique_lock<std::mutex> turn(ix->turnMu, std::defer_lock);
        if (turnstile) { turn.lock(); if (ix->turnEvent && ix->turnEvent != s->evTurn) HIPCHK(hipStreamWaitEvent(s->st, ix->turnEvent, 0)); }
        { int32_t rc_ = acc_enqueue(s, nq, q, nterms, terms, extra_n, extra_docs); if (rc_) return rc_; }
        if (turnstile) { HIPCHK(hipEventRecord(s->evTurn, s->st)); ix->turnEvent = s->evTurn; }
    }
    uint32_t ovf = 0; std::vector<unsigned long long> qbytes(nq);
    if (counts_out) DOWNX(counts_out, s->dCounts, (size_t)nq * INFX_NCLASS * 4);       // host memory, or a device tensor the caller all-reduces in place
    DOWN(&ovf, s->dOverflow, 4);
    DOWN(qbytes.data(), s->dQBytes, (size_t)nq * 8);
    SYNC();
    if (ovf) return fail(INFX_ECAPACITY, "candidate arena overflow (bound violated)%s");
    s->lastAlgBytes = 0; for (auto b : qbytes) s->lastAlgBytes += b;
    s->lastQ.assign(q, q + nq); s->lastNq = nq;
    return INFX_OK;
}

int32_t infx_stage1_select(infx_stream* s, uint32_t nq, const infx_counts* counts, infx_hit* out, uint32_t* out_count) {
    if (!s || !counts || !out || !out_count) return fail(INFX_EINVAL, "null argument%s");
    if (nq == 0) return INFX_OK;
    if (nq != s->lastNq) return fail(INFX_EINVAL, "infx_stage1_select must follow infx_stage1_accumulate of the same batch%s");
    infx_index* ix = s->ix;
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    std::vector<SelRule> rules(nq);
    int maxDepth = 0;
    s->lastCandTotal = 0;
    for (uint32_t i = 0; i < nq; i++) { rules[i] = make_rule(s->lastQ[i], counts[i].c); maxDepth = std::max(maxDepth, rules[i].depth); s->lastCandTotal += rules[i].total; }
    GROW(s->dRules, s->capRules, nq * sizeof(SelRule));
    GROW(s->dHits, s->capHits, (size_t)nq * maxDepth * sizeof(infx_hit));
    GROW(s->dHitCount, s->capHitCount, (size_t)nq * 4);
    UP(s->dRules, rules.data(), nq * sizeof(SelRule));
    Arena ar = make_arena(s);
    HIPCHK(hipEventRecord(s->evS0, s->st));
    const bool exact = exact_possible(s);
    if (exact) HIPCHK(hipMemsetAsync(s->dExactStat + 4, 0, 16, s->st));
    const uint32_t* selOrd = select_order(s, nq);
    k_select<<<nq, SEL_THREADS, 0, s->st>>>(ar, ix->d.nRanges, (const SelRule*)s->dRules, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, maxDepth, exact ? (uint32_t*)s->dExactFlag : nullptr, exact ? s->dExactStat + 4 : nullptr, nullptr, selOrd,
                                            select_giants(s, ar, nq, selOrd));
    if (exact) { int32_t rc_ = mark_wide_queries(s, ix->cfg.max_depth); if (rc_) return rc_; }
    if (exact) { int32_t rc_ = enqueue_exact(s, nq, maxDepth); if (rc_) return rc_; }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evS1, s->st));
    s->timedSel = true;
    std::vector<infx_hit> tmp((size_t)nq * maxDepth);
    DOWN(tmp.data(), s->dHits, tmp.size() * sizeof(infx_hit));
    DOWN(out_count, s->dHitCount, (size_t)nq * 4);
    SYNC();
    // caller layout: nq * depth_i packed with the query's own depth as stride == we use max depth of the batch as stride
    for (uint32_t i = 0; i < nq; i++) memcpy(out + (size_t)i * maxDepth, tmp.data() + (size_t)i * maxDepth, (size_t)out_count[i] * sizeof(infx_hit));
    return INFX_OK;
}

int32_t infx_stage1_batch(infx_stream* s, uint32_t nq, const infx_query* q, uint32_t nterms, const infx_term* terms,
                          uint32_t extra_n, const int32_t* extra_docs, infx_hit* out, uint32_t* out_count) {
    std::vector<infx_counts> counts(nq);
    int32_t rc = infx_stage1_accumulate(s, nq, q, nterms, terms, extra_n, extra_docs, counts.data());
    if (rc) return rc;
    return infx_stage1_select(s, nq, counts.data(), out, out_count);
}

int32_t infx_stage2_long_queries(infx_stream* s, uint32_t n, const infx_cov_query_long* q) {
    if (!s || (n && !q)) return fail(INFX_EINVAL, "null argument%s");
    s->nLongQ = 0;
    if (!n) return INFX_OK;
    infx_index* ix = s->ix;
    HIPCHK(enter_device(ix->cfg.device));
    // (no pin_reset: this call sits between the phases of a batch, whose pending downloads must survive it; its own upload is staged behind theirs)
    for (uint32_t i = 0; i < n; i++) {
        if (q[i].num_tokens < 0 || q[i].num_tokens > INFX_LONGQ_TOKENS || q[i].text_len < 0 || q[i].text_len > INFX_LONGQ_CHARS || q[i].num_fusion_tokens < 0 || q[i].num_fusion_tokens > 2 * INFX_LONGQ_TOKENS)
            return fail(INFX_EUNSUPPORTED, "query exceeds the long Stage-2 envelope%s");
        // the kernel dereferences text + tok_off: a token table that points outside the text would read out of bounds on the device
        for (int32_t t = 0; t < q[i].num_tokens; t++) if ((int32_t)q[i].tok_off[t] + (int32_t)q[i].tok_len[t] > q[i].text_len) return fail(INFX_EINVAL, "long query: a token lies outside the query text%s");
        for (int32_t t = 0; t < q[i].num_fusion_tokens; t++) if ((int32_t)q[i].ftok_off[t] + (int32_t)q[i].ftok_len[t] > q[i].text_len) return fail(INFX_EINVAL, "long query: a fusion token lies outside the query text%s");
    }
    GROW(s->dCovQL, s->capCovQL, (size_t)n * sizeof(infx_cov_query_long));
    UP(s->dCovQL, q, (size_t)n * sizeof(infx_cov_query_long));
    s->longAlias = false;
    for (uint32_t i = 0; i < n && !s->longAlias; i++) for (int32_t k = 0; k < q[i].text_len; k++) if (s2_host_is_alias(q[i].text[k])) { s->longAlias = true; break; }
    s->nLongQ = n;
    return INFX_OK;
}

int32_t infx_stage2_batch(infx_stream* s, uint32_t nq, const infx_cov_query* q, uint32_t ncand, const infx_cov_cand* cand,
                          infx_cov_out* out, int32_t* feat_out) {
    if (!s || (ncand && (!q || !cand || !out))) return fail(INFX_EINVAL, "null argument%s");
    if (ncand == 0) return INFX_OK;
    infx_index* ix = s->ix;
    if (!ix->haveDocs || !ix->d.text) return fail(INFX_EINVAL, "document text not uploaded%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    for (uint32_t i = 0; i < nq; i++) {
        if (q[i].reserved != 0) { if (q[i].reserved < 0 || (uint32_t)q[i].reserved > s->nLongQ) return fail(INFX_EINVAL, "query refers to a missing long-query record (infx_stage2_long_queries)%s"); continue; }
        if (q[i].num_tokens > INFX_MAX_QUERY_TOKENS || q[i].text_len > INFX_MAX_QUERY_CHARS || q[i].num_fusion_tokens > 2 * INFX_MAX_QUERY_TOKENS)
            return fail(INFX_EUNSUPPORTED, "query exceeds the Stage-2 envelope (hand it over as a long query: infx_stage2_long_queries)%s");
        if (q[i].num_tokens < 0 || q[i].text_len < 0 || q[i].num_fusion_tokens < 0) return fail(INFX_EINVAL, "negative count in a coverage query%s");
        for (int32_t t = 0; t < q[i].num_tokens; t++) if ((int32_t)q[i].tok_off[t] + (int32_t)q[i].tok_len[t] > q[i].text_len) return fail(INFX_EINVAL, "coverage query: a token lies outside the query text%s");
        for (int32_t t = 0; t < q[i].num_fusion_tokens; t++) if ((int32_t)q[i].ftok_off[t] + (int32_t)q[i].ftok_len[t] > q[i].text_len) return fail(INFX_EINVAL, "coverage query: a fusion token lies outside the query text%s");
    }
    GROW(s->dCovQ, s->capCovQ, (size_t)nq * sizeof(infx_cov_query));
    GROW(s->dCovC, s->capCovC, (size_t)ncand * sizeof(infx_cov_cand));
    GROW(s->dCovO, s->capCovO, (size_t)ncand * sizeof(infx_cov_out));
    if (feat_out) GROW(s->dCovF, s->capCovF, (size_t)ncand * INFX_NFEAT * 4);
    UP(s->dCovQ, q, (size_t)nq * sizeof(infx_cov_query));
    s->batchAlias = s->longAlias;
    for (uint32_t i = 0; i < nq && !s->batchAlias; i++) if (q[i].reserved == 0) for (int32_t k = 0; k < q[i].text_len && k < (int32_t)INFX_MAX_QUERY_CHARS; k++) if (s2_host_is_alias(q[i].text[k])) { s->batchAlias = true; break; }
    UP(s->dCovC, cand, (size_t)ncand * sizeof(infx_cov_cand));
    HIPCHK(hipEventRecord(s->evC0, s->st));
    S2_LAUNCH_FAST(ix->d, (const infx_cov_query*)s->dCovQ, nq,
                                                                                 (const infx_cov_cand*)s->dCovC, ncand, (infx_cov_out*)s->dCovO, feat_out ? (int32_t*)s->dCovF : nullptr, 0, 0, nullptr);
    S2_LAUNCH_SLOW(ix->d, (const infx_cov_query*)s->dCovQ, nq,
                                                                                 (const infx_cov_cand*)s->dCovC, ncand, (infx_cov_out*)s->dCovO, feat_out ? (int32_t*)s->dCovF : nullptr, 0, 1, nullptr);
    { int32_t rc_ = s2_huge_ready(s); if (rc_) return rc_; }
    S2_LAUNCH_HUGE(ix->d, (const infx_cov_query*)s->dCovQ, nq,
                                                                                 (const infx_cov_cand*)s->dCovC, ncand, (infx_cov_out*)s->dCovO, feat_out ? (int32_t*)s->dCovF : nullptr, 0, 1, nullptr);
    S2_LAUNCH_LONGQ(ix->d, (const infx_cov_query*)s->dCovQ, nq,
                                                                                 (const infx_cov_cand*)s->dCovC, ncand, (infx_cov_out*)s->dCovO, feat_out ? (int32_t*)s->dCovF : nullptr, 0, 1, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evC1, s->st));
    s->timedCov = true;
    DOWN(out, s->dCovO, (size_t)ncand * sizeof(infx_cov_out));
    if (feat_out) DOWN(feat_out, s->dCovF, (size_t)ncand * INFX_NFEAT * 4);
    SYNC();
    return INFX_OK;
}

int32_t infx_upload_wordmatcher(infx_index* ix, uint64_t n_exact, const int32_t* exact_docs, uint64_t n_ld1, const int32_t* ld1_docs) {
    if (!ix || (n_exact && !exact_docs) || (n_ld1 && !ld1_docs)) return fail(INFX_EINVAL, "null argument%s");
    HIPCHK(enter_device(ix->cfg.device));
    int32_t *dE = nullptr, *dL = nullptr;
    HIPCHK(dalloc(ix, &dE, (size_t)n_exact + 1)); HIPCHK(dalloc(ix, &dL, (size_t)n_ld1 + 1));
    if (n_exact) HIPCHK(hipMemcpy(dE, exact_docs, (size_t)n_exact * 4, hipMemcpyHostToDevice));
    if (n_ld1) HIPCHK(hipMemcpy(dL, ld1_docs, (size_t)n_ld1 * 4, hipMemcpyHostToDevice));
    ix->d.wmExact = dE; ix->d.wmLd1 = dL; ix->nWmExact = n_exact; ix->nWmLd1 = n_ld1; ix->haveWm = true;
    return INFX_OK;
}


// ---- dictionaries of the device-side planning lookups (lookup.hip.inc) --------------------------------------------------------------
static int32_t dict_upload(infx_index* ix, DevDict& D, uint32_t nkeys, const uint32_t* key_offs, const uint16_t* chars, const uint64_t* list_offs, uint64_t ndocs) {
    D = DevDict{};
    if (nkeys == 0) return INFX_OK;
    for (uint32_t i = 0; i < nkeys; i++) if (key_offs[i + 1] < key_offs[i] || list_offs[i + 1] < list_offs[i]) return fail(INFX_EINVAL, "dictionary offsets must ascend%s");
    if (list_offs[nkeys] > ndocs) return fail(INFX_EINVAL, "dictionary list offsets exceed the uploaded doc-id array%s");
    { int32_t rc_ = dcopy(ix, &D.keyOff, key_offs, (size_t)nkeys + 1); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &D.chars, chars, (size_t)key_offs[nkeys]); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &D.listOff, list_offs, (size_t)nkeys + 1); if (rc_) return rc_; }
    uint64_t cap = 1024; while (cap < (uint64_t)nkeys * 2) cap <<= 1;
    if (cap > 0x80000000ull) return fail(INFX_ECAPACITY, "dictionary too large%s");
    unsigned long long* slots = nullptr;
    HIPCHK(dalloc(ix, &slots, (size_t)cap));
    HIPCHK(hipMemset(slots, 0, (size_t)cap * 8));
    k_dict_build<<<(nkeys + 255) / 256, 256>>>(slots, (uint32_t)(cap - 1), D.keyOff, D.chars, nkeys);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    D.slots = slots; D.mask = (uint32_t)(cap - 1);
    return INFX_OK;
}
int32_t infx_upload_wm_dictionary(infx_index* ix, uint32_t n_exact, const uint32_t* exact_key_offs, const uint16_t* exact_chars, const uint64_t* exact_list_offs,
                                  uint32_t n_ld1, const uint32_t* ld1_key_offs, const uint16_t* ld1_chars, const uint64_t* ld1_list_offs,
                                  uint32_t n_words, const uint32_t* word_offs, const uint16_t* word_chars, const int32_t* word_last_doc,
                                  uint32_t n_affix, const uint32_t* affix_fwd, const uint32_t* affix_rev, int32_t min_ld1, int32_t max_ld1) {
    if (!ix || (n_exact && (!exact_key_offs || !exact_chars || !exact_list_offs)) || (n_ld1 && (!ld1_key_offs || !ld1_chars || !ld1_list_offs)) ||
        (n_words && (!word_offs || !word_chars || !word_last_doc)) || (n_affix && (!affix_fwd || !affix_rev))) return fail(INFX_EINVAL, "null argument%s");
    if (!ix->haveWm) return fail(INFX_EINVAL, "infx_upload_wordmatcher comes first%s");
    if (ix->haveDict) return fail(INFX_EINVAL, "WordMatcher dictionary already uploaded%s");
    if (min_ld1 < 1 || max_ld1 < min_ld1 || 3 + 2 * max_ld1 > INFX_MAX_WM_LISTS) return fail(INFX_EINVAL, "bad LD1 word-length window%s");
    for (uint32_t i = 0; i < n_affix; i++) if (affix_fwd[i] >= n_words || affix_rev[i] >= n_words) return fail(INFX_EINVAL, "affix word id out of range%s");
    HIPCHK(enter_device(ix->cfg.device));
    if (!ix->lk) ix->lk = new DevLookup{};
    DevLookup& K = *ix->lk;
    { int32_t rc_ = dict_upload(ix, K.exact, n_exact, exact_key_offs, exact_chars, exact_list_offs, ix->nWmExact); if (rc_) return rc_; }
    { int32_t rc_ = dict_upload(ix, K.ld1, n_ld1, ld1_key_offs, ld1_chars, ld1_list_offs, ix->nWmLd1); if (rc_) return rc_; }
    if (n_words) {
        { int32_t rc_ = dcopy(ix, &K.wordOff, word_offs, (size_t)n_words + 1); if (rc_) return rc_; }
        { int32_t rc_ = dcopy(ix, &K.wordChars, word_chars, (size_t)word_offs[n_words]); if (rc_) return rc_; }
        { int32_t rc_ = dcopy(ix, &K.wordLastDoc, word_last_doc, (size_t)n_words); if (rc_) return rc_; }
    }
    if (n_affix) {
        { int32_t rc_ = dcopy(ix, &K.affixFwd, affix_fwd, (size_t)n_affix); if (rc_) return rc_; }
        { int32_t rc_ = dcopy(ix, &K.affixRev, affix_rev, (size_t)n_affix); if (rc_) return rc_; }
    }
    K.nAffix = n_affix; K.minLd1 = min_ld1; K.maxLd1 = max_ld1;
    ix->haveDict = true;
    return INFX_OK;
}
int32_t infx_upload_term_trie(infx_index* ix, uint32_t n_nodes, const uint32_t* edge_start, const uint16_t* edge_label, const uint32_t* edge_child, const int32_t* node_term,
                              uint32_t n_terms, const uint32_t* sorted_terms) {
    if (!ix || !n_nodes || !edge_start || !node_term || (n_terms && !sorted_terms)) return fail(INFX_EINVAL, "null argument%s");
    if (ix->haveTrie) return fail(INFX_EINVAL, "term trie already uploaded%s");
    const uint32_t ne = edge_start[n_nodes];
    if (ne && (!edge_label || !edge_child)) return fail(INFX_EINVAL, "null argument%s");
    for (uint32_t v = 0; v < n_nodes; v++) { if (edge_start[v + 1] < edge_start[v]) return fail(INFX_EINVAL, "edge offsets must ascend%s"); if (node_term[v] >= (int32_t)n_terms) return fail(INFX_EINVAL, "node term id out of range%s"); }
    for (uint32_t k = 0; k < ne; k++) if (edge_child[k] >= n_nodes) return fail(INFX_EINVAL, "edge child out of range%s");
    std::vector<uint32_t> rank(n_terms, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n_terms; i++) { if (sorted_terms[i] >= n_terms || rank[sorted_terms[i]] != 0xFFFFFFFFu) return fail(INFX_EINVAL, "sorted_terms is not a permutation%s"); rank[sorted_terms[i]] = i; }
    HIPCHK(enter_device(ix->cfg.device));
    if (!ix->lk) ix->lk = new DevLookup{};
    DevLookup& K = *ix->lk;
    { int32_t rc_ = dcopy(ix, &K.rEdgeStart, edge_start, (size_t)n_nodes + 1); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &K.rEdgeLabel, edge_label, (size_t)ne); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &K.rEdgeChild, edge_child, (size_t)ne); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &K.rTerm, node_term, (size_t)n_nodes); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &K.termRank, (const uint32_t*)rank.data(), (size_t)n_terms); if (rc_) return rc_; }
    { int32_t rc_ = dcopy(ix, &K.sortedTerms, sorted_terms, (size_t)n_terms); if (rc_) return rc_; }
    K.nTerms = n_terms; K.nNodes = n_nodes;
    ix->haveTrie = true;
    return INFX_OK;
}
int32_t infx_ld1_expand(infx_stream* s, uint32_t nwords, const uint32_t* word_offs, const uint16_t* chars, uint32_t cap, int32_t* members_out, uint32_t* counts_out, uint32_t* status_out) {
    if (!s || (nwords && (!word_offs || !members_out || !counts_out || !status_out || (word_offs[nwords] && !chars))) || cap == 0) return fail(INFX_EINVAL, "null argument%s");
    if (nwords == 0) return INFX_OK;
    infx_index* ix = s->ix;
    if (!ix->haveTrie) return fail(INFX_EINVAL, "infx_upload_term_trie has not been called%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    PlanStream plan(s);
    const size_t nch = word_offs[nwords];
    GROW(s->dLWordOff, s->capLWordOff, ((size_t)nwords + 1) * 4);
    GROW(s->dLChars, s->capLChars, std::max<size_t>(1, nch) * 2);
    GROW(s->dLMembers, s->capLMembers, (size_t)nwords * cap * 4);
    GROW(s->dLCount, s->capLCount, (size_t)nwords * 8);
    UP(s->dLWordOff, word_offs, ((size_t)nwords + 1) * 4);
    UP(s->dLChars, chars, nch * 2);
    uint32_t* dCnt = (uint32_t*)s->dLCount; uint32_t* dSt = dCnt + nwords;
    k_ld1<<<nwords, WAVE, 0, s->st>>>(*ix->lk, (const uint32_t*)s->dLWordOff, (const uint16_t*)s->dLChars, nwords, cap, (int32_t*)s->dLMembers, dCnt, dSt, nullptr, nullptr, 0u);
    HIPCHK(hipGetLastError());
    DOWN(counts_out, dCnt, (size_t)nwords * 4);
    DOWN(status_out, dSt, (size_t)nwords * 4);
    SYNC();
    // only the members that exist travel back: one copy per word of min(count, cap) ids (a few hundred bytes each) would be many small DMAs, so
    // the rows are fetched in one strided pass up to the largest count of the batch
    uint32_t mx = 0; for (uint32_t i = 0; i < nwords; i++) if (!status_out[i]) mx = std::max(mx, std::min(counts_out[i], cap));
    if (mx) {
        void* p = pin_take(s, (size_t)nwords * mx * 4); if (!p) return fail(INFX_ENOMEM, "hipHostMalloc staging failed%s");
        HIPCHK(hipMemcpy2DAsync(p, (size_t)mx * 4, s->dLMembers, (size_t)cap * 4, (size_t)mx * 4, nwords, hipMemcpyDeviceToHost, s->st)); s->unsynced = true;
        SYNC();
        for (uint32_t i = 0; i < nwords; i++) { const uint32_t c = status_out[i] ? 0u : std::min(counts_out[i], cap); std::memcpy(members_out + (size_t)i * cap, (const int32_t*)p + (size_t)i * mx, (size_t)c * 4); }
    }
    return INFX_OK;
}

static uint32_t pow2_at_least(uint32_t v, uint32_t lo) { uint32_t p = lo; while (p < v) p <<= 1; return p; }

// ---- fused pipeline pieces (shared by infx_search_fused and the sharded stage API) ------------------------------------------
static uint32_t wm_words(const infx_cov_query& c) {      // words k_wm looks up: fusion tokens of at least two characters (WordMatcherLookup.cs:27-31)
    uint32_t n = 0; const int nt = std::min(c.num_fusion_tokens, 2 * INFX_MAX_QUERY_TOKENS);
    for (int t = 0; t < nt; t++) if (c.ftok_len[t] >= 2 && (int)c.ftok_off[t] + (int)c.ftok_len[t] <= c.text_len) n++;
    return n;
}
static int32_t fused_check_queries(infx_index* ix, uint32_t nd, uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq,
                                   uint32_t nlists, const infx_wm_list* lists, uint32_t owned_n, int32_t depth, int32_t max_results, uint32_t nLong) {
    if (nd > nq || max_results < 1 || depth < 1 || depth > ix->cfg.max_depth) return fail(INFX_EINVAL, "bad batch shape%s");
    bool anyWm = false;
    for (uint32_t i = 0; i < nq; i++) {
        if (fq[i].dev >= (int32_t)nd) return fail(INFX_EINVAL, "fused query refers to a missing Stage-1 query%s");
        if (fq[i].wm_count > INFX_MAX_WM_LISTS || (uint64_t)fq[i].wm_off + fq[i].wm_count > nlists) return fail(INFX_ECAPACITY, "too many WordMatcher lists for one query%s");
        if (fq[i].wm_count) anyWm = true;
        if (fq[i].flags & INFX_FQ_WMDEV) {
            if (!ix->haveDict) return fail(INFX_EINVAL, "INFX_FQ_WMDEV needs infx_upload_wm_dictionary%s");
            if (fq[i].wm_count) return fail(INFX_EINVAL, "a query takes its WordMatcher lists either from the caller or from the device lookup%s");
            if ((fq[i].flags & INFX_FQ_COV) && !(fq[i].flags & INFX_FQ_SKIP) && wm_words(cq[i]) * (uint32_t)(3 + 2 * ix->lk->maxLd1) > INFX_MAX_WM_LISTS)
                return fail(INFX_ECAPACITY, "too many words for the device WordMatcher lookup (hand the lists over instead)%s");
        }
        if ((fq[i].flags & INFX_FQ_COV) && !(fq[i].flags & INFX_FQ_SKIP) && cq[i].reserved != 0) {
            if (cq[i].reserved < 0 || (uint32_t)cq[i].reserved > nLong) return fail(INFX_EINVAL, "query refers to a missing long-query record (infx_stage2_long_queries)%s");
            if (fq[i].flags & INFX_FQ_WMDEV) return fail(INFX_EINVAL, "the device WordMatcher lookup reads the words of a fast-envelope query: hand the lists of a long query over%s");
        } else if ((fq[i].flags & INFX_FQ_COV) && !(fq[i].flags & INFX_FQ_SKIP) &&
            (cq[i].num_tokens > INFX_MAX_QUERY_TOKENS || cq[i].text_len > INFX_MAX_QUERY_CHARS || cq[i].num_fusion_tokens > 2 * INFX_MAX_QUERY_TOKENS))
            return fail(INFX_EUNSUPPORTED, "query exceeds the Stage-2 envelope (hand it over as a long query: infx_stage2_long_queries)%s");
    }
    for (uint32_t l = 0; l < nlists; l++) {
        const infx_wm_list& L = lists[l];
        const uint64_t lim = L.src == 0 ? ix->nWmExact : (L.src == 1 ? ix->nWmLd1 : (L.src == 2 ? owned_n : 0));
        if (L.off + L.len > lim) return fail(INFX_EINVAL, "WordMatcher list out of range%s");
    }
    if (anyWm && !ix->haveWm) return fail(INFX_EINVAL, "infx_upload_wordmatcher has not been called%s");
    return INFX_OK;
}

// k_rules (from the class histogram in s->dCounts) + k_select -> s->dHits / s->dHitCount (stride = depth)
static int32_t fused_enqueue_select(infx_stream* s, uint32_t nd, int32_t depth, bool shardNext = false, bool markTurn = false) {
    infx_index* ix = s->ix;
    GROW(s->dRules, s->capRules, std::max<size_t>(1, nd) * sizeof(SelRule));
    GROW(s->dHits, s->capHits, std::max<size_t>(1, (size_t)nd) * depth * sizeof(infx_hit));
    GROW(s->dHitCount, s->capHitCount, std::max<size_t>(1, nd) * 4);
    GROW(s->dFQueries, s->capFQueries, std::max<size_t>(1, nd) * sizeof(infx_query));
    UP(s->dFQueries, s->lastQ.data(), (size_t)nd * sizeof(infx_query));
    HIPCHK(hipMemsetAsync(s->dHitCount, 0, std::max<size_t>(1, nd) * 4, s->st));
    Arena ar = make_arena(s);
    HIPCHK(hipEventRecord(s->evS0, s->st));
    if (nd) {
        k_rules<<<(nd + 255) / 256, 256, 0, s->st>>>((const infx_query*)s->dFQueries, (const uint32_t*)s->dCounts, (SelRule*)s->dRules, nd);
        const bool exact = !shardNext && exact_possible(s);
        if (exact) HIPCHK(hipMemsetAsync(s->dExactStat + 4, 0, 16, s->st));
        if (shardNext) GROW(s->dNext, s->capNext, (size_t)nd * 4);       // document shards: no local flags — the cut is global (k_gflag)
#ifdef SEL_PROF
        static unsigned long long* dProf = nullptr; static int profCalls = 0;
        if (getenv("INFX_SEL_PROF") && nd >= 500) { if (!dProf) { hipMalloc((void**)&dProf, 4096 * 64); hipMemcpyToSymbol(HIP_SYMBOL(g_selProf), &dProf, sizeof(dProf)); } hipMemsetAsync(dProf, 0, (size_t)nd * 64, s->st); }
#endif
        const uint32_t* selOrd = select_order(s, nd);
        const SelGiant selG = select_giants(s, ar, nd, selOrd);
        k_select<<<nd, SEL_THREADS, 0, s->st>>>(ar, ix->d.nRanges, (const SelRule*)s->dRules, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, depth, exact ? (uint32_t*)s->dExactFlag : nullptr, exact ? s->dExactStat + 4 : nullptr,
                                                shardNext ? (float*)s->dNext : nullptr, selOrd, selG);
#ifdef SEL_PROF
        if (dProf && nd >= 500 && ++profCalls == 6) { static const char* const ph[5] = {"hist1", "hist2", "gather", "sort", "write"}; wgprof_dump("selprof", dProf, nd, ph, s->st); }
#endif
        if (markTurn) { HIPCHK(hipEventRecord(s->evTurn, s->st)); markTurn = false; }      // the wide phase of this batch ends here (round 6 measured the event in FRONT of k_select — the next batch's accumulation beside this k_select's tail of giant queries: 93.7 / 94.2 k against 93.8 / 94.3 k queries/s, nothing)
        if (exact) { int32_t rc_ = mark_wide_queries(s, ix->cfg.max_depth); if (rc_) return rc_; }
        if (exact) { int32_t rc_ = enqueue_exact(s, nd, depth); if (rc_) return rc_; }
    }
    if (markTurn) HIPCHK(hipEventRecord(s->evTurn, s->st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evS1, s->st));
    s->timedSel = true;
    return INFX_OK;
}

// uploads the per-query tables, then k_prep2 (merging W per-shard hit lists) + k_stage2 -> s->dCovC / s->dCovO / s->dFMeta / s->dFS1
static int32_t fused_enqueue_prep_stage2(infx_stream* s, int W, uint32_t nd, const infx_hit* dHitsAll, const uint32_t* dHcAll,
                                         uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq, uint32_t nlists, const infx_wm_list* lists,
                                         uint32_t owned_n, const int32_t* owned, int32_t depth, int32_t want_debug) {
    infx_index* ix = s->ix;
    const uint32_t stride = 2u * (uint32_t)depth, ncand = nq * stride;
    const uint32_t Dp = pow2_at_least((uint32_t)depth, 8);
    const uint32_t Dall = W > 1 ? pow2_at_least((uint32_t)W * (uint32_t)depth, 8) : 0;
    if (Dall > 8192) return fail(INFX_ECAPACITY, "shards x depth exceeds the in-LDS merge (8192 rows); merge hierarchically%s");
    GROW(s->dFQ, s->capFQ, (size_t)nq * sizeof(infx_fused_query));
    GROW(s->dFLists, s->capFLists, std::max<size_t>(1, nlists) * sizeof(infx_wm_list));
    GROW(s->dFOwned, s->capFOwned, ((size_t)owned_n + 1) * 4);
    GROW(s->dFS1, s->capFS1, (size_t)nq * depth * sizeof(infx_hit));
    GROW(s->dFMeta, s->capFMeta, (size_t)nq * sizeof(FusedMeta));
    GROW(s->dCovQ, s->capCovQ, (size_t)nq * sizeof(infx_cov_query));
    GROW(s->dCovC, s->capCovC, (size_t)ncand * sizeof(infx_cov_cand));
    GROW(s->dCovO, s->capCovO, (size_t)ncand * sizeof(infx_cov_out));
    GROW(s->dFPairs, s->capFPairs, (size_t)ncand * 4);
    if (want_debug) GROW(s->dCovF, s->capCovF, (size_t)ncand * INFX_NFEAT * 4);
    // queries flagged INFX_FQ_WMDEV get their WordMatcher lists from k_wm: a block of INFX_MAX_WM_LISTS descriptors per query behind the caller's lists,
    // one WM_AFFIX_CAP region of `owned` per looked-up word behind the caller's ids (the regions are assigned here: `reserved` = first word slot)
    std::vector<infx_fused_query> fqDev; uint64_t wmSlots = 0;
    for (uint32_t i = 0; i < nq; i++) if (fq[i].flags & INFX_FQ_WMDEV) {
        if (fqDev.empty()) fqDev.assign(fq, fq + nq);
        fqDev[i].reserved = (int32_t)wmSlots;
        if ((fq[i].flags & INFX_FQ_COV) && !(fq[i].flags & INFX_FQ_SKIP)) wmSlots += wm_words(cq[i]);
    }
    const bool wmDev = !fqDev.empty();
    if (wmDev) {
        if (wmSlots * WM_AFFIX_CAP + owned_n > 0x7FFFFFF0ull) return fail(INFX_ECAPACITY, "device WordMatcher lookup: too many query words in one batch%s");
        GROW(s->dFLists, s->capFLists, ((size_t)nlists + (size_t)nq * INFX_MAX_WM_LISTS) * sizeof(infx_wm_list));
        GROW(s->dFOwned, s->capFOwned, ((size_t)owned_n + (size_t)wmSlots * WM_AFFIX_CAP + 1) * 4);
    }
    UP(s->dFQ, wmDev ? fqDev.data() : fq, (size_t)nq * sizeof(infx_fused_query));
    UP(s->dFLists, lists, (size_t)nlists * sizeof(infx_wm_list));
    UP(s->dFOwned, owned, (size_t)owned_n * 4);
    UP(s->dCovQ, cq, (size_t)nq * sizeof(infx_cov_query));
    s->batchAlias = s->longAlias;      // (host scan of the query texts: ~20 characters per query)
    for (uint32_t i = 0; i < nq && !s->batchAlias; i++) if (cq[i].reserved == 0) for (int32_t k = 0; k < cq[i].text_len && k < (int32_t)INFX_MAX_QUERY_CHARS; k++) if (s2_host_is_alias(cq[i].text[k])) { s->batchAlias = true; break; }
    HIPCHK(hipMemsetAsync(s->dCovO, 0, (size_t)ncand * sizeof(infx_cov_out), s->st));
    HIPCHK(hipEventRecord(s->evP0, s->st));
    if (wmDev) {
        k_wm<<<nq, WAVE, 0, s->st>>>(*ix->lk, (infx_fused_query*)s->dFQ, (const infx_cov_query*)s->dCovQ, nq, (infx_wm_list*)s->dFLists, nlists, (int32_t*)s->dFOwned, (uint64_t)owned_n);
        HIPCHK(hipGetLastError());
    }
    {
        const size_t Pp = std::max<size_t>(Dp, P2_CAP);
        const size_t lds = (size_t)Dp * 4 * 4 + Pp * 4 + (size_t)P2_CAP * 4 + (size_t)Dp * 8 + (size_t)Dp * 2 + Pp + (size_t)P2_MAXLISTS * (8 + 4 + 4 + 4) + (P2_THREADS + 2) * 4 + (size_t)Dall * 8 + 64;
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_prep2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef SEL_PROF
        static unsigned long long* dP2 = nullptr; static int p2Calls = 0;
        if (getenv("INFX_SEL_PROF") && nq >= 500) { if (!dP2) { hipMalloc((void**)&dP2, 4096 * 64); hipMemcpyToSymbol(HIP_SYMBOL(g_p2Prof), &dP2, sizeof(dP2)); } hipMemsetAsync(dP2, 0, (size_t)nq * 64, s->st); }
#endif
        k_prep2<<<nq, P2_THREADS, lds, s->st>>>(ix->d, dHitsAll, dHcAll, depth, W, (int)nd, (int)Dall, (const infx_fused_query*)s->dFQ,
                                                 (const infx_wm_list*)s->dFLists, (const int32_t*)s->dFOwned, depth, (int)Dp,
                                                 (infx_hit*)s->dFS1, (infx_cov_cand*)s->dCovC, (int32_t*)s->dFPairs, (FusedMeta*)s->dFMeta);
#ifdef SEL_PROF
        if (dP2 && nq >= 500 && ++p2Calls == 6) { static const char* const ph[5] = {"s1sort", "topsort", "overlap", "wmrounds", "emit"}; wgprof_dump("p2prof", dP2, nq, ph, s->st); }
#endif
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evP1, s->st));
    HIPCHK(hipEventRecord(s->evC0, s->st));
    S2_LAUNCH_FAST(ix->d, (const infx_cov_query*)s->dCovQ, nq, (const infx_cov_cand*)s->dCovC, ncand,
                                                                                 (infx_cov_out*)s->dCovO, want_debug ? (int32_t*)s->dCovF : nullptr, 1, 0, (const int32_t*)s->dFPairs);
    S2_LAUNCH_SLOW(ix->d, (const infx_cov_query*)s->dCovQ, nq, (const infx_cov_cand*)s->dCovC, ncand,
                                                                                 (infx_cov_out*)s->dCovO, want_debug ? (int32_t*)s->dCovF : nullptr, 1, 1, (const int32_t*)s->dFPairs);
    { int32_t rc_ = s2_huge_ready(s); if (rc_) return rc_; }
    S2_LAUNCH_HUGE(ix->d, (const infx_cov_query*)s->dCovQ, nq, (const infx_cov_cand*)s->dCovC, ncand,
                                                                                 (infx_cov_out*)s->dCovO, want_debug ? (int32_t*)s->dCovF : nullptr, 1, 1, (const int32_t*)s->dFPairs);
    S2_LAUNCH_LONGQ(ix->d, (const infx_cov_query*)s->dCovQ, nq, (const infx_cov_cand*)s->dCovC, ncand,
                                                                                 (infx_cov_out*)s->dCovO, want_debug ? (int32_t*)s->dCovF : nullptr, 1, 1, (const int32_t*)s->dFPairs);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evC1, s->st));
    s->timedCov = true;
    s->fusedNq = nq; s->fusedDepth = depth; s->fusedDebug = want_debug != 0;
    return INFX_OK;
}

// k_finalize over s->dCovC / s->dCovO / s->dFMeta / s->dFS1 -> result rows in s->dFKeys ...
static int32_t fused_enqueue_finalize(infx_stream* s, uint32_t nq, int32_t depth, int32_t max_results, bool ties) {
    infx_index* ix = s->ix;
    const uint32_t Cp = pow2_at_least(2u * (uint32_t)depth, 8);
    GROW(s->dFKeys, s->capFKeys, (size_t)nq * max_results * 8);
    GROW(s->dFScores, s->capFScores, (size_t)nq * max_results * 4);
    GROW(s->dFTies, s->capFTies, (size_t)nq * max_results);
    GROW(s->dFCounts, s->capFCounts, (size_t)nq * 4);
    GROW(s->dFFlags, s->capFFlags, (size_t)nq * 4);
    GROW(s->dFErr, s->capFErr, 4);
    HIPCHK(hipMemsetAsync(s->dFErr, 0, 4, s->st));
    HIPCHK(hipEventRecord(s->evF0, s->st));
    const size_t lds = (size_t)Cp * (8 + 4 + 4 + 2 + 1 + 1) + (P2_THREADS + 1) * 4 + 64;
    const bool post = s->postFilter != nullptr || s->nFacet > 0;
    if (post && max_results > INFX_FILTER_MAX_ROWS) return fail(INFX_EUNSUPPORTED, "post-filter / facets run on at most INFX_FILTER_MAX_ROWS returned rows per query%s");
    if (post) GROW(s->dFDocs, s->capFDocs, (size_t)nq * max_results * 4);
    k_finalize<<<nq, P2_THREADS, lds, s->st>>>(ix->d, (const infx_fused_query*)s->dFQ, (const FusedMeta*)s->dFMeta, (const infx_cov_cand*)s->dCovC,
                                                (const infx_cov_out*)s->dCovO, (const infx_hit*)s->dFS1, depth, (int)Cp, max_results,
                                                (long long*)s->dFKeys, (float*)s->dFScores, ties ? (uint8_t*)s->dFTies : nullptr,
                                                (uint32_t*)s->dFCounts, (uint32_t*)s->dFFlags, (uint32_t*)s->dFErr, post ? (int32_t*)s->dFDocs : nullptr);
    HIPCHK(hipGetLastError());
    s->facetNq = 0;
    if (post) {     // ResultProcessor.ApplyFilter + FacetBuilder on the rows just produced, before they leave the device
        const size_t fe = (size_t)nq * std::max<uint32_t>(1, s->nFacet) * INFX_FILTER_MAX_ROWS;
        GROW(s->dFacCodes, s->capFacCodes, fe * 4); GROW(s->dFacCounts, s->capFacCounts, fe * 4); GROW(s->dFacN, s->capFacN, (size_t)nq * std::max<uint32_t>(1, s->nFacet) * 4);
        if (!s->dFacetCols) HIPCHK(hipMalloc(&s->dFacetCols, INFX_MAX_FACET_COLS * 4));
        UP(s->dFacetCols, s->facetCols, INFX_MAX_FACET_COLS * 4);
        DevColumns cols; for (int c = 0; c < FILT_MAXCOL; c++) cols.codes[c] = ix->colCodes[c];
        k_postfilter<<<nq, WAVE, 0, s->st>>>(s->postFilter ? s->postFilter->d : DevFilter{}, s->postFilter ? 1 : 0, cols, max_results, (long long*)s->dFKeys, (float*)s->dFScores,
                                            ties ? (uint8_t*)s->dFTies : nullptr, (int32_t*)s->dFDocs, (uint32_t*)s->dFCounts, (int)s->nFacet, (const uint32_t*)s->dFacetCols,
                                            (uint32_t*)s->dFacCodes, (uint32_t*)s->dFacCounts, (uint32_t*)s->dFacN);
        HIPCHK(hipGetLastError());
        if (s->nFacet) {
            s->hFacCodes.resize(fe); s->hFacCounts.resize(fe); s->hFacN.resize((size_t)nq * s->nFacet); s->facetNq = nq;
            DOWN(s->hFacCodes.data(), s->dFacCodes, fe * 4); DOWN(s->hFacCounts.data(), s->dFacCounts, fe * 4); DOWN(s->hFacN.data(), s->dFacN, (size_t)nq * s->nFacet * 4);
        }
    }
    HIPCHK(hipEventRecord(s->evF1, s->st));
    s->timedFused = true;
    return INFX_OK;
}

// result rows -> caller buffers (after the stream is synchronised); rows beyond a query's count stay untouched
struct FusedResultStage { std::vector<int64_t> k; std::vector<float> sc; std::vector<uint8_t> t; };
static int32_t fused_download_results(infx_stream* s, uint32_t nq, int32_t max_results, FusedResultStage& R, bool ties, uint32_t* out_counts, uint32_t* out_flags, uint32_t* err) {
    R.k.resize((size_t)nq * max_results); R.sc.resize((size_t)nq * max_results); R.t.resize(ties ? (size_t)nq * max_results : 0);
    DOWN(R.k.data(), s->dFKeys, (size_t)nq * max_results * 8);
    DOWN(R.sc.data(), s->dFScores, (size_t)nq * max_results * 4);
    if (ties) DOWN(R.t.data(), s->dFTies, (size_t)nq * max_results);
    DOWN(out_counts, s->dFCounts, (size_t)nq * 4);
    if (out_flags) DOWN(out_flags, s->dFFlags, (size_t)nq * 4);
    DOWN(err, s->dFErr, 4);
    return INFX_OK;
}
static void fused_scatter_results(uint32_t nq, int32_t max_results, const FusedResultStage& R, int64_t* out_keys, float* out_scores, uint8_t* out_ties, const uint32_t* out_counts) {
    for (uint32_t i = 0; i < nq; i++) {
        const size_t o = (size_t)i * max_results, c = std::min<size_t>(out_counts[i], (size_t)max_results);
        std::memcpy(out_keys + o, R.k.data() + o, c * 8); std::memcpy(out_scores + o, R.sc.data() + o, c * 4);
        if (out_ties) std::memcpy(out_ties + o, R.t.data() + o, c);
    }
}
static void fused_take_metas(infx_stream* s, const std::vector<FusedMeta>& metas) {
    s->fusedS1 = s->fusedCands = s->fusedTextBytes = 0;
    for (auto& m : metas) { s->fusedS1 += m.s1Count; s->fusedCands += m.candCount; s->fusedTextBytes += (uint64_t)m.pad0 + ((uint64_t)m.pad1 << 32); }
}

int32_t infx_wm_lookup_debug(infx_stream* s, const infx_cov_query* cq, infx_wm_list* lists_out, uint32_t* nlists_out, int32_t* owned_out, uint64_t owned_cap) {
    if (!s || !cq || !lists_out || !nlists_out || (owned_cap && !owned_out)) return fail(INFX_EINVAL, "null argument%s");
    infx_index* ix = s->ix;
    if (!ix->haveDict) return fail(INFX_EINVAL, "infx_upload_wm_dictionary has not been called%s");
    const uint32_t words = wm_words(*cq);
    if (cq->text_len > INFX_MAX_QUERY_CHARS || cq->num_fusion_tokens > 2 * INFX_MAX_QUERY_TOKENS) return fail(INFX_EUNSUPPORTED, "query exceeds the Stage-2 envelope%s");
    if (words * (uint32_t)(3 + 2 * ix->lk->maxLd1) > INFX_MAX_WM_LISTS) return fail(INFX_ECAPACITY, "too many words for the device WordMatcher lookup%s");
    if ((uint64_t)words * WM_AFFIX_CAP > owned_cap) return fail(INFX_EINVAL, "owned_out too small%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    infx_fused_query fq{}; fq.dev = -1; fq.flags = INFX_FQ_COV | INFX_FQ_WMDEV; fq.max_results = 1; fq.reserved = 0;
    GROW(s->dFQ, s->capFQ, sizeof(infx_fused_query));
    GROW(s->dFLists, s->capFLists, (size_t)INFX_MAX_WM_LISTS * sizeof(infx_wm_list));
    GROW(s->dFOwned, s->capFOwned, ((size_t)words * WM_AFFIX_CAP + 1) * 4);
    GROW(s->dCovQ, s->capCovQ, sizeof(infx_cov_query));
    UP(s->dFQ, &fq, sizeof fq); UP(s->dCovQ, cq, sizeof *cq);
    k_wm<<<1, WAVE, 0, s->st>>>(*ix->lk, (infx_fused_query*)s->dFQ, (const infx_cov_query*)s->dCovQ, 1, (infx_wm_list*)s->dFLists, 0, (int32_t*)s->dFOwned, 0ull);
    HIPCHK(hipGetLastError());
    infx_fused_query back{};
    DOWN(&back, s->dFQ, sizeof back);
    DOWN(lists_out, s->dFLists, (size_t)INFX_MAX_WM_LISTS * sizeof(infx_wm_list));
    if (words) DOWN(owned_out, s->dFOwned, (size_t)words * WM_AFFIX_CAP * 4);
    SYNC();
    *nlists_out = back.wm_count;
    return INFX_OK;
}

int32_t infx_search_fused(infx_stream* s, uint32_t nd, const infx_query* q, uint32_t nterms, const infx_term* terms,
                          uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq,
                          uint32_t nlists, const infx_wm_list* lists, uint32_t owned_n, const int32_t* owned,
                          int32_t depth, int32_t max_results, int32_t want_debug,
                          int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags) {
    if (!s || (nd && (!q || (nterms && !terms))) || (nq && (!fq || !cq || !out_keys || !out_scores || !out_counts)) || (nlists && !lists) || (owned_n && !owned))
        return fail(INFX_EINVAL, "null argument%s");
    infx_index* ix = s->ix;
    if (!ix->havePostings || !ix->haveDocs || !ix->d.text) return fail(INFX_EINVAL, "index not uploaded%s");
    if (ix->nranks > 1 || ix->d.docBase != 0) return fail(INFX_EINVAL, "infx_search_fused needs an unsharded index (sharded engines use infx_shard_*)%s");
    if (nq == 0) return INFX_OK;
    { int32_t rc_ = fused_check_queries(ix, nd, nq, fq, cq, nlists, lists, owned_n, depth, max_results, s->nLongQ); if (rc_) return rc_; }
    for (uint32_t i = 0; i < nd; i++) if (q[i].depth != depth) return fail(INFX_EINVAL, "all queries of a fused batch share one depth%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    // Turnstile.  k_accumulate and k_select fill the GPU on their own; the replay, candidate assembly and Stage 2 behind them are narrow.  Sessions that
    // submit together run their wide kernels against each other and then sit in their narrow phases together — convoys that leave the GPU half empty (the
    // bench timeline showed three batches finishing within a millisecond of each other, then nothing wide to run).  Each batch's k_accumulate therefore waits
    // (stream-side, no host wait) for the k_select of the batch submitted before it on this index: wide phases queue up one behind the other, each batch's
    // narrow tail overlaps the next batch's wide phase.  INFX_TURNSTILE=0 switches it off.
    static const bool turnstile = [] { const char* e = getenv("INFX_TURNSTILE"); return !(e && e[0] == '0'); }();
    {
        std::unique_lock<std::mutex> turn(ix->turnMu, std::defer_lock);
        // (round 6 measured the wait BETWEEN k_accumulate_sparse and the streaming k_accumulate — the address-path-bound kernel of this batch beside the vector-bound
        //  kernels of the batch before: no difference, 94.7 / 94.4 k against 95.8 / 94.3 k queries/s over 200 batches; the streaming kernel's four 128-register
        //  waves per SIMD leave no room for a second kernel's waves)
        if (turnstile) { turn.lock(); if (ix->turnEvent && ix->turnEvent != s->evTurn) HIPCHK(hipStreamWaitEvent(s->st, ix->turnEvent, 0)); }
        if (nd) { int32_t rc_ = acc_enqueue(s, nd, q, nterms, terms, 0, nullptr); if (rc_) return rc_; }
        else { HIPCHK(hipEventRecord(s->evA0, s->st)); HIPCHK(hipEventRecord(s->evA1, s->st)); }
        { int32_t rc_ = fused_enqueue_select(s, nd, depth, false, turnstile); if (rc_) return rc_; }
        if (turnstile) ix->turnEvent = s->evTurn;
    }
    { int32_t rc_ = fused_enqueue_prep_stage2(s, 1, nd, (const infx_hit*)s->dHits, (const uint32_t*)s->dHitCount, nq, fq, cq, nlists, lists, owned_n, owned, depth, want_debug); if (rc_) return rc_; }
    { int32_t rc_ = fused_enqueue_finalize(s, nq, depth, max_results, out_ties != nullptr); if (rc_) return rc_; }
    uint32_t ovf = 0, err = 0; std::vector<unsigned long long> qbytes(nd); std::vector<SelRule> rules(nd); std::vector<FusedMeta> metas(nq);
    FusedResultStage R;
    DOWN(metas.data(), s->dFMeta, (size_t)nq * sizeof(FusedMeta));
    { int32_t rc_ = fused_download_results(s, nq, max_results, R, out_ties != nullptr, out_counts, out_flags, &err); if (rc_) return rc_; }
    if (nd) { DOWN(&ovf, s->dOverflow, 4); DOWN(qbytes.data(), s->dQBytes, (size_t)nd * 8); DOWN(rules.data(), s->dRules, (size_t)nd * sizeof(SelRule)); }
    SYNC();
    if (ovf) return fail(INFX_ECAPACITY, "candidate arena overflow (bound violated)%s");
    (void)err;     // candidates outside the Stage-2 envelope are skipped per query (result flag bit 3), they no longer fail the batch
    fused_scatter_results(nq, max_results, R, out_keys, out_scores, out_ties, out_counts);
    s->lastAlgBytes = 0; for (auto b : qbytes) s->lastAlgBytes += b;
    s->lastCandTotal = 0; for (auto& r : rules) s->lastCandTotal += r.total;
    fused_take_metas(s, metas);
    return INFX_OK;
}

// ---- document-sharded operation: the same device stages with the collectives of SURVEY 8(e) in between (host buffers) ----------
int32_t infx_shard_select(infx_stream* s, uint32_t nd, const infx_counts* global_counts, int32_t depth, infx_hit* hits_out, uint32_t* hitcount_out, float* next_out) {
    if (!s || (nd && (!global_counts || !hits_out || !hitcount_out))) return fail(INFX_EINVAL, "null argument%s");
    if (s) { s->shSelected = false; s->shNd = 0; s->shHead[0] = s->shHead[1] = s->shHead[2] = 0; }
    if (nd == 0) return INFX_OK;
    if (nd != s->lastNq) return fail(INFX_EINVAL, "infx_shard_select must follow infx_stage1_accumulate of the same batch%s");
    infx_index* ix = s->ix;
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    UPX(s->dCounts, global_counts, (size_t)nd * INFX_NCLASS * 4);         // the tier rules see the GLOBAL cardinalities (quirk Q11); host or device memory
    { int32_t rc_ = fused_enqueue_select(s, nd, depth, true); if (rc_) return rc_; }
    std::vector<SelRule> rules(nd);
    DOWNX(hits_out, s->dHits, (size_t)nd * depth * sizeof(infx_hit));
    DOWNX(hitcount_out, s->dHitCount, (size_t)nd * 4);
    if (next_out) DOWNX(next_out, s->dNext, (size_t)nd * 4);
    s->shSelected = true; s->shNd = nd; s->shDepth = depth;
    DOWN(rules.data(), s->dRules, (size_t)nd * sizeof(SelRule));
    SYNC();
    s->lastCandTotal = 0; for (auto& r : rules) s->lastCandTotal += r.total;
    return INFX_OK;
}

int32_t infx_shard_stage2(infx_stream* s, int32_t nshards, uint32_t nd, const infx_hit* all_hits, const uint32_t* all_hitcounts,
                          uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq, uint32_t nlists, const infx_wm_list* lists,
                          uint32_t owned_n, const int32_t* owned, int32_t depth, int32_t max_results, int32_t want_debug, infx_cov_out* outs_out) {
    if (!s || nshards < 1 || (nd && (!all_hits || !all_hitcounts)) || (nq && (!fq || !cq || !outs_out)) || (nlists && !lists) || (owned_n && !owned))
        return fail(INFX_EINVAL, "null argument%s");
    infx_index* ix = s->ix;
    if (!ix->haveDocs || !ix->d.text || !ix->d.docKeyAll) return fail(INFX_EINVAL, "index not uploaded%s");
    if (nq == 0) return INFX_OK;
    { int32_t rc_ = fused_check_queries(ix, nd, nq, fq, cq, nlists, lists, owned_n, depth, max_results, s->nLongQ); if (rc_) return rc_; }
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    const size_t nh = (size_t)nshards * std::max<size_t>(1, nd) * depth;
    GROW(s->dFHitsAll, s->capFHitsAll, nh * sizeof(infx_hit));
    GROW(s->dFHcAll, s->capFHcAll, (size_t)nshards * std::max<size_t>(1, nd) * 4);
    UPX(s->dFHitsAll, all_hits, (size_t)nshards * nd * depth * sizeof(infx_hit));
    UPX(s->dFHcAll, all_hitcounts, (size_t)nshards * nd * 4);
    { int32_t rc_ = fused_enqueue_prep_stage2(s, nshards == 1 ? 1 : nshards, nd, (const infx_hit*)s->dFHitsAll, (const uint32_t*)s->dFHcAll, nq, fq, cq, nlists, lists, owned_n, owned, depth, want_debug); if (rc_) return rc_; }
    std::vector<FusedMeta> metas(nq);
    DOWN(metas.data(), s->dFMeta, (size_t)nq * sizeof(FusedMeta));
    DOWNX(outs_out, s->dCovO, (size_t)nq * 2 * depth * sizeof(infx_cov_out));
    SYNC();
    fused_take_metas(s, metas);
    return INFX_OK;
}

int32_t infx_shard_finalize(infx_stream* s, uint32_t nq, const infx_cov_out* merged_outs, int32_t depth, int32_t max_results,
                            int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags) {
    if (!s || (nq && (!merged_outs || !out_keys || !out_scores || !out_counts))) return fail(INFX_EINVAL, "null argument%s");
    if (nq == 0) return INFX_OK;
    if (nq != s->fusedNq || depth != s->fusedDepth) return fail(INFX_EINVAL, "infx_shard_finalize must follow infx_shard_stage2 of the same batch%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    UPX(s->dCovO, merged_outs, (size_t)nq * 2 * depth * sizeof(infx_cov_out));
    { int32_t rc_ = fused_enqueue_finalize(s, nq, depth, max_results, out_ties != nullptr); if (rc_) return rc_; }
    uint32_t err = 0; FusedResultStage R;
    { int32_t rc_ = fused_download_results(s, nq, max_results, R, out_ties != nullptr, out_counts, out_flags, &err); if (rc_) return rc_; }
    SYNC();
    (void)err;     // candidates outside the Stage-2 envelope are skipped per query (result flag bit 3), they no longer fail the batch
    fused_scatter_results(nq, max_results, R, out_keys, out_scores, out_ties, out_counts);
    return INFX_OK;
}

// ---- exact Stage-1 replay across document shards (exactsh.hip.inc) ----------------------------------------------------------------------------------
static bool shard_exact_possible(infx_stream* s) { return exact_enabled(s->ix) && s->maskWords > 0 && s->ix->cfg.max_depth <= EXS_MAXDEPTH; }

int32_t infx_shard_replay_local(infx_stream* s, int32_t nshards, uint32_t nd, const infx_hit* all_hits, const uint32_t* all_hitcounts, const float* all_next,
                                int32_t depth, uint64_t* blob_bytes) {
    if (!s || nshards < 1 || !blob_bytes || (nd && (!all_hits || !all_hitcounts || !all_next))) return fail(INFX_EINVAL, "null argument%s");
    *blob_bytes = shx_blob_bytes(nd, 0, 0);
    infx_index* ix = s->ix;
    if (nd == 0) { s->shHead[0] = 0; s->shHead[1] = s->shHead[2] = 0; return INFX_OK; }
    if (!s->shSelected || nd != s->shNd || depth != s->shDepth) return fail(INFX_EINVAL, "infx_shard_replay_local must follow infx_shard_select of the same batch%s");
    if (nshards != ix->nranks) return fail(INFX_EINVAL, "nshards differs from infx_set_shard%s");
    if (ix->nranks > 1 && (ix->d.docBase & 0xFFFF)) return fail(INFX_EINVAL, "exact replay across shards needs shard boundaries at multiples of 65536 documents (whole Roaring containers)%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    const uint32_t Dall = pow2_at_least((uint32_t)nshards * (uint32_t)depth, 8);
    if (Dall > 16384) return fail(INFX_ECAPACITY, "shards x depth exceeds the in-LDS merge (16384 rows)%s");
    const size_t nh = (size_t)nshards * nd * depth;
    GROW(s->dFHitsAll, s->capFHitsAll, nh * sizeof(infx_hit));
    GROW(s->dFHcAll, s->capFHcAll, (size_t)nshards * nd * 4);
    GROW(s->dAllNext, s->capAllNext, (size_t)nshards * nd * 4);
    GROW(s->dPrior, s->capPrior, (size_t)nd * EXS_MAXDEPTH * 4);
    UPX(s->dFHitsAll, all_hits, nh * sizeof(infx_hit));
    UPX(s->dFHcAll, all_hitcounts, (size_t)nshards * nd * 4);
    UPX(s->dAllNext, all_next, (size_t)nshards * nd * 4);
    const bool possible = shard_exact_possible(s);
    HIPCHK(hipMemsetAsync(s->dExactStat, 0, 32, s->st));
    HIPCHK(hipEventRecord(s->evX0, s->st));
    {
        const size_t lds = (size_t)Dall * 8 + 257 * 4;
        static std::mutex mu; static size_t attr = 0;
        { std::lock_guard<std::mutex> lk(mu); if (lds > 64 * 1024 && lds > attr) { HIPCHK(hipFuncSetAttribute((const void*)k_gflag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = lds; } }
        k_gflag<<<nd, 256, lds, s->st>>>(ix->d, nshards, (int)nd, depth, (int)Dall, (const infx_hit*)s->dFHitsAll, (const uint32_t*)s->dFHcAll, (const float*)s->dAllNext,
                                        (uint32_t*)s->dExactFlag, (float*)s->dPrior, s->dExactStat + 4, possible ? 1 : 0, (const uint32_t*)s->dWideQ, s->nWide);
    }
    ExBufs xb{};
    if (possible) {
        Arena ar = make_arena(s);
        { int32_t rc_ = exact_chunk_tables(s, nd, xb); if (rc_) return rc_; }
        { int32_t rc_ = launch_scan_and_chunks(s, nd, ar, xb, ix->d.docBase > 0 ? (const float*)s->dPrior : nullptr, false); if (rc_) return rc_; }
        // worst case: every reserved row of every flagged query is emitted
        const size_t need = shx_blob_bytes(nd, s->exChunkCap, 0) + s->exCap * sizeof(infx_hit);
        GROW(s->shBlob, s->capShBlob, need);
        k_exsh_count<<<(nd + 255) / 256, 256, 0, s->st>>>(nd, (const uint32_t*)s->dExactFlag, xb, (ShQHdr*)((unsigned char*)s->shBlob + 16));
        k_exsh_scan<<<1, 256, 0, s->st>>>(nd, (ShQHdr*)((unsigned char*)s->shBlob + 16), (uint32_t*)s->shBlob);
        k_exsh_copy<<<nd, 256, 0, s->st>>>(nd, (const uint32_t*)s->dExactFlag, ar, xb, (unsigned char*)s->shBlob);
        HIPCHK(hipGetLastError());
        DOWN(s->shHead, s->shBlob, 16);
    } else {
        GROW(s->shBlob, s->capShBlob, shx_blob_bytes(nd, 0, 0));
        HIPCHK(hipMemsetAsync(s->shBlob, 0, shx_blob_bytes(nd, 0, 0), s->st));
        const uint32_t head[4] = {nd, 0, 0, 0};
        UP(s->shBlob, head, 16);
        s->shHead[0] = nd; s->shHead[1] = s->shHead[2] = s->shHead[3] = 0;
    }
    HIPCHK(hipGetLastError());
    DOWN(s->lastFlagWhy, s->dExactStat + 4, 16);
    SYNC();
    *blob_bytes = shx_blob_bytes(nd, s->shHead[1], s->shHead[2]);
    return INFX_OK;
}

int32_t infx_shard_replay_blob(infx_stream* s, void* dst, uint64_t padded_bytes) {
    if (!s || !dst) return fail(INFX_EINVAL, "null argument%s");
    const size_t have = shx_blob_bytes(s->shNd, s->shHead[1], s->shHead[2]);
    if (!s->shNd) return INFX_OK;
    if (padded_bytes < have) return fail(INFX_EINVAL, "padded size below this shard's blob%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    DOWNX(dst, s->shBlob, have);       // the padding beyond `have` is never read (the header says how much is valid)
    SYNC();
    return INFX_OK;
}

int32_t infx_shard_replay_merge(infx_stream* s, int32_t nshards, uint32_t nd, const void* all_blobs, uint64_t padded_bytes, int32_t depth,
                                infx_hit* hits_out, uint32_t* hitcount_out) {
    if (!s || nshards < 1 || (nd && (!all_blobs || !hits_out || !hitcount_out))) return fail(INFX_EINVAL, "null argument%s");
    if (nd == 0) return INFX_OK;
    infx_index* ix = s->ix;
    if (!s->shSelected || nd != s->shNd || depth != s->shDepth || nshards != ix->nranks) return fail(INFX_EINVAL, "infx_shard_replay_merge must follow infx_shard_replay_local of the same batch%s");
    if (padded_bytes < shx_blob_bytes(nd, 0, 0)) return fail(INFX_EINVAL, "blob size below the header%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    GROW(s->dAllBlobs, s->capAllBlobs, (size_t)nshards * padded_bytes);
    UPX(s->dAllBlobs, all_blobs, (size_t)nshards * padded_bytes);
    HIPCHK(hipMemsetAsync(s->dExactStat, 0, 16, s->st));
    k_ex_heap_sh<<<nd, WAVE, 0, s->st>>>(nshards, ix->rank, nd, (const unsigned char*)s->dAllBlobs, (size_t)padded_bytes, (const uint32_t*)s->dExactFlag, depth,
                                         exact_slow_only() ? 1 : 0, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, depth, s->dExactStat, ex_heap_in_regs(ix, depth));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->evX1, s->st)); s->timedReplay = true;
    DOWNX(hits_out, s->dHits, (size_t)nd * depth * sizeof(infx_hit));
    DOWNX(hitcount_out, s->dHitCount, (size_t)nd * 4);
    uint32_t st4[4] = {0, 0, 0, 0};
    DOWN(st4, s->dExactStat, 16);
    SYNC();
    s->lastExact2[0] = st4[0]; s->lastExact2[1] = st4[2];       // owned queries replayed here; of the owned ones, handed to the sequential chain
    return INFX_OK;
}

int32_t infx_shard_replay_chain(infx_stream* s, uint32_t nd, const uint32_t* need, int32_t depth, void* state) {
    if (!s || (nd && (!need || !state))) return fail(INFX_EINVAL, "null argument%s");
    if (nd == 0) return INFX_OK;
    infx_index* ix = s->ix;
    if (!s->shSelected || nd != s->shNd || depth != s->shDepth) return fail(INFX_EINVAL, "infx_shard_replay_chain must follow infx_shard_select of the same batch%s");
    if (!shard_exact_possible(s)) return fail(INFX_EINVAL, "no hit masks were kept for this batch: nothing to replay%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    const int depthCap = ix->cfg.max_depth;
    if (depth != depthCap) return fail(INFX_EINVAL, "the chained replay exchanges heaps of infx_config.max_depth entries: search with that depth%s");
    const size_t words = (size_t)nd * (2 + 2 * (size_t)depthCap);
    GROW(s->dChainState, s->capChainState, words * 4);
    GROW(s->dChainNeed, s->capChainNeed, (size_t)nd * 4);
    UPX(s->dChainState, state, words * 4);
    UPX(s->dChainNeed, need, (size_t)nd * 4);
    size_t lds1 = 0; { int32_t rc_ = exact1_lds_ready(ix, s->maskWords, &lds1); if (rc_) return rc_; }
    Arena ar = make_arena(s);
    k_exact1<<<nd, EX_THREADS, lds1, s->st>>>(ix->d, (const DevQuery*)s->dQueries, (const DevRefTerm*)s->dRefTerms, ar, (const SelRule*)s->dRules, (const uint32_t*)s->dExactFlag,
                                             0u, ix->avgdl, (infx_hit*)s->dHits, (uint32_t*)s->dHitCount, depth, depthCap, nullptr, (uint32_t*)s->dChainState, (const uint32_t*)s->dChainNeed);
    HIPCHK(hipGetLastError());
    DOWNX(state, s->dChainState, words * 4);
    SYNC();
    return INFX_OK;
}

int32_t infx_upload_doc_keys_all(infx_index* ix, uint32_t total_docs, const int64_t* keys) {
    if (!ix || (total_docs && !keys)) return fail(INFX_EINVAL, "null argument%s");
    HIPCHK(enter_device(ix->cfg.device));
    int64_t* d = nullptr;
    HIPCHK(dalloc(ix, &d, (size_t)total_docs + 1));
    if (total_docs) HIPCHK(hipMemcpy(d, keys, (size_t)total_docs * 8, hipMemcpyHostToDevice));
    ix->d.docKeyAll = d;
    return INFX_OK;
}

int32_t infx_fused_debug(infx_stream* s, infx_hit* s1, uint32_t* s1_counts, infx_cov_cand* cands, infx_cov_out* outs, int32_t* feat,
                         uint32_t* cand_counts, uint32_t* run_cov, int32_t* idx01) {
    if (!s || !s->fusedNq) return fail(INFX_EINVAL, "no fused batch to read back%s");
    if (feat && !s->fusedDebug) return fail(INFX_EINVAL, "the last fused batch ran without want_debug%s");
    HIPCHK(enter_device(s->ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    const uint32_t nq = s->fusedNq; const size_t depth = (size_t)s->fusedDepth, ncand = (size_t)nq * 2 * depth;
    std::vector<FusedMeta> metas(nq);
    if (s1) DOWN(s1, s->dFS1, (size_t)nq * depth * sizeof(infx_hit));
    if (cands) DOWN(cands, s->dCovC, ncand * sizeof(infx_cov_cand));
    if (outs) DOWN(outs, s->dCovO, ncand * sizeof(infx_cov_out));
    if (feat) DOWN(feat, s->dCovF, ncand * INFX_NFEAT * 4);
    DOWN(metas.data(), s->dFMeta, (size_t)nq * sizeof(FusedMeta));
    SYNC();
    for (uint32_t i = 0; i < nq; i++) {
        if (s1_counts) s1_counts[i] = metas[i].s1Count;
        if (cand_counts) cand_counts[i] = metas[i].candCount;
        if (run_cov) run_cov[i] = metas[i].runCov | (metas[i].wmAny << 1);
        if (idx01) { idx01[2 * i] = metas[i].idx0; idx01[2 * i + 1] = metas[i].idx1; }
    }
    return INFX_OK;
}

int32_t infx_last_fused_stats(infx_stream* s, uint64_t* s1_rows, uint64_t* stage2_rows, uint64_t* stage2_text_bytes) {
    if (!s) return fail(INFX_EINVAL, "null argument%s");
    if (s1_rows) *s1_rows = s->fusedS1; if (stage2_rows) *stage2_rows = s->fusedCands; if (stage2_text_bytes) *stage2_text_bytes = s->fusedTextBytes;
    return INFX_OK;
}

int32_t infx_last_fused_timings(infx_stream* s, float* ms5) {
    if (!s || !ms5) return fail(INFX_EINVAL, "null argument%s");
    if (s->timedFused) {
        hipEventElapsedTime(&s->msFused[0], s->evA0, s->evA1); hipEventElapsedTime(&s->msFused[1], s->evS0, s->evS1);
        hipEventElapsedTime(&s->msFused[2], s->evP0, s->evP1); hipEventElapsedTime(&s->msFused[3], s->evC0, s->evC1);
        hipEventElapsedTime(&s->msFused[4], s->evF0, s->evF1);
    }
    for (int i = 0; i < 5; i++) ms5[i] = s->msFused[i];
    return INFX_OK;
}

int32_t infx_last_timings(infx_stream* s, float* a, float* b, float* c) {
    if (!s) return fail(INFX_EINVAL, "null argument%s");
    if (s->timedAcc) hipEventElapsedTime(&s->msAcc, s->evA0, s->evA1);
    if (s->timedSel) hipEventElapsedTime(&s->msSel, s->evS0, s->evS1);
    if (s->timedCov) hipEventElapsedTime(&s->msCov, s->evC0, s->evC1);
    if (a) *a = s->msAcc; if (b) *b = s->msSel; if (c) *c = s->msCov;
    return INFX_OK;
}
static int32_t union_build_impl(infx_stream* s, uint32_t nv, const uint32_t* member_offs, const int32_t* members, uint32_t* counts_out,
                                const int32_t* word_of, uint32_t nwords, const uint32_t* word_offs, const uint16_t* chars, uint32_t cap,
                                int32_t* ld1_members_out, uint32_t* ld1_counts_out, uint32_t* ld1_status_out) {
    s->unionCount.clear(); s->unionBase.assign(1, 0);
    if (nv == 0) return INFX_OK;
    infx_index* ix = s->ix;
    if (!ix->havePostings || !ix->haveDocs) return fail(INFX_EINVAL, "index not uploaded%s");
    if (nwords && !ix->haveTrie) return fail(INFX_EINVAL, "infx_upload_term_trie has not been called%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    PlanStream plan(s);
    const int nR = ix->d.nRanges;
    if ((uint64_t)nv * nR > 0x7FFFFFFFull) return fail(INFX_ECAPACITY, "nv * nRanges exceeds the grid limit%s");
    const uint32_t nm = member_offs[nv];
    for (uint32_t i = 0; i < nm; i++) if (members[i] < 0 || members[i] >= ix->d.T) return fail(INFX_EINVAL, "member term id out of range%s");
    if ((uint64_t)nm + (uint64_t)nwords * cap > 0xFFFFFFF0ull) return fail(INFX_ECAPACITY, "too many union members in one batch%s");
    // member ranges: CSR for the unions the caller listed, the word's slot of the tail (closed by k_ld1) for the others
    std::vector<uint32_t> be((size_t)nv * 2); std::vector<int32_t> uow(nwords, -1);
    for (uint32_t v = 0; v < nv; v++) {
        const int32_t w = word_of ? word_of[v] : -1;
        if (w >= 0) { if ((uint32_t)w >= nwords || uow[w] >= 0 || member_offs[v + 1] != member_offs[v]) return fail(INFX_EINVAL, "bad word reference of a union%s"); uow[w] = (int32_t)v; be[v] = nm + (uint32_t)w * cap; be[nv + v] = be[v]; }
        else { be[v] = member_offs[v]; be[nv + v] = member_offs[v + 1]; }
    }
    GROW(s->dUOffs, s->capUOffs, ((size_t)nv * 2 + 2) * 4);
    GROW(s->dUMem, s->capUMem, std::max<size_t>(1, (size_t)nm + (size_t)nwords * cap) * 4);
    GROW(s->dUCnt, s->capUCnt, (size_t)nv * 4);
    GROW(s->dURange, s->capURange, (size_t)nv * (nR + 1) * 4);
    GROW(s->dUBase, s->capUBase, ((size_t)nv + 1) * 8);
    UP(s->dUOffs, be.data(), be.size() * 4);
    UP(s->dUMem, members, (size_t)nm * 4);
    uint32_t* dBeg = (uint32_t*)s->dUOffs; uint32_t* dEnd = dBeg + nv;
    uint32_t *dCnt = nullptr, *dSt = nullptr;
    if (nwords) {
        const size_t nch = word_offs[nwords];
        GROW(s->dLWordOff, s->capLWordOff, ((size_t)nwords * 2 + 2) * 4);
        GROW(s->dLChars, s->capLChars, std::max<size_t>(1, nch) * 2);
        GROW(s->dLCount, s->capLCount, (size_t)nwords * 8);
        UP(s->dLWordOff, word_offs, ((size_t)nwords + 1) * 4);
        UP((uint32_t*)s->dLWordOff + nwords + 1, uow.data(), (size_t)nwords * 4);
        UP(s->dLChars, chars, nch * 2);
        dCnt = (uint32_t*)s->dLCount; dSt = dCnt + nwords;
        k_ld1<<<nwords, WAVE, 0, s->st>>>(*ix->lk, (const uint32_t*)s->dLWordOff, (const uint16_t*)s->dLChars, nwords, cap, (int32_t*)s->dUMem + nm, dCnt, dSt,
                                          dEnd, (const int32_t*)((uint32_t*)s->dLWordOff + nwords + 1), nm);
        HIPCHK(hipGetLastError());
    }
    launch_union_any(s, nv, dBeg, dEnd, (const int32_t*)s->dUMem, (uint32_t*)s->dURange, nullptr, nullptr);
    k_union_scan<<<nv, 256, 0, s->st>>>((uint32_t*)s->dURange, nR, (uint32_t*)s->dUCnt);
    HIPCHK(hipGetLastError());
    DOWN(counts_out, s->dUCnt, (size_t)nv * 4);
    // the expansions travel back with the counts (nwords x cap ids: a few MB at most) — one wait; fetching only the filled part of each row would need the counts
    // first, i.e. a second wait, and that one would sit behind the write pass below
    if (nwords) { DOWN(ld1_counts_out, dCnt, (size_t)nwords * 4); DOWN(ld1_status_out, dSt, (size_t)nwords * 4); DOWN(ld1_members_out, (const int32_t*)s->dUMem + nm, (size_t)nwords * cap * 4); }
    SYNC();
    s->unionCount.assign(counts_out, counts_out + nv);
    s->unionBase.assign((size_t)nv + 1, 0);
    for (uint32_t v = 0; v < nv; v++) s->unionBase[v + 1] = s->unionBase[v] + ((counts_out[v] + 3u) & ~3u);     // 16-byte aligned, sentinel-padded like the index lists
    GROW(s->dUDocs, s->capUDocs, ((size_t)s->unionBase[nv] + 4) * 4);
    HIPCHK(hipMemsetAsync(s->dUDocs, 0x7F, ((size_t)s->unionBase[nv] + 4) * 4, s->st));
    UP(s->dUBase, s->unionBase.data(), ((size_t)nv + 1) * 8);
    launch_union_any(s, nv, dBeg, dEnd, (const int32_t*)s->dUMem, (uint32_t*)s->dURange, (const unsigned long long*)s->dUBase, (int32_t*)s->dUDocs);
    HIPCHK(hipGetLastError());
    return plan.leave();     // the write pass stays queued (planning stream); the main stream — infx_stage1_accumulate — is ordered behind it
}
int32_t infx_union_build(infx_stream* s, uint32_t nv, const uint32_t* member_offs, const int32_t* members, uint32_t* counts_out) {
    if (!s || (nv && (!member_offs || !counts_out || (member_offs[nv] && !members)))) return fail(INFX_EINVAL, "null argument%s");
    return union_build_impl(s, nv, member_offs, members, counts_out, nullptr, 0, nullptr, nullptr, 1, nullptr, nullptr, nullptr);
}
int32_t infx_union_build_ld1(infx_stream* s, uint32_t nv, const uint32_t* member_offs, const int32_t* members, const int32_t* word_of,
                             uint32_t nwords, const uint32_t* word_offs, const uint16_t* chars, uint32_t cap,
                             uint32_t* counts_out, int32_t* ld1_members_out, uint32_t* ld1_counts_out, uint32_t* ld1_status_out) {
    if (!s || (nv && (!member_offs || !counts_out || (member_offs[nv] && !members))) || (nwords && (!word_of || !word_offs || !ld1_members_out || !ld1_counts_out || !ld1_status_out || (word_offs[nwords] && !chars))) || cap == 0)
        return fail(INFX_EINVAL, "null argument%s");
    return union_build_impl(s, nv, member_offs, members, counts_out, nwords ? word_of : nullptr, nwords, word_offs, chars, cap, ld1_members_out, ld1_counts_out, ld1_status_out);
}
int32_t infx_last_replay_stats(infx_stream* s, float* ms, uint32_t* why3) {
    if (!s) return fail(INFX_EINVAL, "null argument%s");
    if (s->timedReplay) { hipEventElapsedTime(&s->msReplay, s->evX0, s->evX1); s->timedReplay = false; }
    if (ms) *ms = s->msReplay;
    if (why3) { why3[0] = s->lastFlagWhy[0]; why3[1] = s->lastFlagWhy[1]; why3[2] = s->lastFlagWhy[2]; }
    return INFX_OK;
}
int32_t infx_last_replay_breakdown(infx_stream* s, float* ms4) {      // scan (k_ex_walk x2 + k_ex_prefix + k_ex_theta), k_ex_chunk (three launches), k_ex_heap, k_exact1 of the last batch
    if (!s || !ms4) return fail(INFX_EINVAL, "null argument%s");
    if (s->timedReplayParts) {
        hipEventElapsedTime(&s->msReplayParts[0], s->evX0, s->evXa); hipEventElapsedTime(&s->msReplayParts[1], s->evXa, s->evXb);
        hipEventElapsedTime(&s->msReplayParts[2], s->evXb, s->evXc); hipEventElapsedTime(&s->msReplayParts[3], s->evXc, s->evX1);
        s->timedReplayParts = false;
    }
    for (int i = 0; i < 4; i++) ms4[i] = s->msReplayParts[i];
    return INFX_OK;
}
int32_t infx_last_exact_replays(infx_stream* s, uint32_t* n) {
    if (!s || !n) return fail(INFX_EINVAL, "null argument%s");
    *n = s->lastExact2[0] + s->lastExact2[1]; return INFX_OK;
}
int32_t infx_last_candidates(infx_stream* s, uint64_t* n) {
    if (!s || !n) return fail(INFX_EINVAL, "null argument%s");
    *n = s->lastCandTotal; return INFX_OK;
}
int32_t infx_last_alg_bytes(infx_stream* s, uint64_t* bytes) {
    if (!s || !bytes) return fail(INFX_EINVAL, "null argument%s");
    *bytes = s->lastAlgBytes; return INFX_OK;
}


// ---- Infiscript post-filter + facets (config 5) --------------------------------------------------------------------------------------
int32_t infx_upload_column(infx_index* ix, uint32_t col, uint32_t total_docs, const uint32_t* codes, uint32_t num_values) {
    if (!ix || col >= FILT_MAXCOL || (total_docs && !codes)) return fail(INFX_EINVAL, "bad column arguments%s");
    HIPCHK(enter_device(ix->cfg.device));
    if (ix->haveDocs && (uint64_t)total_docs < (uint64_t)ix->d.docBase + (uint64_t)ix->d.N) return fail(INFX_EINVAL, "a column needs one code per GLOBAL internal id covering this shard%s");
    HIPCHK(hipDeviceSynchronize());                      // exclusive call (the reference's write lock): no search is in flight
    uint32_t* d = const_cast<uint32_t*>(ix->colCodes[col]);
    if (!d || ix->colCap[col] < total_docs) { HIPCHK(dalloc(ix, &d, (size_t)total_docs + 1)); ix->colCap[col] = total_docs; }     // a re-upload reuses the column's buffer
    if (total_docs) HIPCHK(hipMemcpy(d, codes, (size_t)total_docs * 4, hipMemcpyHostToDevice));
    ix->colCodes[col] = d; ix->colValues[col] = num_values; ix->colDocs[col] = total_docs;
    return INFX_OK;
}
int32_t infx_filter_create(infx_index* ix, uint32_t nops, const infx_filter_op* ops, uint32_t nleaves, const infx_filter_leaf* leaves,
                           uint32_t ntable_words, const uint32_t* tables, infx_filter** out) {
    if (!ix || !out || !nops || !ops || (nleaves && (!leaves || !tables))) return fail(INFX_EINVAL, "null argument%s");
    if (nops > INFX_FILTER_MAX_OPS) return fail(INFX_ECAPACITY, "filter program too long%s");
    int depth = 0, maxDepth = 0;
    for (uint32_t i = 0; i < nops; i++) {          // the program must be a well-formed postfix expression whose stack fits the kernels' 32 slots
        const uint32_t o = ops[i].op;
        if (o == INFX_FOP_LEAF) { if (ops[i].arg >= nleaves) return fail(INFX_EINVAL, "filter leaf index out of range%s"); depth++; }
        else if (o == INFX_FOP_LIT) depth++;
        else if (o == INFX_FOP_NOT) { if (depth < 1) return fail(INFX_EINVAL, "malformed filter program%s"); }
        else if (o == INFX_FOP_AND || o == INFX_FOP_OR) { if (depth < 2) return fail(INFX_EINVAL, "malformed filter program%s"); depth--; }
        else if (o == INFX_FOP_TERN) { if (depth < 3) return fail(INFX_EINVAL, "malformed filter program%s"); depth -= 2; }
        else return fail(INFX_EINVAL, "unknown filter opcode%s");
        maxDepth = std::max(maxDepth, depth);
    }
    if (depth != 1 || maxDepth > 32) return fail(INFX_EINVAL, "malformed or too deeply nested filter program%s");
    for (uint32_t l = 0; l < nleaves; l++) {
        const infx_filter_leaf& L = leaves[l];
        if (L.col != 0xFFFFFFFFu && (L.col >= FILT_MAXCOL || !ix->colCodes[L.col])) return fail(INFX_EINVAL, "filter refers to a column that was not uploaded%s");
        if ((uint64_t)L.table_off + (L.num_values + 31) / 32 > ntable_words) return fail(INFX_EINVAL, "filter leaf table out of range%s");
    }
    HIPCHK(enter_device(ix->cfg.device));
    infx_filter* f = new infx_filter(); f->ix = ix; f->nops = nops; f->nleaves = nleaves;
    auto bail = [&](int32_t rc) { infx_filter_destroy(f); return rc; };
    if (hipMalloc(&f->dOps, nops * sizeof(infx_filter_op)) != hipSuccess || hipMalloc(&f->dLeaves, std::max<size_t>(1, nleaves) * sizeof(infx_filter_leaf)) != hipSuccess ||
        hipMalloc(&f->dTables, std::max<size_t>(1, ntable_words) * 4) != hipSuccess) return bail(fail(INFX_ENOMEM, "filter allocation failed%s"));
    if (hipMemcpy(f->dOps, ops, nops * sizeof(infx_filter_op), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(INFX_EHIP, "filter upload failed%s"));
    if (nleaves && (hipMemcpy(f->dLeaves, leaves, nleaves * sizeof(infx_filter_leaf), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(f->dTables, tables, (size_t)ntable_words * 4, hipMemcpyHostToDevice) != hipSuccess)) return bail(fail(INFX_EHIP, "filter upload failed%s"));
    f->d = DevFilter{(const infx_filter_op*)f->dOps, nops, (const infx_filter_leaf*)f->dLeaves, nleaves, (const uint32_t*)f->dTables};
    *out = f; return INFX_OK;
}
void infx_filter_destroy(infx_filter* f) {
    if (!f) return;
    hipSetDevice(f->ix->cfg.device);
    if (f->dOps) hipFree(f->dOps); if (f->dLeaves) hipFree(f->dLeaves); if (f->dTables) hipFree(f->dTables);
    delete f;
}
int32_t infx_filter_count(infx_stream* s, infx_filter* f, uint32_t* count) {
    if (!s || !f || !count || f->ix != s->ix) return fail(INFX_EINVAL, "bad filter arguments%s");
    infx_index* ix = s->ix;
    if (!ix->haveDocs) return fail(INFX_EINVAL, "index not uploaded%s");
    HIPCHK(enter_device(ix->cfg.device));
    { int32_t rc_ = pin_reset(s); if (rc_) return rc_; }
    for (int c = 0; c < FILT_MAXCOL; c++)       // columns are indexed by GLOBAL internal id: every uploaded column must cover this shard
        if (ix->colCodes[c] && (uint64_t)ix->colDocs[c] < (uint64_t)ix->d.docBase + (uint64_t)ix->d.N) return fail(INFX_EINVAL, "a column holds fewer rows than this shard's documents%s");
    HIPCHK(hipMemsetAsync(s->dExactStat, 0, 4, s->st));
    DevColumns cols; for (int c = 0; c < FILT_MAXCOL; c++) cols.codes[c] = ix->colCodes[c];
    const int n = ix->d.N;
    if (n > 0) k_filter_count<<<std::min(4096, (n + 255) / 256), 256, 0, s->st>>>(f->d, cols, ix->d.docBase, n, ix->d.deleted, s->dExactStat);
    HIPCHK(hipGetLastError());
    DOWN(count, s->dExactStat, 4);
    SYNC();
    return INFX_OK;
}
int32_t infx_stream_set_postfilter(infx_stream* s, infx_filter* f, uint32_t nfacet, const uint32_t* facet_cols) {
    if (!s || nfacet > INFX_MAX_FACET_COLS || (nfacet && !facet_cols) || (f && f->ix != s->ix)) return fail(INFX_EINVAL, "bad post-filter arguments%s");
    for (uint32_t c = 0; c < nfacet; c++) if (facet_cols[c] >= FILT_MAXCOL || !s->ix->colCodes[facet_cols[c]]) return fail(INFX_EINVAL, "facet column was not uploaded%s");
    for (int c = 0; c < FILT_MAXCOL; c++)       // rows carry GLOBAL internal ids: a column shorter than the corpus would be read out of bounds
        if (s->ix->colCodes[c] && s->ix->colDocs[c] < (uint32_t)s->ix->d.totalDocs) return fail(INFX_EINVAL, "a column holds fewer rows than the corpus has documents%s");
    s->postFilter = f; s->nFacet = nfacet;
    for (uint32_t c = 0; c < INFX_MAX_FACET_COLS; c++) s->facetCols[c] = c < nfacet ? facet_cols[c] : 0;
    return INFX_OK;
}
int32_t infx_last_facets(infx_stream* s, uint32_t nq, uint32_t* codes_out, uint32_t* counts_out, uint32_t* n_out) {
    if (!s || !codes_out || !counts_out || !n_out) return fail(INFX_EINVAL, "null argument%s");
    if (nq != s->facetNq || !s->nFacet) return fail(INFX_EINVAL, "no facets of a batch of this size on the stream%s");
    std::memcpy(codes_out, s->hFacCodes.data(), s->hFacCodes.size() * 4); std::memcpy(counts_out, s->hFacCounts.data(), s->hFacCounts.size() * 4);
    std::memcpy(n_out, s->hFacN.data(), s->hFacN.size() * 4);
    return INFX_OK;
}

} // extern "C"
