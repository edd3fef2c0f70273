// Host engine of the product: the C++ stand-in for the reference's C# SearchEngine (no .NET toolchain in the image).
// Same surface for the hot path — CreateDefault/CreateMinimal, IndexDocuments, Search(Query) — with the two accelerated
// seams delegated to the HIP kernels through the C ABI of include/infidex_hip.h. There is NO CPU scoring path here: without
// a GPU Search fails with INFX_EHIP.
//   SearchEngine.Search / IndexDocuments     src/Infidex/SearchEngine.cs:96-192, 256-319
//   SearchPipeline.Execute                   src/Infidex/Scoring/SearchPipeline.cs:49-206 (gates), :298-447 (coverage stage)
//   ResultProcessor.CalculateTruncationIndex src/Infidex/Scoring/ResultProcessor.cs:146-178
//   TopKHeap / ScoreEntry / ConsolidateSegments  Core/TopKHeap.cs, Core/ScoreEntry.cs:25-36, Scoring/SegmentProcessor.cs:15-37
#include "query.h"
#include "filter.h"
#include "infdx2.h"
#include "infdx2_verify.h"
#include "infs.h"
#include "hostcache.h"
#include <mutex>
#include <condition_variable>
#include "../../../include/infidex_engine.h"
#include <chrono>
#include <map>
#include <unordered_map>
#include <unordered_set>
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
struct FusedIn;
struct PerQ {
    std::vector<Entry> stage1;          // consolidated Stage-1 (score desc, key asc)
    std::vector<int32_t> stage1Doc;     // global internal ids
    bool runCov = false, wmAny = false, done = false, envelope = false;   // envelope: the query exceeds the Stage-2 query envelope -> answered as unsupported
    uint32_t candOff = 0, candCount = 0; int covIndex = -1;
    int32_t idx0 = -1, idx1 = -1;       // docs with docIndex 0 / 1
};
struct Batch {
    uint32_t nq = 0, nd = 0; int depth = 0, maxResults = 0;
    std::vector<infx_query> dq; std::vector<infx_term> dterms; std::vector<int32_t> extra; std::vector<uint32_t> qmap; std::vector<int> devOf;
    std::vector<infx_counts> counts;
    std::vector<PerQ> pq;
    std::vector<uint32_t> localIdx;     // local Stage-2 candidates -> position in lastCands
    std::vector<std::shared_ptr<FuzzyUnion>> pending; std::vector<uint32_t> pendingCounts; std::unordered_map<const FuzzyUnion*, uint32_t> unionIdx;   // unions whose df this batch counts
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, tPlanPar = 0, tTok = 0, tUnion = 0;
    double tLd1Dev = 0, tUnionDev = 0;      // of the planning time: spent inside infx_ld1_expand / infx_union_build (device work + the wait for it)
    std::shared_ptr<FusedIn> pre;      // sharded phases: per-query device-pipeline inputs prepared in phase 0 (off the collective path)
};

struct infx_session {
    std::vector<uint32_t> facetCols;      // engine column indices whose facets the session's stream counts
    infx_engine* e = nullptr;
    Batch* batch = nullptr;
    infx_stream* stream = nullptr;
    double tPrep1 = 0, tStage1 = 0, tPrep2 = 0, tStage2 = 0, tPost = 0;
    bool kernelTimesPending = false; float msReplayParts[4] = {0, 0, 0, 0};
    uint64_t collCalls[2] = {0, 0}, collBytes[2] = {0, 0};      // all-reduce / all-gather calls and payload bytes this session issued (sharded driver), cumulative
    float msAcc = 0, msSel = 0, msCov = 0, msPrep2 = 0, msFin = 0, msReplay = 0; uint32_t flagWhy[3] = {0, 0, 0}; uint64_t algBytes = 0; uint64_t s2Candidates = 0, s2TextBytes = 0, streamedBytes = 0, s1Candidates = 0; uint32_t exactReplays = 0;
    // last-batch introspection for parity tests
    std::vector<QueryPlan> lastPlans;
    std::vector<infx_hit> lastHits; std::vector<uint32_t> lastHitCount; int lastStride = 0;
    std::vector<infx_cov_cand> lastCands; std::vector<infx_cov_out> lastOuts; std::vector<int32_t> lastFeat;
    // sharded planning (infx_session_prefetch_*): WordMatcher descriptors of the coming batch computed by peer ranks, keyed by the search text;
    // this rank's own share, serialised for the exchange
    struct WmPre { std::vector<infx_wm_list> lists; std::vector<int32_t> owned; };
    std::unordered_map<std::u16string, WmPre> wmPre;
    std::vector<uint8_t> prefetchBlob;
    // plan exchange: per query of the COMING batch, the token-level plan (plan_tokens_text) and the coverage query context (prepare_cov_query) computed by
    // the rank whose slice holds the query — this rank's own slice included; consumed by the next phase 0 (ph_plan, build_fused_inputs)
    // Own slice: parsed (plan, cq filled in by the collect).  A peer's: `rec` points at the query's record inside the retained blob (PlanSlab) and is parsed where
    // the plan is needed — on the planner threads of phase 0, straight into the batch's plan / coverage-query arrays — so the import itself costs a checksum.
    struct OwnPlan { QueryPlan plan; infx_cov_query cq; std::unique_ptr<infx_cov_query_long> cql; };
    struct PlanPre { uint64_t rawHash = 0; int32_t depth = 0, covErr = 0; bool hasCov = false, longCov = false, planTaken = false;      /* planTaken: own->plan was moved into a batch */ const uint8_t* rec = nullptr; uint32_t recLen = 0, covOff = 0; std::unique_ptr<OwnPlan> own; };
    struct PlanSlab { std::vector<uint8_t> bytes; std::vector<PlanPre> pre; };
    std::vector<std::shared_ptr<PlanPre>> planPre;
    uint32_t planFromExchange = 0, planFromPeers = 0;      // last phase 0: queries planned from planPre / of them imported from another rank
    std::vector<uint8_t> planPeer;                          // planPre[i] came from a peer
};

struct CompiledFilter { infx_filter* dev = nullptr; uint32_t inFilter = 0; bool counted = false; };
// Order of the collectives of a rank's pipeline sessions.  Every session has a communicator and a HIP stream of its own, and its host thread enqueues the
// collectives of its batch when it gets there — left alone, the relative order of DIFFERENT communicators' kernels on a device depends on thread timing and
// differs from rank to rank, the classic multi-communicator hang (two collective kernels each holding the CUs the other's peer needs).  The ring makes the
// order a function of the batch schedule alone: the sessions of a stream take turns, one collective per turn, in ring order; a session whose last batch is
// done leaves the ring (infx_session_coll_retire).  Batch i runs on session i mod K on every rank and a batch issues the same collectives everywhere, so every
// rank enqueues every communicator's kernels in the same global order.
struct CollSeq {
    std::mutex m; std::condition_variable cv;
    std::vector<infx_session*> ring; std::vector<uint8_t> active; size_t cur = 0;
    int index_of(const infx_session* S) const { for (size_t i = 0; i < ring.size(); i++) if (ring[i] == S) return (int)i; return -1; }
    void advance_from(size_t i) {                       // next active member after i (stays on i when it is the only one)
        for (size_t k = 1; k <= ring.size(); k++) { const size_t j = (i + k) % ring.size(); if (active[j]) { cur = j; return; } }
        cur = i;
    }
    void set_ring(infx_session* const* ss, uint32_t n) { std::lock_guard<std::mutex> lk(m); ring.assign(ss, ss + n); active.assign(n, 1); cur = 0; cv.notify_all(); }
    bool enter(const infx_session* S) {                 // blocks until it is S's turn; false: S is not (or no longer) in the ring -> unordered
        std::unique_lock<std::mutex> lk(m);
        const int i = index_of(S);
        if (i < 0 || !active[i]) return false;
        cv.wait(lk, [&] { return cur == (size_t)i || !active[i]; });
        return active[i] != 0;
    }
    void leave(const infx_session* S) { std::lock_guard<std::mutex> lk(m); const int i = index_of(S); if (i >= 0 && cur == (size_t)i) { advance_from((size_t)i); cv.notify_all(); } }
    void retire(const infx_session* S) {
        std::lock_guard<std::mutex> lk(m); const int i = index_of(S); if (i < 0 || !active[i]) return;
        active[i] = 0; if (cur == (size_t)i) advance_from((size_t)i); cv.notify_all();
    }
};
// Host planning of concurrent sessions goes through a FIFO gate of `limit` planners at a time.  Each planner fans out over the whole worker pool; with
// six sessions planning at once (the start of a stream, or any moment their phases line up) every one of them takes six times as long and the GPU
// waits for all of them — gated, the first batches reach the device after one planning time and the pipeline fills in order.
struct PlanGate {
    std::mutex m; std::condition_variable cv; int inUse = 0, limit = 0; uint64_t next = 0, serving = 0;
    void enter() { if (limit <= 0) return; std::unique_lock<std::mutex> lk(m); const uint64_t t = next++; cv.wait(lk, [&] { return t == serving && inUse < limit; }); serving++; inUse++; cv.notify_all(); }
    void leave() { if (limit <= 0) return; { std::lock_guard<std::mutex> lk(m); inUse--; } cv.notify_all(); }
};
static thread_local PlanGate* tl_gate = nullptr;      // the gate this thread holds, if any
struct PlanGateHold { PlanGate& g; explicit PlanGateHold(PlanGate& x) : g(x) { g.enter(); tl_gate = &g; } ~PlanGateHold() { tl_gate = nullptr; g.leave(); } };
// A planner that waits for the device (LD1 expansion, union cardinalities) gives its place at the gate to the next planner and queues again afterwards: the
// gate rations CPU time, and a thread blocked on a HIP event uses none (held across the wait it capped the pipeline at `limit` / wait batches per second).
struct PlanGatePause { PlanGate* g; PlanGatePause() : g(tl_gate) { if (g) g->leave(); } ~PlanGatePause() { if (g) g->enter(); } };
struct infx_engine {
    // non-indexed document fields (DocumentFields) as dictionary-encoded columns + compiled Infiscript filters (config 5)
    std::vector<filt::Column> columns; std::mutex filterMu; std::unordered_map<std::string, CompiledFilter> filters;
    std::vector<infx_filter*> retiredFilters;     // compiled against an older column set (a session's stream may still point at one): freed with the engine
    // NumberOfDocumentsInFilter is cached per expression; a Filter parsed after a mutation counts again (the reference keeps the count on the Filter instance)
    void invalidate_filter_counts() { std::lock_guard<std::mutex> lk(filterMu); for (auto& kv : filters) kv.second.counted = false; }
    void retire_filters() { std::lock_guard<std::mutex> lk(filterMu); for (auto& kv : filters) if (kv.second.dev) retiredFilters.push_back(kv.second.dev); filters.clear(); }
    HostIndex ix;
    PlanGate gate;
    CollSeq collSeq;
    infx_engine_config cfg{};
    infx_index* dev = nullptr;
    bool indexed = false;
    FuzzyCache fuzzy;
    std::unordered_map<int64_t, int32_t> keyToFirst;
    bool keysAreIds = false;
    std::vector<uint8_t> deleted;     // Document.Deleted per global internal id; empty = nothing deleted
    std::atomic<long long> ld1OnHost{0}, ld1OnDevice{0}, wmOnHost{0}, wmOnDevice{0};      // where the planning lookups ran (introspection)
    bool devLookups = false;          // WordMatcher dictionaries + term trie uploaded: the LD1 / WordMatcher lookups of planning can run on the GPU (f3)
    // Both sides can answer a batch's lookups with identical results (tests/test_gpu_lookups.py).
    // WordMatcher: always the device once the dictionaries are uploaded — k_wm rides inside the batch's pipeline (0.12 ms, no extra wait), while the host's cost
    // depends on the sizes of the affix ranges (config 3: 37 ms of 16 threads per batch; a word-count model sent it to the host and halved the rate).
    // LD1 expansion: by cost.  The host walk is predictable (~21 us of one core per unknown word, spread over the planner threads); k_ld1 is one latency-bound wave
    // per word whose result the host WAITS for, and behind five other batches' streaming kernels that wait is ~15 ms (0.2 ms on an idle GPU): with cores to spare
    // the host is faster for a few hundred words (config 4, ~220 unknown words per batch, 10 M documents, 16 CPUs: 74.8 k against 68.7 k queries/s); the device takes
    // the batch when the estimated host time exceeds a tenth of the planner threads' time in a 10-ms batch interval — two threads per rank on an 8-GPU node, or about
    // a thousand unknown words: config 3 (every query misspelt, ~960 words per batch) runs 68.0 k against 65.7 k queries/s with the expansion on the device (fused into
    // the union build: one wait) at four sessions, 43.5 k against 39.7 k at one (round 5, profiles/r05_bench_cfg3_device_ld1_*.json; the threshold was 2.5 x before and
    // sent config 3 to the host).  INFX_DEVICE_LOOKUPS=1 pins the device (the GPU test suite does), INFX_HOST_LOOKUPS=1 keeps everything on the host (no upload).
    bool ld1Pinned = false;
    bool ld1_on_device(double estThreadMs) const { return devLookups && (ld1Pinned || estThreadMs > 1.0 * (double)std::max(1, threads)); }
    bool lookups_on_device(double) const { return devLookups; }
    int threads = 1;
    int buildThreads = 0;             // > 0: threads of the index build only (infx_engine_set_build_threads)
    infx_session* def = nullptr;      // default session (single-caller API)
    // document sharding: this engine's GPU holds internal ids [shardBase, shardBase + shardN) of the corpus
    int rank = 0, nranks = 1; int32_t shardBase = 0, shardN = 0;
    std::vector<uint64_t> shardOff;   // sharded: per-term slice lengths prefix (T+1) of the uploaded CSR
    uint64_t shardTermLen(int32_t t) const { return nranks > 1 ? shardOff[t + 1] - shardOff[t] : ix.terms.len((uint32_t)t); }
};

// byte-blob writer / reader of the sharded-planning exchange (infx_session_prefetch_*)
namespace {
// what a peer's descriptors index into: sizes of the host index (FNV-1a over the counts; every rank builds the same index from the same corpus)
uint64_t index_fingerprint(const HostIndex& ix) {
    const uint64_t v[6] = {(uint64_t)ix.N, (uint64_t)ix.terms.K(), (uint64_t)ix.terms.doc.size(), (uint64_t)ix.wmExact.doc.size(), (uint64_t)ix.wmLd1.doc.size(), (uint64_t)ix.text.size()};
    uint64_t h = 1469598103934665603ull;
    for (uint64_t x : v) for (int b = 0; b < 8; b++) { h ^= (x >> (8 * b)) & 0xFFu; h *= 1099511628211ull; }
    return h;
}
struct BlobW { std::vector<uint8_t>& b; template <class T> void put(const T& v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof(T)); }
               void bytes(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); } };
uint64_t raw_hash(const uint16_t* p, size_t n) {      // FNV-1a over the raw query text: ties an exchanged plan to the query it was made for
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
uint64_t bytes_hash(const uint8_t* p, size_t n) {      // checksum of a blob section: eight bytes per step
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    for (; i < n; i++) { h = (h ^ p[i]) * 1099511628211ull; }
    return h ^ (h >> 29);
}
struct BlobR { const uint8_t* p; const uint8_t* e; bool ok = true;
               template <class T> T get() { T v{}; if ((size_t)(e - p) < sizeof(T)) { ok = false; return v; } std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
               const uint8_t* take(size_t n) { if ((size_t)(e - p) < n) { ok = false; return nullptr; } const uint8_t* r = p; p += n; return r; } };
// CoverageEngine.PrepareQuery into the fast record, or — beyond its envelope — into a long record (then `lq` is set and q is left zeroed: its `reserved` member
// gets 1 + the record's index in the batch's long-query table when that is assembled).  INFX_EUNSUPPORTED: beyond the long envelope too.
int32_t prepare_cov_any(const HostIndex& ix, uview st, infx_cov_query& q, std::unique_ptr<infx_cov_query_long>& lq) {
    lq.reset();
    int32_t rc = prepare_cov_query(ix, st, q);
    if (rc != INFX_EUNSUPPORTED) return rc;
    lq.reset(new infx_cov_query_long);
    rc = prepare_cov_query_long(ix, st, *lq);
    if (rc) lq.reset(); else std::memset(&q, 0, sizeof q);
    return rc;
}
// One query's record of the plan exchange (layout: infx_session_prefetch_collect).  parse_plan_record fills the token-level plan and reports where the
// coverage query sits; parse_cov_record rebuilds the infx_cov_query exactly as prepare_cov_query wrote it (its text is the plan's searchText).
// Everything that reaches the device is range-checked.  nullptr = fine, else what is wrong.
const char* parse_plan_record(const HostIndex& ix, const uint8_t* rec, uint32_t len, int depth, QueryPlan& P, bool& hasCov, int32_t& covErr, uint32_t& covOff, bool& longCov) {
    BlobR R{rec, rec + len};
    auto str = [&](ustr& t) { const uint32_t l = R.get<uint32_t>(); const uint8_t* p = R.take((size_t)l * 2); if (!R.ok) return; t.resize(l); if (l) std::memcpy(&t[0], p, (size_t)l * 2); };
    P = QueryPlan(); P.depth = depth;
    R.get<uint64_t>(); const uint8_t fl = R.get<uint8_t>();
    P.blank = (fl & 1) != 0; P.unsupported = (fl & 2) != 0; hasCov = (fl & 16) != 0; covErr = 0; longCov = (fl & 64) != 0;
    str(P.qtext); if (fl & 4) P.searchText = P.qtext; else str(P.searchText); if (fl & 8) P.tfidfQuery = P.searchText; else str(P.tfidfQuery);
    const uint16_t nraw = R.get<uint16_t>();
    if (!R.ok) return "record truncated";
    if (nraw > 128) return "more than 128 raw tokens";
    const int64_t nTerms = (int64_t)ix.terms.K();
    P.rawTok.resize(nraw);
    for (uint16_t k = 0; k < nraw && R.ok; k++) {
        auto& r = P.rawTok[k]; r.id = R.get<int32_t>();
        if (r.id < 0) { r.id = -1; str(r.text); } else if ((int64_t)r.id >= nTerms) return "term id out of range (ranks must hold the same index)";
    }
    if (!R.ok) return "record truncated";
    covOff = (uint32_t)(R.p - rec);
    if (hasCov && (fl & 32)) { covErr = R.get<int32_t>(); if (!R.ok || covErr == 0) return "inconsistent coverage-query status"; }
    return nullptr;
}
template <class QT, int MAXCHARS, int MAXTOK>
const char* parse_cov_record_t(const ustr& searchText, const uint8_t* rec, uint32_t len, uint32_t covOff, QT& C) {
    if (covOff > len) return "record truncated";
    BlobR R{rec + covOff, rec + len};
    std::memset(&C, 0, sizeof C);
    const size_t tl = searchText.size();
    if (tl > (size_t)MAXCHARS) return "coverage query longer than the envelope";
    std::memcpy(C.text, searchText.data(), tl * 2); C.text_len = (int32_t)tl;
    C.num_tokens = R.get<int32_t>();
    if (!R.ok || C.num_tokens < 0 || C.num_tokens > MAXTOK) return "coverage query token count out of range";
    const size_t nt = (size_t)C.num_tokens; const uint8_t* p;
    if ((p = R.take(nt * 2))) std::memcpy(C.tok_off, p, nt * 2);
    if ((p = R.take(nt * 2))) std::memcpy(C.tok_len, p, nt * 2);
    if ((p = R.take(nt * 4))) std::memcpy(C.term_idf, p, nt * 4);
    if ((p = R.take(nt * 4))) std::memcpy(C.word_idf, p, nt * 4);
    C.has_word_idf = R.get<int32_t>(); C.num_fusion_tokens = R.get<int32_t>();
    if (!R.ok || C.num_fusion_tokens < 0 || C.num_fusion_tokens > 2 * MAXTOK) return "coverage query token count out of range";
    const size_t nf = (size_t)C.num_fusion_tokens;
    if ((p = R.take(nf * 2))) std::memcpy(C.ftok_off, p, nf * 2);
    if ((p = R.take(nf * 2))) std::memcpy(C.ftok_len, p, nf * 2);
    C.lcs_tolerance = R.get<int32_t>();
    if (!R.ok) return "record truncated";
    for (size_t k = 0; k < nt; k++) if ((size_t)C.tok_off[k] + C.tok_len[k] > tl) return "coverage token outside its text";      // the token tables go to the device
    for (size_t k = 0; k < nf; k++) if ((size_t)C.ftok_off[k] + C.ftok_len[k] > tl) return "coverage token outside its text";
    return nullptr;
}
const char* parse_cov_record(const ustr& st, const uint8_t* rec, uint32_t len, uint32_t covOff, infx_cov_query& C) { return parse_cov_record_t<infx_cov_query, INFX_MAX_QUERY_CHARS, INFX_MAX_QUERY_TOKENS>(st, rec, len, covOff, C); }
const char* parse_cov_record_long(const ustr& st, const uint8_t* rec, uint32_t len, uint32_t covOff, infx_cov_query_long& C) { return parse_cov_record_t<infx_cov_query_long, INFX_LONGQ_CHARS, INFX_LONGQ_TOKENS>(st, rec, len, covOff, C); }
}

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
    e->threads = cfg->threads > 0 ? cfg->threads : effective_cpus();
    { const char* g = getenv("INFX_PLAN_GATE"); e->gate.limit = g ? atoi(g) : std::max(1, e->threads / 8); }      // 0 = no gate
    if (cfg->device >= 0) {
        infx_config dc{}; dc.device = cfg->device; dc.range_docs = cfg->range_docs; dc.max_depth = h.maxDepth; dc.flags = cfg->no_exact_replay ? INFX_CFG_NO_EXACT_REPLAY : 0;
        int32_t rc = infx_create(&dc, &e->dev);
        if (rc) { g_eerr = infx_last_error(); delete e; return rc; }
    }
    e->def = new infx_session(); e->def->e = e; e->def->batch = new Batch();
    *out = e; return INFX_OK;
}

void infx_engine_destroy(infx_engine* e) {
    if (!e) return;
    infx_session* S = e->def;
    if (S && S->stream) infx_stream_destroy(S->stream);
    if (S) delete S->batch;
    delete S;
    for (auto& kv : e->filters) if (kv.second.dev) infx_filter_destroy(kv.second.dev);
    for (auto* f : e->retiredFilters) infx_filter_destroy(f);
    if (e->dev) infx_destroy(e->dev);
    delete e;
}

static int32_t finish_index(infx_engine* e);
// SearchEngine.IndexDocuments: n documents x field_count fields (UTF-16 arena + offsets), keys may be null (key = index)
int32_t infx_engine_index_documents(infx_engine* e, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs,
                                    int32_t field_count, const int32_t* field_weights) {
    if (!e || n < 0 || (n && (!arena || !offs)) || field_count < 1 || !field_weights) return efail(INFX_EINVAL, "bad arguments");
    if (e->indexed) return efail(INFX_EINVAL, "this engine instance is already indexed (re-indexing: create a new engine)");
    if (n > 0x7FFFFFF0ll) return efail(INFX_EINVAL, "too many documents");
    for (int64_t i = 0, m = n * field_count; i < m; i++) if (offs[i + 1] < offs[i]) return efail(INFX_EINVAL, "field offsets must be ascending (offs[n * field_count] = the arena's length)");
    DocSource src{n, field_count, field_weights, keys, (const u16*)arena, offs};
    {
        const int planThreads = e->ix.cfg.threads;
        if (e->buildThreads > 0) e->ix.cfg.threads = e->buildThreads;      // a node's leader rank builds with every core while the other ranks wait for its cache
        build_index(src, e->ix);
        e->ix.cfg.threads = planThreads;
    }
    e->keysAreIds = (keys == nullptr);
    return finish_index(e);
}
// An engine populated from FLUSHED SEGMENTS + a live tail (VectorModel.Flush, VectorModel.cs:804-815): segment i holds the postings of documents
// [doc_bases[i], doc_bases[i] + its document count), the segments cover the documents from 0 without gaps, the documents behind them are the live tail.  The
// posting lists of the flushed ranges are taken from the files (read and validated by host/infs.h) — only the tail's postings are accumulated from the texts;
// term ids, document lengths, the WordMatcher dictionaries, prefix sets and the Stage-2 texts need the documents, so all n documents are supplied.  The engine
// then searches the whole corpus as ONE index.  (The reference answers a mixed index segment by segment, every segment with its own tier decisions and its own
// top-k heap, merged afterwards — VectorModel.cs:572-584: an index-lifecycle artefact this does not imitate; an unflushed index of the same documents is what
// it equals, and what tests/test_infs.py checks against the oracle.)
int32_t infx_engine_index_from_segments(infx_engine* e, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs, int32_t field_count, const int32_t* field_weights,
                                        int32_t n_segments, const char* const* paths, const int32_t* doc_bases) {
    if (!e || n < 0 || (n && (!arena || !offs)) || field_count < 1 || !field_weights || n_segments < 0 || (n_segments && (!paths || !doc_bases))) return efail(INFX_EINVAL, "bad arguments");
    if (e->indexed) return efail(INFX_EINVAL, "this engine instance is already indexed (re-indexing: create a new engine)");
    if (n > 0x7FFFFFF0ll) return efail(INFX_EINVAL, "too many documents");
    for (int64_t i = 0, m = n * field_count; i < m; i++) if (offs[i + 1] < offs[i]) return efail(INFX_EINVAL, "field offsets must be ascending (offs[n * field_count] = the arena's length)");
    std::vector<infs::Segment> files((size_t)n_segments); std::vector<SegmentPostings> segs((size_t)n_segments);
    for (int32_t i = 0; i < n_segments; i++) {
        if (!paths[i]) return efail(INFX_EINVAL, "null segment path");
        if (!infs::read_file(paths[i], files[i])) return efail(INFX_EINVAL, "INFS segment: " + files[i].error);
        segs[i].docBase = doc_bases[i]; segs[i].docCount = files[i].docCount; segs[i].terms = &files[i].terms; segs[i].off = &files[i].off; segs[i].doc = &files[i].doc; segs[i].w = &files[i].w;
    }
    DocSource src{n, field_count, field_weights, keys, (const u16*)arena, offs};
    const int planThreads = e->ix.cfg.threads;
    if (e->buildThreads > 0) e->ix.cfg.threads = e->buildThreads;
    const char* err = build_index(src, e->ix, &segs);
    e->ix.cfg.threads = planThreads;
    if (err) { const HostConfig cfg = e->ix.cfg; e->ix = HostIndex(); e->ix.cfg = cfg; return efail(INFX_EUNSUPPORTED, std::string("index from segments: ") + err); }      // the engine stays unindexed and reusable
    e->keysAreIds = (keys == nullptr);
    return finish_index(e);
}


// The dictionaries of the planning lookups go to the device next to the lists they resolve to (include/infidex_hip.h "Dictionary lookups"): WordMatcher
// exact / symmetric-delete keys, the word list with its affix orders, the reversed-term trie.  INFX_HOST_LOOKUPS=1 keeps the lookups on the host.
static void key_arena(const KeyTable& K, std::vector<uint32_t>& offs, std::vector<u16>& repack, const u16*& chars) {
    const size_t n = K.size();
    offs.resize(n + 1);
    bool contiguous = true; uint32_t run = n ? K.keyOff[0] : 0;
    for (size_t i = 0; i < n; i++) { if (K.keyOff[i] != run) { contiguous = false; break; } run += K.keyLen[i]; }
    if (contiguous && (n == 0 || K.keyOff[0] == 0)) { for (size_t i = 0; i < n; i++) offs[i] = K.keyOff[i]; offs[n] = run; chars = K.arena.data(); return; }
    repack.clear(); for (size_t i = 0; i < n; i++) { offs[i] = (uint32_t)repack.size(); repack.insert(repack.end(), K.arena.begin() + K.keyOff[i], K.arena.begin() + K.keyOff[i] + K.keyLen[i]); }
    offs[n] = (uint32_t)repack.size(); chars = repack.data();
}
static int32_t upload_lookups(infx_engine* e) {
    const HostIndex& ix = e->ix;
    { const char* h = getenv("INFX_HOST_LOOKUPS"); if (h && h[0] == '1') return INFX_OK; }
    if (ix.rEdgeStart.size() < 2) return INFX_OK;
    int32_t rc = infx_upload_term_trie(e->dev, (uint32_t)ix.rTerm.size(), ix.rEdgeStart.data(), (const uint16_t*)ix.rEdgeLabel.data(), ix.rEdgeChild.data(), ix.rTerm.data(),
                                       (uint32_t)ix.sortedTerms.size(), ix.sortedTerms.data());
    if (rc) return rc;
    if (ix.cfg.wordMatcher) {
        std::vector<uint32_t> eo, lo, wo; std::vector<u16> er, lr, wr; const u16 *ec = nullptr, *lc = nullptr, *wc = nullptr;
        key_arena(ix.wmExact.keys, eo, er, ec); key_arena(ix.wmLd1.keys, lo, lr, lc); key_arena(ix.words, wo, wr, wc);
        rc = infx_upload_wm_dictionary(e->dev, (uint32_t)ix.wmExact.K(), eo.data(), (const uint16_t*)ec, ix.wmExact.off.data(),
                                       (uint32_t)ix.wmLd1.K(), lo.data(), (const uint16_t*)lc, ix.wmLd1.off.data(),
                                       (uint32_t)ix.words.size(), wo.data(), (const uint16_t*)wc, ix.wordLastDoc.data(),
                                       (uint32_t)ix.affixFwd.size(), ix.affixFwd.data(), ix.affixRev.data(), ix.cfg.wmMinLD1, ix.cfg.wmMaxLD1);
        if (rc) return rc;
    }
    e->devLookups = true;
    { const char* d = getenv("INFX_DEVICE_LOOKUPS"); e->ld1Pinned = d && d[0] == '1'; }
    return INFX_OK;
}

// After the host index exists (built here or read from a node-local cache): key map, shard bounds, upload of this rank's slice
static int32_t finish_index(infx_engine* e) {
    if (!e->keysAreIds) { const int64_t n = e->ix.N; e->keyToFirst.reserve((size_t)n * 2); for (int64_t d = 0; d < n; d++) e->keyToFirst.emplace(e->ix.docKey[d], (int32_t)d); }
    {
        // contiguous doc-range shards of whole 65 536-id Roaring containers: the reference scores its candidates in chunks that never span a
        // container (Bm25Scorer.cs:195-280), so with boundaries at container multiples its sequential walk is the shards' walks one after the other and
        // the exact Stage-1 replay stays shard-local (exactsh.hip.inc).  Rank r owns containers [nCont*r/W, nCont*(r+1)/W) — an empty shard if W > nCont.
        const int64_t N = e->ix.N, W = e->nranks, r = e->rank, C = 65536, nCont = (N + C - 1) / C;
        auto bound = [&](int64_t k) { return std::min<int64_t>(N, C * (nCont * k / W)); };
        e->shardBase = (int32_t)bound(r); e->shardN = (int32_t)(bound(r + 1) - bound(r));
    }
    if (e->dev) {
        HostIndex& ix = e->ix;
        int32_t rc;
        if (e->nranks == 1) {
            rc = infx_upload_docs(e->dev, (uint32_t)ix.N, ix.docLen.data(), ix.avgdl, ix.docKey.data(), nullptr, ix.textOff.data(), (const uint16_t*)ix.text.data());
            if (!rc) rc = infx_upload_postings(e->dev, (uint32_t)ix.terms.K(), ix.terms.off.data(), ix.terms.doc.data(), ix.terms.w.data(), ix.df.data());
            if (!rc) rc = infx_upload_prefix_docsets(e->dev, (uint32_t)(ix.psOff.size() - 1), ix.psOff.data(), ix.psDocs.data());
            if (!rc && ix.cfg.wordMatcher) rc = infx_upload_wordmatcher(e->dev, ix.wmExact.doc.size(), ix.wmExact.doc.data(), ix.wmLd1.doc.size(), ix.wmLd1.doc.data());
        } else {
            // global statistics (df, avgdl, N, prefix populations, word IDF) stay on the host; the GPU gets this shard's slices, rebased
            const int32_t sb = e->shardBase, sn = e->shardN, se = sb + sn;
            std::vector<uint64_t> to((size_t)sn + 1);
            for (int32_t d = 0; d <= sn; d++) to[d] = ix.textOff[sb + d] - ix.textOff[sb];
            rc = infx_upload_docs(e->dev, (uint32_t)sn, ix.docLen.data() + sb, ix.avgdl, ix.docKey.data() + sb, nullptr, to.data(), (const uint16_t*)ix.text.data() + ix.textOff[sb]);
            size_t T = ix.terms.K();
            std::vector<uint64_t> lo(T), hi(T); e->shardOff.assign(T + 1, 0);
            parallel_for((int64_t)T, e->threads, [&](int64_t b, int64_t en, int) {
                for (int64_t t = b; t < en; t++) {
                    const int32_t* p = ix.terms.doc.data(); uint64_t a = ix.terms.off[t], z = ix.terms.off[t + 1];
                    lo[t] = std::lower_bound(p + a, p + z, sb) - p; hi[t] = std::lower_bound(p + lo[t], p + z, se) - p;
                }
            });
            for (size_t t = 0; t < T; t++) e->shardOff[t + 1] = e->shardOff[t] + (hi[t] - lo[t]);
            std::vector<int32_t> sd(e->shardOff[T]); std::vector<uint8_t> sw(e->shardOff[T]);
            parallel_for((int64_t)T, e->threads, [&](int64_t b, int64_t en, int) {
                for (int64_t t = b; t < en; t++) { uint64_t o = e->shardOff[t]; for (uint64_t i = lo[t]; i < hi[t]; i++, o++) { sd[o] = ix.terms.doc[i] - sb; sw[o] = ix.terms.w[i]; } }
            });
            if (!rc) rc = infx_upload_postings(e->dev, (uint32_t)T, e->shardOff.data(), sd.data(), sw.data(), ix.df.data());
            size_t ns = ix.psOff.size() - 1;
            std::vector<uint64_t> po(ns + 1, 0); std::vector<int32_t> pd;
            for (size_t k = 0; k < ns; k++) {
                auto a = std::lower_bound(ix.psDocs.begin() + ix.psOff[k], ix.psDocs.begin() + ix.psOff[k + 1], sb);
                auto z = std::lower_bound(a, ix.psDocs.begin() + ix.psOff[k + 1], se);
                for (auto it = a; it != z; ++it) pd.push_back(*it - sb);
                po[k + 1] = pd.size();
            }
            if (!rc) rc = infx_upload_prefix_docsets(e->dev, (uint32_t)ns, po.data(), pd.data());
            if (!rc) rc = infx_set_shard(e->dev, e->rank, e->nranks, sb, ix.N);
            if (!rc && ix.cfg.wordMatcher) rc = infx_upload_wordmatcher(e->dev, ix.wmExact.doc.size(), ix.wmExact.doc.data(), ix.wmLd1.doc.size(), ix.wmLd1.doc.data());
            if (!rc) rc = infx_upload_doc_keys_all(e->dev, (uint32_t)ix.N, ix.docKey.data());
        }
        if (!rc) rc = upload_lookups(e);
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
// ------------------------------------------------------------------------------------------------------------------------
// One batch = four phases.  Unsharded: they run back to back (search_batch_impl).  Document-sharded: the caller
// (infidex_amd/sharded.py) interleaves the collectives — all-reduce of the class histograms after phase 1, all-gather of the
// per-shard top-`depth` after phase 2, all-reduce of the disjoint Stage-2 records after phase 3.

// FstIndex.MatchWithinEditDistance1 for the distinct unknown words of a batch that the expansion cache does not hold: one infx_ld1_expand call; the few
// words the kernel hands back (work lists outgrown, longer than 64 characters) are expanded by the host walk.  Results enter the LRU cache as before.
// Words of the batch that wait for the device expansion fused into the union build (infx_union_build_ld1): placeholders until the members are back
struct DevLd1 { std::vector<ustr> words; std::vector<std::shared_ptr<FuzzyUnion>> fz; };
static int32_t expand_pending(infx_engine* e, infx_session* S, std::vector<QueryPlan>& plans, DevLd1* dev) {
    const HostIndex& ix = e->ix;
    std::vector<const ustr*> words; std::unordered_map<std::u16string, uint32_t> idx;
    for (auto& P : plans) for (auto& r : P.rawTok) if (r.pending && idx.emplace(r.text, (uint32_t)words.size()).second) words.push_back(&r.text);
    if (words.empty()) return INFX_OK;
    const uint32_t nw = (uint32_t)words.size(), cap = 1024;
    static const bool fusedOff = [] { const char* v = getenv("INFX_LD1_FUSED"); return v && v[0] == '0'; }();
    if (dev && !fusedOff && e->ld1_on_device(0.021 * (double)nw)) {
        // the device expands these words inside the union build (one wait for the device instead of two): every occurrence gets the word's placeholder union
        dev->words.resize(nw); dev->fz.resize(nw);
        for (uint32_t i = 0; i < nw; i++) { dev->words[i] = *words[i]; dev->fz[i] = std::make_shared<FuzzyUnion>(); }
        for (auto& P : plans) for (auto& r : P.rawTok) if (r.pending) { r.fz = dev->fz[idx.at(r.text)]; r.pending = false; }
        return INFX_OK;
    }
    std::vector<std::shared_ptr<FuzzyUnion>> made(nw);
    std::vector<uint32_t> offs(nw + 1, 0), counts(nw, 0), status(nw, 2); std::vector<u16> chars; std::vector<int32_t> members((size_t)nw * cap);
    for (uint32_t i = 0; i < nw; i++) { if (words[i]->size() <= 64) chars.insert(chars.end(), words[i]->begin(), words[i]->end()); offs[i + 1] = (uint32_t)chars.size(); }     // longer words: empty -> status 2
    auto t0 = std::chrono::steady_clock::now();
    if (e->ld1_on_device(0.021 * (double)nw)) {
        int32_t rc;
        { PlanGatePause pause; rc = infx_ld1_expand(S->stream, nw, offs.data(), (const uint16_t*)chars.data(), cap, members.data(), counts.data(), status.data()); }
        if (rc) { g_eerr = infx_last_error(); return rc; }
        S->batch->tLd1Dev = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }      // else: status stays 2 for every word -> the host walk below, spread over the planner threads
    std::vector<uint32_t> onHost;
    for (uint32_t i = 0; i < nw; i++) {
        if (status[i] == 0) made[i] = union_of_matches(ix, members.data() + (size_t)i * cap, std::min(counts[i], cap));
        else onHost.push_back(i);
    }
    if (!onHost.empty())
        parallel_dyn((int64_t)onHost.size(), e->threads, 1, [&](int64_t b, int64_t en, int) {
            std::vector<int> m;
            for (int64_t k = b; k < en; k++) { const uint32_t i = onHost[k]; match_ld1(ix, *words[i], m, (int)cap); made[i] = union_of_matches(ix, m.data(), m.size()); }
        });
    e->fuzzy.ld1Ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    e->fuzzy.fuzzyCalls += nw; e->ld1OnHost += (long long)onHost.size(); e->ld1OnDevice += (long long)(nw - onHost.size());
    for (uint32_t i = 0; i < nw; i++) made[i] = e->fuzzy.put(*words[i], made[i]);
    for (auto& P : plans) for (auto& r : P.rawTok) if (r.pending) { r.fz = made[idx.at(r.text)]; r.pending = false; }
    return INFX_OK;
}
static int32_t ph_plan(infx_engine* e, infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth) {
    if (!e || (nq && (!q_arena || !q_offs))) return efail(INFX_EINVAL, "bad arguments");
    if (!e->dev || !S || !S->stream) return efail(INFX_EHIP, "no GPU: the scoring hot path has no CPU fallback");
    if (depth <= 0 || depth > e->ix.cfg.maxDepth) return efail(INFX_EINVAL, "CoverageDepth exceeds the engine's max_depth");
    const HostIndex& ix = e->ix; const int threads = e->threads;
    Batch& B = *S->batch; B = Batch(); B.nq = nq; B.depth = depth;
    g_eerr.clear();
    B.t0 = now_ms();
    std::vector<QueryPlan>& plans = S->lastPlans; plans.assign(nq, QueryPlan());
    const bool devLd1 = e->devLookups;      // unknown words are collected per batch; who expands them is decided once their number is known
    // plan exchange (document shards): the token-level plan of a query may have been made by the rank whose slice holds it (infx_session_prefetch_*);
    // it is used when it was made for exactly this text and depth, and only the cache / expansion pass runs here
    const bool havePre = S->planPre.size() == nq;
    std::atomic<uint32_t> nPre{0}, nPeer{0}; std::atomic<const char*> badRec{nullptr};
    parallel_dyn(nq, threads, devLd1 ? 8 : 1, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; i++) {
            const u16* rp = (const u16*)q_arena + q_offs[i]; const size_t rl = (size_t)(q_offs[i + 1] - q_offs[i]);
            infx_session::PlanPre* pre = havePre ? S->planPre[i].get() : nullptr;
            if (pre && !pre->planTaken && pre->depth == depth && pre->rawHash == raw_hash((const uint16_t*)rp, rl)) {
                // (the entry keeps its coverage query — parsed, or where it sits in the record — for build_fused_inputs)
                if (pre->rec) { const char* why = parse_plan_record(ix, pre->rec, pre->recLen, depth, plans[i], pre->hasCov, pre->covErr, pre->covOff, pre->longCov); if (why) badRec.store(why); }
                else { plans[i] = std::move(pre->own->plan); pre->planTaken = true; }      // a second phase 0 on the same entries (INFX_PHASED, an empty batch: build_fused_inputs did not clear them) plans afresh instead of taking the moved-from plan
                nPre++; if (S->planPeer[i]) nPeer++;
            } else {
                if (havePre) S->planPre[i] = nullptr;      // not this query's: build_fused_inputs must not take its coverage query either
                plan_tokens_text(ix, uview(rp, rl), depth, plans[i]);
            }
            plan_tokens_expand(ix, e->fuzzy, plans[i], false, devLd1);
        }
    });
    S->planFromExchange = nPre.load(); S->planFromPeers = nPeer.load();
    if (badRec.load()) { S->planPre.clear(); S->planPeer.clear(); return efail(INFX_EINVAL, std::string("exchanged plan record: ") + badRec.load()); }
    DevLd1 dl;
    if (devLd1) { int32_t rc = expand_pending(e, S, plans, &dl); if (rc) return rc; }
    B.tTok = now_ms() - B.t0;
    {   // every fuzzy union this batch uses is materialised on the device (this shard's slice); |union| = its df (sharded:
        // summed over the shards by the caller)
        // One union per distinct misspelt WORD of the batch, in first-occurrence order: the list must be the same on every rank (the counts are
        // all-reduced position by position), so it may not depend on the state of the expansion cache — two queries of a batch can hold two
        // FuzzyUnion objects for one word when the LRU cache evicted it in between (the objects have the same members: LD1 is a function of the word).
        B.pending.clear(); B.unionIdx.clear();
        std::unordered_map<std::u16string, uint32_t> byWord;
        for (auto& P : plans) for (auto& r : P.rawTok) if (r.fz && !r.fz->materialised) {
            auto it = byWord.find(r.text);
            uint32_t v;
            if (it == byWord.end()) { v = (uint32_t)B.pending.size(); byWord.emplace(r.text, v); B.pending.push_back(r.fz); } else v = it->second;
            B.unionIdx.emplace(r.fz.get(), v);
        }
        B.pendingCounts.assign(B.pending.size(), 0);
        std::vector<uint32_t> mo(B.pending.size() + 1, 0); std::vector<int32_t> mm;
        for (size_t v = 0; v < B.pending.size(); v++) { mm.insert(mm.end(), B.pending[v]->members.begin(), B.pending[v]->members.end()); mo[v + 1] = (uint32_t)mm.size(); }
        const double tu0 = now_ms();
        int32_t rc;
        if (dl.words.empty()) { PlanGatePause pause; rc = infx_union_build(S->stream, (uint32_t)B.pending.size(), mo.data(), mm.data(), B.pendingCounts.data()); }
        else {
            // k_ld1 for the placeholders' words, k_union for every union of the batch, one wait (infx_union_build_ld1)
            const uint32_t nw = (uint32_t)dl.words.size(), cap = 1024;
            std::unordered_map<const FuzzyUnion*, int32_t> wordOfFz; for (uint32_t w = 0; w < nw; w++) wordOfFz.emplace(dl.fz[w].get(), (int32_t)w);
            std::vector<int32_t> wordOf(B.pending.size(), -1);
            for (size_t v = 0; v < B.pending.size(); v++) { auto it = wordOfFz.find(B.pending[v].get()); if (it != wordOfFz.end()) wordOf[v] = it->second; }
            std::vector<uint32_t> wo(nw + 1, 0), lc(nw, 0), ls(nw, 2); std::vector<u16> wc; std::vector<int32_t> lm((size_t)nw * cap);
            for (uint32_t w = 0; w < nw; w++) { if (dl.words[w].size() <= 64) wc.insert(wc.end(), dl.words[w].begin(), dl.words[w].end()); wo[w + 1] = (uint32_t)wc.size(); }      // longer words: empty -> status 2
            { PlanGatePause pause; rc = infx_union_build_ld1(S->stream, (uint32_t)B.pending.size(), mo.data(), mm.data(), wordOf.data(), nw, wo.data(), (const uint16_t*)wc.data(), cap,
                                                            B.pendingCounts.data(), lm.data(), lc.data(), ls.data()); }
            if (rc) { g_eerr = infx_last_error(); return rc; }
            B.tLd1Dev = now_ms() - tu0;
            std::vector<uint32_t> onHost;
            for (uint32_t w = 0; w < nw; w++) {
                if (ls[w] == 0) { auto u = union_of_matches(ix, lm.data() + (size_t)w * cap, std::min(lc[w], cap)); dl.fz[w]->members.swap(u->members); }
                else onHost.push_back(w);
            }
            e->ld1OnDevice += (long long)(nw - onHost.size()); e->ld1OnHost += (long long)onHost.size(); e->fuzzy.fuzzyCalls += nw;
            if (!onHost.empty()) {
                // the few words the kernel handed back: host walk, then the unions once more with every member list known
                parallel_dyn((int64_t)onHost.size(), threads, 1, [&](int64_t b, int64_t en, int) {
                    std::vector<int> m;
                    for (int64_t k = b; k < en; k++) { const uint32_t w = onHost[k]; match_ld1(ix, dl.words[w], m, (int)cap); auto u = union_of_matches(ix, m.data(), m.size()); dl.fz[w]->members.swap(u->members); }
                });
                mm.clear();
                for (size_t v = 0; v < B.pending.size(); v++) { mm.insert(mm.end(), B.pending[v]->members.begin(), B.pending[v]->members.end()); mo[v + 1] = (uint32_t)mm.size(); }
                { PlanGatePause pause; rc = infx_union_build(S->stream, (uint32_t)B.pending.size(), mo.data(), mm.data(), B.pendingCounts.data()); }
                if (rc) { g_eerr = infx_last_error(); return rc; }
            }
            for (uint32_t w = 0; w < nw; w++) e->fuzzy.put(dl.words[w], dl.fz[w]);      // into the expansion cache (LRU 1000, as the reference's)
        }
        if (rc) { g_eerr = infx_last_error(); return rc; }
        B.tUnionDev = now_ms() - tu0;
    }
    B.tUnion = now_ms() - B.t0;
    return INFX_OK;
}

// second half of planning: global df of the pending unions known
static int32_t ph_plan_finish(infx_engine* e, infx_session* S, const uint32_t* globalUnionCounts) {
    const HostIndex& ix = e->ix; const int threads = e->threads;
    Batch& B = *S->batch; const uint32_t nq = B.nq;
    std::vector<QueryPlan>& plans = S->lastPlans;
    for (size_t v = 0; v < B.pending.size(); v++) B.pending[v]->df.store((int)globalUnionCounts[v]);
    for (auto& kv : B.unionIdx) const_cast<FuzzyUnion*>(kv.first)->df.store((int)globalUnionCounts[kv.second]);      // every object of the word, not only the first
    parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; i++) plan_finish(ix, plans[i]); });
    B.tPlanPar = now_ms() - B.t0;
    const int32_t sb = e->shardBase, se = e->shardBase + e->shardN;
    const bool sharded = e->nranks > 1;
    auto shard_slice = [&](const std::vector<int32_t>& d, size_t& lo, size_t& hi) {
        if (!sharded) { lo = 0; hi = d.size(); return; }
        lo = std::lower_bound(d.begin(), d.end(), sb) - d.begin(); hi = std::lower_bound(d.begin(), d.end(), se) - d.begin();
    };
    size_t nterm = 0, nextra = 0;
    std::vector<size_t> termBase(nq), extraBase(nq);
    B.devOf.assign(nq, -1);
    for (uint32_t i = 0; i < nq; i++) {
        QueryPlan& P = plans[i];
        termBase[i] = nterm; extraBase[i] = nextra;
        if (P.blank || P.unsupported || P.noTerms) continue;
        P.q.term_off = (uint32_t)nterm; nterm += P.terms.size();
        for (size_t k = 0; k < P.terms.size(); k++) if (P.terms[k].term_id < 0) {
            if (P.terms[k].reserved == 1) continue;
            else { size_t lo, hi; shard_slice(P.fuzzy[k]->docs, lo, hi); nextra += hi - lo; }
        }
        B.devOf[i] = (int)B.dq.size(); B.dq.push_back(P.q); B.qmap.push_back(i);
    }
    if (nextra > 0xFFFFFFF0ull) return efail(INFX_ECAPACITY, "fuzzy unions of this batch exceed 2^32 postings; split the batch");
    B.dterms.resize(nterm); B.extra.resize(nextra);
    parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; i++) {
            QueryPlan& P = plans[i];
            if (P.blank || P.unsupported || P.noTerms) continue;
            size_t xo = extraBase[i];
            for (size_t k = 0; k < P.terms.size(); k++) {
                infx_term t = P.terms[k];
                if (t.term_id < 0 && t.reserved == 1) {     // union materialised on the device by ph_plan
                    const uint32_t v = B.unionIdx.at(P.fuzzy[k].get());
                    t.reserved = 2; t.extra_off = v; t.extra_len = B.pendingCounts[v];
                } else if (t.term_id < 0) {     // host-built union: global ids -> this shard's slice, rebased (its df / idf stay global)
                    const auto& d = P.fuzzy[k]->docs; size_t lo, hi; shard_slice(d, lo, hi);
                    t.extra_off = (uint32_t)xo; t.extra_len = (uint32_t)(hi - lo);
                    for (size_t z = lo; z < hi; z++) B.extra[xo + (z - lo)] = d[z] - sb;
                    xo += hi - lo;
                }
                B.dterms[termBase[i] + k] = t;
            }
        }
    });
    B.nd = (uint32_t)B.dq.size();
    B.t1 = now_ms();
    return INFX_OK;
}

static int32_t ph_accumulate(infx_engine* e, infx_session* S) {
    Batch& B = *S->batch;
    B.counts.assign(B.nd, infx_counts{});
    S->msAcc = S->msSel = S->msCov = S->msPrep2 = S->msFin = 0; S->algBytes = 0;
    if (B.nd) {
        int32_t rc = infx_stage1_accumulate(S->stream, B.nd, B.dq.data(), (uint32_t)B.dterms.size(), B.dterms.data(), (uint32_t)B.extra.size(), B.extra.data(), B.counts.data());
        if (rc) { g_eerr = infx_last_error(); return rc; }
    }
    return INFX_OK;
}

static int32_t ph_select(infx_engine* e, infx_session* S, const infx_counts* globalCounts) {
    Batch& B = *S->batch; const HostIndex& ix = e->ix;
    std::vector<infx_hit>& hits = S->lastHits; std::vector<uint32_t>& hitCount = S->lastHitCount;
    hits.assign((size_t)B.nd * B.depth, infx_hit{0, 0.f}); hitCount.assign(B.nd, 0); S->lastStride = B.depth;
    if (B.nd) {
        int32_t rc = infx_stage1_select(S->stream, B.nd, globalCounts, hits.data(), hitCount.data());
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_timings(S->stream, &S->msAcc, &S->msSel, nullptr);
        infx_last_alg_bytes(S->stream, &S->streamedBytes);
        infx_last_candidates(S->stream, &S->s1Candidates); infx_last_exact_replays(S->stream, &S->exactReplays);
        // B_alg(q) = sum_t df_t * 5 B (4 B for fuzzy virtual terms) + card(C_q) * 4 B + depth * 12 B   (this shard's slices)
        uint64_t ab = 0;
        for (auto& t : B.dterms) {
            if (t.term_id >= 0) ab += (uint64_t)e->shardTermLen(t.term_id) * 5ull;
            else ab += (uint64_t)t.extra_len * 4ull;
        }
        uint64_t nh = 0; for (uint32_t c : hitCount) nh += c;
        S->algBytes = ab + S->s1Candidates * 4ull + nh * 12ull;
    }
    (void)ix;
    B.t2 = now_ms();
    return INFX_OK;
}

// allHits: W x nd x depth (global internal ids), allCounts: W x nd.  Runs the local part of Stage 2.
static int32_t ph_stage2(infx_engine* e, infx_session* S, int W, const infx_hit* allHits, const uint32_t* allCounts, int32_t max_results, int32_t enable_coverage) {
    Batch& B = *S->batch; const HostIndex& ix = e->ix; const int threads = e->threads;
    const uint32_t nq = B.nq, nd = B.nd; const int depth = B.depth;
    B.maxResults = max_results;
    std::vector<QueryPlan>& plans = S->lastPlans;
    B.pq.assign(nq, PerQ());
    std::vector<std::vector<infx_cov_cand>> candLocal(nq);
    std::vector<infx_cov_query> covQ(nq); std::vector<std::unique_ptr<infx_cov_query_long>> covL(nq);      // covL: queries beyond the fast Stage-2 envelope
    std::vector<int32_t> covErr(nq, 0);
    const bool covEnabled = ix.cfg.enableCoverage && enable_coverage;
    static const bool dbg = getenv("INFX_DEBUG") != nullptr;
    std::atomic<long long> nsMerge{0}, nsWm{0}, nsSel{0}, nsCovQ{0}, nsPush{0}, nsSelMax{0};
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t) { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); };
    parallel_dyn(nq, threads, 4, [&](int64_t b, int64_t en, int) {
        WmResult wm; std::vector<int32_t> sortedTop, overlap, uniq; std::vector<infx_hit> merged; std::vector<uint8_t> hitFlag; std::vector<uint32_t> sbase;
        for (int64_t i = b; i < en; i++) {
            QueryPlan& P = plans[i]; PerQ& Sq = B.pq[i];
            auto tA = tick();
            if (P.blank || P.unsupported) { Sq.done = true; continue; }
            int j = B.devOf[i];
            if (j >= 0) {
                // global top-`depth` = best `depth` of the union of the per-shard top-`depth` lists, device order (score desc, id asc)
                merged.clear();
                for (int w = 0; w < W; w++) { uint32_t c = allCounts[(size_t)w * nd + j]; const infx_hit* H = allHits + ((size_t)w * nd + j) * depth; merged.insert(merged.end(), H, H + c); }
                if (W > 1) {
                    std::sort(merged.begin(), merged.end(), [](const infx_hit& x, const infx_hit& y) { if (x.score != y.score) return x.score > y.score; return x.doc < y.doc; });
                    if ((int)merged.size() > depth) merged.resize(depth);
                }
                uint32_t c = (uint32_t)merged.size();
                Sq.stage1.resize(c); Sq.stage1Doc.resize(c);
                // TopKHeap -> ConsolidateSegments: (score desc, key asc)
                std::sort(merged.begin(), merged.end(), [&](const infx_hit& x, const infx_hit& y) { if (x.score != y.score) return x.score > y.score; return ix.docKey[x.doc] < ix.docKey[y.doc]; });
                for (uint32_t k = 0; k < c; k++) { Sq.stage1[k] = Entry{merged[k].score, ix.docKey[merged[k].doc], 0}; Sq.stage1Doc[k] = merged[k].doc; }
            }
            const ustr& st = P.searchText;
            bool isShort = !st.empty() && st.size() <= 3;
            if (isShort) for (u16 ch : st) if (is_delim(ch)) { isShort = false; break; }
            if (isShort && (int)Sq.stage1.size() >= max_results) { Sq.done = true; continue; }     // SearchPipeline.cs:114-120
            int shortCount = 0;
            if (isShort) { int64_t pk = ix.prefixKeys.find(st); shortCount = pk >= 0 ? (int)ix.prefixPop[pk] : 0; }
            bool skipCov = isShort && shortCount > 500;
            if (!covEnabled || skipCov) { Sq.done = true; continue; }
            Sq.runCov = true;
            if (dbg) nsMerge += since(tA);
            auto tB = tick();
            // ---- ExecuteCoverageStage preparation ----
            wm_collect(ix, st, true, wm);
            Sq.wmAny = wm.any;
            if (dbg) nsWm += since(tB);
            auto tC = tick();
            size_t ntop = std::min<size_t>(Sq.stage1.size(), (size_t)depth);
            sortedTop.assign(Sq.stage1Doc.begin(), Sq.stage1Doc.begin() + ntop);
            std::sort(sortedTop.begin(), sortedTop.end());
            overlap.clear();
            if (wm.any) { wm_contains_batch(wm, sortedTop.data(), (int)sortedTop.size(), hitFlag, sbase); for (size_t z = 0; z < sortedTop.size(); z++) if (hitFlag[z]) overlap.push_back(sortedTop[z]); }   // ascending
            size_t wmLimit = (size_t)std::max(0, depth - (int)overlap.size());
            size_t need = std::max<size_t>(wmLimit, 2);
            const uint8_t* del = e->deleted.empty() ? nullptr : e->deleted.data();     // Document.Deleted per global internal id
            int32_t live[2] = {-1, -1};
            wm_first_unique(wm, sortedTop, need, uniq, del, (int)std::max<int64_t>(0, 2 - (int64_t)ntop), live);
            // docIndex 0/1 = first two keys in insertion order: Stage-1 docs, then the (live) WordMatcher-only ids ascending
            int32_t first2[2] = {-1, -1}; int nf = 0;
            for (size_t k = 0; k < ntop && nf < 2; k++) first2[nf++] = Sq.stage1Doc[k];
            if (!del) { for (size_t k = 0; k < uniq.size() && nf < 2; k++) first2[nf++] = uniq[k]; }
            else for (int k = 0; k < 2 && nf < 2; k++) if (live[k] >= 0) first2[nf++] = live[k];
            Sq.idx0 = first2[0]; Sq.idx1 = first2[1];
            if (dbg) { long long d = since(tC); nsSel += d; long long cur = nsSelMax.load(); while (d > cur && !nsSelMax.compare_exchange_weak(cur, d)) {} }
            auto tD = tick();
            covErr[i] = prepare_cov_any(ix, st, covQ[i], covL[i]);
            if (dbg) nsCovQ += since(tD);
            if (covErr[i]) continue;
            auto tE = tick();
            auto& CL = candLocal[i];
            auto push = [&](int32_t doc, float base) { infx_cov_cand c{}; c.query = 0; c.doc = doc; c.base_score = base; c.want_lcs = (doc == first2[0] || doc == first2[1]) ? 1 : 0; CL.push_back(c); };
            for (int32_t d : overlap) push(d, 0.f);
            for (size_t k = 0; k < uniq.size() && k < wmLimit; k++) if (!(del && del[uniq[k]])) push(uniq[k], 0.f);     // a deleted id still uses up its wmLimit slot
            float maxT = ntop ? Sq.stage1[0].score : 1.f;
            for (size_t k = 0; k < ntop; k++) {
                push(Sq.stage1Doc[k], maxT > 0 ? Sq.stage1[k].score / maxT : 0.f);
                // a document already evaluated as an overlap row: its LCS is read back from the byte span (infx_cov_cand.want_lcs = 2)
                if (CL.back().want_lcs && std::binary_search(overlap.begin(), overlap.end(), Sq.stage1Doc[k])) CL.back().want_lcs = 2;
            }
            if (dbg) nsPush += since(tE);
        }
    });
    if (dbg) fprintf(stderr, "[infx] prep2 cpu-ms: merge %.1f wm %.1f select %.1f (max %.2f) covq %.1f push %.1f | wall %.1f\n", nsMerge / 1e6, nsWm / 1e6, nsSel / 1e6, nsSelMax / 1e6, nsCovQ / 1e6, nsPush / 1e6, now_ms() - B.t2);
    // a query outside the Stage-2 envelope is answered as "unsupported" (empty result, flag bit 0); the rest of the batch is unaffected
    for (uint32_t i = 0; i < nq; i++) if (covErr[i]) { B.pq[i].runCov = false; B.pq[i].stage1.clear(); B.pq[i].stage1Doc.clear(); B.pq[i].envelope = true; candLocal[i].clear(); }
    std::vector<infx_cov_query> covBatch; std::vector<infx_cov_query_long> covBatchL; std::vector<infx_cov_cand>& cands = S->lastCands;
    size_t ncand = 0;
    for (uint32_t i = 0; i < nq; i++) {
        PerQ& Sq = B.pq[i];
        if (!Sq.runCov) continue;
        Sq.covIndex = (int)covBatch.size(); covBatch.push_back(covQ[i]);
        if (covL[i]) { covBatch.back().reserved = 1 + (int32_t)covBatchL.size(); covBatchL.push_back(*covL[i]); }
        Sq.candOff = (uint32_t)ncand; Sq.candCount = (uint32_t)candLocal[i].size(); ncand += candLocal[i].size();
    }
    cands.resize(ncand);
    const int32_t sb = e->shardBase, se = e->shardBase + e->shardN;
    const bool sharded = e->nranks > 1;
    std::vector<uint64_t> tbytes(nq, 0);
    parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; i++) {
            PerQ& Sq = B.pq[i]; if (!Sq.runCov) continue;
            infx_cov_cand* dst = cands.data() + Sq.candOff; uint64_t tb = 0;
            for (size_t k = 0; k < candLocal[i].size(); k++) {
                infx_cov_cand c = candLocal[i][k]; c.query = (uint32_t)Sq.covIndex; dst[k] = c;
                if (c.doc >= sb && c.doc < se) tb += 2 * (ix.textOff[c.doc + 1] - ix.textOff[c.doc]);
            }
            tbytes[i] = tb;
        }
    });
    S->s2TextBytes = 0; for (uint64_t x : tbytes) S->s2TextBytes += x;
    B.t3 = now_ms();
    // ---------------- Stage 2 on the GPU: the candidates whose text this shard holds ----------------
    std::vector<infx_cov_out>& outs = S->lastOuts; outs.resize(cands.size());
    const bool wantF = e->cfg.want_features != 0;
    if (wantF) S->lastFeat.assign(cands.size() * INFX_NFEAT, 0);
    B.localIdx.clear();
    if (!sharded) {     // every candidate is local: score in place
        S->s2Candidates = cands.size();
        if (!cands.empty()) {
            int32_t rc = covBatchL.empty() ? INFX_OK : infx_stage2_long_queries(S->stream, (uint32_t)covBatchL.size(), covBatchL.data());
            if (!rc) rc = infx_stage2_batch(S->stream, (uint32_t)covBatch.size(), covBatch.data(), (uint32_t)cands.size(), cands.data(), outs.data(), wantF ? S->lastFeat.data() : nullptr);
            if (rc) { g_eerr = infx_last_error(); return rc; }
            infx_last_timings(S->stream, nullptr, nullptr, &S->msCov);
        }
    } else {
        std::fill(outs.begin(), outs.end(), infx_cov_out{});
        std::vector<infx_cov_cand> local;
        for (size_t i = 0; i < cands.size(); i++) {
            const infx_cov_cand& c = cands[i];
            if (c.doc < sb || c.doc >= se) continue;
            infx_cov_cand l = c; l.doc = c.doc - sb; local.push_back(l); B.localIdx.push_back((uint32_t)i);
        }
        S->s2Candidates = local.size();
        std::vector<infx_cov_out> lout(local.size());
        std::vector<int32_t> lfeat;
        if (wantF) lfeat.assign(local.size() * INFX_NFEAT, 0);
        if (!local.empty()) {
            int32_t rc = covBatchL.empty() ? INFX_OK : infx_stage2_long_queries(S->stream, (uint32_t)covBatchL.size(), covBatchL.data());
            if (!rc) rc = infx_stage2_batch(S->stream, (uint32_t)covBatch.size(), covBatch.data(), (uint32_t)local.size(), local.data(), lout.data(), wantF ? lfeat.data() : nullptr);
            if (rc) { g_eerr = infx_last_error(); return rc; }
            infx_last_timings(S->stream, nullptr, nullptr, &S->msCov);
            for (size_t i = 0; i < local.size(); i++) {
                outs[B.localIdx[i]] = lout[i];
                if (wantF) std::memcpy(S->lastFeat.data() + (size_t)B.localIdx[i] * INFX_NFEAT, lfeat.data() + i * INFX_NFEAT, INFX_NFEAT * 4);
            }
        }
    }
    {   std::atomic<int> bad{0};
        parallel_for((int64_t)outs.size(), threads, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; i++) if (outs[i].status) { bad.store(1); break; } });
        (void)bad;     // rows with a non-zero status are skipped per query in ph_finalize (result flag bit 3)
    }
    B.t4 = now_ms();
    return INFX_OK;
}

static int32_t ph_finalize(infx_engine* e, infx_session* S, const infx_cov_out* outs, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                           uint32_t* out_counts, uint32_t* out_flags) {
    Batch& B = *S->batch; const HostIndex& ix = e->ix; const int threads = e->threads;
    const uint32_t nq = B.nq; const int depth = B.depth, max_results = B.maxResults;
    std::vector<QueryPlan>& plans = S->lastPlans; std::vector<infx_cov_cand>& cands = S->lastCands;
    parallel_dyn(nq, threads, 8, [&](int64_t b, int64_t en, int) {
        std::vector<Entry> fin, cons;
        for (int64_t i = b; i < en; i++) {
            PerQ& Sq = B.pq[i]; const QueryPlan& P = plans[i];
            uint32_t flags = 0; const std::vector<Entry>* res = &Sq.stage1;
            if (P.unsupported || Sq.envelope) flags |= 1;
            if (Sq.runCov) {
                flags |= 2;
                int maxWordHits = 0; uint8_t hits01[2] = {0, 0}, lcs01[2] = {0, 0};
                fin.clear();
                for (uint32_t k = 0; k < Sq.candCount; k++) {
                    const infx_cov_cand& c = cands[Sq.candOff + k]; const infx_cov_out& o = outs[Sq.candOff + k];
                    if (o.status) { flags |= 8; continue; }          // outside the Stage-2 envelope: left out of this query's ranking
                    maxWordHits = std::max(maxWordHits, o.word_hits_full);
                    for (int z = 0; z < 2; z++) { int32_t dz = z == 0 ? Sq.idx0 : Sq.idx1; if (dz >= 0 && c.doc == dz) { if (hits01[z] == 0) hits01[z] = o.word_hits; if (lcs01[z] == 0) lcs01[z] = o.lcs; } }
                    fin.push_back(Entry{o.score, ix.docKey[c.doc], o.tiebreaker});
                }
                if (maxWordHits == 0 && !Sq.wmAny) { cons.clear(); }
                else {
                    // TopKHeap(depth): best `depth` by the total order, then ConsolidateSegments (best per key, descending)
                    std::sort(fin.begin(), fin.end(), [](const Entry& a, const Entry& c) { return cmp_entry(a, c) > 0; });
                    if ((int)fin.size() > depth) fin.resize(depth);
                    cons.clear();
                    std::unordered_map<int64_t, char> seen; seen.reserve(fin.size() * 2);
                    for (auto& x : fin) if (seen.emplace(x.key, 1).second) cons.push_back(x);
                    int truncIdx = -1;
                    int minHits = std::max(1, maxWordHits - 0);
                    int64_t k0 = Sq.idx0 >= 0 ? ix.docKey[Sq.idx0] : INT64_MIN, k1 = Sq.idx1 >= 0 ? ix.docKey[Sq.idx1] : INT64_MIN;
                    for (int r = (int)cons.size() - 1; r >= 0; r--) {
                        uint8_t wh = 0, lc = 0;
                        if (Sq.idx0 >= 0 && cons[r].key == k0) { wh = hits01[0]; lc = lcs01[0]; }
                        else if (Sq.idx1 >= 0 && cons[r].key == k1) { wh = hits01[1]; lc = lcs01[1]; }
                        if (wh >= minHits || lc > 0 || cons[r].score >= 254.f) { truncIdx = r; break; }
                    }
                    int resultCount = truncIdx == -1 ? max_results : std::min(std::max(0, truncIdx) + 1, max_results);
                    if ((int)cons.size() > resultCount) cons.resize(resultCount);
                }
                if (cons.empty() && !Sq.stage1.empty()) { flags |= 4; res = &Sq.stage1; } else res = &cons;
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
        fprintf(stderr, "[infx] nq=%u dev=%u terms=%zu extra=%zu cands=%zu | plan %.1f (tokens %.1f unions[%zu] %.1f finish %.1f) s1 %.1f prep2 %.1f s2 %.1f post %.1f ms | fuzzy calls=%lld %.1f ms-cpu (ld1 %.1f) docs=%lld\n",
                nq, B.nd, B.dterms.size(), B.extra.size(), cands.size(), B.t1 - B.t0, B.tTok, B.pending.size(), B.tUnion - B.tTok, B.tPlanPar - B.tUnion, B.t2 - B.t1, B.t3 - B.t2, B.t4 - B.t3, t5 - B.t4,
                (long long)e->fuzzy.fuzzyCalls.exchange(0), e->fuzzy.fuzzyNs.exchange(0) / 1e6, e->fuzzy.ld1Ns.exchange(0) / 1e6, (long long)e->fuzzy.fuzzyDocs.exchange(0));
    }
    S->tPrep1 = B.t1 - B.t0; S->tStage1 = B.t2 - B.t1; S->tPrep2 = B.t3 - B.t2; S->tStage2 = B.t4 - B.t3; S->tPost = t5 - B.t4;
    return INFX_OK;
}

// Per-query inputs of the device pipeline that need the host dictionaries: flags, WordMatcher list descriptors (WordMatcher.Lookup,
// WordMatcher.cs:95-186; affix hits are copied into `owned`), CoverageEngine.PrepareQuery.
// The WordMatcher lists of one search text as device descriptors: (src 0 exact / 1 symmetric-delete dictionary, offset, length) into the uploaded
// doc-id arrays, affix hits (src 2) copied into `owned`.  A pure function of (index, text): document shards compute it for a slice of the batch each
// and exchange the results (infx_session_prefetch_*).
static void wm_descriptors(const HostIndex& ix, const ustr& st, WmResult& wm, std::vector<infx_wm_list>& lists, std::vector<int32_t>& owned) {
    const int32_t* exB = ix.wmExact.doc.data(); const int32_t* exE = exB + ix.wmExact.doc.size();
    const int32_t* l1B = ix.wmLd1.doc.data(); const int32_t* l1E = l1B + ix.wmLd1.doc.size();
    lists.clear(); owned.clear();
    wm_collect(ix, st, true, wm);
    for (auto& l : wm.lists) {
        if (!l.n) continue;
        infx_wm_list L{}; L.len = (uint32_t)l.n;
        if (l.p >= exB && l.p < exE) { L.src = 0; L.off = (uint64_t)(l.p - exB); }
        else if (l.p >= l1B && l.p < l1E) { L.src = 1; L.off = (uint64_t)(l.p - l1B); }
        else { L.src = 2; L.off = owned.size(); owned.insert(owned.end(), l.p, l.p + l.n); }
        lists.push_back(L);
    }
    if (lists.size() > INFX_MAX_WM_LISTS) {
        // very long queries: the device probes at most INFX_MAX_WM_LISTS lists per query, so the lists are merged here into ONE
        // ascending list (membership in any list / first-unique over the union are unchanged by the merge)
        std::vector<int32_t> all;
        for (auto& l : wm.lists) all.insert(all.end(), l.p, l.p + l.n);
        std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
        owned.swap(all);
        infx_wm_list L{}; L.src = 2; L.off = 0; L.len = (uint32_t)owned.size();
        lists.assign(1, L);
    }
}

// May k_wm look this query's words up?  It reads them from the coverage query text as they stand, where WordMatcherLookup normalises each word again
// (lower + Normalize, WordMatcherLookup.cs:27-31) — the identity on an already prepared search text, checked here; and a query may hold at most
// INFX_MAX_WM_LISTS lists (each word: <= 2 + 2*maxLd1 dictionary hits + 1 affix list).
static bool wm_on_device(const HostIndex& ix, const ustr& st) {
    const auto& T = tables();
    int words = 0; bool same = true;
    for_each_word(st, [&](int off, int len) {
        if (len < 2) return;
        words++;
        for (int i = 0; i < len; i++) { const u16 c = st[off + i]; const u16 n = T.norm[T.lower[c]]; if (n != c || n == u' ') same = false; }
    });
    return same && words * (3 + 2 * ix.cfg.wmMaxLD1) <= INFX_MAX_WM_LISTS;
}
struct FusedIn { std::vector<infx_fused_query> fq; std::vector<infx_cov_query> cq; std::vector<infx_wm_list> lists; std::vector<int32_t> owned;
                 std::vector<infx_cov_query_long> cql; };      // cql: the batch's long-query table (queries beyond the fast Stage-2 envelope; cq[i].reserved = 1 + index)
// the long-query table goes to the stream in front of the Stage-2 call that consumes it (infx_search_fused / infx_shard_stage2)
static int32_t stage_long_queries(infx_session* S, const FusedIn& F) {
    if (F.cql.empty()) return INFX_OK;
    int32_t rc = infx_stage2_long_queries(S->stream, (uint32_t)F.cql.size(), F.cql.data());
    if (rc) g_eerr = infx_last_error();
    return rc;
}
static int32_t build_fused_inputs(infx_engine* e, infx_session* S, int32_t max_results, int32_t enable_coverage, FusedIn& F) {
    Batch& B = *S->batch; const HostIndex& ix = e->ix; const int threads = e->threads; const uint32_t nq = B.nq;
    std::vector<QueryPlan>& plans = S->lastPlans;
    auto& fq = F.fq; auto& cq = F.cq; auto& lists = F.lists; auto& owned = F.owned;
    const bool covEnabled = ix.cfg.enableCoverage && enable_coverage;
    fq.assign(nq, infx_fused_query{}); cq.assign(nq, infx_cov_query{});
    std::vector<std::vector<infx_wm_list>> qLists(nq); std::vector<std::vector<int32_t>> qOwned(nq);
    std::vector<int32_t> covErr(nq, 0);
    std::vector<std::unique_ptr<infx_cov_query_long>> qLong(nq);      // queries beyond the fast envelope
    F.cql.clear();
    bool devWm = e->devLookups && ix.cfg.wordMatcher;
    if (devWm) {      // ~2.5 looked-up words per query: estimated from the batch's text volume (one word per ~7 characters), 7 us of one core each
        size_t chars = 0; for (uint32_t i = 0; i < nq; i++) chars += plans[i].searchText.size();
        devWm = e->lookups_on_device(0.007 * (double)chars / 7.0);
    }
    std::atomic<long long> nDev{0}, nHost{0}; std::atomic<const char*> badRec{nullptr};
    parallel_dyn(nq, threads, 4, [&](int64_t b, int64_t en, int) {
        WmResult wm;
        for (int64_t i = b; i < en; i++) {
            const QueryPlan& P = plans[i]; infx_fused_query& F = fq[i];
            F = infx_fused_query{}; F.dev = -1; F.max_results = max_results;
            if (P.blank || P.unsupported) { F.flags = INFX_FQ_SKIP | (P.unsupported ? INFX_FQ_UNSUPPORTED : 0u); continue; }
            F.dev = B.devOf.empty() ? -1 : B.devOf[i];      // (pre-built in phase 0: filled in by fused_inputs_for_phase3)
            const ustr& st = P.searchText;
            bool isShort = !st.empty() && st.size() <= 3;
            if (isShort) for (u16 ch : st) if (is_delim(ch)) { isShort = false; break; }
            if (isShort) { F.flags |= INFX_FQ_SHORT; int64_t pk = ix.prefixKeys.find(st); int shortCount = pk >= 0 ? (int)ix.prefixPop[pk] : 0; if (shortCount > 500) F.flags |= INFX_FQ_SHORTSKIP; }
            if (!covEnabled || (F.flags & INFX_FQ_SHORTSKIP)) continue;
            F.flags |= INFX_FQ_COV;
            const infx_session::PlanPre* pp = S->planPre.size() == nq ? S->planPre[i].get() : nullptr;      // (ph_plan dropped the entries that were not made for this batch)
            if (pp && pp->hasCov) {
                covErr[i] = pp->covErr;
                if (!covErr[i] && pp->longCov) {
                    qLong[i].reset(new infx_cov_query_long);
                    if (pp->rec) { if (const char* why = parse_cov_record_long(st, pp->rec, pp->recLen, pp->covOff, *qLong[i])) badRec.store(why); }
                    else if (pp->own->cql) *qLong[i] = *pp->own->cql; else badRec.store("long coverage query missing");
                }
                else if (!covErr[i] && pp->rec) { if (const char* why = parse_cov_record(st, pp->rec, pp->recLen, pp->covOff, cq[i])) badRec.store(why); }
                else if (!covErr[i]) cq[i] = pp->own->cq;
            } else covErr[i] = prepare_cov_any(ix, st, cq[i], qLong[i]);
            if (devWm && !covErr[i] && !qLong[i] && wm_on_device(ix, st)) { F.flags |= INFX_FQ_WMDEV; nDev++; continue; }      // k_wm resolves the words of cq[i] (lookup.hip.inc; fast-envelope queries)
            auto pre = S->wmPre.find(st);            // computed by a peer rank (sharded planning): same index, same text, same descriptors
            if (pre != S->wmPre.end()) { qLists[i] = pre->second.lists; qOwned[i] = pre->second.owned; }
            else wm_descriptors(ix, st, wm, qLists[i], qOwned[i]);
            nHost++;
        }
    });
    if (badRec.load()) { S->planPre.clear(); S->planPeer.clear(); return efail(INFX_EINVAL, std::string("exchanged plan record: ") + badRec.load()); }
    // a query outside the Stage-2 envelope (INFX_MAX_QUERY_TOKENS / INFX_MAX_QUERY_CHARS / token length) is answered as "unsupported" (empty result,
    // flag bit 0) — it does not fail the other queries of the batch
    for (uint32_t i = 0; i < nq; i++) if (covErr[i]) { fq[i].flags = INFX_FQ_SKIP | INFX_FQ_UNSUPPORTED; qLists[i].clear(); qOwned[i].clear(); }
    e->wmOnDevice += nDev.load(); e->wmOnHost += nHost.load();
    for (uint32_t i = 0; i < nq; i++) if (qLong[i] && !covErr[i]) { std::memset(&cq[i], 0, sizeof cq[i]); cq[i].reserved = 1 + (int32_t)F.cql.size(); F.cql.push_back(*qLong[i]); }
    lists.clear(); owned.clear();
    for (uint32_t i = 0; i < nq; i++) {
        fq[i].wm_off = (uint32_t)lists.size(); fq[i].wm_count = (uint32_t)qLists[i].size();
        if (qLists[i].size() > INFX_MAX_WM_LISTS) return efail(INFX_ECAPACITY, "a query needs more than INFX_MAX_WM_LISTS WordMatcher lists");
        for (auto L : qLists[i]) { if (L.src == 2) L.off += owned.size(); lists.push_back(L); }
        owned.insert(owned.end(), qOwned[i].begin(), qOwned[i].end());
    }
    if (owned.size() > 0xFFFFFFF0ull) return efail(INFX_ECAPACITY, "affix matches of this batch exceed 2^32 ids; split the batch");
    S->wmPre.clear(); S->planPre.clear(); S->planPeer.clear();
    return INFX_OK;
}

// Sharded phases: phase 0 (which may run on a planner thread while another batch is in its collective phases) prepares the inputs;
// phase 3 only fills in what it learns later (the Stage-1 query index, max_results, the caller's coverage switch).
static int32_t fused_inputs_for_phase3(infx_engine* e, infx_session* S, int32_t max_results, int32_t enable_coverage, std::shared_ptr<FusedIn>& out) {
    Batch& B = *S->batch;
    if (!B.pre) { B.pre = std::make_shared<FusedIn>(); int32_t rc = build_fused_inputs(e, S, max_results, 1, *B.pre); if (rc) return rc; }
    FusedIn& F = *B.pre;
    const bool cov = e->ix.cfg.enableCoverage && enable_coverage;
    for (uint32_t i = 0; i < B.nq; i++) {
        infx_fused_query& q = F.fq[i];
        q.max_results = max_results;
        if (!(q.flags & INFX_FQ_SKIP)) q.dev = B.devOf.empty() ? -1 : B.devOf[i];
        if (!cov) { q.flags &= ~(INFX_FQ_COV | INFX_FQ_WMDEV); q.wm_count = 0; }
    }
    out = B.pre;
    return INFX_OK;
}

// Unsharded engines: the whole batch on the device with one synchronisation (infx_search_fused).  The host keeps what needs
// its dictionaries: text preparation, term lookup, LD1 expansion, idf / roles (ph_plan*), and per query the WordMatcher list
// descriptors (WordMatcher.Lookup, WordMatcher.cs:95-186) + CoverageEngine.PrepareQuery.
static void resolve_kernel_times(infx_session* S) {
    if (!S->kernelTimesPending || !S->stream) return;
    S->kernelTimesPending = false;
    float ms5[5] = {0, 0, 0, 0, 0}; infx_last_fused_timings(S->stream, ms5);
    S->msAcc = ms5[0]; S->msSel = ms5[1]; S->msPrep2 = ms5[2]; S->msCov = ms5[3]; S->msFin = ms5[4];
    infx_last_replay_stats(S->stream, &S->msReplay, S->flagWhy);
    infx_last_replay_breakdown(S->stream, S->msReplayParts);
}
static int32_t search_batch_fused(infx_engine* e, infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                  int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                  uint32_t* out_counts, uint32_t* out_flags) {
    int32_t rc; FusedIn FI;
    {   PlanGateHold hold(e->gate);
        rc = ph_plan(e, S, nq, q_arena, q_offs, depth); if (rc) return rc;
        rc = ph_plan_finish(e, S, S->batch->pendingCounts.data()); if (rc) return rc;
        rc = build_fused_inputs(e, S, max_results, enable_coverage, FI); if (rc) return rc;
    }
    Batch& B = *S->batch; const HostIndex& ix = e->ix;
    auto& fq = FI.fq; auto& cq = FI.cq; auto& lists = FI.lists; auto& owned = FI.owned;
    B.t2 = now_ms();
    const bool dbg = e->cfg.want_features != 0;
    rc = stage_long_queries(S, FI); if (rc) return rc;
    rc = infx_search_fused(S->stream, B.nd, B.dq.data(), (uint32_t)B.dterms.size(), B.dterms.data(), nq, fq.data(), cq.data(),
                           (uint32_t)lists.size(), lists.data(), (uint32_t)owned.size(), owned.data(), depth, max_results, dbg ? 1 : 0,
                           out_keys, out_scores, out_ties, out_counts, out_flags);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    B.t3 = now_ms();
    // Kernel durations are resolved when somebody asks (infx_engine_session_last_timings): every hipEventElapsedTime is a HIP API call that queues behind the
    // launches of the other sessions (measured: 2.8-3.3 ms of "post" time per batch on two of three boxes with six sessions).
    S->kernelTimesPending = true;
    float ms5[5] = {0, 0, 0, 0, 0};
    static const bool dbgTimes = getenv("INFX_DEBUG") != nullptr;
    if (dbgTimes) { resolve_kernel_times(S); ms5[0] = S->msAcc; ms5[1] = S->msSel; ms5[2] = S->msPrep2; ms5[3] = S->msCov; ms5[4] = S->msFin; }
    uint64_t s1rows = 0; infx_last_fused_stats(S->stream, &s1rows, &S->s2Candidates, &S->s2TextBytes);
    infx_last_alg_bytes(S->stream, &S->streamedBytes); infx_last_candidates(S->stream, &S->s1Candidates); infx_last_exact_replays(S->stream, &S->exactReplays);
    {   // B_alg(q) = sum_t df_t * 5 B (4 B for fuzzy virtual terms) + card(C_q) * 4 B + depth * 12 B
        uint64_t ab = 0;
        for (auto& t : B.dterms) ab += t.term_id >= 0 ? (uint64_t)e->shardTermLen(t.term_id) * 5ull : (uint64_t)t.extra_len * 4ull;
        S->algBytes = B.nd ? ab + S->s1Candidates * 4ull + s1rows * 12ull : 0;
    }
    if (dbg) {   // parity tests: pull the intermediate device tables into the session's introspection buffers
        const size_t stride = 2 * (size_t)depth;
        std::vector<infx_hit> s1((size_t)nq * depth); std::vector<uint32_t> s1c(nq), cc(nq), rcv(nq);
        std::vector<infx_cov_cand> cands(nq * stride); std::vector<infx_cov_out> outs(nq * stride); std::vector<int32_t> feat(nq * stride * INFX_NFEAT);
        rc = infx_fused_debug(S->stream, s1.data(), s1c.data(), cands.data(), outs.data(), feat.data(), cc.data(), rcv.data(), nullptr);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        S->lastHits.assign((size_t)B.nd * depth, infx_hit{0, 0.f}); S->lastHitCount.assign(B.nd, 0); S->lastStride = depth;
        S->lastCands.clear(); S->lastOuts.clear(); S->lastFeat.clear();
        uint32_t covIndex = 0;
        for (uint32_t i = 0; i < nq; i++) {
            const int j = B.devOf.empty() ? -1 : B.devOf[i];
            if (j >= 0) { S->lastHitCount[j] = s1c[i]; std::memcpy(S->lastHits.data() + (size_t)j * depth, s1.data() + (size_t)i * depth, (size_t)s1c[i] * sizeof(infx_hit)); }
            if (!(rcv[i] & 1)) continue;
            for (uint32_t k = 0; k < cc[i]; k++) {
                infx_cov_cand c = cands[i * stride + k]; c.query = covIndex;
                S->lastCands.push_back(c); S->lastOuts.push_back(outs[i * stride + k]);
                S->lastFeat.insert(S->lastFeat.end(), feat.begin() + (i * stride + k) * INFX_NFEAT, feat.begin() + (i * stride + k + 1) * INFX_NFEAT);
            }
            covIndex++;
        }
    }
    double t5 = now_ms();
    if (getenv("INFX_DEBUG"))
        fprintf(stderr, "[infx] fused nq=%u dev=%u terms=%zu lists=%zu owned=%zu cands=%llu | plan %.1f (tokens %.1f unions[%zu] %.1f finish %.1f) wm-prep %.1f gpu %.1f (acc %.2f sel %.2f prep2 %.2f s2 %.2f fin %.2f) post %.1f ms | fuzzy calls=%lld %.1f ms-cpu\n",
                nq, B.nd, B.dterms.size(), lists.size(), owned.size(), (unsigned long long)S->s2Candidates, B.t1 - B.t0, B.tTok, B.pending.size(), B.tUnion - B.tTok, B.tPlanPar - B.tUnion,
                B.t2 - B.t1, B.t3 - B.t2, ms5[0], ms5[1], ms5[2], ms5[3], ms5[4], t5 - B.t3, (long long)e->fuzzy.fuzzyCalls.exchange(0), e->fuzzy.fuzzyNs.exchange(0) / 1e6);
    S->tPrep1 = B.t1 - B.t0; S->tStage1 = 0; S->tPrep2 = B.t2 - B.t1; S->tStage2 = B.t3 - B.t2; S->tPost = t5 - B.t3;
    return INFX_OK;
}

static int32_t search_batch_impl(infx_engine* e, infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                 int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                 uint32_t* out_counts, uint32_t* out_flags) {
    if (!e || (nq && (!q_arena || !q_offs || !out_keys || !out_scores || !out_counts)) || max_results < 1) return efail(INFX_EINVAL, "bad arguments");
    if (!e->indexed) { for (uint32_t i = 0; i < nq; i++) out_counts[i] = 0; return INFX_OK; }   // Result.MakeEmptyResult(), SearchEngine.cs:261-262
    if (e->nranks > 1) return efail(INFX_EINVAL, "sharded engine: drive the phase API (infx_session_phase1..4) with the collectives in between");
    static const bool phased = getenv("INFX_PHASED") != nullptr;     // host-driven phases (the sharded code path) on one GPU
    if (!phased) return search_batch_fused(e, S, nq, q_arena, q_offs, max_results, depth, enable_coverage, out_keys, out_scores, out_ties, out_counts, out_flags);
    int32_t rc = ph_plan(e, S, nq, q_arena, q_offs, depth); if (rc) return rc;
    rc = ph_plan_finish(e, S, S->batch->pendingCounts.data()); if (rc) return rc;
    rc = ph_accumulate(e, S); if (rc) return rc;
    rc = ph_select(e, S, S->batch->counts.data()); if (rc) return rc;
    rc = ph_stage2(e, S, 1, S->lastHits.data(), S->lastHitCount.data(), max_results, enable_coverage); if (rc) return rc;
    return ph_finalize(e, S, S->lastOuts.data(), out_keys, out_scores, out_ties, out_counts, out_flags);
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
    infx_session* S = new infx_session(); S->e = e; S->batch = new Batch();
    int32_t rc = infx_stream_create(e->dev, &S->stream);
    if (rc) { g_eerr = infx_last_error(); delete S->batch; delete S; return rc; }
    *out = S; return INFX_OK;
}
void infx_engine_session_destroy(infx_session* S) { if (!S) return; if (S->stream) infx_stream_destroy(S->stream); delete S->batch; delete S; }
int32_t infx_engine_session_search_batch(infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                         int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                         uint32_t* out_counts, uint32_t* out_flags) {
    if (!S) return efail(INFX_EINVAL, "null session");
    return search_batch_impl(S->e, S, nq, q_arena, q_offs, max_results, depth, enable_coverage, out_keys, out_scores, out_ties, out_counts, out_flags);
}
// ---- sharded operation: phases with the collectives in between (infidex_amd/sharded.py) ----
int32_t infx_engine_add_synonym(infx_engine* e, const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb) {
    if (!e || !a || !b || la < 0 || lb < 0) return efail(INFX_EINVAL, "bad arguments");
    if (e->indexed) return efail(INFX_EINVAL, "synonyms must be added before IndexDocuments");
    e->ix.cfg.syn.add(uview((const u16*)a, (size_t)la), uview((const u16*)b, (size_t)lb));
    return INFX_OK;
}
int32_t infx_engine_set_shard(infx_engine* e, int32_t rank, int32_t nranks) {
    if (!e || nranks < 1 || rank < 0 || rank >= nranks) return efail(INFX_EINVAL, "bad shard arguments");
    if (e->indexed) return efail(INFX_EINVAL, "set the shard before IndexDocuments");
    e->rank = rank; e->nranks = nranks; return INFX_OK;
}
int32_t infx_engine_shard_info(infx_engine* e, int32_t* base, int32_t* n) { if (!e) return INFX_EINVAL; if (base) *base = e->shardBase; if (n) *n = e->shardN; return INFX_OK; }
int32_t infx_session_phase0(infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, uint32_t* nunions) {
    if (!S) return efail(INFX_EINVAL, "null session");
    // the exact cut across shards falls back to a chained sequential replay whose exchanged state is laid out for max_depth entries (infx_shard_replay_chain)
    if (S->e && S->e->nranks > 1 && depth != S->e->ix.cfg.maxDepth) return efail(INFX_EINVAL, "document shards search with CoverageDepth == the engine's max_depth");
    PlanGateHold hold(S->e->gate);
    int32_t rc = ph_plan(S->e, S, nq, q_arena, q_offs, depth); if (rc) return rc;
    if (nunions) *nunions = (uint32_t)S->batch->pending.size();
    static const bool hostPhases = getenv("INFX_PHASED") != nullptr;
    if (!hostPhases && nq) {     // WordMatcher descriptors + PrepareQuery now: phase 0 is off the collective path (sharded.py search_stream)
        S->batch->pre = std::make_shared<FusedIn>();
        rc = build_fused_inputs(S->e, S, 1, 1, *S->batch->pre); if (rc) return rc;
    }
    return INFX_OK;
}
// ---- sharded planning: the two expensive, index-wide host lookups of a batch — the LD1 expansion of unknown words (plan_tokens) and the
// WordMatcher dictionary / affix lookups (wm_collect), together ~85 % of the host time per query at 10 M documents — are pure functions of
// (index, text) and every rank holds the whole host index.  Rank r therefore computes them for queries [begin, end) of the coming batch only,
// the ranks all-gather the serialised results and import each other's share before phase 0, which then finds every LD1 expansion in the fuzzy
// cache and every descriptor list in the session (infidex_amd/sharded.py: ShardedSearcher._prefetch, on its own process group).
// Blob: u32 nLd1 { u16 len, u16 chars[len], u32 n, i32 members[n] }*  u32 nWm { u16 len, u16 chars[len], u32 nl, {u32 src, u32 len, u64 off}[nl], u32 no, i32 owned[no] }*
int64_t infx_session_prefetch_collect(infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, uint32_t begin, uint32_t end, int32_t depth) {
    if (!S || (nq && (!q_arena || !q_offs)) || begin > end || end > nq) { efail(INFX_EINVAL, "bad arguments"); return -1; }
    infx_engine* e = S->e; const HostIndex& ix = e->ix; const uint32_t n = end - begin;
    // With the dictionaries on the device the two lookups are kernels on every rank's own GPU and sections 1-2 stay empty; section 3 — the
    // plan exchange — is always there: the host work that is left (text preparation, term lookups, the coverage query context) for this rank's slice only.
    const bool hostLookups = !e->devLookups;
    std::vector<std::shared_ptr<infx_session::PlanPre>> pres(n);
    std::vector<std::vector<infx_wm_list>> qL(n); std::vector<std::vector<int32_t>> qO(n);
    parallel_dyn(n, e->threads, 1, [&](int64_t b, int64_t en, int) {
        WmResult wm;
        for (int64_t i = b; i < en; i++) {
            const uint32_t q = begin + (uint32_t)i;
            pres[i] = std::make_shared<infx_session::PlanPre>(); pres[i]->own.reset(new infx_session::OwnPlan); pres[i]->depth = depth;
            infx_session::PlanPre& R = *pres[i]; QueryPlan& P = R.own->plan;
            const u16* rp = (const u16*)q_arena + q_offs[q]; const size_t rl = (size_t)(q_offs[q + 1] - q_offs[q]);
            R.rawHash = raw_hash((const uint16_t*)rp, rl);
            plan_tokens_text(ix, uview(rp, rl), depth, P);
            if (hostLookups) plan_tokens_expand(ix, e->fuzzy, P, false);
            if (!P.blank && !P.unsupported && ix.cfg.enableCoverage) {
                if (hostLookups) wm_descriptors(ix, P.searchText, wm, qL[i], qO[i]);
                R.hasCov = true; R.covErr = prepare_cov_any(ix, P.searchText, R.own->cq, R.own->cql); R.longCov = (bool)R.own->cql;
            }
        }
    });
    std::vector<uint8_t>& blob = S->prefetchBlob; blob.clear(); BlobW W{blob};
    W.put<uint64_t>(index_fingerprint(ix));
    std::map<std::u16string, const FuzzyUnion*> words;          // ordered: the blob is a deterministic function of (index, slice)
    for (auto& R : pres) for (auto& r : R->own->plan.rawTok) if (r.fz) words.emplace(r.text, r.fz.get());
    W.put<uint32_t>((uint32_t)words.size());
    for (auto& kv : words) {
        W.put<uint16_t>((uint16_t)kv.first.size()); W.bytes(kv.first.data(), kv.first.size() * 2);
        W.put<uint32_t>((uint32_t)kv.second->members.size()); W.bytes(kv.second->members.data(), kv.second->members.size() * 4);
    }
    std::map<std::u16string, uint32_t> texts;
    if (hostLookups)
        for (uint32_t i = 0; i < n; i++) { const QueryPlan& P = pres[i]->own->plan; if (!P.blank && !P.unsupported && ix.cfg.enableCoverage && P.searchText.size() <= 0xFFFF) texts.emplace(P.searchText, i); }
    W.put<uint32_t>((uint32_t)texts.size());
    for (auto& kv : texts) {
        const uint32_t i = kv.second;
        W.put<uint16_t>((uint16_t)kv.first.size()); W.bytes(kv.first.data(), kv.first.size() * 2);
        W.put<uint32_t>((uint32_t)qL[i].size());
        for (auto& L : qL[i]) { W.put<uint32_t>(L.src); W.put<uint32_t>(L.len); W.put<uint64_t>(L.off); }
        W.put<uint32_t>((uint32_t)qO[i].size()); W.bytes(qO[i].data(), qO[i].size() * 4);
    }
    // section 3, the plan exchange: u32 begin, end, nq; i32 depth; per query of the slice
    //   (table of n + 1 u32 record offsets first, so that an importer can find a record without parsing the ones before it)
    //   u64 hash of the raw text; u8 flags (1 blank, 2 unsupported, 4 searchText == qtext, 8 tfidfQuery == searchText, 16 coverage query present, 32 .. outside the envelope,
    //   64 .. in the long envelope: infx_cov_query_long)
    //   str qtext; [str searchText]; [str tfidfQuery]; u16 nraw { i32 id; [str text] }*       (str = u32 length + UTF-16 units; unknown words drop their expansion:
    //   the importer looks them up in ITS cache / expands them with the batch, as for its own queries)
    //   coverage query (its text is searchText): i32 status | i32 num_tokens, u16 tok_off[], u16 tok_len[], f32 term_idf[], f32 word_idf[], i32 has_word_idf,
    //   i32 num_fusion_tokens, u16 ftok_off[], u16 ftok_len[], i32 lcs_tolerance
    auto str = [&](const ustr& t) { W.put<uint32_t>((uint32_t)t.size()); W.bytes(t.data(), t.size() * 2); };
    const size_t sec3 = blob.size();
    W.put<uint32_t>(begin); W.put<uint32_t>(end); W.put<uint32_t>(nq); W.put<int32_t>(depth);
    const size_t tab = blob.size(); blob.resize(tab + ((size_t)n + 1) * 4);      // u32 record offsets (n + 1), relative to the first record
    const size_t rec0 = blob.size();
    auto tabput = [&](uint32_t k) { const uint32_t o = (uint32_t)(blob.size() - rec0); std::memcpy(blob.data() + tab + (size_t)k * 4, &o, 4); };
    for (uint32_t i = 0; i < n; i++) {
        tabput(i);
        const infx_session::PlanPre& R = *pres[i]; const QueryPlan& P = R.own->plan;
        const bool sameST = P.searchText == P.qtext, sameTQ = P.tfidfQuery == P.searchText;
        W.put<uint64_t>(R.rawHash);
        W.put<uint8_t>((uint8_t)((P.blank ? 1 : 0) | (P.unsupported ? 2 : 0) | (sameST ? 4 : 0) | (sameTQ ? 8 : 0) | (R.hasCov ? 16 : 0) | (R.covErr ? 32 : 0) | (R.longCov ? 64 : 0)));
        str(P.qtext); if (!sameST) str(P.searchText); if (!sameTQ) str(P.tfidfQuery);
        W.put<uint16_t>((uint16_t)P.rawTok.size());
        for (auto& r : P.rawTok) { W.put<int32_t>(r.id); if (r.id < 0) str(r.text); }
        if (R.hasCov && R.covErr) W.put<int32_t>(R.covErr);
        else if (R.hasCov) {
            auto putcov = [&](const auto& C) {
                const size_t nt = (size_t)C.num_tokens, nf = (size_t)C.num_fusion_tokens;
                W.put<int32_t>(C.num_tokens); W.bytes(C.tok_off, nt * 2); W.bytes(C.tok_len, nt * 2); W.bytes(C.term_idf, nt * 4); W.bytes(C.word_idf, nt * 4);
                W.put<int32_t>(C.has_word_idf); W.put<int32_t>(C.num_fusion_tokens); W.bytes(C.ftok_off, nf * 2); W.bytes(C.ftok_len, nf * 2); W.put<int32_t>(C.lcs_tolerance);
            };
            if (R.longCov) putcov(*R.own->cql); else putcov(R.own->cq);
        }
    }
    tabput(n);
    W.put<uint64_t>(bytes_hash(blob.data() + sec3, blob.size() - sec3));      // the section's checksum: plans steer kernels, a damaged one must not be believed
    // this rank's own slice needs no round trip
    S->planPre.assign(nq, nullptr); S->planPeer.assign(nq, 0);
    for (uint32_t i = 0; i < n; i++) {
        for (auto& r : pres[i]->own->plan.rawTok) { r.fz = nullptr; r.pending = false; }      // phase 0 resolves them against the cache as it stands then
        S->planPre[begin + i] = std::move(pres[i]);
    }
    return (int64_t)blob.size();
}
int64_t infx_session_prefetch_pending(infx_session* S) { return S ? (int64_t)S->wmPre.size() : -1; }
int32_t infx_session_prefetch_blob(infx_session* S, uint8_t* out, int64_t cap) {
    if (!S || !out || cap < (int64_t)S->prefetchBlob.size()) return efail(INFX_EINVAL, "bad arguments");
    std::memcpy(out, S->prefetchBlob.data(), S->prefetchBlob.size()); return INFX_OK;
}
int32_t infx_session_prefetch_import(infx_session* S, const uint8_t* blob, int64_t len) {
    if (!S || !blob || len < 16) return efail(INFX_EINVAL, "bad arguments");
    infx_engine* e = S->e; const HostIndex& ix = e->ix;
    BlobR R{blob, blob + len};
    if (R.get<uint64_t>() != index_fingerprint(ix)) return efail(INFX_EINVAL, "prefetch blob comes from a different index (ranks must index the same corpus with the same build)");
    const uint32_t nw = R.get<uint32_t>();
    for (uint32_t k = 0; k < nw && R.ok; k++) {
        const uint16_t wl = R.get<uint16_t>(); const uint8_t* wp = R.take((size_t)wl * 2);
        const uint32_t nm = R.get<uint32_t>(); const uint8_t* mp = R.take((size_t)nm * 4);
        if (!R.ok) break;
        std::u16string word((size_t)wl, u'\0'); std::memcpy(&word[0], wp, (size_t)wl * 2);
        if (e->fuzzy.get(word)) continue;
        auto nf = std::make_shared<FuzzyUnion>(); nf->members.resize(nm); if (nm) std::memcpy(nf->members.data(), mp, (size_t)nm * 4);
        for (int32_t id : nf->members) if (id < 0 || (size_t)id >= ix.df.size()) return efail(INFX_EINVAL, "prefetch blob: term id out of range (ranks must hold the same index)");
        if (nf->members.empty()) nf->df.store(0);
        e->fuzzy.put(word, nf);
    }
    const uint32_t nt = R.ok ? R.get<uint32_t>() : 0;
    for (uint32_t k = 0; k < nt && R.ok; k++) {
        const uint16_t tl = R.get<uint16_t>(); const uint8_t* tp = R.take((size_t)tl * 2);
        const uint32_t nl = R.get<uint32_t>();
        infx_session::WmPre pre; pre.lists.resize(nl);
        for (uint32_t j = 0; j < nl && R.ok; j++) { pre.lists[j] = infx_wm_list{}; pre.lists[j].src = R.get<uint32_t>(); pre.lists[j].len = R.get<uint32_t>(); pre.lists[j].off = R.get<uint64_t>(); }
        const uint32_t no = R.get<uint32_t>(); const uint8_t* op = R.take((size_t)no * 4);
        if (!R.ok) break;
        pre.owned.resize(no); if (no) std::memcpy(pre.owned.data(), op, (size_t)no * 4);
        for (auto& L : pre.lists) {          // descriptors go straight to the device: reject anything outside the arrays they index
            const uint64_t lim = L.src == 0 ? ix.wmExact.doc.size() : (L.src == 1 ? ix.wmLd1.doc.size() : (L.src == 2 ? (uint64_t)no : 0));
            if (L.src > 2 || L.off > lim || (uint64_t)L.len > lim - L.off) return efail(INFX_EINVAL, "prefetch blob: WordMatcher descriptor out of range");
        }
        for (int32_t id : pre.owned) if (id < 0 || id >= ix.N) return efail(INFX_EINVAL, "prefetch blob: affix document id out of range");
        std::u16string text((size_t)tl, u'\0'); std::memcpy(&text[0], tp, (size_t)tl * 2);
        S->wmPre.emplace(std::move(text), std::move(pre));
    }
    if (!R.ok) return efail(INFX_EINVAL, "prefetch blob truncated");
    // section 3: the peer's slice of the batch's plans
    if ((size_t)(R.e - R.p) < 24 || [&] { uint64_t want; std::memcpy(&want, R.e - 8, 8); return want != bytes_hash(R.p, (size_t)(R.e - R.p) - 8); }())
        return efail(INFX_EINVAL, "prefetch blob: plan section damaged (checksum)");
    R.e -= 8;
    const uint32_t begin = R.get<uint32_t>(), end = R.get<uint32_t>(), bnq = R.get<uint32_t>(); const int32_t depth = R.get<int32_t>();
    if (!R.ok || begin > end || end > bnq) return efail(INFX_EINVAL, "prefetch blob: plan section truncated or its slice is out of range");
    const uint32_t n = end - begin;
    const uint8_t* offp = R.take(((size_t)n + 1) * 4);
    if (!offp) return efail(INFX_EINVAL, "prefetch blob truncated");
    // the records stay in a copy of the section; each query's entry points at its record, which phase 0 parses on its planner threads
    auto slab = std::make_shared<infx_session::PlanSlab>();
    slab->bytes.assign(R.p, R.e); slab->pre.resize(n);
    const size_t total = slab->bytes.size();
    std::vector<uint32_t> offs((size_t)n + 1); std::memcpy(offs.data(), offp, ((size_t)n + 1) * 4);
    if (offs[0] != 0 || offs[n] != total) return efail(INFX_EINVAL, "prefetch blob: plan record table does not match the section");
    for (uint32_t k = 0; k < n; k++) {
        if (offs[k + 1] < offs[k] || offs[k + 1] > total || offs[k + 1] - offs[k] < 9) return efail(INFX_EINVAL, "prefetch blob: plan record table does not match the section");
        infx_session::PlanPre& E = slab->pre[k];
        E.rec = slab->bytes.data() + offs[k]; E.recLen = offs[k + 1] - offs[k]; E.depth = depth;
        std::memcpy(&E.rawHash, E.rec, 8);
    }
    if (S->planPre.size() != bnq) { S->planPre.assign(bnq, nullptr); S->planPeer.assign(bnq, 0); }
    for (uint32_t k = 0; k < n; k++) { S->planPre[begin + k] = std::shared_ptr<infx_session::PlanPre>(slab, &slab->pre[k]); S->planPeer[begin + k] = 1; }
    return INFX_OK;
}
// Plan exchange, observed: out2 = {queries of the last phase 0 whose token-level plan came through infx_session_prefetch_* (own slice + imported), of them imported from peers}
int32_t infx_session_plan_exchange_stats(infx_session* S, uint32_t* out2) {
    if (!S || !out2) return efail(INFX_EINVAL, "null");
    out2[0] = S->planFromExchange; out2[1] = S->planFromPeers; return INFX_OK;
}
// Parity tooling (no device needed): one 64-bit digest per query of everything the plan exchange carries — flags, the three texts, the raw tokens, the coverage
// query context byte for byte — taken from the session's pending exchange entries where they exist and match the text, computed here otherwise; *from_exchange
// counts the former.  Equal digests with and without an exchange = the exchange changes nothing.  Does not consume the entries.
int32_t infx_session_plan_digest(infx_session* S, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, uint64_t* out, uint32_t* from_exchange) {
    if (!S || !out || (nq && (!q_arena || !q_offs))) return efail(INFX_EINVAL, "bad arguments");
    const HostIndex& ix = S->e->ix; uint32_t used = 0;
    for (uint32_t i = 0; i < nq; i++) {
        const u16* rp = (const u16*)q_arena + q_offs[i]; const size_t rl = (size_t)(q_offs[i + 1] - q_offs[i]);
        infx_session::PlanPre local; local.own.reset(new infx_session::OwnPlan); const infx_session::PlanPre* R = S->planPre.size() == nq ? S->planPre[i].get() : nullptr;
        if (R && R->depth == depth && R->rawHash == raw_hash((const uint16_t*)rp, rl)) {
            used++;
            if (R->rec) {      // a peer's record: parsed here as phase 0 would
                const char* why = parse_plan_record(ix, R->rec, R->recLen, depth, local.own->plan, local.hasCov, local.covErr, local.covOff, local.longCov);
                if (!why && local.hasCov && !local.covErr && local.longCov) { local.own->cql.reset(new infx_cov_query_long); why = parse_cov_record_long(local.own->plan.searchText, R->rec, R->recLen, local.covOff, *local.own->cql); }
                else if (!why && local.hasCov && !local.covErr) why = parse_cov_record(local.own->plan.searchText, R->rec, R->recLen, local.covOff, local.own->cq);
                if (why) return efail(INFX_EINVAL, std::string("exchanged plan record: ") + why);
                R = &local;
            }
        } else {
            plan_tokens_text(ix, uview(rp, rl), depth, local.own->plan);
            if (!local.own->plan.blank && !local.own->plan.unsupported && ix.cfg.enableCoverage) { local.hasCov = true; local.covErr = prepare_cov_any(ix, local.own->plan.searchText, local.own->cq, local.own->cql); local.longCov = (bool)local.own->cql; }
            R = &local;
        }
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 1099511628211ull; } const uint64_t nn = n; for (int k = 0; k < 8; k++) { h ^= (nn >> (8 * k)) & 0xFF; h *= 1099511628211ull; } };
        const QueryPlan& P = R->own->plan; const uint8_t fl = (uint8_t)((P.blank ? 1 : 0) | (P.unsupported ? 2 : 0) | (R->hasCov ? 4 : 0)); const int32_t d = P.depth;
        mix(&fl, 1); mix(&d, 4); mix(P.qtext.data(), P.qtext.size() * 2); mix(P.searchText.data(), P.searchText.size() * 2); mix(P.tfidfQuery.data(), P.tfidfQuery.size() * 2);
        for (auto& r : P.rawTok) { mix(&r.id, 4); mix(r.text.data(), r.text.size() * 2); }
        if (R->hasCov) { mix(&R->covErr, 4); const uint8_t lg = R->longCov ? 1 : 0; mix(&lg, 1); if (!R->covErr && R->longCov) mix(R->own->cql.get(), sizeof(infx_cov_query_long)); else if (!R->covErr) mix(&R->own->cq, sizeof R->own->cq); }
        out[i] = h;
    }
    if (from_exchange) *from_exchange = used;
    return INFX_OK;
}
int32_t infx_session_union_counts(infx_session* S, uint32_t* counts) {   // this shard's |union| of every pending fuzzy virtual term
    if (!S || !counts) return efail(INFX_EINVAL, "null");
    std::memcpy(counts, S->batch->pendingCounts.data(), S->batch->pendingCounts.size() * 4); return INFX_OK;
}
int32_t infx_session_phase1(infx_session* S, const uint32_t* global_union_counts, uint32_t* ndev) {
    if (!S) return efail(INFX_EINVAL, "null session");
    static const uint32_t zero = 0;
    int32_t rc = ph_plan_finish(S->e, S, global_union_counts ? global_union_counts : &zero); if (rc) return rc;
    rc = ph_accumulate(S->e, S); if (rc) return rc;
    if (ndev) *ndev = S->batch->nd;
    return INFX_OK;
}
// phase 1 with the class histograms written straight into caller memory on the host OR the device (nq x INFX_NCLASS uint32, the first ndev rows
// are used): with a device tensor the count all-reduce (Exchange 1) runs in place on it and infx_session_phase2x reads it back from HBM.
int32_t infx_session_phase1x(infx_session* S, const uint32_t* global_union_counts, void* counts, uint32_t* ndev) {
    if (!S || !counts) return efail(INFX_EINVAL, "null");
    static const uint32_t zero = 0;
    int32_t rc = ph_plan_finish(S->e, S, global_union_counts ? global_union_counts : &zero); if (rc) return rc;
    Batch& B = *S->batch;
    S->msAcc = S->msSel = S->msCov = S->msPrep2 = S->msFin = 0; S->algBytes = 0;
    if (B.nd) {
        rc = infx_stage1_accumulate(S->stream, B.nd, B.dq.data(), (uint32_t)B.dterms.size(), B.dterms.data(), (uint32_t)B.extra.size(), B.extra.data(), (infx_counts*)counts);
        if (rc) { g_eerr = infx_last_error(); return rc; }
    }
    if (ndev) *ndev = B.nd;
    return INFX_OK;
}
int32_t infx_session_counts(infx_session* S, uint32_t* counts) {   // nd x INFX_NCLASS, this shard
    if (!S || !counts) return efail(INFX_EINVAL, "null");
    std::memcpy(counts, S->batch->counts.data(), S->batch->counts.size() * sizeof(infx_counts)); return INFX_OK;
}
int32_t infx_session_phase2(infx_session* S, const uint32_t* global_counts, infx_hit* hits, uint32_t* hitcounts) {
    if (!S || !global_counts) return efail(INFX_EINVAL, "null");
    static const bool hostPhases = getenv("INFX_PHASED") != nullptr;      // the original host-side candidate assembly / final ordering
    if (hostPhases) {
        int32_t rc = ph_select(S->e, S, (const infx_counts*)global_counts); if (rc) return rc;
        if (hits) std::memcpy(hits, S->lastHits.data(), S->lastHits.size() * sizeof(infx_hit));
        if (hitcounts) std::memcpy(hitcounts, S->lastHitCount.data(), S->lastHitCount.size() * 4);
        return INFX_OK;
    }
    Batch& B = *S->batch;
    S->lastHits.assign((size_t)B.nd * B.depth, infx_hit{0, 0.f}); S->lastHitCount.assign(B.nd, 0); S->lastStride = B.depth;
    if (B.nd) {
        int32_t rc = infx_shard_select(S->stream, B.nd, (const infx_counts*)global_counts, B.depth, S->lastHits.data(), S->lastHitCount.data(), nullptr);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_timings(S->stream, &S->msAcc, &S->msSel, nullptr);
        infx_last_alg_bytes(S->stream, &S->streamedBytes); infx_last_candidates(S->stream, &S->s1Candidates); infx_last_exact_replays(S->stream, &S->exactReplays); infx_last_replay_stats(S->stream, &S->msReplay, S->flagWhy);
        uint64_t ab = 0, nh = 0;
        for (auto& t : B.dterms) ab += t.term_id >= 0 ? (uint64_t)S->e->shardTermLen(t.term_id) * 5ull : (uint64_t)t.extra_len * 4ull;
        for (uint32_t c : S->lastHitCount) nh += c;
        S->algBytes = ab + S->s1Candidates * 4ull + nh * 12ull;
    }
    if (hits) std::memcpy(hits, S->lastHits.data(), S->lastHits.size() * sizeof(infx_hit));
    if (hitcounts) std::memcpy(hitcounts, S->lastHitCount.data(), S->lastHitCount.size() * 4);
    B.t2 = now_ms();
    return INFX_OK;
}
// ---- phase 2 with the exact cut across shards (infx_shard_replay_*) ----
static void take_stage1_stats(infx_session* S, bool withHits) {
    Batch& B = *S->batch;
    infx_last_timings(S->stream, &S->msAcc, &S->msSel, nullptr);
    infx_last_alg_bytes(S->stream, &S->streamedBytes); infx_last_candidates(S->stream, &S->s1Candidates);
    uint64_t ab = 0;
    for (auto& t : B.dterms) ab += t.term_id >= 0 ? (uint64_t)S->e->shardTermLen(t.term_id) * 5ull : (uint64_t)t.extra_len * 4ull;
    S->algBytes = ab + S->s1Candidates * 4ull + (withHits ? (uint64_t)B.nd * B.depth * 12ull : 0ull);
}
int32_t infx_session_phase2a(infx_session* S, const void* global_counts, void* hits, void* hitcounts, void* next) {
    if (!S || !global_counts || !hits || !hitcounts || !next) return efail(INFX_EINVAL, "null");
    Batch& B = *S->batch;
    int32_t rc = infx_shard_select(S->stream, B.nd, (const infx_counts*)global_counts, B.depth, (infx_hit*)hits, (uint32_t*)hitcounts, (float*)next);     // nd == 0: resets the replay state
    if (rc) { g_eerr = infx_last_error(); return rc; }
    if (B.nd) take_stage1_stats(S, true);
    return INFX_OK;
}
int32_t infx_session_phase2b(infx_session* S, int32_t W, const void* all_hits, const void* all_hitcounts, const void* all_next, uint64_t* blob_bytes) {
    if (!S || W < 1 || !blob_bytes) return efail(INFX_EINVAL, "null");
    Batch& B = *S->batch;
    int32_t rc = infx_shard_replay_local(S->stream, W, B.nd, (const infx_hit*)all_hits, (const uint32_t*)all_hitcounts, (const float*)all_next, B.depth, blob_bytes);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    return INFX_OK;
}
int32_t infx_session_phase2b_blob(infx_session* S, void* dst, uint64_t padded) {
    if (!S || !dst) return efail(INFX_EINVAL, "null");
    int32_t rc = infx_shard_replay_blob(S->stream, dst, padded);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    return INFX_OK;
}
int32_t infx_session_phase2c(infx_session* S, int32_t W, const void* all_blobs, uint64_t padded, void* hits, void* hitcounts) {
    if (!S || W < 1 || !hits || !hitcounts) return efail(INFX_EINVAL, "null");
    Batch& B = *S->batch;
    if (B.nd) {
        int32_t rc = infx_shard_replay_merge(S->stream, W, B.nd, all_blobs, padded, B.depth, (infx_hit*)hits, (uint32_t*)hitcounts);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_exact_replays(S->stream, &S->exactReplays); infx_last_replay_stats(S->stream, &S->msReplay, S->flagWhy);
    }
    B.t2 = now_ms();
    return INFX_OK;
}
int32_t infx_session_phase2d(infx_session* S, const uint32_t* need, void* state) {
    if (!S || !need || !state) return efail(INFX_EINVAL, "null");
    Batch& B = *S->batch;
    int32_t rc = infx_shard_replay_chain(S->stream, B.nd, need, B.depth, state);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    return INFX_OK;
}
// ---- native driver of the sharded phases (infidex_engine.h) ---------------------------------------------------------------------------------------------
namespace {
int32_t rccl_allreduce(void* ctx, void* buf, uint64_t count, void* /*stream*/) { return infx_comm_allreduce_sum_u32((infx_stream*)ctx, buf, count); }
int32_t rccl_allgather(void* ctx, const void* send, void* recv, uint64_t bytes, void* /*stream*/) { return infx_comm_allgather((infx_stream*)ctx, send, recv, bytes); }
// exchange buffers of one batch: HBM scratch of the session's stream (RCCL) or host vectors (other transports)
struct XBufs {
    infx_session* S; bool dev; std::vector<std::vector<uint8_t>> host; int slot = 0;
    XBufs(infx_session* s, bool d) : S(s), dev(d) { host.reserve(16); }
    int32_t get(size_t bytes, void** out, bool zero) {
        bytes = std::max<size_t>(bytes, 16);
        if (dev) { int32_t rc = infx_stream_scratch(S->stream, slot++, bytes, out); if (rc) return rc; return zero ? infx_stream_fill0(S->stream, *out, bytes) : INFX_OK; }
        host.emplace_back(bytes, (uint8_t)0); *out = host.back().data(); return INFX_OK;
    }
};
}
int32_t infx_engine_rccl_unique_id(void* id128) { int32_t rc = infx_rccl_unique_id(id128); if (rc) g_eerr = infx_last_error(); return rc; }
int32_t infx_engine_comm_rccl(infx_engine* e, const void* id128, infx_comm* out) {
    if (!e || !id128 || !out) return efail(INFX_EINVAL, "null argument");
    if (!e->dev || !e->indexed) return efail(INFX_EINVAL, "the RCCL communicator is created on an indexed engine with a GPU");
    int32_t rc = infx_set_shard_comm(e->dev, id128);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    std::memset(out, 0, sizeof *out);
    out->ctx = nullptr;                      // the session's stream is supplied per call (the collective runs on ITS HIP stream)
    out->rank = e->rank; out->nranks = e->nranks; out->device_buffers = 1;
    out->allreduce_sum_u32 = rccl_allreduce; out->allgather = rccl_allgather;
    return INFX_OK;
}
int32_t infx_session_comm_rccl(infx_session* S, const void* id128, infx_comm* out) {
    if (!S || !id128 || !out) return efail(INFX_EINVAL, "null argument");
    if (!S->stream) return efail(INFX_EINVAL, "the session has no GPU stream");
    int32_t rc = infx_stream_comm(S->stream, id128);
    if (rc) { g_eerr = infx_last_error(); return rc; }
    std::memset(out, 0, sizeof *out);
    out->rank = S->e->rank; out->nranks = S->e->nranks; out->device_buffers = 1;
    out->allreduce_sum_u32 = rccl_allreduce; out->allgather = rccl_allgather;
    return INFX_OK;
}
int32_t infx_session_sharded_finish(infx_session* S, const infx_comm* comm, int32_t max_results, int32_t enable_coverage,
                                    int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags) {
    if (!S || !comm || !comm->allreduce_sum_u32 || !comm->allgather || !out_keys || !out_scores || !out_counts || max_results < 1) return efail(INFX_EINVAL, "bad arguments");
    infx_engine* e = S->e; Batch& B = *S->batch;
    if (comm->nranks != e->nranks || comm->rank != e->rank) return efail(INFX_EINVAL, "communicator and engine disagree about the shard layout");
    const int W = comm->nranks; const bool dev = comm->device_buffers != 0;
    void* const cctx = comm->ctx ? comm->ctx : (void*)S->stream;       // the in-library RCCL ops run on the session's stream
    void* hs = nullptr;                                                // what a caller-supplied op with device buffers orders itself on: the session's hipStream_t
    if (dev && infx_stream_native(S->stream, &hs) != INFX_OK) return efail(INFX_EINVAL, "device exchange buffers need a session with a GPU stream");
    auto chk = [&](int32_t rc) { if (rc && g_eerr.empty()) g_eerr = infx_last_error(); return rc; };
    // An error return leaves the rank's collective ring: a session that stays a member but issues no further collective would block every peer
    // session of this rank in CollSeq::enter() for ever.  (The peer RANKS then see this session's collectives missing and time out: INFX_COMM_TIMEOUT_S.)
    struct RetireOnError { infx_engine* e; infx_session* S; bool armed = true; ~RetireOnError() { if (armed) e->collSeq.retire(S); } } onError{e, S};
#define XCHK(x) do { int32_t rc_ = chk(x); if (rc_) return rc_; } while (0)
    // one turn of the rank's collective ring per collective (CollSeq above); INFX_COLL_ORDER=0 switches the ordering off
    static const bool orderedEnv = [] { const char* v = getenv("INFX_COLL_ORDER"); return !(v && v[0] == '0'); }();
    const bool ordered = orderedEnv && comm->nranks > 1;      // a single rank has no peer to fall out of step with (measured cost of the lock-step at W = 1: 12 %)
    struct Turn { CollSeq& q; const infx_session* S; bool in; Turn(CollSeq& x, const infx_session* s_, bool on) : q(x), S(s_), in(on && x.enter(s_)) {} ~Turn() { if (in) q.leave(S); } };
    auto all_reduce = [&](void* buf, uint64_t count) { Turn t(e->collSeq, S, ordered); S->collCalls[0]++; S->collBytes[0] += count * 4; return comm->allreduce_sum_u32(cctx, buf, count, hs); };
    auto all_gather = [&](const void* send, void* recv, uint64_t bytes) { Turn t(e->collSeq, S, ordered); S->collCalls[1]++; S->collBytes[1] += bytes; return comm->allgather(cctx, send, recv, bytes, hs); };
    const uint32_t nq = B.nq; const int depth = B.depth;
    if (W > 1 && depth != e->ix.cfg.maxDepth) return efail(INFX_EINVAL, "document shards search with CoverageDepth == the engine's max_depth (the chained replay exchanges heaps of max_depth entries)");
    XBufs X(S, dev);
    // Exchange 1b: global df of the batch's new fuzzy unions (the host needs the values: idf is computed there with the reference's logf)
    std::vector<uint32_t> guc(B.pendingCounts);
    if (!guc.empty()) {
        if (dev) {
            void* d = nullptr; XCHK(X.get(guc.size() * 4, &d, false));
            XCHK(infx_stream_copy(S->stream, d, guc.data(), guc.size() * 4));
            XCHK(all_reduce(d, guc.size()));
            XCHK(infx_stream_copy(S->stream, guc.data(), d, guc.size() * 4)); XCHK(infx_stream_wait(S->stream));
        } else XCHK(all_reduce(guc.data(), guc.size()));
    }
    static const uint32_t zero = 0;
    // phase 1 + Exchange 1: class histograms (tier decisions need GLOBAL cardinalities, Q11)
    void* counts = nullptr; XCHK(X.get((size_t)std::max<uint32_t>(nq, 1) * INFX_NCLASS * 4, &counts, true));
    uint32_t nd = 0; XCHK(infx_session_phase1x(S, guc.empty() ? &zero : guc.data(), counts, &nd));
    XCHK(all_reduce(counts, (uint64_t)std::max<uint32_t>(nq, 1) * INFX_NCLASS));
    // phase 2a + Exchange 2a: first-pass lists, counts, best score left out
    const size_t ndp = std::max<uint32_t>(nd, 1), hitB = ndp * depth * sizeof(infx_hit);
    // the three pieces a rank contributes (lists | counts | best scores left out) are one packed block: ONE all-gather, unpacked into the per-piece arrays
    void *pack = nullptr, *apack = nullptr, *ah = nullptr, *ac = nullptr, *an = nullptr;
    const size_t packB = hitB + 2 * ndp * 4;
    XCHK(X.get(packB, &pack, true)); XCHK(X.get(packB * W, &apack, false));
    void* hits = pack; void* hc = (char*)pack + hitB; void* nxt = (char*)pack + hitB + ndp * 4;
    XCHK(X.get(hitB * W, &ah, false)); XCHK(X.get(ndp * 4 * W, &ac, false)); XCHK(X.get(ndp * 4 * W, &an, false));
    XCHK(infx_session_phase2a(S, counts, hits, hc, nxt));
    XCHK(all_gather(pack, apack, packB));
    { const uint64_t pb[3] = {hitB, ndp * 4, ndp * 4}; void* const ds3[3] = {ah, ac, an}; XCHK(infx_stream_unpack(S->stream, apack, packB, W, 3, pb, ds3)); }
    // phase 2b + Exchange 2c: this shard's part of the exact replay, packed; padded to the largest blob of the world
    uint64_t blobBytes = 0; XCHK(infx_session_phase2b(S, W, ah, ac, an, &blobBytes));
    uint64_t pad = blobBytes;
    {   // max over ranks through the sum-all-reduce the communicator has: one slot per rank
        std::vector<uint32_t> sz((size_t)W, 0u); sz[comm->rank] = (uint32_t)((blobBytes + 15) >> 4);
        if (dev) {
            void* d = nullptr; XCHK(X.get(sz.size() * 4, &d, false));
            XCHK(infx_stream_copy(S->stream, d, sz.data(), sz.size() * 4));
            XCHK(all_reduce(d, sz.size()));
            XCHK(infx_stream_copy(S->stream, sz.data(), d, sz.size() * 4)); XCHK(infx_stream_wait(S->stream));
        } else XCHK(all_reduce(sz.data(), sz.size()));
        pad = 16ull * *std::max_element(sz.begin(), sz.end());
    }
    void *blob = nullptr, *ab = nullptr; XCHK(X.get(pad, &blob, false)); XCHK(X.get(pad * W, &ab, false));
    XCHK(infx_session_phase2b_blob(S, blob, pad));
    XCHK(all_gather(blob, ab, pad));
    // phase 2c + Exchange 2b: owner-side heap; the all-gather of the per-rank final lists
    XCHK(infx_session_phase2c(S, W, ab, pad, hits, hc));
    XCHK(all_gather(pack, apack, hitB + ndp * 4));      // lists | counts, one collective
    { const uint64_t pb[2] = {hitB, ndp * 4}; void* const ds2[2] = {ah, ac}; XCHK(infx_stream_unpack(S->stream, apack, hitB + ndp * 4, W, 2, pb, ds2)); }
    // queries the parallel replay could not certify (rare): the literal sequential replay, shard after shard
    std::vector<uint32_t> fc((size_t)W * ndp);
    XCHK(infx_stream_copy(S->stream, fc.data(), ac, fc.size() * 4)); XCHK(infx_stream_wait(S->stream));
    std::vector<uint32_t> need(ndp, 0u); bool anyNeed = false;
    for (int w = 0; w < W; w++) for (uint32_t q = 0; q < nd; q++) if (fc[(size_t)w * ndp + q] == 0xFFFFFFFFu) { need[q] = 1; anyNeed = true; }
    if (anyNeed) {
        const size_t words = ndp * (2 + 2 * (size_t)depth);
        std::vector<uint32_t> state(words, 0u), all(words * W);
        void *ds = nullptr, *da = nullptr;
        if (dev) { XCHK(X.get(words * 4, &ds, false)); XCHK(X.get(words * 4 * W, &da, false)); }
        for (int r = 0; r < W; r++) {
            if (r == comm->rank) XCHK(infx_session_phase2d(S, need.data(), state.data()));
            if (dev) {
                XCHK(infx_stream_copy(S->stream, ds, state.data(), words * 4)); XCHK(all_gather(ds, da, words * 4));
                XCHK(infx_stream_copy(S->stream, all.data(), da, words * 4 * W)); XCHK(infx_stream_wait(S->stream));
            } else XCHK(all_gather(state.data(), all.data(), words * 4));
            std::memcpy(state.data(), all.data() + (size_t)r * words, words * 4);          // rank r's continuation is the state of record
        }
        std::vector<infx_hit> fh((size_t)W * ndp * depth);
        XCHK(infx_stream_copy(S->stream, fh.data(), ah, fh.size() * sizeof(infx_hit))); XCHK(infx_stream_wait(S->stream));
        for (uint32_t q = 0; q < nd; q++) if (need[q]) {
            const uint32_t* st = state.data() + (size_t)q * (2 + 2 * (size_t)depth); const uint32_t n = st[0];
            for (int w = 0; w < W; w++) { fc[(size_t)w * ndp + q] = 0; std::memset(fh.data() + ((size_t)w * ndp + q) * depth, 0, (size_t)depth * sizeof(infx_hit)); }
            for (uint32_t i = 0; i < n; i++) { infx_hit h; h.doc = (int32_t)st[2 + i]; std::memcpy(&h.score, &st[2 + depth + i], 4); fh[(size_t)q * depth + i] = h; }
            fc[q] = n;
        }
        XCHK(infx_stream_copy(S->stream, ah, fh.data(), fh.size() * sizeof(infx_hit))); XCHK(infx_stream_copy(S->stream, ac, fc.data(), fc.size() * 4));
        if (dev) XCHK(infx_stream_wait(S->stream));
    }
    // phase 3 + the all-reduce of the disjoint Stage-2 rows + phase 4
    void* outs = nullptr; XCHK(X.get((size_t)std::max<uint32_t>(nq, 1) * 2 * depth * sizeof(infx_cov_out), &outs, true));
    XCHK(infx_session_phase3x(S, W, ah, ac, max_results, enable_coverage, outs));
    XCHK(all_reduce(outs, (uint64_t)std::max<uint32_t>(nq, 1) * 2 * depth * 3));
    XCHK(infx_session_phase4(S, (const int32_t*)outs, out_keys, out_scores, out_ties, out_counts, out_flags));
#undef XCHK
    onError.armed = false;
    return INFX_OK;
}

int32_t infx_engine_coll_ring(infx_engine* e, uint32_t n, infx_session* const* sessions) {
    if (!e || (n && !sessions)) return efail(INFX_EINVAL, "null argument");
    for (uint32_t i = 0; i < n; i++) if (!sessions[i] || sessions[i]->e != e) return efail(INFX_EINVAL, "a ring session belongs to another engine");
    e->collSeq.set_ring(sessions, n);
    return INFX_OK;
}
int32_t infx_session_coll_retire(infx_session* S) { if (!S || !S->e) return efail(INFX_EINVAL, "null session"); S->e->collSeq.retire(S); return INFX_OK; }
int32_t infx_session_coll_stats(infx_session* S, uint64_t* out4) {      // all-reduce calls, all-gather calls, all-reduce bytes, all-gather bytes (cumulative)
    if (!S || !out4) return efail(INFX_EINVAL, "null argument");
    out4[0] = S->collCalls[0]; out4[1] = S->collCalls[1]; out4[2] = S->collBytes[0]; out4[3] = S->collBytes[1];
    return INFX_OK;
}
int32_t infx_session_phase3(infx_session* S, int32_t W, const infx_hit* all_hits, const uint32_t* all_counts, int32_t max_results, int32_t enable_coverage, uint64_t* ncand) {
    if (!S || W < 1 || max_results < 1) return efail(INFX_EINVAL, "bad arguments");
    static const bool hostPhases = getenv("INFX_PHASED") != nullptr;
    if (hostPhases) {
        int32_t rc = ph_stage2(S->e, S, W, all_hits, all_counts, max_results, enable_coverage); if (rc) return rc;
        if (ncand) *ncand = S->lastCands.size();
        return INFX_OK;
    }
    infx_engine* e = S->e; Batch& B = *S->batch;
    B.maxResults = max_results;
    std::shared_ptr<FusedIn> FIp; int32_t rc = fused_inputs_for_phase3(e, S, max_results, enable_coverage, FIp); if (rc) return rc;
    FusedIn& FI = *FIp;
    B.t3 = now_ms();
    S->lastOuts.assign((size_t)B.nq * 2 * B.depth, infx_cov_out{});
    if (B.nq) {
        rc = stage_long_queries(S, FI); if (rc) return rc;
        rc = infx_shard_stage2(S->stream, W, B.nd, all_hits, all_counts, B.nq, FI.fq.data(), FI.cq.data(), (uint32_t)FI.lists.size(), FI.lists.data(),
                               (uint32_t)FI.owned.size(), FI.owned.data(), B.depth, max_results, 0, S->lastOuts.data());
        if (rc) { g_eerr = infx_last_error(); return rc; }
        float ms5[5] = {0, 0, 0, 0, 0}; infx_last_fused_timings(S->stream, ms5); S->msPrep2 = ms5[2]; S->msCov = ms5[3];
        infx_last_fused_stats(S->stream, nullptr, &S->s2Candidates, &S->s2TextBytes);
    }
    if (ncand) *ncand = S->lastOuts.size();
    B.t4 = now_ms();
    return INFX_OK;
}
// Variants whose exchange buffers are caller memory on the host OR the device (e.g. the tensors an RCCL collective works on):
// nothing is staged through the session's host vectors.
int32_t infx_session_phase2x(infx_session* S, const uint32_t* global_counts, void* hits, void* hitcounts) {
    if (!S || !global_counts || !hits || !hitcounts) return efail(INFX_EINVAL, "null");
    Batch& B = *S->batch;
    if (B.nd) {
        int32_t rc = infx_shard_select(S->stream, B.nd, (const infx_counts*)global_counts, B.depth, (infx_hit*)hits, (uint32_t*)hitcounts, nullptr);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        infx_last_timings(S->stream, &S->msAcc, &S->msSel, nullptr);
        infx_last_alg_bytes(S->stream, &S->streamedBytes); infx_last_candidates(S->stream, &S->s1Candidates); infx_last_exact_replays(S->stream, &S->exactReplays); infx_last_replay_stats(S->stream, &S->msReplay, S->flagWhy);
        uint64_t ab = 0;
        for (auto& t : B.dterms) ab += t.term_id >= 0 ? (uint64_t)S->e->shardTermLen(t.term_id) * 5ull : (uint64_t)t.extra_len * 4ull;
        S->algBytes = ab + S->s1Candidates * 4ull + (uint64_t)B.nd * B.depth * 12ull;
    }
    B.t2 = now_ms();
    return INFX_OK;
}
int32_t infx_session_phase3x(infx_session* S, int32_t W, const void* all_hits, const void* all_counts, int32_t max_results, int32_t enable_coverage, void* outs) {
    if (!S || W < 1 || max_results < 1 || !outs) return efail(INFX_EINVAL, "bad arguments");
    infx_engine* e = S->e; Batch& B = *S->batch;
    B.maxResults = max_results;
    std::shared_ptr<FusedIn> FIp; int32_t rc = fused_inputs_for_phase3(e, S, max_results, enable_coverage, FIp); if (rc) return rc;
    FusedIn& FI = *FIp;
    B.t3 = now_ms();
    if (B.nq) {
        rc = stage_long_queries(S, FI); if (rc) return rc;
        rc = infx_shard_stage2(S->stream, W, B.nd, (const infx_hit*)all_hits, (const uint32_t*)all_counts, B.nq, FI.fq.data(), FI.cq.data(), (uint32_t)FI.lists.size(), FI.lists.data(),
                               (uint32_t)FI.owned.size(), FI.owned.data(), B.depth, max_results, 0, (infx_cov_out*)outs);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        float ms5[5] = {0, 0, 0, 0, 0}; infx_last_fused_timings(S->stream, ms5); S->msPrep2 = ms5[2]; S->msCov = ms5[3];
        infx_last_fused_stats(S->stream, nullptr, &S->s2Candidates, &S->s2TextBytes);
    }
    B.t4 = now_ms();
    return INFX_OK;
}
int32_t infx_session_outs(infx_session* S, int32_t* outs3) {   // ncand x 3 int32 words; zeros for candidates another shard owns
    if (!S || !outs3) return efail(INFX_EINVAL, "null");
    static_assert(sizeof(infx_cov_out) == 12, "infx_cov_out is exchanged as 3 int32 words");
    std::memcpy(outs3, S->lastOuts.data(), S->lastOuts.size() * sizeof(infx_cov_out)); return INFX_OK;
}
int32_t infx_session_phase4(infx_session* S, const int32_t* merged_outs3, int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags) {
    if (!S || !merged_outs3 || !out_keys || !out_scores || !out_counts) return efail(INFX_EINVAL, "null");
    static const bool hostPhases = getenv("INFX_PHASED") != nullptr;
    if (hostPhases) return ph_finalize(S->e, S, (const infx_cov_out*)merged_outs3, out_keys, out_scores, out_ties, out_counts, out_flags);
    Batch& B = *S->batch;
    if (B.nq) {
        int32_t rc = infx_shard_finalize(S->stream, B.nq, (const infx_cov_out*)merged_outs3, B.depth, B.maxResults, out_keys, out_scores, out_ties, out_counts, out_flags);
        if (rc) { g_eerr = infx_last_error(); return rc; }
        float ms5[5] = {0, 0, 0, 0, 0}; infx_last_fused_timings(S->stream, ms5); S->msFin = ms5[4];
    }
    double t5 = now_ms();
    S->tPrep1 = B.t1 - B.t0; S->tStage1 = B.t2 - B.t1; S->tPrep2 = B.t3 - B.t2; S->tStage2 = B.t4 - B.t3; S->tPost = t5 - B.t4;
    return INFX_OK;
}
int32_t infx_engine_default_session(infx_engine* e, infx_session** out) { if (!e || !out) return INFX_EINVAL; *out = e->def; return INFX_OK; }

int32_t infx_engine_session_last_timings(infx_session* S, double* host_ms5, float* kernel_ms3, uint64_t* alg_bytes3) {
    if (!S) return efail(INFX_EINVAL, "null");
    if (host_ms5) { host_ms5[0] = S->tPrep1; host_ms5[1] = S->tStage1; host_ms5[2] = S->tPrep2; host_ms5[3] = S->tStage2; host_ms5[4] = S->tPost; }
    if (kernel_ms3) resolve_kernel_times(S);
    if (kernel_ms3) { kernel_ms3[0] = S->msAcc; kernel_ms3[1] = S->msSel; kernel_ms3[2] = S->msCov; kernel_ms3[3] = S->msPrep2; kernel_ms3[4] = S->msFin; }
    if (alg_bytes3) { alg_bytes3[0] = S->algBytes; alg_bytes3[1] = S->s2Candidates; alg_bytes3[2] = S->s2TextBytes; alg_bytes3[3] = S->streamedBytes; alg_bytes3[4] = S->s1Candidates; alg_bytes3[5] = S->exactReplays; }
    return INFX_OK;
}

int32_t infx_engine_session_replay_breakdown(infx_session* S, float* ms4) {      // k_ex_scan, k_ex_chunk, k_ex_heap, k_exact1 of the last batch (ms)
    if (!S || !ms4) return INFX_EINVAL;
    resolve_kernel_times(S);
    for (int i = 0; i < 4; i++) ms4[i] = S->msReplayParts[i];
    return INFX_OK;
}
int32_t infx_engine_session_plan_breakdown(infx_session* S, double* out4) {      // of the last batch's plan_ms: tokens + term lookups (incl. the LD1 call), of that inside infx_ld1_expand, inside infx_union_build, idf / roles
    if (!S || !out4 || !S->batch) return INFX_EINVAL;
    const Batch& B = *S->batch;
    out4[0] = B.tTok; out4[1] = B.tLd1Dev; out4[2] = B.tUnionDev; out4[3] = B.t1 - B.t0 - B.tUnion;
    return INFX_OK;
}
int32_t infx_engine_session_last_replay(infx_session* S, float* ms, uint32_t* why3) {
    if (!S) return efail(INFX_EINVAL, "null");
    resolve_kernel_times(S);
    if (ms) *ms = S->msReplay; if (why3) { why3[0] = S->flagWhy[0]; why3[1] = S->flagWhy[1]; why3[2] = S->flagWhy[2]; }
    return INFX_OK;
}

int32_t infx_engine_last_timings(infx_engine* e, double* host_ms5, float* kernel_ms3, uint64_t* alg_bytes3) {
    if (!e) return efail(INFX_EINVAL, "null");
    infx_session* S = e->def;
    if (host_ms5) { host_ms5[0] = S->tPrep1; host_ms5[1] = S->tStage1; host_ms5[2] = S->tPrep2; host_ms5[3] = S->tStage2; host_ms5[4] = S->tPost; }
    if (kernel_ms3) { kernel_ms3[0] = S->msAcc; kernel_ms3[1] = S->msSel; kernel_ms3[2] = S->msCov; kernel_ms3[3] = S->msPrep2; kernel_ms3[4] = S->msFin; }
    if (alg_bytes3) { alg_bytes3[0] = S->algBytes; alg_bytes3[1] = S->s2Candidates; alg_bytes3[2] = S->s2TextBytes; alg_bytes3[3] = S->streamedBytes; alg_bytes3[4] = S->s1Candidates; alg_bytes3[5] = S->exactReplays; }
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
    if (!e || len < 0 || (len && !q) || (cap > 0 && !out)) return -1;
    std::vector<int> m; int c = match_ld1(e->ix, uview((const u16*)q, len), m, cap);
    for (size_t i = 0; i < m.size(); i++) out[i] = m[i];
    return c;
}
// Stage-1 plan of one query (no GPU needed): returns number of terms; mode/prefix_set/n_and/df_s1/df_s2 in meta[5]
int32_t infx_engine_match_ld1_forward(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap) {   // the literal trie walk (test cross-check)
    if (!e) return -1;
    std::vector<int> m; int c = match_ld1_forward(e->ix, uview((const u16*)q, (size_t)len), m, cap);
    for (size_t i = 0; i < m.size() && (int32_t)i < cap; i++) out[i] = m[i];
    return c;
}
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
// Host planning profile (measurement hook, no device needed): the per-query host work of a batch, single-threaded, by stage, in microseconds
// per query: [0] plan_tokens (text preparation, term lookups, LD1 expansion), [1] of that: LD1 walks, [2] plan_finish (idf, roles, modes),
// [3] wm_collect (WordMatcher descriptors), [4] prepare_cov_query, [5] plan_tokens_text (the share of [0] that the plan exchange moves to the slice's owner),
// [6] import of the batch's exchanged plans (with host lookups also the LD1 member lists and WordMatcher descriptors), [7] parsing the imported records in
// phase 0 (what an imported query costs there instead of [5] + [4]).  Pending fuzzy unions get a stand-in df.
int32_t infx_engine_host_plan_profile(infx_engine* e, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, double* out_us) {
    if (!e || !out_us || (nq && (!q_arena || !q_offs))) return efail(INFX_EINVAL, "null argument");
    const HostIndex& ix = e->ix; FuzzyCache fc;
    std::vector<QueryPlan> plans(nq);
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < nq; i++) plan_tokens(ix, fc, uview((const u16*)q_arena + q_offs[i], (size_t)(q_offs[i + 1] - q_offs[i])), depth, plans[i], false);
    auto t1 = std::chrono::steady_clock::now();
    for (auto& P : plans) for (auto& r : P.rawTok) if (r.fz && r.fz->df.load() < 0) r.fz->df.store((int)std::max<size_t>(1, r.fz->members.size()));
    auto t2 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < nq; i++) plan_finish(ix, plans[i]);
    auto t3 = std::chrono::steady_clock::now();
    WmResult wm; size_t sink = 0;
    for (uint32_t i = 0; i < nq; i++) { const QueryPlan& P = plans[i]; if (P.blank || P.unsupported) continue; wm_collect(ix, P.searchText, true, wm); sink += wm.lists.size(); }
    auto t4 = std::chrono::steady_clock::now();
    infx_cov_query cq;
    for (uint32_t i = 0; i < nq; i++) { const QueryPlan& P = plans[i]; if (P.blank || P.unsupported) continue; sink += (size_t)prepare_cov_query(ix, P.searchText, cq); }
    auto t5 = std::chrono::steady_clock::now();
    // the plan exchange: [5] the part of plan_tokens a peer can do (plan_tokens_text: no cache, no expansion), [6] importing a peer's plans (the whole batch as one slice)
    { QueryPlan tmp; for (uint32_t i = 0; i < nq; i++) { plan_tokens_text(ix, uview((const u16*)q_arena + q_offs[i], (size_t)(q_offs[i + 1] - q_offs[i])), depth, tmp); sink += tmp.rawTok.size(); } }
    auto t6 = std::chrono::steady_clock::now();
    auto t7 = t6, t8 = t6, t9 = t6;
    const bool dl = e->devLookups; e->devLookups = true;      // the blob of the deployment this hook is about: dictionaries on the device, the plan section only
    const int64_t blobBytes = e->def ? infx_session_prefetch_collect(e->def, nq, q_arena, q_offs, 0, nq, depth) : -1;
    e->devLookups = dl;
    if (blobBytes >= 0) {
        std::vector<uint8_t> blob = e->def->prefetchBlob;
        e->def->planPre.clear(); e->def->planPeer.clear();      // (a peer's slice never overlaps the own one)
        t7 = std::chrono::steady_clock::now();
        const int32_t rc = infx_session_prefetch_import(e->def, blob.data(), (int64_t)blob.size());
        t8 = std::chrono::steady_clock::now();
        if (!rc) {      // [7]: what phase 0 then pays per imported query instead of [5] + [4]: parsing the record into the plan and the coverage query
            QueryPlan tmp; infx_cov_query tc; bool hc, lg; int32_t ce; uint32_t co;
            for (auto& pp : e->def->planPre) if (pp && pp->rec && !parse_plan_record(ix, pp->rec, pp->recLen, depth, tmp, hc, ce, co, lg) && hc && !ce && !lg) sink += parse_cov_record(tmp.searchText, pp->rec, pp->recLen, co, tc) ? 1 : 0;
        }
        t9 = std::chrono::steady_clock::now();
        e->def->planPre.clear(); e->def->planPeer.clear(); e->def->wmPre.clear();
        if (rc) return rc;
    }
    auto us = [&](auto a, auto b) { return std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3 / std::max<uint32_t>(1, nq); };
    out_us[5] = us(t5, t6); out_us[6] = us(t7, t8); out_us[7] = us(t8, t9);
    out_us[0] = us(t0, t1); out_us[1] = fc.ld1Ns.load() / 1e3 / std::max<uint32_t>(1, nq); out_us[2] = us(t2, t3); out_us[3] = us(t3, t4); out_us[4] = us(t4, t5) + (sink == (size_t)-1 ? 1 : 0);
    return INFX_OK;
}
// WordMatcherLookup.Execute, fully enumerated (tests only): sorted unique ids
int64_t infx_engine_wordmatcher(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int64_t cap) {
    if (!e || len < 0 || (len && !q) || (cap > 0 && !out)) return -1;
    WmResult wm; ustr t = normalize(uview((const u16*)q, len)); lower_inplace(t);
    wm_collect(e->ix, t, true, wm);
    std::vector<int32_t> all;
    for (auto& l : wm.lists) all.insert(all.end(), l.p, l.p + l.n);
    std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
    for (size_t i = 0; i < all.size() && (int64_t)i < cap; i++) out[i] = all[i];
    return (int64_t)all.size();
}
// ---- the planning lookups as the DEVICE answers them (parity tests: tests/test_gpu_lookups.py) ----
int32_t infx_engine_device_lookups(infx_engine* e) { return !e ? -1 : (e->devLookups ? 1 : 0); }
int32_t infx_engine_lookup_stats(infx_engine* e, int64_t* out4) {      // words expanded on the device / on the host, queries whose WordMatcher lists came from the device / the host
    if (!e || !out4) return INFX_EINVAL;
    out4[0] = e->ld1OnDevice.load(); out4[1] = e->ld1OnHost.load(); out4[2] = e->wmOnDevice.load(); out4[3] = e->wmOnHost.load();
    return INFX_OK;
}
int32_t infx_engine_match_ld1_device(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap) {      // like infx_engine_match_ld1; -1 / -2: the kernel handed the word back (status 1 / 2)
    if (!e || len < 0 || (len && !q) || cap < 1 || !out) return -100;
    if (!e->devLookups || !e->def || !e->def->stream) { g_eerr = "no device dictionaries"; return -100; }
    uint32_t offs[2] = {0, (uint32_t)len}, count = 0, status = 0;
    std::vector<int32_t> mem((size_t)cap);
    int32_t rc = infx_ld1_expand(e->def->stream, 1, offs, q, (uint32_t)cap, mem.data(), &count, &status);
    if (rc) { g_eerr = infx_last_error(); return -100 - rc; }
    if (status) return -(int32_t)status;
    for (uint32_t k = 0; k < count && k < (uint32_t)cap; k++) out[k] = mem[k];
    return (int32_t)count;
}
int64_t infx_engine_wordmatcher_device(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int64_t cap) {      // like infx_engine_wordmatcher, lists resolved by k_wm
    if (!e || len < 0 || (len && !q)) return -1;
    if (!e->devLookups || !e->ix.cfg.wordMatcher || !e->def || !e->def->stream) { g_eerr = "no device dictionaries"; return -1; }
    const HostIndex& ix = e->ix;
    ustr st((const u16*)q, (size_t)len);
    if (!wm_on_device(ix, st)) { g_eerr = "query not admissible for the device lookup"; return -2; }
    infx_cov_query cq; if (prepare_cov_query(ix, st, cq)) { g_eerr = "query exceeds the Stage-2 envelope"; return -2; }
    size_t words = 0; for_each_word(st, [&](int, int l) { if (l >= 2) words++; });
    std::vector<infx_wm_list> lists(INFX_MAX_WM_LISTS); std::vector<int32_t> owned(words * 4096 + 1); uint32_t nl = 0;
    int32_t rc = infx_wm_lookup_debug(e->def->stream, &cq, lists.data(), &nl, owned.data(), (uint64_t)words * 4096);
    if (rc) { g_eerr = infx_last_error(); return -1; }
    std::vector<int32_t> all;
    for (uint32_t l = 0; l < nl; l++) {
        const infx_wm_list& L = lists[l];
        const int32_t* p = L.src == 0 ? ix.wmExact.doc.data() : (L.src == 1 ? ix.wmLd1.doc.data() : owned.data());
        const uint64_t lim = L.src == 0 ? ix.wmExact.doc.size() : (L.src == 1 ? ix.wmLd1.doc.size() : owned.size());
        if (L.off + L.len > lim) { g_eerr = "device list out of range"; return -3; }
        for (uint32_t i = 1; i < L.len; i++) if (p[L.off + i] <= p[L.off + i - 1]) { g_eerr = "device list not ascending"; return -4; }
        all.insert(all.end(), p + L.off, p + L.off + L.len);
    }
    std::sort(all.begin(), all.end()); all.erase(std::unique(all.begin(), all.end()), all.end());
    for (size_t i = 0; i < all.size() && (int64_t)i < cap; i++) out[i] = all[i];
    return (int64_t)all.size();
}
int32_t infx_engine_prefix_pop(infx_engine* e, const uint16_t* p, int32_t len) {
    if (!e || len < 0 || (len && !p)) return -1;
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
// CoverageEngine.PrepareQuery for one raw query text, exactly as the engine prepares it for infx_stage2_batch / infx_search_fused (a C# host
// would build the same struct from its CoverageQueryContext).  Returns INFX_OK, or the status prepare_cov_query reports (envelope).
int32_t infx_engine_prepare_cov_query(infx_engine* e, const uint16_t* q, int32_t len, infx_cov_query* out) {
    if (!e || !q || !out || len < 0) return efail(INFX_EINVAL, "bad arguments");
    QueryPlan P; plan_tokens(e->ix, e->fuzzy, uview((const u16*)q, (size_t)len), 500, P, true);
    if (P.blank || P.unsupported) return efail(INFX_EINVAL, "blank or unsupported query");
    std::memset(out, 0, sizeof *out);
    return prepare_cov_query(e->ix, P.searchText, *out);
}
int32_t infx_sizeof_cov_query(void) { return (int32_t)sizeof(infx_cov_query); }
// the same for a query beyond the fast envelope: the record of the long-query table (infx_stage2_long_queries)
int32_t infx_engine_prepare_cov_query_long(infx_engine* e, const uint16_t* q, int32_t len, infx_cov_query_long* out) {
    if (!e || !out || (len && !q)) return efail(INFX_EINVAL, "null argument");
    QueryPlan P; plan_tokens_text(e->ix, uview((const u16*)q, (size_t)len), 500, P);
    if (P.blank || P.unsupported) return efail(INFX_EINVAL, "blank or unsupported query");
    return prepare_cov_query_long(e->ix, P.searchText, *out);
}
int32_t infx_sizeof_cov_query_long(void) { return (int32_t)sizeof(infx_cov_query_long); }
int32_t infx_engine_effective_cpus(void) { return effective_cpus(); }
// parity tooling: switch the introspection downloads (Stage-1 rows, Stage-2 candidates / features of the last batch) on or off at run time
int32_t infx_engine_set_introspection(infx_engine* e, int32_t on) { if (!e) return INFX_EINVAL; e->cfg.want_features = on ? 1 : 0; return INFX_OK; }

int32_t infx_engine_normalize(const uint16_t* s, int32_t len, int32_t lower, uint16_t* out, int32_t cap) {
    ustr r = normalize(uview((const u16*)s, len)); if (lower) lower_inplace(r);
    std::memcpy(out, r.data(), (size_t)std::min<int>(cap, (int)r.size()) * 2);
    return (int32_t)r.size();
}
// raw access for bench.py (HBM-resident inputs are the engine's; these expose the flat host arrays for oracle adoption)
int32_t infx_engine_device_handles(infx_engine* e, infx_index** idx, infx_stream** st) { if (!e) return INFX_EINVAL; if (idx) *idx = e->dev; if (st) *st = e->def->stream; return INFX_OK; }


// ---- INFDX2 index files (host/infdx2.h) -------------------------------------------------------------------------------------------------------
// SearchEngine.Load (SearchEngine.cs:399-441) for files whose documents were indexed as single Med-weight fields: the documents are read and indexed
// (which uploads the shard as infx_engine_index_documents does), Deleted flags are applied, and every stored term is compared with the index just
// built.  *checked3 = {documents, stored terms compared, stored postings compared}.  INFX_EUNSUPPORTED: the stored postings are not what this
// builder produces for the stored texts (multi-field weights, other tokenizer settings); INFX_EINVAL: not an INFDX2 file / corrupted.
// ---- node-local cache of the host index (host/hostcache.h): one build per node instead of one per rank ------------------------------------
static uint64_t syn_hash(const SynMap& m) {
    uint64_t h = 1469598103934665603ull;
    for (auto& pr : m.parent) for (const ustr* t : {&pr.first, &pr.second}) { for (u16 c : *t) { h ^= c; h *= 1099511628211ull; } h ^= 0xFFFFu; h *= 1099511628211ull; }
    return h;
}
int64_t infx_engine_fuzzy_cache_size(infx_engine* e) { return e ? (int64_t)e->fuzzy.size() : -1; }
int32_t infx_engine_set_build_threads(infx_engine* e, int32_t threads) {
    if (!e || threads < 0) return efail(INFX_EINVAL, "bad arguments");
    e->buildThreads = threads; return INFX_OK;
}
int32_t infx_engine_save_host_index(infx_engine* e, const char* path) {
    if (!e || !path) return efail(INFX_EINVAL, "null argument");
    if (!e->indexed) return efail(INFX_EINVAL, "index the documents first");
    const std::string err = hostcache::save(path, e->ix, e->keysAreIds, hostcache::config_signature(e->ix.cfg, syn_hash(e->ix.cfg.syn)), index_fingerprint(e->ix));
    if (!err.empty()) return efail(INFX_EINVAL, err.c_str());
    return INFX_OK;
}
int32_t infx_engine_index_from_host_cache(infx_engine* e, const char* path) {
    if (!e || !path) return efail(INFX_EINVAL, "null argument");
    if (e->indexed) return efail(INFX_EINVAL, "this engine instance is already indexed (re-indexing: create a new engine)");
    bool keysAreIds = true;
    const std::string err = hostcache::load(path, e->ix, keysAreIds, hostcache::config_signature(e->ix.cfg, syn_hash(e->ix.cfg.syn)), &index_fingerprint);
    if (!err.empty()) return efail(INFX_EINVAL, err.c_str());
    e->keysAreIds = keysAreIds;
    return finish_index(e);
}

int32_t infx_engine_load_index(infx_engine* e, const char* path, int64_t* checked3) {
    if (!e || !path) return efail(INFX_EINVAL, "null argument");
    if (e->indexed) return efail(INFX_EINVAL, "this engine instance is already indexed");
    infdx2::File F;
    if (!infdx2::read_file(path, F)) return efail(INFX_EINVAL, F.error);
    const int64_t n = (int64_t)F.docs.size();
    std::vector<int64_t> keys((size_t)n); std::vector<uint64_t> offs((size_t)n + 1, 0); std::vector<uint16_t> arena;
    for (int64_t d = 0; d < n; d++) {
        if (F.docs[d].id != (int32_t)d) return efail(INFX_EUNSUPPORTED, "document ids of the file are not 0..n-1 in order (written after deletions): the stored postings refer to ids Load does not restore");
        keys[d] = F.docs[d].key; arena.insert(arena.end(), F.docs[d].text.begin(), F.docs[d].text.end()); offs[d + 1] = arena.size();
    }
    if (arena.empty()) arena.push_back(0);
    const int32_t w = 1;      // Weight.Med: ReadDocuments adds the text as the single field "content" with Weight.Med (IndexPersistence.cs:338-339)
    if (n > 0x7FFFFFF0ll) return efail(INFX_EINVAL, "too many documents");
    // build the host index, CHECK it against the file, and only then commit (upload, key map): a refused file leaves the engine unindexed and reusable
    {
        DocSource src{n, 1, &w, keys.data(), (const u16*)arena.data(), offs.data()};
        const int planThreads = e->ix.cfg.threads;
        if (e->buildThreads > 0) e->ix.cfg.threads = e->buildThreads;
        build_index(src, e->ix);
        e->ix.cfg.threads = planThreads;
    }
    const HostIndex& ix = e->ix;
    int64_t nterms = 0, npost = 0;
    const char* bad = nullptr;
    infdx2::LossyMatcher termOf(ix.terms.keys);          // the same text, or the lossy UTF-8 image of a term that holds half a surrogate pair (infdx2.h)
    for (auto& t : F.terms) {
        const int64_t id = termOf.match(t.text, [&](uint32_t k) { return ix.df[k] > 0; });
        if (id < 0) { bad = "a stored term does not exist in the index built from the stored documents"; break; }
        const uint64_t b = ix.terms.off[id], len = ix.terms.off[id + 1] - b;
        if (ix.df[id] != t.df || len != t.docs.size()) { bad = "a stored term's document frequency / posting count differs from the rebuilt index"; break; }
        for (size_t i = 0; i < t.docs.size() && !bad; i++)
            if (ix.terms.doc[b + i] != t.docs[i] || ix.terms.w[b + i] != t.w[i]) bad = "a stored posting (document, weight) differs from the rebuilt index: the file was not written from single Med-weight fields";
        if (bad) break;
        nterms++; npost += (int64_t)t.docs.size();
    }
    if (!bad) {   // ... and the other direction: the rebuilt index may not hold non-stop terms the file lacks (e.g. a file written with another stop-term limit)
        int64_t built = 0; for (size_t id = 0; id < ix.terms.K(); id++) if (ix.df[id] > 0) built++;
        if (built != nterms) bad = "the index rebuilt from the stored documents holds another number of non-stop terms than the file";
    }
    if (!bad) bad = infdx2::check_derived(F, ix);
    if (bad) { const HostConfig keep = e->ix.cfg; e->ix = HostIndex{}; e->ix.cfg = keep; return efail(INFX_EUNSUPPORTED, bad); }
    e->keysAreIds = false;
    int32_t rc = finish_index(e);
    if (rc) return rc;
    std::vector<int64_t> gone; for (auto& d : F.docs) if (d.deleted) gone.push_back(d.key);
    if (!gone.empty()) { rc = infx_engine_delete_documents(e, gone.data(), (int64_t)gone.size(), nullptr); if (rc) return rc; }
    if (checked3) { checked3[0] = n; checked3[1] = nterms; checked3[2] = npost; }
    return INFX_OK;
}

// ---- INFS segment files (host/infs.h): SearchEngine.Flush's on-disk segments -------------------------------------------------------------------
struct infx_segment { infs::Segment S; };
int32_t infx_segment_open(const char* path, infx_segment** out) {
    if (!path || !out) return efail(INFX_EINVAL, "null argument");
    infx_segment* g = new infx_segment();
    if (!infs::read_file(path, g->S)) { const std::string m = g->S.error; delete g; return efail(INFX_EINVAL, "INFS segment: " + m); }
    *out = g; return INFX_OK;
}
void infx_segment_close(infx_segment* g) { delete g; }
int32_t infx_segment_info(infx_segment* g, int32_t* doc_count, int32_t* num_terms, int64_t* num_postings, int64_t* term_chars) {
    if (!g) return efail(INFX_EINVAL, "null segment");
    if (doc_count) *doc_count = g->S.docCount; if (num_terms) *num_terms = (int32_t)g->S.terms.size(); if (num_postings) *num_postings = (int64_t)g->S.doc.size();
    if (term_chars) { int64_t c = 0; for (auto& t : g->S.terms) c += (int64_t)t.size(); *term_chars = c; }
    return INFX_OK;
}
int32_t infx_segment_export(infx_segment* g, uint32_t* term_offs, uint16_t* term_chars, uint64_t* post_offs, int32_t* doc_ids, uint8_t* weights) {
    if (!g) return efail(INFX_EINVAL, "null segment");
    const infs::Segment& S = g->S; const size_t T = S.terms.size();
    if (term_offs) { uint32_t o = 0; for (size_t t = 0; t < T; t++) { term_offs[t] = o; if (term_chars) std::memcpy(term_chars + o, S.terms[t].data(), S.terms[t].size() * 2); o += (uint32_t)S.terms[t].size(); } term_offs[T] = o; }
    if (post_offs) std::memcpy(post_offs, S.off.data(), (T + 1) * 8);
    if (doc_ids && !S.doc.empty()) std::memcpy(doc_ids, S.doc.data(), S.doc.size() * 4);
    if (weights && !S.w.empty()) std::memcpy(weights, S.w.data(), S.w.size());
    return INFX_OK;
}
int32_t infx_engine_verify_segment(infx_engine* e, const char* path, int32_t doc_base, int64_t* checked3) {
    if (!e || !path || doc_base < 0) return efail(INFX_EINVAL, "bad arguments");
    if (!e->indexed) return efail(INFX_EINVAL, "index the documents first: a segment holds postings, not documents");
    infs::Segment S;
    if (!infs::read_file(path, S)) return efail(INFX_EINVAL, "INFS segment: " + S.error);
    const HostIndex& ix = e->ix;
    if ((int64_t)doc_base + S.docCount > ix.N) return efail(INFX_EUNSUPPORTED, "the segment's documents [doc_base, doc_base + docCount) lie outside the indexed corpus");
    const int32_t lo = doc_base, hi = doc_base + S.docCount;
    int64_t npost = 0;
    for (size_t t = 0; t < S.terms.size(); t++) {
        const int64_t id = ix.terms.keys.find(uview((const u16*)S.terms[t].data(), S.terms[t].size()));
        if (id < 0) return efail(INFX_EUNSUPPORTED, "a term of the segment does not exist in the index built from the documents");
        const int32_t* p = ix.terms.doc.data(); const uint64_t b = ix.terms.off[id], z = ix.terms.off[id + 1];
        const uint64_t a = std::lower_bound(p + b, p + z, lo) - p, c = std::lower_bound(p + a, p + z, hi) - p;
        const uint64_t n = S.off[t + 1] - S.off[t];
        if (c - a != n) return efail(INFX_EUNSUPPORTED, "a term's posting count inside the segment's document range differs from the index");
        for (uint64_t i = 0; i < n; i++)
            if (p[a + i] != S.doc[S.off[t] + i] + doc_base || ix.terms.w[a + i] != S.w[S.off[t] + i]) return efail(INFX_EUNSUPPORTED, "a posting (document, weight) of the segment differs from the index");
        npost += (int64_t)n;
    }
    // the other direction: every index term with postings in the range must be in the segment (SegmentWriter keeps the terms with DocumentFrequency > 0)
    int64_t present = 0;
    for (size_t id = 0; id < ix.terms.K(); id++) {
        if (ix.df[id] <= 0) continue;
        const int32_t* p = ix.terms.doc.data(); const uint64_t b = ix.terms.off[id], z = ix.terms.off[id + 1];
        const uint64_t a = std::lower_bound(p + b, p + z, lo) - p;
        if (a < z && p[a] < hi) present++;
    }
    if (present != (int64_t)S.terms.size()) return efail(INFX_EUNSUPPORTED, "the index holds terms with postings in the segment's document range that the segment lacks");
    if (checked3) { checked3[0] = S.docCount; checked3[1] = (int64_t)S.terms.size(); checked3[2] = npost; }
    return INFX_OK;
}

// ---- Document.Deleted -----------------------------------------------------------------------------------------------------------------
// DocumentCollection.DeleteDocumentsByKey (Core/DocumentCollection.cs:200-212): every document (alias / segment) with one of the keys is marked
// deleted.  The index itself is not touched — like the reference between a deletion and the next re-index, df / doc lengths / avgdl still
// include the document; the query path skips it (include/infidex_hip.h, infx_set_deleted).  Exclusive: no search may be in flight.
int32_t infx_engine_delete_documents(infx_engine* e, const int64_t* keys, int64_t n, int64_t* out_marked) {
    if (!e || (n > 0 && !keys)) return efail(INFX_EINVAL, "null argument");
    if (!e->indexed) return efail(INFX_EINVAL, "delete before index_documents");
    const int64_t N = e->ix.N; int64_t marked = 0;
    if (e->deleted.empty()) e->deleted.assign((size_t)N, 0);
    if (e->keysAreIds) { for (int64_t i = 0; i < n; i++) if (keys[i] >= 0 && keys[i] < N && !e->deleted[(size_t)keys[i]]) { e->deleted[(size_t)keys[i]] = 1; marked++; } }
    else {
        std::unordered_set<int64_t> ks(keys, keys + n);
        for (int64_t d = 0; d < N; d++) if (!e->deleted[(size_t)d] && ks.count(e->ix.docKey[(size_t)d])) { e->deleted[(size_t)d] = 1; marked++; }
    }
    if (out_marked) *out_marked = marked;
    if (e->dev) { int32_t rc = infx_set_deleted(e->dev, (uint32_t)N, e->deleted.data()); if (rc) { g_eerr = infx_last_error(); return rc; } }
    e->invalidate_filter_counts();
    return INFX_OK;
}
// Clears every Deleted flag (what a reload of the undeleted documents would give).
int32_t infx_engine_restore_documents(infx_engine* e) {
    if (!e) return efail(INFX_EINVAL, "null argument");
    e->deleted.clear();
    if (e->dev && e->indexed) { int32_t rc = infx_set_deleted(e->dev, 0, nullptr); if (rc) { g_eerr = infx_last_error(); return rc; } }
    e->invalidate_filter_counts();
    return INFX_OK;
}

// ---- Infiscript post-filter + facets (config 5): Query.Filter / Query.EnableFacets (SearchEngine.cs:298-316) ------------------------------
// One non-indexed document field for all documents (DocumentFields / Field.Value), by internal id: kind 1 int64, 2 double, 3 UTF-8 strings
// (arena + n+1 offsets).  facetable = Field.Facetable.  After infx_engine_index_documents.
int32_t infx_engine_add_column(infx_engine* e, const char* name, int32_t kind, int32_t facetable, int64_t n, const int64_t* vi, const double* vd,
                               const char* arena, const uint64_t* offs) {
    if (!e || !name || n < 0 || (kind == 1 && !vi) || (kind == 2 && !vd) || (kind == 3 && (!arena || !offs)) || kind < 1 || kind > 3) return efail(INFX_EINVAL, "bad column arguments");
    if (!e->indexed || n != e->ix.N) return efail(INFX_EINVAL, "a column needs one value per indexed document");
    if (e->columns.size() >= 64) return efail(INFX_ECAPACITY, "too many columns");
    for (auto& c : e->columns) if (c.name == name) return efail(INFX_EINVAL, "column exists");
    filt::Column c; c.name = name; c.facetable = facetable != 0;
    if (kind == 1) filt::encode_column(c, (size_t)n, [&](size_t d) { filt::Boxed b; b.kind = 1; b.i = vi[d]; return b; }, (long long)0);
    else if (kind == 2) filt::encode_column(c, (size_t)n, [&](size_t d) { filt::Boxed b; b.kind = 2; b.d = vd[d]; return b; }, (uint64_t)0);
    else filt::encode_column(c, (size_t)n, [&](size_t d) { filt::Boxed b; b.kind = 3; b.s.assign(arena + offs[d], arena + offs[d + 1]); return b; }, std::string());
    if (e->dev) {
        int32_t rc = infx_upload_column(e->dev, (uint32_t)e->columns.size(), (uint32_t)n, c.codes.data(), (uint32_t)c.dict.size());
        if (rc) { g_eerr = infx_last_error(); return rc; }
    }
    e->columns.push_back(std::move(c));
    e->retire_filters();      // leaf tables of filters compiled before this field existed treat it as null: compile again on next use
    return INFX_OK;
}
int32_t infx_engine_column_count(infx_engine* e) { return e ? (int32_t)e->columns.size() : -1; }
int32_t infx_engine_column_info(infx_engine* e, int32_t col, char* name, int32_t cap, int32_t* facetable, int32_t* num_values) {
    if (!e || col < 0 || col >= (int32_t)e->columns.size()) return INFX_EINVAL;
    const filt::Column& c = e->columns[col];
    if (name && cap > 0) snprintf(name, (size_t)cap, "%s", c.name.c_str());
    if (facetable) *facetable = c.facetable ? 1 : 0; if (num_values) *num_values = (int32_t)c.dict.size();
    return INFX_OK;
}
int32_t infx_engine_column_value(infx_engine* e, int32_t col, uint32_t code, char* out, int32_t cap) {      // ToString() of a distinct value (facet key)
    if (!e || col < 0 || col >= (int32_t)e->columns.size() || code >= e->columns[col].text.size()) return -1;
    const std::string& t = e->columns[col].text[code];
    if (out && cap > 0) snprintf(out, (size_t)cap, "%s", t.c_str());
    return (int32_t)t.size();
}
// Installs Query.Filter (expr, UTF-8; NULL = none) and Query.EnableFacets on the session: every following search on it post-filters its rows
// on the device and counts the facetable fields.  n_in_filter = Filter.NumberOfDocumentsInFilter (this shard's share when sharded),
// counted on the device the first time the expression is used.  Status: INFX_EINVAL + message for a syntax error (FilterParseException),
// INFX_EUNSUPPORTED for MATCHES.
int32_t infx_engine_set_filter(infx_session* S, const char* expr, int32_t enable_facets, uint32_t* n_in_filter) {
    if (!S) return efail(INFX_EINVAL, "null session");
    infx_engine* e = S->e;
    if (!e->dev || !S->stream) return efail(INFX_EHIP, "no GPU: the post-filter runs on the device");
    infx_filter* dev = nullptr; uint32_t cnt = 0;
    if (expr) {
        std::lock_guard<std::mutex> lk(e->filterMu);
        auto it = e->filters.find(expr);
        if (it == e->filters.end()) {
            filt::Program P;
            try { P = filt::parse(expr); }
            catch (const filt::Unsupported& x) { return efail(INFX_EUNSUPPORTED, x.what()); }
            catch (const filt::SyntaxError& x) { return efail(INFX_EINVAL, std::string("filter syntax error: ") + x.what()); }
            std::vector<infx_filter_op> ops; std::vector<infx_filter_leaf> leaves; std::vector<uint32_t> tables, w;
            for (auto& in : P.code) ops.push_back(infx_filter_op{in.op, in.arg});
            for (auto& L : P.leaves) {
                int ci = -1; for (size_t c = 0; c < e->columns.size(); c++) if (e->columns[c].name == L.field) ci = (int)c;      // field names are case sensitive (Dictionary<string, Field>)
                filt::leaf_table(L, ci >= 0 ? &e->columns[ci] : nullptr, w);
                leaves.push_back(infx_filter_leaf{ci >= 0 ? (uint32_t)ci : 0xFFFFFFFFu, (uint32_t)tables.size(), ci >= 0 ? (uint32_t)e->columns[ci].dict.size() : 1u, 0});
                tables.insert(tables.end(), w.begin(), w.end());
            }
            CompiledFilter cf;
            int32_t rc = infx_filter_create(e->dev, (uint32_t)ops.size(), ops.data(), (uint32_t)leaves.size(), leaves.data(), (uint32_t)tables.size(), tables.data(), &cf.dev);
            if (rc) { g_eerr = infx_last_error(); return rc; }
            it = e->filters.emplace(expr, cf).first;
        }
        if (!it->second.counted) {       // ResultProcessor.cs:39-54: first use runs the filter over every document
            int32_t rc = infx_filter_count(S->stream, it->second.dev, &it->second.inFilter);
            if (rc) { g_eerr = infx_last_error(); return rc; }
            it->second.counted = true;
        }
        dev = it->second.dev; cnt = it->second.inFilter;
    }
    S->facetCols.clear();
    if (enable_facets) for (size_t c = 0; c < e->columns.size() && S->facetCols.size() < INFX_MAX_FACET_COLS; c++) if (e->columns[c].facetable) S->facetCols.push_back((uint32_t)c);
    int32_t rc = infx_stream_set_postfilter(S->stream, dev, (uint32_t)S->facetCols.size(), S->facetCols.data());
    if (rc) { g_eerr = infx_last_error(); return rc; }
    if (n_in_filter) *n_in_filter = cnt;
    return INFX_OK;
}
// Facets of query qi of the session's last search: for the k-th facetable column (engine column index in *col), up to cap (code, count) pairs ordered
// as FacetBuilder does (count descending, value ascending), at most 100.  Returns the number of pairs, -1 on error.
int32_t infx_engine_last_facets(infx_session* S, uint32_t nq, uint32_t qi, uint32_t k, int32_t* col, uint32_t* codes, uint32_t* counts, int32_t cap) {
    if (!S || !S->stream || qi >= nq || k >= S->facetCols.size()) return -1;
    const uint32_t nf = (uint32_t)S->facetCols.size();
    std::vector<uint32_t> cd((size_t)nq * nf * INFX_FILTER_MAX_ROWS), ct(cd.size()), nn((size_t)nq * nf);
    if (infx_last_facets(S->stream, nq, cd.data(), ct.data(), nn.data())) { g_eerr = infx_last_error(); return -1; }
    const filt::Column& C = S->e->columns[S->facetCols[k]];
    const size_t o = ((size_t)qi * nf + k) * INFX_FILTER_MAX_ROWS; const uint32_t n = nn[(size_t)qi * nf + k];
    std::vector<std::pair<uint32_t, uint32_t>> v;
    for (uint32_t i = 0; i < n; i++) if (cd[o + i] < C.text.size() && !C.text[cd[o + i]].empty()) v.push_back({cd[o + i], ct[o + i]});     // empty strings are not facet values (:95-99)
    std::sort(v.begin(), v.end(), [&](auto& a, auto& b) { if (a.second != b.second) return a.second > b.second; return C.rank[a.first] < C.rank[b.first]; });
    if (v.size() > 100) v.resize(100);
    if (col) *col = (int32_t)S->facetCols[k];
    int32_t m = 0; for (auto& x : v) { if (m >= cap) break; codes[m] = x.first; counts[m] = x.second; m++; }
    return m;
}
int32_t infx_engine_facet_column_count(infx_session* S) { return S ? (int32_t)S->facetCols.size() : -1; }

} // extern "C"
