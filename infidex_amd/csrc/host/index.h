// Host-side index of the product: a parallel builder that produces the flat arrays the C ABI uploads
// (include/infidex_hip.h), plus the host-resident lookup structures query preparation needs.
//
// Mirrors the OBSERVABLE result of the reference's single-threaded indexer (it is not a transliteration: documents are
// tokenised in parallel chunks, every inverted structure goes through one generic thread-local -> global CSR merge,
// and the order-dependent quantities — term ids in first-appearance order, byte tf with banker's rounding, the df /
// stop-term counter — are reconstructed exactly from per-(term,doc) summaries):
//   SearchEngine.IndexDocumentsInternal           SearchEngine.cs:124-192
//   VectorModel.IndexDocument / BuildInvertedLists VectorModel.cs:73-220
//   Term.FirstCycleAdd / IncrementTermUsageCounter Core/Term.cs:71-146   (quirks Q3, Q5)
//   TermCollection.CountTermUsage                  Core/TermCollection.cs:75-139
//   PositionalPrefixIndex.IndexDocument / DocSet   Indexing/ShortQuery/PositionalPrefixIndex.cs:55-120, PrefixPosting.cs:109-137
//   WordMatcher.Load / FinalizeIndex               WordMatcher/WordMatcher.cs:82-164 (Q13: affix keeps the last doc)
//   VectorModel.BuildWordIdfCache                  VectorModel.cs:864-908
#pragma once
#include "text.h"
#include <thread>
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <cstdlib>
#include <cstdio>
#include <sched.h>
#include <unordered_map>
#include <string>
#include <memory>
#include <condition_variable>
#include <mutex>

namespace infx {

// CPUs this process may actually use: hardware threads, capped by the affinity mask and the cgroup CPU quota (a container with
// cpu.max = 16 CPUs on a 256-thread host is throttled for the rest of each 100 ms period once 64 busy threads have burnt the quota —
// measured as 40-70 ms process-wide stalls — so parallel regions are sized to the quota, not to the core count).
inline int effective_cpus() {
    static const int n = [] {
        int hc = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { int c = CPU_COUNT(&set); if (c > 0) hc = std::min(hc, c); }
        auto read2 = [](const char* path, long long& a, long long& b) { FILE* f = fopen(path, "r"); if (!f) return false; char buf[64] = {0}; bool ok = false;
            if (fgets(buf, sizeof buf, f)) { if (strncmp(buf, "max", 3) == 0) { a = -1; ok = true; } else ok = sscanf(buf, "%lld %lld", &a, &b) >= 1; } fclose(f); return ok; };
        long long q = -1, per = 100000;
        if (read2("/sys/fs/cgroup/cpu.max", q, per)) { /* cgroup v2 */ }
        else { long long d = 0; if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, d)) { long long pp = 0; if (read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", pp, d) && pp > 0) per = pp; } }
        if (q > 0 && per > 0) hc = std::min<long long>(hc, std::max<long long>(1, (q + per - 1) / per));
        if (const char* e = getenv("INFX_THREADS")) { int v = atoi(e); if (v > 0) hc = v; }
        return std::max(1, hc);
    }();
    return n;
}

// ---- host worker pool ------------------------------------------------------------------------------------------------
// One process-wide pool (hardware threads - 1 workers; the calling thread always takes part).  A parallel region is a shared
// chunk counter; idle workers join any region that still has chunks and a free slot, so several sessions preparing batches at
// the same time share the cores instead of each spawning its own threads.
struct Pool {
    struct Region {
        std::function<void(int64_t, int)> run;   // (chunk, slot)
        int64_t nchunks = 0; int maxSlots = 1;
        std::atomic<int64_t> next{0};
        int slots = 1, working = 1;              // guarded by Pool::m
        std::condition_variable done;
    };
    std::mutex m; std::condition_variable cv; std::vector<std::shared_ptr<Region>> regions; int nworkers = 0;
    static Pool& get() { static Pool* p = new Pool(); return *p; }    // never destroyed: workers are detached
    static bool& in_worker() { static thread_local bool w = false; return w; }
    Pool() {
        int hc = effective_cpus(); nworkers = hc > 1 ? hc - 1 : 0;
        for (int i = 0; i < nworkers; i++) std::thread([this] { worker(); }).detach();
    }
    void worker() {
        in_worker() = true;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            std::shared_ptr<Region> r; int slot = 0;
            for (auto& x : regions) if (x->slots < x->maxSlots && x->next.load(std::memory_order_relaxed) < x->nchunks) { r = x; slot = x->slots++; x->working++; break; }
            if (!r) { cv.wait(lk); continue; }
            lk.unlock();
            for (;;) { int64_t c = r->next.fetch_add(1); if (c >= r->nchunks) break; r->run(c, slot); }
            lk.lock();
            if (--r->working == 0) r->done.notify_all();
        }
    }
    void run(int64_t nchunks, int maxSlots, std::function<void(int64_t, int)> fn) {
        if (nchunks <= 0) return;
        if (nchunks == 1 || maxSlots <= 1 || nworkers == 0 || in_worker()) { for (int64_t c = 0; c < nchunks; c++) fn(c, 0); return; }
        auto r = std::make_shared<Region>(); r->run = std::move(fn); r->nchunks = nchunks; r->maxSlots = maxSlots;
        { std::lock_guard<std::mutex> g(m); regions.push_back(r); }
        cv.notify_all();
        for (;;) { int64_t c = r->next.fetch_add(1); if (c >= nchunks) break; r->run(c, 0); }
        std::unique_lock<std::mutex> lk(m);
        r->working--;
        r->done.wait(lk, [&] { return r->working == 0; });
        for (size_t i = 0; i < regions.size(); i++) if (regions[i] == r) { regions.erase(regions.begin() + i); break; }
    }
};

template <class F> inline void parallel_for(int64_t n, int threads, F&& f) {   // f(begin, end, partIndex): `threads` equal static parts
    if (threads <= 1 || n < 2) { f((int64_t)0, n, 0); return; }
    const int64_t per = (n + threads - 1) / threads;
    const int64_t parts = (n + per - 1) / per;
    Pool::get().run(parts, threads, [&](int64_t c, int) { f(c * per, std::min(n, (c + 1) * per), (int)c); });
}

// dynamic scheduling: items are handed out in small grains so that a few expensive items (fuzzy expansions, dense
// WordMatcher words) do not serialise behind one thread
template <class F> inline void parallel_dyn(int64_t n, int threads, int64_t grain, F&& f) {   // f(begin, end, slot), slot < threads
    if (threads <= 1 || n <= grain) { f((int64_t)0, n, 0); return; }
    Pool::get().run((n + grain - 1) / grain, threads, [&](int64_t c, int slot) { f(c * grain, std::min(n, (c + 1) * grain), slot); });
}

// ---- string -> dense id table (open addressing, keys in an arena) -----------------------------------------------
struct KeyTable {
    std::vector<u16> arena;
    std::vector<uint32_t> keyOff;   // id -> arena offset
    std::vector<uint16_t> keyLen;
    std::vector<uint64_t> slotHash; // 0 = empty
    std::vector<uint32_t> slotId;
    uint64_t mask = 0;
    KeyTable() { rehash(1 << 10); }
    size_t size() const { return keyOff.size(); }
    uview key(uint32_t id) const { return uview(arena.data() + keyOff[id], keyLen[id]); }
    void rehash(size_t cap) {
        std::vector<uint64_t> h(cap, 0); std::vector<uint32_t> ids(cap, 0);
        uint64_t m = cap - 1;
        for (size_t i = 0; i < slotHash.size(); i++) if (slotHash[i]) {
            uint64_t p = slotHash[i] & m;
            while (h[p]) p = (p + 1) & m;
            h[p] = slotHash[i]; ids[p] = slotId[i];
        }
        slotHash.swap(h); slotId.swap(ids); mask = m;
    }
    int64_t find(uview s) const {
        uint64_t h = hash_u16(s.data(), s.size()) | 1ull;
        uint64_t p = h & mask;
        while (slotHash[p]) {
            if (slotHash[p] == h) { uint32_t id = slotId[p]; if (keyLen[id] == s.size() && std::memcmp(arena.data() + keyOff[id], s.data(), s.size() * 2) == 0) return id; }
            p = (p + 1) & mask;
        }
        return -1;
    }
    uint32_t get_or_add(uview s, bool* isNew = nullptr) {
        uint64_t h = hash_u16(s.data(), s.size()) | 1ull;
        uint64_t p = h & mask;
        while (slotHash[p]) {
            if (slotHash[p] == h) { uint32_t id = slotId[p]; if (keyLen[id] == s.size() && std::memcmp(arena.data() + keyOff[id], s.data(), s.size() * 2) == 0) { if (isNew) *isNew = false; return id; } }
            p = (p + 1) & mask;
        }
        uint32_t id = (uint32_t)keyOff.size();
        keyOff.push_back((uint32_t)arena.size()); keyLen.push_back((uint16_t)s.size());
        arena.insert(arena.end(), s.begin(), s.end());
        slotHash[p] = h; slotId[p] = id;
        if (isNew) *isNew = true;
        if (keyOff.size() * 2 > slotHash.size()) rehash(slotHash.size() * 2);
        return id;
    }
};

// ---- one inverted structure: key -> ascending doc list (+ optional byte weight / df meta) --------------------------
struct Csr {
    KeyTable keys;                 // global ids in first-appearance order
    std::vector<uint64_t> off;     // K+1
    std::vector<int32_t> doc;
    std::vector<uint8_t> w;        // main index only
    std::vector<uint16_t> meta;    // main index only, dropped after the df pass: (net-1)<<8 | (maxTransient-1)
    size_t K() const { return keys.size(); }
    uint64_t len(uint32_t k) const { return off[k + 1] - off[k]; }
};

struct LocalInv {   // thread-local staging of one inverted structure
    KeyTable keys;
    std::vector<uint32_t> occKey; std::vector<int32_t> occDoc; std::vector<uint8_t> occW; std::vector<uint16_t> occMeta;
    std::vector<uint32_t> count;   // per local key
    bool weighted = false;
    void add(uint32_t lid, int32_t doc, uint8_t w = 0, uint16_t meta = 0) {
        occKey.push_back(lid); occDoc.push_back(doc);
        if (weighted) { occW.push_back(w); occMeta.push_back(meta); }
        if (lid >= count.size()) count.resize(lid + 1, 0);
        count[lid]++;
    }
};

inline void merge_inv(std::vector<LocalInv>& locals, Csr& out, int threads) {
    // Phase B: global ids in (thread order, local first-appearance order) == corpus first-appearance order
    std::vector<std::vector<uint32_t>> l2g(locals.size());
    for (size_t t = 0; t < locals.size(); t++) {
        auto& L = locals[t];
        l2g[t].resize(L.keys.size());
        for (uint32_t i = 0; i < L.keys.size(); i++) l2g[t][i] = out.keys.get_or_add(L.keys.key(i));
    }
    size_t K = out.keys.size();
    out.off.assign(K + 1, 0);
    for (size_t t = 0; t < locals.size(); t++) { auto& L = locals[t]; L.count.resize(L.keys.size(), 0); for (uint32_t i = 0; i < L.keys.size(); i++) out.off[l2g[t][i] + 1] += L.count[i]; }
    for (size_t k = 0; k < K; k++) out.off[k + 1] += out.off[k];
    std::vector<uint64_t> cursor(out.off.begin(), out.off.end() - 1);
    std::vector<std::vector<uint64_t>> start(locals.size());
    for (size_t t = 0; t < locals.size(); t++) {
        auto& L = locals[t]; start[t].resize(L.keys.size());
        for (uint32_t i = 0; i < L.keys.size(); i++) { uint32_t g = l2g[t][i]; start[t][i] = cursor[g]; cursor[g] += L.count[i]; }
    }
    bool weighted = !locals.empty() && locals[0].weighted;
    out.doc.resize(out.off[K]);
    if (weighted) { out.w.resize(out.off[K]); out.meta.resize(out.off[K]); }
    // Phase C: parallel fill, each thread writes its own (disjoint) positions
    parallel_for((int64_t)locals.size(), threads, [&](int64_t b, int64_t e, int) {
        for (int64_t t = b; t < e; t++) {
            auto& L = locals[t]; auto& st = start[t];
            for (size_t i = 0; i < L.occKey.size(); i++) {
                uint64_t p = st[L.occKey[i]]++;
                out.doc[p] = L.occDoc[i];
                if (weighted) { out.w[p] = L.occW[i]; out.meta[p] = L.occMeta[i]; }
            }
            std::vector<uint32_t>().swap(L.occKey); std::vector<int32_t>().swap(L.occDoc);
            std::vector<uint8_t>().swap(L.occW); std::vector<uint16_t>().swap(L.occMeta);
        }
    });
}

struct DocInput { const u16* text; uint32_t len; int32_t weight; };   // one field; weight 0 High / 1 Med / 2 Low

struct HostConfig {
    int ngram = 3, startPad = 2, stopPad = 0, stopTermLimit = 1250000;
    float fieldWeights[3] = {1.5f, 1.25f, 1.0f};
    bool enableCoverage = true, wordMatcher = true;
    int wmMinExact = 2, wmMaxExact = 8, wmMinLD1 = 3, wmMaxLD1 = 8;
    int maxDepth = 500;            // prefix DocSets with population > 20*maxDepth are never acceptable (only counted)
    int threads = 0;
    SynMap syn;                    // SearchEngine(..., synonymMap): empty by default
};

struct HostIndex {
    HostConfig cfg;
    int32_t N = 0;
    std::vector<int64_t> docKey;
    std::vector<float> docLen; float avgdl = 0.f;
    std::vector<uint64_t> textOff; std::vector<u16> text;   // Stage-2 text: lower(normalize(IndexedText))
    Csr terms;                      // main index; terms.keys id == reference termId
    std::vector<int32_t> df;        // -1 stop term
    std::vector<uint32_t> sortedTerms;   // ordinal string order
    // trie over all terms (label-sorted children; built from sortedTerms)
    struct TrieNode { u16 label; uint8_t pad0 = 0, pad1 = 0; int32_t term; uint32_t firstChild, nextSibling; };
    std::vector<TrieNode> trie;
    // contiguous, label-sorted out-edges of every node (scanning a node's labels touches 1-3 cache lines, not one per child)
    std::vector<uint32_t> edgeStart; std::vector<u16> edgeLabel; std::vector<uint32_t> edgeChild;
    // trie over the REVERSED terms (suffix trie), CSR edges only: LD1 matching walks it anchored at the term end (query.h match_ld1)
    std::vector<uint32_t> rEdgeStart; std::vector<u16> rEdgeLabel; std::vector<uint32_t> rEdgeChild; std::vector<int32_t> rTerm;
    // prefix DocSets
    Csr prefixAll;                  // temp
    KeyTable prefixKeys; std::vector<uint32_t> prefixPop; std::vector<int32_t> prefixSetId;   // per prefix key: population, uploaded set id or -1
    std::vector<uint64_t> psOff; std::vector<int32_t> psDocs;                                   // uploaded sets (pop <= 20*maxDepth)
    // WordMatcher
    Csr wmExact, wmLd1;
    std::vector<uint32_t> affixFwd, affixRev;     // word ids (of `words`) sorted by word / by reversed word, len >= wmMinLD1
    // words: doc frequency (word-level IDF) and last doc (affix)
    KeyTable words; std::vector<uint32_t> wordDf; std::vector<int32_t> wordLastDoc; std::vector<float> wordIdf;
    // WordIdfCache is an OrdinalIgnoreCase dictionary (VectorModel.cs:864-908): words that differ only in alias characters (final sigma / sigma ...) pool their
    // documents.  Classes with at least one alias member (rare: none in most corpora), keyed by the class representative text; 0 = no entry (df outside (0, N])
    KeyTable icWords; std::vector<float> icWordIdf;
    std::vector<int64_t> keyToFirst;   // optional reverse map is built lazily by the engine
};

inline float compute_idf(int totalDocs, int df) {   // Bm25Scorer.ComputeIdf, Bm25Scorer.cs:686-695
    if (df <= 0 || totalDocs <= 0) return 0.f;
    float d = (float)df, N = (float)totalDocs;
    float ratio = (N - d + 0.5f) / (d + 0.5f);
    return ratio <= 0.f ? 0.f : logf(ratio + 1.f);
}

struct DocSource {   // n docs, fieldCount fields each; offs has n*fieldCount+1 entries into arena
    int64_t n; int fieldCount; const int32_t* fieldWeights; const int64_t* keys; const u16* arena; const uint64_t* offs;
};

// Postings that come from flushed segment files (SearchEngine.Flush -> Indexing/Segments/SegmentWriter.cs; read and validated by host/infs.h) instead of
// being accumulated from the documents: segment i holds the postings of documents [docBase, docBase + docCount) with ids relative to docBase, terms in
// ordinal order.  The segments cover [0, flushed) without gaps, in order (what a sequence of Flush calls leaves: VectorModel.cs:804-815); the documents
// behind them are the live tail.
struct SegmentPostings { int32_t docBase = 0, docCount = 0; const std::vector<ustr>* terms = nullptr; const std::vector<uint64_t>* off = nullptr; const std::vector<int32_t>* doc = nullptr; const std::vector<uint8_t>* w = nullptr; };
// ix.terms is the index accumulated from ALL documents (so the df counter's double counts on saturated weight bytes — Term.cs:118-146, quirk Q5 — and the
// stop-term decisions are those of an unflushed index: a segment stores neither, and its writer drops the lists of terms that were stop terms at flush time).
// The segments' postings then REPLACE the accumulated (document, weight) pairs of their document ranges, list by list: a list of a segment must hold exactly the
// documents the accumulation found for that term in the segment's range (else the segment was written from other documents); a term without a list in a segment
// keeps its accumulated postings (a stop term at flush time stays one: its postings are dropped with the others after the df pass).  Round 5 accumulated only the
// live tail and spliced the segment lists in front: df and stop state of the flushed range were then the posting counts.
inline const char* splice_segments(HostIndex& ix, const std::vector<SegmentPostings>& segs, int threads) {
    Csr& C = ix.terms;
    std::atomic<int> bad{0};
    for (size_t s = 0; s < segs.size(); s++) {
        const auto& names = *segs[s].terms; const auto& o = *segs[s].off; const auto& d = *segs[s].doc; const auto& w = *segs[s].w;
        const int32_t base = segs[s].docBase, cnt = segs[s].docCount;
        parallel_for((int64_t)names.size(), threads, [&](int64_t b, int64_t e, int) {
            for (int64_t t = b; t < e; t++) {
                const int64_t id = C.keys.find(uview(names[t].data(), names[t].size()));
                if (id < 0) { bad.store(1); continue; }
                const int32_t* lo = C.doc.data() + C.off[id]; const int32_t* hi = C.doc.data() + C.off[id + 1];
                const int32_t* a = std::lower_bound(lo, hi, base); const int32_t* z = std::lower_bound(a, hi, base + cnt);
                if ((uint64_t)(z - a) != o[t + 1] - o[t]) { bad.store(2); continue; }
                uint64_t p = (uint64_t)(a - C.doc.data());
                for (uint64_t i = o[t]; i < o[t + 1]; i++, p++) { if (d[i] < 0 || d[i] >= cnt || d[i] + base != C.doc[p]) { bad.store(2); break; } C.w[p] = w[i]; }
            }
        });
        if (bad.load() == 1) return "a segment holds a term the documents do not produce: it was written from other documents";
        if (bad.load()) return "a segment list does not hold the documents the supplied documents produce for its term: it was written from other documents";
    }
    return nullptr;
}

inline const char* build_index(const DocSource& src, HostIndex& ix, const std::vector<SegmentPostings>* segs = nullptr) {
    const HostConfig& cfg = ix.cfg;
    int64_t flushed = 0;                                     // documents [0, flushed): their postings are replaced by the segments' (splice_segments)
    if (segs) for (auto& g : *segs) { if (g.docBase != flushed || g.docCount < 0) return "segments must cover the documents from 0 without gaps, in order"; flushed += g.docCount; }
    if (flushed > src.n) return "the segments hold more documents than were supplied";
    int threads = cfg.threads > 0 ? cfg.threads : effective_cpus();
    if (src.n < 4096) threads = 1;
    const int64_t N = src.n;
    ix.N = (int32_t)N;
    ix.docKey.resize(N);
    for (int64_t d = 0; d < N; d++) ix.docKey[d] = src.keys ? src.keys[d] : d;
    int nChunks = threads;
    std::vector<LocalInv> Lmain(nChunks), Lexact(nChunks), Lld1(nChunks), Lpref(nChunks), Lwords(nChunks);
    for (auto& l : Lmain) l.weighted = true;
    std::vector<std::vector<u16>> textChunks(nChunks);
    std::vector<uint32_t> textLen(N);
    // field order: OrderBy(weight) stable (DocumentFields.cs:71-77)
    std::vector<int> forder(src.fieldCount);
    for (int i = 0; i < src.fieldCount; i++) forder[i] = i;
    std::stable_sort(forder.begin(), forder.end(), [&](int a, int b) { return src.fieldWeights[a] < src.fieldWeights[b]; });

    parallel_for(N, nChunks, [&](int64_t b, int64_t e, int t) {
        LocalInv &M = Lmain[t], &EX = Lexact[t], &LD = Lld1[t], &PF = Lpref[t], &WD = Lwords[t];
        ustr concat, norm1, it, t2, wtext, tmp;
        struct Tk { uint32_t lid; float fw; };
        std::vector<Tk> toks; std::vector<uint32_t> ids;
        std::vector<std::pair<int, int>> bounds;
        for (int64_t d = b; d < e; d++) {
            concat.clear(); bounds.clear();
            for (int fi = 0; fi < src.fieldCount; fi++) {
                int f = forder[fi];
                uint64_t a = src.offs[d * src.fieldCount + f], z = src.offs[d * src.fieldCount + f + 1];
                bounds.push_back({(int)(uint16_t)concat.size(), src.fieldWeights[f]});
                concat.append(src.arena + a, z - a);
                if (fi + 1 < src.fieldCount) concat.push_back(u'§');
            }
            // Stage-2 / index text: lower(normalize(concat))   (VectorModel.cs:83-88)
            normalize_into(concat, it); lower_inplace(it);
            if (cfg.syn.has()) cfg.syn.canonicalize(it);     // VectorModel.cs:90-93 (index text) and SearchPipeline.cs:482-489 (coverage text)
            textLen[d] = (uint32_t)it.size();
            textChunks[t].insert(textChunks[t].end(), it.begin(), it.end());
            // WordMatcher / word-IDF text: normalize(lower(concat))  (WordMatcher.cs:85-89, VectorModel.cs:885-889)
            wtext.assign(concat); lower_inplace(wtext); normalize_into(wtext, tmp); wtext.swap(tmp);
            if (!it.empty()) {
                normalize_into(it, t2);   // Tokenizer normalises its input again (Tokenizer.cs:94-97)
                auto fweight = [&](int pos) -> float {
                    int wi = 0;
                    for (auto& bd : bounds) { if (bd.first <= pos) wi = bd.second; else break; }
                    return wi < 3 ? cfg.fieldWeights[wi] : 1.0f;
                };
                toks.clear();
                const int n = cfg.ngram;
                ustr& padded = tmp; padded.assign((size_t)cfg.startPad, (u16)0xFFFF); padded += t2; padded.append((size_t)cfg.stopPad, (u16)0xFFFE);
                if ((int)padded.size() >= n)
                    for (int i = 0; i + n <= (int)padded.size(); i++) {
                        bool allpad = true;
                        for (int k = 0; k < n; k++) if (padded[i + k] != 0xFFFF && padded[i + k] != 0xFFFE) { allpad = false; break; }
                        if (allpad) continue;
                        toks.push_back({M.keys.get_or_add(uview(padded.data() + i, n)), fweight(i)});
                    }
                for_each_word(t2, [&](int off, int len) { if (len >= n) toks.push_back({M.keys.get_or_add(uview(t2.data() + off, len)), fweight(cfg.startPad + off)}); });
                // aggregate per term in token order: running banker's-rounded byte (Term.cs:84-113)
                ids.resize(toks.size());
                for (size_t i = 0; i < toks.size(); i++) ids[i] = (uint32_t)i;
                std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return toks[x].lid < toks[y].lid; });
                // emit in first-appearance order of the term within the doc so that local ids keep corpus order
                // (local ids were assigned at get_or_add time already; emission order inside a doc is irrelevant)
                for (size_t i = 0; i < ids.size();) {
                    uint32_t lid = toks[ids[i]].lid;
                    double r0 = std::nearbyint((double)toks[ids[i]].fw);
                    uint8_t wgt = (uint8_t)std::min(r0, 255.0);
                    int ovf = 0, maxT = 1;
                    size_t j = i + 1;
                    for (; j < ids.size() && toks[ids[j]].lid == lid; j++) {
                        maxT = std::max(maxT, 1 + ovf + 1);
                        float nw = (float)wgt + toks[ids[j]].fw;
                        if (nw <= 255.f) wgt = (uint8_t)std::nearbyint((double)nw); else ovf++;
                    }
                    int net = 1 + ovf;
                    M.add(lid, (int32_t)d, wgt, (uint16_t)((std::min(net - 1, 255) << 8) | std::min(maxT - 1, 255)));
                    i = j;
                }
                // prefix index over indexText tokens (VectorModel.cs:109)
                ids.clear();
                for_each_word(it, [&](int off, int len) { int mx = std::min(len, 3); for (int L = 1; L <= mx; L++) ids.push_back(PF.keys.get_or_add(uview(it.data() + off, L))); });
                std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                for (uint32_t id : ids) PF.add(id, (int32_t)d);
            }
            // words (word-level df, affix last doc) + WordMatcher exact / LD1 deletions
            {
                std::vector<uint32_t>& wi = ids; wi.clear();
                std::vector<uint32_t> ex, ld;
                for_each_word(wtext, [&](int off, int len) {
                    uview w(wtext.data() + off, len);
                    wi.push_back(WD.keys.get_or_add(w));
                    if (cfg.wordMatcher) {
                        if (len >= cfg.wmMinExact && len <= cfg.wmMaxExact) ex.push_back(EX.keys.get_or_add(w));
                        if (len >= cfg.wmMinLD1 && len <= cfg.wmMaxLD1) {
                            ustr v;
                            for (int i = 0; i < len; i++) { v.assign(w); v.erase(i, 1); ld.push_back(LD.keys.get_or_add(v)); }
                        }
                    }
                });
                std::sort(wi.begin(), wi.end()); wi.erase(std::unique(wi.begin(), wi.end()), wi.end());
                for (uint32_t id : wi) WD.add(id, (int32_t)d);
                std::sort(ex.begin(), ex.end()); ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
                for (uint32_t id : ex) EX.add(id, (int32_t)d);
                std::sort(ld.begin(), ld.end()); ld.erase(std::unique(ld.begin(), ld.end()), ld.end());
                for (uint32_t id : ld) LD.add(id, (int32_t)d);
            }
        }
    });
    // text arena
    ix.textOff.assign(N + 1, 0);
    for (int64_t d = 0; d < N; d++) ix.textOff[d + 1] = ix.textOff[d] + textLen[d];
    ix.text.resize(ix.textOff[N]);
    { uint64_t p = 0; for (auto& c : textChunks) { std::memcpy(ix.text.data() + p, c.data(), c.size() * 2); p += c.size(); std::vector<u16>().swap(c); } }

    // NOTE on id order: a thread-local table numbers keys in the order the chunk first sees them, and chunks are merged
    // in document order, so global ids == the reference's first-appearance order PROVIDED that within one document the
    // n-grams are numbered before the words (they are: toks is filled n-grams first, as Tokenizer.cs:104-138 does).
    merge_inv(Lmain, ix.terms, threads);
    if (segs && !segs->empty()) { if (const char* err = splice_segments(ix, *segs, threads)) return err; }
    merge_inv(Lexact, ix.wmExact, threads);
    merge_inv(Lld1, ix.wmLd1, threads);
    merge_inv(Lpref, ix.prefixAll, threads);
    Csr wordsCsr; merge_inv(Lwords, wordsCsr, threads);

    // ---- df / stop terms (Term.cs:118-146, TermCollection.cs:92-131) -------------------------------------------------
    size_t T = ix.terms.K();
    ix.df.assign(T, 0);
    const int limit = cfg.stopTermLimit;
    parallel_for((int64_t)T, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t k = b; k < e; k++) {
            uint64_t lo = ix.terms.off[k], hi = ix.terms.off[k + 1];
            long running = 0; bool stop = false;
            for (uint64_t p = lo; p < hi; p++) running += 1 + (ix.terms.meta[p] >> 8);
            if (running + 1 > limit) {   // the counter may cross the limit somewhere: replay it (transient +1 on duplicates)
                running = 0;
                for (uint64_t p = lo; p < hi && !stop; p++) { int net = 1 + (ix.terms.meta[p] >> 8), mt = 1 + (ix.terms.meta[p] & 255); if (running + mt > limit) stop = true; running += net; }
            }
            ix.df[k] = stop ? -1 : (int32_t)running;
        }
    });
    // drop stop-term postings (Term.GetDocumentIds() == null) by compacting the CSR
    {
        std::vector<uint64_t> noff(T + 1, 0);
        for (size_t k = 0; k < T; k++) noff[k + 1] = noff[k] + (ix.df[k] > 0 ? ix.terms.len((uint32_t)k) : 0);
        if (noff[T] != ix.terms.off[T]) {
            for (size_t k = 0; k < T; k++) if (ix.df[k] > 0 && noff[k] != ix.terms.off[k]) {
                std::memmove(ix.terms.doc.data() + noff[k], ix.terms.doc.data() + ix.terms.off[k], ix.terms.len((uint32_t)k) * 4);
                std::memmove(ix.terms.w.data() + noff[k], ix.terms.w.data() + ix.terms.off[k], ix.terms.len((uint32_t)k));
            }
            ix.terms.doc.resize(noff[T]); ix.terms.w.resize(noff[T]); ix.terms.off.swap(noff);
        }
        std::vector<uint16_t>().swap(ix.terms.meta);
    }
    // ---- docLength = sum of byte weights over non-stop terms; avgdl = sequential fp32 sum (quirk Q6) --------------------
    ix.docLen.assign(N, 0.f);
    {
        // integer accumulation per doc is exact (== fp32 sums of bytes below 2^24)
        std::vector<std::atomic<uint32_t>> acc(N);
        for (auto& a : acc) a.store(0, std::memory_order_relaxed);
        parallel_for((int64_t)T, threads, [&](int64_t b, int64_t e, int) {
            for (int64_t k = b; k < e; k++) for (uint64_t p = ix.terms.off[k]; p < ix.terms.off[k + 1]; p++) acc[ix.terms.doc[p]].fetch_add(ix.terms.w[p], std::memory_order_relaxed);
        });
        for (int64_t d = 0; d < N; d++) ix.docLen[d] = (float)acc[d].load(std::memory_order_relaxed);
    }
    { float total = 0.f; for (int64_t d = 0; d < N; d++) total += ix.docLen[d]; ix.avgdl = N > 0 ? total / (float)N : 0.f; }

    // ---- sorted terms + trie ------------------------------------------------------------------------------------------
    ix.sortedTerms.resize(T);
    for (size_t i = 0; i < T; i++) ix.sortedTerms[i] = (uint32_t)i;
    std::sort(ix.sortedTerms.begin(), ix.sortedTerms.end(), [&](uint32_t a, uint32_t b) { return ix.terms.keys.key(a) < ix.terms.keys.key(b); });
    {
        ix.trie.clear(); ix.trie.push_back({0, 0, 0, -1, 0, 0});   // root
        std::vector<uint32_t> path{0}, lastChild{0};                 // node index per depth; last child index per depth
        uview prev;
        for (uint32_t id : ix.sortedTerms) {
            uview s = ix.terms.keys.key(id);
            size_t lcp = 0, mx = std::min(prev.size(), s.size());
            while (lcp < mx && prev[lcp] == s[lcp]) lcp++;
            path.resize(lcp + 1); lastChild.resize(lcp + 1);
            for (size_t dpt = lcp; dpt < s.size(); dpt++) {
                uint32_t nn = (uint32_t)ix.trie.size();
                ix.trie.push_back({s[dpt], 0, 0, -1, 0, 0});
                uint32_t parent = path[dpt];
                if (ix.trie[parent].firstChild == 0) ix.trie[parent].firstChild = nn; else ix.trie[lastChild[dpt]].nextSibling = nn;
                lastChild[dpt] = nn;
                path.push_back(nn); lastChild.push_back(0);
            }
            ix.trie[path[s.size()]].term = (int32_t)id;
            prev = s;
        }
    }
    {
        size_t nn = ix.trie.size();
        ix.edgeStart.assign(nn + 1, 0);
        for (size_t v = 0; v < nn; v++) { uint32_t c = 0; for (uint32_t k = ix.trie[v].firstChild; k; k = ix.trie[k].nextSibling) c++; ix.edgeStart[v + 1] = ix.edgeStart[v] + c; }
        ix.edgeLabel.resize(ix.edgeStart[nn]); ix.edgeChild.resize(ix.edgeStart[nn]);
        for (size_t v = 0; v < nn; v++) { uint32_t p = ix.edgeStart[v]; for (uint32_t k = ix.trie[v].firstChild; k; k = ix.trie[k].nextSibling) { ix.edgeLabel[p] = ix.trie[k].label; ix.edgeChild[p] = k; p++; } }
    }
    {   // reversed-term trie
        std::vector<u16> arena; std::vector<uint64_t> ao(T + 1, 0);
        for (size_t i = 0; i < T; i++) ao[i + 1] = ao[i] + ix.terms.keys.key((uint32_t)i).size();
        arena.resize(ao[T]);
        parallel_for((int64_t)T, threads, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; i++) { uview k = ix.terms.keys.key((uint32_t)i); u16* d = arena.data() + ao[i]; for (size_t c = 0; c < k.size(); c++) d[c] = k[k.size() - 1 - c]; }
        });
        auto rkey = [&](uint32_t id) { return uview(arena.data() + ao[id], (size_t)(ao[id + 1] - ao[id])); };
        std::vector<uint32_t> order(T);
        for (size_t i = 0; i < T; i++) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rkey(a) < rkey(b); });
        struct N { u16 label; int32_t term; uint32_t firstChild, nextSibling; };
        std::vector<N> tr; tr.push_back({0, -1, 0, 0});
        std::vector<uint32_t> path{0}, lastChild{0};
        uview prev;
        for (uint32_t id : order) {
            uview s2 = rkey(id);
            size_t lcp = 0, mx = std::min(prev.size(), s2.size());
            while (lcp < mx && prev[lcp] == s2[lcp]) lcp++;
            path.resize(lcp + 1); lastChild.resize(lcp + 1);
            for (size_t dpt = lcp; dpt < s2.size(); dpt++) {
                uint32_t nn = (uint32_t)tr.size();
                tr.push_back({s2[dpt], -1, 0, 0});
                uint32_t parent = path[dpt];
                if (tr[parent].firstChild == 0) tr[parent].firstChild = nn; else tr[lastChild[dpt]].nextSibling = nn;
                lastChild[dpt] = nn;
                path.push_back(nn); lastChild.push_back(0);
            }
            tr[path[s2.size()]].term = (int32_t)id;
            prev = s2;
        }
        size_t nn = tr.size();
        ix.rTerm.resize(nn); ix.rEdgeStart.assign(nn + 1, 0);
        for (size_t v = 0; v < nn; v++) { ix.rTerm[v] = tr[v].term; uint32_t c = 0; for (uint32_t k = tr[v].firstChild; k; k = tr[k].nextSibling) c++; ix.rEdgeStart[v + 1] = ix.rEdgeStart[v] + c; }
        ix.rEdgeLabel.resize(ix.rEdgeStart[nn]); ix.rEdgeChild.resize(ix.rEdgeStart[nn]);
        for (size_t v = 0; v < nn; v++) { uint32_t p2 = ix.rEdgeStart[v]; for (uint32_t k = tr[v].firstChild; k; k = tr[k].nextSibling) { ix.rEdgeLabel[p2] = tr[k].label; ix.rEdgeChild[p2] = k; p2++; } }
    }
    // ---- prefix DocSets: keep the lists prefix precedence can accept, counts for the rest --------------------------------
    {
        size_t PK = ix.prefixAll.K();
        ix.prefixKeys = std::move(ix.prefixAll.keys);
        ix.prefixPop.resize(PK); ix.prefixSetId.assign(PK, -1);
        ix.psOff.assign(1, 0);
        uint64_t cap = (uint64_t)20 * cfg.maxDepth;
        int32_t nset = 0;
        for (size_t k = 0; k < PK; k++) {
            uint64_t len = ix.prefixAll.off[k + 1] - ix.prefixAll.off[k];
            ix.prefixPop[k] = (uint32_t)len;
            if (len > 0 && len <= cap) {
                ix.prefixSetId[k] = nset++;
                ix.psDocs.insert(ix.psDocs.end(), ix.prefixAll.doc.begin() + ix.prefixAll.off[k], ix.prefixAll.doc.begin() + ix.prefixAll.off[k + 1]);
                ix.psOff.push_back(ix.psDocs.size());
            }
        }
        ix.prefixAll = Csr();
    }
    // ---- words: df -> word-level IDF; last doc -> affix lists ------------------------------------------------------------
    {
        size_t W = wordsCsr.K();
        ix.words = std::move(wordsCsr.keys);
        ix.wordDf.resize(W); ix.wordLastDoc.resize(W); ix.wordIdf.resize(W);
        for (size_t k = 0; k < W; k++) {
            uint64_t lo = wordsCsr.off[k], hi = wordsCsr.off[k + 1];
            ix.wordDf[k] = (uint32_t)(hi - lo); ix.wordLastDoc[k] = hi > lo ? wordsCsr.doc[hi - 1] : -1;
            ix.wordIdf[k] = compute_idf((int)N, (int)(hi - lo));
        }
        {   // OrdinalIgnoreCase classes with alias members: document frequency of the CLASS = documents holding any of its words (per document the reference's HashSet is
            // OrdinalIgnoreCase too: a document counts once)
            const auto& TT = tables();
            std::unordered_map<std::u16string, std::vector<uint32_t>> cls;
            for (uint32_t k = 0; k < W; k++) {
                uview key = ix.words.key(k); bool alias = false;
                for (u16 c : key) if (TT.icrep[c] != c) { alias = true; break; }
                if (!alias) continue;
                std::u16string F(key.begin(), key.end()); for (auto& c : F) c = (char16_t)TT.icrep[(u16)c];
                cls[F].push_back(k);
            }
            ix.icWords = KeyTable(); ix.icWordIdf.clear();
            std::vector<std::u16string> order; order.reserve(cls.size());
            for (auto& kv : cls) order.push_back(kv.first);
            std::sort(order.begin(), order.end());      // deterministic ids
            for (auto& F : order) {
                std::vector<uint32_t> members = cls[F];
                const int64_t rep = ix.words.find(uview((const u16*)F.data(), F.size())); if (rep >= 0) members.push_back((uint32_t)rep);
                std::vector<int32_t> docs;
                for (uint32_t m : members) docs.insert(docs.end(), wordsCsr.doc.begin() + wordsCsr.off[m], wordsCsr.doc.begin() + wordsCsr.off[m + 1]);
                std::sort(docs.begin(), docs.end()); docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
                const int df = (int)docs.size();
                ix.icWords.get_or_add(uview((const u16*)F.data(), F.size()));
                ix.icWordIdf.push_back((df > 0 && df <= (int)N) ? compute_idf((int)N, df) : 0.f);
            }
        }
        if (cfg.wordMatcher) {
            for (uint32_t k = 0; k < W; k++) if ((int)ix.words.keyLen[k] >= cfg.wmMinLD1) ix.affixFwd.push_back(k);
            ix.affixRev = ix.affixFwd;
            std::sort(ix.affixFwd.begin(), ix.affixFwd.end(), [&](uint32_t a, uint32_t b) { return ix.words.key(a) < ix.words.key(b); });
            auto revLess = [&](uint32_t a, uint32_t b) {
                uview x = ix.words.key(a), y = ix.words.key(b);
                size_t n = std::min(x.size(), y.size());
                for (size_t i = 0; i < n; i++) { u16 cx = x[x.size() - 1 - i], cy = y[y.size() - 1 - i]; if (cx != cy) return cx < cy; }
                return x.size() < y.size();
            };
            std::sort(ix.affixRev.begin(), ix.affixRev.end(), revLess);
        }
    }
    return nullptr;
}

} // namespace infx
