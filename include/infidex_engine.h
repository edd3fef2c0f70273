/*
 * infidex_engine.h — C ABI of the host engine inside libinfidex_hip.so.
 *
 * The reference's host is C# (src/Infidex/SearchEngine.cs); this image has no .NET toolchain, so the host side above the
 * device ABI (infidex_hip.h) is C++ and is exported here with the same surface for the hot path:
 *   SearchEngine.CreateDefault()/CreateMinimal()  SearchEngine.cs:78-94   -> infx_engine_create
 *   SearchEngine.IndexDocuments(IEnumerable<Document>)  :96-192           -> infx_engine_index_documents
 *   SearchEngine.Search(Query)                          :256-319          -> infx_engine_search_batch (one Query per row;
 *        Query.MaxNumberOfRecordsToReturn / CoverageDepth / EnableCoverage, Api/Query.cs:19,28,40)
 *   SearchEngine.Dispose                                 :477             -> infx_engine_destroy
 * A C# SearchEngine shim would call either this layer or infidex_hip.h directly.
 * All functions return int32 status (INFX_OK = 0) unless documented otherwise; infx_engine_last_error() gives the text.
 */
#ifndef INFIDEX_ENGINE_H
#define INFIDEX_ENGINE_H
#include "infidex_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct infx_engine infx_engine;

typedef struct infx_engine_config {
    int32_t device;           /* HIP device ordinal; -1 = host-only engine (indexing / planning introspection; Search fails) */
    int32_t range_docs;       /* see infx_config */
    int32_t max_depth;        /* largest Query.CoverageDepth (default 500) */
    int32_t threads;          /* host threads for indexing / query preparation; 0 = infx_engine_effective_cpus() */
    int32_t enable_coverage;  /* CreateDefault: 1, CreateMinimal: 0 */
    int32_t word_matcher;     /* CreateDefault: 1 (config 400 WordMatcherSetup), CreateMinimal: 0 */
    int32_t stop_term_limit;  /* 0 = 1 250 000 */
    int32_t want_features;    /* 1: Stage 2 also returns the integer feature vector (parity tests) */
    int32_t no_exact_replay;  /* 1: INFX_CFG_NO_EXACT_REPLAY (infidex_hip.h) — Stage-1 cut ties by (score, doc id) instead of the reference's heap order */
} infx_engine_config;

const char* infx_engine_last_error(void);
int32_t infx_engine_create(const infx_engine_config* cfg, infx_engine** out);
void    infx_engine_destroy(infx_engine* e);

/* n documents x field_count fields; field k of document d is arena[offs[d*field_count+k] .. offs[d*field_count+k+1]) (UTF-16);
 * field_weights[k] in {0 High, 1 Med, 2 Low} (Api/Weight.cs); keys == NULL means DocumentKey = document index. */
int32_t infx_engine_index_documents(infx_engine* e, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs,
                                    int32_t field_count, const int32_t* field_weights);

/* out_keys/out_scores/out_ties: nq x max_results (row-major); out_counts: nq; out_flags (may be NULL): bit0 query needs the
 * short-query path or exceeds the Stage-2 query envelope (empty result), bit1 coverage stage ran, bit2 coverage returned nothing -> Stage-1
 * fallback, bit3 a candidate document was left out of the ranking: the batch's over-long documents (> INFX_MAX_DOC_TOKENS words) exceeded the token-table pool. */
int32_t infx_engine_search_batch(infx_engine* e, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                 int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                 uint32_t* out_counts, uint32_t* out_flags);

/* Sessions: one in-flight batch each (own HIP stream + scratch). Calling infx_engine_session_search_batch from several host
 * threads, one session per thread, overlaps the host preparation of one batch with the GPU stages of another — the
 * reference's concurrent-reader model (ReaderWriterLockSlim read lock, SearchEngine.cs:33,258). */
typedef struct infx_session infx_session;
int32_t infx_engine_session_create(infx_engine* e, infx_session** out);
void    infx_engine_session_destroy(infx_session* s);
int32_t infx_engine_session_search_batch(infx_session* s, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t max_results,
                                         int32_t depth, int32_t enable_coverage, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                                         uint32_t* out_counts, uint32_t* out_flags);
int32_t infx_engine_session_last_timings(infx_session* s, double* host_ms5, float* kernel_ms5, uint64_t* alg6);
/* exact Stage-1 replay of the session's last batch: kernel milliseconds (part of kernel_ms5[1]) and the k_select flag reasons (infx_last_replay_stats) */
/* Where the last batch's planning time went (ms): [0] text preparation + term lookups incl. the LD1 expansion call, [1] of that inside infx_ld1_expand
 * (device work and the wait for it), [2] inside infx_union_build (ditto), [3] idf / roles / device records. */
int32_t infx_engine_session_plan_breakdown(infx_session* S, double* out4);
int32_t infx_engine_session_replay_breakdown(infx_session* S, float* ms4 /* scan (k_ex_walk x2 + k_ex_prefix + k_ex_theta), k_ex_chunk, k_ex_heap, k_exact1 of the last batch */);
int32_t infx_engine_session_last_replay(infx_session* s, float* ms, uint32_t* why3);

/* Document-sharded operation: every rank indexes the whole corpus on the host (global df / avgdl / N), uploads
 * its contiguous doc range, and a batch runs as four phases with the collectives in between:
 *   phase0 (text prep, LD1 member lists, k_union_count) -> all-reduce(sum) of infx_session_union_counts    [Exchange 1b: fuzzy df]
 *   phase1 (idf/roles + k_accumulate)  -> all-reduce(sum) of infx_session_counts (ndev x INFX_NCLASS)     [Exchange 1]
 *   phase2 (k_select, global counts) -> all-gather of hits (ndev x depth) and hit counts                  [Exchange 2, RCCL over xGMI]
 *   phase3 (merge to the global top-depth, Stage-2 prep, k_stage2 on the OWNED candidates)
 *          -> all-reduce(sum) of infx_session_outs (ncand x 3 int32; every candidate is scored by exactly one shard)
 *   phase4 (final ordering / truncation). */
/* SynonymMap.AddSynonym (Synonyms/SynonymMap.cs:33-62), before infx_engine_index_documents: index text, query text and coverage text are
 * canonicalised with the map (UTF-16 terms). */
int32_t infx_engine_add_synonym(infx_engine* e, const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb);
int32_t infx_engine_set_shard(infx_engine* e, int32_t rank, int32_t nranks);      /* before infx_engine_index_documents */
int32_t infx_engine_shard_info(infx_engine* e, int32_t* doc_base, int32_t* num_docs);
int32_t infx_engine_default_session(infx_engine* e, infx_session** out);
int32_t infx_session_phase0(infx_session* s, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, uint32_t* nunions);
int32_t infx_session_union_counts(infx_session* s, uint32_t* counts);
int32_t infx_session_phase1(infx_session* s, const uint32_t* global_union_counts, uint32_t* ndev);
int32_t infx_session_counts(infx_session* s, uint32_t* counts);
int32_t infx_session_phase2(infx_session* s, const uint32_t* global_counts, infx_hit* hits, uint32_t* hitcounts);
/* phase 2 in three steps for the exact cut across shards (infidex_hip.h, infx_shard_replay_*); every buffer may be host or device memory:
 *   phase2a: tier rules + first-pass top-`depth` + best score left out                 -> all-gather (hits, hitcounts, next)
 *   phase2b: global ambiguity test + this shard's part of the replay; *blob_bytes      -> max over ranks; infx_session_phase2b_blob -> all-gather
 *   phase2c: owner-side heap replay; hits / hitcounts = this rank's contribution         -> all-gather -> phase 3
 *   phase2d: (only if a hitcount is 0xFFFFFFFF) sequential chain step for the queries with need[q] != 0 */
int32_t infx_session_phase2a(infx_session* s, const void* global_counts, void* hits, void* hitcounts, void* next);
int32_t infx_session_phase2b(infx_session* s, int32_t nranks, const void* all_hits, const void* all_hitcounts, const void* all_next, uint64_t* blob_bytes);
int32_t infx_session_phase2b_blob(infx_session* s, void* dst, uint64_t padded_bytes);
int32_t infx_session_phase2c(infx_session* s, int32_t nranks, const void* all_blobs, uint64_t padded_bytes, void* hits, void* hitcounts);
int32_t infx_session_phase2d(infx_session* s, const uint32_t* need, void* state);
/* ---- native driver of the sharded phases -------------------------------------------------------------------------------------------------------------
 * Everything after phase 0 of a batch — phases 1, 2a-2d, 3, 4 with every collective in between — in ONE call, driven from C++ with no interpreter on the
 * path.  The collectives come through an infx_comm:
 *   infx_engine_comm_rccl : RCCL inside the library (infidex_hip.h: infx_set_shard_comm).  Exchange buffers live in HBM, the collectives are enqueued on the
 *                           session's HIP stream between the kernels that produce and consume them; the host synchronises only where it needs a value
 *                           (global fuzzy df for idf, the padded blob size, the "needs the sequential chain" flags).
 *   caller-supplied ops   : any other transport (the tests pass torch.distributed/gloo callbacks; device_buffers = 0: host exchange buffers).
 * Results are those of the phase-by-phase API (infidex_amd/sharded.py drives either). */
typedef struct infx_comm {
    void*   ctx;
    int32_t rank, nranks;
    int32_t device_buffers;      /* 1: buffers handed to the ops are device memory and the ops are stream-ordered on `stream`; 0: host memory, blocking ops */
    int32_t reserved;
    int32_t (*allreduce_sum_u32)(void* ctx, void* buf, uint64_t count, void* stream);                       /* in place */
    int32_t (*allgather)(void* ctx, const void* send, void* recv, uint64_t bytes_per_rank, void* stream);   /* recv: nranks x bytes_per_rank */
} infx_comm;
int32_t infx_engine_rccl_unique_id(void* id128);                                  /* rank 0: 128 bytes to ship to every rank */
int32_t infx_engine_comm_rccl(infx_engine* e, const void* id128, infx_comm* out);  /* every rank, after infx_engine_index_documents */
int32_t infx_session_comm_rccl(infx_session* s, const void* id128, infx_comm* out); /* a communicator per session: several batches in flight per rank */
/* phase 0 first (infx_session_phase0, possibly on a planner thread); then this call, in the same order on every rank */
int32_t infx_session_sharded_finish(infx_session* s, const infx_comm* comm, int32_t max_results, int32_t enable_coverage,
                                    int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags);

/* Collective order across a rank's pipeline sessions.  Each session has its own communicator and HIP stream; left alone, the order in which the sessions' host
 * threads enqueue their collectives — hence the relative order of different communicators' kernels on the device — depends on thread timing and differs between
 * ranks (the multi-communicator hang).  infx_engine_coll_ring declares the sessions of the coming stream (batch i on sessions[i mod n], the same on every rank):
 * inside infx_session_sharded_finish they then take turns, one collective per turn, in ring order; a session calls infx_session_coll_retire after its last batch
 * of the stream.  n = 0: no ordering.  INFX_COLL_ORDER=0 switches it off. */
int32_t infx_engine_coll_ring(infx_engine* e, uint32_t n, infx_session* const* sessions);
int32_t infx_session_coll_retire(infx_session* s);
int32_t infx_session_coll_stats(infx_session* s, uint64_t* out4 /* all-reduce calls, all-gather calls, all-reduce bytes, all-gather bytes; cumulative */);
int32_t infx_session_phase3(infx_session* s, int32_t nranks, const infx_hit* all_hits, const uint32_t* all_hitcounts, int32_t max_results,
                            int32_t enable_coverage, uint64_t* ncand);
int32_t infx_session_outs(infx_session* s, int32_t* outs3);
/* the same phases with caller-owned exchange buffers on the host OR the device (nd x depth hits + nd counts; nshards x ... ; nq x 2*depth
 * rows of 3 int32): with device tensors the collectives (RCCL) work in place and nothing crosses PCIe. infx_session_phase4 accepts either. */
int32_t infx_session_phase1x(infx_session* s, const uint32_t* global_union_counts, void* counts /* nq x INFX_NCLASS, host or device */, uint32_t* ndev);
int32_t infx_session_phase2x(infx_session* s, const uint32_t* global_counts /* host or device */, void* hits, void* hitcounts);
int32_t infx_session_phase3x(infx_session* s, int32_t nranks, const void* all_hits, const void* all_hitcounts, int32_t max_results, int32_t enable_coverage, void* outs);
int32_t infx_session_phase4(infx_session* s, const int32_t* merged_outs3, int64_t* out_keys, float* out_scores, uint8_t* out_ties,
                            uint32_t* out_counts, uint32_t* out_flags);

/* host_ms5: plan; Stage 1 incl. transfers (phase API only, 0 for the fused pipeline); Stage-2 preparation (fused pipeline:
 * WordMatcher list descriptors + PrepareQuery); GPU stage (phase API: Stage 2 incl. transfers; fused: the whole device
 * pipeline, one synchronisation); final ordering on the host (phase API only);
 * kernel_ms5: accumulate, rules + select, stage2, and for the fused pipeline candidate assembly (k_prep2) and final ordering
 * (k_finalize) — kernel durations from HIP events on the launch stream;
 * alg (6 entries): algorithmic bytes of the accumulate launch per SURVEY 8(d), Stage-2 candidate count, Stage-2 text bytes,
 * bytes the accumulate launch actually streamed, Stage-1 candidates, queries replayed by k_exact1. */
int32_t infx_engine_last_timings(infx_engine* e, double* host_ms5, float* kernel_ms5, uint64_t* alg6);

/* ---- introspection used by the parity tests (host logic runs without a GPU) ---- */
int32_t infx_engine_index_stats(infx_engine* e, int64_t* n_docs, int64_t* n_terms, int64_t* n_postings, float* avgdl);
int32_t infx_engine_export_index(infx_engine* e, int32_t* df, uint64_t* post_off, int32_t* post_doc, uint8_t* post_w, float* doc_len);
int32_t infx_engine_term_text(infx_engine* e, int32_t t, uint16_t* out, int32_t cap);
int32_t infx_engine_match_ld1(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap);
int32_t infx_engine_match_ld1_forward(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap);
int32_t infx_engine_plan(infx_engine* e, const uint16_t* q, int32_t len, int32_t depth, int32_t* term_ids, int32_t* dfs, float* idfs,
                         uint8_t* roles, uint8_t* ranks, int32_t cap, int32_t* meta5, int32_t* flags);
int64_t infx_engine_wordmatcher(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int64_t cap);
/* The LD1 / WordMatcher lookups of planning as the DEVICE answers them (infidex_hip.h "Dictionary lookups"; uploaded by IndexDocuments unless
 * INFX_HOST_LOOKUPS=1).  _match_ld1_device: like _match_ld1 (count, first `cap` term ids in ordinal order); -1 / -2 = the kernel handed the word back to the
 * host (work lists outgrown / length outside 1..64).  _wordmatcher_device: like _wordmatcher (ascending unique doc ids over all of the query's lists). */
int32_t infx_engine_device_lookups(infx_engine* e);
int32_t infx_engine_lookup_stats(infx_engine* e, int64_t* out4 /* LD1 words on device / on host, WordMatcher queries on device / on host */);
int32_t infx_engine_match_ld1_device(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int32_t cap);
int64_t infx_engine_wordmatcher_device(infx_engine* e, const uint16_t* q, int32_t len, int32_t* out, int64_t cap);
int32_t infx_engine_prefix_pop(infx_engine* e, const uint16_t* p, int32_t len);
int32_t infx_engine_last_stage1(infx_engine* e, uint32_t qi, int64_t* keys, float* scores, int32_t cap);
int64_t infx_engine_last_stage2(infx_engine* e, uint32_t* query_of, int32_t* docs, float* base, float* scores, uint8_t* ties,
                                int32_t* feat, int64_t cap);
/* CPUs usable by this process: hardware threads capped by the affinity mask and the cgroup CPU quota (INFX_THREADS overrides);
 * the default size of the host worker pool and of `threads`. */
/* ---- Document.Deleted ----------------------------------------------------------------------------------------------------------------------
 * DocumentCollection.DeleteDocumentsByKey (Core/DocumentCollection.cs:200-212): marks every document with one of the keys as deleted; the
 * index statistics are not rebuilt (as in the reference until the next re-index).  Searches skip deleted documents where the reference does
 * (Bm25Scorer.cs:322-323,455-459,622-624; SearchPipeline.cs:404-406,463-465,532-537).  Exclusive — no search in flight (the reference's write
 * lock).  On a sharded engine every rank must make the same call.  *out_marked (optional) = documents newly marked. */
int32_t infx_engine_delete_documents(infx_engine* e, const int64_t* keys, int64_t n, int64_t* out_marked);
/* SearchEngine.Load (SearchEngine.cs:399-441) of an INFDX2 file written by SearchEngine.Save (Indexing/IndexPersistence.cs:33-99): header and data
 * checksums are verified, the host index is built from the documents { DocumentKey, IndexedText as the single Med-weight field "content", Deleted }
 * and — BEFORE anything is uploaded — compared with what the file stores: every stored term (text, document frequency, postings with their weight
 * bytes; both ways) and every derived section the reference's Load would read and search with (csrc/host/infdx2_verify.h): the term FST (every term
 * with its collection index, forward and reverse trie in FstBuilder's layout), the short-query index (every (prefix, document, token position) entry
 * regenerated from the index texts), the document metadata cache (first token, token count) and the WordMatcher section behind the checksum (exact and
 * symmetric-delete dictionaries with their Roaring document sets, the affix FST with its last-occurrence documents).  A WordMatcher section on an engine
 * configured without one, or none on an engine configured with one, is refused as SearchEngine.cs:436-439 throws.
 * checked3 (optional): documents, stored terms compared, stored postings compared.  INFX_EUNSUPPORTED when a stored structure is not what the builder
 * produces for the stored texts (e.g. written from differently weighted fields, another stop-term limit, a derived section that says something else than
 * the documents) — the engine is left unindexed and can load another file; INFX_EINVAL for a foreign or corrupted file. */
int32_t infx_engine_load_index(infx_engine* e, const char* path, int64_t* checked3);
/* SearchEngine.Flush's on-disk segments (INFS: Indexing/Segments/SegmentWriter.cs:13-94, BlockPostingsWriter.cs:24-161 — blocks of 64..256 delta-coded postings
 * with min / max doc and max weight per block, Compression/GroupVarInt.cs, the "FST2" term tries, Elias-Fano list offsets).  infx_segment_open reads and
 * VALIDATES a file (every offset, the skip tables against the decoded blocks, the tries against each other, the Elias-Fano select index); infx_segment_export hands its
 * content over as CSR in ordinal term order — term texts (UTF-16 arena + offsets) and, per term, ascending segment-local doc ids with their weight bytes: the layout
 * infx_upload_postings takes (map the ordinals to the host's term ids first).  Any output pointer may be NULL. */
typedef struct infx_segment infx_segment;
int32_t infx_segment_open(const char* path, infx_segment** out);
void    infx_segment_close(infx_segment* seg);
int32_t infx_segment_info(infx_segment* seg, int32_t* doc_count, int32_t* num_terms, int64_t* num_postings, int64_t* term_chars);
int32_t infx_segment_export(infx_segment* seg, uint32_t* term_offs /* T+1 */, uint16_t* term_chars, uint64_t* post_offs /* T+1 */, int32_t* doc_ids, uint8_t* weights);
/* A flushed segment against the engine's index of the same documents: the segment covers the documents [doc_base, doc_base + its docCount) (VectorModel.Flush writes
 * ids relative to the documents flushed before, VectorModel.cs:804-815).  Every term of the segment must exist in the index with exactly the segment's (document,
 * weight) postings inside that range, and every index term with postings in the range must be in the segment.  checked3 (optional): documents, terms, postings
 * compared.  INFX_EINVAL for a foreign or corrupted file, INFX_EUNSUPPORTED when segment and index disagree. */
int32_t infx_engine_verify_segment(infx_engine* e, const char* path, int32_t doc_base, int64_t* checked3);
/* An engine populated from FLUSHED SEGMENTS + a live tail instead of infx_engine_index_documents (VectorModel.Flush, Indexing/VectorModel.cs:804-815;
 * Indexing/Segments/SegmentReader.cs): segment i holds the postings of documents [doc_bases[i], doc_bases[i] + its document count); the segments cover the
 * documents from 0 without gaps, in order; documents behind the last segment are the live tail.  All n documents are supplied and accumulated (term ids, the
 * df counter with its double counts on saturated weight bytes, the stop-term decisions, document lengths, WordMatcher dictionaries and Stage-2 texts need them: a
 * segment stores none of that); the segments' (document, weight) postings then replace the accumulated ones of their ranges, list by list, and must name exactly
 * the documents the accumulation found (a term without a list in a segment was a stop term at flush time and stays one).
 * The corpus is then searched as ONE index (= an unflushed index of the same documents), not segment by segment as VectorModel.cs:572-584 does.
 * INFX_EUNSUPPORTED: a segment written from other documents; the engine stays unindexed. */
int32_t infx_engine_index_from_segments(infx_engine* e, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs, int32_t field_count, const int32_t* field_weights,
                                        int32_t n_segments, const char* const* paths, const int32_t* doc_bases);
int32_t infx_engine_restore_documents(infx_engine* e);      /* clears every Deleted flag */
/* One host-index build per node instead of one per rank (document shards: every process needs the whole host index — global df / avgdl / N, the term and
 * word dictionaries, the WordMatcher lists).  The node's leader indexes the documents (infx_engine_set_build_threads lets that build use every
 * core of the node while planning keeps the rank's share), saves the host index to a node-local file (e.g. under /dev/shm) and the other ranks call
 * infx_engine_index_from_host_cache INSTEAD of infx_engine_index_documents: they read the arrays back and upload their own shard.  The file pins its
 * layout version, the configuration (n-gram, stop-term, WordMatcher, synonym settings), the index fingerprint and a checksum of its payload; anything
 * else is refused (INFX_EINVAL).  It is created exclusively (O_EXCL | O_NOFOLLOW, mode 0600) and read without following a link.  A transient hand-off between the processes of one build — not the reference's INFDX2 format (infx_engine_load_index). */
int32_t infx_engine_set_build_threads(infx_engine* e, int32_t threads);     /* 0 = the engine's thread count */
int32_t infx_engine_save_host_index(infx_engine* e, const char* path);
int32_t infx_engine_index_from_host_cache(infx_engine* e, const char* path);

/* ---- Query.Filter (Infiscript, Api/FilterParser.cs) and Query.EnableFacets (config 5) -------------------------------------------------
 * Non-indexed document fields are given as columns (one value per indexed document, in indexing order); the post-filter of the returned
 * rows (Scoring/ResultProcessor.cs:35-70), Filter.NumberOfDocumentsInFilter and the facet counts (Core/FacetBuilder.cs:19-105) run on the
 * device (include/infidex_hip.h).  kind: 1 int64, 2 double, 3 UTF-8 strings (arena + n+1 offsets). */
int32_t infx_engine_add_column(infx_engine* e, const char* name, int32_t kind, int32_t facetable, int64_t n, const int64_t* vals_i, const double* vals_d,
                               const char* arena, const uint64_t* offs);
int32_t infx_engine_column_count(infx_engine* e);
int32_t infx_engine_column_info(infx_engine* e, int32_t col, char* name, int32_t cap, int32_t* facetable, int32_t* num_values);
int32_t infx_engine_column_value(infx_engine* e, int32_t col, uint32_t code, char* out, int32_t cap);
int32_t infx_engine_set_filter(infx_session* s, const char* expr_utf8 /* NULL = no filter */, int32_t enable_facets, uint32_t* n_in_filter);
int32_t infx_engine_facet_column_count(infx_session* s);
int32_t infx_engine_last_facets(infx_session* s, uint32_t nq, uint32_t qi, uint32_t k, int32_t* col, uint32_t* codes, uint32_t* counts, int32_t cap);
/* The infx_cov_query (CoverageEngine.PrepareQuery) the engine would hand to the device for this raw query text: lets a caller of the device ABI
 * (infx_stage2_batch) prepare Stage-2 inputs without the engine's search path. */
int32_t infx_engine_prepare_cov_query(infx_engine* e, const uint16_t* q, int32_t len, infx_cov_query* out);
int32_t infx_sizeof_cov_query(void);
/* the same for a query beyond the fast Stage-2 envelope (infx_engine_prepare_cov_query returns INFX_EUNSUPPORTED for it): the record of the long-query table
 * (infx_stage2_long_queries, include/infidex_hip.h) */
int32_t infx_engine_prepare_cov_query_long(infx_engine* e, const uint16_t* q, int32_t len, infx_cov_query_long* out);
int32_t infx_sizeof_cov_query_long(void);
int32_t infx_engine_effective_cpus(void);
/* Sharded planning — the PLAN EXCHANGE.  Every rank of a document-sharded job answers the same queries and holds the whole host index, and planning is a
 * pure function of (index, query text): rank r therefore plans queries [begin, end) of the coming batch only — text preparation and term lookups
 * (VectorModel.cs:376-420), the coverage query context (CoverageEngine.PrepareQuery, CoverageEngine.cs:61-126) and, when the dictionaries are NOT on the device,
 * the LD1 expansions and WordMatcher descriptors — the ranks exchange the serialised results (one all-gather of byte blobs on a transport of the caller's,
 * infidex_amd/sharded.py: a gloo group per pipeline session) and import each other's before infx_session_phase0, which then only runs what depends on
 * the batch as a whole or on this shard (expansion cache, fuzzy unions and their global df, idf / roles, list descriptors).  Per-rank host work per query
 * goes from the whole plan to 1/W of it plus the import.  collect returns the blob size (or -1) and keeps this rank's own slice, blob copies it out, import
 * takes a peer's blob (validated: everything in it that reaches the device is range-checked; a plan is only used for the text and depth it was made for).
 * Results are identical with or without the exchange. */
int64_t infx_session_prefetch_collect(infx_session* s, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, uint32_t begin, uint32_t end, int32_t depth);
int32_t infx_session_prefetch_blob(infx_session* s, uint8_t* out, int64_t cap);
int32_t infx_session_prefetch_import(infx_session* s, const uint8_t* blob, int64_t len);
int64_t infx_session_prefetch_pending(infx_session* s);        /* imported WordMatcher descriptor sets waiting for the next phase 0 */
int32_t infx_session_plan_exchange_stats(infx_session* s, uint32_t* out2 /* last phase 0: queries planned from the exchange (own slice + imported), of them imported */);
/* parity tooling, no device needed: a digest per query of everything the exchange carries for it, from the pending exchange entries where they match the text,
 * computed locally otherwise (*from_exchange counts the former); entries are not consumed */
int32_t infx_session_plan_digest(infx_session* s, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, uint64_t* out, uint32_t* from_exchange);
/* Measurement hook (no device needed): single-threaded host planning cost of a batch by stage, microseconds per query:
 * out_us[0] plan_tokens, [1] of it LD1 walks, [2] plan_finish, [3] wm_collect, [4] prepare_cov_query, [5] plan_tokens_text (the exchangeable share of [0]),
 * [6] import of the batch's exchanged plans, [7] parsing them in phase 0 (replaces [5] + [4] for an imported query).  out_us holds 8 doubles. */
int32_t infx_engine_host_plan_profile(infx_engine* e, uint32_t nq, const uint16_t* q_arena, const uint64_t* q_offs, int32_t depth, double* out_us);
/* Entries of the LD1 expansion cache (least recently used, at most 1000: VectorModel.cs:42). */
int64_t infx_engine_fuzzy_cache_size(infx_engine* e);
/* Switches infx_engine_config.want_features at run time (the introspection buffers behind infx_engine_last_stage1 / _last_stage2). */
int32_t infx_engine_set_introspection(infx_engine* e, int32_t on);
int32_t infx_engine_normalize(const uint16_t* s, int32_t len, int32_t lower, uint16_t* out, int32_t cap);
int32_t infx_engine_device_handles(infx_engine* e, infx_index** idx, infx_stream** st);

#ifdef __cplusplus
}
#endif
#endif
