/*
 * infidex_hip.h — C ABI of libinfidex_hip.so: the MI355X (gfx950) drop-in for the query-time scoring hot path
 * of lofcz/Infidex (SearchEngine.Search -> SearchPipeline.Execute).
 *
 * The reference has no FFI seam today; the two INTERNAL calls this library replaces are
 *   S1  Bm25Scorer.Search(TermScoreInfo[] termInfos, int topK, int totalDocs, float[] docLengths, float avgdl, ...)
 *         -> TopKHeap                       src/Infidex/Indexing/Bm25Scorer.cs:56-67  (caller VectorModel.cs:567-569)
 *   S2  CoverageEngine.CalculateFeatures(ctx, docText, lcsSum, buffer, docId) + FusionScorer.Calculate(...)
 *         -> (float score, byte tiebreaker) src/Infidex/Coverage/CoverageEngine.cs:174, Scoring/FusionScorer.cs:19-25
 *                                           (caller SearchPipeline.ProcessCandidate, Scoring/SearchPipeline.cs:449-522)
 * A C# host binds these entry points with [DllImport("infidex_hip")] (stub in INTEGRATION.md).
 *
 * Conventions: every function returns an int32 status (0 = INFX_OK); no exceptions cross the ABI; the caller owns
 * all buffers; uploads are copied to HBM and host pointers are never retained; handles are opaque and freed only by
 * infx_destroy. infx_stage*_batch are re-entrant on an uploaded (immutable) index when each caller uses its own
 * infx_stream; infx_upload_* / infx_destroy are exclusive (the reference holds its write lock there,
 * SearchEngine.cs:96-104). Thread-local error text: infx_last_error().
 */
#ifndef INFIDEX_HIP_H
#define INFIDEX_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INFX_OK            0
#define INFX_EINVAL        1   /* bad argument / unsupported size */
#define INFX_ENOMEM        2   /* hipMalloc failed */
#define INFX_EHIP          3   /* HIP runtime error (incl. "no GPU") */
#define INFX_ECAPACITY     4   /* a per-batch workspace bound would be exceeded; split the batch */
#define INFX_EUNSUPPORTED  5   /* input outside the supported envelope (e.g. > INFX_MAX_DOC_TOKENS tokens) */
#define INFX_ENCCL         6   /* RCCL error (or librccl could not be loaded) */

#define INFX_MAX_QUERY_TERMS   128  /* VectorModel.cs:381 rents 128 raw tokens */
#define INFX_MAX_QUERY_TOKENS  32   /* Stage-2 query words after dedupe: the fast envelope (infx_cov_query) */
#define INFX_MAX_QUERY_CHARS   512
#define INFX_LONGQ_TOKENS      128  /* the long envelope (infx_cov_query_long): queries beyond the fast one take k_stage2's long-query launches — slower, same results */
#define INFX_LONGQ_CHARS       2048
#define INFX_MAX_DOC_TOKENS    192  /* Stage-2 words per document text handled in registers / scratch; longer texts (the reference allows 65 535 characters,
                                       Api/DocumentFields.cs:140) take k_stage2's global-workspace pass: slower, same results */
#define INFX_NFEAT             32   /* ints per infx_cov_out.feat */

typedef struct infx_index infx_index;     /* device-resident immutable index (one shard) */
typedef struct infx_stream infx_stream;   /* per-caller HIP stream + workspace */

/* Replaces the constants of Bm25Scorer.cs:21-23 / ConfigurationParameters.cs:101-104. */
typedef struct infx_config {
    int32_t device;          /* HIP device ordinal */
    int32_t range_docs;      /* documents per LDS range block (power of two, 512..16384); 0 = chosen from the shard size at
                                infx_upload_docs: 1024 below 256k documents ... 8192 from 4M documents */
    int32_t max_depth;       /* largest Query.CoverageDepth that will be used (default 500) */
    int32_t flags;           /* INFX_CFG_* */
} infx_config;
/* Queries whose Stage-1 top-`depth` cut is ambiguous (rows within a few ulp of each other around the cut, exact ties included) are
 * replayed with the reference's sequential semantics — chunked Vector256 / scalar-tail BM25 rounding (Bm25Scorer.cs:395-444) and the
 * BCL PriorityQueue eviction order (Bm25Scorer.cs:654-670) — by k_exact1, so the Stage-1 set and scores are the reference's bit for
 * bit.  This flag turns the replay off (the cut is then taken by (score, doc id)).  Document shards replay through infx_shard_replay_* (below). */
#define INFX_CFG_NO_EXACT_REPLAY 1

int32_t infx_create(const infx_config* cfg, infx_index** out);
void    infx_destroy(infx_index* idx);
const char* infx_last_error(void);

/* Postings: Term.docIds/weights (Core/Term.cs:21-22) flattened to CSR. df[t] < 0 marks a stop term (Term.cs:134-146);
 * its slice must be empty. doc ids are shard-local internal ids in ascending order. */
int32_t infx_upload_postings(infx_index* idx, uint32_t num_terms, const uint64_t* offs /* T+1 */,
                             const int32_t* doc_ids, const uint8_t* tf, const int32_t* df);

/* Documents: VectorModel._docLengths/_avgDocLength (VectorModel.cs:25-26, global avgdl!), DocumentKey, and the
 * Stage-2 text = ToLowerInvariant(TextNormalizer.Normalize(doc.IndexedText)) as UTF-16 (SegmentProcessor.cs:42-75;
 * every Stage-2 comparison in the reference is OrdinalIgnoreCase, so folding once at upload is equivalent). */
int32_t infx_upload_docs(infx_index* idx, uint32_t num_docs, const float* doc_len, float avgdl,
                         const int64_t* doc_key, const uint8_t* deleted /* N flags (Document.Deleted) or NULL; unsharded indexes only */,
                         const uint64_t* text_offs /* N+1 */, const uint16_t* text_utf16);

/* Document.Deleted (Core/Document.cs) after the upload: `deleted` holds one flag per GLOBAL internal id (`total_docs` of them — the same array on
 * every shard; NULL clears all flags).  Nothing else changes, exactly as in the reference between a deletion and the next re-index: postings, df,
 * doc lengths and avgdl keep the deleted documents.  The query path then skips them where the reference does — a deleted document is scored with
 * its chunk but never offered to the top-K heap (Bm25Scorer.cs:322-323, 455-459, 622-624), is not scored by Stage 2 (SearchPipeline.cs:463-465), and
 * a deleted WordMatcher id gets no docIndex (SearchPipeline.cs:532-537) while still counting against the WordMatcher-only limit (:387-397).
 * Call it while no search is in flight on this index (the reference mutates documents under its write lock). */
int32_t infx_set_deleted(infx_index* idx, uint32_t total_docs, const uint8_t* deleted);

/* Prefix DocSets (PrefixPostingList.DocSet, Indexing/ShortQuery/PrefixPosting.cs:64,109-137) that prefix precedence
 * can accept (population <= 20*max_depth), CSR over sets; referenced by set index from infx_query.prefix_set. */
int32_t infx_upload_prefix_docsets(infx_index* idx, uint32_t num_sets, const uint64_t* offs, const int32_t* doc_ids);

/* Document-sharded operation: this index holds internal ids [doc_base, doc_base+num_docs) of a corpus
 * of total_docs; corpus statistics passed in queries are global. */
int32_t infx_set_shard(infx_index* idx, int32_t rank, int32_t nranks, int32_t doc_base, int32_t total_docs);

/* RCCL inside the boundary (infx_set_shard(..., ncclUniqueId)).  Rank 0 creates the 128-byte ncclUniqueId and ships it to every rank
 * (any transport); every rank then joins the communicator of its index (ncclCommInitRank(nranks, id, rank) on the index's device, after infx_set_shard).
 * The collectives below are enqueued on the stream's HIP stream over DEVICE buffers — stream-ordered behind the kernels that produced the data and in
 * front of the ones that consume it, no host synchronisation: count all-reduce (Exchange 1 / 1b), all-gather of the per-shard top-k (Exchange 2).
 * librccl is loaded with dlopen at the first call: an unsharded deployment does not need it. */
#define INFX_RCCL_ID_BYTES 128
int32_t infx_rccl_unique_id(void* id128);
int32_t infx_set_shard_comm(infx_index* idx, const void* id128);
/* a communicator of its own for one stream (another ncclUniqueId; same call order on every rank): batches in flight on different streams then exchange
 * without serialising on one communicator.  A stream without one uses the index's. */
int32_t infx_stream_comm(infx_stream* s, const void* id128);
int32_t infx_comm_allreduce_sum_u32(infx_stream* s, void* buf /* device, in place */, uint64_t count);
int32_t infx_comm_allgather(infx_stream* s, const void* send /* device */, void* recv /* device: nranks x bytes_per_rank */, uint64_t bytes_per_rank);
/* Plumbing for a native host driver of the sharded phases (infidex_engine.h: infx_session_sharded_finish): per-stream device scratch buffers (slot < 16,
 * grown on demand, valid until the next call with the same slot), copies between host and device memory of either kind, stream synchronisation. */
int32_t infx_stream_scratch(infx_stream* s, int32_t slot, uint64_t bytes, void** out);
int32_t infx_stream_copy(infx_stream* s, void* dst, const void* src, uint64_t bytes);      /* stream-ordered; host pointers are staged; call infx_stream_wait before reading a host destination */
int32_t infx_stream_fill0(infx_stream* s, void* dev, uint64_t bytes);
/* After an all-gather of ONE packed block per rank (parts of part_bytes[p] bytes back to back, blocks `stride` bytes apart): dsts[p] receives the nranks
 * pieces of part p consecutively (rank-major), as separate all-gathers of the parts would have left them.  Device memory: a stream-ordered kernel; host: memcpy. */
int32_t infx_stream_unpack(infx_stream* s, const void* src, uint64_t stride, int32_t nranks, int32_t nparts /* <= 4 */, const uint64_t* part_bytes, void* const* dsts);
int32_t infx_stream_wait(infx_stream* s);
int32_t infx_stream_native(infx_stream* s, void** hip_stream);      /* the hipStream_t the stream's kernels, copies and collectives are ordered on */

int32_t infx_stream_create(infx_index* idx, infx_stream** out);
void    infx_stream_destroy(infx_stream* s);

/* ---- Stage 1 ------------------------------------------------------------------------------------------------- */
/* One term of a query, in Bm25Scorer order (ascending termId; fuzzy virtual terms first — VectorModel.cs:442). */
typedef struct infx_term {
    int32_t  term_id;      /* >= 0: index term; -1: virtual (fuzzy-union) term, postings in the batch's extra arrays */
    uint32_t extra_off;    /* virtual term: offset into infx_stage1_batch.extra_docs; its tf == 1 (RoaringPostingsEnum.cs:21) */
    uint32_t extra_len;
    float    idf;          /* Bm25Scorer.ComputeIdf on the host (MathF.Log), Bm25Scorer.cs:686-695 */
    float    max_score;    /* VectorModel.cs:525-531 */
    uint8_t  role;         /* INFX_ROLE_* bits: membership in the candidate tiers */
    uint8_t  rank;         /* disjunctive: position in the IDF-descending order among eligible terms (TieredCandidateSelector.cs:253) */
    uint16_t reserved;     /* virtual term only: 0 = extra_docs[extra_off..+len) are the union's doc ids (shard-local, ascending);
                              1 = they are the MEMBER TERM IDS (<= LD1 matches, VectorModel.cs:660-683): the union is formed on the
                              device inside the accumulate launch (one posting stream per member, each document counted once);
                              2 = extra_off is the index of a union materialised on the device by the last infx_union_build */
} infx_term;

#define INFX_ROLE_AND      1   /* AND mode: member of terms[0..n-2] (everything but the lowest-IDF term) */
#define INFX_ROLE_LOWEST   2   /* AND mode: the lowest-IDF term (dropped by Tier 1)                      */
#define INFX_ROLE_S1       4   /* AND mode: first selective term of Tier 2                               */
#define INFX_ROLE_S2       8   /* AND mode: second selective term of Tier 2                              */
#define INFX_ROLE_ELIGIBLE 16  /* disjunctive: not low-quality (idf >= 0.2*maxIdf)                       */
#define INFX_ROLE_LOWQ     32  /* disjunctive: low-quality term (only used when nothing selective hit)   */

#define INFX_MODE_PREFIX   1   /* candidates = accepted prefix DocSet alone (TieredCandidateSelector.cs:66-82)   */
#define INFX_MODE_DISJ     2   /* SelectCandidatesDisjunctive (:108-125, :243-322)                               */
#define INFX_MODE_AND      3   /* Tier 0/1/2 (:128-234)                                                          */

typedef struct infx_query {
    uint32_t term_off;     /* into the batch's terms[] */
    uint32_t num_terms;
    int32_t  mode;         /* INFX_MODE_* (decided on the host from global df / DocSet populations) */
    int32_t  prefix_set;   /* MODE_PREFIX: the candidate set; other modes: set of pre-"seen" docs (< min(2k,100) docs) or -1 */
    int32_t  depth;        /* topK handed to Bm25Scorer.Search == Query.CoverageDepth (SearchPipeline.cs:282) */
    int32_t  n_and;        /* AND mode: number of terms (n); needed for the Tier-0 / Tier-1 membership tests */
    int32_t  df_s1;        /* AND mode: global df of S1 / S2 (Tier-2 cardinality tests, :221) */
    int32_t  df_s2;
} infx_query;

typedef struct infx_hit { int32_t doc; float score; } infx_hit;   /* doc = global internal id; key via infx_upload_docs */

/* Per-query class counts of the candidate tiers on THIS shard (sum over shards before infx_stage1_select when sharded). */
#define INFX_NCLASS 136
typedef struct infx_counts { uint32_t c[INFX_NCLASS]; } infx_counts;

/* Stage 1a: stream postings, accumulate BM25+ in LDS, emit candidate supersets + class counts (device resident). */
int32_t infx_stage1_accumulate(infx_stream* s, uint32_t nq, const infx_query* q, uint32_t nterms, const infx_term* terms,
                               uint32_t extra_n, const int32_t* extra_docs, infx_counts* counts_out /* nq; host memory or a device buffer */);
/* Stage 1b: apply the tier rules with (global) counts, select the top-`depth` per query ordered by
 * (score desc, DocumentKey asc). out: nq*depth hits; out_count: nq. Replaces UpdateTopK/PriorityQueue (Bm25Scorer.cs:654-670). */
int32_t infx_stage1_select(infx_stream* s, uint32_t nq, const infx_counts* counts /* nq, host, global */,
                           infx_hit* out, uint32_t* out_count);
/* Convenience for the unsharded case: accumulate + select. == Bm25Scorer.Search for a batch of queries. */
int32_t infx_stage1_batch(infx_stream* s, uint32_t nq, const infx_query* q, uint32_t nterms, const infx_term* terms,
                          uint32_t extra_n, const int32_t* extra_docs, infx_hit* out, uint32_t* out_count);

/* Fuzzy virtual terms on the device (ExpandMissingTerm, VectorModel.cs:643-743): for v in [0, nv) the union of the doc sets of
 * members[member_offs[v] .. member_offs[v+1]) is materialised in HBM (ascending, shard-local ids) and counts_out[v] = its
 * cardinality on this shard (RoaringBitmap.Cardinality, :722-729; sum over shards for the global df — Exchange 1b). The unions stay
 * valid on this stream until the next infx_union_build and are referenced by infx_term {term_id -1, reserved 2, extra_off v}. */
int32_t infx_union_build(infx_stream* s, uint32_t nv, const uint32_t* member_offs, const int32_t* members, uint32_t* counts_out);

/* ---- Stage 2 ------------------------------------------------------------------------------------------------- */
/* CoverageQueryContext (Coverage/CoverageEngine.cs:9-43) of one query, prepared on the host (PrepareQuery :61-126). */
typedef struct infx_cov_query {
    uint16_t text[INFX_MAX_QUERY_CHARS];          /* normalised, lower-cased query */
    int32_t  text_len;
    int32_t  num_tokens;                          /* deduplicated, MinWordSize-filtered */
    uint16_t tok_off[INFX_MAX_QUERY_TOKENS];
    uint16_t tok_len[INFX_MAX_QUERY_TOKENS];
    float    term_idf[INFX_MAX_QUERY_TOKENS];     /* n-gram averaged (ComputeTermIdf :388-427) */
    float    word_idf[INFX_MAX_QUERY_TOKENS];     /* WordIdfCache (VectorModel.cs:864-908), 0 when absent */
    int32_t  has_word_idf;
    int32_t  num_fusion_tokens;                   /* unfiltered tokens (minWordSize 0), CoverageEngine.cs:348-352 */
    uint16_t ftok_off[INFX_MAX_QUERY_TOKENS * 2];
    uint16_t ftok_len[INFX_MAX_QUERY_TOKENS * 2];
    int32_t  lcs_tolerance;                       /* SearchPipeline.cs:498-500 */
    int32_t  reserved;                            /* 0, or 1 + the index of this query's record in the stream's long-query table (infx_stage2_long_queries): the
                                                     query is beyond the fast envelope, lives there, and the other members of this struct are ignored */
} infx_cov_query;
/* The same context for a query beyond INFX_MAX_QUERY_TOKENS distinct words / INFX_MAX_QUERY_CHARS characters (the reference has no limit: PrepareQuery rents
 * query.Length / 2 + 1 token slots, CoverageEngine.cs:68): up to INFX_LONGQ_TOKENS / INFX_LONGQ_CHARS.  Same members, larger tables. */
typedef struct infx_cov_query_long {
    uint16_t text[INFX_LONGQ_CHARS];
    int32_t  text_len;
    int32_t  num_tokens;
    uint16_t tok_off[INFX_LONGQ_TOKENS];
    uint16_t tok_len[INFX_LONGQ_TOKENS];
    float    term_idf[INFX_LONGQ_TOKENS];
    float    word_idf[INFX_LONGQ_TOKENS];
    int32_t  has_word_idf;
    int32_t  num_fusion_tokens;
    uint16_t ftok_off[INFX_LONGQ_TOKENS * 2];
    uint16_t ftok_len[INFX_LONGQ_TOKENS * 2];
    int32_t  lcs_tolerance;
    int32_t  reserved;
} infx_cov_query_long;

typedef struct infx_cov_cand {
    uint32_t query;        /* index into the batch's infx_cov_query[] */
    int32_t  doc;          /* shard-local internal id */
    float    base_score;   /* normBm25 = s / s_top1, or 0 for WordMatcher candidates (SearchPipeline.cs:380-414) */
    int32_t  want_lcs;     /* 1 for docIndex < 2 (quirk Q7, SearchPipeline.cs:492-503); 2: the same document's SECOND evaluation (its Stage-1 row after its WordMatcher
                              overlap row): the reference reads the LCS back from a byte span there, so a value above 255 comes back as 255 (quirk Q18, :494-503) */
} infx_cov_cand;

typedef struct infx_cov_out {
    float   score;         /* FusionScorer.Calculate: (float)precedence + semantic */
    uint8_t tiebreaker;
    uint8_t word_hits;     /* min(WordHits,255) */
    uint8_t lcs;           /* min(lcs,255) when want_lcs */
    uint8_t status;        /* 0 ok; INFX_EUNSUPPORTED: the over-long documents (> INFX_MAX_DOC_TOKENS words) of one launch exceeded the 64 MB token-table pool */
    int32_t word_hits_full;
} infx_cov_out;

/* feat_out (may be NULL): ncand x INFX_NFEAT ints — CoverageFeatures / FusionSignals (the bit-exact parity target; the
 * reference keeps them internal, Coverage/CoverageFeatures.cs:3-88), layout in DESIGN.md. */
/* Long queries of the NEXT Stage-2 call on this stream (infx_stage2_batch, infx_search_fused, infx_shard_stage2): n records, referred to by
 * infx_cov_query.reserved = 1 + index.  The table is consumed by that call; n = 0 clears it. */
int32_t infx_stage2_long_queries(infx_stream* s, uint32_t n, const infx_cov_query_long* q);
int32_t infx_stage2_batch(infx_stream* s, uint32_t nq, const infx_cov_query* q, uint32_t ncand,
                          const infx_cov_cand* cand, infx_cov_out* out, int32_t* feat_out);

/* ---- Infiscript post-filter + facet aggregation on device-resident columns (BASELINE config 5) ---------------------------------------
 * Replaces ResultProcessor.ApplyFilter (Scoring/ResultProcessor.cs:35-70: FilterVM.Execute per returned row, and over ALL documents for
 * Filter.NumberOfDocumentsInFilter on first use) and FacetBuilder.BuildFacets (Core/FacetBuilder.cs:19-105) for the rows a search returns.
 * A column = one non-indexed document field, dictionary-encoded by the host: codes[d] = index of document d's value among the field's
 * distinct values (GLOBAL internal ids: every shard holds the whole column, 4 B per document).
 * A filter = a postfix program over LEAVES; leaf l is a bitmap over the codes of column leaf.col (bit v = "the leaf's predicate holds for
 * distinct value v", evaluated by the host with FilterVM's coercion rules); three-valued evaluation F / T / N as in the reference's
 * untyped VM: AND(l,r) = l==F ? F : r, OR(l,r) = l==T ? T : r, NOT(x) = x==T ? F : T, TERN(c,a,b) = c==F ? b : a, LIT = N; match iff T. */
typedef struct infx_filter infx_filter;
#define INFX_FOP_LEAF 0
#define INFX_FOP_AND  1
#define INFX_FOP_OR   2
#define INFX_FOP_NOT  3
#define INFX_FOP_TERN 4
#define INFX_FOP_LIT  5
#define INFX_FILTER_MAX_OPS 256
#define INFX_FILTER_MAX_ROWS 64         /* post-filter / facets run on <= 64 returned rows per query (Query.MaxNumberOfRecordsToReturn) */
#define INFX_MAX_FACET_COLS 8
typedef struct infx_filter_op { uint32_t op; uint32_t arg; } infx_filter_op;                 /* arg: leaf index for INFX_FOP_LEAF */
typedef struct infx_filter_leaf { uint32_t col; uint32_t table_off; uint32_t num_values; uint32_t reserved; } infx_filter_leaf;   /* col 0xFFFFFFFF: no such field (null) */
int32_t infx_upload_column(infx_index* idx, uint32_t col, uint32_t total_docs, const uint32_t* codes, uint32_t num_values);
int32_t infx_filter_create(infx_index* idx, uint32_t nops, const infx_filter_op* ops, uint32_t nleaves, const infx_filter_leaf* leaves,
                           uint32_t ntable_words, const uint32_t* tables, infx_filter** out);
void    infx_filter_destroy(infx_filter* f);
/* Documents of THIS shard the filter accepts (sum over shards = Filter.NumberOfDocumentsInFilter). */
int32_t infx_filter_count(infx_stream* s, infx_filter* f, uint32_t* count);
/* Installs (f != NULL) or clears the post-filter and the facet columns of this stream: every following infx_search_fused /
 * infx_shard_finalize filters its result rows on the device before they are returned and counts the facet values of the kept rows. */
int32_t infx_stream_set_postfilter(infx_stream* s, infx_filter* f, uint32_t nfacet, const uint32_t* facet_cols);
/* Facets of the last search on this stream: for query q and facet column k, n_out[q*nfacet+k] pairs at codes_out / counts_out
 * [(q*nfacet+k)*INFX_FILTER_MAX_ROWS ..], in row order of first occurrence (the host orders them: count desc, value asc). */
int32_t infx_last_facets(infx_stream* s, uint32_t nq, uint32_t* codes_out, uint32_t* counts_out, uint32_t* n_out);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------------------ */
/* Durations (ms) of the last Stage-1 accumulate / select / Stage-2 launches on this stream, from HIP events recorded on
 * the stream the kernels ran on. */
/* ---- Whole batch on the device (unsharded index) -----------------------------------------------------------------
 * infx_search_fused runs SearchEngine.Search for a batch with ONE host synchronisation: accumulate -> tier rules -> select ->
 * coverage-stage candidate assembly (SearchPipeline.ExecuteCoverageStage, SearchPipeline.cs:330-420, incl. the WordMatcher
 * overlap / first-unique lookups of WordMatcher.Lookup, WordMatcher.cs:95-186) -> Stage 2 -> TopKHeap + ConsolidateSegments +
 * CalculateTruncationIndex (SearchPipeline.cs:422-560); only max_results rows per query return to the host.
 * The host supplies what needs the dictionaries: query terms / idf / roles (as for infx_stage1_accumulate), the prepared
 * coverage query, and per query the WordMatcher doc-id lists as ranges of the uploaded arrays. */
int32_t infx_upload_wordmatcher(infx_index* idx, uint64_t n_exact, const int32_t* exact_docs, uint64_t n_ld1, const int32_t* ld1_docs);

typedef struct infx_wm_list {      /* one ascending doc-id list */
    uint32_t src;                  /* 0: exact_docs, 1: ld1_docs (infx_upload_wordmatcher), 2: the call's `owned` buffer (affix matches) */
    uint32_t len;
    uint64_t off;
} infx_wm_list;

/* ---- Dictionary lookups of query planning on the device ------------------------------------------------------------
 * The reference resolves a query word to doc-id lists through three host dictionaries; uploaded once, the lookups run on the GPU:
 *   - WordMatcher.Lookup (WordMatcher/WordMatcher.cs:201-246): the word in the exact-word dictionary; for words of min_ld1..max_ld1 characters the word
 *     and each of its single-character deletions in the symmetric-delete dictionary and the exact dictionary;
 *   - WordMatcher.LookupAffix (:277-354): the words the query word is a prefix of, then those it is a suffix of — at most 4096 trie terms in that
 *     order — each contributing the LAST document that contains it (quirk Q13);
 *   - FstIndex.MatchWithinEditDistance1 (Indexing/Fst/FstIndex.cs:202-351): index terms with a suffix within Levenshtein distance 1 of the word.
 * Keys are UTF-16 strings in one arena per dictionary (key k = chars[key_offs[k] .. key_offs[k+1])); the library builds its own hash tables.
 * exact/ld1 list_offs index the doc-id arrays handed to infx_upload_wordmatcher (call that first).  affix_fwd / affix_rev: ids of the words of at
 * least min_ld1 characters, in ordinal order of the word / of the reversed word (the order the reference's trie walk meets them in). */
int32_t infx_upload_wm_dictionary(infx_index* idx,
                                  uint32_t n_exact_keys, const uint32_t* exact_key_offs /* n+1 */, const uint16_t* exact_chars, const uint64_t* exact_list_offs /* n+1 */,
                                  uint32_t n_ld1_keys, const uint32_t* ld1_key_offs, const uint16_t* ld1_chars, const uint64_t* ld1_list_offs,
                                  uint32_t n_words, const uint32_t* word_offs /* n+1 */, const uint16_t* word_chars, const int32_t* word_last_doc,
                                  uint32_t n_affix, const uint32_t* affix_fwd, const uint32_t* affix_rev, int32_t min_ld1, int32_t max_ld1);
/* Trie of the REVERSED index terms as CSR (node 0 = root; children of node v = edges edge_start[v] .. edge_start[v+1), ascending label), node_term[v] =
 * term id ending at v or -1; sorted_terms = all term ids in ordinal order of their text (FstIndex returns matches in that order). */
int32_t infx_upload_term_trie(infx_index* idx, uint32_t n_nodes, const uint32_t* edge_start /* n_nodes+1 */, const uint16_t* edge_label, const uint32_t* edge_child,
                              const int32_t* node_term, uint32_t n_terms, const uint32_t* sorted_terms);
/* FstIndex.MatchWithinEditDistance1 for a batch of words (word w = chars[word_offs[w] .. word_offs[w+1])): counts_out[w] = number of matching terms,
 * members_out[w*cap ..] = the first min(count, cap) of them in ordinal order (the reference keeps the first 1024, VectorModel.cs:660).
 * status_out[w]: 0 ok; 1 the walk outgrew the kernel's work lists (very short words fan out over the whole vocabulary) and 2 word length outside 1..64
 * (the reference switches to another walk, FstIndex.cs:362-440) — in both cases nothing was written for w and the caller expands it on the host. */
int32_t infx_ld1_expand(infx_stream* s, uint32_t nwords, const uint32_t* word_offs /* n+1 */, const uint16_t* chars, uint32_t cap,
                        int32_t* members_out /* nwords x cap */, uint32_t* counts_out, uint32_t* status_out);
/* infx_union_build with the expansion fused in: union v takes its members from members[member_offs[v] .. member_offs[v+1]) when word_of[v] < 0, or from the LD1
 * expansion of word word_of[v] (each word at most once; its member range must be empty).  k_ld1 closes the word unions' member ranges on the device and the union
 * kernels follow on the same stream: ONE wait for the device per batch (the counts) instead of two.  A word the kernel hands back (status != 0) yields an empty
 * union: expand it on the host and build again.  The expansions come back as from infx_ld1_expand. */
int32_t infx_union_build_ld1(infx_stream* s, uint32_t nv, const uint32_t* member_offs, const int32_t* members, const int32_t* word_of /* nv */,
                             uint32_t nwords, const uint32_t* word_offs /* n+1 */, const uint16_t* chars, uint32_t cap,
                             uint32_t* counts_out /* nv */, int32_t* ld1_members_out /* nwords x cap */, uint32_t* ld1_counts_out, uint32_t* ld1_status_out);
/* WordMatcher lists of ONE coverage query as the device produces them (parity tests): lists_out[INFX_MAX_WM_LISTS], *nlists_out of them; src 2 offsets
 * index owned_out (owned_cap ints, >= 4096 per word of the query). */
int32_t infx_wm_lookup_debug(infx_stream* s, const infx_cov_query* cq, infx_wm_list* lists_out, uint32_t* nlists_out, int32_t* owned_out, uint64_t owned_cap);

#define INFX_FQ_SKIP 1u            /* blank / unsupported query text: empty result */
#define INFX_FQ_SHORT 2u           /* 1..3 characters without delimiter (SearchPipeline.cs:108-120) */
#define INFX_FQ_SHORTSKIP 4u       /* short query whose prefix population exceeds 500: coverage is skipped */
#define INFX_FQ_COV 8u             /* coverage enabled (engine setup && Query.EnableCoverage) */
#define INFX_FQ_UNSUPPORTED 16u    /* result flag bit0 is set */
#define INFX_FQ_WMDEV 32u          /* the WordMatcher lists of this query are looked up on the device (infx_upload_wm_dictionary) from the words of its
                                      infx_cov_query; wm_off / wm_count are ignored (leave wm_count 0).  Admissible while words x (3 + 2*max_ld1) <= INFX_MAX_WM_LISTS */
#define INFX_MAX_WM_LISTS 256
typedef struct infx_fused_query {
    int32_t  dev;                  /* index into q[] of the Stage-1 part, or -1 (no index term) */
    uint32_t flags;                /* INFX_FQ_* */
    uint32_t wm_off, wm_count;     /* lists[wm_off .. +wm_count), non-empty lists only, <= INFX_MAX_WM_LISTS */
    int32_t  max_results;          /* Query.MaxNumberOfRecordsToReturn */
    int32_t  reserved;             /* set by the library */
} infx_fused_query;

/* nd Stage-1 queries q[] (+ terms), nq search queries fq[]/cq[] (nq >= nd; cq[i] is ignored unless coverage runs for i).
 * All queries must share one depth.  out_*: nq x max_results rows; out_flags bits as infx_engine_search_batch. */
int32_t infx_search_fused(infx_stream* s, uint32_t nd, const infx_query* q, uint32_t nterms, const infx_term* terms,
                          uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq,
                          uint32_t nlists, const infx_wm_list* lists, uint32_t owned_n, const int32_t* owned,
                          int32_t depth, int32_t max_results, int32_t want_debug,
                          int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags);
/* Document-sharded operation: the same device stages, cut where the collectives go. Rows carry GLOBAL internal ids.
 *   infx_stage1_accumulate (local) -> all-reduce(sum) of the class histograms
 *   infx_shard_select   : tier rules from the GLOBAL counts + local first-pass top-`depth` -> all-gather; infx_shard_replay_* (exact cut) -> all-gather of hits / hit counts (RCCL)
 *   infx_shard_stage2   : merge of the nshards lists, candidate assembly, Stage 2 on the rows whose document this shard holds
 *                         (all other rows are zero)                                  -> all-reduce(sum) of outs (nq x 2*depth x 12 B)
 *   infx_shard_finalize : final ordering / truncation from the merged rows (identical on every rank).
 * The exchange buffers (hits_out / hitcount_out, all_hits / all_hitcounts, outs_out, merged_outs) may be host OR device memory: device
 * pointers (e.g. the tensors a torch.distributed / RCCL collective works on) are copied device-to-device, nothing crosses PCIe.
 * Every shard needs the GLOBAL DocumentKey table (infx_upload_doc_keys_all) and the WordMatcher lists (global ids). */
int32_t infx_upload_doc_keys_all(infx_index* idx, uint32_t total_docs, const int64_t* keys);
int32_t infx_shard_select(infx_stream* s, uint32_t nd, const infx_counts* global_counts, int32_t depth, infx_hit* hits_out, uint32_t* hitcount_out,
                          float* next_out /* nd: best first-pass score this shard left out of its list (0: none); NULL = not wanted */);
/* The reference's Stage-1 cut across document shards, bit for bit (Bm25Scorer.cs:195-329 chunks, :350-368 MaxScore skips, :654-670 UpdateTopK on the
 * BCL heap): shards must begin at multiples of 65 536 documents (whole Roaring containers), so the reference's sequential walk is the shards' walks one
 * after the other, coupled only through the heap.  Between infx_shard_select and infx_shard_stage2:
 *   infx_shard_select (hits, counts, next)                                 -> all-gather of the three                                [Exchange 2a]
 *   infx_shard_replay_local : global ambiguity test (identical on every rank) + this shard's chunks: exact scores, candidates the heap could
 *                             still take, validity intervals of the MaxScore skips; *blob_bytes = size of the packed result
 *   infx_shard_replay_blob  : copies it to host or device memory             -> all-gather of the blobs, padded to the largest        [Exchange 2c]
 *   infx_shard_replay_merge : the OWNER of a flagged query (query mod nshards) replays UpdateTopK over shard 0's chunks, then shard 1's, ...;
 *                             hits_out / hitcount_out = this rank's contribution to the final lists: the exact list for owned flagged queries,
 *                             nothing for other flagged queries, the first-pass list otherwise                                      -> all-gather [Exchange 2b]
 *                             hitcount 0xFFFFFFFF: the owner could not certify the parallel replay (rare) -> infx_shard_replay_chain
 *   infx_shard_replay_chain : the literal sequential replay of the queries with need[q] != 0, continued from shard to shard: state = nd x (2 + 2*depth)
 *                             words {n, threshold bits, doc[depth], score bits[depth]} (heap array order), zero for shard 0; pass the output of shard r
 *                             to shard r+1; the last shard's state is the reference's final heap.  depth == infx_config.max_depth. */
int32_t infx_shard_replay_local(infx_stream* s, int32_t nshards, uint32_t nd, const infx_hit* all_hits /* nshards x nd x depth */, const uint32_t* all_hitcounts,
                                const float* all_next, int32_t depth, uint64_t* blob_bytes);
int32_t infx_shard_replay_blob(infx_stream* s, void* dst, uint64_t padded_bytes);
int32_t infx_shard_replay_merge(infx_stream* s, int32_t nshards, uint32_t nd, const void* all_blobs /* nshards x padded_bytes */, uint64_t padded_bytes, int32_t depth,
                                infx_hit* hits_out, uint32_t* hitcount_out);
int32_t infx_shard_replay_chain(infx_stream* s, uint32_t nd, const uint32_t* need, int32_t depth, void* state);
int32_t infx_shard_stage2(infx_stream* s, int32_t nshards, uint32_t nd, const infx_hit* all_hits /* nshards x nd x depth */, const uint32_t* all_hitcounts,
                          uint32_t nq, const infx_fused_query* fq, const infx_cov_query* cq, uint32_t nlists, const infx_wm_list* lists,
                          uint32_t owned_n, const int32_t* owned, int32_t depth, int32_t max_results, int32_t want_debug, infx_cov_out* outs_out);
int32_t infx_shard_finalize(infx_stream* s, uint32_t nq, const infx_cov_out* merged_outs, int32_t depth, int32_t max_results,
                            int64_t* out_keys, float* out_scores, uint8_t* out_ties, uint32_t* out_counts, uint32_t* out_flags);

/* After an infx_search_fused call with want_debug != 0: the intermediate device results (parity tests / introspection).
 * s1: nq x depth (ConsolidateSegments order) + counts; cands/outs/feat: nq x 2*depth rows, cand_counts[i] of them valid;
 * idx01: nq x 2 (documents with docIndex 0 / 1, -1 if absent).  Any pointer may be NULL. */
int32_t infx_fused_debug(infx_stream* s, infx_hit* s1, uint32_t* s1_counts, infx_cov_cand* cands, infx_cov_out* outs, int32_t* feat,
                         uint32_t* cand_counts, uint32_t* run_cov, int32_t* idx01);
/* kernel durations of the last fused call: accumulate, rules+select, prep2, stage2, finalize */
int32_t infx_last_fused_timings(infx_stream* s, float* ms5);
/* totals of the last fused call: Stage-1 rows kept, Stage-2 candidate rows scored, UTF-16 bytes of their texts */
int32_t infx_last_fused_stats(infx_stream* s, uint64_t* s1_rows, uint64_t* stage2_rows, uint64_t* stage2_text_bytes);

int32_t infx_last_timings(infx_stream* s, float* accumulate_ms, float* select_ms, float* stage2_ms);
/* Bytes the last accumulate launch actually streamed (posting slices of the doc ranges that held candidates: 5 B/posting,
 * 4 B for virtual terms, + 12 B per emitted arena entry) — an implementation figure, <= the algorithmic bytes of SURVEY 8(d). */
int32_t infx_last_alg_bytes(infx_stream* s, uint64_t* bytes);
/* Sum over the batch of card(C_q): candidates the tier rules let through (upper bound +128/query in disjunctive mode). */
int32_t infx_last_candidates(infx_stream* s, uint64_t* n);
/* Queries of the last batch whose Stage-1 cut was ambiguous and was replayed with the reference's sequential semantics (k_exact1). */
int32_t infx_last_exact_replays(infx_stream* s, uint32_t* n);
/* The replay of the last batch: duration of its kernels (ms, HIP events on the stream) and why k_select flagged the queries:
 * why3[0] exact plateau (equal first-pass scores on both sides of the cut), [1] the best row left out lies inside the rounding band below the cut,
 * [2] the best row left out was not gathered (flagged conservatively). */
int32_t infx_last_replay_stats(infx_stream* s, float* ms, uint32_t* why3);
/* ... and its parts: the scan (k_ex_walk x2 + k_ex_prefix + k_ex_theta), k_ex_chunk (three launches), k_ex_heap, k_exact1 (ms, HIP events on the stream). */
int32_t infx_last_replay_breakdown(infx_stream* s, float* ms4);

#ifdef __cplusplus
}
#endif
#endif
