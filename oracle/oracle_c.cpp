// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// C ABI over the CPU restatement so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// drive it through ctypes. Nothing under infidex_amd/ may link or load this library.
#include "pipeline.hpp"
#include "filter.hpp"
#include <sstream>
#include <chrono>
#include <thread>
#include <atomic>

using namespace orc;

namespace {
constexpr int NFEAT = 32;
void pack_features(const CoverageFeatures& f, int lcs, int32_t* o) {
    // integer "coverage counts" (bit-exact parity target) — see include/infidex_hip.h infx_cov_out
    o[0] = f.CoverageScore; o[1] = f.TermsCount; o[2] = f.TermsWithAnyMatch; o[3] = f.TermsFullyMatched;
    o[4] = f.TermsStrictMatched; o[5] = f.TermsPrefixMatched; o[6] = f.FirstMatchIndex; o[7] = f.WordHits;
    o[8] = f.DocTokenCount; o[9] = f.LongestPrefixRun; o[10] = f.SuffixPrefixRun; o[11] = f.PhraseSpan;
    o[12] = f.PrecedingStrictCount; o[13] = f.LastTokenHasPrefix; o[14] = f.LastTermIsTypeAhead;
    o[15] = f.Fusion.UnfilteredQueryTokenCount; o[16] = f.Fusion.LexicalPrefixLast; o[17] = f.Fusion.AllPrecedingExact;
    o[18] = f.Fusion.IsPerfectDocLexical; o[19] = f.Fusion.HasStemEvidence; o[20] = f.Fusion.HasAnchorStem;
    o[21] = f.Fusion.TrailingMatchDensity; o[22] = f.Fusion.SingleTermLexicalSim; o[23] = f.Fusion.SingleCharLastTokenBoost;
    o[24] = lcs;
    auto fbits = [](float x) { int32_t b; std::memcpy(&b, &x, 4); return b; };
    o[25] = fbits(f.SumCi); o[26] = fbits(f.IdfCoverage); o[27] = fbits(f.TotalIdf); o[28] = fbits(f.MissingIdf);
    o[29] = fbits(f.LastTermCi); o[30] = fbits(f.WeightedCoverage); o[31] = 0;
}
struct Column { std::string name; bool facetable; std::vector<flt::Value> vals; };     // a non-indexed document field (DocumentFields), by internal doc id
struct Handle {
    Engine eng;
    SearchOutput last;
    std::vector<Column> cols;
    std::map<std::string, int> filterCount;       // Filter.NumberOfDocumentsInFilter, computed on first use (ResultProcessor.cs:39-54)
    std::string lastFacets; int lastInFilter = 0;
    explicit Handle(const Config& c) : eng(c) {}
    flt::Fields fields_of(int doc) const { flt::Fields f; for (auto& c : cols) if ((size_t)doc < c.vals.size()) f[c.name] = c.vals[doc]; return f; }
};
std::string json_escape(const std::string& s) { std::string o; for (char c : s) { if (c == '"' || c == '\\') o.push_back('\\'); o.push_back(c); } return o; }
}

extern "C" {

void* orc_create(int enable_coverage, int word_matcher, int stop_term_limit) {
    Config c; c.enableCoverage = enable_coverage != 0; c.wordMatcher = word_matcher != 0;
    if (stop_term_limit > 0) c.stopTermLimit = stop_term_limit;
    return new Handle(c);
}
void orc_destroy(void* h) { delete (Handle*)h; }
// SynonymMap.AddSynonym (before any document is added)
void orc_add_synonym(void* h, const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb) {
    ((Handle*)h)->eng.ix.syn.add(uview((const u16*)a, la), uview((const u16*)b, lb));
}

void orc_add_document(void* h, int64_t key, int nfields, const uint16_t* const* texts, const int32_t* lens, const int32_t* weights) {
    std::vector<FieldIn> f;
    for (int i = 0; i < nfields; i++) f.push_back({ustr((const u16*)texts[i], lens[i]), weights[i]});
    ((Handle*)h)->eng.add_document(key, f);
}
// bulk: fieldCount fields per doc, laid out doc-major; offs has n*fieldCount+1 entries into arena
void orc_add_documents_flat(void* h, int64_t n, const int64_t* keys, const uint16_t* arena, const uint64_t* offs,
                            int field_count, const int32_t* field_weights) {
    Handle* H = (Handle*)h;
    std::vector<FieldIn> f(field_count);
    for (int64_t d = 0; d < n; d++) {
        for (int k = 0; k < field_count; k++) {
            uint64_t a = offs[d * field_count + k], b = offs[d * field_count + k + 1];
            f[k].text.assign((const u16*)arena + a, b - a);
            f[k].weight = field_weights[k];
        }
        H->eng.add_document(keys ? keys[d] : d, f);
    }
}
void orc_finalize(void* h) { ((Handle*)h)->eng.finalize(); }
void orc_set_trace(void* h, int on) { ((Handle*)h)->eng.keepTrace = on != 0; }

// flags: bit0 = unsupported (short-query path), bit1 = coverage stage ran
int32_t orc_search(void* h, const uint16_t* q, int32_t qlen, int32_t max_results, int32_t depth, int32_t enable_cov,
                   int64_t* keys, float* scores, uint8_t* ties, int32_t cap, int32_t* flags) {
    Handle* H = (Handle*)h;
    QueryParams qp; qp.maxResults = max_results; qp.coverageDepth = depth; qp.enableCoverage = enable_cov != 0;
    H->last = H->eng.search(uview((const u16*)q, qlen), qp);
    int n = std::min<int>(cap, (int)H->last.records.size());
    for (int i = 0; i < n; i++) { keys[i] = H->last.records[i].key; scores[i] = H->last.records[i].score; ties[i] = H->last.records[i].tie; }
    if (flags) *flags = (H->last.unsupported ? 1 : 0) | (H->last.usedCoverage ? 2 : 0);
    return n;
}
int32_t orc_last_stage1(void* h, int64_t* keys, float* scores, int32_t cap) {
    Handle* H = (Handle*)h;
    int n = std::min<int>(cap, (int)H->last.stage1.size());
    for (int i = 0; i < n; i++) { keys[i] = H->last.stage1[i].key; scores[i] = H->last.stage1[i].score; }
    return n;
}
int32_t orc_feature_count() { return NFEAT; }
int32_t orc_last_trace(void* h, int32_t* internal_ids, float* base, float* scores, uint8_t* ties, int32_t* feat, int32_t cap) {
    Handle* H = (Handle*)h;
    int n = std::min<int>(cap, (int)H->last.trace.size());
    for (int i = 0; i < n; i++) {
        auto& t = H->last.trace[i];
        internal_ids[i] = t.internalId; base[i] = t.baseScore; scores[i] = t.score; ties[i] = t.tie;
        pack_features(t.f, t.lcs, feat + (size_t)i * NFEAT);
    }
    return n;
}
// last Stage-1 term list (ascending termId order as fed to Bm25Scorer): termId (-1 fuzzy), df, idf, maxScore
int32_t orc_last_terms(void* h, int32_t* term_ids, int32_t* dfs, float* idfs, float* maxs, int32_t cap) {
    Handle* H = (Handle*)h;
    auto& lt = H->eng.s1->lastTerms;
    int n = std::min<int>(cap, (int)lt.size());
    for (int i = 0; i < n; i++) { term_ids[i] = lt[i].termId; dfs[i] = lt[i].df; idfs[i] = lt[i].idf; maxs[i] = lt[i].maxScore; }
    return n;
}
void orc_last_stats(void* h, int64_t* out3) {
    Handle* H = (Handle*)h;
    out3[0] = H->eng.s1->stats.candidates; out3[1] = H->eng.s1->stats.postingsTouched; out3[2] = H->eng.s1->stats.mode;
}

// ---- index introspection (array-level parity with the product's builder) -----------------------
int64_t orc_num_docs(void* h) { return ((Handle*)h)->eng.ix.N; }
int64_t orc_num_terms(void* h) { return (int64_t)((Handle*)h)->eng.ix.termText.size(); }
int64_t orc_num_postings(void* h) { return (int64_t)((Handle*)h)->eng.ix.postDocFlat.size(); }
float orc_avgdl(void* h) { return ((Handle*)h)->eng.ix.avgdl; }
void orc_export_index(void* h, int32_t* df, uint64_t* post_off, int32_t* post_doc, uint8_t* post_w, float* doc_len) {
    Index& ix = ((Handle*)h)->eng.ix;
    size_t T = ix.termText.size();
    if (df) std::memcpy(df, ix.termDf.data(), T * 4);
    if (post_off) std::memcpy(post_off, ix.postOff.data(), (T + 1) * 8);
    if (post_doc) std::memcpy(post_doc, ix.postDocFlat.data(), ix.postDocFlat.size() * 4);
    if (post_w) std::memcpy(post_w, ix.postWFlat.data(), ix.postWFlat.size());
    if (doc_len) std::memcpy(doc_len, ix.docLen.data(), (size_t)ix.N * 4);
}
// term text of id t -> returns length, copies up to cap units
int32_t orc_term_text(void* h, int32_t t, uint16_t* out, int32_t cap) {
    Index& ix = ((Handle*)h)->eng.ix;
    const ustr& s = ix.termText[t];
    int n = std::min<int>(cap, (int)s.size());
    std::memcpy(out, s.data(), (size_t)n * 2);
    return (int)s.size();
}
int32_t orc_term_id(void* h, const uint16_t* s, int32_t len) { return ((Handle*)h)->eng.ix.get_term(uview((const u16*)s, len)); }
int32_t orc_prefix_docset(void* h, const uint16_t* p, int32_t len, int32_t* out, int32_t cap) {
    auto* v = ((Handle*)h)->eng.ix.prefix_docset(uview((const u16*)p, len));
    if (!v) return 0;
    int n = std::min<int>(cap, (int)v->size());
    if (out) std::memcpy(out, v->data(), (size_t)n * 4);
    return (int)v->size();
}
int32_t orc_match_ld1(void* h, const uint16_t* q, int32_t len, int32_t* out, int32_t cap) {
    std::vector<int> m;
    int c = ((Handle*)h)->eng.ix.match_ld1(uview((const u16*)q, len), m, cap);
    for (size_t i = 0; i < m.size(); i++) out[i] = m[i];
    return c;
}
int32_t orc_wordmatcher(void* h, const uint16_t* q, int32_t len, int32_t* out, int32_t cap) {
    std::vector<int32_t> r;
    ((Handle*)h)->eng.wm.execute(uview((const u16*)q, len), true, r);
    int n = std::min<int>(cap, (int)r.size());
    if (out) std::memcpy(out, r.data(), (size_t)n * 4);
    return (int)r.size();
}
int32_t orc_wm_lookup(void* h, const uint16_t* q, int32_t len, int affix, int32_t* out, int32_t cap) {
    std::vector<int32_t> r;
    bool ok = affix ? ((Handle*)h)->eng.wm.lookup_affix(uview((const u16*)q, len), r) : ((Handle*)h)->eng.wm.lookup(uview((const u16*)q, len), r);
    if (!ok) return -1;
    int n = std::min<int>(cap, (int)r.size());
    if (out) std::memcpy(out, r.data(), (size_t)n * 4);
    return (int)r.size();
}

// ---- primitives (unit KATs) -----------------------------------------------------------------------
int32_t orc_levenshtein(const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb, int32_t max_err, int32_t ic) {
    return lev_calculate(uview((const u16*)a, la), uview((const u16*)b, lb), max_err, ic != 0);
}
int32_t orc_damerau(const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb, int32_t max_d, int32_t ic) {
    return lev_damerau(uview((const u16*)a, la), uview((const u16*)b, lb), max_d, ic != 0);
}
int32_t orc_lcs(const uint16_t* a, int32_t la, const uint16_t* b, int32_t lb, int32_t tol) {
    return lcs_metric(uview((const u16*)a, la), uview((const u16*)b, lb), tol);
}
// the generated case tables as the oracle uses them: 65536 entries each (lower, upper: code units; letter: 0 / 1)
int32_t orc_case_tables(uint16_t* lower, uint16_t* upper, uint8_t* letter) {
    for (int c = 0; c < 65536; c++) { lower[c] = to_lower_inv((u16)c); upper[c] = to_upper_inv((u16)c); letter[c] = is_letter((u16)c) ? 1 : 0; }
    return 65536;
}
int32_t orc_normalize(const uint16_t* s, int32_t len, int lower, uint16_t* out, int32_t cap) {
    ustr r = default_normalizer().normalize(uview((const u16*)s, len));
    if (lower) r = to_lower_inv(r);
    int n = std::min<int>(cap, (int)r.size());
    std::memcpy(out, r.data(), (size_t)n * 2);
    return (int)r.size();
}
// CoverageEngine without an index (CoverageEngineTests.cs): returns coverage byte; feat gets NFEAT ints; also fusion score
// word_idf: optional fixed word-level IDF cache (BugReproductionTests.cs:13-67): n_idf entries of (utf16 word, len, idf)
int32_t orc_coverage_standalone(const uint16_t* q, int32_t ql, const uint16_t* d, int32_t dl, double lcs_sum, float bm25,
                                int32_t* feat, float* score, uint8_t* tie,
                                int32_t n_idf, const uint16_t* idf_words, const uint64_t* idf_offs, const float* idf_vals) {
    CoverageEngine ce;
    std::unordered_map<ustr, float, UHash> fixed;
    if (n_idf > 0) {
        for (int i = 0; i < n_idf; i++) {
            ustr k((const u16*)idf_words + idf_offs[i], (size_t)(idf_offs[i + 1] - idf_offs[i]));
            for (auto& ch : k) ch = to_upper_inv(ch);
            fixed[k] = idf_vals[i];
        }
        ce.fixedWordIdf = &fixed;
    }
    QueryContext ctx = ce.prepare_query(uview((const u16*)q, ql));
    CoverageFeatures f = ce.calculate_features(ctx, uview((const u16*)d, dl), lcs_sum);
    if (feat) pack_features(f, (int)lcs_sum, feat);
    auto sc = fusion_calculate(ctx.query, uview((const u16*)d, dl), f, bm25);
    if (score) *score = sc.first;
    if (tie) *tie = sc.second;
    return f.CoverageScore;
}

// per-stage nanoseconds summed over all threads since the last reset (stage1.hpp StageClock): planning, candidate selection, BM25+ scoring, WordMatcher, coverage
void orc_stage_times(double* out_ms5, int32_t reset) {
    for (int i = 0; i < 5; i++) { out_ms5[i] = (double)stage_ns(i).load() * 1e-6; if (reset) stage_ns(i).store(0); }
}

// ---- timed batch for bench.py's cpu_baseline leg ---------------------------------------------------
// Runs nq queries (concatenated UTF-16 with offsets) on `threads` threads (one in-flight query per thread, each
// with its own Stage1 scratch — the reference's concurrent-reader model, SearchEngine.cs:258). Returns seconds.
double orc_timed_batch(void* h, int32_t nq, const uint16_t* arena, const uint64_t* offs, int32_t max_results, int32_t depth,
                       int32_t threads, int64_t* out_keys /* nq*max_results, -1 padded */, double* lat_ms /* nq or null */) {
    Handle* H = (Handle*)h;
    Engine& E = H->eng;
    std::atomic<int> next(0);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        Stage1 s1(E.ix);
        // private engine view: share index/wm/cov, own Stage1
        while (true) {
            int i = next.fetch_add(1);
            if (i >= nq) break;
            auto a = std::chrono::steady_clock::now();
            QueryParams qp; qp.maxResults = max_results; qp.coverageDepth = depth;
            SearchOutput o;
            {
                // Engine::search uses E.s1; emulate with a thread-local Stage1 by swapping pointers is racy, so inline:
                uview raw((const u16*)arena + offs[i], (size_t)(offs[i + 1] - offs[i]));
                size_t b = 0, e = raw.size();
                while (b < e && is_whitespace(raw[b])) b++;
                while (e > b && is_whitespace(raw[e - 1])) e--;
                ustr q = to_lower_inv(default_normalizer().normalize(raw.substr(b, e - b)));
                o = E.search_with(s1, q, qp);
            }
            if (out_keys) for (int k = 0; k < max_results; k++) out_keys[(size_t)i * max_results + k] = k < (int)o.records.size() ? o.records[k].key : -1;
            if (lat_ms) lat_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        }
    };
    if (threads <= 1) worker();
    else { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(worker); for (auto& t : th) t.join(); }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}


// ---- Infiscript filter / facets (config 5) ------------------------------------------------------------------------------
// Evaluates `expr` against one document given as parallel field arrays. kinds: 0 null, 1 int64, 2 double, 3 string (UTF-8).
// Returns 1 / 0, -1 on a parse error, -2 for a construct the oracle does not restate.
int32_t orc_filter_eval(const char* expr, int32_t nfields, const char* const* names, const int32_t* kinds, const int64_t* ints, const double* dbls, const char* const* strs) {
    try {
        flt::Compiled c = flt::compile(flt::parse(expr));
        flt::Fields f;
        for (int i = 0; i < nfields; i++) {
            flt::Value v;
            if (kinds[i] == 1) v = flt::Value::integer(ints[i]); else if (kinds[i] == 2) v = flt::Value::num(dbls[i]); else if (kinds[i] == 3) v = flt::Value::str(strs[i]);
            f[names[i]] = v;
        }
        return flt::execute(c, f) ? 1 : 0;
    } catch (const flt::ParseError& e) { return std::string(e.what()).find("not restated") != std::string::npos ? -2 : -1; }
      catch (const std::exception&) { return -2; }
}
int32_t orc_double_to_string(double x, char* out, int32_t cap) { std::string s = flt::dbl_to_string(x); snprintf(out, (size_t)cap, "%s", s.c_str()); return (int32_t)s.size(); }
// Document.Deleted for every document whose DocumentKey is listed (DocumentCollection.DeleteDocumentsByKey, Core/DocumentCollection.cs:200-212)
// without the Count bookkeeping: index statistics stay as indexed.  Returns the number of documents newly marked.
int32_t orc_delete_keys(void* h, const int64_t* keys, int64_t n) {
    Handle* H = (Handle*)h; Index& ix = H->eng.ix; if (ix.deleted.empty()) ix.deleted.assign((size_t)ix.N, 0);
    std::unordered_set<int64_t> ks(keys, keys + n); int32_t c = 0;
    for (int d = 0; d < ix.N; d++) if (!ix.deleted[d] && ks.count(ix.docKey[d])) { ix.deleted[d] = 1; c++; }
    H->filterCount.clear();      // a Filter object parsed after the deletion counts again (NumberOfDocumentsInFilter lives on the Filter instance, Api/Filter.cs:17)
    return c;
}
void orc_restore_all(void* h) { Handle* H = (Handle*)h; H->eng.ix.deleted.clear(); H->filterCount.clear(); }     // clears every Deleted flag (test fixture reuse)
// One column of non-indexed document fields, by internal doc id. kind: 1 int64 (vals_i), 2 double (vals_d), 3 string (arena + offs, UTF-8)
void orc_set_column(void* h, const char* name, int32_t kind, int32_t facetable, int64_t n, const int64_t* vals_i, const double* vals_d, const char* arena, const uint64_t* offs) {
    Handle* H = (Handle*)h; Column c; c.name = name; c.facetable = facetable != 0; c.vals.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (kind == 1) c.vals[i] = flt::Value::integer(vals_i[i]); else if (kind == 2) c.vals[i] = flt::Value::num(vals_d[i]);
        else c.vals[i] = flt::Value::str(std::string(arena + offs[i], arena + offs[i + 1]));
    }
    H->filterCount.clear();
    for (auto& x : H->cols) if (x.name == c.name) { x = c; return; }
    H->cols.push_back(std::move(c));
}
// SearchEngine.Search with Query.Filter / Query.EnableFacets (SearchEngine.cs:298-316): Execute(max) -> ApplyFilter -> facets -> Take(max).
// filter may be NULL.  Returns the row count, or -1 / -2 as orc_filter_eval.
int32_t orc_search_filtered(void* h, const uint16_t* q, int32_t qlen, int32_t max_results, int32_t depth, int32_t enable_cov, const char* filter, int32_t enable_facets,
                            int64_t* keys, float* scores, uint8_t* ties, int32_t cap, int32_t* flags, int32_t* n_in_filter) {
    Handle* H = (Handle*)h;
    QueryParams qp; qp.maxResults = max_results; qp.coverageDepth = depth; qp.enableCoverage = enable_cov != 0;
    H->last = H->eng.search(uview((const u16*)q, qlen), qp);
    std::vector<ScoreEntry> rows(H->last.records.begin(), H->last.records.end());
    H->lastInFilter = 0; H->lastFacets = "{}";
    auto doc_of = [&](int64_t key) -> int { auto it = H->eng.ix.keyToFirstId.find(key); return it == H->eng.ix.keyToFirstId.end() ? -1 : it->second; };
    if (filter) {
        flt::Compiled c;
        try { c = flt::compile(flt::parse(filter)); }
        catch (const flt::ParseError& e) { return std::string(e.what()).find("not restated") != std::string::npos ? -2 : -1; }
        auto fc = H->filterCount.find(filter);
        if (fc == H->filterCount.end() || fc->second == 0) {                 // NumberOfDocumentsInFilter == 0 -> count over ALL documents
            // over DocumentCollection.GetAllDocuments() == the documents that are not Deleted (Core/DocumentCollection.cs:216-219)
            int m = 0; for (int d = 0; d < H->eng.ix.N; d++) if (!H->eng.ix.is_deleted(d) && flt::execute(c, H->fields_of(d))) m++;
            H->filterCount[filter] = m;
        }
        H->lastInFilter = H->filterCount[filter];
        std::vector<ScoreEntry> kept;
        for (auto& r : rows) { int d = doc_of(r.key); if (d < 0) continue; if (flt::execute(c, H->fields_of(d))) kept.push_back(r); }
        rows.swap(kept);
    }
    if (enable_facets && !rows.empty()) {
        std::vector<flt::Fields> fs; std::vector<const flt::Fields*> ps;
        for (auto& r : rows) { int d = doc_of(r.key); if (d >= 0) fs.push_back(H->fields_of(d)); }
        for (auto& f : fs) ps.push_back(&f);
        std::ostringstream o; o << "{"; bool first = true;
        for (auto& col : H->cols) if (col.facetable) {
            auto fc = flt::facet(ps, col.name);
            if (fc.empty()) continue;
            if (!first) o << ","; first = false;
            o << "\"" << json_escape(col.name) << "\":[";
            for (size_t i = 0; i < fc.size(); i++) { if (i) o << ","; o << "[\"" << json_escape(fc[i].first) << "\"," << fc[i].second << "]"; }
            o << "]";
        }
        o << "}"; H->lastFacets = o.str();
    }
    if (n_in_filter) *n_in_filter = H->lastInFilter;
    int n = std::min<int>(std::min<int>(cap, max_results), (int)rows.size());
    for (int i = 0; i < n; i++) { keys[i] = rows[i].key; scores[i] = rows[i].score; ties[i] = rows[i].tie; }
    if (flags) *flags = (H->last.unsupported ? 1 : 0) | (H->last.usedCoverage ? 2 : 0);
    return n;
}
int32_t orc_last_facets_json(void* h, char* out, int32_t cap) { Handle* H = (Handle*)h; snprintf(out, (size_t)cap, "%s", H->lastFacets.c_str()); return (int32_t)H->lastFacets.size(); }

} // extern "C"