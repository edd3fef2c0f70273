// Host-side query preparation of the product (what the reference does on the managed heap before / between the two
// accelerated seams). Produces the flat records of include/infidex_hip.h.
//   VectorModel.SearchWithMaxScore        Indexing/VectorModel.cs:376-602   -> plan_stage1
//   FstIndex.MatchWithinEditDistance1     Indexing/Fst/FstIndex.cs:202-351  -> match_ld1 (same result set and order,
//        computed with a pruned trie walk instead of visiting every node to depth m+1)
//   TieredCandidateSelector (host-decidable part: prefix precedence, mode, tier roles)  Scoring/TieredCandidateSelector.cs:53-237
//   WordMatcherLookup.Execute / WordMatcher.Lookup / LookupAffix   Scoring/WordMatcherLookup.cs, WordMatcher/WordMatcher.cs:201-354
//   CoverageEngine.PrepareQuery           Coverage/CoverageEngine.cs:61-126,388-427
//   List<T>.Sort semantics for the IDF ordering (BCL introsort; unstable above 16 elements) -> bcl_sort
#pragma once
#include "index.h"
#include "../../../include/infidex_hip.h"
#include <list>
#include <mutex>
#include <unordered_map>
#include <memory>
#include <queue>
#include <chrono>

namespace infx {

// ---- BCL ArraySortHelper<T>.IntrospectiveSort (Comparison<T>) ------------------------------------------------------------
// The reference orders terms with List<T>.Sort((a,b) => b.Idf.CompareTo(a.Idf)); equal-IDF terms end up in the order this
// algorithm leaves them in, and that order decides which terms the 100*topK early stop / Tier 1 / Tier 2 pick.
template <class T, class Cmp> struct BclSort {
    T* k; Cmp cmp;
    void sig(int i, int j) { if (cmp(k[i], k[j]) > 0) std::swap(k[i], k[j]); }
    void ins(int lo, int n) { for (int i = 0; i < n - 1; i++) { T t = k[lo + i + 1]; int j = i; while (j >= 0 && cmp(t, k[lo + j]) < 0) { k[lo + j + 1] = k[lo + j]; j--; } k[lo + j + 1] = t; } }
    void down(int lo, int i, int n) { T d = k[lo + i - 1]; while (i <= n / 2) { int c = 2 * i; if (c < n && cmp(k[lo + c - 1], k[lo + c]) < 0) c++; if (!(cmp(d, k[lo + c - 1]) < 0)) break; k[lo + i - 1] = k[lo + c - 1]; i = c; } k[lo + i - 1] = d; }
    void heap(int lo, int n) { for (int i = n / 2; i >= 1; i--) down(lo, i, n); for (int i = n; i > 1; i--) { std::swap(k[lo], k[lo + i - 1]); down(lo, 1, i - 1); } }
    int part(int lo, int n) {
        int hi = n - 1, mid = hi >> 1; T* a = k + lo;
        auto s2 = [&](int i, int j) { if (cmp(a[i], a[j]) > 0) std::swap(a[i], a[j]); };
        s2(0, mid); s2(0, hi); s2(mid, hi);
        T pivot = a[mid]; std::swap(a[mid], a[hi - 1]);
        int l = 0, r = hi - 1;
        while (l < r) { while (cmp(a[++l], pivot) < 0) {} while (cmp(pivot, a[--r]) < 0) {} if (l >= r) break; std::swap(a[l], a[r]); }
        if (l != hi - 1) std::swap(a[l], a[hi - 1]);
        return l;
    }
    void intro(int lo, int n, int depth) {
        while (n > 1) {
            if (n <= 16) { if (n == 2) { sig(lo, lo + 1); return; } if (n == 3) { sig(lo, lo + 1); sig(lo, lo + 2); sig(lo + 1, lo + 2); return; } ins(lo, n); return; }
            if (depth == 0) { heap(lo, n); return; }
            depth--;
            int p = part(lo, n);
            intro(lo + p + 1, n - (p + 1), depth);
            n = p;
        }
    }
};
template <class T, class Cmp> inline void bcl_sort(std::vector<T>& v, Cmp cmp) {
    int n = (int)v.size(); if (n < 2) return;
    int lg = 0; for (unsigned x = (unsigned)n; x >>= 1;) lg++;
    BclSort<T, Cmp> s{v.data(), cmp}; s.intro(0, n, 2 * (lg + 1));
}

// ---- LD1 term matching ----------------------------------------------------------------------------------------------------
// Semantics of the reference walk: D[i][0] = i, D[0][j] = 0 (Myers' SEARCH variant: the query may match a suffix of the
// path), a final node at depth j <= m+1 is reported when D[m][j] <= 1; nodes are visited in label order (pre-order).
// A subtree is skipped when min_i(D[i][j] + max(0, j-1-i)) > 1: completing the pattern from row i needs m-i more text
// characters but at most m+1-j remain below depth m+1, and a fresh start (row 0) deeper than column 2 costs >= 2.
// When every live row has no error budget left, only children whose label continues an exact match are visited.
inline int match_ld1_forward(const HostIndex& ix, uview q, std::vector<int>& out, int cap = 1024) {
    out.clear();
    const int m = (int)q.size();
    if (m == 0 || m > 64 || ix.trie.empty()) return 0;
    int count = 0;
    // column j of the DP lives at col[j*(m+1) ..]; depth <= m+1
    std::vector<int> colv((size_t)(m + 2) * (m + 1));
    int* col = colv.data();
    for (int i = 0; i <= m; i++) col[i] = i;
    struct Fr { uint32_t node; int depth; };
    Fr st[4096]; int sp = 0;                      // <= (m+1) levels x fan-out; fan-out is bounded by the alphabet in practice
    std::vector<Fr> big;                          // overflow stack for pathological fan-outs
    auto push = [&](uint32_t node, int depth) { if (sp < 4096) st[sp++] = {node, depth}; else big.push_back({node, depth}); };
    auto pop = [&](Fr& f) { if (!big.empty()) { f = big.back(); big.pop_back(); return true; } if (sp == 0) return false; f = st[--sp]; return true; };
    // children are linked in ascending label order; a stack must receive them in reverse to pop ascending
    auto push_children = [&](uint32_t node, int depth, const u16* want, int nwant) {
        const uint32_t eb = ix.edgeStart[node], ee = ix.edgeStart[node + 1];
        for (uint32_t e = ee; e-- > eb;) {     // reverse: the stack then pops ascending labels
            if (want) { const u16 lb = ix.edgeLabel[e]; bool ok = false; for (int k = 0; k < nwant; k++) if (lb == want[k]) { ok = true; break; } if (!ok) continue; }
            push(ix.edgeChild[e], depth);
        }
    };
    push_children(0, 1, nullptr, 0);              // root: row 0 has slack 1
    Fr f;
    while (pop(f)) {
        const auto& nd = ix.trie[f.node];
        const int j = f.depth;
        const int* p = col + (size_t)(j - 1) * (m + 1); int* c = col + (size_t)j * (m + 1);
        c[0] = 0;
        for (int i = 1; i <= m; i++) {
            int v = p[i - 1] + (q[i - 1] == nd.label ? 0 : 1);
            int u = p[i] + 1; if (u < v) v = u;
            u = c[i - 1] + 1; if (u < v) v = u;
            c[i] = v;
        }
        if (nd.term >= 0 && c[m] <= 1) { if (count < cap) out.push_back(nd.term); count++; }
        if (j >= m + 1) continue;
        // slack of row i: 1 - (D[i][j] + max(0, j-1-i)).  slack 1: any next label may still match; slack 0: only a diagonal
        // MATCH keeps it alive, i.e. the next label must be q[i]; negative: dead.
        bool any1 = false; u16 want[66]; int nwant = 0;
        for (int i = 0; i <= m; i++) {
            int s = 1 - (c[i] + (j - 1 - i > 0 ? j - 1 - i : 0));
            if (s >= 1) { any1 = true; break; }
            if (s == 0 && i < m) { bool dup = false; for (int k = 0; k < nwant; k++) if (want[k] == q[i]) { dup = true; break; } if (!dup) want[nwant++] = q[i]; }
        }
        if (any1) push_children(f.node, j + 1, nullptr, 0);
        else if (nwant) push_children(f.node, j + 1, want, nwant);
    }
    return count;
}

// The same set through the reversed-term trie.  match_ld1_forward accepts a term t (|t| <= m+1) iff some SUFFIX s of t has
// LD(q, s) <= 1 (the search variant's free start); |s| >= m-1, so at most two leading "junk" characters precede s.  Walking the
// reversed trie anchors the match at the term END: an ordinary LD<=1 automaton over reverse(q) visits O(m * fan-out) nodes
// (instead of every 1-2 character prefix of the vocabulary), and each accepted node is extended by the <= 2 junk characters.
// Results are returned in the forward trie's pre-order (= lexicographic order of the terms), first `cap` kept, like the forward walk.
// Words longer than 64 characters take the reference's Wagner-Fischer walk (FstIndex.MatchEditDistance1Slow, FstIndex.cs:362-440), which is a
// DIFFERENT predicate: whole-term Levenshtein distance <= 1 (row[0] grows with the depth: no free start), subtrees pruned once
// min(row) > 1, children visited in DESCENDING label order (ascending pushes on a LIFO stack), and the walk stops as soon as the
// output buffer is full (so the count it returns is capped).
inline int match_ld1_slow(const HostIndex& ix, uview q, std::vector<int>& out, int cap) {
    out.clear();
    const int m = (int)q.size();
    if (ix.trie.empty() || cap <= 0) return 0;
    struct Fr { uint32_t node; std::vector<int> row; };
    std::vector<Fr> st;
    { Fr f; f.node = 0; f.row.resize(m + 1); for (int i = 0; i <= m; i++) f.row[i] = i; st.push_back(std::move(f)); }
    int count = 0;
    while (!st.empty()) {
        Fr f = std::move(st.back()); st.pop_back();
        const auto& nd = ix.trie[f.node];
        if (f.row[m] <= 1 && nd.term >= 0) { out.push_back(nd.term); if (++count >= cap) return count; }
        int mn = f.row[0]; for (int i = 1; i <= m; i++) mn = std::min(mn, f.row[i]);
        if (mn > 1) continue;
        for (uint32_t e = ix.edgeStart[f.node]; e < ix.edgeStart[f.node + 1]; e++) {      // ascending pushes: popped in descending label order
            const u16 c = ix.edgeLabel[e];
            Fr g; g.node = ix.edgeChild[e]; g.row.resize(m + 1);
            g.row[0] = f.row[0] + 1;
            for (int i = 1; i <= m; i++) {
                int v = g.row[i - 1] + 1;
                if (f.row[i] + 1 < v) v = f.row[i] + 1;
                const int sub = f.row[i - 1] + (q[i - 1] == c ? 0 : 1);
                if (sub < v) v = sub;
                g.row[i] = v;
            }
            st.push_back(std::move(g));
        }
    }
    return count;
}

inline int match_ld1(const HostIndex& ix, uview q, std::vector<int>& out, int cap = 1024) {
    out.clear();
    const int m = (int)q.size();
    if (m > 64) return match_ld1_slow(ix, q, out, cap);
    if (m == 0 || ix.rEdgeStart.empty()) return 0;
    struct St { uint32_t node; int16_t i, d; uint8_t e; };
    St st[1024]; int sp = 0; std::vector<St> big;
    auto push = [&](uint32_t node, int i, int e, int d) { St x{node, (int16_t)i, (int16_t)d, (uint8_t)e}; if (sp < 1024) st[sp++] = x; else big.push_back(x); };
    auto pop = [&](St& x) { if (!big.empty()) { x = big.back(); big.pop_back(); return true; } if (sp == 0) return false; x = st[--sp]; return true; };
    std::vector<int> found;
    // accepted node at depth d: every terminal descendant within g <= m+1-d further levels (g <= 2) is a match
    auto accept = [&](uint32_t node, int d) {
        uint32_t fr[2][0]; (void)fr;
        struct E { uint32_t node; int g; };
        E es[256]; int ep = 0; std::vector<E> eb;
        auto epush = [&](uint32_t n2, int g) { if (ep < 256) es[ep++] = {n2, g}; else eb.push_back({n2, g}); };
        epush(node, 0);
        for (;;) {
            E x; if (!eb.empty()) { x = eb.back(); eb.pop_back(); } else if (ep > 0) x = es[--ep]; else break;
            if (ix.rTerm[x.node] >= 0) found.push_back(ix.rTerm[x.node]);
            if (d + x.g >= m + 1) continue;
            for (uint32_t k = ix.rEdgeStart[x.node]; k < ix.rEdgeStart[x.node + 1]; k++) epush(ix.rEdgeChild[k], x.g + 1);
        }
    };
    push(0, 0, 0, 0);
    St x;
    while (pop(x)) {
        const int i = x.i, e = x.e, d = x.d;
        // pattern fully consumed (possibly after deleting its last characters within the budget) -> accepted at this node
        if (i == m) accept(x.node, d);
        else if (e == 0 && i == m - 1) accept(x.node, d);            // delete the last pattern character
        if (i < m && e == 0) push(x.node, i + 1, 1, d);              // deletion of rq[i] (no text consumed); its acceptance is handled when popped
        if (d >= m + 1) continue;                                     // terms longer than m+1 cannot match
        const u16 want = i < m ? q[m - 1 - i] : 0;
        for (uint32_t k = ix.rEdgeStart[x.node]; k < ix.rEdgeStart[x.node + 1]; k++) {
            const u16 lb = ix.rEdgeLabel[k]; const uint32_t ch = ix.rEdgeChild[k];
            if (i < m && lb == want) push(ch, i + 1, e, d + 1);                       // match
            if (e == 0) {
                if (i < m && lb != want) push(ch, i + 1, 1, d + 1);                   // substitution
                push(ch, i, 1, d + 1);                                                  // insertion (extra text character)
            }
        }
    }
    std::sort(found.begin(), found.end()); found.erase(std::unique(found.begin(), found.end()), found.end());
    std::sort(found.begin(), found.end(), [&](int a, int c) { return ix.terms.keys.key((uint32_t)a) < ix.terms.keys.key((uint32_t)c); });
    const int count = (int)found.size();
    for (int k = 0; k < count && k < cap; k++) out.push_back(found[k]);
    return count;
}

// ---- Stage-1 plan -----------------------------------------------------------------------------------------------------------
// A fuzzy virtual term (ExpandMissingTerm): the LD1-matched member terms; its df = |union of their doc sets| is counted on the
// device (infx_union_counts) and cached with the members per misspelt word, like the reference's LruCache (VectorModel.cs:42).
struct FuzzyUnion {
    std::vector<int32_t> members;        // index term ids (df > 0), in trie pre-order
    std::atomic<int> df{-1};             // -1 = not counted yet
    bool materialised = false;           // host-built union (no device, or more than FUZZY_MAX_MEMBERS members)
    std::vector<int32_t> docs;           // only when materialised
};
constexpr size_t FUZZY_MAX_MEMBERS = 512;
struct FuzzyCache {
    std::atomic<long long> fuzzyNs{0}, fuzzyCalls{0}, fuzzyDocs{0}, ld1Ns{0};   // instrumentation (INFX_DEBUG)
    // least-recently-used, 1000 expansions — the reference's _fuzzyExpansionCache (VectorModel.cs:42, LruCache :745-800): a larger cache would let a long
    // query stream plan warmer than the reference can.  Entries are shared_ptr: a batch in flight keeps the unions it planned with after their eviction.
    static constexpr size_t CAPACITY = 1000;
    typedef std::list<std::pair<std::u16string, std::shared_ptr<FuzzyUnion>>> Order;     // front = most recently used
    std::mutex mu; Order order; std::unordered_map<std::u16string, Order::iterator> map;
    std::shared_ptr<FuzzyUnion> get(const ustr& k) {
        std::lock_guard<std::mutex> l(mu);
        auto it = map.find(k);
        if (it == map.end()) return nullptr;
        order.splice(order.begin(), order, it->second);
        return it->second->second;
    }
    std::shared_ptr<FuzzyUnion> put(const ustr& k, std::shared_ptr<FuzzyUnion> v) {   // first writer wins (two planner threads may expand the same word)
        std::lock_guard<std::mutex> l(mu);
        auto it = map.find(k);
        if (it != map.end()) { order.splice(order.begin(), order, it->second); return it->second->second; }
        if (map.size() >= CAPACITY) { map.erase(order.back().first); order.pop_back(); }
        order.emplace_front(k, v); map.emplace(k, order.begin());
        return v;
    }
    size_t size() { std::lock_guard<std::mutex> l(mu); return map.size(); }
};
// ExpandMissingTerm's filter over the LD1 matches (VectorModel.cs:660-683): members = matched terms that have postings
inline std::shared_ptr<FuzzyUnion> union_of_matches(const HostIndex& ix, const int* m, size_t n) {
    auto nf = std::make_shared<FuzzyUnion>();
    for (size_t k = 0; k < n; k++) { const int id = m[k]; if (ix.df[id] > 0 && ix.terms.len((uint32_t)id)) nf->members.push_back(id); }
    if (nf->members.empty()) nf->df.store(0);
    return nf;
}
inline void materialise_union(const HostIndex& ix, FuzzyUnion& fz) {    // union of the (sorted) member lists: pairwise merges, smallest first
    std::vector<std::pair<const int32_t*, size_t>> lists;
    for (int id : fz.members) lists.push_back({ix.terms.doc.data() + ix.terms.off[id], (size_t)ix.terms.len((uint32_t)id)});
    std::sort(lists.begin(), lists.end(), [](auto& a, auto& c) { return a.second < c.second; });
    std::vector<int32_t> acc, tmp;
    for (auto& l : lists) {
        if (acc.empty()) { acc.assign(l.first, l.first + l.second); continue; }
        tmp.resize(acc.size() + l.second);
        tmp.resize(std::set_union(acc.begin(), acc.end(), l.first, l.first + l.second, tmp.begin()) - tmp.begin());
        acc.swap(tmp);
    }
    fz.docs.swap(acc); fz.materialised = true; fz.df.store((int)fz.docs.size());
}

struct QueryPlan {
    bool blank = false, unsupported = false;
    ustr qtext;          // lower(normalize(trim(raw)))  == Query.Text inside SearchEngine.Search
    ustr searchText;     // normalised again by SearchPipeline.Execute
    ustr tfidfQuery;
    std::vector<infx_term> terms;                    // Bm25Scorer order
    std::vector<std::shared_ptr<FuzzyUnion>> fuzzy;  // per term (null for index terms)
    infx_query q{};
    bool noTerms = false;
    struct Raw { int id; ustr text; std::shared_ptr<FuzzyUnion> fz; bool pending = false; };      // pending: expansion deferred to the batch (device LD1 lookup)
    std::vector<Raw> rawTok;                          // between plan_tokens and plan_finish
    int depth = 0;
};

inline void analyze_query(uview text, int minIndexSize, bool& canUse, bool& mixed, ustr& longWords) {   // QueryAnalyzer.cs:10-54
    canUse = false; mixed = false; longWords.assign(text);
    int shortCnt = 0, longCnt = 0; ustr joined; bool any = false;
    for_each_word(text, [&](int off, int len) { any = true; if (len >= minIndexSize) { if (longCnt++) joined.push_back(u' '); joined.append(text.substr(off, len)); } else shortCnt++; });
    if (!any) { canUse = (int)text.size() >= minIndexSize; return; }
    if (longCnt > 0) { canUse = true; longWords = joined; }
    if (shortCnt > 0 && longCnt > 0) mixed = true;
}

// Pass 1a: text preparation and term lookup — a pure function of (index, raw query text): no cache, no device.  Under document sharding rank r runs it
// for its slice of a batch only and the ranks exchange the results (infx_session_prefetch_*: the plan exchange).
inline void plan_tokens_text(const HostIndex& ix, uview raw, int depth, QueryPlan& P) {
    P = QueryPlan(); P.depth = depth;
    size_t b = 0, e = raw.size();
    while (b < e && is_ws(raw[b])) b++;
    while (e > b && is_ws(raw[e - 1])) e--;
    normalize_into(raw.substr(b, e - b), P.qtext); lower_inplace(P.qtext);
    if (ix.cfg.syn.has()) ix.cfg.syn.canonicalize(P.qtext);        // SearchEngine.cs:276-286
    bool allws = true; for (u16 c : P.qtext) if (!is_ws(c)) { allws = false; break; }
    if (allws) { P.blank = true; return; }
    normalize_into(P.qtext, P.searchText);
    const int n = ix.cfg.ngram;
    bool canUse, mixed; ustr longWords;
    analyze_query(P.searchText, n, canUse, mixed, longWords);
    if (!canUse) { P.unsupported = true; return; }
    P.tfidfQuery = mixed ? longWords : P.searchText;
    { bool ws = true; for (u16 c : P.tfidfQuery) if (!is_ws(c)) { ws = false; break; } if (ws) P.tfidfQuery = P.searchText; }

    // raw tokens: words first, then n-grams of the padded text; at most 128 (VectorModel.cs:381,407)
    using Raw = QueryPlan::Raw;
    std::vector<Raw>& rawTok = P.rawTok;
    ustr text = normalize(P.tfidfQuery);
    auto visit = [&](uview s) { if (rawTok.size() >= 128) return; int64_t id = ix.terms.keys.find(s); if (id >= 0) rawTok.push_back({(int)id, ustr(), nullptr}); else rawTok.push_back({-1, ustr(s), nullptr}); };
    for_each_word(text, [&](int off, int len) { if (len >= n) visit(uview(text.data() + off, len)); });
    ustr padded((size_t)ix.cfg.startPad, (u16)0xFFFF); padded += text; padded.append((size_t)ix.cfg.stopPad, (u16)0xFFFE);
    if ((int)padded.size() >= n)
        for (int i = 0; i + n <= (int)padded.size(); i++) {
            bool allpad = true; for (int k = 0; k < n; k++) if (padded[i + k] != 0xFFFF && padded[i + k] != 0xFFFE) { allpad = false; break; }
            if (!allpad) visit(uview(padded.data() + i, n));
        }
    std::sort(rawTok.begin(), rawTok.end(), [](const Raw& a, const Raw& c) { return a.id != c.id ? a.id < c.id : a.text < c.text; });
    rawTok.erase(std::unique(rawTok.begin(), rawTok.end(), [](const Raw& a, const Raw& c) { return a.id == c.id && a.text == c.text; }), rawTok.end());
}
// Pass 1b: LD1 member lists of the unknown words (df of new unions still pending) — expansion cache first.
// hostUnions: build the unions on the host (engines without a device: planning introspection only).
// deferLd1: an unknown word the expansion cache does not hold is only marked `pending`; the caller expands the distinct pending words of the whole
// batch at once (ph_plan: on the device, infx_ld1_expand) and fills in `fz`.
inline void plan_tokens_expand(const HostIndex& ix, FuzzyCache& fc, QueryPlan& P, bool hostUnions, bool deferLd1 = false) {
    if (P.blank || P.unsupported) return;
    for (auto& r : P.rawTok) {
        if (r.id >= 0 || r.text.size() < 4) continue;      // ExpandMissingTerm (VectorModel.cs:643-743): unknown words of length >= 4
        auto fz = fc.get(r.text);
        if (!fz && deferLd1) { r.pending = true; continue; }
        if (!fz) {
            auto tF0 = std::chrono::steady_clock::now();
            std::vector<int> m; match_ld1(ix, r.text, m, 1024);
            fc.ld1Ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tF0).count();
            auto nf = std::make_shared<FuzzyUnion>();
            for (int id : m) if (ix.df[id] > 0 && ix.terms.len((uint32_t)id)) nf->members.push_back(id);
            if (nf->members.empty()) nf->df.store(0);
            else if (hostUnions) materialise_union(ix, *nf);
            fz = fc.put(r.text, nf);
            fc.fuzzyNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tF0).count();
            fc.fuzzyCalls++;
        }
        r.fz = fz;
    }
}
inline void plan_tokens(const HostIndex& ix, FuzzyCache& fc, uview raw, int depth, QueryPlan& P, bool hostUnions, bool deferLd1 = false) {
    plan_tokens_text(ix, raw, depth, P);
    plan_tokens_expand(ix, fc, P, hostUnions, deferLd1);
}

// Pass 2 (every union's df known): idf / maxScore, candidate-selection mode and tier roles.
inline void plan_finish(const HostIndex& ix, QueryPlan& P) {
    if (P.blank || P.unsupported) return;
    const int depth = P.depth;
    auto& rawTok = P.rawTok;
    const int N = ix.N;
    const float avgdl = ix.avgdl > 0.f ? ix.avgdl : 1.f;
    struct TI { int termId; int df; float idf, maxScore; std::shared_ptr<FuzzyUnion> fz; };
    std::vector<TI> tis;
    for (auto& r : rawTok) {
        int df = 0; std::shared_ptr<FuzzyUnion> fz;
        if (r.id >= 0) df = ix.df[r.id];
        else if (r.fz) { fz = r.fz; df = fz->df.load(); if (df <= 0) fz = nullptr; }
        if (df <= 0 || df > ix.cfg.stopTermLimit) continue;
        float idf = compute_idf(N, df);
        const float maxTf = 255.f, k1 = 1.2f, bb = 0.75f, delta = 1.0f;
        float minDlNorm = 1.f - bb + bb * (1.f / avgdl);
        float maxCore = (maxTf * (k1 + 1.f)) / (maxTf + k1 * minDlNorm);
        tis.push_back({r.id, df, idf, idf * (maxCore + delta), fz});
    }
    if (tis.empty()) { P.noTerms = true; return; }
    const int nT = (int)tis.size();
    P.terms.resize(nT); P.fuzzy.resize(nT);
    for (int i = 0; i < nT; i++) {
        infx_term& t = P.terms[i]; std::memset(&t, 0, sizeof t);
        t.term_id = tis[i].termId; t.idf = tis[i].idf; t.max_score = tis[i].maxScore; P.fuzzy[i] = tis[i].fz;
        if (tis[i].fz) { t.extra_len = (uint32_t)(tis[i].fz->materialised ? tis[i].fz->docs.size() : tis[i].fz->members.size()); t.reserved = tis[i].fz->materialised ? 0 : 1; }
    }
    infx_query& Q = P.q; std::memset(&Q, 0, sizeof Q);
    Q.num_terms = (uint32_t)nT; Q.depth = depth; Q.prefix_set = -1;
    const long k = depth;
    // prefix precedence (TieredCandidateSelector.cs:66-82, 455-532); originalQuery == the text handed to SearchWithMaxScore
    {
        ustr ql = P.tfidfQuery; lower_inplace(ql);
        int maxLen = std::min((int)ql.size(), 3);
        for (int len = maxLen; len >= 1; len--) {
            int64_t pk = ix.prefixKeys.find(uview(ql.data(), len));
            if (pk < 0) continue;
            long pop = ix.prefixPop[pk];
            if (pop == 0) continue;
            if (pop > k * 20) continue;
            if (pop <= k * 10) {
                Q.prefix_set = ix.prefixSetId[pk];
                if (pop >= std::min(k * 2, 100L)) { Q.mode = INFX_MODE_PREFIX; return; }
                break;   // pre-seen docs only
            }
        }
    }
    bool typo = false; float maxIdf = 0.f;
    for (auto& t : tis) { if (t.df < 10) typo = true; if (t.idf > maxIdf) maxIdf = t.idf; }
    struct Ord { int idx; float idf; };
    std::vector<Ord> ord(nT);
    for (int i = 0; i < nT; i++) ord[i] = {i, tis[i].idf};
    bcl_sort(ord, [](const Ord& a, const Ord& c) { return c.idf < a.idf ? -1 : (c.idf > a.idf ? 1 : 0); });
    if (typo || nT == 1) {
        Q.mode = INFX_MODE_DISJ;
        int nElig = 0;
        for (int r = 0; r < nT; r++) {
            infx_term& t = P.terms[ord[r].idx];
            bool lowq = ord[r].idf < (maxIdf * 0.2f);
            t.role = lowq ? INFX_ROLE_LOWQ : INFX_ROLE_ELIGIBLE; t.rank = (uint8_t)r;
            if (!lowq) nElig = r + 1;
        }
        Q.n_and = nElig; Q.df_s1 = nT;
        // Once a processed term alone holds >= 100*topK (+ the < 100 pre-seen) documents, localCount >= 100*topK after it and the
        // loop breaks (:317-318): no later rank can generate candidates. They keep scoring, but stop marking (smaller supersets).
        for (int r = 0; r < nT; r++) {
            if ((long)tis[ord[r].idx].df >= k * 100 + 100) {
                for (int r2 = r + 1; r2 < nT; r2++) P.terms[ord[r2].idx].role = 0;
                break;
            }
        }
        return;
    }
    Q.mode = INFX_MODE_AND; Q.n_and = nT;
    for (int r = 0; r < nT; r++) P.terms[ord[r].idx].role = (r < nT - 1) ? INFX_ROLE_AND : INFX_ROLE_LOWEST;
    {
        int sel = 0, cap = std::min(2, nT); float cutoff = maxIdf * 0.3f;
        for (int r = 0; r < nT && sel < cap; r++) {
            if (ord[r].idf <= 0.f || ord[r].idf < cutoff) continue;
            P.terms[ord[r].idx].role |= (sel == 0 ? INFX_ROLE_S1 : INFX_ROLE_S2);
            if (sel == 0) Q.df_s1 = tis[ord[r].idx].df; else Q.df_s2 = tis[ord[r].idx].df;
            sel++;
        }
    }
}

inline void plan_stage1(const HostIndex& ix, FuzzyCache& fc, uview raw, int depth, QueryPlan& P) {   // host-only (introspection)
    plan_tokens(ix, fc, raw, depth, P, true);
    for (auto& r : P.rawTok) if (r.fz && r.fz->df.load() < 0) materialise_union(ix, *r.fz);
    plan_finish(ix, P);
}

// ---- WordMatcher lookups ---------------------------------------------------------------------------------------------------
struct DocList { const int32_t* p; size_t n; };
struct WmResult { std::vector<DocList> lists; std::vector<std::vector<int32_t>> owned; bool any = false; };

inline void wm_collect(const HostIndex& ix, uview queryText, bool coverPrefixSuffix, WmResult& R) {
    R = WmResult();
    if (!ix.cfg.wordMatcher) return;
    auto add = [&](const Csr& c, uview key) { int64_t id = c.keys.find(key); if (id >= 0 && c.len((uint32_t)id) > 0) R.lists.push_back({c.doc.data() + c.off[id], (size_t)c.len((uint32_t)id)}); };
    std::vector<std::pair<int, int>> words;
    for_each_word(queryText, [&](int off, int len) { if (len >= 2) words.push_back({off, len}); });
    R.owned.reserve(words.size() * 2);
    for (auto& w : words) {
        uview word = queryText.substr(w.first, w.second);
        ustr nw(word); lower_inplace(nw); nw = normalize(nw);
        int L = (int)nw.size();
        add(ix.wmExact, nw);
        if (L >= ix.cfg.wmMinLD1 && L <= ix.cfg.wmMaxLD1) {
            add(ix.wmLd1, nw);
            ustr d;
            for (int i = 0; i < L; i++) { d.assign(nw); d.erase(i, 1); add(ix.wmLd1, d); add(ix.wmExact, d); }
        }
        if (coverPrefixSuffix && !nw.empty()) {    // LookupAffix: <= 4096 trie terms, prefix hits first, then suffix hits
            auto lo = std::lower_bound(ix.affixFwd.begin(), ix.affixFwd.end(), nw, [&](uint32_t a, const ustr& kx) { return ix.words.key(a) < uview(kx); });
            auto hi = lo; while (hi != ix.affixFwd.end() && ix.words.key(*hi).substr(0, nw.size()) == uview(nw)) ++hi;
            ustr rn(nw.rbegin(), nw.rend());
            auto revKeyLess = [&](uint32_t a, const ustr& kx) {   // compare reversed(word a) < kx
                uview x = ix.words.key(a); size_t nx = x.size(), nk = kx.size(), nmin = std::min(nx, nk);
                for (size_t i = 0; i < nmin; i++) { u16 cx = x[nx - 1 - i]; if (cx != kx[i]) return cx < kx[i]; }
                return nx < nk;
            };
            auto rlo = std::lower_bound(ix.affixRev.begin(), ix.affixRev.end(), rn, revKeyLess);
            auto ends_with = [&](uint32_t a) { uview x = ix.words.key(a); return x.size() >= nw.size() && x.substr(x.size() - nw.size()) == uview(nw); };
            auto rhi = rlo; while (rhi != ix.affixRev.end() && ends_with(*rhi)) ++rhi;
            size_t pc = hi - lo, sc = rhi - rlo;
            if (pc || sc) {
                std::vector<int32_t> docs; size_t budget = 4096;
                size_t take = std::min(pc, budget);
                for (size_t i = 0; i < take; i++) docs.push_back(ix.wordLastDoc[*(lo + i)]);
                budget -= take; take = std::min(sc, budget);
                for (size_t i = 0; i < take; i++) docs.push_back(ix.wordLastDoc[*(rlo + i)]);
                std::sort(docs.begin(), docs.end()); docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
                R.owned.push_back(std::move(docs));
            }
        }
    }
    for (auto& o : R.owned) if (!o.empty()) R.lists.push_back({o.data(), o.size()});
    R.any = !R.lists.empty();
}
inline bool wm_contains(const WmResult& R, int32_t doc) {
    for (auto& l : R.lists) if (std::binary_search(l.p, l.p + l.n, doc)) return true;
    return false;
}
// hit[i] |= (q[i] in list) for m probes: branch-free binary searches advanced level by level with the next level's cache line
// prefetched, so the probes' DRAM misses overlap instead of serialising (a common word's list has ~10^5..10^6 ids)
inline void wm_contains_batch(const WmResult& R, const int32_t* q, int m, std::vector<uint8_t>& hit, std::vector<uint32_t>& base) {
    hit.assign(m, 0);
    for (auto& l : R.lists) {
        if (l.n == 0) continue;
        if (l.n < 32 || m < 8) { for (int i = 0; i < m; i++) if (!hit[i] && std::binary_search(l.p, l.p + l.n, q[i])) hit[i] = 1; continue; }
        base.assign(m, 0);
        size_t len = l.n;
        while (len > 1) {
            const size_t half = len / 2, nlen = len - half, nhalf = nlen / 2;
            for (int i = 0; i < m; i++) {
                uint32_t b = base[i];
                b += (l.p[b + half - 1] < q[i]) ? (uint32_t)half : 0u;
                base[i] = b;
                if (nhalf) __builtin_prefetch(l.p + b + nhalf - 1);
            }
            len = nlen;
        }
        for (int i = 0; i < m; i++) { size_t b = base[i] + (l.p[base[i]] < q[i] ? 1 : 0); if (b < l.n && l.p[b] == q[i]) hit[i] = 1; }
    }
}
// first `limit` ids of the union in ascending order that are not in `exclude` (sorted)
// With deletions (del != nullptr): live[0..1] receive the first two ids of that sequence that are not deleted (-1 = none); the walk goes on past
// `limit` until `wantLive` of them are known (SearchPipeline.cs:532-537: only live WordMatcher ids get a docIndex).
inline void wm_first_unique(const WmResult& R, const std::vector<int32_t>& excludeSorted, size_t limit, std::vector<int32_t>& out,
                            const uint8_t* del = nullptr, int wantLive = 0, int32_t* live = nullptr) {
    out.clear();
    int nLive = 0; if (live) { live[0] = -1; live[1] = -1; }
    if (!del) wantLive = 0;
    if ((limit == 0 && wantLive == 0) || R.lists.empty()) return;
    using E = std::pair<int32_t, size_t>;   // (doc, list)
    std::priority_queue<E, std::vector<E>, std::greater<E>> pq;
    std::vector<size_t> pos(R.lists.size(), 0);
    for (size_t i = 0; i < R.lists.size(); i++) if (R.lists[i].n) pq.push({R.lists[i].p[0], i});
    int32_t last = -1;
    while (!pq.empty() && (out.size() < limit || nLive < wantLive)) {
        E e2 = pq.top(); pq.pop();
        size_t li = e2.second;
        if (++pos[li] < R.lists[li].n) pq.push({R.lists[li].p[pos[li]], li});
        if (e2.first == last) continue;
        last = e2.first;
        if (std::binary_search(excludeSorted.begin(), excludeSorted.end(), e2.first)) continue;
        if (out.size() < limit) out.push_back(e2.first);
        if (del && live && nLive < 2 && !del[e2.first]) live[nLive++] = e2.first;
    }
}

// ---- CoverageEngine.PrepareQuery -------------------------------------------------------------------------------------------
// QT = infx_cov_query (the fast envelope: INFX_MAX_QUERY_CHARS / INFX_MAX_QUERY_TOKENS) or infx_cov_query_long (INFX_LONGQ_CHARS / INFX_LONGQ_TOKENS); same members.
template <class QT, int MAXCHARS, int MAXTOK>
inline int32_t prepare_cov_query_t(const HostIndex& ix, uview query, QT& C) {
    std::memset(&C, 0, sizeof C);
    if ((int)query.size() > MAXCHARS) return INFX_EUNSUPPORTED;
    std::memcpy(C.text, query.data(), query.size() * 2); C.text_len = (int)query.size();
    struct Tk { int off, len; };
    std::vector<Tk> raw, uq, fus;
    for_each_word(query, [&](int off, int len) { fus.push_back({off, len}); if (len >= 2) raw.push_back({off, len}); });
    for (auto& t : raw) {
        bool dup = false;
        for (auto& u : uq) if (ic_equal(query.substr(u.off, u.len), query.substr(t.off, t.len))) { dup = true; break; }      // CoverageTokenizer.cs:50-57: OrdinalIgnoreCase
        if (!dup) uq.push_back(t);
    }
    if ((int)uq.size() > MAXTOK || (int)fus.size() > 2 * MAXTOK) return INFX_EUNSUPPORTED;
    C.num_tokens = (int)uq.size();
    const int n = ix.cfg.ngram;
    for (int i = 0; i < (int)uq.size(); i++) {
        C.tok_off[i] = (uint16_t)uq[i].off; C.tok_len[i] = (uint16_t)uq[i].len;
        uview term = query.substr(uq[i].off, uq[i].len);
        float sum = 0.f; int cnt = 0;
        if (ix.N > 0 && (int)term.size() >= n)
            for (int j = 0; j + n <= (int)term.size(); j++) { int64_t id = ix.terms.keys.find(term.substr(j, n)); if (id >= 0 && ix.df[id] > 0) { sum += compute_idf(ix.N, ix.df[id]); cnt++; } }
        C.term_idf[i] = cnt > 0 ? sum / (float)cnt : log2f((float)(term.size() + 1));
        // WordIdfCache is OrdinalIgnoreCase and both sides are lower-case here: an exact lookup — unless the word's class has alias members (icWords, keyed by the
        // class representative: the corpus may hold 'ϑερμος' where the query says 'θερμος', or both)
        bool classHit = false;
        const auto& TT = tables(); ustr F(term); bool qAlias = false;
        for (auto& c : F) { const u16 r = TT.icrep[c]; if (r != c) { qAlias = true; c = r; } }      // the class representative of the query word
        if (ix.icWords.size()) {
            const int64_t ci = ix.icWords.find(F);
            if (ci >= 0) { C.word_idf[i] = ix.icWordIdf[ci]; classHit = true; }
        }
        if (!classHit) {      // a class without alias members in the corpus: its only possible word is the representative itself
            int64_t w = ix.words.find(qAlias ? uview(F) : term);
            C.word_idf[i] = (w >= 0 && ix.wordDf[w] > 0 && (int)ix.wordDf[w] <= ix.N) ? ix.wordIdf[w] : 0.f;
        }
    }
    C.has_word_idf = uq.empty() ? 0 : 1;
    C.num_fusion_tokens = (int)fus.size();
    for (int i = 0; i < (int)fus.size(); i++) { C.ftok_off[i] = (uint16_t)fus[i].off; C.ftok_len[i] = (uint16_t)std::min(fus[i].len, 65535); }
    C.lcs_tolerance = (int)query.size() >= 5 ? (int)((double)query.size() * 0.2) : 0;
    return INFX_OK;
}
inline int32_t prepare_cov_query(const HostIndex& ix, uview query, infx_cov_query& C) { return prepare_cov_query_t<infx_cov_query, INFX_MAX_QUERY_CHARS, INFX_MAX_QUERY_TOKENS>(ix, query, C); }
// a query beyond the fast envelope (prepare_cov_query returned INFX_EUNSUPPORTED): the long record (include/infidex_hip.h); INFX_EUNSUPPORTED again = beyond that too
inline int32_t prepare_cov_query_long(const HostIndex& ix, uview query, infx_cov_query_long& C) { return prepare_cov_query_t<infx_cov_query_long, INFX_LONGQ_CHARS, INFX_LONGQ_TOKENS>(ix, query, C); }

} // namespace infx
