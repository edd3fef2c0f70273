// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's WordMatcher (extra Stage-2 candidate source).
//
// Follows (paths relative to /root/reference/src/Infidex):
//   WordMatcher/WordMatcher.cs:82-118   Load (lower-case THEN normalise; exact 2..8, LD1 deletions 3..8, affix >=3)
//   WordMatcher/WordMatcher.cs:166-196  IndexWordInFst — _fstIndex is null while loading, so every word
//                                       OCCURRENCE gets a fresh term id and the trie node keeps the last one:
//                                       an affix hit yields only the LAST document containing that word (quirk Q13)
//   WordMatcher/WordMatcher.cs:201-246  Lookup (exact + symmetric-delete LD1)
//   WordMatcher/WordMatcher.cs:277-354  LookupAffix (<= 4096 trie terms: prefix hits first, then suffix hits)
//   Scoring/WordMatcherLookup.cs:11-68  Execute (per query word of length >= 2)
// RoaringBitmap is restated as a sorted unique int32 vector.
#pragma once
#include "index.hpp"

namespace orc {

inline void union_into(std::vector<int32_t>& acc, const std::vector<int32_t>& add) {
    if (add.empty()) return;
    if (acc.empty()) { acc = add; return; }
    std::vector<int32_t> r; r.reserve(acc.size() + add.size());
    std::set_union(acc.begin(), acc.end(), add.begin(), add.end(), std::back_inserter(r));
    acc.swap(r);
}

struct WordMatcher {
    const Config* cfg = nullptr;
    std::unordered_map<ustr, std::vector<int32_t>, UHash> exact, ld1;
    std::unordered_map<ustr, int32_t, UHash> affixLast;   // word -> last doc containing it
    std::vector<std::pair<ustr, int32_t>> affixFwd;        // sorted by word (trie pre-order)
    std::vector<std::pair<ustr, int32_t>> affixRev;        // sorted by reversed word (reverse-trie pre-order)
    static constexpr int MaxFstAffixTermsPerQuery = 4096;

    static void add_to(std::unordered_map<ustr, std::vector<int32_t>, UHash>& m, const ustr& k, int doc) {
        auto& v = m[k];
        if (v.empty() || v.back() != doc) v.push_back(doc);
    }
    void load(uview rawText, int doc) {
        ustr norm = default_normalizer().normalize(to_lower_inv(rawText));
        std::vector<Slice> words; split_words(norm, words);
        for (auto& w : words) {
            ustr word(norm.data() + w.off, w.len);
            int L = w.len;
            if (L >= cfg->wmMinExact && L <= cfg->wmMaxExact) add_to(exact, word, doc);
            if (L >= cfg->wmMinLD1 && L <= cfg->wmMaxLD1) {
                for (int i = 0; i < L; i++) { ustr v = word; v.erase(i, 1); add_to(ld1, v, doc); }
            }
            if (L >= cfg->wmMinLD1) affixLast[word] = doc;
        }
    }
    void finalize_index() {
        affixFwd.assign(affixLast.begin(), affixLast.end());
        std::sort(affixFwd.begin(), affixFwd.end(), [](auto& a, auto& b) { return a.first < b.first; });
        affixRev.clear(); affixRev.reserve(affixFwd.size());
        for (auto& p : affixFwd) { ustr r(p.first.rbegin(), p.first.rend()); affixRev.push_back({std::move(r), p.second}); }
        std::sort(affixRev.begin(), affixRev.end(), [](auto& a, auto& b) { return a.first < b.first; });
        std::unordered_map<ustr, int32_t, UHash>().swap(affixLast);
    }
    void accumulate(const std::unordered_map<ustr, std::vector<int32_t>, UHash>& m, const ustr& k, std::vector<int32_t>& res, bool& any) const {
        auto it = m.find(k);
        if (it == m.end()) return;
        any = true;
        union_into(res, it->second);
    }
    // returns false when the reference would return null (no key matched at all)
    bool lookup(uview query, std::vector<int32_t>& res) const {
        ustr n = default_normalizer().normalize(to_lower_inv(query));
        int L = (int)n.size();
        bool any = false;
        accumulate(exact, n, res, any);
        if (L >= cfg->wmMinLD1 && L <= cfg->wmMaxLD1) {
            accumulate(ld1, n, res, any);
            for (int i = 0; i < L; i++) {
                ustr d = n; d.erase(i, 1);
                accumulate(ld1, d, res, any);
                accumulate(exact, d, res, any);
            }
        }
        return any;
    }
    static std::pair<size_t, size_t> prefix_range(const std::vector<std::pair<ustr, int32_t>>& v, const ustr& p) {
        auto lo = std::lower_bound(v.begin(), v.end(), p, [](auto& a, const ustr& k) { return a.first < k; });
        auto hi = lo;
        while (hi != v.end() && starts_with(hi->first, p)) ++hi;   // contiguous in sorted order
        return {size_t(lo - v.begin()), size_t(hi - v.begin())};
    }
    bool lookup_affix(uview query, std::vector<int32_t>& res) const {
        ustr n = default_normalizer().normalize(to_lower_inv(query));
        if (n.empty()) return false;
        auto pr = prefix_range(affixFwd, n);
        ustr rn(n.rbegin(), n.rend());
        auto sr = prefix_range(affixRev, rn);
        size_t pc = pr.second - pr.first, sc = sr.second - sr.first;
        if (pc == 0 && sc == 0) return false;
        size_t budget = MaxFstAffixTermsPerQuery;
        std::vector<int32_t> docs;
        size_t take = std::min(pc, budget);
        for (size_t i = 0; i < take; i++) docs.push_back(affixFwd[pr.first + i].second);
        budget -= take;
        take = std::min(sc, budget);
        for (size_t i = 0; i < take; i++) docs.push_back(affixRev[sr.first + i].second);
        std::sort(docs.begin(), docs.end());
        docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
        union_into(res, docs);
        return true;
    }
    // WordMatcherLookup.Execute
    void execute(uview queryText, bool coverPrefixSuffix, std::vector<int32_t>& result) const {
        result.clear();
        std::vector<Slice> words; split_words(queryText, words);
        for (auto& w : words) {
            if (w.len < 2) continue;
            uview word(queryText.data() + w.off, w.len);
            bool ws = true; for (u16 c : word) if (!is_whitespace(c)) { ws = false; break; }
            if (ws) continue;
            std::vector<int32_t> ids;
            if (lookup(word, ids)) union_into(result, ids);
            if (coverPrefixSuffix) { std::vector<int32_t> a; if (lookup_affix(word, a)) union_into(result, a); }
        }
    }
};

} // namespace orc
