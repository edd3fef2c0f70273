// ORACLE — test infrastructure only (see oracle/README.md).
// SynonymMap: union-find canonicalisation of tokens.  Restates src/Infidex/Synonyms/SynonymMap.cs:
//   AddSynonym :33-62, GetCanonical :124-137, CanonicalizeText :146-181, Find :199-209, Union :211-246.
// Used at three places of the hot path: index text (Indexing/VectorModel.cs:90-93), query text (SearchEngine.cs:276-286) and the
// coverage document text (Scoring/SearchPipeline.cs:482-489).  Only needed to replay SchoolSearchParityTests.cs (3 synonym pairs).
#pragma once
#include "text.hpp"
#include <unordered_map>

namespace orc {

struct SynonymMap {
    std::unordered_map<ustr, ustr> parent;       // keys are lower-invariant strings
    bool has() const { return !parent.empty(); }  // HasCanonicalMappings

    static ustr trim_lower(uview s) {
        size_t b = 0, e = s.size();
        while (b < e && is_whitespace(s[b])) b++;
        while (e > b && is_whitespace(s[e - 1])) e--;
        return to_lower_inv(ustr(s.substr(b, e - b)));
    }
    void ensure(const ustr& t) { if (!parent.count(t)) parent[t] = t; }
    ustr find(const ustr& t) {
        ensure(t);
        ustr p = parent[t];
        if (p != t) { ustr r = find(p); parent[t] = r; }
        return parent[t];
    }
    void add(uview a, uview b) {                  // AddSynonym: blank terms and identical terms are ignored
        auto blank = [](uview s) { for (u16 c : s) if (!is_whitespace(c)) return false; return true; };
        if (blank(a) || blank(b)) return;
        ustr t1 = trim_lower(a), t2 = trim_lower(b);
        if (t1 == t2) return;
        ensure(t1); ensure(t2);
        ustr r1 = find(t1), r2 = find(t2);
        if (r1 == r2) return;
        // the longer surface form is the canonical root; equal lengths: ordinal order
        ustr canon, other;
        if (r1.size() != r2.size()) { if (r1.size() >= r2.size()) { canon = r1; other = r2; } else { canon = r2; other = r1; } }
        else { if (r1.compare(r2) <= 0) { canon = r1; other = r2; } else { canon = r2; other = r1; } }
        parent[other] = canon;
    }
    ustr canonical(uview token) {                 // GetCanonical: trims and lower-cases EVERY token, mapped or not
        bool blank = true; for (u16 c : token) if (!is_whitespace(c)) { blank = false; break; }
        if (blank) return ustr();
        ustr t = trim_lower(token);
        if (!parent.count(t)) return t;
        return find(t);
    }
    ustr canonicalize(uview text) {               // CanonicalizeText with TokenizerSetup.Delimiters
        if (text.empty() || parent.empty()) return ustr(text);
        const Delims& dl = default_delims();
        ustr out; out.reserve(text.size());
        size_t i = 0;
        while (i < text.size()) {
            if (dl.is(text[i])) { out.push_back(text[i]); i++; continue; }
            size_t st = i;
            while (i < text.size() && !dl.is(text[i])) i++;
            out += canonical(text.substr(st, i - st));
        }
        return out;
    }
};

} // namespace orc
