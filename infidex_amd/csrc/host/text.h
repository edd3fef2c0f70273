// Host-side text layer of the product (C++ stands in for the reference's C# host; no .NET toolchain in the image).
// Mirrors, for the hot path only:
//   Tokenization/TextNormalizer.cs:120-200, 203-304   Normalize + default diacritic map
//   Tokenization/TokenizerSetup.cs:36-43              delimiters
//   string.ToLowerInvariant / char.IsLetter / OrdinalIgnoreCase: the BMP simple case mappings and the letter set, generated FROM DATA
//   (../unicode_tables.h, tools/gen_unicode_tables.py: Unicode 13 simple mappings with .NET's invariant exceptions for U+0130 / U+0131)
// Table-driven: one 64 Ki-entry fold table per operation, built once.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <string_view>
#include "../unicode_tables.h"

namespace infx {

using u16 = char16_t;
using ustr = std::u16string;
using uview = std::u16string_view;

struct TextTables {
    std::vector<u16> lower;      // ToLowerInvariant
    std::vector<u16> norm;       // TextNormalizer char map (+ \t \n \r -> ' ')
    std::vector<uint8_t> delim;  // tokenizer delimiters
    std::vector<u16> icrep;      // OrdinalIgnoreCase on LOWER-cased text: lower-case characters that share their upper-case image with another one
                                 // (final sigma / sigma, long s / s, micro sign / mu ...) -> the class representative lower(upper(c))
    TextTables() : lower(65536), norm(65536), delim(65536, 0), icrep(65536) {
        for (int i = 0; i < 65536; i++) { lower[i] = (u16)i; norm[i] = (u16)i; }
        static const uint16_t lowerPairs[][2] = { INFX_UC_LOWER_PAIRS };
        for (auto& pr : lowerPairs) lower[pr[0]] = pr[1];
        static const uint16_t repPairs[][2] = { INFX_UC_ICREP_PAIRS };
        for (int i = 0; i < 65536; i++) icrep[i] = (u16)i;
        for (auto& pr : repPairs) icrep[pr[0]] = pr[1];
        static const char16_t* from = u"ÆæØøÅåÄäÖöÜüßŠšČčŘřŽžŇňŤťĎďĚěÁáÉéÍíÓóÚúÝýŮůĄąĆćĘęŁłŃńŚśŹźŻżŐőŰűĂăÂâÎîȘșȚțĞğİıŞşÀàÇçÈèÊêËëÌìÏïÑñÒòÔôÕõÙùÛûŸÿÐðÞþ";
        static const char16_t* to   = u"EeOoAaAaOoUusSsCcRrZzNnTtDdEeAaEeIiOoUuYyUuAaCcEeLlNnSsZzZzOoUuAaAaIiSsTtGgIiSsAaCcEeEeEeIiIiNnOoOoOoUuUuYyDdTt";
        for (int i = 0; from[i]; i++) norm[from[i]] = to[i];
        norm[u'\t'] = norm[u'\n'] = norm[u'\r'] = u' ';
        const u16 d[] = {u' ',u'-',u'/',u'.',u',',u':',u';',u'\'',u'`',0x2013,0x2014,u'*',u'&',u'\\',u'_',u'(',u')',u'{',u'}',u'[',u']',u'\t'};
        for (u16 c : d) delim[c] = 1;
    }
};
inline const TextTables& tables() { static TextTables t; return t; }

// TextNormalizer.NormalizeWithStandardWhitespace: map chars, collapse runs of spaces.
inline void normalize_into(uview in, ustr& out) {
    const auto& T = tables();
    out.clear(); out.reserve(in.size());
    bool prev = false;
    for (u16 c : in) {
        u16 m = T.norm[c];
        bool sp = m == u' ';
        if (sp && prev) continue;
        out.push_back(m); prev = sp;
    }
}
inline ustr normalize(uview in) { ustr o; normalize_into(in, o); return o; }
inline void lower_inplace(ustr& s) { const auto& T = tables(); for (auto& c : s) c = T.lower[c]; }
// string.Equals(a, b, OrdinalIgnoreCase) for lower-cased a and b (ToUpperInvariant images compared, char by char)
inline bool ic_equal(uview a, uview b) {
    if (a.size() != b.size()) return false;
    const auto& T = tables();
    for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i] && T.icrep[a[i]] != T.icrep[b[i]]) return false;
    return true;
}
inline bool is_ws(u16 c) {
    return c == 0x20 || (c >= 0x09 && c <= 0x0D) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline bool is_delim(u16 c) { return tables().delim[c] != 0; }

struct Span { int off, len; };
// maximal runs of non-delimiters (Tokenizer.cs:108-138, string.Split(delims, RemoveEmptyEntries))
template <class F> inline void for_each_word(uview s, F&& f) {
    const auto& T = tables();
    int n = (int)s.size(), i = 0;
    while (i < n) {
        while (i < n && T.delim[s[i]]) i++;
        if (i >= n) break;
        int st = i;
        while (i < n && !T.delim[s[i]]) i++;
        f(st, i - st);
    }
}

inline uint64_t hash_u16(const u16* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; h ^= h >> 29; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}


// ---- synonyms ---------------------------------------------------------------------------------------------------------------
// SynonymMap (Synonyms/SynonymMap.cs): tokens of one equivalence class are replaced by the class's canonical form in the index
// text (VectorModel.cs:90-93), the query text (SearchEngine.cs:276-286) and the coverage document text (SearchPipeline.cs:482-489).
// Union rule (:211-246): the longer root wins, equal lengths: ordinal order.  GetCanonical (:124-137) trims and lower-cases every
// token, mapped or not.  Pairs are added before indexing; lookups afterwards are read-only (roots resolved eagerly).
struct SynMap {
    std::vector<std::pair<ustr, ustr>> parent;      // small (a handful of pairs): linear search
    bool has() const { return !parent.empty(); }
    static ustr trim_lower(uview t) {
        size_t b = 0, e = t.size();
        while (b < e && is_ws(t[b])) b++;
        while (e > b && is_ws(t[e - 1])) e--;
        ustr r(t.substr(b, e - b)); lower_inplace(r); return r;
    }
    int idx(const ustr& t) const { for (size_t i = 0; i < parent.size(); i++) if (parent[i].first == t) return (int)i; return -1; }
    ustr root(ustr t) const { for (;;) { int i = idx(t); if (i < 0 || parent[i].second == t) return t; t = parent[i].second; } }
    void add(uview a, uview b) {
        auto blank = [](uview x) { for (u16 c : x) if (!is_ws(c)) return false; return true; };
        if (blank(a) || blank(b)) return;
        ustr t1 = trim_lower(a), t2 = trim_lower(b);
        if (t1 == t2) return;
        if (idx(t1) < 0) parent.push_back({t1, t1});
        if (idx(t2) < 0) parent.push_back({t2, t2});
        ustr r1 = root(t1), r2 = root(t2);
        if (r1 == r2) return;
        const bool firstWins = r1.size() != r2.size() ? r1.size() >= r2.size() : r1.compare(r2) <= 0;
        const ustr& canon = firstWins ? r1 : r2; const ustr& other = firstWins ? r2 : r1;
        parent[idx(other)].second = canon;
    }
    void canonicalize(ustr& text) const {            // CanonicalizeText with the tokenizer's delimiters
        if (text.empty() || parent.empty()) return;
        ustr out; out.reserve(text.size());
        size_t i = 0;
        while (i < text.size()) {
            if (is_delim(text[i])) { out.push_back(text[i]); i++; continue; }
            size_t st = i;
            while (i < text.size() && !is_delim(text[i])) i++;
            uview tok(text.data() + st, i - st);
            bool blank = true; for (u16 c : tok) if (!is_ws(c)) { blank = false; break; }
            if (blank) continue;
            out += root(trim_lower(tok));
        }
        text.swap(out);
    }
};
} // namespace infx
