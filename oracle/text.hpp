// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's text layer. Not part of the product path.
//
// Follows (reference paths relative to /root/reference/src/Infidex):
//   Tokenization/TextNormalizer.cs:120-200,203-304  (Normalize, default char map)
//   Tokenization/TokenizerSetup.cs:36-43            (default delimiters)
//   .NET BCL char.ToLowerInvariant / ToUpperInvariant / IsWhiteSpace / IsLetter
//     (simple 1:1 case mappings of the whole BMP and the letter categories, from unicode_tables.hpp — generated from Unicode 13 data with .NET's
//      two invariant exceptions; .NET 8 ships a newer Unicode: characters added after 13.0 are not covered)
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <array>
#include "unicode_tables.hpp"

namespace orc {

using u16 = char16_t;
using ustr = std::u16string;
using uview = std::u16string_view;

// ---- .NET invariant simple case mapping and letter set: generated from data (oracle/unicode_tables.hpp, tools/gen_unicode_tables.py) -------------
struct CaseTables {
    std::vector<u16> lower, upper; std::vector<uint32_t> letter;
    CaseTables() : lower(65536), upper(65536) {
        for (int i = 0; i < 65536; i++) { lower[i] = (u16)i; upper[i] = (u16)i; }
        static const uint16_t lo[][2] = { INFX_UC_LOWER_PAIRS };
        static const uint16_t up[][2] = { INFX_UC_UPPER_PAIRS };
        static const uint32_t lt[] = { INFX_UC_LETTER_WORDS };
        for (auto& pr : lo) lower[pr[0]] = pr[1];
        for (auto& pr : up) upper[pr[0]] = pr[1];
        letter.assign(lt, lt + 2048);
    }
};
inline const CaseTables& case_tables() { static CaseTables t; return t; }
inline u16 to_lower_inv(u16 c) { return case_tables().lower[c]; }      // char.ToLowerInvariant (U+0130 maps to itself)
inline u16 to_upper_inv(u16 c) { return case_tables().upper[c]; }      // char.ToUpperInvariant (U+0131 maps to itself); OrdinalIgnoreCase compares these images
inline ustr to_lower_inv(uview s) { ustr r(s); for (auto& c : r) c = to_lower_inv(c); return r; }

inline bool is_whitespace(u16 c) {   // char.IsWhiteSpace
    return c == 0x20 || (c >= 0x09 && c <= 0x0D) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline bool is_letter(u16 c) { return (case_tables().letter[c >> 5] >> (c & 31)) & 1u; }      // char.IsLetter: Lu | Ll | Lt | Lm | Lo

// OrdinalIgnoreCase primitives (compare after ToUpperInvariant per code unit)
inline bool eq_ic(uview a, uview b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) if (to_upper_inv(a[i]) != to_upper_inv(b[i])) return false;
    return true;
}
inline bool starts_with_ic(uview s, uview p) { return s.size() >= p.size() && eq_ic(s.substr(0, p.size()), p); }
inline bool ends_with_ic(uview s, uview p) { return s.size() >= p.size() && eq_ic(s.substr(s.size() - p.size()), p); }
inline int index_of_ic(uview s, uview p) {
    if (p.empty()) return 0;
    if (s.size() < p.size()) return -1;
    for (size_t i = 0; i + p.size() <= s.size(); i++) if (eq_ic(s.substr(i, p.size()), p)) return (int)i;
    return -1;
}
inline bool contains_ic(uview s, uview p) { return index_of_ic(s, p) >= 0; }
inline bool starts_with(uview s, uview p) { return s.size() >= p.size() && s.substr(0, p.size()) == p; }
inline bool ends_with(uview s, uview p) { return s.size() >= p.size() && s.substr(s.size() - p.size()) == p; }

// ---- TextNormalizer.CreateDefault (TextNormalizer.cs:203-304) -----------------------------
struct Normalizer {
    std::vector<u16> map;   // 65536 entries
    Normalizer() : map(65536) {
        for (int i = 0; i < 65536; i++) map[i] = (u16)i;
        static const u16 pairs[][2] = {
            {u'Æ',u'E'},{u'æ',u'e'},{u'Ø',u'O'},{u'ø',u'o'},{u'Å',u'A'},{u'å',u'a'},{u'Ä',u'A'},{u'ä',u'a'},
            {u'Ö',u'O'},{u'ö',u'o'},{u'Ü',u'U'},{u'ü',u'u'},{u'ß',u's'},
            {u'Š',u'S'},{u'š',u's'},{u'Č',u'C'},{u'č',u'c'},{u'Ř',u'R'},{u'ř',u'r'},{u'Ž',u'Z'},{u'ž',u'z'},
            {u'Ň',u'N'},{u'ň',u'n'},{u'Ť',u'T'},{u'ť',u't'},{u'Ď',u'D'},{u'ď',u'd'},{u'Ě',u'E'},{u'ě',u'e'},
            {u'Á',u'A'},{u'á',u'a'},{u'É',u'E'},{u'é',u'e'},{u'Í',u'I'},{u'í',u'i'},{u'Ó',u'O'},{u'ó',u'o'},
            {u'Ú',u'U'},{u'ú',u'u'},{u'Ý',u'Y'},{u'ý',u'y'},{u'Ů',u'U'},{u'ů',u'u'},
            {u'Ą',u'A'},{u'ą',u'a'},{u'Ć',u'C'},{u'ć',u'c'},{u'Ę',u'E'},{u'ę',u'e'},{u'Ł',u'L'},{u'ł',u'l'},
            {u'Ń',u'N'},{u'ń',u'n'},{u'Ś',u'S'},{u'ś',u's'},{u'Ź',u'Z'},{u'ź',u'z'},{u'Ż',u'Z'},{u'ż',u'z'},
            {u'Ő',u'O'},{u'ő',u'o'},{u'Ű',u'U'},{u'ű',u'u'},
            {u'Ă',u'A'},{u'ă',u'a'},{u'Â',u'A'},{u'â',u'a'},{u'Î',u'I'},{u'î',u'i'},{u'Ș',u'S'},{u'ș',u's'},{u'Ț',u'T'},{u'ț',u't'},
            {u'Ğ',u'G'},{u'ğ',u'g'},{u'İ',u'I'},{u'ı',u'i'},{u'Ş',u'S'},{u'ş',u's'},
            {u'À',u'A'},{u'à',u'a'},{u'Ç',u'C'},{u'ç',u'c'},{u'È',u'E'},{u'è',u'e'},{u'Ê',u'E'},{u'ê',u'e'},
            {u'Ë',u'E'},{u'ë',u'e'},{u'Ì',u'I'},{u'ì',u'i'},{u'Ï',u'I'},{u'ï',u'i'},{u'Ñ',u'N'},{u'ñ',u'n'},
            {u'Ò',u'O'},{u'ò',u'o'},{u'Ô',u'O'},{u'ô',u'o'},{u'Õ',u'O'},{u'õ',u'o'},{u'Ù',u'U'},{u'ù',u'u'},
            {u'Û',u'U'},{u'û',u'u'},{u'Ÿ',u'Y'},{u'ÿ',u'y'},
            {u'Ð',u'D'},{u'ð',u'd'},{u'Þ',u'T'},{u'þ',u't'},
        };
        for (auto& p : pairs) map[p[0]] = p[1];
    }
    // NormalizeWithStandardWhitespace (TextNormalizer.cs:137-200): \t \n \r -> ' ', char map, collapse runs of ' '.
    ustr normalize(uview text) const {
        ustr out; out.reserve(text.size());
        bool prevSpace = false;
        for (u16 o : text) {
            u16 m = (o == u'\t' || o == u'\n' || o == u'\r') ? u' ' : map[o];
            bool sp = (m == u' ');
            if (sp && prevSpace) continue;
            out.push_back(m);
            prevSpace = sp;
        }
        return out;
    }
};
inline const Normalizer& default_normalizer() { static Normalizer n; return n; }

// ---- delimiters (ConfigurationParameters.cs:58-62) -----------------------------------------
struct Delims {
    std::array<bool, 65536>* tbl;
    Delims() {
        tbl = new std::array<bool, 65536>();
        tbl->fill(false);
        const u16 d[] = {u' ',u'-',u'/',u'.',u',',u':',u';',u'\'',u'`',u'\u2013',u'\u2014',u'*',u'&',u'\\',u'_',u'(',u')',u'{',u'}',u'[',u']',u'\t'};
        for (u16 c : d) (*tbl)[c] = true;
    }
    bool is(u16 c) const { return (*tbl)[c]; }
};
inline const Delims& default_delims() { static Delims d; return d; }

struct Slice { int off, len; };

// string.Split(delims, RemoveEmptyEntries) / the tokenizer word loops: maximal runs of non-delimiters
inline void split_words(uview s, std::vector<Slice>& out) {
    const Delims& D = default_delims();
    out.clear();
    int n = (int)s.size(), i = 0;
    while (i < n) {
        while (i < n && D.is(s[i])) i++;
        if (i >= n) break;
        int st = i;
        while (i < n && !D.is(s[i])) i++;
        out.push_back({st, i - st});
    }
}

// ---- UTF-8 <-> UTF-16 (for the C API / fixtures) -------------------------------------------
inline ustr utf8_to_u16(const char* p, size_t n) {
    ustr r; r.reserve(n);
    size_t i = 0;
    while (i < n) {
        unsigned char c = (unsigned char)p[i];
        uint32_t cp;
        if (c < 0x80) { cp = c; i += 1; }
        else if ((c >> 5) == 6 && i + 1 < n) { cp = ((c & 0x1F) << 6) | (p[i+1] & 0x3F); i += 2; }
        else if ((c >> 4) == 14 && i + 2 < n) { cp = ((c & 0x0F) << 12) | ((p[i+1] & 0x3F) << 6) | (p[i+2] & 0x3F); i += 3; }
        else if ((c >> 3) == 30 && i + 3 < n) { cp = ((c & 0x07) << 18) | ((p[i+1] & 0x3F) << 12) | ((p[i+2] & 0x3F) << 6) | (p[i+3] & 0x3F); i += 4; }
        else { cp = 0xFFFD; i += 1; }
        if (cp >= 0x10000) { cp -= 0x10000; r.push_back((u16)(0xD800 + (cp >> 10))); r.push_back((u16)(0xDC00 + (cp & 0x3FF))); }
        else r.push_back((u16)cp);
    }
    return r;
}
inline ustr utf8_to_u16(const std::string& s) { return utf8_to_u16(s.data(), s.size()); }
inline std::string u16_to_utf8(uview s) {
    std::string r;
    for (size_t i = 0; i < s.size(); i++) {
        uint32_t cp = s[i];
        if (cp >= 0xD800 && cp <= 0xDBFF && i + 1 < s.size()) { cp = 0x10000 + ((cp - 0xD800) << 10) + (s[i+1] - 0xDC00); i++; }
        if (cp < 0x80) r.push_back((char)cp);
        else if (cp < 0x800) { r.push_back((char)(0xC0 | (cp >> 6))); r.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { r.push_back((char)(0xE0 | (cp >> 12))); r.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); r.push_back((char)(0x80 | (cp & 0x3F))); }
        else { r.push_back((char)(0xF0 | (cp >> 18))); r.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); r.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); r.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    return r;
}

} // namespace orc
