// Infiscript post-filter and facet aggregation, host side (BASELINE config 5).
//
// Reference: Api/FilterParser.cs (grammar), Filtering/FilterCompiler.cs + FilterVM.cs (evaluation semantics), Scoring/ResultProcessor.cs:35-70
// (post-filter of the returned rows; NumberOfDocumentsInFilter over the whole collection on first use), Core/FacetBuilder.cs:19-105.
//
// Design (not the reference's bytecode interpreter over boxed values):
//   * every non-indexed document field is a COLUMN, dictionary-encoded once: the distinct values (boxed: int64 / double / string) and one
//     uint32 code per document; the codes live in HBM.
//   * an expression is parsed into a postfix boolean program over LEAVES (field <op> constants).  A leaf depends on one field only, so it
//     is evaluated here, once per DISTINCT value, with the reference's coercion rules (FilterVM.AreEqual / CompareTo: case-insensitive
//     ToString() equality; numeric order when both sides parse as doubles, else case-insensitive string order) — the result is a bitmap
//     over the column's codes.  The device evaluates program + bitmaps per document (k_filter_count over all documents, k_postfilter over
//     the returned rows), so arbitrary string / number coercions cost nothing there.
//   * the reference's VM is untyped: AND / OR / ?: pass non-boolean operands through (a literal in a ternary branch).  The program keeps
//     that as a three-valued logic: F, T, N (not a bool): AND(l,r) = l==F ? F : r; OR(l,r) = l==T ? T : r; NOT(x) = x==T ? F : T;
//     TERN(c,a,b) = c==F ? b : a; a document matches iff the result is T  (FilterCompiler.cs:84-128,212-240; FilterVM.cs:26-45,136-148).
// MATCHES (a .NET regular expression) is rejected.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <unordered_map>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <stdexcept>

namespace infx { namespace filt {

struct Boxed { int kind = 0; long long i = 0; double d = 0; std::string s; };      // 0 null, 1 int64, 2 double, 3 string

inline std::string fmt_double(double x) {          // System.Double.ToString(): shortest round-trip digits, scientific outside [1e-4, 1e15)
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0" : "0";
    char buf[48]; int p = 0;
    do { snprintf(buf, sizeof buf, "%.*e", p, x); p++; } while (p <= 17 && strtod(buf, nullptr) != x);
    const char* e = strchr(buf, 'e'); int ex = atoi(e + 1);
    std::string dg; bool neg = false;
    for (const char* c = buf; c < e; c++) { if (*c == '-') neg = true; else if (*c != '.') dg.push_back(*c); }
    while (dg.size() > 1 && dg.back() == '0') dg.pop_back();
    std::string r;
    if (ex >= 15 || ex < -4) { r = dg.substr(0, 1); if (dg.size() > 1) r += "." + dg.substr(1); char t[12]; snprintf(t, sizeof t, "E%c%02d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex); r += t; }
    else if (ex < 0) r = "0." + std::string((size_t)(-ex - 1), '0') + dg;
    else if ((int)dg.size() > ex + 1) r = dg.substr(0, (size_t)ex + 1) + "." + dg.substr((size_t)ex + 1);
    else r = dg + std::string((size_t)ex + 1 - dg.size(), '0');
    return neg ? "-" + r : r;
}
inline std::string text_of(const Boxed& v) { return v.kind == 1 ? std::to_string(v.i) : v.kind == 2 ? fmt_double(v.d) : v.kind == 3 ? v.s : std::string(); }
inline bool as_number(const std::string& t, double& out) {        // double.TryParse: NumberStyles.Float | AllowThousands, invariant culture
    size_t a = 0, b = t.size();
    while (a < b && isspace((unsigned char)t[a])) a++;
    while (b > a && isspace((unsigned char)t[b - 1])) b--;
    if (a == b) return false;
    const std::string s = t.substr(a, b - a);
    if (s == "NaN") { out = NAN; return true; }
    if (s == "Infinity" || s == "+Infinity") { out = INFINITY; return true; }
    if (s == "-Infinity") { out = -INFINITY; return true; }
    std::string plain; plain.reserve(s.size());
    bool sawDigit = false, mantissaInt = true;            // group separators are legal only among the integer digits ("1,000", "12,34.5")
    for (unsigned char c : s) {
        if (isdigit(c)) { sawDigit = true; plain.push_back((char)c); continue; }
        if (c == ',') { if (mantissaInt && sawDigit) continue; return false; }
        if (c == '.' || c == 'e' || c == 'E') { mantissaInt = false; plain.push_back((char)c); continue; }
        if (c == '+' || c == '-') { plain.push_back((char)c); continue; }
        return false;
    }
    char* end = nullptr; out = strtod(plain.c_str(), &end);
    return end != plain.c_str() && *end == 0;
}
// OrdinalIgnoreCase (and RegexOptions.IgnoreCase for LIKE): .NET upper-cases each UTF-16 code unit with the invariant simple mapping and compares the
// units.  Case pairs are described as runs: {first lower, last lower, stride, delta to upper}; U+0131 / U+017F are left alone as in the BCL's ordinal
// casing.  Covers Latin (incl. the Czech school corpus), Greek, Cyrillic, Armenian, Latin Extended Additional.
struct CaseRun { uint16_t lo, hi, step; int32_t delta; };
inline uint16_t fold_unit(uint16_t c) {
    static const CaseRun runs[] = {
        {0x0061, 0x007A, 1, -32}, {0x00B5, 0x00B5, 1, 0x39C - 0xB5}, {0x00E0, 0x00F6, 1, -32}, {0x00F8, 0x00FE, 1, -32}, {0x00FF, 0x00FF, 1, 0x178 - 0xFF},
        {0x0101, 0x012F, 2, -1}, {0x0133, 0x0137, 2, -1}, {0x013A, 0x0148, 2, -1}, {0x014B, 0x0177, 2, -1}, {0x017A, 0x017E, 2, -1},
        {0x01CE, 0x01DC, 2, -1}, {0x01DF, 0x01EF, 2, -1}, {0x01F9, 0x021F, 2, -1}, {0x0223, 0x0233, 2, -1}, {0x0247, 0x024F, 2, -1},
        {0x03AC, 0x03AC, 1, 0x386 - 0x3AC}, {0x03AD, 0x03AF, 1, -0x25}, {0x03B1, 0x03C1, 1, -32}, {0x03C2, 0x03C2, 1, 0x3A3 - 0x3C2}, {0x03C3, 0x03CB, 1, -32},
        {0x03CC, 0x03CC, 1, 0x38C - 0x3CC}, {0x03CD, 0x03CE, 1, -0x3F},
        {0x0430, 0x044F, 1, -32}, {0x0450, 0x045F, 1, -80}, {0x0461, 0x0481, 2, -1}, {0x048B, 0x04BF, 2, -1}, {0x04C2, 0x04CE, 2, -1}, {0x04CF, 0x04CF, 1, 0x4C0 - 0x4CF},
        {0x04D1, 0x052F, 2, -1}, {0x0561, 0x0586, 1, -48}, {0x1E01, 0x1E95, 2, -1}, {0x1EA1, 0x1EFF, 2, -1},
    };
    for (const CaseRun& r : runs) if (c >= r.lo && c <= r.hi && (c - r.lo) % r.step == 0) return (uint16_t)(c + r.delta);
    return c;
}
inline std::u16string folded(const std::string& s) {               // UTF-8 -> folded UTF-16 units
    std::u16string o; o.reserve(s.size());
    size_t i = 0; const size_t n = s.size();
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp; size_t len;
        if (c < 0x80 || c < 0xC0) { cp = c; len = 1; }
        else if (c < 0xE0) { cp = c & 0x1Fu; len = 2; }
        else if (c < 0xF0) { cp = c & 0x0Fu; len = 3; }
        else { cp = c & 0x07u; len = 4; }
        if (i + len > n) { cp = c; len = 1; }
        for (size_t k = 1; k < len; k++) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3Fu);
        i += len;
        if (cp > 0xFFFF) { cp -= 0x10000; o.push_back((char16_t)(0xD800 | (cp >> 10))); o.push_back((char16_t)(0xDC00 | (cp & 0x3FF))); }
        else o.push_back((char16_t)fold_unit((uint16_t)cp));
    }
    return o;
}
inline int icmp(const std::string& a, const std::string& b) {     // StringComparison.OrdinalIgnoreCase
    const std::u16string x = folded(a), y = folded(b);
    return x < y ? -1 : (y < x ? 1 : 0);                           // char16_t compares as unsigned code units
}
inline std::string upper(std::string s) { for (auto& c : s) if (c >= 'a' && c <= 'z') c = (char)(c - 32); return s; }     // ASCII: keywords of the expression language
// value <op> constant with the VM's coercions; `isnull`: the document has no such field / a null value
inline bool same(bool isnull, const std::string& v, const std::string& c) { return !isnull && icmp(v, c) == 0; }          // constants are never null
inline int order(bool isnull, const std::string& v, const std::string& c) {
    if (isnull) return -1;
    double x, y;
    if (as_number(v, x) && as_number(c, y)) return x < y ? -1 : (x > y ? 1 : (x == y ? 0 : (std::isnan(x) ? (std::isnan(y) ? 0 : -1) : 1)));
    return icmp(v, c);
}
inline bool wildcard(const std::string& text0, const std::string& pat0) {     // LIKE: % any run, _ any one UTF-16 unit (not a newline), whole string, ignore case
    const std::u16string text = folded(text0), pat = folded(pat0);
    const size_t n = text.size(), m = pat.size();
    std::vector<char> prev(m + 1, 0), cur(m + 1, 0);
    prev[0] = 1; for (size_t j = 1; j <= m; j++) prev[j] = prev[j - 1] && pat[j - 1] == u'%';
    for (size_t i = 1; i <= n; i++) {
        cur[0] = 0;
        for (size_t j = 1; j <= m; j++) {
            const char16_t p = pat[j - 1]; const bool nl = text[i - 1] == u'\n';
            cur[j] = p == u'%' ? (cur[j - 1] || (prev[j] && !nl)) : p == u'_' ? (prev[j - 1] && !nl) : (prev[j - 1] && p == text[i - 1]);
        }
        prev.swap(cur);
    }
    return prev[m] != 0;
}

enum LeafOp { L_EQ, L_GT, L_GE, L_LT, L_LE, L_BETWEEN, L_IN, L_CONTAINS, L_STARTS, L_ENDS, L_LIKE, L_ISNULL, L_NOTNULL };
struct Leaf { std::string field; LeafOp op; std::vector<std::string> consts; };
enum POp : uint8_t { P_LEAF = 0, P_AND = 1, P_OR = 2, P_NOT = 3, P_TERN = 4, P_LIT = 5 };
struct PIns { uint8_t op; uint32_t arg; };
struct Program { std::vector<PIns> code; std::vector<Leaf> leaves; };
struct SyntaxError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

inline bool leaf_holds(const Leaf& L, const Boxed& v) {
    const bool isnull = v.kind == 0; const std::string t = text_of(v);
    switch (L.op) {
        case L_EQ: return same(isnull, t, L.consts[0]);
        case L_GT: return order(isnull, t, L.consts[0]) > 0;
        case L_GE: return order(isnull, t, L.consts[0]) >= 0;
        case L_LT: return order(isnull, t, L.consts[0]) < 0;
        case L_LE: return order(isnull, t, L.consts[0]) <= 0;
        case L_BETWEEN: return order(isnull, t, L.consts[0]) >= 0 && order(isnull, t, L.consts[1]) <= 0;
        case L_IN: for (auto& c : L.consts) if (same(isnull, t, c)) return true; return false;
        case L_CONTAINS: return folded(t).find(folded(L.consts[0])) != std::u16string::npos;
        case L_STARTS: { const std::u16string a = folded(t), b = folded(L.consts[0]); return a.size() >= b.size() && a.compare(0, b.size(), b) == 0; }
        case L_ENDS: { const std::u16string a = folded(t), b = folded(L.consts[0]); return a.size() >= b.size() && a.compare(a.size() - b.size(), b.size(), b) == 0; }
        case L_LIKE: return wildcard(t, L.consts[0]);
        case L_ISNULL: return isnull || (v.kind == 3 && v.s.empty());
        case L_NOTNULL: return !(isnull || (v.kind == 3 && v.s.empty()));
    }
    return false;
}

// ---- recursive descent straight to the postfix program (grammar of Api/FilterParser.cs:84-453; precedence ?: < OR < AND < NOT) -------------
class Reader {
    enum K { END, NAME, VALUE, CMP, KW_AND, KW_OR, KW_NOT, KW_BETWEEN, KW_IN, KW_CONTAINS, KW_STARTS, KW_ENDS, KW_LIKE, KW_MATCHES, KW_IS, KW_NULL, KW_WITH, LPAR, RPAR, COMMA, QM, COLON };
    struct T { K k; std::string s; };
    std::vector<T> toks; size_t at = 0; Program& P;
    K peek() const { return at < toks.size() ? toks[at].k : END; }
    std::string take() { return toks[at++].s; }
    void need(K k, const char* msg) { if (peek() != k) throw SyntaxError(msg); at++; }
    void lex(const std::string& e) {
        static const std::pair<const char*, K> kws[] = {{"AND", KW_AND}, {"OR", KW_OR}, {"NOT", KW_NOT}, {"BETWEEN", KW_BETWEEN}, {"IN", KW_IN}, {"CONTAINS", KW_CONTAINS}, {"STARTS", KW_STARTS},
            {"ENDS", KW_ENDS}, {"LIKE", KW_LIKE}, {"MATCHES", KW_MATCHES}, {"IS", KW_IS}, {"NULL", KW_NULL}, {"WITH", KW_WITH}};
        for (size_t i = 0; i < e.size();) {
            const unsigned char c = (unsigned char)e[i];
            if (isspace(c)) { i++; continue; }
            if (c == '(') { toks.push_back({LPAR, "("}); i++; } else if (c == ')') { toks.push_back({RPAR, ")"}); i++; }
            else if (c == ',') { toks.push_back({COMMA, ","}); i++; } else if (c == '?') { toks.push_back({QM, "?"}); i++; } else if (c == ':') { toks.push_back({COLON, ":"}); i++; }
            else if (c == '&') { i += (i + 1 < e.size() && e[i + 1] == '&') ? 2 : 1; toks.push_back({KW_AND, "&"}); }
            else if (c == '|') { i += (i + 1 < e.size() && e[i + 1] == '|') ? 2 : 1; toks.push_back({KW_OR, "|"}); }
            else if (c == '=' || c == '<' || c == '>') { std::string o(1, (char)c); i++; if (i < e.size() && e[i] == '=') { o.push_back('='); i++; } toks.push_back({CMP, o}); }
            else if (c == '!') { i++; if (i < e.size() && e[i] == '=') { toks.push_back({CMP, "!="}); i++; } else toks.push_back({KW_NOT, "!"}); }
            else if (c == '\'' || c == '"') { size_t j = e.find((char)c, i + 1); if (j == std::string::npos) throw SyntaxError("unterminated string literal"); toks.push_back({VALUE, e.substr(i + 1, j - i - 1)}); i = j + 1; }
            else if (isalpha(c) || c == '_' || c >= 0x80) {
                size_t j = i; while (j < e.size() && (isalnum((unsigned char)e[j]) || e[j] == '_' || (unsigned char)e[j] >= 0x80)) j++;
                std::string w = e.substr(i, j - i), u = upper(w); K k = NAME;
                for (auto& kw : kws) if (u == kw.first) k = kw.second;
                toks.push_back({k, w}); i = j;
            }
            else if (isdigit(c)) { size_t j = i; while (j < e.size() && (isdigit((unsigned char)e[j]) || e[j] == '.')) j++; toks.push_back({VALUE, e.substr(i, j - i)}); i = j; }
            else throw SyntaxError("unexpected character");
        }
    }
    void emit(uint8_t op, uint32_t arg = 0) { P.code.push_back({op, arg}); }
    void leaf(const std::string& f, LeafOp op, std::vector<std::string> c) { P.leaves.push_back({f, op, std::move(c)}); emit(P_LEAF, (uint32_t)P.leaves.size() - 1); }
    std::string value(const char* msg) { if (peek() != VALUE) throw SyntaxError(msg); return take(); }
    void ternary() { disj(); if (peek() == QM) { at++; ternary(); need(COLON, "expected ':' in a ternary expression"); ternary(); emit(P_TERN); } }
    void disj() { conj(); while (peek() == KW_OR) { at++; conj(); emit(P_OR); } }
    void conj() { unary(); while (peek() == KW_AND) { at++; unary(); emit(P_AND); } }
    void unary() {
        if (peek() == KW_NOT) { at++; unary(); emit(P_NOT); return; }
        if (peek() == LPAR) { at++; ternary(); need(RPAR, "expected ')'"); return; }
        if (peek() == VALUE) { at++; emit(P_LIT); return; }              // a bare literal (ternary branch): never a bool
        cond();
    }
    void cond() {
        if (peek() != NAME) throw SyntaxError("expected a field name");
        const std::string f = take();
        switch (peek()) {
            case KW_IN: { at++; need(LPAR, "expected '(' after IN"); std::vector<std::string> vs; while (peek() != RPAR && peek() != END) { vs.push_back(value("expected a value in the IN list")); if (peek() == COMMA) at++; }
                          need(RPAR, "expected ')' after the IN list"); leaf(f, L_IN, vs); return; }
            case KW_CONTAINS: at++; leaf(f, L_CONTAINS, {value("expected a value after CONTAINS")}); return;
            case KW_STARTS: at++; need(KW_WITH, "expected WITH after STARTS"); leaf(f, L_STARTS, {value("expected a value after STARTS WITH")}); return;
            case KW_ENDS: at++; need(KW_WITH, "expected WITH after ENDS"); leaf(f, L_ENDS, {value("expected a value after ENDS WITH")}); return;
            case KW_LIKE: at++; leaf(f, L_LIKE, {value("expected a pattern after LIKE")}); return;
            case KW_MATCHES: throw Unsupported("MATCHES (.NET regular expressions) is not supported");
            case KW_IS: { at++; bool neg = false; if (peek() == KW_NOT) { neg = true; at++; } need(KW_NULL, "expected NULL after IS"); leaf(f, neg ? L_NOTNULL : L_ISNULL, {}); return; }
            case KW_BETWEEN: { at++; std::string lo = value("expected a value after BETWEEN"); need(KW_AND, "expected AND in BETWEEN"); std::string hi = value("expected a value after AND"); leaf(f, L_BETWEEN, {lo, hi}); return; }
            case CMP: { const std::string o = take(); const std::string v = value("expected a value after the operator");
                        if (o == "=") leaf(f, L_EQ, {v}); else if (o == "!=") { leaf(f, L_EQ, {v}); emit(P_NOT); } else if (o == ">") leaf(f, L_GT, {v}); else if (o == ">=") leaf(f, L_GE, {v});
                        else if (o == "<") leaf(f, L_LT, {v}); else if (o == "<=") leaf(f, L_LE, {v}); else throw SyntaxError("unknown operator"); return; }
            default: throw SyntaxError("expected a comparison operator");
        }
    }
public:
    explicit Reader(Program& p) : P(p) {}
    void run(const std::string& e) {
        bool blank = true; for (unsigned char c : e) if (!isspace(c)) blank = false;
        if (blank) throw SyntaxError("filter expression cannot be empty");
        lex(e); ternary();
        if (at < toks.size()) throw SyntaxError("unexpected token after a complete expression");
    }
};
inline Program parse(const std::string& expr) { Program p; Reader(p).run(expr); return p; }

// ---- columns ------------------------------------------------------------------------------------------------------------------
struct Column {
    std::string name; bool facetable = false;
    std::vector<Boxed> dict;                // distinct values, code = position
    std::vector<uint32_t> codes;            // per document
    std::vector<uint32_t> rank;             // facet tie order of the codes: value ascending (FacetBuilder.cs:44 ThenBy key)
    std::vector<std::string> text;          // ToString() of every distinct value (facet keys)
};
template <class Key, class Get> inline void encode_column(Column& c, size_t n, Get get, Key) {
    std::unordered_map<Key, uint32_t> ids; c.codes.resize(n);
    for (size_t d = 0; d < n; d++) { Boxed v = get(d); Key k; if constexpr (std::is_same<Key, std::string>::value) k = v.s; else if constexpr (std::is_same<Key, long long>::value) k = v.i; else { uint64_t b; std::memcpy(&b, &v.d, 8); k = b; }
        auto it = ids.find(k); if (it == ids.end()) { it = ids.emplace(k, (uint32_t)c.dict.size()).first; c.dict.push_back(v); } c.codes[d] = it->second; }
    c.text.resize(c.dict.size()); for (size_t i = 0; i < c.dict.size(); i++) c.text[i] = text_of(c.dict[i]);
    std::vector<uint32_t> ord(c.dict.size()); for (size_t i = 0; i < ord.size(); i++) ord[i] = (uint32_t)i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { const int x = icmp(c.text[a], c.text[b]); return x ? x < 0 : c.text[a] < c.text[b]; });
    c.rank.resize(ord.size()); for (size_t i = 0; i < ord.size(); i++) c.rank[ord[i]] = (uint32_t)i;
}

// leaf -> bitmap over the codes of its column (bit = the leaf holds for that distinct value); an unknown field is null for every document
inline void leaf_table(const Leaf& L, const Column* col, std::vector<uint32_t>& words) {
    const size_t nv = col ? col->dict.size() : 1;
    words.assign((nv + 31) / 32, 0u);
    for (size_t v = 0; v < nv; v++) { const Boxed b = col ? col->dict[v] : Boxed(); if (leaf_holds(L, b)) words[v >> 5] |= 1u << (v & 31); }
}

}} // namespace
