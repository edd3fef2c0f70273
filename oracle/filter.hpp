// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// CPU restatement of the reference's Infiscript post-filter and facet aggregation (BASELINE config 5).
//
// Follows (paths relative to /root/reference/src/Infidex):
//   Api/FilterParser.cs:35-693          Tokenize, ParseTernaryExpression / ParseExpression / ParseTerm / ParseFactor / ParseCondition
//   Api/{ValueFilter,RangeFilter,InFilter,StringFilter,NullFilter,CompositeFilter,TernaryFilter,LiteralFilter}.cs   (tree node kinds)
//   Filtering/FilterCompiler.cs:19-283  tree -> bytecode (short-circuit AND/OR through DUP / JUMP_IF_* / POP, ternary jumps)
//   Filtering/BytecodeInstruction.cs:3-47   opcodes
//   Filtering/FilterVM.cs:26-359        stack VM; AreEqual = OrdinalIgnoreCase equality of ToString(); CompareTo = numeric if both sides
//                                        double.TryParse, else OrdinalIgnoreCase string compare; `as bool? ?? false` for AND/OR/NOT operands
//   Scoring/ResultProcessor.cs:35-70    ApplyFilter: post-filter of the <= k result rows; first use counts the matches over ALL documents
//   Core/FacetBuilder.cs:19-105         value counts of the facetable fields over the result rows, (count desc, value asc), <= 100 per field
//   SearchEngine.cs:298-316             Execute(...maxResults) -> ApplyPostProcessing -> facets -> Take(max)
// NOT restated (an expression using them is rejected by the oracle AND by the product): MATCHES (a .NET regex), SortBy / Boosts.
// PARITY UNPINNED (BCL behaviour outside /root/reference): double.ToString() ("shortest round-trip" digits, restated), double.TryParse
// (NumberStyles.Float | AllowThousands under the current culture: restated for invariant-culture plain / exponent numbers, no thousands
// separators), and the facet tie order `ThenBy(kvp => kvp.Key)` (current-culture string comparer: restated as ordinal-ignore-case, then
// ordinal — identical for the digit strings and capitalised words the reference's tests and the synthetic config use).
#pragma once
#include "text.hpp"
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <stdexcept>

namespace orc { namespace flt {

// ---- boxed values ------------------------------------------------------------------------------------------------
struct Value {
    enum Kind { Null, Bool, Str, Int, Dbl, Arr } kind = Null;
    bool b = false; std::string s; long long i = 0; double d = 0; std::vector<std::string> arr;
    static Value str(std::string x) { Value v; v.kind = Str; v.s = std::move(x); return v; }
    static Value num(double x) { Value v; v.kind = Dbl; v.d = x; return v; }
    static Value integer(long long x) { Value v; v.kind = Int; v.i = x; return v; }
    static Value boolean(bool x) { Value v; v.kind = Bool; v.b = x; return v; }
};

// double.ToString(): shortest digits that round-trip; fixed notation for exponents in [-4, 15) (0.0001 -> "0.0001", 0.00001 -> "1E-05"), else d.dddE+XX
inline std::string dbl_to_string(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
    if (x == 0) return std::signbit(x) ? "-0" : "0";
    char buf[64]; int prec = 1;
    for (; prec <= 17; prec++) { snprintf(buf, sizeof buf, "%.*e", prec - 1, x); if (strtod(buf, nullptr) == x) break; }
    std::string e(buf); size_t ep = e.find('e');
    std::string mant = e.substr(0, ep); int exp10 = atoi(e.c_str() + ep + 1);
    bool neg = mant[0] == '-'; if (neg) mant.erase(0, 1);
    std::string digits; for (char c : mant) if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    if (exp10 >= -4 && exp10 < 15) {
        if (exp10 >= 0) {
            if ((int)digits.size() <= exp10 + 1) { out = digits + std::string(exp10 + 1 - digits.size(), '0'); }
            else out = digits.substr(0, exp10 + 1) + "." + digits.substr(exp10 + 1);
        } else out = "0." + std::string(-exp10 - 1, '0') + digits;
    } else {
        out = digits.substr(0, 1); if (digits.size() > 1) out += "." + digits.substr(1);
        char eb[16]; snprintf(eb, sizeof eb, "E%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10)); out += eb;
    }
    return neg ? "-" + out : out;
}
inline std::string to_string(const Value& v) {      // object.ToString()
    switch (v.kind) {
        case Value::Bool: return v.b ? "True" : "False";
        case Value::Str: return v.s;
        case Value::Int: return std::to_string(v.i);
        case Value::Dbl: return dbl_to_string(v.d);
        case Value::Arr: return "System.Object[]";
        default: return "";
    }
}
inline bool try_parse_double(const std::string& s0, double& out) {      // double.TryParse(string): NumberStyles.Float | AllowThousands, invariant culture
    size_t a = 0, b = s0.size();
    while (a < b && isspace((unsigned char)s0[a])) a++;
    while (b > a && isspace((unsigned char)s0[b - 1])) b--;
    if (a == b) return false;
    std::string s = s0.substr(a, b - a);
    if (s == "Infinity" || s == "+Infinity") { out = INFINITY; return true; }
    if (s == "-Infinity") { out = -INFINITY; return true; }
    if (s == "NaN") { out = NAN; return true; }
    // AllowThousands: ',' is accepted inside the integer part once a digit has been read (group sizes are not checked by the BCL parser)
    std::string t; bool digitSeen = false, intPart = true;
    for (char c : s) {
        if (isdigit((unsigned char)c)) { digitSeen = true; t.push_back(c); }
        else if (c == ',' && intPart && digitSeen) continue;
        else if (c == '.' || c == 'e' || c == 'E') { intPart = false; t.push_back(c); }
        else if (c == '+' || c == '-') t.push_back(c);
        else return false;
    }
    char* end = nullptr; out = strtod(t.c_str(), &end);
    return end && *end == 0 && end != t.c_str();
}
// StringComparison.OrdinalIgnoreCase / RegexOptions.IgnoreCase: both sides are upper-cased per UTF-16 code unit with the invariant SIMPLE case mapping
// and compared ordinally.  Restated for the scripts the corpora use (Basic Latin, Latin-1, Latin Extended-A/B pairs, Greek, Cyrillic, Armenian);
// U+0131 and U+017F keep their value (the BCL's ordinal casing does not fold them onto ASCII I / S).  Strings here are UTF-8.
inline uint32_t up_cp(uint32_t c) {
    if (c < 0x80) return (c >= 'a' && c <= 'z') ? c - 32 : c;
    if (c == 0xB5) return 0x39C;
    if (c >= 0xE0 && c <= 0xFE && c != 0xF7) return c - 0x20;
    if (c == 0xFF) return 0x178;
    if (c >= 0x100 && c <= 0x17F) {
        if (c == 0x131 || c == 0x138 || c == 0x149 || c == 0x17F) return c;
        if ((c >= 0x139 && c <= 0x148) || (c >= 0x179 && c <= 0x17E)) return (c & 1) ? c : c - 1;      // upper = odd code point
        return (c & 1) ? c - 1 : c;                                                                   // upper = even code point
    }
    if (c >= 0x180 && c <= 0x24F) {
        if ((c >= 0x1CD && c <= 0x1DC)) return (c & 1) ? c : c - 1;
        if ((c >= 0x1DE && c <= 0x1EF) || (c >= 0x1F8 && c <= 0x21F) || (c >= 0x222 && c <= 0x233) || (c >= 0x246 && c <= 0x24F)) return (c & 1) ? c - 1 : c;
        return c;
    }
    if (c == 0x3AC) return 0x386; if (c >= 0x3AD && c <= 0x3AF) return c - 0x25; if (c == 0x3CC) return 0x38C; if (c == 0x3CD || c == 0x3CE) return c - 0x3F;
    if (c == 0x3C2) return 0x3A3;
    if (c >= 0x3B1 && c <= 0x3CB) return c - 0x20;
    if (c >= 0x430 && c <= 0x44F) return c - 0x20;
    if (c >= 0x450 && c <= 0x45F) return c - 0x50;
    if ((c >= 0x460 && c <= 0x481) || (c >= 0x48A && c <= 0x4BF) || (c >= 0x4D0 && c <= 0x52F)) return (c & 1) ? c - 1 : c;
    if (c >= 0x4C1 && c <= 0x4CE) return (c & 1) ? c : c - 1;
    if (c == 0x4CF) return 0x4C0;
    if (c >= 0x561 && c <= 0x586) return c - 0x30;
    if (c >= 0x1E00 && c <= 0x1EFF) { if (c >= 0x1E96 && c <= 0x1E9F) return c; return (c & 1) ? c - 1 : c; }
    return c;
}
inline std::vector<uint16_t> up_units(const std::string& s) {       // UTF-8 -> upper-cased UTF-16 code units (invalid bytes pass through as single units)
    std::vector<uint16_t> o; o.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        unsigned char c = (unsigned char)s[i]; uint32_t cp = c; int n = 1;
        if (c >= 0xF0 && i + 3 < s.size()) { cp = ((c & 7u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) | ((unsigned char)s[i + 3] & 0x3Fu); n = 4; }
        else if (c >= 0xE0 && c < 0xF0 && i + 2 < s.size()) { cp = ((c & 0xFu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu); n = 3; }
        else if (c >= 0xC0 && c < 0xE0 && i + 1 < s.size()) { cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); n = 2; }
        i += (size_t)n;
        if (cp >= 0x10000) { cp -= 0x10000; o.push_back((uint16_t)(0xD800 + (cp >> 10))); o.push_back((uint16_t)(0xDC00 + (cp & 0x3FF))); }
        else o.push_back((uint16_t)up_cp(cp));
    }
    return o;
}
inline int cmp_oic(const std::string& a0, const std::string& b0) {
    const std::vector<uint16_t> a = up_units(a0), b = up_units(b0);
    size_t n = std::min(a.size(), b.size());
    for (size_t i = 0; i < n; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0);
}
inline bool are_equal(const Value* l, const Value* r) {          // FilterVM.AreEqual :329-338
    if (!l && !r) return true;
    if (!l || !r) return false;
    return cmp_oic(to_string(*l), to_string(*r)) == 0;
}
inline int compare_to(const Value* l, const Value* r) {           // FilterVM.CompareTo :340-358
    if (!l && !r) return 0;
    if (!l) return -1;
    if (!r) return 1;
    std::string ls = to_string(*l), rs = to_string(*r);
    double a, b;
    if (try_parse_double(ls, a) && try_parse_double(rs, b)) return a < b ? -1 : (a > b ? 1 : (a == b ? 0 : (std::isnan(a) ? (std::isnan(b) ? 0 : -1) : 1)));
    return cmp_oic(ls, rs);
}
inline bool like_match(const std::string& text0, const std::string& pat0) {      // ^escape(pat) with % -> .*, _ -> . $, IgnoreCase; '.' = one UTF-16 unit
    const std::vector<uint16_t> text = up_units(text0), pat = up_units(pat0);
    size_t n = text.size(), m = pat.size();
    std::vector<std::vector<char>> dp(n + 1, std::vector<char>(m + 1, 0));
    dp[0][0] = 1;
    for (size_t j = 1; j <= m; j++) dp[0][j] = dp[0][j - 1] && pat[j - 1] == '%';
    for (size_t i = 1; i <= n; i++) for (size_t j = 1; j <= m; j++) {
        uint16_t p = pat[j - 1];
        if (p == '%') dp[i][j] = dp[i][j - 1] || (dp[i - 1][j] && text[i - 1] != '\n');
        else if (p == '_') dp[i][j] = dp[i - 1][j - 1] && text[i - 1] != '\n';
        else dp[i][j] = dp[i - 1][j - 1] && p == text[i - 1];
    }
    return dp[n][m] != 0;
}

// ---- filter tree (Api/*Filter.cs) -----------------------------------------------------------------------------------
struct Node {
    enum Kind { ValueF, RangeF, InF, StringF, RegexF, NullF, And, Or, Not, Ternary, Literal } kind;
    std::string field, v1, v2; bool has1 = false, has2 = false, inc1 = true, inc2 = true;     // Range: min / max
    std::vector<std::string> values; int strOp = 0; bool isNull = true;
    bool litIsNum = false; double litNum = 0;
    std::shared_ptr<Node> a, b, c;
};
using NodeP = std::shared_ptr<Node>;
struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

struct Tok { enum T { Ident, Op, Val, And, Or, Not, Between, In, Contains, Starts, Ends, Like, Matches, Is, Null, With, LP, RP, Comma, Q, Colon } t; std::string v; };

inline std::vector<Tok> tokenize(const std::string& e) {          // FilterParser.Tokenize :456-650
    std::vector<Tok> out; size_t i = 0;
    auto isl = [](unsigned char c) { return isalpha(c) || c >= 0x80; };
    while (i < e.size()) {
        unsigned char c = (unsigned char)e[i];
        if (isspace(c)) { i++; continue; }
        if (c == '(') { out.push_back({Tok::LP, "("}); i++; continue; }
        if (c == ')') { out.push_back({Tok::RP, ")"}); i++; continue; }
        if (c == ',') { out.push_back({Tok::Comma, ","}); i++; continue; }
        if (c == '?') { out.push_back({Tok::Q, "?"}); i++; continue; }
        if (c == ':') { out.push_back({Tok::Colon, ":"}); i++; continue; }
        if (c == '&') { if (i + 1 < e.size() && e[i + 1] == '&') { out.push_back({Tok::And, "&&"}); i += 2; } else { out.push_back({Tok::And, "&"}); i++; } continue; }
        if (c == '|') { if (i + 1 < e.size() && e[i + 1] == '|') { out.push_back({Tok::Or, "||"}); i += 2; } else { out.push_back({Tok::Or, "|"}); i++; } continue; }
        if (c == '=' || c == '<' || c == '>') { std::string op(1, (char)c); i++; if (i < e.size() && e[i] == '=') { op += '='; i++; } out.push_back({Tok::Op, op}); continue; }
        if (c == '!') { i++; if (i < e.size() && e[i] == '=') { out.push_back({Tok::Op, "!="}); i++; } else out.push_back({Tok::Not, "!"}); continue; }
        if (c == '\'' || c == '"') {
            char q = (char)c; i++; std::string s;
            while (i < e.size() && e[i] != q) s.push_back(e[i++]);
            if (i >= e.size()) throw ParseError("Unterminated string literal");
            i++; out.push_back({Tok::Val, s}); continue;
        }
        if (isl(c) || c == '_') {
            std::string w; while (i < e.size() && (isl((unsigned char)e[i]) || isdigit((unsigned char)e[i]) || e[i] == '_')) w.push_back(e[i++]);
            std::string u = w; for (auto& ch : u) ch = (char)up_cp((unsigned char)ch);      // keywords are ASCII
            Tok::T t = Tok::Ident;
            if (u == "AND") t = Tok::And; else if (u == "OR") t = Tok::Or; else if (u == "NOT") t = Tok::Not; else if (u == "BETWEEN") t = Tok::Between;
            else if (u == "IN") t = Tok::In; else if (u == "CONTAINS") t = Tok::Contains; else if (u == "STARTS") t = Tok::Starts; else if (u == "ENDS") t = Tok::Ends;
            else if (u == "LIKE") t = Tok::Like; else if (u == "MATCHES") t = Tok::Matches; else if (u == "IS") t = Tok::Is; else if (u == "NULL") t = Tok::Null; else if (u == "WITH") t = Tok::With;
            out.push_back({t, w}); continue;
        }
        if (isdigit(c)) { std::string s; while (i < e.size() && (isdigit((unsigned char)e[i]) || e[i] == '.')) s.push_back(e[i++]); out.push_back({Tok::Val, s}); continue; }
        throw ParseError(std::string("Unexpected character: ") + (char)c);
    }
    return out;
}

struct Parser {
    std::vector<Tok> t; size_t p = 0;
    bool at(Tok::T k) const { return p < t.size() && t[p].t == k; }
    static NodeP mk(Node::Kind k) { auto n = std::make_shared<Node>(); n->kind = k; return n; }
    NodeP ternary() {                                              // :84-115
        NodeP cond = expr();
        if (at(Tok::Q)) { p++; NodeP tv = ternary(); if (!at(Tok::Colon)) throw ParseError("Expected ':'"); p++; NodeP fv = ternary();
            NodeP n = mk(Node::Ternary); n->a = cond; n->b = tv; n->c = fv; return n; }
        return cond;
    }
    NodeP expr() { NodeP l = term(); while (at(Tok::Or)) { p++; NodeP r = term(); NodeP n = mk(Node::Or); n->a = l; n->b = r; l = n; } return l; }      // :117-136
    NodeP term() { NodeP l = factor(); while (at(Tok::And)) { p++; NodeP r = factor(); NodeP n = mk(Node::And); n->a = l; n->b = r; l = n; } return l; }   // :138-150
    NodeP factor() {                                               // :152-195
        if (at(Tok::Not)) { p++; NodeP n = mk(Node::Not); n->a = factor(); return n; }
        if (at(Tok::LP)) { p++; NodeP in = ternary(); if (!at(Tok::RP)) throw ParseError("Expected ')'"); p++; return in; }
        if (at(Tok::Val)) { NodeP n = mk(Node::Literal); n->v1 = t[p].v; p++; double d; if (try_parse_double(n->v1, d)) { n->litIsNum = true; n->litNum = d; } return n; }
        return condition();
    }
    std::string val(const char* what) { if (!at(Tok::Val)) throw ParseError(std::string("Expected value ") + what); return t[p++].v; }
    NodeP condition() {                                            // :197-453
        if (!at(Tok::Ident)) throw ParseError("Expected field name");
        std::string f = t[p++].v;
        if (at(Tok::In)) { p++; if (!at(Tok::LP)) throw ParseError("Expected '(' after IN"); p++; NodeP n = mk(Node::InF); n->field = f;
            while (p < t.size() && !at(Tok::RP)) { if (!at(Tok::Val)) throw ParseError("Expected value in IN list"); n->values.push_back(t[p++].v); if (at(Tok::Comma)) p++; }
            if (!at(Tok::RP)) throw ParseError("Expected ')' after IN list"); p++; return n; }
        if (at(Tok::Contains)) { p++; NodeP n = mk(Node::StringF); n->field = f; n->strOp = 0; n->v1 = val("after CONTAINS"); return n; }
        if (at(Tok::Starts)) { p++; if (!at(Tok::With)) throw ParseError("Expected WITH"); p++; NodeP n = mk(Node::StringF); n->field = f; n->strOp = 1; n->v1 = val("after STARTS WITH"); return n; }
        if (at(Tok::Ends)) { p++; if (!at(Tok::With)) throw ParseError("Expected WITH"); p++; NodeP n = mk(Node::StringF); n->field = f; n->strOp = 2; n->v1 = val("after ENDS WITH"); return n; }
        if (at(Tok::Like)) { p++; NodeP n = mk(Node::StringF); n->field = f; n->strOp = 3; n->v1 = val("after LIKE"); return n; }
        if (at(Tok::Matches)) { p++; NodeP n = mk(Node::RegexF); n->field = f; n->v1 = val("after MATCHES"); return n; }
        if (at(Tok::Is)) { p++; bool isNot = false; if (at(Tok::Not)) { isNot = true; p++; } if (!at(Tok::Null)) throw ParseError("Expected NULL"); p++;
            NodeP n = mk(Node::NullF); n->field = f; n->isNull = !isNot; return n; }
        if (at(Tok::Between)) { p++; NodeP n = mk(Node::RangeF); n->field = f; n->v1 = val("after BETWEEN"); n->has1 = true; if (!at(Tok::And)) throw ParseError("Expected AND"); p++;
            n->v2 = val("after AND"); n->has2 = true; return n; }
        if (!at(Tok::Op)) throw ParseError("Expected comparison operator");
        std::string op = t[p++].v; std::string v = val("after operator");
        if (op == "=") { NodeP n = mk(Node::ValueF); n->field = f; n->v1 = v; return n; }
        if (op == "!=") { NodeP n = mk(Node::ValueF); n->field = f; n->v1 = v; NodeP nn = mk(Node::Not); nn->a = n; return nn; }
        NodeP n = mk(Node::RangeF); n->field = f;
        if (op == ">") { n->v1 = v; n->has1 = true; n->inc1 = false; } else if (op == ">=") { n->v1 = v; n->has1 = true; n->inc1 = true; }
        else if (op == "<") { n->v2 = v; n->has2 = true; n->inc2 = false; } else if (op == "<=") { n->v2 = v; n->has2 = true; n->inc2 = true; }
        else throw ParseError("Unknown operator " + op);
        return n;
    }
};
inline NodeP parse(const std::string& e) {                          // FilterParser.Parse :39-70
    bool ws = true; for (char c : e) if (!isspace((unsigned char)c)) ws = false;
    if (ws) throw ParseError("Filter expression cannot be empty");
    Parser P; P.t = tokenize(e);
    NodeP r = P.ternary();
    if (P.p < P.t.size()) throw ParseError("Unexpected token after complete expression");
    return r;
}

// ---- bytecode (FilterCompiler / BytecodeInstruction) -------------------------------------------------------------------------
enum Op : uint8_t { PUSH_FIELD = 0x01, PUSH_CONST = 0x02, POP = 0x03, DUP = 0x04, EQ = 0x10, NEQ = 0x11, LT = 0x12, LTE = 0x13, GT = 0x14, GTE = 0x15,
                    AND = 0x20, OR = 0x21, NOT = 0x22, CONTAINS = 0x30, STARTS_WITH = 0x31, ENDS_WITH = 0x32, LIKE = 0x33, MATCHES = 0x34, IN = 0x40, BETWEEN = 0x41,
                    IS_NULL = 0x50, IS_NOT_NULL = 0x51, JUMP = 0x60, JUMP_IF_FALSE = 0x61, JUMP_IF_TRUE = 0x62, HALT = 0xFF };
struct Ins { Op op; int a; };
struct Compiled { std::vector<Value> pool; std::vector<Ins> code; };
struct Compiler {
    Compiled c;
    int add(const Value& v) { c.pool.push_back(v); return (int)c.pool.size() - 1; }      // ConstantPool: indices only matter through Get()
    void emit(Op o, int a = 0) { c.code.push_back({o, a}); }
    void comp(const NodeP& n) {
        switch (n->kind) {
            case Node::And: { comp(n->a); emit(DUP); size_t j = c.code.size(); emit(JUMP_IF_FALSE, 0); emit(POP); comp(n->b); c.code[j].a = (int)c.code.size(); break; }
            case Node::Or: { comp(n->a); emit(DUP); size_t j = c.code.size(); emit(JUMP_IF_TRUE, 0); emit(POP); comp(n->b); c.code[j].a = (int)c.code.size(); break; }
            case Node::Not: comp(n->a); emit(NOT); break;
            case Node::ValueF: emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(Value::str(n->v1))); emit(EQ); break;
            case Node::RangeF:
                if (n->has1 && n->has2) { emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(Value::str(n->v1))); emit(PUSH_CONST, add(Value::str(n->v2))); emit(BETWEEN); }
                else if (n->has1) { emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(Value::str(n->v1))); emit(n->inc1 ? GTE : GT); }
                else if (n->has2) { emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(Value::str(n->v2))); emit(n->inc2 ? LTE : LT); }
                break;
            case Node::InF: { Value a; a.kind = Value::Arr; a.arr = n->values; emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(a)); emit(IN); break; }
            case Node::StringF: emit(PUSH_FIELD, add(Value::str(n->field))); emit(PUSH_CONST, add(Value::str(n->v1)));
                emit(n->strOp == 0 ? CONTAINS : n->strOp == 1 ? STARTS_WITH : n->strOp == 2 ? ENDS_WITH : LIKE); break;
            case Node::RegexF: throw ParseError("MATCHES is not restated");
            case Node::NullF: emit(PUSH_FIELD, add(Value::str(n->field))); emit(n->isNull ? IS_NULL : IS_NOT_NULL); break;
            case Node::Ternary: {                                    // CompileTernary :212-240
                comp(n->a); size_t jf = c.code.size(); emit(JUMP_IF_FALSE, 0); emit(POP); comp(n->b); size_t je = c.code.size(); emit(JUMP, 0);
                c.code[jf].a = (int)c.code.size(); emit(POP); comp(n->c); c.code[je].a = (int)c.code.size(); break; }
            case Node::Literal: emit(PUSH_CONST, add(n->litIsNum ? Value::num(n->litNum) : Value::str(n->v1))); break;
        }
    }
};
inline Compiled compile(const NodeP& n) { Compiler c; c.comp(n); c.emit(HALT); return c.c; }

using Fields = std::map<std::string, Value>;       // DocumentFields: name -> boxed value (absent or Null kind = null)

inline bool execute(const Compiled& f, const Fields& doc) {          // FilterVM.Execute :26-45
    struct Slot { bool isNull; Value v; };
    std::vector<Slot> st;
    auto pop = [&]() -> Slot { if (st.empty()) throw std::runtime_error("stack underflow"); Slot s = st.back(); st.pop_back(); return s; };
    auto pushb = [&](bool b) { st.push_back({false, Value::boolean(b)}); };
    auto asbool = [](const Slot& s) { return !s.isNull && s.v.kind == Value::Bool && s.v.b; };
    auto ptr = [](const Slot& s) -> const Value* { return s.isNull ? nullptr : &s.v; };
    auto str = [](const Slot& s) { return s.isNull ? std::string() : to_string(s.v); };
    for (size_t ip = 0; ip < f.code.size(); ip++) {
        const Ins in = f.code[ip];
        switch (in.op) {
            case PUSH_FIELD: { auto it = doc.find(f.pool[in.a].s); if (it == doc.end() || it->second.kind == Value::Null) st.push_back({true, Value()}); else st.push_back({false, it->second}); break; }
            case PUSH_CONST: st.push_back({false, f.pool[in.a]}); break;
            case POP: pop(); break;
            case DUP: { Slot s = st.back(); st.push_back(s); break; }
            case EQ: { Slot r = pop(), l = pop(); pushb(are_equal(ptr(l), ptr(r))); break; }
            case NEQ: { Slot r = pop(), l = pop(); pushb(!are_equal(ptr(l), ptr(r))); break; }
            case LT: { Slot r = pop(), l = pop(); pushb(compare_to(ptr(l), ptr(r)) < 0); break; }
            case LTE: { Slot r = pop(), l = pop(); pushb(compare_to(ptr(l), ptr(r)) <= 0); break; }
            case GT: { Slot r = pop(), l = pop(); pushb(compare_to(ptr(l), ptr(r)) > 0); break; }
            case GTE: { Slot r = pop(), l = pop(); pushb(compare_to(ptr(l), ptr(r)) >= 0); break; }
            case AND: { Slot r = pop(), l = pop(); pushb(asbool(l) && asbool(r)); break; }
            case OR: { Slot r = pop(), l = pop(); pushb(asbool(l) || asbool(r)); break; }
            case NOT: { Slot v = pop(); pushb(!asbool(v)); break; }
            case CONTAINS: case STARTS_WITH: case ENDS_WITH: case LIKE: {
                Slot p = pop(), t = pop(); std::string pat = str(p), text = str(t);
                const std::vector<uint16_t> P = up_units(pat), T = up_units(text);      // StringComparison.OrdinalIgnoreCase
                bool r;
                if (in.op == CONTAINS) r = std::search(T.begin(), T.end(), P.begin(), P.end()) != T.end();
                else if (in.op == STARTS_WITH) r = T.size() >= P.size() && std::equal(P.begin(), P.end(), T.begin());
                else if (in.op == ENDS_WITH) r = T.size() >= P.size() && std::equal(P.begin(), P.end(), T.end() - (long)P.size());
                else r = like_match(text, pat);
                pushb(r); break; }
            case MATCHES: throw std::runtime_error("MATCHES is not restated");
            case IN: { Slot a = pop(), v = pop(); bool found = false;
                if (!a.isNull && a.v.kind == Value::Arr) for (auto& item : a.v.arr) { Value iv = Value::str(item); if (are_equal(ptr(v), &iv)) { found = true; break; } }
                pushb(found); break; }
            case BETWEEN: { Slot mx = pop(), mn = pop(), v = pop(); pushb(compare_to(ptr(v), ptr(mn)) >= 0 && compare_to(ptr(v), ptr(mx)) <= 0); break; }
            case IS_NULL: case IS_NOT_NULL: { Slot v = pop(); bool isn = v.isNull || (v.v.kind == Value::Str && v.v.s.empty()); pushb(in.op == IS_NULL ? isn : !isn); break; }
            case JUMP: ip = (size_t)in.a - 1; break;
            case JUMP_IF_FALSE: { const Slot& c = st.back(); if (!c.isNull && c.v.kind == Value::Bool && !c.v.b) ip = (size_t)in.a - 1; break; }   // peeks (:136-142)
            case JUMP_IF_TRUE: { const Slot& c = st.back(); if (!c.isNull && c.v.kind == Value::Bool && c.v.b) ip = (size_t)in.a - 1; break; }
            case HALT: ip = f.code.size(); break;
        }
    }
    if (st.empty()) return false;
    Slot r = st.back();
    return asbool(r);
}

// FacetBuilder.BuildFacetForField + the ordering of BuildFacets (:36-50, :58-105)
inline std::vector<std::pair<std::string, int>> facet(const std::vector<const Fields*>& rows, const std::string& field, int maxFacets = 100) {
    std::map<std::string, int> counts;
    for (const Fields* d : rows) { auto it = d->find(field); if (it == d->end() || it->second.kind == Value::Null) continue; std::string v = to_string(it->second); if (!v.empty()) counts[v]++; }
    std::vector<std::pair<std::string, int>> out(counts.begin(), counts.end());
    std::stable_sort(out.begin(), out.end(), [](auto& a, auto& b) { if (a.second != b.second) return a.second > b.second; int c = cmp_oic(a.first, b.first); if (c) return c < 0; return a.first < b.first; });
    if ((int)out.size() > maxFacets) out.resize(maxFacets);
    return out;
}

}} // namespace
