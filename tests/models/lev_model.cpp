// Host check of the distance-1 fast path of the Stage-2 matchers: infidex_amd/csrc/lev.hip.inc compiled for the host, s2_dam1(a, b) against s2_damerau(a, b, 1)
// (the banded dynamic programme + the reference's transposition rule) clipped to 2, on every pair of strings over {a, b, c} up to length 6 and on random strings
// with planted edits (substitutions, insertions, deletions, adjacent swaps, at the ends and in the middle).  Test infrastructure (tests/test_lev_model.py).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
using std::min; using std::max;
#define S2_FN static inline
#define S2_FNX static
#include "../../infidex_amd/csrc/lev.hip.inc"
typedef std::basic_string<uint16_t> ustr;
static int check(const ustr& a, const ustr& b, long long& n) {
    S2Str A{a.data(), (int)a.size()}, B{b.data(), (int)b.size()};
    bool ov = false;
    const int ref = std::min(s2_damerau(A, B, 1, &ov), 2), got = s2_dam1(A, B);
    n++;
    if (ref != got) { fprintf(stdout, "MISMATCH ref %d got %d: a=", ref, got); for (auto c : a) fputc((int)c, stdout); fputs(" b=", stdout); for (auto c : b) fputc((int)c, stdout); fputc('\n', stdout); return 1; }
    return 0;
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200000;
    long long n = 0; int bad = 0;
    std::vector<ustr> all; all.push_back(ustr());
    for (size_t lo = 0, len = 1; len <= 6; len++) { const size_t hi = all.size(); for (size_t i = lo; i < hi; i++) for (uint16_t c = 'a'; c <= 'c'; c++) { ustr s = all[i]; s.push_back(c); all.push_back(s); } lo = hi; }
    for (auto& a : all) for (auto& b : all) { if (std::abs((int)a.size() - (int)b.size()) > 2) continue; bad += check(a, b, n); if (bad > 5) return 1; }
    std::mt19937 rng(12345);
    for (int r = 0; r < rounds; r++) {
        const int len = 1 + rng() % 14, alpha = 2 + rng() % 6;
        ustr a; for (int i = 0; i < len; i++) a.push_back((uint16_t)('a' + rng() % alpha));
        ustr b = a;
        const int edits = rng() % 4;
        for (int e = 0; e < edits && !b.empty(); e++) {
            const int kind = rng() % 4; const size_t pos = rng() % b.size();
            if (kind == 0) b[pos] = (uint16_t)('a' + rng() % alpha);
            else if (kind == 1) b.insert(b.begin() + pos, (uint16_t)('a' + rng() % alpha));
            else if (kind == 2) b.erase(b.begin() + pos);
            else if (pos + 1 < b.size()) std::swap(b[pos], b[pos + 1]);
        }
        bad += check(a, b, n); bad += check(b, a, n);
        if (bad > 5) return 1;
    }
    if (bad) return 1;
    printf("OK %lld pairs\n", n);
    return 0;
}
