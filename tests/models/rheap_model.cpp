// Host model of RHeap (infidex_amd/csrc/exact3.hip.inc): the wave's lanes are arrays, v_readlane / v_writelane are element accesses, everything
// else is the device code line by line.  Checked against the plain array-backed 4-ary PriorityQueue (BCL rules: MoveDown picks the first minimal
// child with `<`, stops on `<=`; EnqueueDequeue only for a priority above the root's) on random streams with many equal priorities: the final
// ARRAY (node by node) must be equal.  Test infrastructure (tests/test_rheap_model.py); nothing in the product links it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define RH_DOCM 0x3FFFFFFFu
#define RH_KM 0xC0000000u
static bool rh_gt(uint32_t pa, uint32_t la, uint32_t pb, uint32_t lb) { return pa > pb || (pa == pb && la > lb); }
struct Plain {                       // nodes {priority bits, doc}
    std::vector<uint32_t> p; std::vector<uint32_t> d; int n = 0;
    void move_up(uint32_t np, uint32_t nd, int idx) { while (idx > 0) { int parent = (idx - 1) >> 2; if (np < p[parent]) { p[idx] = p[parent]; d[idx] = d[parent]; idx = parent; } else break; } p[idx] = np; d[idx] = nd; }
    void move_down(uint32_t np, uint32_t nd, int idx) {
        int i;
        while ((i = (idx << 2) + 1) < n) {
            uint32_t mp = p[i]; int mi = i;
            for (int k = 1; k < 4 && i + k < n; k++) if (p[i + k] < mp) { mp = p[i + k]; mi = i + k; }
            if (np <= mp) break;
            p[idx] = mp; d[idx] = d[mi]; idx = mi;
        }
        p[idx] = np; d[idx] = nd;
    }
    void enqueue(uint32_t nd, uint32_t np) { p.push_back(0); d.push_back(0); n++; move_up(np, nd, n - 1); }
    void enqueue_dequeue(uint32_t nd, uint32_t np) { if (n > 0 && np > p[0]) move_down(np, nd, 0); }
};
struct Model {
    uint32_t p0[64][4], l0[64][4], p1[64][4], l1[64][4], np0[64], nl0[64], np1[64], nl1[64], rootP, rootD;
    static int group0(int lane) { return lane < 21 ? lane : lane + 64; }
    static int group1(int lane) { return lane + 21; }
    static void cswap(uint32_t& pa, uint32_t& la, uint32_t& pb, uint32_t& lb) { if (rh_gt(pa, la, pb, lb)) { std::swap(pa, pb); std::swap(la, lb); } }
    void load(const Plain& H) {
        const int n = H.n;
        for (int lane = 0; lane < 64; lane++) {
            const int g0 = group0(lane), g1 = group1(lane);
            for (int k = 0; k < 4; k++) {
                const int a = 4 * g0 + 1 + k, b = 4 * g1 + 1 + k;
                p0[lane][k] = a < n ? H.p[a] : 0xFFFFFFFFu; l0[lane][k] = ((uint32_t)k << 30) | (a < n ? H.d[a] & RH_DOCM : 0u);
                p1[lane][k] = b < n ? H.p[b] : 0xFFFFFFFFu; l1[lane][k] = ((uint32_t)k << 30) | (b < n ? H.d[b] & RH_DOCM : 0u);
            }
            auto s4 = [&](uint32_t (&p)[4], uint32_t (&l)[4]) { cswap(p[0], l[0], p[1], l[1]); cswap(p[2], l[2], p[3], l[3]); cswap(p[0], l[0], p[2], l[2]); cswap(p[1], l[1], p[3], l[3]); cswap(p[1], l[1], p[2], l[2]); };
            s4(p0[lane], l0[lane]); s4(p1[lane], l1[lane]);
            np0[lane] = p0[lane][0]; nl0[lane] = l0[lane][0]; np1[lane] = p1[lane][0]; nl1[lane] = l1[lane][0];
        }
        rootP = H.p[0]; rootD = H.d[0];
    }
    static void settle(uint32_t (&p)[4], uint32_t (&l)[4], uint32_t& np, uint32_t& nl) {
        const uint32_t xp = np, xl = (l[0] & RH_KM) | (nl & RH_DOCM);
        const bool c1 = (uint32_t)(xp + (xl > l[1] ? 1u : 0u)) > p[1], c2 = (uint32_t)(xp + (xl > l[2] ? 1u : 0u)) > p[2], c3 = (uint32_t)(xp + (xl > l[3] ? 1u : 0u)) > p[3];
        const uint32_t a0 = c1 ? p[1] : xp, b0 = c1 ? l[1] : xl;
        const uint32_t a1 = c2 ? p[2] : (c1 ? xp : p[1]), b1 = c2 ? l[2] : (c1 ? xl : l[1]);
        const uint32_t a2 = c3 ? p[3] : (c2 ? xp : p[2]), b2 = c3 ? l[3] : (c2 ? xl : l[2]);
        const uint32_t a3 = c3 ? xp : p[3], b3 = c3 ? xl : l[3];
        p[0] = a0; l[0] = b0; p[1] = a1; l[1] = b1; p[2] = a2; l[2] = b2; p[3] = a3; l[3] = b3;
        np = a0; nl = b0;
    }
    static int child_group(int c, uint32_t ml) { return 4 * c + (int)(ml >> 30) + 1; }
    void replace_root(uint32_t np, uint32_t nd) {
        int hl = -1; bool in1 = false;
        do {
            uint32_t mp = p0[0][0];
            if (np <= mp) { rootP = np; rootD = nd; break; }
            uint32_t ml = l0[0][0];
            rootP = mp; rootD = ml & RH_DOCM;
            hl = 0; int c = child_group(0, ml);
            mp = p0[c][0]; if (np <= mp) break;
            ml = l0[c][0]; np0[hl] = mp; nl0[hl] = ml;
            hl = c; c = child_group(c, ml);
            mp = p0[c][0]; if (np <= mp) break;
            ml = l0[c][0]; np0[hl] = mp; nl0[hl] = ml;
            hl = c; c = child_group(c, ml);
            mp = p1[c - 21][0]; if (np <= mp) break;
            ml = l1[c - 21][0]; np0[hl] = mp; nl0[hl] = ml;
            hl = c - 21; c = child_group(c, ml); in1 = true;
            if (c <= 127) {
                mp = p0[c - 64][0];
                if (np > mp) { ml = l0[c - 64][0]; np1[hl] = mp; nl1[hl] = ml; hl = c - 64; in1 = false; }
            }
        } while (0);
        if (hl >= 0) { if (in1) { np1[hl] = np; nl1[hl] = nd; } else { np0[hl] = np; nl0[hl] = nd; } }
        for (int lane = 0; lane < 64; lane++) { settle(p0[lane], l0[lane], np0[lane], nl0[lane]); settle(p1[lane], l1[lane], np1[lane], nl1[lane]); }
    }
    void store(std::vector<uint32_t>& P, std::vector<uint32_t>& D, int n) const {
        P.assign(n, 0); D.assign(n, 0); P[0] = rootP; D[0] = rootD;
        for (int lane = 0; lane < 64; lane++) for (int i = 0; i < 4; i++) {
            const int a = 4 * group0(lane) + 1 + (int)(l0[lane][i] >> 30), b = 4 * group1(lane) + 1 + (int)(l1[lane][i] >> 30);
            if (a < n) { P[a] = p0[lane][i]; D[a] = l0[lane][i] & RH_DOCM; }
            if (b < n) { P[b] = p1[lane][i]; D[b] = l1[lane][i] & RH_DOCM; }
        }
    }
};
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937 rng(12345);
    long ops = 0;
    for (int r = 0; r < rounds; r++) {
        const int depths[] = {1, 2, 5, 6, 20, 21, 22, 85, 86, 100, 340, 341, 342, 499, 500, 501, 509, 510, 511, 512};
        const int depth = depths[r % 20];
        const int levels = 1 + (int)(rng() % (r % 3 == 0 ? 4 : 40));          // few distinct priorities -> plateaus
        const int n = depth + (int)(rng() % 6000);
        Plain H; Model M; bool inReg = false;
        for (int i = 0; i < n; i++) {
            // rising trend + ties: priorities are bit patterns of positive floats
            float f = 1.0f + (float)(rng() % levels) * 0.125f + (rng() % 4 == 0 ? (float)i * 1e-3f : 0.f);
            uint32_t pb; memcpy(&pb, &f, 4); const uint32_t d = (uint32_t)(rng() & RH_DOCM);
            if (H.n < depth) { H.enqueue(d, pb); if (H.n == depth) { M.load(H); inReg = true; } }
            else { const uint32_t th = H.p[0]; if (pb > th) { H.enqueue_dequeue(d, pb); M.replace_root(pb, d); ops++; if (M.rootP != H.p[0]) { printf("FAIL threshold round %d op %d\n", r, i); return 1; } } }
        }
        if (inReg) {
            std::vector<uint32_t> P, D; M.store(P, D, depth);
            for (int i = 0; i < depth; i++) if (P[i] != H.p[i] || D[i] != H.d[i]) { printf("FAIL round %d depth %d node %d: model (%08x,%u) heap (%08x,%u)\n", r, depth, i, P[i], D[i], H.p[i], H.d[i]); return 1; }
        }
    }
    printf("OK %d rounds, %ld replace operations\n", rounds, ops);
    return 0;
}
