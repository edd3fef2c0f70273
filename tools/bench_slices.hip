// What does the MI355X memory system deliver for k_accumulate's access pattern?  One wave per workgroup (as k_accumulate), 10 KB of LDS per wave
// (16 waves per CU), every "visit" = one 16-byte load per lane of a random 16-byte-aligned 1 KB slice of a 2 GB array (the posting array of the
// 10 M-doc index), DEPTH loads in flight per wave.  Variants: bare loads (XOR-reduced), + the 4 ds_write_b8 scatter per lane, + the candidate probe
// (ds_read_u8 + ds_write_b8 + ballot).  Prints GB/s per variant: the ceiling the Stage-1 kernel can approach without changing its bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_slices.hip -o gpurun_out/bench_slices && gpurun_out/bench_slices
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4i_t* gint4p;
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
#define LDS8(off) (*(lds_u8_t*)(uint32_t)(off))

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int DEPTH, int MODE>
__global__ __launch_bounds__(64, 4) void k_slices(const int* __restrict__ data, uint64_t nVec, int visits, int sliceVec, uint32_t* __restrict__ out, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const gint4p p = (gint4p)(uintptr_t)data;
    if (MODE >= 1) for (int i = lane * 16; i < 8192 + 128; i += 64 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
    const uint32_t cAddr = mix(b * 77u + lane) & 8191u;
    uint32_t acc = 0;
    v4i_t c[DEPTH];
    auto slice_of = [&](int i) -> uint64_t { const uint64_t h = ((uint64_t)mix(b * 2654435761u + (uint32_t)i + seed) << 8) ^ mix((uint32_t)i * 40503u + b); return h & (nVec - 1); };      // nVec: a power of two (+ 64 vectors of slack behind it)
#pragma unroll
    for (int d = 0; d < DEPTH; d++) c[d] = p[slice_of(d) + (lane < sliceVec ? lane : sliceVec - 1)];
    for (int i = 0; i < visits; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            v4i_t v = c[d];
            c[d] = p[slice_of(i + DEPTH + d) + (lane < sliceVec ? lane : sliceVec - 1)];
            if (MODE == 0) acc ^= (uint32_t)(v.x ^ v.y ^ v.z ^ v.w);
            else {
                const uint32_t base = (uint32_t)(i + d) * 8192u;
                const uint32_t R = 0xFFFFFF00u + lane;             // (the data is random: the clamp never bites, the cells are random)
                const uint32_t l0 = min(((uint32_t)v.x >> 8) - base, R), l1 = min(((uint32_t)v.y >> 8) - base, R), l2 = min(((uint32_t)v.z >> 8) - base, R), l3 = min(((uint32_t)v.w >> 8) - base, R);
                LDS8(l0 & 8191u) = (uint8_t)v.x; LDS8(l1 & 8191u) = (uint8_t)v.y; LDS8(l2 & 8191u) = (uint8_t)v.z; LDS8(l3 & 8191u) = (uint8_t)v.w;
                if (MODE >= 2) {
                    const uint32_t t = LDS8(cAddr); LDS8(cAddr) = 0;
                    if (__ballot(t != 0)) acc += t;
                }
            }
        }
    }
    if (MODE >= 1) acc ^= LDS8(cAddr);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= (uint32_t)c[d].x;
    if (acc == 0x12345678u) out[b & 1023] = acc;
}

template <int DEPTH, int MODE> static void run(const int* d, uint64_t nVec, uint32_t* out, int blocks, int visits, int sliceVec, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 10152;
    k_slices<DEPTH, MODE><<<blocks, 64, lds>>>(d, nVec, visits, sliceVec, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_slices<DEPTH, MODE><<<blocks, 64, lds>>>(d, nVec, visits, sliceVec, out, 7u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * (visits + DEPTH) * sliceVec * 16.0;
    printf("%-46s depth %d slice %4d B: %7.3f ms  %7.1f GB/s  %6.1f M visits/s\n", what, DEPTH, sliceVec * 16, ms, bytes / ms * 1e-6, (double)blocks * visits / ms * 1e-3);
}

int main() {
    const uint64_t bytes = 2048ull << 20; const uint64_t nVec = bytes / 16;
    int* d; hipMalloc(&d, bytes + 4096);
    std::vector<uint32_t> h(1 << 20); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    for (uint64_t o = 0; o < bytes; o += h.size() * 4) hipMemcpy((char*)d + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    uint32_t* out; hipMalloc(&out, 4096);
    const int blocks = 306000, visits = 128;
    for (int sv : {64, 52, 32}) {
        run<2, 0>(d, nVec, out, blocks, visits, sv, "loads only");
        run<4, 0>(d, nVec, out, blocks, visits, sv, "loads only");
        run<8, 0>(d, nVec, out, blocks, visits, sv, "loads only");
    }
    run<2, 1>(d, nVec, out, blocks, visits, 52, "loads + 4 ds_write_b8 scatter");
    run<4, 1>(d, nVec, out, blocks, visits, 52, "loads + 4 ds_write_b8 scatter");
    run<2, 2>(d, nVec, out, blocks, visits, 52, "loads + scatter + probe (read, restore, ballot)");
    run<4, 2>(d, nVec, out, blocks, visits, 52, "loads + scatter + probe (read, restore, ballot)");
    run<8, 2>(d, nVec, out, blocks, visits, 52, "loads + scatter + probe (read, restore, ballot)");
    return 0;
}
