// Micro-benchmark behind the walk's record fetch: a wave follows K independent chains of 24-byte records (the walk: one
// cursor, the pair loop: two), 8 waves per SIMD, every CU busy.  Is the rate set by the SCALAR memory path (s_load into
// SGPRs, what the walk uses: operands are wave-uniform) - and would the VECTOR path (every lane loads the same address:
// one request after coalescing, in-order return, far more requests in flight per CU) move more records?
// Per variant: hops per microsecond over the whole chip and shader cycles per hop per wave, for K = 1, 2, 4 chains and
// for F = 0 / 16 dependent-free VALU instructions per hop (the walk's visit has 16).  The records form one random
// cycle over `span` records: every hop misses the CU's scalar cache / L1 and - for spans below the L2 size - hits L2,
// like the walk (TCC hit rate 90 %).
//   hipcc --offload-arch=gfx950 -O3 -o chain_loads chain_loads.hip && ./chain_loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct alignas(8) Rec { float a, b, c, d, e; unsigned next; };  // next: byte offset of the next record
constexpr int kHops = 2048;

template <int K, int F, bool VEC>
__global__ __launch_bounds__(256) void k_chain(const Rec *__restrict__ recs, unsigned span, float *out, float p) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned off[K];
#pragma unroll
    for (int k = 0; k < K; k++) off[k] = __builtin_amdgcn_readfirstlane(((unsigned)(wave * 977 + k * 7919) * 2654435761u % span) * 24u);
    float acc = (float)(threadIdx.x & 63) * 1e-3f, acc2 = 0.f;
    for (int h = 0; h < kHops; h++) {
        Rec r[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (VEC) {
                unsigned ov = off[k];
                asm volatile("" : "+v"(ov));  // the compiler no longer knows the address is uniform: a vector load
                r[k] = *reinterpret_cast<const Rec *>(reinterpret_cast<const char *>(recs) + ov);
            } else {
                r[k] = *reinterpret_cast<const Rec *>(reinterpret_cast<const char *>(recs) + off[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            float x = r[k].a - p, y = r[k].b - p;
#pragma unroll
            for (int f = 0; f < F / 2; f++) { x = fmaf(x, r[k].c, acc); y = fmaf(y, r[k].d, acc2); }
            acc += x; acc2 += y + r[k].e;
            off[k] = __builtin_amdgcn_readfirstlane(r[k].next);
        }
    }
    if (acc + acc2 == 12345.678f) out[0] = acc;
}

template <int K, int F, bool VEC>
static int run(const Rec *d, unsigned span, float *out, int blocks, const char *name) {
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    k_chain<K, F, VEC><<<blocks, 256>>>(d, span, out, 0.5f);
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0));
    for (int i = 0; i < 3; i++) k_chain<K, F, VEC><<<blocks, 256>>>(d, span, out, 0.5f);
    HC(hipEventRecord(e1));
    HC(hipEventSynchronize(e1));
    float ms = 0.f;
    HC(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3.f;
    const double hops = (double)blocks * 4 * K * kHops;
    // shader cycles per hop per wave at 2.4 GHz with 8 waves per SIMD: (ms * 2.4e6) / (hops of one wave = K * kHops) when
    // every SIMD holds exactly its 8 waves (blocks = 256 CUs x 8)
    printf("%-8s K=%d F=%2d span=%8u: %8.3f ms  %9.1f hops/us  %7.1f cycles per hop per wave  (%6.1f per chain step)\n", name, K, F, span, ms,
           hops / (ms * 1e3), ms * 2.4e6 / (K * kHops), ms * 2.4e6 / kHops);
    return 0;
}

int main() {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float *out;
    HC(hipMalloc(&out, 64));
    for (unsigned span : {65536u, 1500000u}) {
        std::vector<unsigned> order(span);
        std::iota(order.begin(), order.end(), 0u);
        std::mt19937 rng(7);
        std::shuffle(order.begin(), order.end(), rng);
        std::vector<Rec> h(span);
        for (unsigned i = 0; i < span; i++) {
            Rec &r = h[order[i]];
            r.a = 0.1f * (i % 7); r.b = 0.2f; r.c = 0.999f; r.d = 1.001f; r.e = 0.f;
            r.next = order[(i + 1) % span] * 24u;
        }
        Rec *d;
        HC(hipMalloc(&d, (size_t)span * sizeof(Rec)));
        HC(hipMemcpy(d, h.data(), (size_t)span * sizeof(Rec), hipMemcpyHostToDevice));
        if (run<1, 0, false>(d, span, out, blocks, "scalar") || run<2, 0, false>(d, span, out, blocks, "scalar") ||
            run<4, 0, false>(d, span, out, blocks, "scalar") || run<1, 16, false>(d, span, out, blocks, "scalar") ||
            run<2, 16, false>(d, span, out, blocks, "scalar") || run<4, 16, false>(d, span, out, blocks, "scalar") ||
            run<1, 0, true>(d, span, out, blocks, "vector") || run<2, 0, true>(d, span, out, blocks, "vector") ||
            run<4, 0, true>(d, span, out, blocks, "vector") || run<1, 16, true>(d, span, out, blocks, "vector") ||
            run<2, 16, true>(d, span, out, blocks, "vector") || run<4, 16, true>(d, span, out, blocks, "vector"))
            return 1;
        HC(hipFree(d));
    }
    return 0;
}
