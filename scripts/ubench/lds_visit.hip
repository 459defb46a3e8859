// Micro-benchmark for the batched-frontier walk's inner loop: per visit 2 x ds_read_b128 (node) +
// 1 x ds_read_b64 (lane mask) at a wave-uniform LDS address, then the walk's per-lane maths.
// LDS per wave is padded to model the stack footprint (occupancy).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
struct alignas(16) Rec { float cx, cy, cz, gm, s2t; int child; int nchild; int pad; unsigned long long mask; unsigned long long pad2; };

template <int kLdsPerWave>
__global__ __launch_bounds__(256) void k_visit(float *out, const Rec *src, float eps2) {
    __shared__ char lds[4][kLdsPerWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Rec *buf = reinterpret_cast<Rec *>(&lds[wave][0]);
    buf[lane] = src[(blockIdx.x * 4 + wave) % 8 * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0);
    float px = lane * 0.37f, py = lane * 0.11f, pz = lane * 0.23f, ax = 0, ay = 0, az = 0;
    unsigned long long omask = 0;
    for (int i = 0; i < ITER; i++) {
        const Rec nd = buf[i & 63];  // wave-uniform LDS address
        const float dx = nd.cx - px, dy = nd.cy - py, dz = nd.cz - pz;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
        const unsigned long long m = __builtin_amdgcn_readfirstlane((unsigned)nd.mask) |
                                     ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(nd.mask >> 32)) << 32);
        const bool geom = __float_as_int(nd.s2t) < __float_as_int(d2);
        const unsigned long long g = __builtin_amdgcn_ballot_w64(geom);
        const unsigned long long take = m & g, open = m & ~g;
        const bool tk = (take >> lane) & 1;
        const float inv = __builtin_amdgcn_rsqf(d2);
        const float f = tk ? nd.gm * inv * inv * inv : 0.f;
        ax = fmaf(dx, f, ax); ay = fmaf(dy, f, ay); az = fmaf(dz, f, az);
        omask ^= open;
    }
    out[blockIdx.x * 256 + threadIdx.x] = ax + ay + az + (float)(omask & 1);
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
    Rec *h = (Rec *)malloc(512 * sizeof(Rec)), *d;
    for (int i = 0; i < 512; i++) { h[i] = Rec{i * 0.5f, i * 0.25f, i * 0.125f, 1.f, (i % 3) * 100.f, 0, 0, 0, 0xF0F0F0F0F0F0F0F0ull ^ (unsigned long long)i * 0x9E3779B97F4A7C15ull, 0}; }
    (void)hipMalloc(&d, 512 * sizeof(Rec)); (void)hipMemcpy(d, h, 512 * sizeof(Rec), hipMemcpyHostToDevice);
    // blocks per CU limited by LDS: 4 waves * kLdsPerWave per block
    {
        float t = timeit([&] { k_visit<4096><<<256 * 8, 256>>>(out, d, 2.25f); });   // 16 KB/block -> 8+ blocks/CU
        printf("lds/wave 4 KB  (up to 8 waves/SIMD): %.3f ms  %.1f cyc/visit/SIMD @2.4GHz\n", t, t * 1e-3 * 2.4e9 / (ITER * 8.0));
    }
    {
        float t = timeit([&] { k_visit<12288><<<256 * 3, 256>>>(out, d, 2.25f); });  // 48 KB/block -> 3 blocks/CU
        printf("lds/wave 12 KB (3 waves/SIMD):       %.3f ms  %.1f cyc/visit/SIMD @2.4GHz\n", t, t * 1e-3 * 2.4e9 / (ITER * 3.0));
    }
    {
        float t = timeit([&] { k_visit<16384><<<256 * 2, 256>>>(out, d, 2.25f); });  // 64 KB/block -> 2 blocks/CU
        printf("lds/wave 16 KB (2 waves/SIMD):       %.3f ms  %.1f cyc/visit/SIMD @2.4GHz\n", t, t * 1e-3 * 2.4e9 / (ITER * 2.0));
    }
    return 0;
}
