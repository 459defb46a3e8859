// Micro-benchmark: VALU / SALU / SMEM issue behaviour on gfx950 (inline asm so that the compiler
// cannot re-pack or delete anything).  Prints cycles per loop iteration per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define V4 "v_fma_f32 %0, %0, %5, %6\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %5, %6\n v_fma_f32 %3, %3, %5, %6\n"
#define S4 "s_add_u32 %4, %4, 7\n s_xor_b32 %4, %4, 0x55\n s_lshl_b32 %4, %4, 1\n s_lshr_b32 %4, %4, 1\n"
#define P2 "v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n"

// 16 plain VALU per iteration
__global__ __launch_bounds__(256) void k_v16(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    for (int i = 0; i < ITER; i++)
        asm volatile(V4 V4 V4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
// 16 VALU + 12 SALU per iteration (interleaved)
__global__ __launch_bounds__(256) void k_v16s12(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    for (int i = 0; i < ITER; i++)
        asm volatile(V4 S4 V4 S4 V4 S4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)s;
}
// 12 SALU only
__global__ __launch_bounds__(256) void k_s12(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    for (int i = 0; i < ITER; i++)
        asm volatile(S4 S4 S4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
    out[blockIdx.x * 256 + threadIdx.x] = x0 + (float)s;
}
// 8 packed fma
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk8(float *out, float a, float b) {
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = {2.f, 3.f}, av = {a, a}, bv = {b, b};
    for (int i = 0; i < ITER; i++)
        asm volatile(P2 P2 P2 P2 : "+v"(x0), "+v"(x1) : "v"(av), "v"(bv));
    out[blockIdx.x * 256 + threadIdx.x] = x0.x + x0.y + x1.x + x1.y;
}
// dependent scalar-load chain (pointer chase through a ring of 32-byte records) + 16 VALU + 12 SALU
__global__ __launch_bounds__(256) void k_chase(float *out, const int *__restrict__ ring, float a, float b, int mask) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    int c = (blockIdx.x * 4 + (threadIdx.x >> 6)) & mask;
    for (int i = 0; i < ITER; i++) {
        c = __builtin_amdgcn_readfirstlane(c);
        c = ring[c * 8];  // scalar load, next index
        asm volatile(V4 S4 V4 S4 V4 S4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)s + c;
}
// same chain, but the record is fetched through the VECTOR memory path (every lane loads the same
// address) by all waves (mode 1) or by the odd waves only (mode 2: half scalar, half vector)
__global__ __launch_bounds__(256) void k_chase_mix(float *out, const int *__restrict__ ring, float a, float b, int mask,
                                                   int mode) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    const int wave = threadIdx.x >> 6;
    int c = (blockIdx.x * 4 + wave) & mask;
    const bool vec = mode == 1 || (mode == 2 && (wave & 1));
    if (vec) {
        int cv = c + (threadIdx.x & 0);  // keep it in a VGPR: vector loads
        for (int i = 0; i < ITER; i++) {
            cv = ring[cv * 8];
            asm volatile(V4 S4 V4 S4 V4 S4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s), "+v"(cv) : "v"(a), "v"(b) : "scc");
        }
        c = cv;
    } else {
        for (int i = 0; i < ITER; i++) {
            c = __builtin_amdgcn_readfirstlane(c);
            c = ring[c * 8];
            asm volatile(V4 S4 V4 S4 V4 S4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)s + c;
}
// two / four independent scalar chains per wave (same VALU + SALU work per iteration as k_chase)
template <int NC>
__global__ __launch_bounds__(256) void k_chase_n(float *out, const int *__restrict__ ring, float a, float b, int mask) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; unsigned s = 1;
    int c[NC];
    for (int k = 0; k < NC; k++) c[k] = (blockIdx.x * 4 + (threadIdx.x >> 6) + k * 7919) & mask;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < NC; k++) { c[k] = __builtin_amdgcn_readfirstlane(c[k]); c[k] = ring[c[k] * 8]; }
        asm volatile(V4 S4 V4 S4 V4 S4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s) : "v"(a), "v"(b) : "scc");
    }
    int sum = 0;
    for (int k = 0; k < NC; k++) sum += c[k];
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)s + sum;
}
template <typename F>
float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float *out; HC(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    // ring: random permutation cycle over nrec records of 8 ints
    for (int logn = 14; logn <= 22; logn += 4) {
        int nrec = 1 << logn; int *h = (int *)malloc((size_t)nrec * 32); unsigned r = 12345;
        int *perm = (int *)malloc(nrec * 4); for (int i = 0; i < nrec; i++) perm[i] = i;
        for (int i = nrec - 1; i > 0; i--) { r = r * 1664525u + 1013904223u; int j = r % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        for (int i = 0; i < nrec; i++) h[(size_t)perm[i] * 8] = perm[(i + 1) % nrec];
        int *ring; HC(hipMalloc(&ring, (size_t)nrec * 32)); HC(hipMemcpy(ring, h, (size_t)nrec * 32, hipMemcpyHostToDevice));
        for (int wps = 2; wps <= 8; wps *= 2) {
            float t = timeit([&] { k_chase<<<256 * wps, 256>>>(out, ring, 1.0001f, 0.5f, nrec - 1); });
            printf("chase ring %5.1f MB  waves/SIMD %d: %.3f ms  %.1f cyc/iter/SIMD (%.0f cyc/iter/wave) @2.4GHz\n", nrec * 32 / 1048576.0, wps, t,
                   t * 1e-3 * 2.4e9 / ((double)ITER * wps), t * 1e-3 * 2.4e9 / ITER);
        }
        {
            float t2 = timeit([&] { k_chase_n<2><<<256 * 8, 256>>>(out, ring, 1.0001f, 0.5f, nrec - 1); });
            float t4 = timeit([&] { k_chase_n<4><<<256 * 8, 256>>>(out, ring, 1.0001f, 0.5f, nrec - 1); });
            printf("chase ring %5.1f MB  8 waves/SIMD  2 chains/wave: %.1f cyc/iter/SIMD   4 chains/wave: %.1f (same VALU/SALU work per iteration)\n",
                   nrec * 32 / 1048576.0, t2 * 1e-3 * 2.4e9 / ((double)ITER * 8), t4 * 1e-3 * 2.4e9 / ((double)ITER * 8));
        }
        for (int mode = 0; mode <= 2; mode++) {
            float t = timeit([&] { k_chase_mix<<<256 * 8, 256>>>(out, ring, 1.0001f, 0.5f, nrec - 1, mode); });
            printf("chase ring %5.1f MB  8 waves/SIMD  %s: %.1f cyc/iter/SIMD\n", nrec * 32 / 1048576.0,
                   mode == 0 ? "scalar loads" : (mode == 1 ? "vector loads" : "half scalar, half vector"),
                   t * 1e-3 * 2.4e9 / ((double)ITER * 8));
        }
        (void)hipFree(ring); free(h); free(perm);
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        int grid = 256 * wps; double it = (double)ITER * wps;
        float t1 = timeit([&] { k_v16<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        float t2 = timeit([&] { k_v16s12<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        float t3 = timeit([&] { k_s12<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        float t4 = timeit([&] { k_pk8<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        printf("waves/SIMD %d: cycles/iter/SIMD @2.4GHz: 16 VALU %.1f | 16 VALU + 12 SALU %.1f | 12 SALU %.1f | 8 pk_fma %.1f\n", wps,
               t1 * 1e-3 * 2.4e9 / it, t2 * 1e-3 * 2.4e9 / it, t3 * 1e-3 * 2.4e9 / it, t4 * 1e-3 * 2.4e9 / it);
    }
    return 0;
}
