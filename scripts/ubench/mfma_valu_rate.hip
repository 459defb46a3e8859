// How much vector work fits beside an fp32 MFMA stream on one SIMD?  ("price the primitive first")
// Per loop trip: M independent MFMAs (own accumulators) interleaved with V independent v_fma_f32 per MFMA, W waves per
// SIMD (blocks of 256 threads = one wave per SIMD each, W blocks per CU).  Reports shader cycles per trip per SIMD
// (all resident waves together), i.e. what the SIMD spends on M MFMAs + M*V FMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_rate mfma_valu_rate.hip && ./mfma_valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTrips = 4096;

// KIND 0: v_mfma_f32_4x4x1_16b_f32 (8 cycles by the ISA tables), 1: v_mfma_f32_16x16x4_f32 (32 cycles), 2: no MFMA
template <int KIND, int V, bool RSQ>
__global__ __launch_bounds__(256) void k_rate(float *out, float seed) {
    f4 acc[8];
    float x[8], y[8];
    for (int i = 0; i < 8; i++) { acc[i] = f4{seed, seed, seed, seed}; x[i] = seed + i; y[i] = seed * 0.5f + i; }
    const float a = seed * 1.0001f, b = seed * 0.9999f;
    for (int t = 0; t < kTrips; t++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < V; k++) {
                if (RSQ && k == 0) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(x[(i + k) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[(i + k) & 7]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + y[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int V, bool RSQ>
static int run(float *out, const char *name) {
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (int W : {1, 2, 4, 8}) {
        const int blocks = 256 * W;
        k_rate<KIND, V, RSQ><<<blocks, 256>>>(out, 1.0f);
        HC(hipDeviceSynchronize());
        HC(hipEventRecord(e0));
        k_rate<KIND, V, RSQ><<<blocks, 256>>>(out, 1.0f);
        HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        // one SIMD runs W waves, each kTrips trips of 8 (MFMA + V VALU) groups
        const double cyc_per_group = ms * 1e-3 * 2.4e9 / ((double)kTrips * 8 * W);
        printf("%-22s V=%d%s waves/SIMD=%d: %7.2f cycles per (MFMA + %d VALU) per SIMD\n", name, V, RSQ ? " (1 rsq)" : "        ", W, cyc_per_group, V);
    }
    return 0;
}

int main() {
    float *out;
    HC(hipMalloc(&out, 64));
    if (run<2, 1, false>(out, "no MFMA") || run<2, 4, false>(out, "no MFMA") || run<2, 4, true>(out, "no MFMA") ||
        run<0, 0, false>(out, "4x4x1_16b") || run<0, 1, false>(out, "4x4x1_16b") || run<0, 2, false>(out, "4x4x1_16b") ||
        run<0, 3, false>(out, "4x4x1_16b") || run<0, 4, false>(out, "4x4x1_16b") || run<0, 3, true>(out, "4x4x1_16b") ||
        run<1, 0, false>(out, "16x16x4") || run<1, 4, false>(out, "16x16x4") || run<1, 8, false>(out, "16x16x4") ||
        run<1, 12, false>(out, "16x16x4") || run<1, 16, false>(out, "16x16x4") || run<1, 12, true>(out, "16x16x4"))
        return 1;
    return 0;
}
