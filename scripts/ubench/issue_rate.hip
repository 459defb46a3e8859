// Micro-benchmark: VALU / SALU issue rates on gfx950 with 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fma(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < ITER; i++) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ __launch_bounds__(256) void k_pkfma(float *out, float a, float b) {
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float2v av = {a, a}, bv = {b, b};
    for (int i = 0; i < ITER; i++) {
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
    }
    float2v s = x0 + x1 + x2 + x3;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
// 8 VALU + 8 SALU per iteration (scalar work on a uniform value)
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b, unsigned seed) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned s = seed;
    for (int i = 0; i < ITER; i++) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        asm volatile("s_add_u32 %0, %0, 7\n s_xor_b32 %0, %0, 0x55\n s_lshl_b32 %0, %0, 1\n s_add_u32 %0, %0, 3\n"
                     "s_xor_b32 %0, %0, 0x33\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, 5\n s_xor_b32 %0, %0, 0x0f\n"
                     : "+s"(s));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)s;
}
// 8 SALU only
__global__ __launch_bounds__(256) void k_salu(float *out, unsigned seed) {
    unsigned s = seed;
    for (int i = 0; i < ITER; i++) {
        asm volatile("s_add_u32 %0, %0, 7\n s_xor_b32 %0, %0, 0x55\n s_lshl_b32 %0, %0, 1\n s_add_u32 %0, %0, 3\n"
                     "s_xor_b32 %0, %0, 0x33\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, 5\n s_xor_b32 %0, %0, 0x0f\n"
                     : "+s"(s));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
// rsq + cmp + cndmask mix like the walk
__global__ __launch_bounds__(256) void k_rsq(float *out, float a) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < ITER; i++) {
        x0 = __builtin_amdgcn_rsqf(x0) + a; x1 = __builtin_amdgcn_rsqf(x1) + a;
        x2 = __builtin_amdgcn_rsqf(x2) + a; x3 = __builtin_amdgcn_rsqf(x3) + a;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <typename F>
float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD = blocks per CU (256-thread blocks = 1 wave per SIMD)
        int grid = 256 * wps;
        double waves_per_simd = wps;
        double simd_instr = (double)ITER * 8 * waves_per_simd;  // per SIMD
        float t1 = timeit([&] { k_fma<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        float t2 = timeit([&] { k_pkfma<<<grid, 256>>>(out, 1.0001f, 0.5f); });
        float t3 = timeit([&] { k_mix<<<grid, 256>>>(out, 1.0001f, 0.5f, 1u); });
        float t4 = timeit([&] { k_salu<<<grid, 256>>>(out, 1u); });
        float t5 = timeit([&] { k_rsq<<<grid, 256>>>(out, 0.5f); });
        double clk = 2.4e9;
        printf("waves/SIMD %d: fma %.3f ms (%.2f cyc/instr/SIMD @2.4GHz)  pk_fma %.3f ms (%.2f)  mix8v+8s %.3f ms (%.2f per valu)  salu-only %.3f ms (%.2f cyc/sinstr/SIMD)  rsq+add %.3f ms (%.2f cyc per pair)\n",
               wps, t1, t1 * 1e-3 * clk / simd_instr, t2, t2 * 1e-3 * clk / simd_instr, t3, t3 * 1e-3 * clk / simd_instr,
               t4, t4 * 1e-3 * clk / simd_instr, t5, t5 * 1e-3 * clk / ((double)ITER * 4 * waves_per_simd));
    }
    return 0;
}
