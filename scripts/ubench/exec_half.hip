// Micro-benchmark: does a VALU instruction whose EXEC mask leaves one 32-lane half (or all but one lane) idle
// issue faster on gfx950's SIMD-32?  16 dependent-free v_fma per iteration under four EXEC masks.
// Prints cycles per iteration per SIMD with 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define V4 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"

__global__ __launch_bounds__(256) void k_mask(float *out, float a, float b, unsigned long long mask_in, int mixed, unsigned long long *stamps) {
    // mixed: only the odd waves of a workgroup narrow EXEC, the even ones keep all lanes
    const bool keep_all = mixed && !((threadIdx.x >> 6) & 1);
    const unsigned mlo = __builtin_amdgcn_readfirstlane(keep_all ? ~0u : (unsigned)mask_in);
    const unsigned mhi = __builtin_amdgcn_readfirstlane(keep_all ? ~0u : (unsigned)(mask_in >> 32));
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float *o = out + blockIdx.x * 256 + threadIdx.x;
    int it = ITER;
    // everything the compiler computes per lane must exist BEFORE EXEC is narrowed (it does not know about it)
    asm volatile("" : "+v"(o), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(it));
    asm volatile("s_mov_b32 exec_lo, %0\n s_mov_b32 exec_hi, %1" ::"s"(mlo), "s"(mhi));
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < it; i++)
        asm volatile(V4 V4 V4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b64 exec, -1");
    *o = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0 && blockIdx.x == 7) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

int main() {
    float *out;
    unsigned long long *stamps, hst[2];
    HC(hipMalloc(&stamps, 16));
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    HC(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    unsigned long long masks[24];
    char names[24][48];
    int nm = 0;
    auto add = [&](unsigned long long m, const char *n) { masks[nm] = m; snprintf(names[nm], 48, "%s", n); nm++; };
    add(~0ull, "all 64 lanes");
    add(0xffffffffull, "lanes 0-31");
    add(0xffffffff00000000ull, "lanes 32-63");
    add(0x0000ffff0000ffffull, "0-15 + 32-47");
    for (int k : {1, 2, 4, 8, 12, 16, 20, 24, 28}) {
        char nb[48];
        snprintf(nb, 48, "lowest %d lanes", k);
        add((1ull << k) - 1, nb);
    }
    add(0x0000000100000001ull, "lane 0 + lane 32");
    add(0x000000ff000000ffull, "0-7 + 32-39");
    add(0x1111111111111111ull, "every 4th lane (16)");
    add(0x0101010101010101ull, "every 8th lane (8)");
    add(0xff00ull, "lanes 8-15");
    for (int m = 0; m < 2 * nm + 4; m++) {
        int mixed = m >= nm && m < 2 * nm, mm = m % nm, bl = blocks;
        if (m >= 2 * nm) { mm = (m - 2 * nm) & 1 ? 7 : 0; mixed = 0; bl = (m - 2 * nm) < 2 ? 256 : 512; }  // 1 / 2 waves per SIMD
        if (mixed && mm != 0 && mm != 4 && mm != 7 && mm != 8) continue;
        k_mask<<<bl, 256>>>(out, 1.0001f, 0.5f, masks[mm], mixed, stamps);
        HC(hipDeviceSynchronize());
        HC(hipEventRecord(e0));
        k_mask<<<bl, 256>>>(out, 1.0001f, 0.5f, masks[mm], mixed, stamps);
        HC(hipEventRecord(e1));
        HC(hipEventSynchronize(e1));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1));
        // each SIMD runs 8 waves x ITER iterations; assume 2.4 GHz
        HC(hipMemcpy(hst, stamps, 16, hipMemcpyDeviceToHost));
        printf("[wave 0 of block 7: %.0f MHz, %.2f shader cycles per VALU] ", 100.0 * hst[0] / hst[1], (double)hst[0] / ITER / 16);
        printf("%-22s %s blocks %4d  %.3f ms  -> %.2f cycles per VALU per SIMD (at 8 waves/SIMD), %.1f cycles per VALU per wave\n", names[mm],
               mixed ? "[odd waves only]" : "                ", bl, ms, ms * 1e-3 * 2.4e9 / ITER / 8 / 16, ms * 1e-3 * 2.4e9 / ITER / 16);
    }
    return 0;
}
