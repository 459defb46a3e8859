// Internal helpers shared by the translation units of libnbmi.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace nbmi {

void set_error(const char *fmt, ...);
const char *get_error();
void clear_error();

#define NBMI_HIP_CHECK(expr)                                                                \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (void)hipGetLastError(); /* clear the sticky error so later calls start clean */ \
            nbmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                      \
            return -2;                                                                      \
        }                                                                                   \
    } while (0)

// {m, m*x, m*y, m*z} prefix-sum element, each a double-double (high word, low word).
struct Moment {
    double m, ml, x, xl, y, yl, z, zl;
};

// ---- sort / scan primitives (sortscan.hip) ---------------------------------------------
size_t sort_pairs_temp_bytes(size_t n, int begin_bit, int end_bit);
hipError_t sort_pairs_u64_u32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit,
                              int end_bit, hipStream_t s);
size_t sort_pairs32_temp_bytes(size_t n, int begin_bit, int end_bit);
hipError_t sort_init_temp(void *temp, hipStream_t s);                          // once per freshly allocated temp buffer
hipError_t sort_error_word(const void *temp, unsigned *out, hipStream_t s);  // sticky: 1 = some sort since then went wrong
hipError_t sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit,
                              int end_bit, hipStream_t s);

// the hand-written radix sort (radix.hip): by key bits [begin_bit, end_bit), stable, ping-pong inside `temp`
size_t radix_temp_bytes_u64(size_t n, int bits);
size_t radix_temp_bytes_u32(size_t n, int bits);
hipError_t radix_sort_pairs_u64(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t s);
hipError_t radix_sort_pairs_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t s);
hipError_t radix_init_temp(void *temp, hipStream_t s);
hipError_t radix_error_word(const void *temp, unsigned *out, hipStream_t s);

// One body of a key-sorted run as it travels between GPUs (multi-GPU run exchange): the two
// octant-path key words and the fp32 {x, y, z, G*m} the octree is built from.  32 bytes.
struct RunRec {
    uint64_t hi, lo;
    float x, y, z, gm;
};

// ---- device-side initial conditions (icgen.hip) ---------------------------------------------
#define NBMI_IC_GALAXY 0
#define NBMI_IC_COLLISION 1
#define NBMI_IC_CLUSTER 2
struct IcArrays {
    double *x, *y, *z, *vx, *vy, *vz, *m;
    int32_t *id;
};
int ic_generate(int distribution, int64_t n, double R, double G, uint64_t seed, IcArrays out, hipStream_t s);

// Philox4x32-10 (Salmon et al., SC'11): counter-based generator, 4 x 32 bits per call.
__host__ __device__ inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
}  // namespace nbmi
