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

// {m, m*x, m*y, m*z} prefix-sum element (float64).
struct Moment {
    double m, x, y, z;
};

// ---- sort / scan primitives (sortscan.hip) ---------------------------------------------
size_t sort_pairs_temp_bytes(size_t n, int begin_bit, int end_bit);
hipError_t sort_pairs_u64_u32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit,
                              int end_bit, hipStream_t s);
size_t sort_pairs32_temp_bytes(size_t n, int begin_bit, int end_bit);
hipError_t sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit,
                              int end_bit, hipStream_t s);

// One body of a key-sorted run as it travels between GPUs (multi-GPU run exchange): the two
// octant-path key words and the fp32 {x, y, z, G*m} the octree is built from.  32 bytes.
struct RunRec {
    uint64_t hi, lo;
    float x, y, z, gm;
};
}  // namespace nbmi
