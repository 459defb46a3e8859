// Device sorts of (key, 32-bit value) pairs used by the octree build and the boids grid: the hand-written radix sort
// of radix.hip behind one small interface.  [r4] rocPRIM is no longer compiled into the product: its cross-check lives
// in tests/native/rocprim_check.hip (a test-only library that tests/test_gpu_sort.py compares this sort with, bit for
// bit); nbmi_debug_sort_pairs runs the product's sort on caller-supplied arrays for that comparison.
#include <cstdlib>
#include <cstring>

#include "../../include/nbmi.h"
#include "common.h"

namespace nbmi {

size_t sort_pairs_temp_bytes(size_t n, int begin_bit, int end_bit) { return radix_temp_bytes_u64(n, end_bit - begin_bit); }

hipError_t sort_pairs_u64_u32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    return radix_sort_pairs_u64(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

// the sticky error word of the sort (a look-back spin that timed out)
hipError_t sort_init_temp(void *temp, hipStream_t s) { return radix_init_temp(temp, s); }
hipError_t sort_error_word(const void *temp, unsigned *out, hipStream_t s) { return radix_error_word(temp, out, s); }

size_t sort_pairs32_temp_bytes(size_t n, int begin_bit, int end_bit) { return radix_temp_bytes_u32(n, end_bit - begin_bit); }

hipError_t sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    return radix_sort_pairs_u32(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

}  // namespace nbmi

// Test / measurement hook: sorts caller-supplied host arrays on the device with the product's sort (impl must be 0).
extern "C" int nbmi_debug_sort_pairs(int key_bytes, int64_t n, const void *keys, const uint32_t *values, void *keys_out,
                                     uint32_t *values_out, int bits, int impl, int repeats, double *ms_per_sort) {
    if ((key_bytes != 4 && key_bytes != 8) || n < 0 || bits < 1 || bits > 8 * key_bytes || (n && (!keys || !values)) || impl != 0) {
        nbmi::set_error("nbmi_debug_sort_pairs: bad arguments");
        return NBMI_ERR_ARG;
    }
    if (n == 0) return 0;
    const size_t kb = (size_t)n * key_bytes, vb = (size_t)n * 4;
    const size_t own = key_bytes == 8 ? nbmi::radix_temp_bytes_u64(n, bits) : nbmi::radix_temp_bytes_u32(n, bits);
    size_t tb = own + 256;
    void *dk = nullptr, *dko = nullptr, *dv = nullptr, *dvo = nullptr, *tmp = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto fail = [&](const char *what, hipError_t e) {
        nbmi::set_error("nbmi_debug_sort_pairs: %s: %s", what, hipGetErrorString(e));
        rc = NBMI_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&dk, kb)) || (e = hipMalloc(&dko, kb)) || (e = hipMalloc(&dv, vb)) || (e = hipMalloc(&dvo, vb)) ||
        (e = hipMalloc(&tmp, tb)) || (e = hipStreamCreate(&st)) || (e = nbmi::radix_init_temp(tmp, st)) || (e = hipEventCreate(&e0)) || (e = hipEventCreate(&e1)) ||
        (e = hipMemcpyAsync(dk, keys, kb, hipMemcpyHostToDevice, st)) ||
        (e = hipMemcpyAsync(dv, values, vb, hipMemcpyHostToDevice, st)))
        fail("setup", e);
    for (int r = 0; rc == 0 && r < (repeats < 1 ? 1 : repeats) + 1; r++) {  // first run untimed
        if (r == 1) (void)hipEventRecord(e0, st);
        if (key_bytes == 8) {
            e = nbmi::radix_sort_pairs_u64(tmp, tb, (const uint64_t *)dk, (uint64_t *)dko, (const uint32_t *)dv,
                                           (uint32_t *)dvo, (size_t)n, 0, bits, st);
        } else {
            e = nbmi::radix_sort_pairs_u32(tmp, tb, (const uint32_t *)dk, (uint32_t *)dko, (const uint32_t *)dv,
                                           (uint32_t *)dvo, (size_t)n, 0, bits, st);
        }
        if (e != hipSuccess) fail("sort", e);
    }
    if (rc == 0) {
        (void)hipEventRecord(e1, st);
        unsigned err = 0;
        (void)nbmi::radix_error_word(tmp, &err, st);
        if ((e = hipMemcpyAsync(keys_out, dko, kb, hipMemcpyDeviceToHost, st)) ||
            (e = hipMemcpyAsync(values_out, dvo, vb, hipMemcpyDeviceToHost, st)) || (e = hipStreamSynchronize(st)))
            fail("copy back", e);
        float ms = 0.f;
        if (rc == 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms_per_sort)
            *ms_per_sort = ms / (repeats < 1 ? 1 : repeats);
        if (rc == 0 && err) {
            nbmi::set_error("nbmi_debug_sort_pairs: a look-back spin timed out");
            rc = NBMI_ERR_HIP;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    for (void *q : {dk, dko, dv, dvo, tmp})
        if (q) (void)hipFree(q);
    return rc;
}
