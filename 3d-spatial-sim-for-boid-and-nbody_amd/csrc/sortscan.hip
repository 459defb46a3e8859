// Device sorts of (key, 32-bit value) pairs used by the octree build and the boids grid.
// The product path is the hand-written radix sort of radix.hip.  rocPRIM's radix sort stays linked as the
// cross-check (NBMI_SORT=rocprim selects it for a whole process; nbmi_debug_sort_pairs runs either on
// caller-supplied arrays so that tests can compare the two bit for bit).  Kept in its own translation
// unit because the rocPRIM templates dominate compile time.
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "../../include/nbmi.h"
#include "common.h"

namespace nbmi {

static bool use_rocprim() {
    static const bool v = [] {
        const char *e = getenv("NBMI_SORT");
        return e && !strcmp(e, "rocprim");
    }();
    return v;
}

static size_t rocprim_bytes_u64(size_t n, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint64_t *, uint64_t *, const uint32_t *,
                                    uint32_t *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, begin_bit,
                                                end_bit, 0);
    return bytes;
}
static size_t rocprim_bytes_u32(size_t n, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const uint32_t *,
                                    uint32_t *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, begin_bit,
                                                end_bit, 0);
    return bytes;
}

size_t sort_pairs_temp_bytes(size_t n, int begin_bit, int end_bit) {
    const size_t a = rocprim_bytes_u64(n, begin_bit, end_bit), b = radix_temp_bytes_u64(n, end_bit - begin_bit);
    return a > b ? a : b;
}

hipError_t sort_pairs_u64_u32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    if (use_rocprim())
        return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
    return radix_sort_pairs_u64(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

// The sticky error word of the hand-written sort (radix.hip) behind the dispatcher: rocPRIM owns the whole temp
// buffer when it is selected and has no such word.
hipError_t sort_init_temp(void *temp, hipStream_t s) { return radix_init_temp(temp, s); }
hipError_t sort_error_word(const void *temp, unsigned *out, hipStream_t s) {
    if (use_rocprim()) {
        *out = 0u;
        return hipSuccess;
    }
    return radix_error_word(temp, out, s);
}

size_t sort_pairs32_temp_bytes(size_t n, int begin_bit, int end_bit) {
    const size_t a = rocprim_bytes_u32(n, begin_bit, end_bit), b = radix_temp_bytes_u32(n, end_bit - begin_bit);
    return a > b ? a : b;
}

hipError_t sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    if (use_rocprim())
        return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
    return radix_sort_pairs_u32(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

}  // namespace nbmi

// Test / measurement hook: sorts caller-supplied host arrays on the device with either implementation.
extern "C" int nbmi_debug_sort_pairs(int key_bytes, int64_t n, const void *keys, const uint32_t *values, void *keys_out,
                                     uint32_t *values_out, int bits, int impl, int repeats, double *ms_per_sort) {
    if ((key_bytes != 4 && key_bytes != 8) || n < 0 || bits < 1 || bits > 8 * key_bytes || (n && (!keys || !values))) {
        nbmi::set_error("nbmi_debug_sort_pairs: bad arguments");
        return NBMI_ERR_ARG;
    }
    if (n == 0) return 0;
    const size_t kb = (size_t)n * key_bytes, vb = (size_t)n * 4;
    const size_t own = key_bytes == 8 ? nbmi::radix_temp_bytes_u64(n, bits) : nbmi::radix_temp_bytes_u32(n, bits);
    const size_t lib = key_bytes == 8 ? nbmi::rocprim_bytes_u64(n, 0, bits) : nbmi::rocprim_bytes_u32(n, 0, bits);
    size_t tb = (own > lib ? own : lib) + 256;
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
            e = impl ? rocprim::radix_sort_pairs(tmp, tb, (const uint64_t *)dk, (uint64_t *)dko, (const uint32_t *)dv,
                                                 (uint32_t *)dvo, (size_t)n, 0, bits, st)
                     : nbmi::radix_sort_pairs_u64(tmp, tb, (const uint64_t *)dk, (uint64_t *)dko, (const uint32_t *)dv,
                                                  (uint32_t *)dvo, (size_t)n, 0, bits, st);
        } else {
            e = impl ? rocprim::radix_sort_pairs(tmp, tb, (const uint32_t *)dk, (uint32_t *)dko, (const uint32_t *)dv,
                                                 (uint32_t *)dvo, (size_t)n, 0, bits, st)
                     : nbmi::radix_sort_pairs_u32(tmp, tb, (const uint32_t *)dk, (uint32_t *)dko, (const uint32_t *)dv,
                                                  (uint32_t *)dvo, (size_t)n, 0, bits, st);
        }
        if (e != hipSuccess) fail("sort", e);
    }
    if (rc == 0) {
        (void)hipEventRecord(e1, st);
        unsigned err = 0;
        if (!impl) (void)nbmi::radix_error_word(tmp, &err, st);
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
