// TEST INFRASTRUCTURE ONLY: rocPRIM's radix sort on caller-supplied host arrays, the cross-check of the product's own
// sort (csrc/radix.hip) in tests/test_gpu_sort.py.  Built into tests/native/librocprim_check.so by
// __graft_entry__.build(); the product library libnbmi.so does not contain or link rocPRIM.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdint>

template <class K>
static int run(int64_t n, const K *keys, const uint32_t *values, K *keys_out, uint32_t *values_out, int bits, int repeats,
               double *ms_per_sort) {
    const size_t kb = (size_t)n * sizeof(K), vb = (size_t)n * 4;
    size_t tb = 0;
    if (rocprim::radix_sort_pairs<rocprim::default_config, const K *, K *, const uint32_t *, uint32_t *>(
            nullptr, tb, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, bits, 0) != hipSuccess)
        return -1;
    void *dk = nullptr, *dko = nullptr, *dv = nullptr, *dvo = nullptr, *tmp = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    if (hipMalloc(&dk, kb) || hipMalloc(&dko, kb) || hipMalloc(&dv, vb) || hipMalloc(&dvo, vb) || hipMalloc(&tmp, tb + 256) ||
        hipStreamCreate(&st) || hipEventCreate(&e0) || hipEventCreate(&e1) ||
        hipMemcpyAsync(dk, keys, kb, hipMemcpyHostToDevice, st) || hipMemcpyAsync(dv, values, vb, hipMemcpyHostToDevice, st))
        rc = -2;
    for (int r = 0; rc == 0 && r < (repeats < 1 ? 1 : repeats) + 1; r++) {  // first run untimed
        if (r == 1) (void)hipEventRecord(e0, st);
        if (rocprim::radix_sort_pairs(tmp, tb, (const K *)dk, (K *)dko, (const uint32_t *)dv, (uint32_t *)dvo, (size_t)n, 0, bits, st) != hipSuccess)
            rc = -3;
    }
    if (rc == 0) {
        (void)hipEventRecord(e1, st);
        if (hipMemcpyAsync(keys_out, dko, kb, hipMemcpyDeviceToHost, st) || hipMemcpyAsync(values_out, dvo, vb, hipMemcpyDeviceToHost, st) ||
            hipStreamSynchronize(st))
            rc = -4;
        float ms = 0.f;
        if (rc == 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms_per_sort) *ms_per_sort = ms / (repeats < 1 ? 1 : repeats);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    for (void *q : {dk, dko, dv, dvo, tmp})
        if (q) (void)hipFree(q);
    return rc;
}

extern "C" int rocprim_check_sort_pairs(int key_bytes, int64_t n, const void *keys, const uint32_t *values, void *keys_out,
                                        uint32_t *values_out, int bits, int repeats, double *ms_per_sort) {
    if (n <= 0) return 0;
    if (key_bytes == 8) return run<uint64_t>(n, (const uint64_t *)keys, values, (uint64_t *)keys_out, values_out, bits, repeats, ms_per_sort);
    if (key_bytes == 4) return run<uint32_t>(n, (const uint32_t *)keys, values, (uint32_t *)keys_out, values_out, bits, repeats, ms_per_sort);
    return -1;
}
