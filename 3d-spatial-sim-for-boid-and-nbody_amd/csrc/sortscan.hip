// Device radix sorts used by the octree build and the boids grid.
// These are plain library primitives (rocPRIM); the hand-written kernels live in nbmi.hip /
// bdmi.hip.  Kept in their own translation unit because the rocPRIM templates dominate
// compile time.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace nbmi {

size_t sort_pairs_temp_bytes(size_t n, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint64_t *, uint64_t *, const uint32_t *,
                                    uint32_t *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, begin_bit,
                                                end_bit, 0);
    return bytes;
}

hipError_t sort_pairs_u64_u32(void *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

size_t sort_pairs32_temp_bytes(size_t n, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const uint32_t *,
                                    uint32_t *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, begin_bit,
                                                end_bit, 0);
    return bytes;
}

hipError_t sort_pairs_u32_u32(void *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout,
                              const uint32_t *vin, uint32_t *vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

}  // namespace nbmi
