// Render-side reductions shared by libnbmi's two handles: frustum test per body,
// then an order-preserving compaction in the caller's ORIGINAL body order, so that only what a
// viewer would upload crosses PCIe.  The bodies live on the device in key / cell order with their
// original index in `id`, hence: mark by id -> count per tile of ids -> scan the tile counts ->
// emit.  All float64, same operation order as the reference's functions (built with
// -ffp-contract=off): the masks are bit-identical to a CPython run of
//   compute_visibility_points   nbody/simulation.py:403-434   (z_near 0.1, margin 1.2)
//   compute_visibility_numba    boids/flock.py:311-348        (z_near 0.5, margin 1.0)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vis {

constexpr int kBlock = 256;
constexpr int kItems = 8;                  // ids per thread
constexpr int kTile = kBlock * kItems;     // ids per block

struct Camera {
    double p[3], f[3], r[3], u[3];
    double tan_h, tan_v, z_near, z_far, margin;
};

__device__ __forceinline__ bool frustum_visible(const Camera &c, double px, double py, double pz) {
    const double dx = px - c.p[0], dy = py - c.p[1], dz = pz - c.p[2];
    const double z = dx * c.f[0] + dy * c.f[1] + dz * c.f[2];
    if (z < c.z_near || z > c.z_far) return false;
    const double x = dx * c.r[0] + dy * c.r[1] + dz * c.r[2];
    const double y = dx * c.u[0] + dy * c.u[1] + dz * c.u[2];
    const double half_width = z * c.tan_h * c.margin;   // margin 1.0 multiplies exactly
    const double half_height = z * c.tan_v * c.margin;
    return fabs(x) < half_width && fabs(y) < half_height;
}

// slot = where the body is stored now; id[slot] = its original index
static __global__ __launch_bounds__(kBlock) void k_mark(const double *__restrict__ x, const double *__restrict__ y,
                                                        const double *__restrict__ z, const int32_t *__restrict__ id,
                                                        int64_t n, Camera c, uint8_t *__restrict__ flag,
                                                        uint32_t *__restrict__ slot_of) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const int32_t i = id[s];
    flag[i] = frustum_visible(c, x[s], y[s], z[s]) ? 1 : 0;
    slot_of[i] = (uint32_t)s;
}

__device__ __forceinline__ unsigned tile_flags(const uint8_t *flag, int64_t n, int64_t first) {
    // 8 flags of this thread as a bit mask (flag[] is padded to a multiple of 8 and zero beyond n)
    const unsigned long long w = *reinterpret_cast<const unsigned long long *>(flag + first);
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < kItems; k++) m |= (unsigned)((w >> (8 * k)) & 1ull) << k;
    return first < n ? m : 0u;
}

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *total) {
    __shared__ unsigned wave_sum[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        if (w < wave) base += wave_sum[w];
        all += wave_sum[w];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

static __global__ __launch_bounds__(kBlock) void k_count(const uint8_t *__restrict__ flag, int64_t n,
                                                         uint32_t *__restrict__ tile_cnt) {
    const int64_t first = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
    unsigned total;
    (void)block_exclusive_scan(__popc(tile_flags(flag, n, first)), &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// exclusive scan of the tile counts in place; tile_cnt[ntiles] = number of visible bodies
static __global__ __launch_bounds__(kBlock) void k_scan_tiles(uint32_t *__restrict__ tile_cnt, int64_t ntiles) {
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += kBlock) {
        const int64_t i = base + threadIdx.x;
        const unsigned v = i < ntiles ? tile_cnt[i] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan(v, &total);
        const unsigned c0 = carry;
        if (i < ntiles) tile_cnt[i] = c0 + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c0 + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_cnt[ntiles] = carry;
}

// Emit(k, i, slot): write output row k for original index i stored at `slot`
template <class Emit>
static __global__ __launch_bounds__(kBlock) void k_emit(const uint8_t *__restrict__ flag,
                                                        const uint32_t *__restrict__ slot_of,
                                                        const uint32_t *__restrict__ tile_cnt, int64_t n,
                                                        int64_t capacity, Emit emit) {
    const int64_t first = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kItems;
    const unsigned m = tile_flags(flag, n, first);
    unsigned total;
    int64_t k = (int64_t)tile_cnt[blockIdx.x] + block_exclusive_scan(__popc(m), &total);
#pragma unroll
    for (int b = 0; b < kItems; b++) {
        if ((m >> b) & 1u) {
            if (k < capacity) emit(k, first + b, slot_of[first + b]);
            k++;
        }
    }
}

inline int64_t tiles_for(int64_t n) { return (n + kTile - 1) / kTile; }
inline size_t flag_bytes(int64_t n) { return (size_t)tiles_for(n) * kTile + 8; }

}  // namespace vis
