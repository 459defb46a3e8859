// Hand-written device radix sort of (key, 32-bit value) pairs for gfx950: the octant-path keys of the
// octree build (63-bit key word + body index) and the boids' 24-bit cell indices.
//
// Least-significant-digit first, 8-bit digits, ONE kernel per pass ("onesweep"): a workgroup owns a tile of
// 4 096 pairs, ranks them by digit with wave-level match operations (ballots), learns how many pairs of
// each digit the tiles before it hold through a decoupled look-back over per-tile status words, reorders
// the tile in LDS so that every digit's pairs leave as one contiguous run, and writes them to their final
// place of this pass.  Stable.  Per pass every pair is read once and written once (24 B of traffic for a
// u64 key + u32 value); the digit histograms of ALL passes come from one extra read of the keys up front.
//
// Look-back notes (gfx950: eight XCDs, L2s not coherent with each other): a status word carries flag and
// count in ONE 32-bit granule and is written / polled with agent-scope relaxed atomics (sc1), so no
// payload has to be ordered behind a flag.  Tile numbers come from an atomic ticket, so a tile only ever
// waits for tiles whose workgroups are already running.  A tile inspects 8 predecessors per round trip
// (their loads are issued together): with a whole grid starting at once the serial form would walk up to
// `tiles` predecessors one memory round trip at a time.  Every spin is bounded; a timeout sets an error
// word instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include <stddef.h>

namespace nbmi {
namespace {

constexpr int kRadixBits = 8;
constexpr int kBins = 1 << kRadixBits;
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 16, kItemsMin = 4;   // pairs per thread: a workgroup's tile is 4 096 (or 2 048) pairs, a wave ranks 1 024 (512)
                                            // consecutive ones (template parameter of k_radix_pass; the status words are
                                            // sized for the smaller tile)
constexpr unsigned kFlagAgg = 1u << 30, kFlagIncl = 2u << 30, kValueMask = (1u << 30) - 1;
constexpr int kLookBatch = 8;
constexpr int kMaxPasses = 8;

struct Control {                 // lives at the start of the temp buffer
    unsigned error;              // 1: a look-back spin timed out.  STICKY: cleared by radix_init_temp() only, not by the
    unsigned pad0[15];           // per-sort clear (which starts at tile_ticket), so that the owner of the buffer still
                                 // finds it at its next synchronisation however many sorts have run since
    unsigned tile_ticket[kMaxPasses];
    unsigned pad[8];
    unsigned hist[kMaxPasses][kBins];  // global digit counts, then exclusive offsets
};

template <typename K>
__device__ __forceinline__ unsigned digit_of(K k, int shift) {
    return (unsigned)(k >> shift) & (kBins - 1);
}

// ---- digit histograms of all passes: one read of the keys ------------------------------------
template <typename K>
__global__ __launch_bounds__(kThreads) void k_radix_hist(const K *__restrict__ keys, int64_t n, int passes, int first_bit,
                                                        Control *ctl) {
    __shared__ unsigned h[kMaxPasses][kBins];
    for (int i = threadIdx.x; i < kMaxPasses * kBins; i += kThreads) (&h[0][0])[i] = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const K k = keys[i];
        // [r3] The octree keys arrive nearly sorted (the state is kept in last step's key order), so the 64 keys of a
        // wave share their upper digits: 64 LDS atomics on ONE counter, serialised (62 % of this kernel's wave cycles
        // were LDS stalls).  A digit the whole wave agrees on is counted by one lane.
        const unsigned live = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
        for (int p = 0; p < passes; p++) {
            const unsigned d = digit_of(k, first_bit + p * kRadixBits);
            const unsigned d0 = __builtin_amdgcn_readfirstlane(d);
            if (__builtin_amdgcn_ballot_w64(d != d0) == 0ull) {
                if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) atomicAdd(&h[p][d0], live);
            } else {
                atomicAdd(&h[p][d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * kBins; i += kThreads) {
        const unsigned v = (&h[0][0])[i];
        if (v) atomicAdd(&(&ctl->hist[0][0])[i], v);
    }
}

// exclusive scan of each pass's 256 counts (one workgroup per pass)
__global__ __launch_bounds__(kBins) void k_radix_offsets(Control *ctl) {
    __shared__ unsigned wsum[kBins / 64];
    unsigned *h = ctl->hist[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned own = h[t];
    unsigned v = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    h[t] = base + v - own;
}

// ---- one pass --------------------------------------------------------------------------------
template <typename K, int ITEMS>
__global__ __launch_bounds__(kThreads) void k_radix_pass(const K *__restrict__ kin, K *__restrict__ kout,
                                                        const uint32_t *__restrict__ vin, uint32_t *__restrict__ vout,
                                                        int64_t n, int shift, int pass, Control *ctl,
                                                        unsigned *__restrict__ status /* [tiles][256] of this pass */) {
    __shared__ unsigned s_tile;
    __shared__ unsigned cnt_w[kWaves][kBins];   // per-wave digit counts, then the wave's base inside the digit
    __shared__ unsigned tile_off[kBins];        // first slot of the digit inside the reordered tile
    __shared__ unsigned glob_off[kBins];        // global slot of reordered slot q of digit d = glob_off[d] + q
    __shared__ unsigned wave_tot[kWaves];
    // the reorder buffer is used twice, for the keys and then for the values (38 KB instead of 54 KB of LDS:
    // four workgroups per CU instead of two)
    __shared__ K lds_k[(kThreads * ITEMS)];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) s_tile = atomicAdd(&ctl->tile_ticket[pass], 1u);
    for (int i = t; i < kWaves * kBins; i += kThreads) (&cnt_w[0][0])[i] = 0u;
    __syncthreads();
    const unsigned tile = s_tile;
    const int64_t tile_base = (int64_t)tile * (kThreads * ITEMS);
    const int valid_in_tile = (int)((n - tile_base) < (kThreads * ITEMS) ? (n - tile_base) : (kThreads * ITEMS));

    // a wave owns (64 * ITEMS) consecutive pairs and loads them 64 at a time (coalesced); the order inside
    // the tile is wave-major, then round, then lane
    K key[ITEMS];
    uint32_t val[ITEMS];
    unsigned rank[ITEMS];
    const int64_t wave_base = tile_base + (int64_t)w * (64 * ITEMS);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int64_t idx = wave_base + i * 64 + lane;
        const bool ok = idx < n;
        key[i] = ok ? kin[idx] : (K)~(K)0;
        val[i] = ok ? vin[idx] : 0u;
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int64_t idx = wave_base + i * 64 + lane;
        const bool ok = idx < n;
        const unsigned d = digit_of(key[i], shift);
        // lanes of this round with the same digit
        unsigned long long peers = __builtin_amdgcn_ballot_w64(ok);
#pragma unroll
        for (int b = 0; b < kRadixBits; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const unsigned before = (unsigned)__popcll(peers & lt_mask);
        const unsigned pre = ok ? cnt_w[w][d] : 0u;  // pairs of this digit in the wave's earlier rounds
        rank[i] = pre + before;
        // the wave's LDS operations execute in order: every peer has read `pre` before the leader's update
        if (ok && before == 0u) cnt_w[w][d] = pre + (unsigned)__popcll(peers);
    }
    __syncthreads();

    // thread t = digit t: totals, wave bases, tile offsets, look-back
    unsigned total = 0;
    {
        unsigned c[kWaves];
#pragma unroll
        for (int k = 0; k < kWaves; k++) { c[k] = cnt_w[k][t]; }
#pragma unroll
        for (int k = 0; k < kWaves; k++) { cnt_w[k][t] = total; total += c[k]; }
    }
    // exclusive scan of the 256 totals across the workgroup
    unsigned incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int k = 0; k < w; k++) wbase += wave_tot[k];
    const unsigned my_tile_off = wbase + incl - total;
    tile_off[t] = my_tile_off;

    // decoupled look-back: pairs of digit t in the tiles before this one
    unsigned *mine = status + (size_t)tile * kBins + t;
    unsigned prefix = 0;
    if (tile == 0) {
        __hip_atomic_store(mine, kFlagIncl | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(mine, kFlagAgg | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int p = (int)tile - 1;
        unsigned spins = 0;
        bool done = false;
        while (!done) {
            unsigned sw[kLookBatch];
#pragma unroll
            for (int j = 0; j < kLookBatch; j++) {
                const int q = p - j;
                sw[j] = q >= 0 ? __hip_atomic_load(status + (size_t)q * kBins + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kFlagIncl;  // before tile 0: nothing
            }
#pragma unroll
            for (int j = 0; j < kLookBatch; j++) {
                if (done) break;
                const unsigned flag = sw[j] & ~kValueMask;
                if (flag == 0u) {  // not published yet: poll again from this tile
                    if (++spins > (1u << 22)) { ctl->error = 1u; done = true; }
                    __builtin_amdgcn_s_sleep(2);
                    break;
                }
                prefix += sw[j] & kValueMask;
                if (flag == kFlagIncl) done = true;
                else p--;
            }
        }
        __hip_atomic_store(mine, kFlagIncl | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    glob_off[t] = ctl->hist[pass][t] + prefix - my_tile_off;
    __syncthreads();

    // reorder inside the tile: digit runs, each in input order
    unsigned slot[ITEMS / 2];  // two 16-bit slots per register
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int64_t idx = wave_base + i * 64 + lane;
        unsigned q = 0;
        if (idx < n) {
            const unsigned d = digit_of(key[i], shift);
            q = tile_off[d] + cnt_w[w][d] + rank[i];
            lds_k[q] = key[i];
        }
        if (i & 1) slot[i >> 1] |= q << 16; else slot[i >> 1] = q;
    }
    __syncthreads();
    unsigned dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int q = i * kThreads + t;
        if (q < valid_in_tile) {
            const K k = lds_k[q];
            dst[i] = glob_off[digit_of(k, shift)] + (unsigned)q;
            kout[dst[i]] = k;
        }
    }
    __syncthreads();
    uint32_t *lds_v = reinterpret_cast<uint32_t *>(lds_k);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int64_t idx = wave_base + i * 64 + lane;
        if (idx < n) lds_v[(slot[i >> 1] >> ((i & 1) * 16)) & 0xffffu] = val[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int q = i * kThreads + t;
        if (q < valid_in_tile) vout[dst[i]] = lds_v[q];
    }
}

inline int passes_for(int bits) { return (bits + kRadixBits - 1) / kRadixBits; }
inline size_t tiles_for(size_t n, int items = kItemsMin) { return (n + (size_t)kThreads * items - 1) / ((size_t)kThreads * items); }
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename K>
size_t temp_bytes(size_t n, int bits) {
    const int passes = passes_for(bits);
    return align256(sizeof(Control)) + align256((size_t)passes * tiles_for(n) * kBins * sizeof(unsigned)) +
           align256(n * sizeof(K)) + align256(n * sizeof(uint32_t));
}

template <typename K>
hipError_t sort_pairs(void *temp, size_t temp_size, const K *kin, K *kout, const uint32_t *vin, uint32_t *vout,
                      size_t n, int begin_bit, int end_bit, hipStream_t st) {
    const int bits = end_bit - begin_bit;
    if (n == 0) return hipSuccess;
    if (n > (size_t)kValueMask) return hipErrorInvalidValue;  // counts travel in 30 bits
    const int passes = passes_for(bits);
    if (passes < 1 || passes > kMaxPasses || temp_size < temp_bytes<K>(n, bits)) return hipErrorInvalidValue;
    char *base = (char *)temp;
    Control *ctl = (Control *)base;
    // [r4] pairs per thread by size (profiles/r04_small_systems.txt): a pass over few pairs is a chain of latencies, and
    // more, smaller workgroups shorten it - sort phase at 10 k bodies 0.067 / 0.054 / 0.050 ms with 16 / 8 / 4, at 262 k
    // 0.126 / 0.120 / 0.125; from 1 M on the large tile wins (0.151 against 0.169 ms with 8, 10 M: 0.70 against 0.82)
    const int items = n <= 65536 ? 4 : (n <= 524288 ? 8 : kItems);
    const size_t tiles = tiles_for(n, items);
    unsigned *status = (unsigned *)(base + align256(sizeof(Control)));
    const size_t status_bytes = align256((size_t)passes * tiles_for(n) * kBins * sizeof(unsigned));
    K *ktmp = (K *)((char *)status + status_bytes);
    uint32_t *vtmp = (uint32_t *)((char *)ktmp + align256(n * sizeof(K)));
    // everything but the sticky error word at the head of the control block
    hipError_t e = hipMemsetAsync(base + offsetof(Control, tile_ticket), 0,
                                  align256(sizeof(Control)) - offsetof(Control, tile_ticket) + status_bytes, st);
    if (e != hipSuccess) return e;
    int hb = (int)((n + kThreads * 8 - 1) / (kThreads * 8));
    if (hb > 1024) hb = 1024;
    k_radix_hist<K><<<hb, kThreads, 0, st>>>(kin, (int64_t)n, passes, begin_bit, ctl);
    k_radix_offsets<<<passes, kBins, 0, st>>>(ctl);
    // ping-pong so that the last pass writes the caller's output: ... -> tmp -> out
    const K *ksrc = kin;
    const uint32_t *vsrc = vin;
    for (int p = 0; p < passes; p++) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        K *kdst = to_out ? kout : ktmp;
        uint32_t *vdst = to_out ? vout : vtmp;
        if (items == 4)
            k_radix_pass<K, 4><<<(int)tiles, kThreads, 0, st>>>(ksrc, kdst, vsrc, vdst, (int64_t)n, begin_bit + p * kRadixBits, p, ctl,
                                                                status + (size_t)p * tiles * kBins);
        else if (items == 8)
            k_radix_pass<K, 8><<<(int)tiles, kThreads, 0, st>>>(ksrc, kdst, vsrc, vdst, (int64_t)n, begin_bit + p * kRadixBits, p, ctl,
                                                                status + (size_t)p * tiles * kBins);
        else
            k_radix_pass<K, 16><<<(int)tiles, kThreads, 0, st>>>(ksrc, kdst, vsrc, vdst, (int64_t)n, begin_bit + p * kRadixBits, p, ctl,
                                                                 status + (size_t)p * tiles * kBins);
        ksrc = kdst;
        vsrc = vdst;
    }
    return hipGetLastError();
}

}  // namespace

// ---- entry points used by nbmi.hip / bdmi.hip ---------------------------------------------------
size_t radix_temp_bytes_u64(size_t n, int bits) { return temp_bytes<uint64_t>(n, bits); }
size_t radix_temp_bytes_u32(size_t n, int bits) { return temp_bytes<uint32_t>(n, bits); }
hipError_t radix_sort_pairs_u64(void *temp, size_t temp_size, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    return sort_pairs<uint64_t>(temp, temp_size, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}
hipError_t radix_sort_pairs_u32(void *temp, size_t temp_size, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t>(temp, temp_size, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}
// A freshly allocated temp buffer: clears the sticky error word (once, by whoever allocated the buffer).
hipError_t radix_init_temp(void *temp, hipStream_t s) { return hipMemsetAsync(temp, 0, offsetof(Control, tile_ticket), s); }
// 1 if a look-back of ANY sort on this temp buffer since radix_init_temp() timed out (never observed; the spin is
// bounded so that a lost status word cannot hang the GPU).  Such a pass scattered to wrong offsets: the product
// paths read this word wherever they synchronise anyway and report NBMI_ERR_HIP.
hipError_t radix_error_word(const void *temp, unsigned *out, hipStream_t s) {
    return hipMemcpyAsync(out, &((const Control *)temp)->error, sizeof(unsigned), hipMemcpyDeviceToHost, s);
}

}  // namespace nbmi
