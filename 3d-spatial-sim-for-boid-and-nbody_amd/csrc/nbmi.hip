// libnbmi.so - MI355X (gfx950) N-body backend: Barnes-Hut octree + stackless wave-shared tree
// walk with fused kick-drift, and the LDS-tiled direct O(N^2) fallback.  C ABI in include/nbmi.h.
//
// Reference behaviour being reproduced (file:line in /root/reference):
//   compute_bounds                nbody/simulation.py:308-317
//   get_octant / _center          nbody/simulation.py:38-60      (key digits)
//   build_octree                  nbody/simulation.py:63-198     (one body per leaf; cell SET is
//                                                                  insertion-order independent)
//   compute_forces_barnes_hut     nbody/simulation.py:201-278
//   update_positions_velocities   nbody/simulation.py:281-305
//   compute_colors_by_velocity    nbody/simulation.py:320-400
//   compute_forces_*_cuda, update_bodies_cuda   nbody/gpu_backend.py:145-257
//
// Design (DESIGN.md has the full story):
//   * master state float64 SoA in HBM, kept in key-sorted order (re-sorted every step; the
//     permutation is nearly the identity so the gather is almost coalesced); `id` maps a
//     sorted rank back to the caller's body index.
//   * keys: per-body replay of the reference's compare/halve recurrence in float64, 42 levels
//     (2 x 63 bit), so the cell set equals the reference's bit for bit.
//   * tree: for key-sorted bodies, delta[r] = common octal prefix length of bodies r, r+1.
//     Body r starts internal cells at levels (delta[r-1], delta[r]] and owns one leaf at level
//     max(delta[r-1], delta[r]) + 1.  Nodes are emitted in DFS pre-order with a `next` (skip
//     subtree) index, COM from float64 prefix sums of {m, m x} over the sorted bodies.
//   * walk: one wave64 per 64 consecutive sorted bodies walks the pre-order array with a single
//     wave-uniform cursor (scalar loads); every lane applies the reference's own per-body
//     opening test; the wave descends if ANY lane opens, lanes that accepted an ancestor sit
//     out until the cursor leaves that subtree.  Each lane's accepted set == the reference's.
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "../../include/nbmi.h"
#include "common.h"
#include "hilbert.h"
#include "visible.h"

namespace nbmi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }
void clear_error() { g_err[0] = 0; }
}  // namespace nbmi

using nbmi::Moment;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxLevel = 42;  // key digits available (2 x 21)

// 24-byte octree node, DFS pre-order, read by the walk with scalar loads.  The node that follows in
// memory (first child, or for a leaf the next sibling) is always at own offset + 24, so only the
// "skip my subtree" link is stored; 24 instead of 32 bytes per node is 25 % more nodes per cache
// line (the walk slows by 30 % when the record is padded to 64 bytes: it is that sensitive).
struct alignas(8) Node {
    float cx, cy, cz;   // centre of mass (leaf: the body's position)
    float gm;           // G * mass
    float s2t;          // (2*half_size)^2 / theta^2 (accept when s2t < dist_sq); 0 for leaves
    unsigned next_off;  // BYTE offset (index * 24) of the first node after this node's subtree
};
constexpr unsigned kNodeBytes = 24;
// Float64 twin of an internal cell, row = node index: centre of mass and half size as the reference holds
// them (node_com / node_half_sizes, simulation.py:478-484).  Read only when the fp32 opening test lands
// inside its uncertainty band (K9); leaves have no row content (a leaf is always accepted).
struct alignas(16) Node64 {
    double cx, cy, cz, hs;
};
// [r3] The float64 node record, one per node (leaves too), read by the waves that compute their forces in float64
// (k_walk, "force precision"): centre of mass / body position and G m as the float64 state has them, the opening
// threshold and the skip link.  Same pre-order as `Node`; links are byte offsets in units of THIS record.
struct alignas(8) NodeD {
    double cx, cy, cz, gm;
    float s2t;          // as Node::s2t
    unsigned next_off;  // index * 40
};
constexpr unsigned kNodeDBytes = 40;
constexpr unsigned kBand64 = 2;  // half width (ulps of fp32 d^2) of the float64 loop's uncertainty band, see k_emit_tile
static_assert(sizeof(NodeD) == kNodeDBytes, "NodeD must be 40 bytes");
constexpr int64_t kMaxNodeDRows = 107000000;  // 32-bit byte offsets: 2^32 / 40
// Node links are 32-bit byte offsets: at most 2^32 / 24 rows.  The reference allocates min(8 M, 4N)
// rows (simulation.py:477) and uses ~1.5 N; this build allocates 4N + 4096 rows up to that ceiling,
// which still leaves 1.7 N rows at the largest supported body count.
constexpr int64_t kMaxNodeRows = 178000000;
constexpr int64_t kMaxBodies = 100000000;  // the reference's largest presets have 50 M bodies
inline int64_t node_rows_for(int64_t n) { return 4 * n + 4096 < kMaxNodeRows ? 4 * n + 4096 : kMaxNodeRows; }
// Node `num_nodes` is a sentinel that loops onto itself (next = own offset, zero mass, at
// "infinity", never opened): the unrolled walk may step onto it a few times after the traversal
// has ended.
static_assert(sizeof(Node) == kNodeBytes, "Node must be 24 bytes");

struct Bodies {
    double *x, *y, *z, *vx, *vy, *vz, *m;
    int32_t *id;
};

struct TreeInfo {
    unsigned long long maxabs_bits;  // max |coordinate| (non-negative double bit pattern)
    double bounds;                   // root half size
    long long num_nodes;             // N + number of internal cells (reference numbering)
    long long walk_nodes;            // nodes the walk covers: num_nodes, plus received trees in owner mode
    int max_level;                   // deepest leaf level
    int error;                       // 1 = node capacity exceeded
    int max_run;                     // longest run of bodies equal in the radix-sorted key prefix (> 64 only)
    int pad0;
    unsigned long long maxabs_next;  // max |coordinate| of the positions this step's integrating walk WRITES: the next
                                     // step's maxabs_bits without a pass over the bodies (k_header)
    unsigned long long wave_visits, lane_visits, lane_accepts;
    unsigned long long win_miss[4];  // counted walk: node-window misses for windows of 8/16/32/64 nodes
    unsigned long long jumps;        // cursor moves other than to the next node in memory
    unsigned long long xcd_visits[8];  // counted walk: wave-level visits executed on each XCD
    unsigned long long band_visits;    // counted walk: lane visits decided by the float64 re-test (near-ties)
    unsigned band2;                    // width (ulps of d^2) of the opening test's uncertainty band, see K9
    // Sticky: set together with `error`, but outside the range every step clears.  While it is set the
    // walk does not advance the state (the bodies keep the last good step), so an overflow in substep 3
    // of 10 is still there when the host looks (nbmi_sync / getters), which reports and clears it.
    int sticky_error;
    long long sticky_nodes;  // num_nodes of the build that overflowed
    // force precision "auto": 1 while a large part of the system asks for float64 - then every wave computes in it.
    // Entered when more than a third of the step's waves ask, left when fewer than a quarter do (hysteresis: a system
    // hovering at the threshold must not flip the whole walk between the two loops step by step).  [r4] Round 3
    // entered above one half: the held-out 1 M collision at dt 0.25 (36 % of the waves asking at the start, 61 % after
    // 100 steps) then ended at 3.1e-5 - 3 x inside the bound - where every wave in float64 ends at 2.5e-14 and costs
    // nothing extra (the waves that ask are the ones that visit most nodes: with 48 % of them in float64 the walk
    // already takes the all-float64 time; profiles/r04_precision_cases.jsonl).  Decided by k_scan_subtiles from
    // (ask_waves, n_waves) of the handle's own bodies; in owner mode with several ranks the host sums the ranks'
    // votes and sets the verdict for all of them (nbmi_owner_set_all64) - one system, one decision, whatever the
    // world size.  Outside the ranges the per-step header reset clears: the value of the last step is the state.
    int force_all64;
    int ask_waves, n_waves;  // this build's votes: waves whose own density asked for float64 / waves
    int pad1;
    // owner mode [r3]: the own tree in its global form (k_chain_fix) - how many of its first nodes are copies of cells
    // that begin on a lower rank (not walked, not exported), and where the walk array begins behind the jump node
    int own_skip, own_added;
    long long walk_first;
};

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware block remap.  Hardware deals consecutive blocks round-robin over the 8 XCDs (blocks b
// and b+8 share an XCD and its L2).  Neighbouring body groups walk nearly the same nodes, so every
// XCD should get CONTIGUOUS runs of logical blocks; runs of `chunk` blocks are interleaved over
// the XCDs so that dense and sparse regions of the key order are spread evenly (chunk = 0: one
// contiguous eighth per XCD; chunk < 0: identity).  Speed only, never correctness.
__device__ __forceinline__ int logical_block(int b, int nb, int chunk) {
    if (chunk < 0) return b;
    const int xcd = b & 7, j = b >> 3;
    if (chunk == 0) {
        const int q = nb >> 3, rem = nb & 7;
        return xcd * q + (xcd < rem ? xcd : rem) + j;
    }
    const int span = 8 * chunk;
    const int full = (nb / span) * span;  // blocks covered by complete rounds of 8 chunks
    if (b >= full) return b;
    return ((j / chunk) * 8 + xcd) * chunk + (j % chunk);
}

__device__ __forceinline__ int cpl_digits(uint64_t ahi, uint64_t alo, uint64_t bhi, uint64_t blo) {
    const uint64_t xh = ahi ^ bhi;
    if (xh) return (__clzll((long long)xh) - 1) / 3;
    const uint64_t xl = alo ^ blo;
    if (xl) return 21 + (__clzll((long long)xl) - 1) / 3;
    return kMaxLevel;
}

// ---------------------------------------------------------------------------------------
// K1: max |coordinate|  (compute_bounds, simulation.py:308-317; max is order independent)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_maxabs(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, int64_t n, TreeInfo *info) {
    __shared__ double red[kBlock / 64];
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        mx = fmax(mx, fabs(x[i]));
        mx = fmax(mx, fabs(y[i]));
        mx = fmax(mx, fabs(z[i]));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) mx = fmax(mx, red[w]);
        atomicMax(&info->maxabs_bits, (unsigned long long)__double_as_longlong(mx));
    }
}

// The step's tree header when the previous step's integrating walk has already left max |coordinate| of the
// positions it wrote in maxabs_next: take it over and clear the rest (instead of a memset + a pass over the bodies).
__global__ void k_header(TreeInfo *info) {
    if (threadIdx.x != 0) return;
    info->maxabs_bits = info->maxabs_next;
    info->maxabs_next = 0ull;
    info->bounds = 0.0;
    info->num_nodes = 0;
    info->walk_nodes = 0;
    info->max_level = 0;
    info->error = 0;
    info->max_run = 0;
    info->pad0 = 0;
}

// ---------------------------------------------------------------------------------------
// K2: octant-path keys.  Replays get_octant/get_octant_center (simulation.py:38-60) from the
// root cube [-bounds, bounds]^3: digit = x>=cx | (y>=cy)<<1 | (z>=cz)<<2, centre +- hs/2.
// Only compares, adds and exact halvings: bit-identical to the float64 reference.
// ---------------------------------------------------------------------------------------
// [r2] The digits are then RELABELLED along the 3-D Hilbert curve (hilbert.h: digit and next orientation from the
// octant and the cell's orientation, a 24-state machine carried down the 42 levels).  Which bodies share a
// level-L cell - every common-prefix length, hence the cell set, the node count, every cell's bodies - is
// untouched; what changes is the order of a cell's eight children in the sorted array, and with it which 64
// bodies share a wave: waves along the Hilbert curve are more compact and visit 7 % fewer nodes (1 M galaxy;
// 5 % at the 10 M collision; scripts/analysis/hilbert_groups.py).  nbmi_get_keys / nbmi_get_cells decode back
// to the reference's octant digits.  kHilbert = false (NBMI_HILBERT=0): plain octant digits.
template <bool kHilbert>
__global__ __launch_bounds__(kBlock) void k_keys(const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, int64_t n, TreeInfo *info,
                                                 uint64_t *__restrict__ key_hi, uint64_t *__restrict__ key_lo,
                                                 uint32_t *__restrict__ idx, const uint8_t *__restrict__ dead = nullptr) {
    __shared__ uint32_t hd[nbmi::kHilStates];
    __shared__ uint64_t hn[nbmi::kHilStates];
    if (kHilbert) {
        if (threadIdx.x < nbmi::kHilStates) {
            hd[threadIdx.x] = nbmi::kHilDigit[threadIdx.x];
            hn[threadIdx.x] = nbmi::kHilNext[threadIdx.x];
        }
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    // bounds = max_extent * 1.1 + 10.0 with two roundings (no FMA contraction)
    const double maxabs = __longlong_as_double((long long)info->maxabs_bits);
    const double bounds = __dadd_rn(__dmul_rn(maxabs, 1.1), 10.0);
    if (i == 0) info->bounds = bounds;
    if (i >= n) return;
    if (dead && dead[i]) {  // owner mode: this row's body now lives on another rank; all ones sorts behind every key
        key_hi[i] = ~0ull;
        key_lo[i] = ~0ull;
        idx[i] = (uint32_t)i;
        return;
    }
    const double px = x[i], py = y[i], pz = z[i];
    double cx = 0.0, cy = 0.0, cz = 0.0, hs = bounds;
    uint64_t k[2];
    unsigned st = 0u;  // orientation of the current cell (root: 0)
#pragma unroll
    for (int w = 0; w < 2; w++) {
        uint64_t kk = 0;
        for (int l = 0; l < 21; l++) {
            const double q = hs * 0.5;
            const bool bx = px >= cx, by = py >= cy, bz = pz >= cz;
            cx = bx ? cx + q : cx - q;
            cy = by ? cy + q : cy - q;
            cz = bz ? cz + q : cz - q;
            hs = q;
            const unsigned oct = (bx ? 1u : 0u) | (by ? 2u : 0u) | (bz ? 4u : 0u);
            if (kHilbert) {
                kk = (kk << 3) | (uint64_t)((hd[st] >> (3u * oct)) & 7u);
                st = (unsigned)(hn[st] >> (5u * oct)) & 31u;
            } else {
                kk = (kk << 3) | (uint64_t)oct;
            }
        }
        k[w] = kk;
    }
    key_hi[i] = k[0];
    key_lo[i] = k[1];
    idx[i] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------
// K4: finish the order.  The radix sort only looks at the top `sort_bits` bits of the upper key word
// (13 levels by default: 5 digit passes instead of 8); bodies that agree on those bits form a run - none or
// pairs in the bench systems, thousands inside one cell once an escaper has inflated the root cube - and
// are put in order of their full 126-bit key here.  One thread per body; a body inside a run finds the
// run's ends by galloping + binary search on the sorted prefixes and its place by counting the run's
// members that sort before it: L reads per member, all members in parallel, so a run of 10^5 bodies costs
// about a millisecond (a one-thread insertion sort of it took minutes).  Equal keys keep the input order
// (the sort is stable and idx ascends), like the reference's insertion order.  Writes the final
// permutation and the fully sorted upper words; the longest run is recorded so that the host can widen
// the sorted prefix for the following steps.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tiefix(const uint64_t *__restrict__ hi_s, const uint64_t *__restrict__ key_lo,
                                                   const uint32_t *__restrict__ perm, int shift,
                                                   uint32_t *__restrict__ perm_out, uint64_t *__restrict__ hi_out, int64_t n,
                                                   TreeInfo *info, const int32_t *__restrict__ dead_rank = nullptr,
                                                   int64_t n_dead = 0) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const uint64_t h = hi_s[r], hp = h >> shift;
    const uint32_t p = perm[r];
    if (dead_rank && h == ~0ull) {
        // owner mode: the row of a body that has left (all-ones key, larger than any real one): the n_dead of them
        // take the last n_dead places, in row order - no run to search, however many there are
        const int64_t slot = n - n_dead + dead_rank[p];
        perm_out[slot] = p;
        hi_out[slot] = h;
        return;
    }
    const bool tie = (r + 1 < n && (hi_s[r + 1] >> shift) == hp) || (r > 0 && (hi_s[r - 1] >> shift) == hp);
    if (!tie) {
        perm_out[r] = p;
        hi_out[r] = h;
        return;
    }
    // run = [s, e): first / one past the last rank with this prefix
    int64_t lo_ok = r, lo_bad, step = 1;
    for (;;) {  // backwards
        const int64_t t = r - step;
        if (t < 0) { lo_bad = -1; break; }
        if ((hi_s[t] >> shift) == hp) { lo_ok = t; step <<= 1; } else { lo_bad = t; break; }
    }
    while (lo_ok - lo_bad > 1) {
        const int64_t mid = lo_bad + ((lo_ok - lo_bad) >> 1);
        if ((hi_s[mid] >> shift) == hp) lo_ok = mid; else lo_bad = mid;
    }
    int64_t hi_ok = r, hi_bad;
    step = 1;
    for (;;) {  // forwards
        const int64_t t = r + step;
        if (t >= n) { hi_bad = n; break; }
        if ((hi_s[t] >> shift) == hp) { hi_ok = t; step <<= 1; } else { hi_bad = t; break; }
    }
    while (hi_bad - hi_ok > 1) {
        const int64_t mid = hi_ok + ((hi_bad - hi_ok) >> 1);
        if ((hi_s[mid] >> shift) == hp) hi_ok = mid; else hi_bad = mid;
    }
    const int64_t s = lo_ok, e = hi_bad;
    if (r == s && e - s > 64 && !(dead_rank && hi_s[e - 1] == ~0ull)) atomicMax(&info->max_run, (int)(e - s < 0x7fffffff ? e - s : 0x7fffffff));
    const uint64_t la = key_lo[p];
    int64_t before = 0;
    for (int64_t j = s; j < e; j++) {
        const uint64_t hj = hi_s[j];
        const uint32_t pj = perm[j];
        if (hj != h) {
            before += hj < h ? 1 : 0;
        } else {
            const uint64_t lj = key_lo[pj];
            before += (lj < la || (lj == la && pj < p)) ? 1 : 0;
        }
    }
    perm_out[s + before] = p;
    hi_out[s + before] = h;
}

// ---------------------------------------------------------------------------------------
// K5 - K7 [r3]: ONE pass puts the bodies in key order and lays the ground for every cell's mass and centre of mass.
// A workgroup owns a tile (SUB-TILE of the prefix sums) of kScanTile = 2048 sorted ranks, swept in 8 rounds of 256:
//   * gather through the sort permutation: fp32 {x,y,z,G m} (walk / leaf data), the low key word;
//   * from the sorted keys of the two neighbours
//       delta[r] = common prefix digits of sorted bodies r and r+1 (delta[N-1] = -1),
//       cnt[r]   = number of internal cells whose first body is r = max(0, delta[r] - delta[r-1]);
//   * exclusive prefix sums INSIDE the tile (plain float64, straight from the float64 state):
//       S[r]    = sum over the sub-tile's bodies before r of {G m, G m x, G m y, G m z}     (32 bytes)
//       PexL[r] = the same for cnt
//     and the sub-tile's totals (sub_tot / sub_cnt).
// k_scan_subtiles then turns the totals into exclusive prefixes over the sub-tiles (T: double-double, subPex).
// A cell's moments are   (T[sub(e)] - T[sub(r)])  +  (S[e] - S[r])   for its body range [r, e):
// the first difference is exact to 1e-32 (double-double), the second is between sums of at most 2047 terms, so a
// cell's centre of mass is good to ~1e-13 of the coordinate wherever the cell sits in the array.  (A plain float64
// running sum over 10^6 bodies loses 1e-8, the size of the opening-test ties K9 re-decides in float64; round 2
// carried double-double sums through a three-phase scan of 64-byte records instead - 1.0 GB more traffic at 10 M
// bodies and three more kernels.)  Pre-order index helpers: pex_at(r) = subPex[r / 256] + PexL[r].
// ---------------------------------------------------------------------------------------
constexpr int kScanItems = 8;                      // rounds (sub-tiles) per workgroup
constexpr int kScanTile = kBlock * kScanItems;     // 2048 ranks per workgroup
constexpr int kSubShift = 11;                      // sub-tile = the workgroup's 2048 ranks
static_assert(kScanTile == (1 << kSubShift), "a sub-tile is one workgroup's tile");

// double-double (unevaluated sum of two float64)
struct dd {
    double h, l;
};
__device__ __forceinline__ dd dd_add(const dd &a, const dd &b) {
    const double s = a.h + b.h;
    const double bb = s - a.h;
    double e = (a.h - (s - bb)) + (b.h - bb);
    e += a.l + b.l;
    const double h = s + e;
    return dd{h, e - (h - s)};
}
__device__ __forceinline__ double dd_diff(const dd &a, const dd &b) {  // a - b, rounded once
    const dd d = dd_add(a, dd{-b.h, -b.l});
    return d.h + d.l;
}

__device__ __forceinline__ int64_t pex_at(const int32_t *__restrict__ PexL, const int32_t *__restrict__ subPex, int64_t r) {
    return (int64_t)subPex[r >> kSubShift] + PexL[r];
}

struct Mom4 {
    double m, x, y, z;
    int c;
};
__device__ __forceinline__ Mom4 m4_add(const Mom4 &a, const Mom4 &b) { return Mom4{a.m + b.m, a.x + b.x, a.y + b.y, a.z + b.z, a.c + b.c}; }
__device__ __forceinline__ Mom4 m4_shfl_up(const Mom4 &v, int d) {
    return Mom4{__shfl_up(v.m, d), __shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.c, d)};
}

// lane exchange inside a row of 16 lanes as a DPP operand (CTRL: 0xB1 / 0x4E quad_perm [1,0,3,2] / [2,3,0,1], 0x141
// row_half_mirror, 0x140 row_mirror)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float add_f(float a, float b) { return a + b; }

__global__ __launch_bounds__(kBlock) void k_gather_scan(Bodies cur, const uint32_t *__restrict__ perm,
                                                        const uint64_t *__restrict__ hi_s, const uint64_t *__restrict__ key_lo,
                                                        int64_t n, double G, float4 *__restrict__ posm_s,
                                                        double4 *__restrict__ p64_s /* may be null */, uint64_t *__restrict__ lo_s,
                                                        int32_t *__restrict__ delta, double4 *__restrict__ S,
                                                        int32_t *__restrict__ PexL, double4 *__restrict__ sub_tot,
                                                        int32_t *__restrict__ sub_cnt, unsigned char *__restrict__ wave_flag,
                                                        int32_t *__restrict__ sub_flag, float dens_thr, float edge_floor) {
    __shared__ Mom4 wtot[kBlock / 64];
    __shared__ int wflag[kBlock / 64];
    int nflag = 0;  // (thread 0) waves of this tile whose density asks for float64 forces
    __shared__ int dl[kBlock + 1];  // delta of the round's ranks, dl[0] = delta of the rank before the round
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    Mom4 carry{0.0, 0.0, 0.0, 0.0, 0};  // the rounds before this one (plain float64: at most 2047 terms)
#pragma unroll 1
    for (int k = 0; k < kScanItems; k++) {
        const int64_t r0 = (int64_t)blockIdx.x * kScanTile + (int64_t)k * kBlock;
        if (r0 > n) break;  // entry n (the totals' slot) is the last one anybody reads
        const int64_t r = r0 + t;
        Mom4 v{0.0, 0.0, 0.0, 0.0, 0};
        int d = -1;
        uint64_t h = 0, l = 0;
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (r < n) {
            const uint32_t j = perm[r];
            const double x = cur.x[j], y = cur.y[j], z = cur.z[j], gm = G * cur.m[j];
            fx = (float)x; fy = (float)y; fz = (float)z;
            posm_s[r] = make_float4((float)x, (float)y, (float)z, (float)gm);
            if (p64_s) p64_s[r] = make_double4(x, y, z, gm);
            h = hi_s[r];
            l = key_lo[j];
            lo_s[r] = l;
            if (r + 1 < n) {
                const uint64_t hn = hi_s[r + 1];
                d = cpl_digits(h, l, hn, hn == h ? key_lo[perm[r + 1]] : 0ull);
            }
            delta[r] = d;
            v = Mom4{gm, gm * x, gm * y, gm * z, 0};
        }
        if (wave_flag) {
            // [r3] force precision "auto", decided here where the sorted bodies pass through registers: G rho of the
            // DENSEST quarter of the wave (16 key-adjacent bodies: sum of G m over the volume of their bounding box, every
            // edge at least one softening length) against tau / dt^2.  Not the whole wave's box: 64 consecutive bodies
            // of the key order can straddle a gap (the two disks of the collision preset, a cell boundary high up in the
            // tree) and their common box then says nothing about where they sit.
            const bool have = r < n;
            const float big = 3.0e38f;
            float lox = have ? fx : big, hix = have ? fx : -big, loy = have ? fy : big, hiy = have ? fy : -big;
            float loz = have ? fz : big, hiz = have ? fz : -big, gsum = have ? (float)v.m : 0.f;
            // butterflies inside a row of 16 lanes as DPP operands (quad swaps, half-row mirror, row mirror: min, max and
            // the sum are symmetric) - one vector instruction each, where __shfl_xor is an LDS round trip (28 of them per
            // round cost 0.15 ms at 10 M bodies)
#define NBMI_ROW16(V, OP)                      \
    V = OP(V, dpp_f<0xB1>(V));                 \
    V = OP(V, dpp_f<0x4E>(V));                 \
    V = OP(V, dpp_f<0x141>(V));                \
    V = OP(V, dpp_f<0x140>(V));
            NBMI_ROW16(lox, fminf) NBMI_ROW16(hix, fmaxf) NBMI_ROW16(loy, fminf) NBMI_ROW16(hiy, fmaxf)
            NBMI_ROW16(loz, fminf) NBMI_ROW16(hiz, fmaxf) NBMI_ROW16(gsum, add_f)
#undef NBMI_ROW16
            const float vol = fmaxf(hix - lox, edge_floor) * fmaxf(hiy - loy, edge_floor) * fmaxf(hiz - loz, edge_floor);
            float dens = gsum > 0.f ? gsum / vol : 0.f;
            dens = fmaxf(dens, __shfl_xor(dens, 16));
            dens = fmaxf(dens, __shfl_xor(dens, 32));
            const int flag = dens > dens_thr ? 1 : 0;
            if (lane == 0) {
                if (r0 + 64 * w < n) wave_flag[(r0 >> 6) + w] = (unsigned char)flag;
                wflag[w] = (r0 + 64 * w < n) ? flag : 0;
            }
        }
        dl[t + 1] = d;
        if (t == 0) {
            int dp = -1;
            if (r > 0 && r < n) {
                const uint64_t hp = hi_s[r - 1];
                dp = cpl_digits(hp, hp == h ? key_lo[perm[r - 1]] : 0ull, h, l);
            }
            dl[0] = dp;
        }
        __syncthreads();
        if (r < n) {
            const int dp = dl[t];
            v.c = d > dp ? d - dp : 0;
        }
        // inclusive scan over the 256 ranks of the round
        Mom4 inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const Mom4 u = m4_shfl_up(inc, o);
            if (lane >= o) inc = m4_add(u, inc);
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        if (wave_flag && t == 0) nflag += wflag[0] + wflag[1] + wflag[2] + wflag[3];
        Mom4 off = carry, tot{0.0, 0.0, 0.0, 0.0, 0};
#pragma unroll
        for (int q = 0; q < kBlock / 64; q++) {
            if (q < w) off = m4_add(off, wtot[q]);
            tot = m4_add(tot, wtot[q]);
        }
        inc = m4_add(off, inc);
        carry = m4_add(carry, tot);
        if (r <= n) {
            // exclusive = inclusive - own (own is exactly representable in the sum only for cnt; for the moments take
            // the neighbour's inclusive value instead of subtracting)
            Mom4 ex = m4_shfl_up(inc, 1);
            if (lane == 0) ex = off;
            S[r] = make_double4(ex.m, ex.x, ex.y, ex.z);
            PexL[r] = ex.c;
        }
        __syncthreads();  // wtot / dl are reused by the next round
    }
    if (t == 0) {
        sub_tot[blockIdx.x] = make_double4(carry.m, carry.x, carry.y, carry.z);
        sub_cnt[blockIdx.x] = carry.c;
        if (wave_flag) sub_flag[blockIdx.x] = nflag;
    }
}

// Exclusive prefixes over the sub-tile totals: T (double-double moments, 64 bytes) and subPex.  One workgroup of
// 1024 threads: a thread adds up a contiguous chunk, the chunk sums are scanned across the workgroup, a second
// sweep writes the prefixes.  (39 k sub-tiles at 10 M bodies: 1.6 MB in, 2.7 MB out.)
constexpr int kSubScanThreads = 1024;
struct SubVal {
    dd m, x, y, z;
    long long c;
};
__device__ __forceinline__ SubVal sub_add(const SubVal &a, const SubVal &b) {
    return SubVal{dd_add(a.m, b.m), dd_add(a.x, b.x), dd_add(a.y, b.y), dd_add(a.z, b.z), a.c + b.c};
}
__device__ __forceinline__ SubVal sub_zero() { return SubVal{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, 0}; }
__device__ __forceinline__ dd dd_shfl_up(const dd &v, int d) { return dd{__shfl_up(v.h, d), __shfl_up(v.l, d)}; }
__device__ __forceinline__ SubVal sub_shfl_up(const SubVal &v, int d) {
    return SubVal{dd_shfl_up(v.m, d), dd_shfl_up(v.x, d), dd_shfl_up(v.y, d), dd_shfl_up(v.z, d), __shfl_up(v.c, d)};
}
// the system-wide half of force precision "auto" (TreeInfo::force_all64): enter above a third of the waves, leave below a quarter
constexpr int kAll64Enter = 333, kAll64Leave = 250;
// (thresholds in per mille of the waves: kAll64Enter / kAll64Leave; NBMI_ALL64_ENTER / NBMI_ALL64_LEAVE for experiments)
__host__ __device__ inline int all64_rule(int prev, long long ask, long long waves, int enter_pm, int leave_pm) {
    return prev ? (1000 * ask >= (long long)leave_pm * waves ? 1 : 0) : (1000 * ask > (long long)enter_pm * waves ? 1 : 0);
}
__global__ void k_set_all64(TreeInfo *info, int v) { info->force_all64 = v; }

__global__ __launch_bounds__(kSubScanThreads) void k_scan_subtiles(const double4 *__restrict__ sub_tot,
                                                                   const int32_t *__restrict__ sub_cnt, int64_t nsub,
                                                                   Moment *__restrict__ T, int32_t *__restrict__ subPex,
                                                                   const int32_t *__restrict__ sub_flag, int64_t nwaves,
                                                                   int enter_pm, int leave_pm, TreeInfo *info) {
    __shared__ SubVal wsum[kSubScanThreads / 64];
    __shared__ long long fsum[kSubScanThreads / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t chunk = (nsub + kSubScanThreads - 1) / kSubScanThreads;
    const int64_t b = (int64_t)t * chunk, e = b + chunk < nsub ? b + chunk : nsub;
    SubVal acc = sub_zero();
    long long fl = 0;
    for (int64_t i = b; i < e; i++) {
        const double4 q = sub_tot[i];
        acc = sub_add(acc, SubVal{{q.x, 0.0}, {q.y, 0.0}, {q.z, 0.0}, {q.w, 0.0}, (long long)sub_cnt[i]});
        if (sub_flag) fl += sub_flag[i];
    }
    if (sub_flag) {
        // [r3, thresholds r4] force precision "auto": while a large part of the waves ask for float64 (all64_rule: entered
        // above a third, left below a quarter), every wave gets it - where much of the system needs float64, the sparse rest
        // interacts with the same massive cells (10 M collision at dt 0.25: the 5 % of the waves below any density threshold
        // end at 3.8e-4 after 50 steps in fp32, at 1e-10 in float64; at the 1 M galaxy a quarter of the waves ask at first and
        // the fp32 rest stays at 2e-6 after 100 steps)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) fl += __shfl_xor(fl, o);
        if (lane == 0) fsum[w] = fl;
    }
    SubVal inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const SubVal u = sub_shfl_up(inc, o);
        if (lane >= o) inc = sub_add(u, inc);
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (sub_flag && t == 0) {
        long long tot = 0;
        for (int q = 0; q < kSubScanThreads / 64; q++) tot += fsum[q];
        info->ask_waves = (int)tot;
        info->n_waves = (int)nwaves;
        info->force_all64 = all64_rule(info->force_all64, tot, nwaves, enter_pm, leave_pm);
    }
    SubVal off = sub_zero();
    for (int q = 0; q < w; q++) off = sub_add(off, wsum[q]);
    SubVal run = sub_shfl_up(inc, 1);
    if (lane == 0) run = sub_zero();
    run = sub_add(off, run);  // exclusive prefix of this thread's chunk
    for (int64_t i = b; i < e; i++) {
        T[i] = Moment{run.m.h, run.m.l, run.x.h, run.x.l, run.y.h, run.y.l, run.z.h, run.z.l};
        subPex[i] = (int32_t)run.c;
        const double4 q = sub_tot[i];
        run = sub_add(run, SubVal{{q.x, 0.0}, {q.y, 0.0}, {q.z, 0.0}, {q.w, 0.0}, (long long)sub_cnt[i]});
    }
}

// Half width K (ulps of d^2) of the opening test's uncertainty band; derivation at K9.
__device__ __forceinline__ unsigned band_half_ulps(double maxabs, double eps) {
    double k = 8.0 + 7.0 * maxabs / eps;  // eps == 0: inf
    if (!(k < 2097152.0)) k = 2097152.0;  // cap 2^21 (band 2^22 ulps = a factor 1.5 ... 2 in d^2)
    return (unsigned)k + 1u;
}

// ---------------------------------------------------------------------------------------
// K8 [r3]: emit the nodes in DFS pre-order, one workgroup per tile of TILE sorted ranks (kEmitTile / kEmitTileSmall).
// Pre-order index of a node "started" by sorted body r (dp = delta[r-1], d = delta[r], cnt = max(0, d - dp)):
//   internal cell k of r (level dp+1+k, k < cnt) -> r + pex(r) + k,     leaf of r -> r + pex(r) + cnt.
// A cell (r, lev) holds the bodies [r, e), e - 1 = the first j >= r with delta[j] < lev ("nearest smaller value to
// the right").  Round 2 found e with a gallop + binary search on the sorted KEYS in global memory, one thread per
// cell: ~2 log2(size) dependent 16-byte probes each, 527 us at 10 M bodies (64 % of the wave cycles parked at
// s_waitcnt).  Now the tile's delta values sit in LDS as bytes under a min-tree (heap layout); the query "first
// j >= r with delta[j] < lev" climbs from leaf r to the first subtree on its right whose minimum is small enough
// and descends to its leftmost such leaf - a handful of LDS byte reads for the small cells that make up the
// bulk.  Only cells that reach beyond their tile (<= depth of the tree per tile boundary) fall back to the key
// search, started at the tile's end.  The same workgroup writes the leaves and lists the tile's cells in LDS
// (chunks of kCellChunk), so the cell list never travels through global memory.
// Mass / centre of mass of [r, e): see K5-K7.
// ---------------------------------------------------------------------------------------
constexpr int kEmitTile = 512, kEmitTileSmall = 256;
constexpr int64_t kEmitSmallBodies = 262144;  // up to here the small tile (>= 1024 workgroups from 262 k bodies on either way)

// moments (G m, G m x, G m y, G m z summed) of the bodies at sorted ranks [r, e): in-tile prefix differences, plus
// the double-double tile prefixes where the range crosses tiles
__device__ __forceinline__ void range_moments(const double4 *__restrict__ S, const Moment *__restrict__ T, int64_t r, int64_t e,
                                              double &M, double &mx, double &my, double &mz) {
    const double4 s0 = S[r], s1 = S[e];
    M = s1.x - s0.x; mx = s1.y - s0.y; my = s1.z - s0.z; mz = s1.w - s0.w;
    const int64_t ur = r >> kSubShift, ue = e >> kSubShift;
    if (ue != ur) {
        const Moment a = T[ur], b = T[ue];
        M += dd_diff(dd{b.m, b.ml}, dd{a.m, a.ml});
        mx += dd_diff(dd{b.x, b.xl}, dd{a.x, a.xl});
        my += dd_diff(dd{b.y, b.yl}, dd{a.y, a.yl});
        mz += dd_diff(dd{b.z, b.zl}, dd{a.z, a.zl});
    }
}

__device__ __forceinline__ void write_sentinel(Node *__restrict__ nodes, int32_t *__restrict__ node_ref, int64_t total,
                                               int64_t link_base = 0) {
    Node sn;
    sn.cx = sn.cy = sn.cz = 1.0e30f;
    sn.gm = 0.f; sn.s2t = 0.f;
    sn.next_off = (unsigned)(total + link_base) * kNodeBytes;
    nodes[total] = sn;
    if (node_ref) node_ref[total] = -1;
}

// [r4] TILE is a template parameter.  Round 3 used 2 048 ranks per workgroup throughout; measured this round
// (profiles/r04_emit_tile_sweep.txt): tree phase at 1 M bodies 0.147 ms with 2 048, 0.134 / 0.133 / 0.135 with 1 024 / 512 /
// 256; at 10 M 1.32 / 1.18 / 1.16 / 1.26 ms; at 10 k bodies five workgroups of 2 048 ranks took 58 us (the tile's serial
// phases, an empty chip), forty of 256 a quarter of that.  512 ships, 256 up to 262 k bodies.  Only the decomposition
// changes (a cell that reaches beyond its tile is found by the key search either way): the nodes written are the same.
template <int TILE>
__global__ __launch_bounds__(kBlock) void k_emit_tile(const int32_t *__restrict__ delta, const int32_t *__restrict__ PexL,
                                                      const int32_t *__restrict__ subPex, const double4 *__restrict__ S,
                                                      const Moment *__restrict__ T, const float4 *__restrict__ posm_s,
                                                      const double4 *__restrict__ p64_s, const uint64_t *__restrict__ hi_s,
                                                      const uint64_t *__restrict__ lo_s, int64_t n, int64_t capacity,
                                                      double eps, double inv_theta2, Node *__restrict__ nodes,
                                                      Node64 *__restrict__ nodes64, uint8_t *__restrict__ node_level,
                                                      int32_t *__restrict__ node_ref, double4 *__restrict__ diag64,
                                                      NodeD *__restrict__ nodesd /* may be null */, Bodies cur,
                                                      const uint32_t *__restrict__ perm, double G, TreeInfo *info,
                                                      int64_t link_base /* owner mode: the arrays passed in begin at this row of
                                                                           the walk array, and the links count from ITS start */) {
    __shared__ uint8_t tree[2 * TILE];   // heap: tree[TILE + i] = delta[base + i] + 1, inner nodes = min of children
    __shared__ uint8_t dprev;                 // delta[base - 1] + 1
    __shared__ uint32_t cells[(2 * TILE)];    // node slot -> (local body index << 6) | k (k-th node of that body)
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * TILE;
    const int64_t total = n + pex_at(PexL, subPex, n);
    if (blockIdx.x == 0 && t == 0) {
        info->num_nodes = total;
        info->walk_nodes = total;
        info->band2 = 2u * band_half_ulps(__longlong_as_double((long long)info->maxabs_bits), eps);
        if (total + 1 > capacity) {  // + 1: the sentinel
            info->error = 1;
            if (info->sticky_error == 0) {
                info->sticky_error = 1;
                info->sticky_nodes = total;
            }
        } else {
            write_sentinel(nodes, node_ref, total, link_base);
            if (nodesd) nodesd[total] = NodeD{1.0e30, 1.0e30, 1.0e30, 0.0, 0.0f, (unsigned)(total + link_base) * kNodeDBytes};
        }
    }
    if (total + 1 > capacity) return;
    // delta of the tile as bytes (ranks >= n - 1 count as -1: nothing reaches across the end of the array)
    for (int i = t; i < TILE; i += kBlock) {
        const int64_t r = base + i;
        tree[TILE + i] = (uint8_t)((r < n ? delta[r] : -1) + 1);
    }
    if (t == 0) dprev = (uint8_t)((base > 0 ? delta[base - 1] : -1) + 1);
    __syncthreads();
    for (int width = TILE / 2; width >= 1; width >>= 1) {
        for (int i = t; i < width; i += kBlock) {
            const uint8_t a = tree[2 * (width + i)], b = tree[2 * (width + i) + 1];
            tree[width + i] = a < b ? a : b;
        }
        __syncthreads();
    }
    const int64_t q_tile = pex_at(PexL, subPex, base < n ? base : n);
    const int64_t tile_end = base + TILE < n ? base + TILE : n;
    const int64_t ncell = pex_at(PexL, subPex, tile_end) - q_tile;  // cells started inside this tile
    const int64_t nnode = (tile_end - base) + ncell;                // the tile's nodes: one contiguous run of indices
    const int64_t idx0 = base + q_tile;                             // ... starting here
    const double bounds = info->bounds;
    const unsigned bandk = band_half_ulps(__longlong_as_double((long long)info->maxabs_bits), eps);
    // The tile's nodes are written in INDEX order, consecutive lanes = consecutive nodes (a wave's stores of the
    // 24 / 40 / 32-byte records then cover one contiguous stretch of each array; a thread per body / per cell wrote
    // every record into a different cache line: 2.3 TB/s of mostly partial-line traffic at 10 M bodies).  Which
    // (body, k) a node index belongs to comes from a table in LDS that the bodies fill, (2 * TILE) slots at a time:
    // local node slot of body i's k-th node (its cnt cells, then its leaf) = i + (pex(r) - q_tile) + k.
    for (int64_t c0 = 0; c0 < nnode; c0 += (2 * TILE)) {
        for (int i = t; i < TILE; i += kBlock) {
            const int64_t r = base + i;
            if (r >= n) break;
            const int d = (int)tree[TILE + i] - 1;
            const int dp = (int)(i > 0 ? tree[TILE + i - 1] : dprev) - 1;
            const int cnt = d > dp ? d - dp : 0;
            const int64_t s0 = i + (pex_at(PexL, subPex, r) - q_tile) - c0;
            for (int k = 0; k <= cnt; k++) {
                const int64_t c = s0 + k;
                if (c >= 0 && c < (2 * TILE)) cells[c] = ((uint32_t)i << 6) | (uint32_t)k;
            }
        }
        __syncthreads();
        const int64_t here = nnode - c0 < (2 * TILE) ? nnode - c0 : (2 * TILE);
        for (int64_t c = t; c < here; c += kBlock) {
            const uint32_t cl = cells[c];
            const int i = (int)(cl >> 6), k = (int)(cl & 63u);
            const int64_t r = base + i;
            const int64_t idx = idx0 + c0 + c;
            const int d = (int)tree[TILE + i] - 1;
            const int dp = (int)(i > 0 ? tree[TILE + i - 1] : dprev) - 1;
            const int cnt = d > dp ? d - dp : 0;
            if (k == cnt) {  // the body's leaf
                const float4 p = posm_s[r];
                Node lf;
                lf.cx = p.x; lf.cy = p.y; lf.cz = p.z; lf.gm = p.w;
                lf.s2t = 0.0f;
                lf.next_off = (unsigned)(idx + 1 + link_base) * kNodeBytes;
                nodes[idx] = lf;
                if (diag64) diag64[idx] = p64_s ? p64_s[r] : make_double4((double)p.x, (double)p.y, (double)p.z, (double)p.w);
                if (nodesd) {  // the body as the float64 state has it (nearly sequential: the state is in last step's key order)
                    const uint32_t j = perm[r];
                    nodesd[idx] = NodeD{cur.x[j], cur.y[j], cur.z[j], G * cur.m[j], 0.0f, (unsigned)(idx + 1 + link_base) * kNodeDBytes};
                }
                if (node_ref) {  // queries and the owner-mode kernels only: a plain step does not pay for them
                    node_ref[idx] = (int32_t)r;
                    node_level[idx] = (uint8_t)((d > dp ? d : dp) + 1);
                }
                continue;
            }
            const int lev = dp + 1 + k;
            // first local index j >= i with delta[j] + 1 <= lev
            unsigned h = (unsigned)(TILE + i);
            bool found = false;
            for (;;) {
                if ((int)tree[h] <= lev) { found = true; break; }
                while ((h & 1u) && h > 1u) h >>= 1;
                if (h == 1u) break;
                h += 1u;
            }
            int64_t e;
            if (found) {
                while (h < (unsigned)TILE) {
                    h <<= 1;
                    if ((int)tree[h] > lev) h += 1u;
                }
                e = base + (int64_t)(h - (unsigned)TILE) + 1;
            } else {
                // the cell reaches beyond the tile: gallop + binary search on the sorted keys from the tile's end on
                const uint64_t kh = hi_s[r], kl = lo_s[r];
                int64_t ok = base + TILE - 1, bad, step = 1;
                for (;;) {
                    const int64_t u = ok + step;
                    if (u >= n) { bad = n; break; }
                    if (cpl_digits(kh, kl, hi_s[u], lo_s[u]) >= lev) { ok = u; step <<= 1; }
                    else { bad = u; break; }
                }
                while (bad - ok > 1) {
                    const int64_t mid = ok + ((bad - ok) >> 1);
                    if (cpl_digits(kh, kl, hi_s[mid], lo_s[mid]) >= lev) ok = mid; else bad = mid;
                }
                e = bad;
            }
            double M, mx, my, mz;
            range_moments(S, T, r, e, M, mx, my, mz);
            double cx = 0.0, cy = 0.0, cz = 0.0;
            if (M > 0.0) { cx = mx / M; cy = my / M; cz = mz / M; }
            const double size = ldexp(bounds, 1 - lev);  // 2 * bounds / 2^lev, exact
            Node nd;
            nd.cx = (float)cx; nd.cy = (float)cy; nd.cz = (float)cz;
            nd.gm = (float)M;  // the moments are sums of G*m
            // upper edge of the uncertainty band: bits((2 hs)^2 / theta^2) + K  (theta == 0: +inf, never accepted)
            const float s2t = (float)(size * size * inv_theta2);
            nd.s2t = __int_as_float(__float_as_int(s2t) + (int)bandk);
            const int64_t nxt = e + pex_at(PexL, subPex, e) + link_base;
            nd.next_off = (unsigned)nxt * kNodeBytes;
            nodes[idx] = nd;
            // [r3] the float64 loop tests a correctly rounded d^2: its uncertainty band is 2 ulps on either side of the
            // threshold, not the K the fp32 loop needs for its rounded coordinates (re-decisions in float64 waves: 1 000 x fewer)
            if (nodesd) nodesd[idx] = NodeD{cx, cy, cz, M, __int_as_float(__float_as_int(s2t) + (int)kBand64), (unsigned)nxt * kNodeDBytes};
            nodes64[idx] = Node64{cx, cy, cz, ldexp(bounds, -lev)};
            if (diag64) diag64[idx] = make_double4(cx, cy, cz, M);
            if (node_ref) {
                node_ref[idx] = (int32_t)r;
                node_level[idx] = (uint8_t)lev;
            }
        }
        __syncthreads();
    }
}

// deepest leaf level = max(delta) + 1; only run when nbmi_tree_stats() asks (same-address
// atomics cost ~11 ns each: one per block here, never in the per-step path)
__global__ __launch_bounds__(kBlock) void k_max_level(const int32_t *__restrict__ delta, int64_t n, TreeInfo *info) {
    __shared__ int red[kBlock / 64];
    int m = -1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int d = delta[i];
        m = d > m ? d : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = red[w] > m ? red[w] : m;
        atomicMax(&info->max_level, m + 1);
    }
}

// ---------------------------------------------------------------------------------------
// K8b: child table (stack-walk prototype only).  Row i of an internal cell holds the byte offsets of its (up to
// 8) children in pre-order (0 = no more children; offset 0 is the root and never a child).  One thread per node
// row, after all nodes exist: first child = next row, siblings via the skip links.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_child_table(const Node *__restrict__ nodes, const TreeInfo *__restrict__ info,
                                                        int64_t capacity, uint32_t *__restrict__ child_tab) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (info->error || idx >= info->num_nodes || idx >= capacity) return;
    if (__float_as_int(nodes[idx].s2t) == 0) return;  // a leaf
    const unsigned end = nodes[idx].next_off;
    unsigned c = (unsigned)(idx + 1) * kNodeBytes;
    uint32_t t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        t[k] = c < end ? c : 0u;
        if (c < end) c = reinterpret_cast<const Node *>(reinterpret_cast<const char *>(nodes) + c)->next_off;
    }
    uint4 *row = reinterpret_cast<uint4 *>(child_tab + 8 * idx);
    row[0] = make_uint4(t[0], t[1], t[2], t[3]);
    row[1] = make_uint4(t[4], t[5], t[6], t[7]);
}

// ---------------------------------------------------------------------------------------
// K9: the walk.  One wave64 = 64 consecutive sorted bodies; wave-uniform cursor over the
// pre-order node array (scalar loads of the 24-byte record); per lane the reference's test
// (simulation.py:245-274):
//     d = com - p; dist_sq = |d|^2 + eps^2;
//     accept if leaf or 2 hs / dist < theta          [== (2hs)^2/theta^2 < dist_sq; leaves carry 0]
//     accepted && mass>0 && dist_sq > eps^2  ->  a += G m d / dist^3
// The reference's explicit "skip my own leaf" needs no instruction here: the own leaf has d = 0
// exactly, so its term is 0 (eps > 0) or fails the dist_sq > eps^2 guard (kGuard, eps == 0).
// `resume` = first node index at which the lane takes part again (it accepted an ancestor of
// everything before that).  The cursor moves to the next node in memory if any lane opens the node, else to
// next_off.  Fused epilogue: v = (v + a dt) * damping; x += v dt  (simulation.py:291-305),
// written at the body's NEW sorted rank (state re-ordering is fused into this kernel).
//
// Opening-test ties.  The test runs in fp32 on fp32-rounded positions, the reference's in float64; a
// (body, node) pair whose d^2 lies within the rounding error of the threshold can be decided the other
// way, and such a body then carries a different (equally legitimate) Barnes-Hut approximation - enough
// to miss the 1e-4 position bound after 100 steps at 1 M bodies.  So the node stores the UPPER edge of
// an uncertainty band, bits(s2t) + K, the walk derives the lower edge bits(s2t) - K, and a lane whose
// d^2 falls between the two is re-decided exactly as the reference does it (simulation.py:252-258):
// float64 body position, float64 centre of mass (double-double prefix sums, see k_scan_*), sqrt and
// divide.  K (ulps of d^2) bounds the fp32 error: both positions are rounded by <= 2^-24 maxabs, so
// d^2 is off by <= 2 sqrt(3) d 2^-23 maxabs + a few ulps, i.e. by <= 6.93 maxabs / d + 4 ulps (one ulp
// is >= 2^-24 of d^2); the test only matters where d >= eps (smaller cells pass it through eps^2
// alone), hence K = 8 + 7 maxabs / eps (TreeInfo.band2 = 2 K; 6.5 k ulps = 4e-4 ... 8e-4 of d^2 for the
// 1 M-body galaxy).  One extra compare per visit; the float64 path runs for ~1.5e-4 of the lane visits.
// ---------------------------------------------------------------------------------------
struct WalkParams {
    int64_t rank_begin, rank_end;  // shard of sorted ranks handled by this launch

    float eps2;
    int xcd_chunk;  // see logical_block()
    int balance;    // 1: blocks are dealt to the XCDs by last step's measured times (WalkTable::xcd_bounds)
    int pair;       // one-wave walk with two cursors (the two halves of the array)
    double dt, damping;
    int curbuf;  // which of WalkTable.buf holds the current state
    int acc64;   // measurement (counted walk only, NBMI_ACC64=1): every visit's contribution summed in float64
    int force_prec;  // 0 = per wave by local density (see k_walk), 1 = fp32 pair forces everywhere, 2 = float64 everywhere
    float prec_tau;  // force_prec 0: a wave takes float64 when G rho dt^2 of its bodies exceeds this
    int prec;    // measurement (k_walk_diag, NBMI_PREC=<mode>): which visits compute their force in which arithmetic
    float near2; // k_walk_diag: "near" visits have fp32 dist_sq below this
};

// Per-handle constants the walk needs only rarely (float64 re-decision) or only at its end (the state
// pointers of the fused kick-drift).  They live in device memory, not in kernel arguments: arguments sit
// in SGPRs for the whole kernel, and with 16 state pointers among them the walk needed 92 SGPRs - over
// the 80 at which the hardware still admits eight 256-thread blocks per CU.
struct WalkTable {
    Bodies buf[2];
    const Node64 *n64;
    const NodeD *nodesd;  // float64 node records (null: the handle computes every force in fp32)
    const int *xcd_bounds;   // [9] logical walk blocks [b[x], b[x + 1]) belong to XCD x (balance mode)
    unsigned *wave_cycles;   // [4 per logical block] how long each wave of the last walk took (shader clocks)
    unsigned char *wave_flag;  // [one per wave] 1 = the wave's own density asked for float64 forces (auto mode)
    const int32_t *pex, *subpex;  // leaf of the body at sorted rank r = node r + pex_at(pex, subpex, r + 1)
    unsigned long long *maxabs_next;  // TreeInfo::maxabs_next
    double theta, eps2;
    long long own_base;  // owner mode: row of the walk array at which the handle's own tree begins (0 otherwise)
};

// where a lane finds its body's float64 position (only read on the re-decision path)
struct Body64 {
    const WalkTable *tab;
    int curbuf;
    uint32_t j;
};

// the reference's own test in float64 (simulation.py:249-258), operation for operation
__device__ __forceinline__ bool exact_take_idx(unsigned idx, const Body64 &b) {
    const WalkTable *t = b.tab;
    // (Measured and dropped [r3]: taking the centre of mass from the NodeD row and the half size from bounds / 2^level,
    // so that Node64 need not be written - 0.1 ms less build at 10 M bodies, but the walk lost 0.5 ms: a wave leaves the
    // loop for ~12 re-decisions per walk and each one waits for this function's dependent loads; one more level of them
    // is 4 % of the walk.)
    const Node64 c = t->n64[idx];
    const Bodies &cur = t->buf[b.curbuf];
    const double dx = c.cx - cur.x[b.j], dy = c.cy - cur.y[b.j], dz = c.cz - cur.z[b.j];
    const double dist_sq = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)), t->eps2);
    const double dist = sqrt(dist_sq);
    return (c.hs * 2.0) / dist < t->theta;
}
__device__ __forceinline__ bool exact_take(unsigned off, const Body64 &b) { return exact_take_idx(off / kNodeBytes, b); }

// One node visit (C++ form: counted / eps == 0 kernels, seek(), and the product walk's re-decision visits).
// `off` is the cursor as a byte offset into the node array; `resume` likewise.  Returns the next cursor.
template <bool kGuard>
__device__ __forceinline__ unsigned visit(const Node *__restrict__ nodes, unsigned off,
                                          float px, float py, float pz, const Body64 &b64, const WalkParams &P,
                                          unsigned band2, unsigned &resume, float &ax, float &ay, float &az,
                                          bool &active_out, bool &force_out, bool &jumped, bool &band_out) {
    off = __builtin_amdgcn_readfirstlane(off);
    const Node nd = *reinterpret_cast<const Node *>(reinterpret_cast<const char *>(nodes) + off);
    const float dx = nd.cx - px, dy = nd.cy - py, dz = nd.cz - pz;
    const float dist_sq = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, P.eps2)));
    const bool active = resume <= off;
    // non-negative floats: compare bit patterns as integers.  hi / lo = edges of the uncertainty band.
    const int d2b = __float_as_int(dist_sq), hi = __float_as_int(nd.s2t), lo = hi - (int)band2;
    bool geom = hi < d2b;
    const bool band = active && !geom && lo < d2b;
    if (band) geom = (hi == 0) || exact_take(off, b64);  // rare, divergent
    const unsigned long long m_active = __builtin_amdgcn_ballot_w64(active);
    const unsigned long long m_geom = __builtin_amdgcn_ballot_w64(geom);
    const bool take = active && geom;
    bool force = take;
    if (kGuard) force = take && (dist_sq > P.eps2);
    // same association as the hand-scheduled loop, (G m / d) (1 / d^2): a body's sums must not depend on
    // which of its visits went through this C++ form (that depends on the other bodies of its wave)
    const float inv = __builtin_amdgcn_rsqf(dist_sq);
    const float f = force ? (nd.gm * inv) * (inv * inv) : 0.f;
    ax = fmaf(dx, f, ax);
    ay = fmaf(dy, f, ay);
    az = fmaf(dz, f, az);
    resume = take ? nd.next_off : resume;
    const unsigned long long any_open = m_active & ~m_geom;
    active_out = active;
    force_out = take && (dist_sq > P.eps2);
    band_out = band;
    jumped = !any_open && nd.next_off != off + kNodeBytes;
    return any_open ? off + kNodeBytes : nd.next_off;
}

// Hand-scheduled walk loop for the product kernel (eps > 0, no counters): 16 VALU + 3 SALU + 2 SMEM
// instructions on a visit nobody opens (80 % of them).  s_load_dwordx4 + s_load_dwordx2 at an SGPR byte offset.
// The two deciding compares are v_cmpx: EXEC narrows to the lanes that take part in the visit, then to those
// that TAKE the node, so the force instructions and the `resume` update need no per-lane selects; s_andn2 of
// the two masks leaves "some lane opens" in SCC.  Nobody opens: the cursor follows next_off.  Somebody opens
// (20 % of the visits): a branch to an out-of-line block behind the loop (NBMI_OPEN_X), which holds everything
// only such a visit needs - the near-tie check (only an opener can lie inside the uncertainty band: one v_cmp
// of the openers against the lower band edge s2t - band2; a hit leaves the loop BEFORE the visit has changed
// anything - cursor, resume, sums - the caller performs that one visit in C++ with the float64 test and
// re-enters) and the step to the next node in memory - and comes back.  (Measured: walk 1.071 -> 1.053 ms at
// 1 M, 10.23 -> 10.06 ms at 10 M against the check on every visit.)  EXEC is all-ones again before the next
// visit (every launched wave is full; lanes without a body carry resume = ~0 and never take part).  The node
// record of the NEXT visit is requested as soon as its offset is known - before the nine force instructions of
// the current visit are issued - into the other of two SGPR banks (A = s[36:41], B = s[48:53]: cx cy cz gm s2t
// next_off), which takes those instructions' issue time out of the per-wave dependent chain.  The whole loop
// is one asm statement (4 visits per trip, banks A B A B) so that no compiler-generated code runs while a load
// is in flight; a self-looping sentinel node after the last one makes overshooting harmless.
#define NBMI_VISIT_X(OFF, RES, ACC, WAIT, CX, CY, CZ, GM, S2T, NXT, NEXTLO, NEXTHI, LOPEN, LJOIN, LSKIP) \
    "v_cmpx_ge_u32_e64 s[44:45], " OFF ", " RES "\n"           \
    WAIT                                                       \
    "v_sub_f32_e32 %[dx], " CX ", %[px]\n"                     \
    "v_sub_f32_e32 %[dy], " CY ", %[py]\n"                     \
    "v_sub_f32_e32 %[dz], " CZ ", %[pz]\n"                     \
    "v_fma_f32 %[d2], %[dx], %[dx], %[eps2]\n"                 \
    "v_fmac_f32_e32 %[d2], %[dy], %[dy]\n"                     \
    "v_fmac_f32_e32 %[d2], %[dz], %[dz]\n"                     \
    "v_cmpx_lt_i32_e64 s[46:47], " S2T ", %[d2]\n"             \
    "s_andn2_b64 s[56:57], s[44:45], s[46:47]\n"               \
    "s_cbranch_scc1 " LOPEN "f\n"                              \
    "s_mov_b32 " OFF ", " NXT "\n"                             \
    LJOIN ":\n"                                                \
    "s_load_dwordx4 " NEXTLO ", %[base], " OFF "\n"            \
    "s_load_dwordx2 " NEXTHI ", %[base], " OFF " offset:16\n"  \
    "v_rsq_f32_e32 %[inv], %[d2]\n"                            \
    "v_mov_b32_e32 " RES ", " NXT "\n"                         \
    "v_mul_f32_e32 %[f], " GM ", %[inv]\n"                     \
    "v_mul_f32_e32 %[t], %[inv], %[inv]\n"                     \
    "v_mul_f32_e32 %[f], %[f], %[t]\n"                         \
    "v_fmac_f32_e32 %[ax" ACC "], %[dx], %[f]\n"               \
    "v_fmac_f32_e32 %[ay" ACC "], %[dy], %[f]\n"               \
    "v_fmac_f32_e32 %[az" ACC "], %[dz], %[f]\n"               \
    "s_mov_b64 exec, -1\n"                                    \
    LSKIP ":\n"
// the out-of-line half of a visit that some lane opens (s[56:57] = openers, s[46:47] = takers; 20 % of the visits):
// only an opener can lie inside the uncertainty band, so the near-tie check lives here - one v_cmp of the openers
// against the lower band edge; a hit leaves the loop BEFORE the visit has changed anything, otherwise the cursor
// moves to the next node in memory and the visit goes on
#define NBMI_OPEN_X(OFF, S2T, NEXTLO, NEXTHI, LOPEN, LJOIN, LSKIP, EXITL)   \
    LOPEN ":\n"                                      \
    "s_sub_u32 s54, " S2T ", %[band2]\n"             \
    "s_mov_b64 exec, s[56:57]\n"                     \
    "v_cmp_lt_i32_e64 s[42:43], s54, %[d2]\n"        \
    "s_mov_b64 exec, s[46:47]\n"                     \
    "s_cmp_lg_u64 s[42:43], 0\n"                     \
    "s_cbranch_scc1 " EXITL "\n"                     \
    "s_add_u32 " OFF ", " OFF ", 24\n"               \
    "s_cmp_lg_u64 s[46:47], 0\n"                     \
    "s_cbranch_scc1 " LJOIN "b\n"                    \
    /* [r3] nobody takes the node (every lane that takes part opens it): request the next record and skip the */ \
    /* eight force instructions */                   \
    "s_load_dwordx4 " NEXTLO ", %[base], " OFF "\n"  \
    "s_load_dwordx2 " NEXTHI ", %[base], " OFF " offset:16\n" \
    "s_mov_b64 exec, -1\n"                           \
    "s_branch " LSKIP "b\n"
#define NBMI_WAIT "s_waitcnt lgkmcnt(0)\n"
#define NBMI_VISIT_A(LO, LJ, LS) \
    NBMI_VISIT_X("%[off]", "%[resume]", "", NBMI_WAIT, "s36", "s37", "s38", "s39", "s40", "s41", "s[48:51]", "s[52:53]", LO, LJ, LS)
#define NBMI_VISIT_B(LO, LJ, LS) \
    NBMI_VISIT_X("%[off]", "%[resume]", "", NBMI_WAIT, "s48", "s49", "s50", "s51", "s52", "s53", "s[36:39]", "s[40:41]", LO, LJ, LS)
#define NBMI_OPEN_A(LO, LJ, LS) NBMI_OPEN_X("%[off]", "s40", "s[48:51]", "s[52:53]", LO, LJ, LS, "7f")
#define NBMI_OPEN_B(LO, LJ, LS) NBMI_OPEN_X("%[off]", "s52", "s[36:39]", "s[40:41]", LO, LJ, LS, "7f")
// two cursors in one wave (walk_pair_asm): cursor 1 uses banks s[36:41] / s[48:53], cursor 2 uses
// s[60:65] / s[68:73]; the trip waits ONCE for both cursors' records
#define NBMI_VISIT_1A NBMI_VISIT_A
#define NBMI_VISIT_1B NBMI_VISIT_B
#define NBMI_VISIT_2A(LO, LJ, LS) \
    NBMI_VISIT_X("%[off2]", "%[resume2]", "2", "", "s60", "s61", "s62", "s63", "s64", "s65", "s[68:71]", "s[72:73]", LO, LJ, LS)
#define NBMI_VISIT_2B(LO, LJ, LS) \
    NBMI_VISIT_X("%[off2]", "%[resume2]", "2", "", "s68", "s69", "s70", "s71", "s72", "s73", "s[60:63]", "s[64:65]", LO, LJ, LS)
#define NBMI_OPEN_2A(LO, LJ, LS) NBMI_OPEN_X("%[off2]", "s64", "s[68:71]", "s[72:73]", LO, LJ, LS, "8f")
#define NBMI_OPEN_2B(LO, LJ, LS) NBMI_OPEN_X("%[off2]", "s72", "s[60:63]", "s[64:65]", LO, LJ, LS, "8f")
// Two-level sums: the loops add into fp32 accumulators (one instruction per component and visit); every few
// trips those are emptied into float64 sums (NBMI_FLUSH: convert, add, clear), so an fp32 running sum never grows
// beyond a dozen terms.  Measured at 1 M bodies against the float64 oracle, per-body relative acceleration error:
// fp32 running sums over the whole walk rms 5.4e-7 (median 4.1e-7), every contribution summed in float64 1.1e-7
// (5.4e-8) - the running sums were four fifths of the error.  The flush blocks sit behind the loops (reached by a
// branch once per NBMI_FLUSH_TRIPS trips).
#define NBMI_FLUSH(AX, AY, AZ)                      \
    "v_cvt_f64_f32_e32 %[t64], " AX "\n"            \
    "v_add_f64 %[sx], %[sx], %[t64]\n"              \
    "v_mov_b32_e32 " AX ", 0\n"                     \
    "v_cvt_f64_f32_e32 %[t64], " AY "\n"            \
    "v_add_f64 %[sy], %[sy], %[t64]\n"              \
    "v_mov_b32_e32 " AY ", 0\n"                     \
    "v_cvt_f64_f32_e32 %[t64], " AZ "\n"            \
    "v_add_f64 %[sz], %[sz], %[t64]\n"              \
    "v_mov_b32_e32 " AZ ", 0\n"
// at the loop-back point: count the trip down; on underflow empty the accumulators (label 40, returns to 41)
#define NBMI_TRIP_COUNT                 \
    "s_sub_u32 %[cnt], %[cnt], 1\n"     \
    "s_cbranch_scc1 40f\n"              \
    "41:\n"
#define NBMI_EXITS(OPENS, FLUSHES, TRIPS) \
    "s_branch 9f\n" OPENS               \
    "40:\n" FLUSHES                     \
    "s_mov_b32 %[cnt], " TRIPS "\n"     \
    "s_branch 41b\n"                    \
    "7:\n"                              \
    "s_mov_b64 exec, -1\n"              \
    "s_mov_b32 %[which], 1\n"           \
    "s_branch 9f\n"                     \
    "8:\n"                              \
    "s_mov_b64 exec, -1\n"              \
    "s_mov_b32 %[which], 2\n"           \
    "9:\n"                              \
    "s_waitcnt lgkmcnt(0)\n"
#define NBMI_CLOBBERS                                                                                               \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
        "s52", "s53", "s54", "s55", "s56", "s57", "s58", "vcc", "scc", "memory"

// walks from `off` until the cursor reaches `end` (the end of the array: 4 visits per loop test, the
// sentinel absorbs the overshoot) or a near-tie stops it (which = 1, cursor on the tied node)
__device__ __forceinline__ void walk4_asm(const Node *nodes, unsigned &off, unsigned end, float px, float py, float pz,
                                          float eps2, unsigned band2, unsigned &resume, float &ax, float &ay, float &az,
                                          double &sx, double &sy, double &sz, unsigned &which) {
    float dx, dy, dz, d2, inv, f, t;
    double t64;
    unsigned cnt;
    asm volatile("s_mov_b32 %[cnt], 3\n"
                 "s_load_dwordx4 s[36:39], %[base], %[off]\n"
                 "s_load_dwordx2 s[40:41], %[base], %[off] offset:16\n"
                 "1:\n" NBMI_VISIT_A("21", "31", "51") NBMI_VISIT_B("22", "32", "52") NBMI_VISIT_A("23", "33", "53") NBMI_VISIT_B("24", "34", "54") NBMI_TRIP_COUNT
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc1 1b\n"
                 NBMI_EXITS(NBMI_OPEN_A("21", "31", "51") NBMI_OPEN_B("22", "32", "52") NBMI_OPEN_A("23", "33", "53") NBMI_OPEN_B("24", "34", "54"),
                            NBMI_FLUSH("%[ax]", "%[ay]", "%[az]"), "3")
                 : [off] "+s"(off), [which] "+s"(which), [resume] "+v"(resume), [ax] "+v"(ax), [ay] "+v"(ay),
                   [az] "+v"(az), [sx] "+v"(sx), [sy] "+v"(sy), [sz] "+v"(sz), [t64] "=&v"(t64), [cnt] "=&s"(cnt),
                   [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [d2] "=&v"(d2), [inv] "=&v"(inv),
                   [f] "=&v"(f), [t] "=&v"(t)
                 : [base] "s"(nodes), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps2] "s"(eps2), [end] "s"(end),
                   [band2] "s"(band2)
                 : NBMI_CLOBBERS);
}

// one visit per loop test: a part must not step past its end (the next part starts there)
__device__ __forceinline__ void walk1_asm(const Node *nodes, unsigned &off, unsigned end, float px, float py, float pz,
                                          float eps2, unsigned band2, unsigned &resume, float &ax, float &ay, float &az,
                                          double &sx, double &sy, double &sz, unsigned &which) {
    float dx, dy, dz, d2, inv, f, t;
    double t64;
    unsigned cnt;
    asm volatile("s_mov_b32 %[cnt], 7\n"
                 "s_load_dwordx4 s[36:39], %[base], %[off]\n"
                 "s_load_dwordx2 s[40:41], %[base], %[off] offset:16\n"
                 "1:\n" NBMI_VISIT_A("21", "31", "51")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc0 9f\n" NBMI_VISIT_B("22", "32", "52") NBMI_TRIP_COUNT
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc1 1b\n"
                 NBMI_EXITS(NBMI_OPEN_A("21", "31", "51") NBMI_OPEN_B("22", "32", "52"), NBMI_FLUSH("%[ax]", "%[ay]", "%[az]"), "7")
                 : [off] "+s"(off), [which] "+s"(which), [resume] "+v"(resume), [ax] "+v"(ax), [ay] "+v"(ay),
                   [az] "+v"(az), [sx] "+v"(sx), [sy] "+v"(sy), [sz] "+v"(sz), [t64] "=&v"(t64), [cnt] "=&s"(cnt),
                   [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [d2] "=&v"(d2), [inv] "=&v"(inv),
                   [f] "=&v"(f), [t] "=&v"(t)
                 : [base] "s"(nodes), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps2] "s"(eps2), [end] "s"(end),
                   [band2] "s"(band2)
                 : NBMI_CLOBBERS);
}

// Two cursors in one wave: the same 64 bodies walk [off1, end1) and [off2, end2) of the array at
// once, one visit of each per trip.  A wave issues in order and a scalar load can only be awaited with
// lgkmcnt(0), so the trip waits once, for both records, and both cursors' next loads are in flight
// while the other cursor's instructions issue: two dependent load chains per wave instead of one.
// Runs while BOTH cursors are inside their ranges: the first is checked after every trip (it must not
// step into the second's range), the second after every other trip (its range ends with the
// self-looping sentinel, one idle visit at worst); the caller finishes the longer one alone.
__device__ __forceinline__ void walk_pair_asm(const Node *nodes, unsigned &off1, unsigned end1, unsigned &off2,
                                              unsigned end2, float px, float py, float pz, float eps2, unsigned band2,
                                              unsigned &resume1, unsigned &resume2, float &ax, float &ay, float &az,
                                              float &ax2, float &ay2, float &az2, double &sx, double &sy, double &sz,
                                              unsigned &which) {
    float dx, dy, dz, d2, inv, f, t;
    double t64;
    unsigned cnt;
    asm volatile("s_mov_b32 %[cnt], 3\n"
                 "s_load_dwordx4 s[36:39], %[base], %[off]\n"
                 "s_load_dwordx2 s[40:41], %[base], %[off] offset:16\n"
                 "s_load_dwordx4 s[60:63], %[base], %[off2]\n"
                 "s_load_dwordx2 s[64:65], %[base], %[off2] offset:16\n"
                 "1:\n" NBMI_VISIT_1A("21", "31", "51") NBMI_VISIT_2A("22", "32", "52")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc0 9f\n" NBMI_VISIT_1B("23", "33", "53") NBMI_VISIT_2B("24", "34", "54")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc0 9f\n" NBMI_TRIP_COUNT
                 "s_cmp_lt_u32 %[off2], %[end2]\n"
                 "s_cbranch_scc1 1b\n"
                 NBMI_EXITS(NBMI_OPEN_A("21", "31", "51") NBMI_OPEN_2A("22", "32", "52") NBMI_OPEN_B("23", "33", "53") NBMI_OPEN_2B("24", "34", "54"),
                            NBMI_FLUSH("%[ax]", "%[ay]", "%[az]") NBMI_FLUSH("%[ax2]", "%[ay2]", "%[az2]"), "3")
                 : [off] "+s"(off1), [off2] "+s"(off2), [which] "+s"(which), [resume] "+v"(resume1),
                   [resume2] "+v"(resume2), [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az), [ax2] "+v"(ax2),
                   [ay2] "+v"(ay2), [az2] "+v"(az2), [sx] "+v"(sx), [sy] "+v"(sy), [sz] "+v"(sz), [t64] "=&v"(t64),
                   [cnt] "=&s"(cnt), [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [d2] "=&v"(d2),
                   [inv] "=&v"(inv), [f] "=&v"(f), [t] "=&v"(t)
                 : [base] "s"(nodes), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps2] "s"(eps2), [end] "s"(end1),
                   [end2] "s"(end2), [band2] "s"(band2)
                 : NBMI_CLOBBERS, "s60", "s61", "s62", "s63", "s64", "s65", "s68", "s69", "s70", "s71", "s72", "s73");
}

// ---------------------------------------------------------------------------------------
// [r3] The float64 visit.  Why it exists (scripts/gpu_prec_diag.py, profiles/r03_precision_modes.jsonl): at 1 M
// bodies x 100 steps the position error against the float64 reference is made by fp32 pair arithmetic, and not by
// its random roundings but by its SYSTEMATIC ones - G m rounded to fp32 (alone: max 6.6e-5 of the largest
// coordinate), v_rsq_f32's one-ulp error pattern (alone: 9.5e-5), fp32-rounded coordinates of near pairs - which act
// like a slightly different force law step after step on the bodies of the dense inner disk, where a difference,
// once it flips an opening decision, cascades.  With the force of every accepted visit in float64 (same accepted
// sets) the GPU follows the reference to 3e-14 over the 100 steps.  Same lock-step scheme as the fp32 loop, operands
// of the 40-byte NodeD record in SGPR pairs:
//   d = c - p, d2 = |d|^2 + eps^2 in float64;  the opening test on fp32(d2) against the same s2t / band as the fp32
//   loop (fp32(d2) is within half an ulp of the true value: well inside what the band allows for, so the accepted
//   sets are the reference's here too);  y0 = v_rsq_f32(fp32(d2)), e = 1 - d2 y0^2 (one FMA, exact to 1e-16),
//   G m d2^(-3/2) = G m y0^3 (1 + 1.5 e) [+ O(e^2) = 1e-14];  three float64 FMAs into the sums.
// 17 float64-rate + 4 fp32-rate vector instructions per visit (the fp32 visit: 16 fp32-rate).
// ---------------------------------------------------------------------------------------
#define NBMI_V64_X(CX, CY, CZ, GM, S2T, NXT, NEXT8, NEXT2, LOPEN, LJOIN, LSKIP) \
    "v_cmpx_ge_u32_e64 s[58:59], %[off], %[resume]\n"          \
    "s_waitcnt lgkmcnt(0)\n"                                   \
    "v_add_f64 %[dx], " CX ", -%[px]\n"                        \
    "v_add_f64 %[dy], " CY ", -%[py]\n"                        \
    "v_add_f64 %[dz], " CZ ", -%[pz]\n"                        \
    "v_fma_f64 %[d2], %[dx], %[dx], %[eps2]\n"                 \
    "v_fma_f64 %[d2], %[dy], %[dy], %[d2]\n"                   \
    "v_fma_f64 %[d2], %[dz], %[dz], %[d2]\n"                   \
    "v_cvt_f32_f64_e32 %[d2f], %[d2]\n"                        \
    "v_cmpx_lt_i32_e64 s[60:61], " S2T ", %[d2f]\n"            \
    "s_andn2_b64 s[62:63], s[58:59], s[60:61]\n"               \
    "s_cbranch_scc1 " LOPEN "f\n"                              \
    "s_mov_b32 %[off], " NXT "\n"                              \
    LJOIN ":\n"                                                \
    "s_load_dwordx8 " NEXT8 ", %[base], %[off]\n"              \
    "s_load_dwordx2 " NEXT2 ", %[base], %[off] offset:32\n"    \
    "v_rsq_f32_e32 %[d2f], %[d2f]\n"                           \
    "v_mov_b32_e32 %[resume], " NXT "\n"                       \
    "v_cvt_f64_f32_e32 %[y0], %[d2f]\n"                        \
    "v_mul_f64 %[t], %[y0], %[y0]\n"                           \
    "v_mul_f64 %[w], " GM ", %[y0]\n"                          \
    "v_fma_f64 %[d2], -%[d2], %[t], 1.0\n"                     \
    "v_mul_f64 %[w], %[w], %[t]\n"                             \
    "v_mul_f64 %[d2], %[d2], %[c15]\n"                         \
    "v_fma_f64 %[w], %[w], %[d2], %[w]\n"                      \
    "v_fma_f64 %[sx], %[dx], %[w], %[sx]\n"                    \
    "v_fma_f64 %[sy], %[dy], %[w], %[sy]\n"                    \
    "v_fma_f64 %[sz], %[dz], %[w], %[sz]\n"                    \
    "s_mov_b64 exec, -1\n"                                    \
    LSKIP ":\n"
#define NBMI_O64_X(S2T, NEXT8, NEXT2, LOPEN, LJOIN, LSKIP) \
    LOPEN ":\n"                                      \
    "s_sub_u32 s66, " S2T ", %[band2]\n"             \
    "s_mov_b64 exec, s[62:63]\n"                     \
    "v_cmp_lt_i32_e64 s[64:65], s66, %[d2f]\n"       \
    "s_mov_b64 exec, s[60:61]\n"                     \
    "s_cmp_lg_u64 s[64:65], 0\n"                     \
    "s_cbranch_scc1 7f\n"                            \
    "s_add_u32 %[off], %[off], 40\n"                 \
    "s_cmp_lg_u64 s[60:61], 0\n"                     \
    "s_cbranch_scc1 " LJOIN "b\n"                    \
    /* nobody takes the node: request the next record, skip the force block */ \
    "s_load_dwordx8 " NEXT8 ", %[base], %[off]\n"    \
    "s_load_dwordx2 " NEXT2 ", %[base], %[off] offset:32\n" \
    "s_mov_b64 exec, -1\n"                           \
    "s_branch " LSKIP "b\n"
#define NBMI_V64_A(LO, LJ, LS) \
    NBMI_V64_X("s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s44", "s45", "s[48:55]", "s[56:57]", LO, LJ, LS)
#define NBMI_V64_B(LO, LJ, LS) \
    NBMI_V64_X("s[48:49]", "s[50:51]", "s[52:53]", "s[54:55]", "s56", "s57", "s[36:43]", "s[44:45]", LO, LJ, LS)
#define NBMI_O64_A(LO, LJ, LS) NBMI_O64_X("s44", "s[48:55]", "s[56:57]", LO, LJ, LS)
#define NBMI_O64_B(LO, LJ, LS) NBMI_O64_X("s56", "s[36:43]", "s[44:45]", LO, LJ, LS)

// walks from `off` to the end of the NodeD array (4 visits per loop test, the sentinel absorbs the overshoot) or
// until a near-tie stops it (which = 1, cursor on the tied node)
__device__ __forceinline__ void walk4_asm64(const NodeD *nodesd, unsigned &off, unsigned end, double px, double py,
                                            double pz, double eps2, unsigned band2, unsigned &resume, double &sx,
                                            double &sy, double &sz, unsigned &which) {
    double dx, dy, dz, d2, y0, t, w;
    float d2f;
    const double c15 = 1.5;
    asm volatile("s_load_dwordx8 s[36:43], %[base], %[off]\n"
                 "s_load_dwordx2 s[44:45], %[base], %[off] offset:32\n"
                 "1:\n" NBMI_V64_A("21", "31", "51") NBMI_V64_B("22", "32", "52") NBMI_V64_A("23", "33", "53") NBMI_V64_B("24", "34", "54")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc1 1b\n"
                 "s_branch 9f\n"
                 NBMI_O64_A("21", "31", "51") NBMI_O64_B("22", "32", "52") NBMI_O64_A("23", "33", "53") NBMI_O64_B("24", "34", "54")
                 "7:\n"
                 "s_mov_b64 exec, -1\n"
                 "s_mov_b32 %[which], 1\n"
                 "9:\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : [off] "+s"(off), [which] "+s"(which), [resume] "+v"(resume), [sx] "+v"(sx), [sy] "+v"(sy), [sz] "+v"(sz),
                   [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [d2] "=&v"(d2), [y0] "=&v"(y0), [t] "=&v"(t),
                   [w] "=&v"(w), [d2f] "=&v"(d2f)
                 : [base] "s"(nodesd), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps2] "s"(eps2), [c15] "s"(c15),
                   [end] "s"(end), [band2] "s"(band2)
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                   "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
                   "s66", "vcc", "scc", "memory");
}

// the same with one visit per loop test: a part of the array must not be stepped past (split walk)
__device__ __forceinline__ void walk1_asm64(const NodeD *nodesd, unsigned &off, unsigned end, double px, double py,
                                            double pz, double eps2, unsigned band2, unsigned &resume, double &sx,
                                            double &sy, double &sz, unsigned &which) {
    double dx, dy, dz, d2, y0, t, w;
    float d2f;
    const double c15 = 1.5;
    asm volatile("s_load_dwordx8 s[36:43], %[base], %[off]\n"
                 "s_load_dwordx2 s[44:45], %[base], %[off] offset:32\n"
                 "1:\n" NBMI_V64_A("21", "31", "51")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc0 9f\n" NBMI_V64_B("22", "32", "52")
                 "s_cmp_lt_u32 %[off], %[end]\n"
                 "s_cbranch_scc1 1b\n"
                 "s_branch 9f\n"
                 NBMI_O64_A("21", "31", "51") NBMI_O64_B("22", "32", "52")
                 "7:\n"
                 "s_mov_b64 exec, -1\n"
                 "s_mov_b32 %[which], 1\n"
                 "9:\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : [off] "+s"(off), [which] "+s"(which), [resume] "+v"(resume), [sx] "+v"(sx), [sy] "+v"(sy), [sz] "+v"(sz),
                   [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [d2] "=&v"(d2), [y0] "=&v"(y0), [t] "=&v"(t),
                   [w] "=&v"(w), [d2f] "=&v"(d2f)
                 : [base] "s"(nodesd), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [eps2] "s"(eps2), [c15] "s"(c15),
                   [end] "s"(end), [band2] "s"(band2)
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",
                   "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65",
                   "s66", "vcc", "scc", "memory");
}

// one float64 visit in C++ with the float64 re-decision of the lanes inside the band (the asm loop stopped on this
// node).  Operation for operation the asm visit: a body's sums do not depend on which of its visits came through here.
__device__ __forceinline__ unsigned tie_visit64(const NodeD *nodesd, unsigned off, double qx, double qy, double qz,
                                                double eps2, unsigned band2, const Body64 &b64, unsigned &resume,
                                                double &sx, double &sy, double &sz) {
    off = __builtin_amdgcn_readfirstlane(off);
    const NodeD *np = reinterpret_cast<const NodeD *>(reinterpret_cast<const char *>(nodesd) + off);
    const bool active = resume <= off;
    bool geom;
    {   // the decision first, in a scope of its own: the float64 re-test is register hungry, and nothing of the
        // force arithmetic below needs to be alive across it (the pointer is laundered so that it is recomputed)
        const double dx = np->cx - qx, dy = np->cy - qy, dz = np->cz - qz;
        const float d2f = (float)__builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
        const int d2b = __float_as_int(d2f), hi = __float_as_int(np->s2t), lo = hi - (int)band2;
        geom = hi < d2b;
        if (active && !geom && lo < d2b) geom = (hi == 0) || exact_take_idx(off / kNodeDBytes, b64);
    }
    asm volatile("" : "+s"(np));
    const NodeD nd = *np;
    const double dx = nd.cx - qx, dy = nd.cy - qy, dz = nd.cz - qz;
    const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, __builtin_fma(dx, dx, eps2)));
    const float d2f = (float)d2;
    const bool take = active && geom;
    const double y0 = (double)__builtin_amdgcn_rsqf(d2f);
    const double t = y0 * y0;
    double w = nd.gm * y0;
    const double e = __builtin_fma(-d2, t, 1.0);
    w = w * t;
    const double h = e * 1.5;
    w = __builtin_fma(w, h, w);
    if (take) {
        sx = __builtin_fma(dx, w, sx); sy = __builtin_fma(dy, w, sy); sz = __builtin_fma(dz, w, sz);
        resume = nd.next_off;
    }
    const unsigned long long any_open = __builtin_amdgcn_ballot_w64(active && !geom);
    return __builtin_amdgcn_readfirstlane(any_open ? off + kNodeDBytes : nd.next_off);
}

// a wave-uniform 64-bit value the compiler no longer knows to be uniform (loaded behind something it treats as a
// possible store, e.g. the cycle counter read of the balance mode) back into scalar registers
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// everything a lane carries through a walk
struct WalkCtx {
    const Node *nodes;
    float px, py, pz;
    Body64 b64;
    unsigned band2;
};

// one visit with the float64 re-decision (the asm loop stopped on this node)
__device__ __forceinline__ unsigned tie_visit(const WalkCtx &C, const WalkParams &P, unsigned off, unsigned &resume,
                                              float &ax, float &ay, float &az) {
    bool a_, f_, j_, b_;
    return __builtin_amdgcn_readfirstlane(
        visit<false>(C.nodes, off, C.px, C.py, C.pz, C.b64, P, C.band2, resume, ax, ay, az, a_, f_, j_, b_));
}

// the cursor `off` walks until it reaches `end`; kToEnd: `end` is the end of the array (unrolled loop)
template <bool kToEnd>
__device__ __forceinline__ void walk_span(const WalkCtx &C, const WalkParams &P, unsigned off, unsigned end,
                                          unsigned &resume, float &ax, float &ay, float &az, double &sx, double &sy,
                                          double &sz) {
    off = __builtin_amdgcn_readfirstlane(off);
    while (off < end) {
        unsigned which = 0u;
        if (kToEnd) walk4_asm(C.nodes, off, end, C.px, C.py, C.pz, P.eps2, C.band2, resume, ax, ay, az, sx, sy, sz, which);
        else walk1_asm(C.nodes, off, end, C.px, C.py, C.pz, P.eps2, C.band2, resume, ax, ay, az, sx, sy, sz, which);
        off = __builtin_amdgcn_readfirstlane(off);
        if (!__builtin_amdgcn_readfirstlane(which)) break;
        off = tie_visit(C, P, off, resume, ax, ay, az);
    }
}

// Split / pair walks start a cursor in the middle of the array, at offset S.  The lane's `resume` there
// is what the full walk would have left: only the ancestors of S can have set it beyond S, so seek()
// replays the opening test down that chain (no forces: an ancestor lies before S and belongs to
// another part).  Returns the first offset >= S the walk visits.
__device__ __forceinline__ unsigned seek(const WalkCtx &C, const WalkParams &P, unsigned S, unsigned &resume) {
    unsigned off = 0u;
    while (off < S) {
        off = __builtin_amdgcn_readfirstlane(off);
        const Node nd = *reinterpret_cast<const Node *>(reinterpret_cast<const char *>(C.nodes) + off);
        const float dx = nd.cx - C.px, dy = nd.cy - C.py, dz = nd.cz - C.pz;
        const float dist_sq = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, P.eps2)));
        const bool active = resume <= off;
        const int d2b = __float_as_int(dist_sq), hi = __float_as_int(nd.s2t), lo = hi - (int)C.band2;
        bool geom = hi < d2b;
        if (active && !geom && lo < d2b) geom = (hi == 0) || exact_take(off, C.b64);
        if (active && geom) resume = nd.next_off;
        if (__builtin_amdgcn_ballot_w64(active && !geom) == 0ull) {
            off = nd.next_off;  // nobody opens this ancestor: the walk never enters it
            continue;
        }
        // descend to the child whose subtree contains S
        unsigned c = off + kNodeBytes;
        for (;;) {
            c = __builtin_amdgcn_readfirstlane(c);
            const unsigned nxt = reinterpret_cast<const Node *>(reinterpret_cast<const char *>(C.nodes) + c)->next_off;
            if (nxt > S) break;
            c = nxt;
        }
        off = c;
    }
    return off;
}

// fused kick-drift (simulation.py:291-305) of the body at sorted rank `rank`, written at that rank of the
// other buffer; frozen (a capacity error is pending): the body moves to its rank unchanged
__device__ __forceinline__ double integrate(const WalkTable *tab, uint32_t j, int64_t rank, double ax,
                                            double ay, double az, const WalkParams &P, bool frozen) {
    const Bodies cur = tab->buf[P.curbuf], nxt = tab->buf[1 - P.curbuf];
    double vx = cur.vx[j], vy = cur.vy[j], vz = cur.vz[j];
    double x0 = cur.x[j], y0 = cur.y[j], z0 = cur.z[j];
    const double m0 = cur.m[j];
    const int32_t id0 = cur.id[j];
    if (!frozen) {
        vx += ax * P.dt; vy += ay * P.dt; vz += az * P.dt;
        vx *= P.damping; vy *= P.damping; vz *= P.damping;
        x0 += vx * P.dt; y0 += vy * P.dt; z0 += vz * P.dt;
    }
    nxt.vx[rank] = vx; nxt.vy[rank] = vy; nxt.vz[rank] = vz;
    nxt.x[rank] = x0; nxt.y[rank] = y0; nxt.z[rank] = z0;
    nxt.m[rank] = m0;
    nxt.id[rank] = id0;
    return fmax(fmax(fabs(x0), fabs(y0)), fabs(z0));
}

// The largest |coordinate| a wave has just written joins TreeInfo::maxabs_next (all lanes of the wave call this;
// lanes without a body pass 0).  Most waves find a value at least as large already there and skip the atomic.
__device__ __forceinline__ void publish_maxabs(const WalkTable *tab, double lane_max) {
    const double m = wave_max(lane_max);
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        unsigned long long *dst = tab->maxabs_next;
        if (bits > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, bits);
    }
}

// [r3] Cuts for the balance mode of k_walk: XCD x gets the logical blocks [b[x], b[x + 1]) such that every range
// holds about an eighth of last step's total wave time (bodies move little between steps), at most `jmax` blocks
// (the launch has 8 jmax workgroups) and at least one.  One workgroup: per-thread chunk sums, scan, the seven
// thread(s) whose chunk holds a cut find it.  All-zero times (first step) give equal ranges.
__global__ __launch_bounds__(1024) void k_xcd_bounds(const unsigned *__restrict__ wave_cycles, int nb, int jmax, int *__restrict__ bounds) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long total_s;
    __shared__ int cut[9];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int chunk = (nb + 1023) / 1024;
    const int b = t * chunk < nb ? t * chunk : nb, e = b + chunk < nb ? b + chunk : nb;
    unsigned long long acc = 0;
    for (int i = b; i < e; i++) {
        const uint4 q = reinterpret_cast<const uint4 *>(wave_cycles)[i];
        acc += (unsigned long long)q.x + q.y + q.z + q.w + 1ull;  // + 1: all-zero input still cuts into equal parts
    }
    unsigned long long inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned long long off = 0, total = 0;
    for (int q = 0; q < 16; q++) {
        if (q < w) off += wsum[q];
        total += wsum[q];
    }
    inc += off;
    if (t == 0) { total_s = total; cut[0] = 0; cut[8] = nb; }
    __syncthreads();
    const unsigned long long ex = inc - acc;  // work before this thread's chunk
    for (int k = 1; k < 8; k++) {
        const unsigned long long target = total_s / 8ull * (unsigned long long)k;
        if (b < e && ex <= target && target < inc) {
            unsigned long long run = ex;
            int i = b;
            for (; i < e; i++) {
                const uint4 q = reinterpret_cast<const uint4 *>(wave_cycles)[i];
                run += (unsigned long long)q.x + q.y + q.z + q.w + 1ull;
                if (run > target) break;
            }
            cut[k] = i + 1 < nb ? i + 1 : nb;
        }
    }
    __syncthreads();
    if (t == 0) {
        // ranges of 1 .. jmax blocks, in order; what the clamps push out goes to the later XCDs
        int prev = 0;
        for (int k = 1; k < 8; k++) {
            int c = cut[k];
            const int lo = prev + 1, hi = prev + jmax;
            const int need_after = (8 - k);  // blocks the remaining XCDs need at least ...
            const int room_after = (8 - k) * jmax;  // ... and can take at most
            if (c < lo) c = lo;
            if (c > hi) c = hi;
            if (nb - c < need_after) c = nb - need_after;
            if (nb - c > room_after) c = nb - room_after;
            bounds[k] = c;
            prev = c;
        }
        bounds[0] = 0;
        bounds[8] = nb;
    }
}

// The walk kernel.  kCount = parity/measurement build (C++ visit, work counters);
// otherwise the hand-scheduled loop (eps > 0) or the C++ visit with the distance guard (eps == 0).
template <bool kIntegrate, bool kCount, bool kGuard>
__global__ __launch_bounds__(kBlock) void k_walk(const Node *__restrict__ nodes, const WalkTable *tab,
                                                 const TreeInfo *info_in, const float4 *__restrict__ posm_s,
                                                 const uint32_t *__restrict__ perm,
                                                 double *__restrict__ acc_out, WalkParams P, TreeInfo *info_out) {
    // [r3] Which logical block (= which 256 consecutive ranks) this workgroup walks.  Hardware deals workgroups to the
    // eight XCDs round-robin; logical_block() gives every XCD one contiguous eighth of the blocks (neighbouring
    // groups walk the same nodes: L2 locality) - equal in blocks, not in work: the XCD that got the densest part
    // of the system sets the kernel's time (10 M collision: the busiest XCD has 5.5 % more visits than the mean).
    // Balance mode keeps the contiguous ranges but cuts them where last step's measured wave times say an eighth of
    // the WORK ends (k_xcd_bounds); a workgroup beyond its XCD's range has nothing to do.
    int lb;
    if (kIntegrate && P.balance) {
        const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
        const int b0 = __builtin_amdgcn_readfirstlane(tab->xcd_bounds[xcd]);
        const int b1 = __builtin_amdgcn_readfirstlane(tab->xcd_bounds[xcd + 1]);
        lb = b0 + jb;
        if (lb >= b1) return;
    } else {
        lb = logical_block(blockIdx.x, gridDim.x, P.xcd_chunk);
    }
    const int lane = threadIdx.x & 63;
    const int64_t rank = P.rank_begin + (int64_t)lb * blockDim.x + threadIdx.x;
    const bool valid = rank < P.rank_end;
    const bool frozen = info_in->error != 0 || info_in->sticky_error != 0;
    const unsigned nn = frozen ? 0u : ((unsigned)info_in->walk_nodes * kNodeBytes);  // end offset

    WalkCtx C;
    C.nodes = nodes;
    C.px = C.py = C.pz = 0.f;
    C.band2 = __builtin_amdgcn_readfirstlane(info_in->band2);
    uint32_t j = 0;
    if (valid) {
        const float4 p = posm_s[rank];
        C.px = p.x; C.py = p.y; C.pz = p.z;
        j = perm[rank];
    }
    C.b64 = Body64{tab, P.curbuf, j};
    unsigned resume = valid ? 0u : 0xffffffffu;
    float ax = 0.f, ay = 0.f, az = 0.f;  // fp32 accumulators of the loops ...
    double sx = 0.0, sy = 0.0, sz = 0.0;  // ... emptied into these every few trips (two-level sums, NBMI_FLUSH)
    double acc64x = 0.0, acc64y = 0.0, acc64z = 0.0;

    // [r3] force precision of this wave (see NBMI_V64_X).  force_prec 0: float64 where the bodies' own neighbourhood is
    // dense enough that an error, once made, is amplified within a few hundred steps (decided in k_gather_scan, which
    // has the sorted bodies in registers anyway), or where most of the system is.  All lanes of a wave agree; which
    // 64 ranks form a wave does not depend on the sharding.
    bool use64 = false;
    if (!kCount && !kGuard && kIntegrate && tab->nodesd && P.force_prec != 1) {
        if (P.force_prec == 2) {
            use64 = true;
        } else {
            // k_gather_scan has already decided, wave by wave (and k_scan_subtiles for the system as a whole)
            use64 = tab->wave_flag[(P.rank_begin + (int64_t)lb * blockDim.x + threadIdx.x) >> 6] != 0 || info_in->force_all64 != 0;
        }
        use64 = __builtin_amdgcn_readfirstlane((int)use64) != 0;
    }
    // (the clock is read here, behind the prologue's loads: the compiler treats the read as a possible store and
    // turns every scalar load that follows it into a vector load)
    const unsigned long long t_start = (kIntegrate && P.balance) ? __builtin_readcyclecounter() : 0ull;
    if (use64) {
        const NodeD *nodesd = reinterpret_cast<const NodeD *>(uniform_u64(reinterpret_cast<unsigned long long>(tab->nodesd)));
        double qx = 0.0, qy = 0.0, qz = 0.0;
        if (valid) {
            const Bodies &cur = tab->buf[P.curbuf];
            qx = cur.x[j]; qy = cur.y[j]; qz = cur.z[j];
        }
        const double eps2d = __longlong_as_double((long long)uniform_u64((unsigned long long)__double_as_longlong(tab->eps2)));
        const unsigned nnd = __builtin_amdgcn_readfirstlane(frozen ? 0u : ((unsigned)info_in->walk_nodes * kNodeDBytes));
        unsigned off = 0u;
        while (off < nnd) {
            unsigned which = 0u;
            walk4_asm64(nodesd, off, nnd, qx, qy, qz, eps2d, 2u * kBand64, resume, sx, sy, sz, which);
            off = __builtin_amdgcn_readfirstlane(off);
            if (!__builtin_amdgcn_readfirstlane(which)) break;
            off = tie_visit64(nodesd, off, qx, qy, qz, eps2d, 2u * kBand64, C.b64, resume, sx, sy, sz);
        }
    } else if (!kCount && !kGuard) {
        if (nn && P.pair) {
            // two cursors: [0, mid) and [mid, nn); the second needs the lanes' state at mid (seek).
            // Where to cut: a wave spends ~40 % of its visits inside the 1/64 of the array around its own bodies
            // (scripts/analysis/range_balance.py), so equal halves keep both cursors busy for only a fifth of the
            // visits; cutting at the leaf of the wave's middle body does for two thirds of them.
            // (owner mode: the array begins with a jump node and unused rows up to walk_first)
            unsigned mid = __builtin_amdgcn_readfirstlane((unsigned)((info_in->walk_first + info_in->walk_nodes) / 2) * kNodeBytes);
            if (P.pair == 2) {
                const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                const int64_t r0 = P.rank_begin + (int64_t)lb * blockDim.x + wv * 64;
                const int64_t rl = r0 + 64 < P.rank_end ? r0 + 64 : P.rank_end;  // the wave's bodies: [r0, rl)
                const int64_t rm = r0 < rl ? r0 + (rl - r0) / 2 : 0;
                const unsigned home = __builtin_amdgcn_readfirstlane((unsigned)(rm + pex_at(tab->pex, tab->subpex, rm + 1) + tab->own_base) * kNodeBytes);
                mid = (r0 < rl && home > 0u && home < nn) ? home : mid;
            }
            // (each half sums into its own accumulator, added at the end: a body's result does not depend on how the two
            // cursors' visits interleave.  With the home cut it does depend on which 64 ranks form the wave - the same
            // ones for every sharding whose ranges start at multiples of 64 ranks)
            unsigned resume2 = resume;
            float bx = 0.f, by = 0.f, bz = 0.f;
            unsigned o1 = 0u, o2 = __builtin_amdgcn_readfirstlane(mid ? seek(C, P, mid, resume2) : 0u);
            while (mid && o1 < mid && o2 < nn) {
                unsigned which = 0u;
                walk_pair_asm(nodes, o1, mid, o2, nn, C.px, C.py, C.pz, P.eps2, C.band2, resume, resume2, ax, ay, az,
                              bx, by, bz, sx, sy, sz, which);
                o1 = __builtin_amdgcn_readfirstlane(o1);
                o2 = __builtin_amdgcn_readfirstlane(o2);
                which = __builtin_amdgcn_readfirstlane(which);
                if (which == 1u) o1 = tie_visit(C, P, o1, resume, ax, ay, az);
                else if (which == 2u) o2 = tie_visit(C, P, o2, resume2, bx, by, bz);
                else break;
            }
            if (o1 < mid) walk_span<false>(C, P, o1, mid, resume, ax, ay, az, sx, sy, sz);
            if (o2 < nn && o2 >= mid) walk_span<true>(C, P, o2, nn, resume2, bx, by, bz, sx, sy, sz);
            sx += (double)bx; sy += (double)by; sz += (double)bz;
        } else if (nn) {
            walk_span<true>(C, P, 0u, nn, resume, ax, ay, az, sx, sy, sz);
        }
    } else {
        unsigned long long wv = 0, lv = 0, la = 0, jm = 0, bd = 0;
        double dax = 0.0, day = 0.0, daz = 0.0;
        unsigned long long wm[4] = {0, 0, 0, 0};
        int wbase[4] = {-1000, -1000, -1000, -1000};
        unsigned off = 0u;
        while (off < nn) {
            bool a_, f_, j_, b_;
            const int c_old = (int)(off / kNodeBytes);
            if (kCount && P.acc64) {
                float tx = 0.f, ty = 0.f, tz = 0.f;
                off = visit<kGuard>(nodes, off, C.px, C.py, C.pz, C.b64, P, C.band2, resume, tx, ty, tz, a_, f_, j_, b_);
                dax += (double)tx; day += (double)ty; daz += (double)tz;
            } else {
                off = visit<kGuard>(nodes, off, C.px, C.py, C.pz, C.b64, P, C.band2, resume, ax, ay, az, a_, f_, j_, b_);
            }
            if (kCount) {
                wv += 1; lv += a_ ? 1 : 0; la += f_ ? 1 : 0; jm += j_ ? 1 : 0; bd += b_ ? 1 : 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    if (c_old < wbase[w] || c_old >= wbase[w] + (8 << w)) { wm[w]++; wbase[w] = c_old; }
                }
            }
        }
        if (kCount && P.acc64) { acc64x = dax; acc64y = day; acc64z = daz; }
        if (kCount) {
            // wave_visits counted once per wave (lane 0), lane counters summed over lanes
            if (lane == 0) {
                const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u;
                atomicAdd(&info_out->xcd_visits[xcc], wv);
                atomicAdd(&info_out->wave_visits, wv);
                for (int w = 0; w < 4; w++) atomicAdd(&info_out->win_miss[w], wm[w]);
                atomicAdd(&info_out->jumps, jm);
            }
            atomicAdd(&info_out->lane_visits, lv);
            atomicAdd(&info_out->lane_accepts, la);
            if (bd) atomicAdd(&info_out->band_visits, bd);
        }
    }
    if (kIntegrate) {
        publish_maxabs(tab, valid ? integrate(tab, j, rank, sx + (double)ax, sy + (double)ay, sz + (double)az, P, frozen) : 0.0);
        if (P.balance && lane == 0) {
            const unsigned long long dtc = __builtin_readcyclecounter() - t_start;
            tab->wave_cycles[4 * lb + (threadIdx.x >> 6)] = (unsigned)(dtc > 0xffffffffull ? 0xffffffffull : dtc);
        }
    } else if (valid) {
        const int64_t o = 3 * (int64_t)tab->buf[P.curbuf].id[j];
        const bool d64 = kCount && P.acc64;
        acc_out[o] = d64 ? acc64x : (double)ax; acc_out[o + 1] = d64 ? acc64y : (double)ay; acc_out[o + 2] = d64 ? acc64z : (double)az;
    }
}

// ---------------------------------------------------------------------------------------
// Split walk for small systems.  Below ~250 k bodies there are fewer groups than wave slots and the
// step time is ONE wave's serial chain of ~1 700 dependent visits (0.32 ms at 100 k bodies, where
// the same work spread over the chip would take 0.1 ms).  Here a block of K waves shares one group
// of 64 bodies and wave w walks the w-th K-th of the pre-order array, [w N/K, (w+1) N/K) nodes.
// (Measured at 100 k bodies, K = 4: equal quarters 0.179 ms; quarters centred on the group's own
// leaves with a near range of 1/8 ... 1/8192 of the array 0.204 ... 0.257 ms - a galaxy's far field
// is where most visits are, so plain equal parts balance best, and they do not depend on the
// group.)  A part that starts at S gets its lanes' `resume` from seek().  The K partial
// sums meet in LDS and are added in fixed order, then the usual fused kick-drift.  Same accepted
// (body, node) set as the one-wave walk; the fp32 sums associate differently (by fixed node ranges,
// so a body's result still does not depend on its group or on the sharding).
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(64 * K) void k_walk_split(const Node *__restrict__ nodes, const WalkTable *tab,
                                                       const TreeInfo *info_in, const float4 *__restrict__ posm_s,
                                                       const uint32_t *__restrict__ perm, WalkParams P) {
    __shared__ double part[K][3][64];
    const int lb = logical_block(blockIdx.x, gridDim.x, P.xcd_chunk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t rank = P.rank_begin + (int64_t)lb * 64 + lane;
    const bool valid = rank < P.rank_end;
    const bool frozen = info_in->error != 0 || info_in->sticky_error != 0;
    const int64_t num_nodes = frozen ? 0 : info_in->walk_nodes;

    WalkCtx C;
    C.nodes = nodes;
    C.px = C.py = C.pz = 0.f;
    C.band2 = __builtin_amdgcn_readfirstlane(info_in->band2);
    uint32_t j = 0;
    if (valid) {
        const float4 p = posm_s[rank];
        C.px = p.x; C.py = p.y; C.pz = p.z;
        j = perm[rank];
    }
    C.b64 = Body64{tab, P.curbuf, j};
    unsigned resume = valid ? 0u : 0xffffffffu;
    float ax = 0.f, ay = 0.f, az = 0.f;
    double sx = 0.0, sy = 0.0, sz = 0.0;  // two-level sums, see NBMI_FLUSH
    // wave-uniform range (w is the wave index): tell the compiler so
    // (owner mode: rows [1, walk_first) behind the jump node at row 0 are unused; the parts divide the rest)
    const int64_t first = frozen ? 0 : info_in->walk_first, span = num_nodes - first;
    // ([r4] measured and dropped: parts graded by distance from the group's own leaves - the K waves sharing the two sides
    // of the home leaf in proportion to their octaves of distance, a side's parts ending at home +- 48 * 2^(t octaves / parts)
    // nodes.  Walk 0.076 -> 0.096 ms at 10 k bodies, 0.157 -> 0.182 at 30 k, 0.216 -> 0.244 at 100 k, 0.475 -> 0.454 at
    // 262 k: a galaxy's visits are spread over the far field more evenly than one per octave, equal K-ths stay.)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(w == 0 ? 0 : first + span * w / K) * kNodeBytes);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(first + span * (w + 1) / K) * kNodeBytes);
    // [r3] force precision of the group (all K waves of the workgroup walk the same 64 bodies): as in k_walk
    bool use64 = false;
    if (tab->nodesd && P.force_prec != 1)
        use64 = P.force_prec == 2 || tab->wave_flag[(P.rank_begin + (int64_t)lb * 64) >> 6] != 0 || info_in->force_all64 != 0;
    use64 = __builtin_amdgcn_readfirstlane((int)use64) != 0;
    if (lo < hi) {
        // (seek works on the fp32 records in both cases: it only replays opening decisions, and those are the same)
        const unsigned c0 = __builtin_amdgcn_readfirstlane(lo == 0u ? 0u : seek(C, P, lo, resume));
        if (c0 < hi && !use64) {
            walk_span<false>(C, P, c0, hi, resume, ax, ay, az, sx, sy, sz);
        } else if (c0 < hi) {
            const NodeD *nodesd = reinterpret_cast<const NodeD *>(uniform_u64(reinterpret_cast<unsigned long long>(tab->nodesd)));
            double qx = 0.0, qy = 0.0, qz = 0.0;
            if (valid) {
                const Bodies &cur = tab->buf[P.curbuf];
                qx = cur.x[j]; qy = cur.y[j]; qz = cur.z[j];
            }
            const double eps2d = __longlong_as_double((long long)uniform_u64((unsigned long long)__double_as_longlong(tab->eps2)));
            // offsets of the 24-byte records -> offsets of the 40-byte ones (same node indices)
            unsigned off = c0 / kNodeBytes * kNodeDBytes;
            const unsigned end = hi / kNodeBytes * kNodeDBytes;
            unsigned res64 = resume == 0xffffffffu ? resume : resume / kNodeBytes * kNodeDBytes;
            while (off < end) {
                unsigned which = 0u;
                walk1_asm64(nodesd, off, end, qx, qy, qz, eps2d, 2u * kBand64, res64, sx, sy, sz, which);
                off = __builtin_amdgcn_readfirstlane(off);
                if (!__builtin_amdgcn_readfirstlane(which)) break;
                off = tie_visit64(nodesd, off, qx, qy, qz, eps2d, 2u * kBand64, C.b64, res64, sx, sy, sz);
            }
        }
    }
    part[w][0][lane] = sx + (double)ax; part[w][1][lane] = sy + (double)ay; part[w][2][lane] = sz + (double)az;
    __syncthreads();
    if (w != 0) return;
    sx = part[0][0][lane]; sy = part[0][1][lane]; sz = part[0][2][lane];
#pragma unroll
    for (int k = 1; k < K; k++) {
        sx += part[k][0][lane]; sy += part[k][1][lane]; sz += part[k][2][lane];
    }
    publish_maxabs(tab, valid ? integrate(tab, j, rank, sx, sy, sz, P, frozen) : 0.0);
}

// ---------------------------------------------------------------------------------------
// Stack walk (NBMI_WALK_STACK=1, prototype): the same 64 bodies per wave and the same per-lane opening test,
// but the wave keeps a stack of (cell, mask of the lanes that opened it) and expands a cell by visiting all
// its children back to back: their offsets come from ONE load of the child table, their records are requested
// together, the lanes taking part are a scalar mask (no per-visit `resume` compare / update) and there is no
// skip / descend decision per visit.  Accepted (body, node) sets are unchanged; sums associate differently.
// ---------------------------------------------------------------------------------------
constexpr int kStackCap = 320;  // > 7 pending siblings on each of the 43 possible levels
__global__ __launch_bounds__(kBlock) void k_walk_stack(const Node *__restrict__ nodes, const uint32_t *__restrict__ child_tab,
                                                       const WalkTable *tab, const TreeInfo *info_in,
                                                       const float4 *__restrict__ posm_s, const uint32_t *__restrict__ perm,
                                                       WalkParams P) {
    __shared__ unsigned st_off[kBlock / 64][kStackCap];
    __shared__ unsigned long long st_mask[kBlock / 64][kStackCap];
    const int lb = logical_block(blockIdx.x, gridDim.x, P.xcd_chunk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t rank = P.rank_begin + (int64_t)lb * blockDim.x + threadIdx.x;
    const bool valid = rank < P.rank_end;
    const bool frozen = info_in->error != 0 || info_in->sticky_error != 0;
    const unsigned band2 = __builtin_amdgcn_readfirstlane(info_in->band2);
    float px = 0.f, py = 0.f, pz = 0.f;
    uint32_t j = 0;
    if (valid) {
        const float4 p = posm_s[rank];
        px = p.x; py = p.y; pz = p.z;
        j = perm[rank];
    }
    const Body64 b64{tab, P.curbuf, j};
    float ax = 0.f, ay = 0.f, az = 0.f;
    int sp = 0;
    const unsigned long long lane_bit = 1ull << lane;

    // one node for the lanes in M: force for the lanes that take it, (offset, openers) pushed if anybody opens it
    auto visit = [&](unsigned off, unsigned long long M) {
        const Node nd = *reinterpret_cast<const Node *>(reinterpret_cast<const char *>(nodes) + off);
        const float dx = nd.cx - px, dy = nd.cy - py, dz = nd.cz - pz;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, P.eps2)));
        const bool in = (M & lane_bit) != 0ull;
        const int d2b = __float_as_int(d2), hi = __float_as_int(nd.s2t), lo = hi - (int)band2;
        bool geom = hi < d2b;
        if (in && !geom && lo < d2b) geom = (hi == 0) || exact_take(off, b64);
        const bool take = in && geom;
        const float inv = __builtin_amdgcn_rsqf(d2);
        const float f = take ? (nd.gm * inv) * (inv * inv) : 0.f;
        ax = fmaf(dx, f, ax); ay = fmaf(dy, f, ay); az = fmaf(dz, f, az);
        const unsigned long long openers = __builtin_amdgcn_ballot_w64(in && !geom);
        if (openers && sp < kStackCap) {
            if (lane == 0) { st_off[w][sp] = off; st_mask[w][sp] = openers; }
            sp++;
        }
    };

    if (!frozen) {
        visit(0u, __builtin_amdgcn_ballot_w64(valid));
        while (sp > 0) {
            sp--;
            const unsigned cell = __builtin_amdgcn_readfirstlane(st_off[w][sp]);
            const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)st_mask[w][sp]);
            const unsigned mhi = __builtin_amdgcn_readfirstlane((unsigned)(st_mask[w][sp] >> 32));
            const unsigned long long M = ((unsigned long long)mhi << 32) | mlo;
            const uint4 *row = reinterpret_cast<const uint4 *>(child_tab + 8 * (size_t)(cell / kNodeBytes));
            const uint4 t0 = row[0], t1 = row[1];
            const unsigned ch[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned c = __builtin_amdgcn_readfirstlane(ch[k]);
                if (c == 0u) break;
                visit(c, M);
            }
        }
    }
    if (!valid) return;
    integrate(tab, j, rank, ax, ay, az, P, frozen);
}

// ---------------------------------------------------------------------------------------
// Measurement only (NBMI_WALK_LANE=1): every lane walks on its own (own cursor, vector loads of the node
// record).  No lane ever idles for another's descent, but every visit is a 64-address gather.  Same
// accepted sets; kept as the yardstick for what a divergent visit costs on this chip.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_walk_lane(const Node *__restrict__ nodes, const WalkTable *tab,
                                                      const TreeInfo *info_in, const float4 *__restrict__ posm_s,
                                                      const uint32_t *__restrict__ perm, WalkParams P) {
    const int lb = logical_block(blockIdx.x, gridDim.x, P.xcd_chunk);
    const int64_t rank = P.rank_begin + (int64_t)lb * blockDim.x + threadIdx.x;
    if (rank >= P.rank_end) return;
    const bool frozen = info_in->error != 0 || info_in->sticky_error != 0;
    const unsigned nn = frozen ? 0u : ((unsigned)info_in->walk_nodes * kNodeBytes);
    const unsigned band2 = info_in->band2;
    const float4 p = posm_s[rank];
    const uint32_t j = perm[rank];
    const Body64 b64{tab, P.curbuf, j};
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned off = 0u;
    while (off < nn) {
        const char *q = reinterpret_cast<const char *>(nodes) + off;
        const float2 a0 = *reinterpret_cast<const float2 *>(q), a1 = *reinterpret_cast<const float2 *>(q + 8);
        const float2 b = *reinterpret_cast<const float2 *>(q + 16);
        const float4 a = make_float4(a0.x, a0.y, a1.x, a1.y);
        const float dx = a.x - p.x, dy = a.y - p.y, dz = a.z - p.z;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, P.eps2)));
        const int d2b = __float_as_int(d2), hi = __float_as_int(b.x), lo = hi - (int)band2;
        bool take = hi < d2b;
        if (!take && lo < d2b) take = (hi == 0) || exact_take(off, b64);
        const float inv = __builtin_amdgcn_rsqf(d2);
        const float f = take ? (a.w * inv) * (inv * inv) : 0.f;
        ax = fmaf(dx, f, ax); ay = fmaf(dy, f, ay); az = fmaf(dz, f, az);
        off = take ? __float_as_uint(b.y) : off + kNodeBytes;
    }
    integrate(tab, j, rank, ax, ay, az, P, frozen);
}

// ---------------------------------------------------------------------------------------
// Measurement only (NBMI_PREC=<mode>): the lock-step walk in C++ with the product walk's opening decisions
// (fp32 test, uncertainty band, float64 re-decision - the accepted sets are the product's) and a selectable
// arithmetic for the force of an accepted visit.  Answers "which rounding makes the 100-step error at 1 M
// bodies" (scripts/gpu_prec_diag.py).  diag64 = {cx, cy, cz, G m} of EVERY node in float64.
//   1  fp32 pair arithmetic on fp32-rounded coordinates, every contribution added to float64 sums (= the product's
//      arithmetic with ideal accumulation)
//   2  float64 throughout
//   3  float64 for leaves, 1 otherwise          4  float64 where fp32 dist_sq < near2, 1 otherwise
//   5  two-word coordinates (hi + lo floats of body and node), fp32 arithmetic
//   6  exact float64 differences rounded to fp32, then fp32 arithmetic (coordinate rounding removed altogether)
//   7  as 1 with fp32 running sums (no float64 accumulation at all)
//   8  as 6, one Newton step on the reciprocal square root
//   9  as 2 but G m rounded to fp32
//   10 ... 14  as 2 with ONE quantity rounded to fp32: the coordinate differences / dist_sq / the reciprocal square
//      root (v_rsq_f32) / [13: v_rsq_f32 seed + one Newton step in float64 - the candidate product form] / each
//      contribution before it is added
//   15 fp32 with the systematic errors removed (exact-residual Newton step on v_rsq_f32, G m as two floats), float64
//      accumulation;  16 the same on exact coordinate differences rounded to fp32;  17 = 16 with two-level fp32 sums
//   20 float64 (as 13) for the bodies inside the cylindrical radius NBMI_PREC_NEAR (length units), 1 for the others
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_walk_diag(const Node *__restrict__ nodes, const double4 *__restrict__ diag64,
                                                      const WalkTable *tab, const TreeInfo *info_in,
                                                      const float4 *__restrict__ posm_s, const uint32_t *__restrict__ perm,
                                                      WalkParams P) {
    const int64_t rank = P.rank_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = rank < P.rank_end;
    const bool frozen = info_in->error != 0 || info_in->sticky_error != 0;
    const unsigned nn = frozen ? 0u : ((unsigned)info_in->walk_nodes * kNodeBytes);
    const unsigned band2 = __builtin_amdgcn_readfirstlane(info_in->band2);
    float px = 0.f, py = 0.f, pz = 0.f;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    uint32_t j = 0;
    if (valid) {
        const float4 p = posm_s[rank];
        px = p.x; py = p.y; pz = p.z;
        j = perm[rank];
        const Bodies &cur = tab->buf[P.curbuf];
        qx = cur.x[j]; qy = cur.y[j]; qz = cur.z[j];
    }
    const float plx = (float)(qx - (double)px), ply = (float)(qy - (double)py), plz = (float)(qz - (double)pz);
    const Body64 b64{tab, P.curbuf, j};
    unsigned resume = valid ? 0u : 0xffffffffu;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    int nflush = 0;
    const int mode = P.prec;
    const double eps2d = tab->eps2;
    unsigned off = 0u;
    while (off < nn) {
        off = __builtin_amdgcn_readfirstlane(off);
        const Node nd = *reinterpret_cast<const Node *>(reinterpret_cast<const char *>(nodes) + off);
        const float dx = nd.cx - px, dy = nd.cy - py, dz = nd.cz - pz;
        const float dist_sq = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, P.eps2)));
        const bool active = resume <= off;
        const int d2b = __float_as_int(dist_sq), hi = __float_as_int(nd.s2t), lo = hi - (int)band2;
        bool geom = hi < d2b;
        if (active && !geom && lo < d2b) geom = (hi == 0) || exact_take(off, b64);
        const bool take = active && geom;
        if (take) {
            const bool leaf = hi == 0;
            const double rc2 = (double)P.near2;  // modes >= 20: bodies inside this cylindrical radius^2 (x, z) take float64
            const bool core = mode >= 20 && (qx * qx + qz * qz) < rc2;
            const bool use64 = mode == 2 || (mode >= 9 && mode <= 14) || (mode == 3 && leaf) || (mode == 4 && dist_sq < P.near2) || core;
            if (use64) {
                const double4 c = diag64[off / kNodeBytes];
                double ex = c.x - qx, ey = c.y - qy, ez = c.z - qz;
                if (mode == 10) { ex = (double)(float)ex; ey = (double)(float)ey; ez = (double)(float)ez; }
                double d2 = ex * ex + ey * ey + ez * ez + eps2d;
                if (mode == 11) d2 = (double)(float)d2;
                double inv;
                if (mode == 12) {
                    inv = (double)__builtin_amdgcn_rsqf((float)d2);
                } else if (mode == 13 || mode >= 20) {  // fp32 seed + one Newton step in float64
                    const double y0 = (double)__builtin_amdgcn_rsqf((float)d2);
                    const double e = fma(-d2 * y0, y0, 1.0);
                    inv = fma(0.5 * y0, e, y0);
                } else {
                    inv = 1.0 / sqrt(d2);
                }
                const double gm = mode == 9 ? (double)nd.gm : c.w;
                const double f = gm * inv * inv * inv;
                if (mode == 14) {
                    sx += (double)(float)(ex * f); sy += (double)(float)(ey * f); sz += (double)(float)(ez * f);
                } else {
                    sx += ex * f; sy += ey * f; sz += ez * f;
                }
            } else {
                float ex = dx, ey = dy, ez = dz;
                if (mode == 5) {
                    const double4 c = diag64[off / kNodeBytes];
                    const float clx = (float)(c.x - (double)nd.cx), cly = (float)(c.y - (double)nd.cy), clz = (float)(c.z - (double)nd.cz);
                    ex = dx + (clx - plx); ey = dy + (cly - ply); ez = dz + (clz - plz);
                } else if (mode == 6 || mode == 8 || mode == 16 || mode == 17) {
                    const double4 c = diag64[off / kNodeBytes];
                    ex = (float)(c.x - qx); ey = (float)(c.y - qy); ez = (float)(c.z - qz);
                }
                const float d2 = (mode == 5 || mode == 6 || mode == 8 || mode == 16 || mode == 17) ? fmaf(ez, ez, fmaf(ey, ey, fmaf(ex, ex, P.eps2))) : dist_sq;
                float inv = __builtin_amdgcn_rsqf(d2);
                if (mode == 8) inv = inv * fmaf(-0.5f * d2 * inv, inv, 1.5f);
                float f;
                if (mode >= 15 && mode <= 17) {
                    // "enhanced fp32": the SYSTEMATIC errors removed (one Newton step with the exact residual, G m as two
                    // floats), the random roundings of fp32 kept
                    const float t = d2 * inv, tl = fmaf(d2, inv, -t);
                    float e = fmaf(-t, inv, 1.0f);
                    e = fmaf(-tl, inv, e);
                    inv = fmaf(0.5f * inv, e, inv);
                    const double4 c = diag64[off / kNodeBytes];
                    const float gml = (float)(c.w - (double)nd.gm);
                    const float g3 = (inv * inv) * inv;
                    f = fmaf(gml, g3, nd.gm * g3);
                } else {
                    f = (nd.gm * inv) * (inv * inv);
                }
                if (mode == 7 || mode == 17) {
                    fx = fmaf(ex, f, fx); fy = fmaf(ey, f, fy); fz = fmaf(ez, f, fz);
                    if (mode == 17 && (++nflush & 15) == 0) {  // two-level sums like the product loop
                        sx += (double)fx; sy += (double)fy; sz += (double)fz;
                        fx = fy = fz = 0.f;
                    }
                } else {
                    sx += (double)(ex * f); sy += (double)(ey * f); sz += (double)(ez * f);
                }
            }
            resume = nd.next_off;
        }
        const unsigned long long any_open = __builtin_amdgcn_ballot_w64(active && !geom);
        off = any_open ? off + kNodeBytes : nd.next_off;
    }
    publish_maxabs(tab, valid ? integrate(tab, j, rank, sx + (double)fx, sy + (double)fy, sz + (double)fz, P, frozen) : 0.0);
}

// ---------------------------------------------------------------------------------------
// Direct O(N^2): a_i = sum_{j != i} G m_j d (|d|^2 + eps^2)^(-3/2)   (gpu_backend.py:145-240)
// 256-thread blocks, IB bodies per thread, 256-body tiles of {x,y,z,G m} staged in LDS and read
// back as wave-uniform broadcasts.  fp32 pair arithmetic, per-tile fp32 partial sums folded
// into float64 accumulators.  The j == i term is exactly zero when eps > 0 (d = 0); with
// kGuard (eps == 0) pairs at zero distance are skipped.  Fused update_bodies_cuda
// (gpu_backend.py:243-257): v = (v + a dt) * damping; x += v dt, new positions go to `nxt`.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_posm(Bodies cur, int64_t n, double G, float4 *__restrict__ posm) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    posm[i] = make_float4((float)cur.x[i], (float)cur.y[i], (float)cur.z[i], (float)(G * cur.m[i]));
}

// [r4] kUniform: every body has the same mass (the reference's presets set masses = 1: tools/presets.py), so G m is ONE number:
// it leaves the pair loop - f = inv^3 instead of G m inv^3, 12 vector instructions per pair instead of 13 - and multiplies
// the float64 sums once per body.
template <int IB, bool kGuard, bool kIntegrate, bool kUniform>
__global__ __launch_bounds__(kBlock) void k_direct(const float4 *__restrict__ posm, int64_t n, int64_t ibeg, int64_t iend,
                                                   float eps2, Bodies cur, Bodies nxt, double *__restrict__ acc_out,
                                                   double dt, double damping, double uniform_gm) {
    // bodies [ibeg, iend) (this launch's shard) against all n
    __shared__ float4 tile[kBlock];
    const int64_t i0 = ibeg + (int64_t)blockIdx.x * (kBlock * IB) + threadIdx.x;
    float px[IB], py[IB], pz[IB];
    double ax[IB], ay[IB], az[IB];
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const int64_t i = i0 + (int64_t)k * kBlock;
        const float4 p = i < iend ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        px[k] = p.x; py[k] = p.y; pz[k] = p.z;
        ax[k] = ay[k] = az[k] = 0.0;
    }
    const int64_t ntiles = (n + kBlock - 1) / kBlock;
    for (int64_t t = 0; t < ntiles; t++) {
        const int64_t j = t * kBlock + threadIdx.x;
        // pads of the last tile: zero mass - or, where the mass is not part of the pair arithmetic, so far away that
        // d^2 overflows to +inf and v_rsq_f32 returns exactly 0
        tile[threadIdx.x] = j < n ? posm[j] : (kUniform ? make_float4(1.0e20f, 1.0e20f, 1.0e20f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
        __syncthreads();
        float sx[IB], sy[IB], sz[IB];
#pragma unroll
        for (int k = 0; k < IB; k++) sx[k] = sy[k] = sz[k] = 0.f;
        // [r4] The j-bodies are fetched a chunk AHEAD: the next eight LDS reads are issued before the current eight bodies'
        // arithmetic, so no read is waited for where it is issued (the compiler's 8-unrolled form of the plain loop put an
        // s_waitcnt lgkmcnt(0) behind each of its last four ds_read_b128).  Same operations in the same order: results
        // unchanged bit for bit.  62.8 -> 73.3 TFLOP/s at 1 M bodies (318 -> 273 ms per step) (scripts/ubench/direct_valu_sched.hip:
        // 47.3 -> 41.2 cycles per 64 pairs per SIMD; chunks of 4: 41.9; more bodies per thread: no better).
        constexpr int CH = 8;
        float4 cur[CH], nxt[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) cur[c] = tile[c];
#pragma unroll 1
        for (int jj = 0; jj < kBlock; jj += CH) {
            const int nb = jj + CH < kBlock ? jj + CH : 0;
#pragma unroll
            for (int c = 0; c < CH; c++) nxt[c] = tile[nb + c];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const float4 q = cur[c];
#pragma unroll
                for (int k = 0; k < IB; k++) {
                    const float dx = q.x - px[k], dy = q.y - py[k], dz = q.z - pz[k];
                    const float r2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
                    const float inv = __builtin_amdgcn_rsqf(r2);
                    float f = kUniform ? inv * inv * inv : q.w * inv * inv * inv;
                    if (kGuard) f = (r2 > 0.f) ? f : 0.f;
                    sx[k] = fmaf(f, dx, sx[k]);
                    sy[k] = fmaf(f, dy, sy[k]);
                    sz[k] = fmaf(f, dz, sz[k]);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; c++) cur[c] = nxt[c];
        }
#pragma unroll
        for (int k = 0; k < IB; k++) {
            ax[k] += (double)sx[k]; ay[k] += (double)sy[k]; az[k] += (double)sz[k];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const int64_t i = i0 + (int64_t)k * kBlock;
        if (i >= iend) continue;
        if (kUniform) { ax[k] *= uniform_gm; ay[k] *= uniform_gm; az[k] *= uniform_gm; }
        if (kIntegrate) {
            const double vx = (cur.vx[i] + ax[k] * dt) * damping;
            const double vy = (cur.vy[i] + ay[k] * dt) * damping;
            const double vz = (cur.vz[i] + az[k] * dt) * damping;
            nxt.vx[i] = vx; nxt.vy[i] = vy; nxt.vz[i] = vz;
            nxt.x[i] = cur.x[i] + vx * dt;
            nxt.y[i] = cur.y[i] + vy * dt;
            nxt.z[i] = cur.z[i] + vz * dt;
            nxt.m[i] = cur.m[i];
            nxt.id[i] = cur.id[i];
        } else {
            const int64_t o = 3 * (int64_t)cur.id[i];
            acc_out[o] = ax[k]; acc_out[o + 1] = ay[k]; acc_out[o + 2] = az[k];
        }
    }
}

// ---------------------------------------------------------------------------------------
// colour ramp (simulation.py:320-400 == gpu_backend.py:259-325), float64 maths, f32 stores,
// rows written in the caller's body order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_colors(Bodies cur, int64_t n, double max_speed, bool by_rank,
                                                   float *__restrict__ colors) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const double vx = cur.vx[r], vy = cur.vy[r], vz = cur.vz[r];
    const double speed = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(vx, vx), __dmul_rn(vy, vy)), __dmul_rn(vz, vz)));
    double t = speed / max_speed;
    t = t > 1.0 ? 1.0 : t;
    double cr, cg, cb, s, s2;
    if (t < 0.55) {
        if (t < 0.15) {
            s = t / 0.15;
            cr = __dsub_rn(0.4, __dmul_rn(0.2, s)); cg = __dadd_rn(0.2, __dmul_rn(0.2, s)); cb = __dadd_rn(0.8, __dmul_rn(0.1, s));
        } else if (t < 0.30) {
            s = (t - 0.15) / 0.15;
            cr = __dadd_rn(0.2, __dmul_rn(0.1, s)); cg = __dadd_rn(0.4, __dmul_rn(0.1, s)); cb = __dadd_rn(0.9, __dmul_rn(0.05, s));
        } else {
            s = (t - 0.30) / 0.25;
            if (s < 0.6) {
                s2 = s / 0.6;
                cr = __dsub_rn(0.3, __dmul_rn(0.1, s2)); cg = __dadd_rn(0.5, __dmul_rn(0.3, s2)); cb = __dadd_rn(0.95, __dmul_rn(0.05, s2));
            } else {
                s2 = (s - 0.6) / 0.4;
                cr = __dadd_rn(0.2, __dmul_rn(0.8, s2)); cg = __dadd_rn(0.8, __dmul_rn(0.2, s2)); cb = 1.0;
            }
        }
    } else if (t < 0.90) {
        cr = 1.0; cg = 1.0; cb = 1.0;
    } else if (t < 0.95) {
        s = (t - 0.90) / 0.05;
        cr = 1.0; cg = __dsub_rn(1.0, __dmul_rn(0.05, s)); cb = __dsub_rn(1.0, __dmul_rn(1.0, s));
    } else if (t < 0.99) {
        s = (t - 0.95) / 0.04;
        cr = 1.0; cg = __dsub_rn(0.95, __dmul_rn(0.45, s)); cb = 0.0;
    } else {
        s = (t - 0.99) / 0.01;
        cr = 1.0; cg = __dsub_rn(0.5, __dmul_rn(0.5, s)); cb = 0.0;
    }
    const int64_t o = 3 * (by_rank ? r : (int64_t)cur.id[r]);  // owner mode: rows stay in rank order
    colors[o] = (float)cr; colors[o + 1] = (float)cg; colors[o + 2] = (float)cb;
}

// ---- un-permuting getters ------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_unperm3_f32(const double *__restrict__ a, const double *__restrict__ b,
                                                        const double *__restrict__ c, const int32_t *__restrict__ id,
                                                        int64_t n, float *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t o = 3 * (id ? (int64_t)id[r] : r);  // owner mode: rows stay in rank order
    out[o] = (float)a[r]; out[o + 1] = (float)b[r]; out[o + 2] = (float)c[r];
}
__global__ __launch_bounds__(kBlock) void k_unperm3_f64(const double *__restrict__ a, const double *__restrict__ b,
                                                        const double *__restrict__ c, const int32_t *__restrict__ id,
                                                        int64_t n, double *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t o = 3 * (id ? (int64_t)id[r] : r);
    out[o] = a[r]; out[o + 1] = b[r]; out[o + 2] = c[r];
}
// ---- frame codec on the device (tools/record.py:231-326) ------------------------------------------
// Format-2 payload of a .zstd frame: int16((cur - prev) * 1000) against the PREVIOUS DECODED frame
// (record.py:254-262; decoder :313-322), float32 arithmetic like NumPy's, C-cast wrap beyond +-32.767 kept
// [quirk].  `prev` (float32, caller's body order) lives in HBM and is advanced to what the decoder will
// reconstruct, prev + int16 / 1000, so only 6 bytes per body and array cross PCIe instead of 12.
__global__ __launch_bounds__(kBlock) void k_frame_delta(const float *__restrict__ cur, float *__restrict__ prev, int64_t count,
                                                        int16_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const float p = prev[i];
    const float t = __fmul_rn(__fsub_rn(cur[i], p), 1000.0f);
    // float -> int32 (truncation toward zero) -> low 16 bits: what ndarray.astype(np.int16) does on x86-64
    const int16_t q = (int16_t)(int)t;
    out[i] = q;
    prev[i] = __fadd_rn(p, __fdiv_rn((float)q, 1000.0f));
}

__global__ __launch_bounds__(kBlock) void k_split_state(const double *__restrict__ pos, const double *__restrict__ vel,
                                                        const double *__restrict__ mass, Bodies cur, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    cur.x[i] = pos[3 * i]; cur.y[i] = pos[3 * i + 1]; cur.z[i] = pos[3 * i + 2];
    cur.vx[i] = vel[3 * i]; cur.vy[i] = vel[3 * i + 1]; cur.vz[i] = vel[3 * i + 2];
    if (mass) cur.m[i] = mass[i];
    cur.id[i] = (int32_t)i;
}
// like k_split_state but keeps the current ordering: row of body id[r] goes to rank r
__global__ __launch_bounds__(kBlock) void k_set_state_perm(const double *__restrict__ pos, const double *__restrict__ vel,
                                                           Bodies cur, int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t i = cur.id[r];
    cur.x[r] = pos[3 * i]; cur.y[r] = pos[3 * i + 1]; cur.z[r] = pos[3 * i + 2];
    cur.vx[r] = vel[3 * i]; cur.vy[r] = vel[3 * i + 1]; cur.vz[r] = vel[3 * i + 2];
}
// the reference's octant digits back from the sort key: `levels` digits of (hi, lo) decoded (hilbert.h), the rest 0
__device__ __forceinline__ void octant_digits(uint64_t hi, uint64_t lo, int levels, uint64_t &ohi, uint64_t &olo) {
    unsigned st = 0u;
    ohi = 0ull; olo = 0ull;
    for (int l = 0; l < levels; l++) {
        const uint64_t w = l < 21 ? hi : lo;
        const unsigned d = (unsigned)(w >> (3 * (20 - (l < 21 ? l : l - 21)))) & 7u;
        const unsigned oct = (nbmi::kHilOctant[st] >> (3u * d)) & 7u;
        st = (unsigned)(nbmi::kHilNextByDigit[st] >> (5u * d)) & 31u;
        if (l < 21) ohi |= (uint64_t)oct << (3 * (20 - l));
        else olo |= (uint64_t)oct << (3 * (41 - l));
    }
}
// keys in the caller's body order; decode: as the reference's octant-path digits, else the raw sort keys
__global__ __launch_bounds__(kBlock) void k_keys_to_orig(const uint64_t *__restrict__ hi_s, const uint64_t *__restrict__ lo_s,
                                                         const uint32_t *__restrict__ perm, const int32_t *__restrict__ id,
                                                         int64_t n, int decode, uint64_t *__restrict__ out_hi,
                                                         uint64_t *__restrict__ out_lo) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t o = id[perm[r]];
    uint64_t h = hi_s[r], l = lo_s[r];
    if (decode) octant_digits(h, l, kMaxLevel, h, l);
    out_hi[o] = h;
    out_lo[o] = l;
}
__global__ __launch_bounds__(kBlock) void k_order(const uint32_t *__restrict__ perm, const int32_t *__restrict__ id,
                                                  int64_t n, int32_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r < n) out[r] = id[perm[r]];
}

__global__ __launch_bounds__(kBlock) void k_cells(const int32_t *__restrict__ node_ref, const uint8_t *__restrict__ node_level,
                                                  const uint64_t *__restrict__ hi_s, int64_t num_nodes, int decode,
                                                  int32_t *__restrict__ level, uint64_t *__restrict__ key) {
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (u >= num_nodes) return;
    const int r = node_ref[u];  // first body of the node's range
    const int lev = node_level[u];
    level[u] = lev;
    uint64_t h = hi_s[r], l = 0ull;
    if (decode && lev <= 21) octant_digits(h, 0ull, lev, h, l);  // the cell's path in the reference's octant digits
    key[u] = lev <= 21 ? (lev == 0 ? 0ull : (h >> (63 - 3 * lev))) : ~0ull;
}
// multi-GPU row pack / unpack: {x,y,z,vx,vy,vz,m,id}
__global__ __launch_bounds__(kBlock) void k_pack_rows(Bodies cur, int64_t begin, int64_t end, double *__restrict__ rows) {
    const int64_t r = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= end) return;
    double *o = rows + 8 * (r - begin);
    o[0] = cur.x[r]; o[1] = cur.y[r]; o[2] = cur.z[r];
    o[3] = cur.vx[r]; o[4] = cur.vy[r]; o[5] = cur.vz[r];
    o[6] = cur.m[r]; o[7] = (double)cur.id[r];
}
__global__ __launch_bounds__(kBlock) void k_unpack_rows(Bodies cur, int64_t begin, int64_t end, const double *__restrict__ rows) {
    const int64_t r = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= end) return;
    const double *o = rows + 8 * (r - begin);
    cur.x[r] = o[0]; cur.y[r] = o[1]; cur.z[r] = o[2];
    cur.vx[r] = o[3]; cur.vy[r] = o[4]; cur.vz[r] = o[5];
    cur.m[r] = o[6]; cur.id[r] = (int32_t)o[7];
}

inline int nblocks(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

namespace {
// =========================================================================================
// Multi-GPU stage 2 ("owner mode"): every rank OWNS the bodies of one octant-key range, builds the octree
// of its own bodies inside the GLOBAL root cube and receives from every other rank only the part of that
// rank's tree its own bodies can open (a locally essential tree).  Kernels for: the key samples the
// splitters come from, the destination of every body, the body bounding box, and the extraction /
// appending of locally essential trees.  The collectives themselves are the host framework's (RCCL).
// =========================================================================================
constexpr int kMaxWorld = 64;
constexpr int kSampleCap = 4096;  // world x samples_per_rank, ranked in LDS

// `nvalid` regular samples of the (nearly key-ordered) local keys; the other slots are all ones (ignored: they sort to
// the very end).  [r3] A rank emits samples IN PROPORTION to the bodies it holds: with the same number from every rank
// the pooled samples describe "one W-th of the bodies per rank" whatever the ranks actually hold, equal quantiles then
// reproduce the current counts, and an imbalance, once there, stays (measured: 27 k ... 49 k bodies per rank after 500
// steps of a 300 k-body collision on 8 ranks).
__global__ __launch_bounds__(kBlock) void k_key_samples(const uint64_t *__restrict__ key_hi, int64_t n, int nsamples, int nvalid,
                                                        uint64_t *__restrict__ out) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nsamples) return;
    out[k] = (n > 0 && k < nvalid) ? key_hi[(int64_t)((2 * (int64_t)k + 1) * n / (2 * (int64_t)nvalid))] : ~0ull;
}

// world - 1 splitters at equal quantiles of the valid samples: every sample finds its rank among all of them by
// counting (all samples in LDS, broadcast reads), and the ones whose rank is a quantile write themselves.
// Rank j owns the keys in [split[j-1], split[j]).  Any number of workgroups (one sample per thread).
__global__ __launch_bounds__(kBlock) void k_splitters(const uint64_t *__restrict__ samples, int total, int world,
                                                      uint64_t *__restrict__ split) {
    __shared__ uint64_t a[kSampleCap];
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < total; i += kBlock) {
        const uint64_t v = samples[i];
        a[i] = v;
        cnt += v != ~0ull ? 1 : 0;
    }
    if (cnt) atomicAdd(&s_valid, cnt);
    __syncthreads();
    const int valid = s_valid;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (valid == 0) {
        if (i < world - 1) split[i] = ~0ull;
        return;
    }
    if (i >= total) return;
    const uint64_t x = a[i];
    if (x == ~0ull) return;
    int rank = 0;
#pragma unroll 16
    for (int j = 0; j < total; j++) {
        const uint64_t y = a[j];
        rank += (y < x || (y == x && j < i)) ? 1 : 0;
    }
    for (int j = 0; j < world - 1; j++)
        if ((int)((int64_t)(j + 1) * valid / world) == rank) split[j] = x;
}

// destination rank of every body: number of splitters <= its upper key word (bodies that agree on all 21
// upper digits always travel together)
__global__ __launch_bounds__(kBlock) void k_dest(const uint64_t *__restrict__ key_hi, int64_t n, const uint64_t *__restrict__ split,
                                                 int world, uint32_t *__restrict__ dest, uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key_hi[i];
    int lo = 0, hi = world - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (split[mid] <= k) lo = mid + 1; else hi = mid; }
    dest[i] = (uint32_t)lo;
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void k_dead_flags(const uint8_t *__restrict__ dead, int64_t n, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) flag[i] = dead[i] ? 1 : 0;
}
// owner mode, migration: flag row j of destination d = 1 if body i goes to rank d (d != me); dead[i] = leaves
__global__ __launch_bounds__(kBlock) void k_emigrant_flags(const uint32_t *__restrict__ dest, int64_t n, int world, int me,
                                                           int64_t stride, int32_t *__restrict__ flag, uint8_t *__restrict__ dead) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int d = (int)dest[i];
    for (int j = 0; j < world; j++) flag[(int64_t)j * stride + i] = (j == d && j != me) ? 1 : 0;
    dead[i] = d != me ? 1 : 0;
}
// emigrants' rows, grouped by destination (segment offsets from k_let_counts), current order inside a group
__global__ __launch_bounds__(kBlock) void k_pack_emigrants(Bodies cur, const uint32_t *__restrict__ dest, int64_t n, int me,
                                                           int64_t stride, const int32_t *__restrict__ slot,
                                                           const int64_t *__restrict__ seg_off, double *__restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int d = (int)dest[i];
    if (d == me) return;
    double *o = rows + 8 * (seg_off[d] + slot[(int64_t)d * stride + i]);
    o[0] = cur.x[i]; o[1] = cur.y[i]; o[2] = cur.z[i];
    o[3] = cur.vx[i]; o[4] = cur.vy[i]; o[5] = cur.vz[i];
    o[6] = cur.m[i]; o[7] = (double)cur.id[i];
}

// Where a rank's bodies are, for the pruning of the trees the others send it: tight bounding boxes of the
// bodies inside octree cells of the rank's OWN tree - the cells of level kBoxLevel, refined down to
// kBoxLevelMax along the two cells that hold the rank's first and last body (a rank's key range ends in the
// middle of cells; unrefined, those two boxes would overlap the neighbours' whole working set).  (Bounding
// boxes of equal chunks of the key order do not work: a chunk that crosses a high-level cell boundary spans
// two distant corners of the system, and one such box keeps everything alive.)  The boxes come out in key
// order (slot = exclusive scan of the emit flags over the pre-order array), so every kSuper consecutive ones
// are neighbours in space and get a common "super box" for a two-level test.
constexpr int kChainLevels = 43;  // cell levels 0 .. 42 (21 digits of the upper key word, 21 of the lower)
constexpr int kBoxLevel = 4, kBoxLevelMax = 11;
constexpr int kBoxesPerRank = 2048, kSuper = 32, kSupersPerRank = kBoxesPerRank / kSuper;
constexpr int kMegasPerRank = 8, kSupersPerMega = kSupersPerRank / kMegasPerRank;

// flag[i] = 1 if node i is one of the cells / leaves whose bodies get a box
__global__ __launch_bounds__(kBlock) void k_box_flags(const Node *__restrict__ nodes, const uint8_t *__restrict__ node_level,
                                                      const int32_t *__restrict__ node_ref, const uint64_t *__restrict__ hi_s,
                                                      const uint64_t *__restrict__ lo_s, const TreeInfo *__restrict__ info,
                                                      int64_t n, int64_t rows, int32_t *__restrict__ flag,
                                                      int32_t *__restrict__ chainR, int64_t link_base) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= rows) return;  // the grid is rounded up to whole blocks: nothing behind the row budget is written
    const int64_t num_nodes = info->error ? 0 : info->num_nodes;
    if (i >= num_nodes) {
        flag[i] = 0;  // launched for the row budget: rows behind the tree count nothing
        return;
    }
    const int l = node_level[i];
    int f = 0;
    {   // [r3] the cells whose subtree ends with the array hold the rank's last body: one per level (k_chain_table)
        const Node nd = nodes[i];
        if (__float_as_int(nd.s2t) != 0 && (int64_t)(nd.next_off / kNodeBytes) - link_base == num_nodes && l < kChainLevels) chainR[l] = (int32_t)i;
    }
    if (l <= kBoxLevelMax) {
        const Node nd = nodes[i];
        const bool leaf = __float_as_int(nd.s2t) == 0;
        const int64_t a = node_ref[i];
        const int64_t nx = (int64_t)(nd.next_off / kNodeBytes) - link_base;
        const int64_t b = nx < num_nodes ? node_ref[nx] : n;
        const bool boundary = a == 0 || b == n;  // holds the rank's first or last body
        if (l < kBoxLevel) {
            f = leaf ? 1 : 0;
        } else if (l == kBoxLevel) {
            f = (!boundary || leaf) ? 1 : 0;
        } else {
            // only below the two boundary chains: the parent (level l - 1) holds body 0 or body n - 1
            const uint64_t h = hi_s[a], lo = lo_s[a];
            const bool under = cpl_digits(h, lo, hi_s[0], lo_s[0]) >= l - 1 || cpl_digits(h, lo, hi_s[n - 1], lo_s[n - 1]) >= l - 1;
            f = (under && (!boundary || leaf || l == kBoxLevelMax)) ? 1 : 0;
        }
    }
    flag[i] = f;
}
// body ranges of the flagged nodes, in key order
__global__ __launch_bounds__(kBlock) void k_box_ranges(const Node *__restrict__ nodes, const int32_t *__restrict__ node_ref,
                                                       const int32_t *__restrict__ flag, const int32_t *__restrict__ slot,
                                                       const TreeInfo *__restrict__ info, int64_t rows, int64_t n,
                                                       int32_t *__restrict__ ranges /* 2 x kBoxesPerRank */, int64_t link_base) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t num_nodes = info->error ? 0 : info->num_nodes;
    if (i >= num_nodes || !flag[i]) return;
    const int k = slot[i];
    if (k >= kBoxesPerRank) return;  // more cells than boxes: the last box takes everything from its cell on
    const int64_t nx = (int64_t)(nodes[i].next_off / kNodeBytes) - link_base;
    ranges[2 * k] = node_ref[i];
    ranges[2 * k + 1] = (k == kBoxesPerRank - 1 && slot[rows] > kBoxesPerRank) ? (int32_t)n
                                                                                    : (int32_t)(nx < num_nodes ? node_ref[nx] : n);
}
// tight bounding boxes of the bodies of every range (empty box: lo = +inf > hi = -inf).  The ranges are
// consecutive in key order and very unequal (a level-4 cell of a galaxy core holds 10^5 bodies, a refined
// boundary cell a handful), so the work is cut by BODIES: a wave takes kBoxChunk consecutive bodies, finds the
// ranges that overlap them and adds its part of each with float64 atomic min / max.
constexpr int kBoxChunk = 1024;
__global__ void k_boxes_init(double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * kBoxesPerRank) out[i] = (i % 6) < 3 ? INFINITY : -INFINITY;
}
__global__ __launch_bounds__(kBlock) void k_range_boxes(const double4 *__restrict__ p64_s, const int32_t *__restrict__ ranges,
                                                        const int32_t *__restrict__ total, int64_t n, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t c0 = wave * kBoxChunk, c1 = c0 + kBoxChunk < n ? c0 + kBoxChunk : n;
    if (c0 >= n) return;
    const int nb = *total < kBoxesPerRank ? *total : kBoxesPerRank;
    int lo = 0, hi = nb;  // first range that ends behind c0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)ranges[2 * mid + 1] > c0) hi = mid; else lo = mid + 1; }
    for (int k = lo; k < nb; k++) {
        const int64_t r0 = ranges[2 * k], r1 = ranges[2 * k + 1];
        if (r0 >= c1) break;
        const int64_t a = r0 > c0 ? r0 : c0, b = r1 < c1 ? r1 : c1;
        double v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int64_t i = a + lane; i < b; i += 64) {
            const double4 q = p64_s[i];
            v[0] = fmin(v[0], q.x); v[1] = fmin(v[1], q.y); v[2] = fmin(v[2], q.z);
            v[3] = fmax(v[3], q.x); v[4] = fmax(v[4], q.y); v[5] = fmax(v[5], q.z);
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double t = __shfl_xor(v[c], o);
                v[c] = c < 3 ? fmin(v[c], t) : fmax(v[c], t);
            }
        }
        if (lane < 6 && a < b) {
            double r = v[0];
#pragma unroll
            for (int c = 1; c < 6; c++) r = lane == c ? v[c] : r;
            if (lane < 3) atomicMin(&out[6 * k + lane], r);
            else atomicMax(&out[6 * k + lane], r);
        }
    }
}
// super box = union of kSuper consecutive boxes (of every rank)
__global__ void k_super_boxes(const double *__restrict__ boxes, int nsupers, double *__restrict__ supers) {
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nsupers) return;
    double v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < kSuper; k++) {
        const double *b = boxes + 6 * ((int64_t)sidx * kSuper + k);
        if (!(b[0] <= b[3])) continue;
        for (int c = 0; c < 3; c++) { v[c] = fmin(v[c], b[c]); v[3 + c] = fmax(v[3 + c], b[3 + c]); }
    }
    for (int c = 0; c < 6; c++) supers[6 * sidx + c] = v[c];
}

// ---- plain int32 exclusive scan (three phases, like the moment scan); blockIdx.y selects one of several
// equally long arrays `stride` elements apart (one per destination rank) ---------------------------------
__global__ __launch_bounds__(kBlock) void k_iscan_reduce(const int32_t *__restrict__ in, int64_t n, int64_t stride,
                                                         int32_t *__restrict__ tile_sum, int64_t tstride) {
    __shared__ int red[kBlock / 64];
    in += (int64_t)blockIdx.y * stride;
    tile_sum += (int64_t)blockIdx.y * tstride;
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
        acc += i < n ? in[i] : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; w++) t += red[w];
        tile_sum[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(kBlock) void k_iscan_tiles(int32_t *__restrict__ tile_sum, int64_t ntiles, int64_t tstride) {
    __shared__ int wsum[kBlock / 64];
    __shared__ int carry_s;
    tile_sum += (int64_t)blockIdx.y * tstride;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < ntiles; base += kBlock) {
        const int64_t i = base + threadIdx.x;
        const int own = i < ntiles ? tile_sum[i] : 0;
        int v = own;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d);
            if (lane >= d) v += o;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < w; k++) off += wsum[k];
        if (i < ntiles) tile_sum[i] = off + v - own;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = off + v;
        __syncthreads();
    }
}
// out[i] = sum of in[0..i); entry n receives the total
__global__ __launch_bounds__(kBlock) void k_iscan_apply(const int32_t *__restrict__ in, int64_t n, int64_t stride,
                                                        const int32_t *__restrict__ tile_off, int64_t tstride,
                                                        int32_t *__restrict__ out) {
    __shared__ int wsum[kBlock / 64];
    in += (int64_t)blockIdx.y * stride;
    out += (int64_t)blockIdx.y * stride;
    tile_off += (int64_t)blockIdx.y * tstride;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { v[k] = base + k < n ? in[base + k] : 0; sum += v[k]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = tile_off[blockIdx.x] + inc - sum;
    for (int k = 0; k < w; k++) run += wsum[k];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (base + k <= n) out[base + k] = run;
        run += v[k];
    }
}

// ---- locally essential tree -----------------------------------------------------------------
// [r3] One GLOBAL octree, cut into the ranks' pieces.  A rank's build (k_emit_tile) makes the octree of its own
// bodies; the octree of ALL bodies differs from the ranks' trees side by side only along the rank boundaries:
//   * cells that contain a rank's last body AND bodies of higher ranks: one cell in the global tree (first body on
//     some rank A, the "owner"), a partial copy in every rank it reaches into;
//   * cells that exist only because a rank's last body and the next rank's first body share more key digits than
//     either shares with its neighbour on its own rank ("boundary-born" cells).
// Round 2 walked the partial copies as they were - valid Barnes-Hut, but a DIFFERENT approximation than the
// single-GPU tree: 2e-6 of the largest coordinate after 10 steps, 1.2e-4 after 50, 1.7e-3 after 100 at 1 M bodies
// on 8 ranks, whatever the force precision (profiles/r03_owner_100_steps.jsonl).  Now every rank publishes a small
// table about its two ends (ChainTable, all-gathered with the bounding boxes), and k_chain_fix turns the own tree
// into the rank's piece of the global pre-order array:
//   * the copies of cells that begin on a lower rank (always the FIRST own_skip nodes of the array) are dropped;
//   * the cells along the path to the last body that reach into higher ranks get the global moments, the
//     boundary-born ones are inserted in front of the last leaf (the last node of the array: nothing moves);
//   * skip links that leave the rank name (rank, node) and are resolved where the pieces are put together.
// Concatenated in rank order the pieces ARE the global pre-order array: every rank walks its own piece in full and a
// pruned copy of the others', in rank order, i.e. the reference's accepted sets in the reference's order.
constexpr unsigned kLinkTag = 0xFFFFFF00u;  // next_off >= this: a link that leaves the rank, slot = low byte (k_chain_fix)
constexpr int kLinkEnd = 0xFE, kLinkLocal = 0xFF;
struct ChainTable {
    long long n;       // bodies of the rank (0: every other field is meaningless)
    long long nodes;   // nodes of its tree as built (before k_chain_fix)
    unsigned long long first_hi, first_lo, last_hi, last_lo;  // keys of its first / last body
    long long d0, dR;  // common key digits of its first two / last two bodies (-1: a single body)
    // level by level: how many of the rank's FIRST bodies share `level` digits with its first body (Lc), the node
    // behind them in the rank's array (Ln; = nodes if all do) and their moments (Ls); the same from the end (Rc, Rs)
    long long Lc[kChainLevels], Ln[kChainLevels], Rc[kChainLevels];
    double Ls[kChainLevels][4], Rs[kChainLevels][4];
};
static_assert(sizeof(ChainTable) % 8 == 0, "ChainTable travels as doubles");
constexpr int kChainDoubles = (int)(sizeof(ChainTable) / 8);

__device__ __forceinline__ int chain_prev(const ChainTable *T, int x) {
    for (int p = x - 1; p >= 0; p--) if (T[p].n > 0) return p;
    return -1;
}
__device__ __forceinline__ int chain_next(const ChainTable *T, int W, int x) {
    for (int z = x + 1; z < W; z++) if (T[z].n > 0) return z;
    return -1;
}
// key digits the last body of rank x shares with the first body of rank y
__device__ __forceinline__ int chain_cpl(const ChainTable *T, int x, int y) {
    return cpl_digits(T[x].last_hi, T[x].last_lo, T[y].first_hi, T[y].first_lo);
}
// how many of rank x's first nodes are copies of cells that begin on a lower rank
__device__ __forceinline__ int chain_skip(const ChainTable *T, int x) {
    const int p = chain_prev(T, x);
    if (p < 0) return 0;
    const int db = chain_cpl(T, p, x), d0 = (int)T[x].d0;
    return (db < d0 ? db : d0) + 1;
}
struct ChainCell {
    double M, mx, my, mz;
    int link_rank;        // rank on which the cell's subtree ends (kLinkEnd: with the last body of the last rank)
    long long link_node;  // ... and the node (in that rank's numbering as built) that follows it
};
// The level-`lev` cell that holds the LAST body of rank x and reaches into the next rank.  The same function with
// the same gathered tables on every rank that needs the cell (its owner for the record, the ranks it reaches into
// for their pruning decisions): bit-identical moments everywhere.
__device__ ChainCell chain_cell(const ChainTable *T, int W, int x, int lev) {
    int a = x;  // the owner: leftwards while the whole rank lies inside the cell and the cell comes from further left
    while (T[a].Rc[lev] == T[a].n) {
        const int p = chain_prev(T, a);
        if (p < 0 || chain_cpl(T, p, a) < lev) break;
        a = p;
    }
    ChainCell c;
    c.M = T[a].Rs[lev][0]; c.mx = T[a].Rs[lev][1]; c.my = T[a].Rs[lev][2]; c.mz = T[a].Rs[lev][3];
    int y = a;
    for (;;) {
        const int z = chain_next(T, W, y);
        if (z < 0) { c.link_rank = kLinkEnd; c.link_node = 0; break; }
        if (chain_cpl(T, y, z) < lev) { c.link_rank = z; c.link_node = chain_skip(T, z); break; }
        c.M += T[z].Ls[lev][0]; c.mx += T[z].Ls[lev][1]; c.my += T[z].Ls[lev][2]; c.mz += T[z].Ls[lev][3];
        if (T[z].Lc[lev] < T[z].n) { c.link_rank = z; c.link_node = T[z].Ln[lev]; break; }
        y = z;
    }
    return c;
}
// what follows a rank's last leaf in the global array: the first node the next rank keeps
__device__ __forceinline__ void chain_after(const ChainTable *T, int W, int x, int &link_rank, long long &link_node) {
    const int z = chain_next(T, W, x);
    link_rank = z < 0 ? kLinkEnd : z;
    link_node = z < 0 ? 0 : chain_skip(T, z);
}

// the rank's table (one wave, lane = level).  chainR[l] = node of the level-l cell that holds the last body
// (k_box_flags notes them: the cells whose subtree ends with the array).
__global__ void k_chain_table(const Node *__restrict__ nodes, const int32_t *__restrict__ node_ref, const int32_t *__restrict__ delta,
                              const double4 *__restrict__ S, const Moment *__restrict__ T, const uint64_t *__restrict__ hi_s,
                              const uint64_t *__restrict__ lo_s, int64_t n, const TreeInfo *__restrict__ info,
                              const int32_t *__restrict__ chainR, int64_t link_base, ChainTable *__restrict__ out) {
    const int lev = threadIdx.x;
    const int64_t N = info->error ? 0 : info->num_nodes;
    if (N == 0) n = 0;  // (a build that overflowed: the rank takes no part in this step's global tree; the error is sticky)
    const int d0 = n >= 2 ? delta[0] : -1, dR = n >= 2 ? delta[n - 2] : -1;
    if (lev == 0) {
        out->n = n; out->nodes = N;
        out->first_hi = n > 0 ? hi_s[0] : 0; out->first_lo = n > 0 ? lo_s[0] : 0;
        out->last_hi = n > 0 ? hi_s[n - 1] : 0; out->last_lo = n > 0 ? lo_s[n - 1] : 0;
        out->d0 = d0; out->dR = dR;
    }
    if (lev >= kChainLevels || n == 0) return;
    int64_t Lc, Ln, Rc;
    if (lev <= d0) {  // node `lev` is the level-lev cell of the first body (its cells are the first nodes of the array)
        const int64_t nx = (int64_t)(nodes[lev].next_off / kNodeBytes) - link_base;
        Lc = nx < N ? node_ref[nx] : n;
        Ln = nx;
    } else {
        Lc = 1;
        Ln = n >= 2 ? d0 + 2 : N;
    }
    if (lev <= dR) Rc = n - node_ref[chainR[lev]];
    else Rc = 1;
    out->Lc[lev] = Lc; out->Ln[lev] = Ln; out->Rc[lev] = Rc;
    double M, mx, my, mz;
    range_moments(S, T, 0, Lc, M, mx, my, mz);
    out->Ls[lev][0] = M; out->Ls[lev][1] = mx; out->Ls[lev][2] = my; out->Ls[lev][3] = mz;
    range_moments(S, T, n - Rc, n, M, mx, my, mz);
    out->Rs[lev][0] = M; out->Rs[lev][1] = mx; out->Rs[lev][2] = my; out->Rs[lev][3] = mz;
}

struct OwnLink { int node; int rank; long long target; };  // node of the own array, (rank, node there) it links to
constexpr int kOwnLinks = kChainLevels + 1;                  // a slot per level, one for the last leaf

// The own tree becomes the rank's piece of the global array (see above).  One workgroup of 64, lane = level.
__global__ void k_chain_fix(const ChainTable *__restrict__ T, int W, int me, Node *__restrict__ nodes, Node64 *__restrict__ n64,
                            NodeD *__restrict__ nodesd, uint8_t *__restrict__ node_level, int32_t *__restrict__ node_ref,
                            const int32_t *__restrict__ chainR, double inv_theta2, int64_t capacity, OwnLink *__restrict__ links,
                            TreeInfo *info) {
    const int lev = threadIdx.x;
    const ChainTable &t = T[me];
    const int64_t n = t.n, N = t.nodes;
    if (lev < kOwnLinks) links[lev] = OwnLink{-1, 0, 0};
    if (n == 0 || info->error) {
        if (lev == 0) { info->own_skip = 0; info->own_added = 0; }
        return;
    }
    const int p = chain_prev(T, me), z = chain_next(T, W, me);
    const int dprev = p >= 0 ? chain_cpl(T, p, me) : -1, dnext = z >= 0 ? chain_cpl(T, me, z) : -1;
    const int dR = (int)t.dR;
    const int lo_new = n == 1 ? (dprev > dR ? dprev : dR) : dR;  // levels above this one and up to dnext are boundary-born
    const int c = dnext > lo_new ? dnext - lo_new : 0;
    if (N + c + 1 > capacity) {
        if (lev == 0) {
            info->error = 1;
            if (info->sticky_error == 0) { info->sticky_error = 1; info->sticky_nodes = N + c; }
        }
        return;
    }
    // the last leaf first (the new cells take its place)
    const Node leaf = nodes[N - 1];
    const uint8_t leaf_level = node_level[N - 1];
    NodeD leafd = NodeD{0.0, 0.0, 0.0, 0.0, 0.0f, 0u};
    if (nodesd) leafd = nodesd[N - 1];
    __syncthreads();
    const bool exists = lev < kChainLevels && lev <= dR, born = lev < kChainLevels && lev > lo_new && lev <= dnext;
    if (exists || born) {
        const int64_t i = exists ? chainR[lev] : (N - 1) + (lev - lo_new - 1);
        const bool from_left = t.Rc[lev] == n && dprev >= lev;  // begins on a lower rank: that rank's to describe
        if (!from_left) {
            int lr; long long ln;
            if (lev <= dnext) {
                const ChainCell cc = chain_cell(T, W, me, lev);
                lr = cc.link_rank; ln = cc.link_node;
                double cx = 0.0, cy = 0.0, cz = 0.0;
                if (cc.M > 0.0) { cx = cc.mx / cc.M; cy = cc.my / cc.M; cz = cc.mz / cc.M; }
                const double bounds = info->bounds;
                const double size = ldexp(bounds, 1 - lev);
                const float s2t = (float)(size * size * inv_theta2);
                Node nd;
                nd.cx = (float)cx; nd.cy = (float)cy; nd.cz = (float)cz; nd.gm = (float)cc.M;
                nd.s2t = __int_as_float(__float_as_int(s2t) + (int)(info->band2 >> 1));
                nd.next_off = kLinkTag | (unsigned)lev;
                nodes[i] = nd;
                n64[i] = Node64{cx, cy, cz, ldexp(bounds, -lev)};
                if (nodesd) nodesd[i] = NodeD{cx, cy, cz, cc.M, __int_as_float(__float_as_int(s2t) + (int)kBand64), kLinkTag | (unsigned)lev};
                if (born) { node_level[i] = (uint8_t)lev; node_ref[i] = (int32_t)(n - 1); }
            } else {  // ends with the rank's last body: only its skip link leaves the rank
                chain_after(T, W, me, lr, ln);
                nodes[i].next_off = kLinkTag | (unsigned)lev;
                if (nodesd) nodesd[i].next_off = kLinkTag | (unsigned)lev;
            }
            links[lev] = OwnLink{(int)i, lr, ln};
        }
    }
    if (lev == kChainLevels) {  // the last leaf, behind the boundary-born cells
        const int64_t i = N - 1 + c;
        Node lf = leaf;
        lf.next_off = kLinkTag | (unsigned)kChainLevels;
        nodes[i] = lf;
        if (nodesd) { leafd.next_off = kLinkTag | (unsigned)kChainLevels; nodesd[i] = leafd; }
        node_ref[i] = (int32_t)(n - 1);
        node_level[i] = leaf_level + (uint8_t)c;
        int lr; long long ln;
        chain_after(T, W, me, lr, ln);
        links[kChainLevels] = OwnLink{(int)i, lr, ln};
        info->num_nodes = N + c;
        info->own_skip = chain_skip(T, me);
        info->own_added = c;
    }
}

// A cell can only be opened by a body of another rank if the opening test can fail somewhere in one of that
// rank's bounding boxes; if it cannot (for any other rank), nobody else ever looks below it and its subtree
// stays home.  Conservative by a 1e-9 margin on both sides of the float64 test.  diff[] marks the dropped
// pre-order ranges (+1 at the first node of the subtree, -1 behind it): a node is kept iff the running sum
// over diff up to and including it is zero.
__device__ __forceinline__ bool box_may_open(const double *__restrict__ b, const Node64 &c, double eps2, double thr) {
    if (!(b[0] <= b[3])) return false;  // empty
    const double dx = fmax(0.0, fmax(b[0] - c.cx, c.cx - b[3]));
    const double dy = fmax(0.0, fmax(b[1] - c.cy, c.cy - b[4]));
    const double dz = fmax(0.0, fmax(b[2] - c.cz, c.cz - b[5]));
    return (dx * dx + dy * dy + dz * dz + eps2) * (1.0 - 1e-9) <= thr;  // some point of the box may fail "size / dist < theta"
}
// may a body of rank j open the cell?  Four-level test: the union box of the rank, its mega boxes, super boxes, boxes.
__device__ __forceinline__ bool rank_may_open(int j, const Node64 &c, double eps2, double thr, const double *__restrict__ boxes,
                                              const double *__restrict__ supers, const double *__restrict__ megas,
                                              const double *__restrict__ rankbox) {
    if (!box_may_open(rankbox + 6 * j, c, eps2, thr)) return false;
    for (int g = j * kMegasPerRank; g < (j + 1) * kMegasPerRank; g++) {
        if (!box_may_open(megas + 6 * g, c, eps2, thr)) continue;
        for (int sb = g * kSupersPerMega; sb < (g + 1) * kSupersPerMega; sb++) {
            if (!box_may_open(supers + 6 * sb, c, eps2, thr)) continue;
            for (int k = 0; k < kSuper; k++)
                if (box_may_open(boxes + 6 * ((int64_t)sb * kSuper + k), c, eps2, thr)) return true;
        }
    }
    return false;
}
// one pass over the own tree decides for EVERY destination rank: diff row j marks the pre-order ranges rank j
// does not need.
__global__ __launch_bounds__(kBlock) void k_let_mark(const Node *__restrict__ nodes, const Node64 *__restrict__ n64,
                                                     int64_t num_nodes, const double *__restrict__ boxes,
                                                     const double *__restrict__ supers, const double *__restrict__ megas,
                                                     const double *__restrict__ rankbox, int world, int me, double theta, double eps2,
                                                     int32_t *__restrict__ diff, int64_t stride, const TreeInfo *__restrict__ info,
                                                     int64_t link_base) {
    // (a wave-uniform form of these loops - a level entered if ANY lane needs it, box data by scalar loads -
    // was slower, 266 vs 190 us at 8 ranks: the lanes' early exits are worth more than the cheaper loads)
    // `num_nodes` is the host's launch bound (it may be an estimate from the previous step); the tree's own count is
    // on the device
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t N = info->num_nodes;
    if (i >= num_nodes || i >= N || i < info->own_skip) return;  // (the first own_skip nodes: k_let_mark_left speaks for them)
    const Node nd = nodes[i];
    if (__float_as_int(nd.s2t) == 0) return;  // a leaf: nothing below it
    const Node64 c = n64[i];
    const double size = c.hs * 2.0;
    const bool all = !(theta > 0.0);  // theta == 0: every cell is opened by everybody
    const double thr = all ? 0.0 : (size / theta) * (size / theta) * (1.0 + 1e-9);
    const int64_t nx = nd.next_off >= kLinkTag ? N : (int64_t)(nd.next_off / kNodeBytes) - link_base;  // (a subtree that leaves the rank: to the end of the array)
    if (nx <= i + 1) return;
    // [r4] A cell whose PARENT no body of rank j can open lies inside the range the parent drops for j: it has nothing to
    // add (13 of 14 marks of a deep cell were of that kind: two atomics each).  The parent is the cube of edge 2 size around
    // this cell; its centre of mass and ours both lie in it, at most its diagonal 2 sqrt(3) size apart, so every point b of
    // rank j's box is at least d(b, our centre of mass) - 2 sqrt(3) size from the parent's: the parent passes the opening
    // test for all of rank j when d^2 > reach^2.  (Skipping a mark can only keep a node that could have been dropped -
    // the exported piece stays correct whatever this test does; the parent's own test is the exact one.)
    const double far = sqrt(fmax(0.0, 4.0 * thr * (1.0 + 1e-6) - eps2)) + 3.4642 * size;
    const double reach2 = i > 0 ? far * far : INFINITY;
    for (int j = 0; j < world; j++) {
        if (j == me) continue;
        if (!all) {
            const double *rb = rankbox + 6 * j;
            if (rb[0] <= rb[3]) {
                const double dx = fmax(0.0, fmax(rb[0] - c.cx, c.cx - rb[3])), dy = fmax(0.0, fmax(rb[1] - c.cy, c.cy - rb[4])),
                             dz = fmax(0.0, fmax(rb[2] - c.cz, c.cz - rb[5]));
                if (dx * dx + dy * dy + dz * dz > reach2) continue;
            }
        }
        if (!all && !rank_may_open(j, c, eps2, thr, boxes, supers, megas, rankbox)) {
            int32_t *d = diff + (int64_t)j * stride;
            atomicAdd(&d[i + 1], 1);
            atomicAdd(&d[nx], -1);
        }
    }
}
// The cells that begin on a lower rank and reach into this one: their owner decides with the global moments whether
// a destination needs their subtree; this rank takes the same decision from the same numbers (chain_cell) for the
// part of the subtree that lies in ITS array: nodes [0, Ln[level]).  One wave per destination rank, lane = level.
__global__ void k_let_mark_left(const ChainTable *__restrict__ T, int world, int me, const double *__restrict__ boxes,
                                const double *__restrict__ supers, const double *__restrict__ megas,
                                const double *__restrict__ rankbox, double theta, double eps2, int32_t *__restrict__ diff,
                                int64_t stride, const TreeInfo *__restrict__ info) {
    const int lev = threadIdx.x;
    const ChainTable &t = T[me];
    if (lev >= kChainLevels || t.n == 0 || info->error) return;
    const int p = chain_prev(T, me);
    if (p < 0 || chain_cpl(T, p, me) < lev) return;
    if (!(theta > 0.0)) return;  // everybody opens everything
    const ChainCell cc = chain_cell(T, world, p, lev);
    Node64 c{0.0, 0.0, 0.0, ldexp(info->bounds, -lev)};
    if (cc.M > 0.0) { c.cx = cc.mx / cc.M; c.cy = cc.my / cc.M; c.cz = cc.mz / cc.M; }
    const double size = c.hs * 2.0;
    const double thr = (size / theta) * (size / theta) * (1.0 + 1e-9);
    const int64_t end = t.Lc[lev] < t.n ? t.Ln[lev] : info->num_nodes;
    const int j = blockIdx.x;
    if (j == me) return;
    if (!rank_may_open(j, c, eps2, thr, boxes, supers, megas, rankbox)) {
        int32_t *d = diff + (int64_t)j * stride;
        atomicAdd(&d[0], 1);
        atomicAdd(&d[end], -1);
    }
}
__global__ __launch_bounds__(kBlock) void k_let_keep(const int32_t *__restrict__ diff, const int32_t *__restrict__ diff_ex,
                                                     int64_t num_nodes, int64_t stride, int32_t *__restrict__ keep,
                                                     const TreeInfo *__restrict__ info) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t o = (int64_t)blockIdx.y * stride;
    if (i < num_nodes) keep[o + i] = (i >= info->own_skip && i < info->num_nodes && (diff_ex[o + i] + diff[o + i]) == 0) ? 1 : 0;
}
// kept nodes move to their new index in the destination's segment; links are re-based to the compacted
// numbering (a kept node's successor is always kept: it hangs off one of the node's own ancestors - on whichever
// rank that is: all ranks decide about a cell from the same numbers).  A row of the exchange buffer is 48 bytes,
// everything in float64 (LetRow): the receiver rounds the fp32 walk record out of it exactly as the sender's build
// did, and rebuilds the float64 record of the float64 force loop from the same numbers ([r3]; round 2 shipped the
// 24-byte fp32 record + a 32-byte float64 twin and owner mode had fp32 forces only).
struct LetRow {
    double cx, cy, cz, gm;  // centre of mass / body position, G * mass
    float s2t;              // as in Node (the fp32 loop's band included; 0: a leaf)
    unsigned next;          // skip link: node index in the compacted segment (link_rank = kLinkLocal), or the node of
                            // rank link_rank (in that rank's numbering behind ITS dropped copies) that follows the subtree
    unsigned orig;          // the node's index in the sender's array, behind the sender's dropped copies
    uint8_t level;          // cell level (half size = root half size / 2^level)
    uint8_t link_rank;
    uint16_t pad;
};
static_assert(sizeof(LetRow) == 48, "LetRow is a 48-byte wire row");
constexpr int kLetRow = 48;
__global__ __launch_bounds__(kBlock) void k_let_compact(const Node *__restrict__ nodes, const Node64 *__restrict__ n64,
                                                        const NodeD *__restrict__ nodesd /* may be null */,
                                                        const uint8_t *__restrict__ node_level, const OwnLink *__restrict__ links,
                                                        const ChainTable *__restrict__ T,
                                                        const int32_t *__restrict__ keep, const int32_t *__restrict__ newidx,
                                                        int64_t num_nodes, int64_t stride, int64_t capacity, int me,
                                                        const int64_t *__restrict__ seg_off, char *__restrict__ out,
                                                        const TreeInfo *__restrict__ info, int64_t link_base) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int j = blockIdx.y;
    if (j == me || i >= num_nodes || i >= info->num_nodes) return;
    const int64_t o = (int64_t)j * stride;
    if (!keep[o + i]) return;
    const int64_t k = seg_off[j] + newidx[o + i];
    if (k >= capacity) return;  // reported through the counts
    const Node nd = nodes[i];
    const bool leaf = __float_as_int(nd.s2t) == 0;
    LetRow row;
    row.s2t = nd.s2t;
    row.orig = (unsigned)(i - info->own_skip);
    row.level = node_level[i];
    row.pad = 0;
    if (nd.next_off >= kLinkTag) {
        const OwnLink L = links[nd.next_off & 0xffu];
        row.link_rank = (uint8_t)L.rank;
        row.next = L.rank == kLinkEnd ? 0u : (unsigned)(L.target - chain_skip(T, L.rank));
    } else {
        row.link_rank = (uint8_t)kLinkLocal;
        row.next = (unsigned)newidx[o + (int64_t)(nd.next_off / kNodeBytes) - link_base];
    }
    if (nodesd) {  // the float64 record has the unrounded numbers of leaves and cells alike
        const NodeD d = nodesd[i];
        row.cx = d.cx; row.cy = d.cy; row.cz = d.cz; row.gm = d.gm;
    } else if (!leaf) {
        const Node64 c = n64[i];
        row.cx = c.cx; row.cy = c.cy; row.cz = c.cz; row.gm = (double)nd.gm;
    } else {  // an fp32-only handle keeps no float64 copy of a leaf
        row.cx = (double)nd.cx; row.cy = (double)nd.cy; row.cz = (double)nd.cz; row.gm = (double)nd.gm;
    }
    *reinterpret_cast<LetRow *>(out + k * kLetRow) = row;
}
// rows per destination and where each destination's segment starts in the (packed) send buffer
__global__ void k_let_counts(const int32_t *__restrict__ newidx, int64_t num_nodes, int64_t stride, int world, int me,
                             int64_t *__restrict__ counts /* [world] counts, then [world] offsets */) {
    if (threadIdx.x != 0) return;
    int64_t off = 0;
    for (int j = 0; j < world; j++) {
        const int64_t c = j == me ? 0 : newidx[(int64_t)j * stride + num_nodes];
        counts[j] = c;
        counts[world + j] = off;
        off += c;
    }
}
// two more levels above the super boxes: kMegasPerRank "mega boxes" (kSupersPerMega consecutive super boxes each)
// and the union box of the rank.  One workgroup, one thread per mega box.
__global__ __launch_bounds__(kMaxWorld * kMegasPerRank) void k_rank_boxes(const double *__restrict__ supers, int world,
                                                                          double *__restrict__ megas, double *__restrict__ rankbox) {
    __shared__ double m[kMaxWorld * kMegasPerRank][6];
    const int t = threadIdx.x, j = t / kMegasPerRank;
    double v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (j < world) {
        for (int k = 0; k < kSupersPerMega; k++) {
            const double *b = supers + 6 * ((int64_t)t * kSupersPerMega + k);
            if (!(b[0] <= b[3])) continue;
            for (int c = 0; c < 3; c++) { v[c] = fmin(v[c], b[c]); v[3 + c] = fmax(v[3 + c], b[3 + c]); }
        }
        for (int c = 0; c < 6; c++) { megas[6 * t + c] = v[c]; m[t][c] = v[c]; }
    }
    __syncthreads();
    if (j < world && t % kMegasPerRank == 0) {
        for (int g = 1; g < kMegasPerRank; g++)
            for (int c = 0; c < 3; c++) { v[c] = fmin(v[c], m[t + g][c]); v[3 + c] = fmax(v[3 + c], m[t + g][3 + c]); }
        for (int c = 0; c < 6; c++) rankbox[6 * j + c] = v[c];
    }
}
// Where the pieces lie in a rank's walk array, in rank order:
//   row 0: a jump node (a massless leaf everybody accepts, its skip link leads to walk_first)
//   [walk_first, own_base + own_skip): the received pieces of the lower ranks, packed up against
//   [own_base + own_skip, own_base + num_nodes): the own piece, where k_emit_tile / k_chain_fix built it
//   behind it the received pieces of the higher ranks, then the sentinel.
struct Pieces {
    const char *src[kMaxWorld];  // received rows of each rank (null: none / the own rank)
    long long count[kMaxWorld];
    long long base[kMaxWorld];   // first row of the rank's piece in the walk array (own rank: own_base + own_skip)
    long long total;             // row of the sentinel
    int me;
};
// a link (rank, node behind that rank's dropped copies) -> row of the walk array.  The rows of a received piece are
// in pre-order, so the first row at or behind the node is looked up.  A link whose target was pruned belongs to a
// node nobody here can reach (a cell above both of them is never opened - deep cells below theta * softening are
// opened by nobody, not even by the bodies inside them); it then leads to the next node that IS here, which is
// where a walk leaves that never-entered subtree anyway.  The pieces follow each other without gaps in rank order,
// so "behind the last row of a piece" is the first row of the next.
__device__ __forceinline__ long long piece_row(const Pieces &P, int rank, long long node) {
    if (rank == kLinkEnd) return P.total;
    if (rank == P.me) return P.base[rank] + node;
    long long lo = 0, hi = P.count[rank];
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        const unsigned o = reinterpret_cast<const LetRow *>(P.src[rank] + mid * kLetRow)->orig;
        if (o < (unsigned)node) lo = mid + 1; else hi = mid;
    }
    return P.base[rank] + lo;
}
__global__ __launch_bounds__(kBlock) void k_let_append(Pieces P, int from, Node *__restrict__ nodes, Node64 *__restrict__ n64,
                                                       NodeD *__restrict__ nodesd /* may be null */, TreeInfo *info) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= P.count[from]) return;
    const LetRow row = *reinterpret_cast<const LetRow *>(P.src[from] + k * kLetRow);
    const long long base = P.base[from];
    const long long nx = row.link_rank == kLinkLocal ? base + row.next : piece_row(P, row.link_rank, row.next);
    const bool leaf = __float_as_int(row.s2t) == 0;
    Node nd;
    nd.cx = (float)row.cx; nd.cy = (float)row.cy; nd.cz = (float)row.cz; nd.gm = (float)row.gm;
    nd.s2t = row.s2t;
    nd.next_off = (unsigned)nx * kNodeBytes;
    nodes[base + k] = nd;
    n64[base + k] = Node64{row.cx, row.cy, row.cz, leaf ? 0.0 : ldexp(info->bounds, -(int)row.level)};
    if (nodesd) {
        // the float64 loop's narrow band instead of the fp32 loop's (every rank derives the same band from the
        // global extent, k_emit_tile)
        const int bits = __float_as_int(row.s2t);
        const float s2d = bits == 0 ? 0.0f : __int_as_float(bits - (int)(info->band2 >> 1) + (int)kBand64);
        nodesd[base + k] = NodeD{row.cx, row.cy, row.cz, row.gm, s2d, (unsigned)nx * kNodeDBytes};
    }
}
// the own piece's links that leave the rank, the jump node, the sentinel (one wave)
__global__ void k_let_finish(Pieces P, const OwnLink *__restrict__ links, const ChainTable *__restrict__ T, long long own_base,
                             long long walk_first, Node *__restrict__ nodes, NodeD *__restrict__ nodesd, TreeInfo *info) {
    const int t = threadIdx.x;
    if (t < kOwnLinks && links[t].node >= 0) {
        const int lr = links[t].rank;
        const long long nx = piece_row(P, lr, lr == kLinkEnd ? 0 : links[t].target - chain_skip(T, lr));
        const long long i = own_base + links[t].node;
        nodes[i].next_off = (unsigned)nx * kNodeBytes;
        if (nodesd) nodesd[i].next_off = (unsigned)nx * kNodeDBytes;
    }
    if (t == 0) {
        Node sn;
        sn.cx = sn.cy = sn.cz = 1.0e30f;
        sn.gm = 0.f; sn.s2t = 0.f;
        sn.next_off = (unsigned)P.total * kNodeBytes;
        nodes[P.total] = sn;
        if (nodesd) nodesd[P.total] = NodeD{1.0e30, 1.0e30, 1.0e30, 0.0, 0.0f, (unsigned)P.total * kNodeDBytes};
        sn.next_off = (unsigned)walk_first * kNodeBytes;  // the jump node: accepted by every lane, adds nothing
        nodes[0] = sn;
        if (nodesd) nodesd[0] = NodeD{1.0e30, 1.0e30, 1.0e30, 0.0, 0.0f, (unsigned)walk_first * kNodeDBytes};
        info->walk_nodes = P.total;
        info->walk_first = walk_first;
    }
}
// world 1: nothing was received, the walk array is the own tree as built (its sentinel is in place)
__global__ void k_let_finish_alone(TreeInfo *info) {
    info->walk_nodes = info->num_nodes;
    info->walk_first = 0;
}
__global__ void k_copy_ids(const int32_t *__restrict__ ids, int32_t *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ids[i];
}

}  // namespace

// =========================================================================================
// handle
// =========================================================================================
// Measured (MI355X, theta 0.5, walk ms, middle cut -> cut at the wave's own leaves): galaxy 1 M 1.251 -> 1.251, 2 M
// 2.515 -> 2.409, 4 M 5.13 -> 4.95, 10 M 12.85 -> 12.28; collision 1 M 1.402 -> 1.389, 2 M 2.477 -> 2.386, 4 M
// 4.68 -> 4.40, 10 M 11.50 -> 10.79.  The home cut keeps both cursors busy for 68-81 % of the visits instead of
// ~20 % (scripts/analysis/range_balance.py), but what two cursors overlap is the latency of far-field jumps, and
// that only shows once the node array is far beyond the L2.  (A cut placed by a two-component model of where the
// visits are - 75 % instead of 68 % at 1 M - measured the same or slower.)
constexpr int64_t kHomeSplitBodies = 1500000;

struct nbmi_sim {
    int64_t n = 0;
    int method = 0, device = 0;
    double G = 0, softening = 0, damping = 1, theta = 0.5;
    hipStream_t stream = nullptr;
    int curbuf = 0;
    int64_t steps_taken = 0;  // nbmi_step_count()
    Bodies buf[2] = {};
    // scratch
    uint64_t *key_hi = nullptr, *key_lo = nullptr, *hi_s = nullptr, *lo_s = nullptr;
    uint32_t *idx = nullptr, *perm = nullptr;
    int32_t *delta = nullptr, *Pex = nullptr;  // Pex: exclusive prefix of the cell counts INSIDE a sub-tile (PexL)
    int32_t *subPex = nullptr, *sub_cnt = nullptr;  // per sub-tile: exclusive prefix / total of the cell counts
    float4 *posm_s = nullptr;
    double4 *p64_s = nullptr;  // float64 twin of posm_s: owner mode (bounding boxes) and NBMI_PREC only
    double4 *S = nullptr;      // in-sub-tile exclusive prefix of {G m, G m x, G m y, G m z}
    double4 *sub_tot = nullptr;  // per sub-tile totals of the same
    Moment *T = nullptr;         // per sub-tile exclusive prefix, double-double
    Node *nodes = nullptr;
    uint32_t *child_tab = nullptr;  // 8 child offsets per node row (stack walk)
    Node64 *nodes64 = nullptr;  // float64 twin rows of the internal cells (near-tie re-decision)
    NodeD *nodesd = nullptr;    // float64 node records of every node (waves that compute forces in float64)
    int force_prec = 0;         // 0 = per wave by local density, 1 = fp32 everywhere, 2 = float64 everywhere (NBMI_FORCE_PREC)
    int all64_enter_pm = kAll64Enter, all64_leave_pm = kAll64Leave;  // "auto": every wave float64 while most of the system asks (per mille of the waves)
    double prec_tau = 5.0e-5;   // force_prec 0: float64 where G rho dt^2 exceeds this (NBMI_PREC_TAU)
    WalkTable *wtab = nullptr;  // device copy of the walk's per-handle constants
    uint8_t *node_level = nullptr;
    int32_t *node_ref = nullptr;  // first body (sorted rank) of every node; queries only
    int64_t node_capacity = 0;   // rows of the walk array (own tree + received trees)
    int64_t own_node_rows = 0;   // rows the handle's own tree may use
    TreeInfo *info = nullptr;  // device
    void *tmp_sort = nullptr;
    size_t tmp_sort_bytes = 0;
    float *colors = nullptr;  // (N,3) original order
    void *stage = nullptr;    // getter staging, 3N doubles
    bool tree_valid = false;
    int64_t shard_begin = 0, shard_end = 0;
    bool exchange_sync = true;  // export / import block the host (false: the caller orders streams itself)
    // octree inputs in key order: the handle's own sorted arrays (nt == n)
    int64_t nt = 0;
    uint64_t *t_hi = nullptr, *t_lo = nullptr;
    float4 *t_posm = nullptr;
    // owner mode (multi-GPU stage 2): this handle holds the bodies of one octant-key range
    bool owner = false;
    int world = 0, rank = 0;
    int64_t cap = 0;            // body capacity (0: exactly n)
    int64_t let_capacity = 0;   // rows of one locally essential tree in the exchange buffers
    int64_t node_extra = 0;     // node rows reserved behind the own tree for received trees
    uint64_t *let_split = nullptr;
    uint8_t *let_dead = nullptr;   // rows whose body has just been handed to another rank
    int64_t n_leaving = 0;
    uint32_t *let_dest = nullptr;
    int64_t *let_counts = nullptr;
    int32_t *let_diff = nullptr, *let_scan = nullptr, *let_keep = nullptr, *let_tiles = nullptr, *let_ranges = nullptr;
    double *let_supers = nullptr, *let_megas = nullptr, *let_rankbox = nullptr;
    int64_t let_stride = 0, let_tile_stride = 0;  // rows between two destinations' work arrays
    TreeInfo *h_info = nullptr;   // pinned host copy of the tree header, refreshed by every nbmi_owner_adopt
    hipEvent_t ev_info = nullptr;  // ... complete when this event is
    int64_t last_nodes = -1;      // num_nodes of the previous step's own tree (launch bound of this step's tree export)
    int64_t own_base = 0;         // [r3] row of the walk array at which the own tree is built (world > 1: behind the room for the lower ranks' pieces)
    int32_t *chain_r = nullptr;   // nodes of the cells that hold the rank's last body, by level
    OwnLink *own_links = nullptr; // the own piece's links that leave the rank
    const ChainTable *chains = nullptr;  // every rank's table (the caller's gathered buffer of this step)
    // frame codec: previous decoded frame (positions then colours, float32, caller's order) and the int16 payload
    float *frame_prev = nullptr;
    int16_t *frame_q = nullptr;
    bool frame_have_prev = false;
    // render-side reduction scratch (nbmi_visible_points), allocated on first use
    uint8_t *vis_flag = nullptr;
    uint32_t *vis_slot = nullptr, *vis_tiles = nullptr;
    int xcd_chunk = 0;  // walk block -> XCD mapping, see logical_block()
    int xcd_balance = 1;  // XCD ranges cut by last step's measured wave times (NBMI_XCD_BALANCE=0: equal eighths)
    int *xcd_bounds = nullptr;        // device [9]
    unsigned *wave_cycles = nullptr;  // device [4 per walk block]
    unsigned char *wave_flag = nullptr;  // device [one per wave]
    int32_t *sub_flag = nullptr;         // device [one per tile]: waves of the tile that ask for float64
    double step_dt = 0.0;                // dt of the step being enqueued (0: a build without a step)
    double uniform_gm = -1.0;            // direct N^2: G m when every body has the same positive mass (the reference's presets: masses = 1), else < 0
    int owner_all64 = -1;                // owner mode: the system-wide "every wave float64" verdict for the next walk (-1: this rank's own rule)
    double owner_dt = 0.0;               // owner mode: the dt the next nbmi_owner_step will use (nbmi_owner_set_dt; "auto" needs it at build time)
    int balance_blocks = 0;           // the block count the bounds on the device were made for (0: none yet)
    // the cuts for the NEXT walk are made on a stream of their own, beside the next step's build (one workgroup, 70 us
    // at 10 M bodies: off the critical path)
    hipStream_t side = nullptr;
    hipEvent_t ev_walked = nullptr, ev_cut = nullptr;
    bool cut_pending = false;         // a k_xcd_bounds on `side` that the next balanced walk has to wait for
    int walk_block = kBlock;  // threads per walk block (64, 128 or 256; measurement knob NBMI_WALK_BLOCK)
    int sort_bits = 0;   // upper-word bits the radix sort looks at (0: chosen from n; NBMI_SORT_BITS); widened when long runs show up
    bool maxabs_fused = false;  // TreeInfo::maxabs_next holds max |coordinate| of the CURRENT positions (set by a full
                                // integrating walk, dropped by anything else that writes positions); NBMI_FUSE_MAXABS=0: never
    bool fuse_maxabs = true;
    bool hilbert = true;  // sort keys relabelled along the Hilbert curve (k_keys); NBMI_HILBERT=0: plain octant digits
    int walk_stack = 0;  // prototype: stack walk with batched children (NBMI_WALK_STACK=1)
    int walk_lane = 0;  // measurement: per-lane walk (NBMI_WALK_LANE=1)
    int prec = 0;       // measurement: k_walk_diag arithmetic mode (NBMI_PREC), 0 = product walk
    double prec_near = 4.0;  // NBMI_PREC_NEAR: "near" = closer than this many softening lengths
    double4 *diag64 = nullptr;  // float64 {cx, cy, cz, G m} of every node (only with NBMI_PREC)
    // one-wave walk: cursors per wave and where the array is cut.  -1 = by size: two cursors, cut at the middle of the
    // array, or (from kHomeSplitBodies = 1.5 M bodies on) at the leaf of the wave's middle body; NBMI_WALK_PAIR = 0 / 1 / 2 forces
    // one cursor / the middle cut / the home cut
    int walk_pair = -1;
    int64_t split_max_waves = 9400;  // split walk: K waves per group while groups x K fits; NBMI_SPLIT_WAVES (0 = off)
    // timers
    bool timers = false;
    hipEvent_t ev[6] = {};
    double ms[5] = {0, 0, 0, 0, 0};
    int64_t timed_steps = 0;
    std::vector<void *> allocs;
};

namespace {

template <typename T>
int dev_alloc(nbmi_sim *s, T **p, size_t count) {
    void *q = nullptr;
    NBMI_HIP_CHECK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    s->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

int alloc_bodies(nbmi_sim *s, Bodies *b, int64_t n) {
    if (dev_alloc(s, &b->x, n) || dev_alloc(s, &b->y, n) || dev_alloc(s, &b->z, n) || dev_alloc(s, &b->vx, n) ||
        dev_alloc(s, &b->vy, n) || dev_alloc(s, &b->vz, n) || dev_alloc(s, &b->m, n) || dev_alloc(s, &b->id, n))
        return -2;
    return 0;
}

// (re)writes the device copy of the walk's per-handle constants
int upload_walk_table(nbmi_sim *s) {
    WalkTable t;
    t.buf[0] = s->buf[0];
    t.buf[1] = s->buf[1];
    t.n64 = s->nodes64;
    t.nodesd = s->nodesd;
    t.xcd_bounds = s->xcd_bounds;
    t.wave_cycles = s->wave_cycles;
    t.wave_flag = s->wave_flag;
    t.pex = s->Pex;
    t.subpex = s->subPex;
    t.maxabs_next = &s->info->maxabs_next;
    t.theta = s->theta;
    t.eps2 = s->softening * s->softening;
    t.own_base = s->own_base;
    NBMI_HIP_CHECK(hipMemcpyAsync(s->wtab, &t, sizeof(t), hipMemcpyHostToDevice, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));  // `t` is a stack object
    return 0;
}

int check_handle(nbmi_sim *s) {
    if (!s) {
        nbmi::set_error("null nbmi_sim handle");
        return NBMI_ERR_ARG;
    }
    if (hipSetDevice(s->device) != hipSuccess) {
        nbmi::set_error("hipSetDevice(%d) failed", s->device);
        return NBMI_ERR_HIP;
    }
    return 0;
}

// The octree build in three enqueue stages (one after the other for a single-GPU step; the
// multi-GPU run exchange puts its two collectives between them):
//   enqueue_maxabs      reset the tree header, max |coordinate| of the handle's own bodies
//   enqueue_local_sort  keys -> sort -> tie fix -> gather: the handle's bodies in key order
//   enqueue_global_tree delta -> scans -> node emission over the nt bodies of t_hi/t_lo/t_posm
int enqueue_maxabs(nbmi_sim *s) {
    const int64_t n = s->n;
    hipStream_t st = s->stream;
    Bodies cur = s->buf[s->curbuf];
    if (s->maxabs_fused) {  // the last walk left it behind (and nothing has touched the positions since)
        s->maxabs_fused = false;
        k_header<<<1, 64, 0, st>>>(s->info);
        NBMI_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // reset maxabs/num_nodes/max_level/error (keep counters)
    NBMI_HIP_CHECK(hipMemsetAsync(s->info, 0, offsetof(TreeInfo, wave_visits), st));
    if (n == 0) return 0;  // an owner-mode rank may hold no bodies: the cleared header is all there is (a 0-block launch is an error)
    int gb = nblocks(n);
    if (gb > 256) gb = 256;  // one same-address atomic per block: keep them few
    k_maxabs<<<gb, kBlock, 0, st>>>(cur.x, cur.y, cur.z, n, s->info);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

// n_sort rows are keyed and sorted (owner mode: rows of emigrants are still in place, flagged dead, and sort to
// the end); the first n_live of the order are gathered for the tree
// force precision "auto" is decided during the build of a step (it needs dt); anything else leaves the flags alone
static inline bool auto_prec(const nbmi_sim *s) { return s->nodesd && s->force_prec == 0 && s->step_dt > 0.0; }

int enqueue_local_sort(nbmi_sim *s, int ev_base, int64_t n_sort = -1, int64_t n_live = -1, const uint8_t *dead = nullptr) {
    const int64_t n = n_sort < 0 ? s->n : n_sort;
    if (n_live < 0) n_live = n;
    hipStream_t st = s->stream;
    Bodies cur = s->buf[s->curbuf];
    if (s->hilbert) k_keys<true><<<nblocks(n), kBlock, 0, st>>>(cur.x, cur.y, cur.z, n, s->info, s->key_hi, s->key_lo, s->idx, dead);
    else k_keys<false><<<nblocks(n), kBlock, 0, st>>>(cur.x, cur.y, cur.z, n, s->info, s->key_hi, s->key_lo, s->idx, dead);
    if (ev_base >= 0) NBMI_HIP_CHECK(hipEventRecord(s->ev[1], st));
    // radix sort on the top sort_bits bits of the upper word, then the tie-fix completes the 126-bit order
    if (s->sort_bits == 0) {
        // enough levels that cells of that level hold about one body on average, plus four: 3 (log8 n + 4) bits,
        // rounded up to whole 8-bit digit passes (1 M and 10 M bodies: 40 bits = 5 passes; measured at 1 M: with
        // 32 bits the tie-fix's runs in the galaxy core cost more (sort phase 0.25 ms) than the pass saved (0.15))
        int levels = 4;
        for (int64_t c = 1; c < n; c *= 8) levels++;
        int bits = ((3 * levels + 7) / 8) * 8;
        if (bits < 16) bits = 16;
        s->sort_bits = bits < 63 ? bits : 63;
    }
    const int shift = 63 - s->sort_bits;
    NBMI_HIP_CHECK(nbmi::sort_pairs_u64_u32(s->tmp_sort, s->tmp_sort_bytes, s->key_hi, s->hi_s, s->idx, s->perm,
                                            (size_t)n, shift, 63, st));
    k_tiefix<<<nblocks(n), kBlock, 0, st>>>(s->hi_s, s->key_lo, s->perm, shift, s->idx, s->key_hi, n, s->info,
                                            dead ? s->let_scan : nullptr, n - n_live);
    // the finished permutation / sorted upper words are `perm` / `hi_s` from here on; the old buffers take
    // the next step's indices and keys
    std::swap(s->perm, s->idx);
    std::swap(s->hi_s, s->key_hi);
    s->t_hi = s->hi_s;
    if (ev_base >= 0) NBMI_HIP_CHECK(hipEventRecord(s->ev[2], st));
    // ranks 0 .. n_live: entry n_live is the slot of the totals
    k_gather_scan<<<(int)((n_live + 1 + kScanTile - 1) / kScanTile), kBlock, 0, st>>>(
        cur, s->perm, s->hi_s, s->key_lo, n_live, s->G, s->posm_s, s->p64_s, s->lo_s, s->delta, s->S, s->Pex, s->sub_tot, s->sub_cnt,
        auto_prec(s) ? s->wave_flag : nullptr, s->sub_flag, auto_prec(s) ? (float)(s->prec_tau / (s->step_dt * s->step_dt)) : 0.f,
        (float)s->softening);
    return 0;
}

// aux: also write node_ref / node_level (cell queries, owner-mode kernels)
int enqueue_global_tree(nbmi_sim *s, bool aux = true) {
    const int64_t n = s->nt;
    hipStream_t st = s->stream;
    // (delta, the in-sub-tile prefixes S / PexL and the sub-tile totals: written by k_gather_scan)
    const int64_t nsub = (n + 1 + kScanTile - 1) / kScanTile;  // sub-tiles that hold the entries 0 .. n
    k_scan_subtiles<<<1, kSubScanThreads, 0, st>>>(s->sub_tot, s->sub_cnt, nsub, s->T, s->subPex,
                                                   auto_prec(s) ? s->sub_flag : nullptr, (n + 63) / 64, s->all64_enter_pm, s->all64_leave_pm, s->info);
    // theta = 0 means "never accept an internal node": s2t = +inf
    const double inv_theta2 = s->theta > 0.0 ? 1.0 / (s->theta * s->theta) : INFINITY;
    const int64_t ob = s->own_base;  // (owner mode: the own tree begins at this row of the walk array; node_level / node_ref / diag64 count from the tree's start)
    {
        const int tile = n <= kEmitSmallBodies ? kEmitTileSmall : kEmitTile;
#define NBMI_EMIT(TV) k_emit_tile<TV><<<(int)((n + TV - 1) / TV), kBlock, 0, st>>>(                                         \
        s->delta, s->Pex, s->subPex, s->S, s->T, s->t_posm, s->p64_s, s->t_hi, s->t_lo, n, s->own_node_rows, s->softening,       \
        inv_theta2, s->nodes + ob, s->nodes64 + ob, aux ? s->node_level : nullptr, aux ? s->node_ref : nullptr, s->diag64,   \
        s->force_prec != 1 && s->nodesd ? s->nodesd + ob : nullptr, s->buf[s->curbuf], s->perm, s->G, s->info, ob)
        if (tile == kEmitTileSmall) NBMI_EMIT(kEmitTileSmall);
        else NBMI_EMIT(kEmitTile);
#undef NBMI_EMIT
    }
    if (s->walk_stack)
        k_child_table<<<nblocks(s->own_node_rows), kBlock, 0, st>>>(s->nodes, s->info, s->own_node_rows, s->child_tab);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

// Single-GPU build: the tree over the handle's own bodies.
int enqueue_tree(nbmi_sim *s, int ev_base, bool aux = true) {
    if (s->owner) {
        nbmi::set_error("this handle is in owner mode: use the nbmi_owner_* calls");
        return NBMI_ERR_ARG;
    }
    if (ev_base >= 0) NBMI_HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
    if (int rc = enqueue_maxabs(s)) return rc;
    if (int rc = enqueue_local_sort(s, ev_base)) return rc;
    if (int rc = enqueue_global_tree(s, aux)) return rc;
    if (ev_base >= 0) NBMI_HIP_CHECK(hipEventRecord(s->ev[3], s->stream));
    s->tree_valid = aux;
    return 0;
}

int enqueue_walk(nbmi_sim *s, bool integrate, double dt, double *acc_out) {
    const int64_t n = s->n;
    hipStream_t st = s->stream;
    if (!s->wtab || !s->nodes64) {  // the kernels dereference both: never launch without them
        nbmi::set_error("internal: walk table not initialised");
        return NBMI_ERR_ARG;
    }
    WalkParams P;
    P.rank_begin = integrate ? s->shard_begin : 0;
    P.rank_end = integrate ? s->shard_end : n;
    P.eps2 = (float)(s->softening * s->softening);
    const bool guard = !(P.eps2 > 0.f);
    P.dt = dt;
    P.damping = s->damping;
    const int64_t cntr = P.rank_end - P.rank_begin;
    if (cntr <= 0) return 0;
    P.xcd_chunk = s->xcd_chunk;
    P.pair = s->walk_pair >= 0 ? s->walk_pair : (s->nt >= kHomeSplitBodies ? 2 : 1);
    if (s->owner && s->world > 1 && P.pair == 1) P.pair = 2;  // (the middle of the array may lie in the unused rows in front of the pieces)
    P.curbuf = s->curbuf;
    P.acc64 = getenv("NBMI_ACC64") ? atoi(getenv("NBMI_ACC64")) : 0;
    P.balance = 0;
    P.force_prec = s->nodesd ? s->force_prec : 1;
    P.prec_tau = (float)s->prec_tau;
    P.prec = s->prec;
    P.near2 = s->prec >= 20 ? (float)(s->prec_near * s->prec_near) : (float)(s->prec_near * s->prec_near * s->softening * s->softening);
    if (integrate && s->prec && s->diag64 && !guard && !s->owner) {  // measurement only, see k_walk_diag
        k_walk_diag<<<(int)((cntr + kBlock - 1) / kBlock), kBlock, 0, st>>>(s->nodes, s->diag64, s->wtab, s->info, s->posm_s, s->perm, P);
        NBMI_HIP_CHECK(hipGetLastError());
        return 0;
    }

    // few groups: a block of K waves per group, each walking one K-th of the array.  K depends only on
    // the size of the tree (not on the shard), so that every sharding adds up the same partial sums.
    const int64_t tree_groups = (s->nt + 63) / 64, groups = (cntr + 63) / 64;
    // measured (galaxy, 10 k ... 300 k bodies): the largest K <= 16 with groups x K <= ~9 400 waves is the
    // fastest or within 5 % of it; beyond ~280 k bodies the one-wave walk wins
    int parts = 1;
    while (parts < 16 && tree_groups <= 4300 && tree_groups * parts * 2 <= s->split_max_waves) parts *= 2;
    if (integrate && !guard && parts > 1) {
#define NBMI_SPLIT(KV) \
    k_walk_split<KV><<<(int)groups, 64 * KV, 0, st>>>(s->nodes, s->wtab, s->info, s->posm_s, s->perm, P)
        if (parts == 2) NBMI_SPLIT(2);
        else if (parts == 4) NBMI_SPLIT(4);
        else if (parts == 8) NBMI_SPLIT(8);
        else NBMI_SPLIT(16);
#undef NBMI_SPLIT
        NBMI_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int wb = s->walk_block;
    const int gb = (int)((cntr + wb - 1) / wb);
    if (integrate && s->walk_stack && !guard && !s->owner) {
        k_walk_stack<<<(int)((cntr + kBlock - 1) / kBlock), kBlock, 0, st>>>(s->nodes, s->child_tab, s->wtab, s->info, s->posm_s,
                                                                             s->perm, P);
        NBMI_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (integrate && s->walk_lane) {  // measurement only, see k_walk_lane
        k_walk_lane<<<gb, wb, 0, st>>>(s->nodes, s->wtab, s->info, s->posm_s, s->perm, P);
        NBMI_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define NBMI_WALK(I, C, G) \
    k_walk<I, C, G><<<gb, wb, 0, st>>>(s->nodes, s->wtab, s->info, s->posm_s, s->perm, acc_out, P, s->info)
    // balance mode: full, unsharded integrating walks of the product kernel with the default block mapping
    // (measured: 10 M collision walk 15.9 -> 14.8 ms, fp32 10.4 -> 9.8; 4 M galaxy 6.96 -> 6.90; at 1 M bodies the eighths are
    // within 1 % of each other already and the half-empty launch costs 2 %: from 8 192 blocks = 2 M bodies on)
    const bool balance = integrate && !guard && s->xcd_balance && s->xcd_chunk == 0 && wb == kBlock && !s->owner &&
                         (gb >= 8192 || s->xcd_balance > 1) &&
                         P.rank_begin == 0 && P.rank_end == n && getenv("NBMI_XCD_CHUNK") == nullptr;
    if (balance) {
        const int jmax = ((gb + 7) / 8) * 3 / 2 + 1;
        if (s->cut_pending) {  // the cuts made from the last walk's times (on the side stream)
            NBMI_HIP_CHECK(hipStreamWaitEvent(st, s->ev_cut, 0));
            s->cut_pending = false;
        }
        if (s->balance_blocks != gb) {  // first use (or another shard size): no times yet -> equal eighths
            NBMI_HIP_CHECK(hipMemsetAsync(s->wave_cycles, 0, (size_t)gb * 16, st));
            k_xcd_bounds<<<1, 1024, 0, st>>>(s->wave_cycles, gb, jmax, s->xcd_bounds);
            s->balance_blocks = gb;
        }
        P.balance = 1;
        if (!s->side) {
            NBMI_HIP_CHECK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
            NBMI_HIP_CHECK(hipEventCreateWithFlags(&s->ev_walked, hipEventDisableTiming));
            NBMI_HIP_CHECK(hipEventCreateWithFlags(&s->ev_cut, hipEventDisableTiming));
        }
        k_walk<true, false, false><<<8 * jmax, wb, 0, st>>>(s->nodes, s->wtab, s->info, s->posm_s, s->perm, acc_out, P, s->info);
        NBMI_HIP_CHECK(hipGetLastError());
        // cuts for the next step: beside whatever the main stream does next (the next step's keys, sort and build)
        NBMI_HIP_CHECK(hipEventRecord(s->ev_walked, st));
        NBMI_HIP_CHECK(hipStreamWaitEvent(s->side, s->ev_walked, 0));
        k_xcd_bounds<<<1, 1024, 0, s->side>>>(s->wave_cycles, gb, jmax, s->xcd_bounds);
        NBMI_HIP_CHECK(hipGetLastError());
        NBMI_HIP_CHECK(hipEventRecord(s->ev_cut, s->side));
        s->cut_pending = true;
        return 0;
    }
    if (integrate) {
        if (guard) NBMI_WALK(true, false, true); else NBMI_WALK(true, false, false);
    } else {
        if (guard) NBMI_WALK(false, true, true); else NBMI_WALK(false, true, false);
    }
#undef NBMI_WALK
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

template <bool kIntegrate>
int launch_direct(nbmi_sim *s, double dt, double *acc_out) {
    const int64_t n = s->n;
    hipStream_t st = s->stream;
    Bodies cur = s->buf[s->curbuf], nxt = s->buf[1 - s->curbuf];
    k_pack_posm<<<nblocks(n), kBlock, 0, st>>>(cur, n, s->G, s->posm_s);
    const float eps2 = (float)(s->softening * s->softening);
    const bool guard = !(eps2 > 0.f);
    // multi-GPU: a sharded handle integrates only the bodies [shard_begin, shard_end) (index order:
    // the direct method never re-orders the state); the force pass always covers everything
    const int64_t ibeg = kIntegrate ? s->shard_begin : 0, iend = kIntegrate ? s->shard_end : n;
    const int64_t cnt = iend - ibeg;
    if (cnt <= 0) return 0;
    // bodies per thread: enough blocks to cover 256 CUs a few times over
    int ib = cnt >= 512 * 1024 ? 4 : (cnt >= 128 * 1024 ? 2 : 1);
#define NBMI_DIRECT(IBV)                                                                                      \
    do {                                                                                                      \
        const int gb = (int)((cnt + (int64_t)kBlock * IBV - 1) / ((int64_t)kBlock * IBV));                    \
        if (guard)                                                                                            \
            k_direct<IBV, true, kIntegrate, false><<<gb, kBlock, 0, st>>>(s->posm_s, n, ibeg, iend, eps2, cur, nxt, \
                                                                         acc_out, dt, s->damping, 0.0);        \
        else if (s->uniform_gm > 0.0)                                                                         \
            k_direct<IBV, false, kIntegrate, true><<<gb, kBlock, 0, st>>>(s->posm_s, n, ibeg, iend, eps2, cur, nxt, \
                                                                         acc_out, dt, s->damping, s->uniform_gm); \
        else                                                                                                  \
            k_direct<IBV, false, kIntegrate, false><<<gb, kBlock, 0, st>>>(s->posm_s, n, ibeg, iend, eps2, cur, nxt, \
                                                                          acc_out, dt, s->damping, 0.0);      \
    } while (0)
    if (ib == 4) NBMI_DIRECT(4);
    else if (ib == 2) NBMI_DIRECT(2);
    else NBMI_DIRECT(1);
#undef NBMI_DIRECT
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

int check_device_error(nbmi_sim *s) {
    TreeInfo h;
    unsigned sort_err = 0u;
    NBMI_HIP_CHECK(hipMemcpyAsync(&h, s->info, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    if (s->tmp_sort) NBMI_HIP_CHECK(nbmi::sort_error_word(s->tmp_sort, &sort_err, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (sort_err) {  // a look-back spin of the radix sort timed out: that pass scattered to wrong offsets
        NBMI_HIP_CHECK(nbmi::sort_init_temp(s->tmp_sort, s->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
        s->tree_valid = false;
        nbmi::set_error("device radix sort: a look-back spin timed out; the steps since the last synchronisation are invalid");
        return NBMI_ERR_HIP;
    }
    if (h.max_run > 4096 && s->sort_bits < 63) {
        // many bodies agree on the sorted prefix (a dense core inside one level-13 cell): the tie-fix did the
        // rest correctly but at L reads per member - sort on more bits from the next step on
        s->sort_bits = s->sort_bits + 8 < 63 ? s->sort_bits + 8 : 63;
    }
    if (h.error || h.sticky_error) {
        // reported once: clear the sticky word so the handle can go on after nbmi_set_state / a retry.  The
        // bodies stand at the last step that completed (the walk froze them while the word was set).
        NBMI_HIP_CHECK(hipMemsetAsync(&s->info->sticky_error, 0, sizeof(int), s->stream));
        NBMI_HIP_CHECK(hipMemsetAsync(&s->info->error, 0, sizeof(int), s->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
        s->tree_valid = false;
        s->maxabs_fused = false;
    }
    if (h.error || h.sticky_error) {
        nbmi::set_error("octree needs %lld nodes, more than the %lld rows allocated (4N, as the reference); the "
                        "bodies were not advanced from that step on",
                        (long long)(h.sticky_error ? h.sticky_nodes : h.num_nodes), (long long)s->node_capacity);
        return NBMI_ERR_CAPACITY;
    }
    return 0;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int nbmi_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

const char *nbmi_last_error(void) { return nbmi::get_error(); }

void nbmi_destroy(nbmi_sim *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->side) {
        (void)hipStreamSynchronize(s->side);
        (void)hipStreamDestroy(s->side);
        (void)hipEventDestroy(s->ev_walked);
        (void)hipEventDestroy(s->ev_cut);
    }
    for (void *p : s->allocs) (void)hipFree(p);
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->ev_info) (void)hipEventDestroy(s->ev_info);
    if (s->h_info) (void)hipHostFree(s->h_info);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// measurement / tuning knobs, read once per handle by both constructors
static void read_env_knobs(nbmi_sim *s) {
    if (const char *e = getenv("NBMI_XCD_CHUNK")) s->xcd_chunk = atoi(e);
    if (const char *e = getenv("NBMI_XCD_BALANCE")) s->xcd_balance = atoi(e);
    if (const char *e = getenv("NBMI_SPLIT_WAVES")) s->split_max_waves = atoll(e);
    if (const char *e = getenv("NBMI_WALK_PAIR")) s->walk_pair = atoi(e);
    if (const char *e = getenv("NBMI_FUSE_MAXABS")) s->fuse_maxabs = atoi(e) != 0;
    if (const char *e = getenv("NBMI_HILBERT")) s->hilbert = atoi(e) != 0;
    if (const char *e = getenv("NBMI_WALK_LANE")) s->walk_lane = atoi(e);
    if (const char *e = getenv("NBMI_WALK_STACK")) s->walk_stack = atoi(e);
    if (const char *e = getenv("NBMI_FORCE_PREC")) {
        const int v = atoi(e);
        if (v >= 0 && v <= 2) s->force_prec = v;
    }
    if (const char *e = getenv("NBMI_PREC_TAU")) s->prec_tau = atof(e);
    if (const char *e = getenv("NBMI_ALL64_ENTER")) s->all64_enter_pm = (int)(1000.0 * atof(e) + 0.5);
    if (const char *e = getenv("NBMI_ALL64_LEAVE")) s->all64_leave_pm = (int)(1000.0 * atof(e) + 0.5);
    if (const char *e = getenv("NBMI_PREC")) s->prec = atoi(e);
    if (const char *e = getenv("NBMI_PREC_NEAR")) s->prec_near = atof(e);
    if (const char *e = getenv("NBMI_SORT_BITS")) {
        const int b = atoi(e);
        if (b >= 8 && b <= 63) s->sort_bits = b;
    }
    if (const char *e = getenv("NBMI_WALK_BLOCK")) {
        const int b = atoi(e);
        if (b == 64 || b == 128 || b == 256) s->walk_block = b;
    }
}

static int create_impl(nbmi_sim *s, const double *pos, const double *vel, const double *mass) {
    const int64_t n = s->n;
    const int64_t c = s->cap > n ? s->cap : n;  // rows allocated (owner mode keeps head room for immigrants)
    NBMI_HIP_CHECK(hipSetDevice(s->device));
    NBMI_HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (auto &e : s->ev) NBMI_HIP_CHECK(hipEventCreate(&e));
    if (alloc_bodies(s, &s->buf[0], c) || alloc_bodies(s, &s->buf[1], c)) return -2;
    if (dev_alloc(s, &s->posm_s, c) || dev_alloc(s, &s->colors, 3 * (c ? c : 1)) || dev_alloc(s, &s->info, 1)) return -2;
    void *stage = nullptr;
    NBMI_HIP_CHECK(hipMalloc(&stage, (size_t)(c ? c : 1) * 7 * sizeof(double)));
    s->allocs.push_back(stage);
    s->stage = stage;
    NBMI_HIP_CHECK(hipMemsetAsync(s->info, 0, sizeof(TreeInfo), s->stream));
    NBMI_HIP_CHECK(hipMemsetAsync(s->colors, 0, (size_t)(c ? c : 1) * 3 * sizeof(float), s->stream));
    if (s->method == NBMI_METHOD_BARNES_HUT) {
        // reference: max_nodes = min(8M, 4N) (simulation.py:477), + slack for tiny N; owner mode: + received trees
        s->node_capacity = node_rows_for(c) + s->node_extra;
        const int64_t own_rows = node_rows_for(c);
        if (dev_alloc(s, &s->key_hi, c) || dev_alloc(s, &s->key_lo, c) || dev_alloc(s, &s->hi_s, c) ||
            dev_alloc(s, &s->lo_s, c) || ((s->owner || s->prec) && dev_alloc(s, &s->p64_s, c)) || dev_alloc(s, &s->idx, c) ||
            dev_alloc(s, &s->perm, c) || dev_alloc(s, &s->delta, c) || dev_alloc(s, &s->Pex, c + 1) ||
            dev_alloc(s, &s->S, c + 1) || dev_alloc(s, &s->sub_tot, (c + 1) / kScanTile + 2) ||
            dev_alloc(s, &s->sub_cnt, (c + 1) / kScanTile + 2) || dev_alloc(s, &s->subPex, (c + 1) / kScanTile + 2) ||
            dev_alloc(s, &s->T, (c + 1) / kScanTile + 2) ||
            dev_alloc(s, &s->nodes, s->node_capacity + 2) || dev_alloc(s, &s->nodes64, s->node_capacity + 2) ||
            dev_alloc(s, &s->node_level, own_rows) || dev_alloc(s, &s->node_ref, own_rows) ||
            (s->walk_stack && dev_alloc(s, &s->child_tab, (size_t)8 * own_rows)) ||
            (s->prec && dev_alloc(s, &s->diag64, own_rows)) ||
            dev_alloc(s, &s->xcd_bounds, 16) || dev_alloc(s, &s->wave_cycles, (size_t)4 * ((c + 63) / 64 + 8)) ||
            dev_alloc(s, &s->wave_flag, (size_t)(c + 63) / 64 + 64) || dev_alloc(s, &s->sub_flag, (c + 1) / kScanTile + 2) ||
            (s->force_prec != 1 && s->softening > 1e-12 && s->node_capacity + 2 <= kMaxNodeDRows &&
             dev_alloc(s, &s->nodesd, s->node_capacity + 2)) ||
            false)
            return -2;
        s->own_node_rows = own_rows;
        s->tmp_sort_bytes = nbmi::sort_pairs_temp_bytes((size_t)c, 0, 63);
        char *t = nullptr;
        if (dev_alloc(s, &t, s->tmp_sort_bytes + 256)) return -2;
        s->tmp_sort = t;
        NBMI_HIP_CHECK(nbmi::sort_init_temp(t, s->stream));
        if (dev_alloc(s, &s->wtab, 1) || upload_walk_table(s)) return -2;
    }
    // upload AoS host arrays through the staging buffer and split to SoA
    double *dpos = (double *)s->stage, *dvel = dpos + 3 * n, *dm = dvel + 3 * n;
    if (n > 0 && pos) {  // pos == NULL: the caller fills buf[0] on the device (nbmi_create_generated)
        NBMI_HIP_CHECK(hipMemcpyAsync(dpos, pos, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, s->stream));
        NBMI_HIP_CHECK(hipMemcpyAsync(dvel, vel, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, s->stream));
        NBMI_HIP_CHECK(hipMemcpyAsync(dm, mass, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s->stream));
        k_split_state<<<nblocks(n), kBlock, 0, s->stream>>>(dpos, dvel, dm, s->buf[0], n);
        NBMI_HIP_CHECK(hipGetLastError());
    }
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->shard_begin = 0;
    s->shard_end = n;
    s->nt = n;
    s->t_hi = s->hi_s;
    s->t_lo = s->lo_s;
    s->t_posm = s->posm_s;
    return 0;
}

nbmi_sim *nbmi_create(int64_t n, const double *pos, const double *vel, const double *mass, double G,
                      double softening, double damping, double theta, int method, int device) {
    nbmi::clear_error();
    if (n < 0 || n > kMaxBodies || (n > 0 && (!pos || !vel || !mass))) {
        nbmi::set_error("nbmi_create: bad arguments (n=%lld)", (long long)n);
        return nullptr;
    }
    if (method != NBMI_METHOD_BARNES_HUT && method != NBMI_METHOD_DIRECT) {
        nbmi::set_error("nbmi_create: unknown method %d", method);
        return nullptr;
    }
    if (!(softening >= 0.0) || !(theta >= 0.0)) {
        nbmi::set_error("nbmi_create: softening and theta must be >= 0");
        return nullptr;
    }
    int count = nbmi_device_count();
    if (count <= 0) {
        nbmi::set_error("nbmi_create: no HIP device available");
        return nullptr;
    }
    if (device < 0 || device >= count) {
        nbmi::set_error("nbmi_create: device %d out of range (have %d)", device, count);
        return nullptr;
    }
    nbmi_sim *s = new nbmi_sim();
    s->n = n; s->method = method; s->device = device;
    s->G = G; s->softening = softening; s->damping = damping; s->theta = theta;
    read_env_knobs(s);
    if (create_impl(s, pos, vel, mass) != 0) {
        std::string keep = nbmi::get_error();
        nbmi_destroy(s);
        nbmi::set_error("%s", keep.c_str());
        return nullptr;
    }
    if (method == NBMI_METHOD_DIRECT && n > 0 && mass[0] > 0.0) {
        bool same = true;
        for (int64_t i = 1; i < n && same; i++) same = mass[i] == mass[0];
        if (same) s->uniform_gm = G * mass[0];
    }
    return s;
}

nbmi_sim *nbmi_create_generated(int distribution, int64_t n, double spawn_radius, uint64_t seed, double G,
                                double softening, double damping, double theta, int method, int device) {
    nbmi::clear_error();
    if (n < 0 || n > kMaxBodies || distribution < NBMI_IC_GALAXY || distribution > NBMI_IC_CLUSTER ||
        (method != NBMI_METHOD_BARNES_HUT && method != NBMI_METHOD_DIRECT) || !(softening >= 0.0) || !(theta >= 0.0) ||
        !(spawn_radius > 0.0)) {
        nbmi::set_error("nbmi_create_generated: bad arguments (n=%lld, distribution=%d)", (long long)n, distribution);
        return nullptr;
    }
    const int count = nbmi_device_count();
    if (count <= 0 || device < 0 || device >= count) {
        nbmi::set_error("nbmi_create_generated: no HIP device %d (have %d)", device, count);
        return nullptr;
    }
    nbmi_sim *s = new nbmi_sim();
    s->n = n; s->method = method; s->device = device;
    s->G = G; s->softening = softening; s->damping = damping; s->theta = theta;
    read_env_knobs(s);
    int rc = create_impl(s, nullptr, nullptr, nullptr);
    if (rc == 0) {
        Bodies &b = s->buf[0];
        rc = nbmi::ic_generate(distribution, n, spawn_radius, G, seed, nbmi::IcArrays{b.x, b.y, b.z, b.vx, b.vy, b.vz, b.m, b.id},
                               s->stream);
    }
    if (rc != 0) {
        std::string keep = nbmi::get_error();
        nbmi_destroy(s);
        nbmi::set_error("%s", keep.c_str());
        return nullptr;
    }
    return s;
}

void nbmi_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    nbmi::philox4x32_10(counter, key, out);
}

int nbmi_get_masses_f64(nbmi_sim *s, double *out) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!out) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    Bodies cur = s->buf[s->curbuf];
    // scatter to the caller's order through the 3-component un-permute (components 1, 2 unused)
    k_unperm3_f64<<<nblocks(n), kBlock, 0, s->stream>>>(cur.m, cur.m, cur.m, s->owner ? nullptr : cur.id, n, (double *)s->stage);
    NBMI_HIP_CHECK(hipGetLastError());
    std::vector<double> tmp((size_t)n * 3);
    NBMI_HIP_CHECK(hipMemcpyAsync(tmp.data(), s->stage, (size_t)n * 24, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    for (int64_t i = 0; i < n; i++) out[i] = tmp[3 * i];
    return 0;
}

int nbmi_step(nbmi_sim *s, double dt, int substeps) {
    if (int rc = check_handle(s)) return rc;
    if (substeps < 0) {
        nbmi::set_error("nbmi_step: substeps < 0");
        return NBMI_ERR_ARG;
    }
    if (s->n == 0) return 0;
    if (s->owner) {
        nbmi::set_error("nbmi_step: this handle is in owner mode, use the nbmi_owner_* calls");
        return NBMI_ERR_ARG;
    }
    if (substeps > 1 && (s->shard_begin != 0 || s->shard_end != s->n)) {
        nbmi::set_error("nbmi_step: a sharded handle needs nbmi_import_ranks between steps (substeps must be 1)");
        return NBMI_ERR_ARG;
    }
    for (int k = 0; k < substeps; k++) {
        if (s->method == NBMI_METHOD_BARNES_HUT) {
            const int evb = s->timers ? 0 : -1;
            s->step_dt = dt;
            const int rc_tree = enqueue_tree(s, evb, false);
            s->step_dt = 0.0;
            if (rc_tree) return rc_tree;
            if (int rc = enqueue_walk(s, true, dt, nullptr)) return rc;
            if (s->timers) {
                NBMI_HIP_CHECK(hipEventRecord(s->ev[4], s->stream));
                NBMI_HIP_CHECK(hipEventSynchronize(s->ev[4]));
                for (int p = 0; p < 4; p++) {
                    float ms = 0.f;
                    NBMI_HIP_CHECK(hipEventElapsedTime(&ms, s->ev[p], s->ev[p + 1]));
                    s->ms[p] += ms;
                }
                s->timed_steps++;
            }
            // a full (unsharded) step leaves every rank of the other buffer written
            // (sharded handles: the ranks outside [shard_begin, shard_end) arrive via nbmi_import_ranks)
            s->curbuf ^= 1;
            s->steps_taken++;
            s->tree_valid = false;
            s->maxabs_fused = s->fuse_maxabs && s->shard_begin == 0 && s->shard_end == s->n && !s->walk_stack && !s->walk_lane;
        } else {
            if (s->timers) NBMI_HIP_CHECK(hipEventRecord(s->ev[0], s->stream));
            if (int rc = launch_direct<true>(s, dt, nullptr)) return rc;
            if (s->timers) {
                NBMI_HIP_CHECK(hipEventRecord(s->ev[1], s->stream));
                NBMI_HIP_CHECK(hipEventSynchronize(s->ev[1]));
                float ms = 0.f;
                NBMI_HIP_CHECK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
                s->ms[3] += ms;
                s->timed_steps++;
            }
            s->curbuf ^= 1;
            s->steps_taken++;
        }
    }
    return 0;
}

int64_t nbmi_step_count(nbmi_sim *s) { return s ? s->steps_taken : -1; }

int nbmi_compute_colors(nbmi_sim *s, double max_speed) {
    if (int rc = check_handle(s)) return rc;
    if (s->n == 0) return 0;
    k_colors<<<nblocks(s->n), kBlock, 0, s->stream>>>(s->buf[s->curbuf], s->n, max_speed, s->owner, s->colors);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

int nbmi_sync(nbmi_sim *s) {
    if (int rc = check_handle(s)) return rc;
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->method == NBMI_METHOD_BARNES_HUT) return check_device_error(s);
    return 0;
}

static int get3(nbmi_sim *s, const double *a, const double *b, const double *c, void *out, bool f32) {
    const int64_t n = s->n;
    if (n == 0) return 0;
    Bodies cur = s->buf[s->curbuf];
    const int32_t *id = s->owner ? nullptr : cur.id;
    if (f32) k_unperm3_f32<<<nblocks(n), kBlock, 0, s->stream>>>(a, b, c, id, n, (float *)s->stage);
    else k_unperm3_f64<<<nblocks(n), kBlock, 0, s->stream>>>(a, b, c, id, n, (double *)s->stage);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemcpyAsync(out, s->stage, (size_t)n * 3 * (f32 ? sizeof(float) : sizeof(double)),
                                  hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->method == NBMI_METHOD_BARNES_HUT) return check_device_error(s);
    return 0;
}

int nbmi_get_positions_f32(nbmi_sim *s, float *out) {
    if (int rc = check_handle(s)) return rc;
    if (!out && s->n) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    Bodies cur = s->buf[s->curbuf];
    return get3(s, cur.x, cur.y, cur.z, out, true);
}
int nbmi_get_positions_f64(nbmi_sim *s, double *out) {
    if (int rc = check_handle(s)) return rc;
    if (!out && s->n) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    Bodies cur = s->buf[s->curbuf];
    return get3(s, cur.x, cur.y, cur.z, out, false);
}
int nbmi_get_velocities_f64(nbmi_sim *s, double *out) {
    if (int rc = check_handle(s)) return rc;
    if (!out && s->n) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    Bodies cur = s->buf[s->curbuf];
    return get3(s, cur.vx, cur.vy, cur.vz, out, false);
}
int nbmi_get_colors_f32(nbmi_sim *s, float *out) {
    if (int rc = check_handle(s)) return rc;
    if (s->n == 0) return 0;
    if (!out) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    NBMI_HIP_CHECK(hipMemcpyAsync(out, s->colors, (size_t)s->n * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_set_state(nbmi_sim *s, const double *pos, const double *vel) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!pos || !vel) { nbmi::set_error("null input"); return NBMI_ERR_ARG; }
    double *dpos = (double *)s->stage, *dvel = dpos + 3 * n;
    NBMI_HIP_CHECK(hipMemcpyAsync(dpos, pos, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, s->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(dvel, vel, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, s->stream));
    k_set_state_perm<<<nblocks(n), kBlock, 0, s->stream>>>(dpos, dvel, s->buf[s->curbuf], n);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemsetAsync(&s->info->sticky_error, 0, sizeof(int), s->stream));  // a fresh state: drop a pending capacity error
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    k_set_all64<<<1, 1, 0, s->stream>>>(s->info, 0);  // a new state: the "most of the system asks" history is the old system's
    NBMI_HIP_CHECK(hipGetLastError());
    s->tree_valid = false;
    s->maxabs_fused = false;
    return 0;
}

int nbmi_build_tree(nbmi_sim *s) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT) { nbmi::set_error("not a Barnes-Hut handle"); return NBMI_ERR_ARG; }
    if (s->n == 0) return 0;
    if (int rc = enqueue_tree(s, -1)) return rc;
    return check_device_error(s);
}

int nbmi_get_accelerations_f64(nbmi_sim *s, double *out) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!out) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    double *acc = (double *)s->stage;
    if (s->method == NBMI_METHOD_BARNES_HUT) {
        if (int rc = enqueue_tree(s, -1)) return rc;
        NBMI_HIP_CHECK(hipMemsetAsync(&s->info->wave_visits, 0, 17 * sizeof(unsigned long long), s->stream));
        if (int rc = enqueue_walk(s, false, 0.0, acc)) return rc;
    } else {
        if (int rc = launch_direct<false>(s, 0.0, acc)) return rc;
    }
    NBMI_HIP_CHECK(hipMemcpyAsync(out, acc, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->method == NBMI_METHOD_BARNES_HUT) return check_device_error(s);
    return 0;
}

int nbmi_tree_stats(nbmi_sim *s, int64_t *num_nodes, int32_t *max_depth, double *bounds) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT) { nbmi::set_error("not a Barnes-Hut handle"); return NBMI_ERR_ARG; }
    if (s->n == 0) {  // reference: empty root leaf, bounds = 0*1.1+10
        if (num_nodes) *num_nodes = 1;
        if (max_depth) *max_depth = 0;
        if (bounds) *bounds = 10.0;
        return 0;
    }
    if (max_depth) {
        int gb = nblocks(s->nt);
        if (gb > 64) gb = 64;
        k_max_level<<<gb, kBlock, 0, s->stream>>>(s->delta, s->nt, s->info);
        NBMI_HIP_CHECK(hipGetLastError());
    }
    TreeInfo h;
    NBMI_HIP_CHECK(hipMemcpyAsync(&h, s->info, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (num_nodes) *num_nodes = h.num_nodes;
    if (max_depth) *max_depth = h.max_level;
    if (bounds) *bounds = h.bounds;
    if (h.error) {
        nbmi::set_error("octree needs %lld nodes, more than the %lld rows allocated", (long long)h.num_nodes,
                        (long long)s->node_capacity);
        return NBMI_ERR_CAPACITY;
    }
    return 0;
}

int nbmi_get_keys(nbmi_sim *s, uint64_t *key_hi, uint64_t *key_lo) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT || !s->tree_valid) {
        nbmi::set_error("nbmi_get_keys: call nbmi_build_tree first");
        return NBMI_ERR_ARG;
    }
    const int64_t n = s->n;
    if (n == 0) return 0;
    uint64_t *o_hi = (uint64_t *)s->stage, *o_lo = o_hi + n;
    k_keys_to_orig<<<nblocks(n), kBlock, 0, s->stream>>>(s->hi_s, s->lo_s, s->perm, s->buf[s->curbuf].id, n, s->hilbert ? 1 : 0, o_hi, o_lo);
    NBMI_HIP_CHECK(hipGetLastError());
    if (key_hi) NBMI_HIP_CHECK(hipMemcpyAsync(key_hi, o_hi, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    if (key_lo) NBMI_HIP_CHECK(hipMemcpyAsync(key_lo, o_lo, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_get_sort_keys(nbmi_sim *s, uint64_t *key_hi, uint64_t *key_lo) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT || !s->tree_valid) {
        nbmi::set_error("nbmi_get_sort_keys: call nbmi_build_tree first");
        return NBMI_ERR_ARG;
    }
    const int64_t n = s->n;
    if (n == 0) return 0;
    uint64_t *o_hi = (uint64_t *)s->stage, *o_lo = o_hi + n;
    k_keys_to_orig<<<nblocks(n), kBlock, 0, s->stream>>>(s->hi_s, s->lo_s, s->perm, s->buf[s->curbuf].id, n, 0, o_hi, o_lo);
    NBMI_HIP_CHECK(hipGetLastError());
    if (key_hi) NBMI_HIP_CHECK(hipMemcpyAsync(key_hi, o_hi, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    if (key_lo) NBMI_HIP_CHECK(hipMemcpyAsync(key_lo, o_lo, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_get_order(nbmi_sim *s, int32_t *order) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT || !s->tree_valid) {
        nbmi::set_error("nbmi_get_order: call nbmi_build_tree first");
        return NBMI_ERR_ARG;
    }
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!order) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    k_order<<<nblocks(n), kBlock, 0, s->stream>>>(s->perm, s->buf[s->curbuf].id, n, (int32_t *)s->stage);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemcpyAsync(order, s->stage, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_get_cells(nbmi_sim *s, int32_t *level, uint64_t *key, int64_t capacity) {
    if (int rc = check_handle(s)) return rc;
    if (s->method != NBMI_METHOD_BARNES_HUT || !s->tree_valid) {
        nbmi::set_error("nbmi_get_cells: call nbmi_build_tree first");
        return NBMI_ERR_ARG;
    }
    int64_t nn = 0;
    if (int rc = nbmi_tree_stats(s, &nn, nullptr, nullptr)) return rc;
    if (nn > capacity || !level || !key) {
        nbmi::set_error("nbmi_get_cells: need room for %lld cells", (long long)nn);
        return NBMI_ERR_ARG;
    }
    if (s->n == 0) { level[0] = 0; key[0] = 0; return 0; }
    int32_t *dl = nullptr;
    uint64_t *dk = nullptr;
    NBMI_HIP_CHECK(hipMalloc((void **)&dl, (size_t)nn * 4));
    NBMI_HIP_CHECK(hipMalloc((void **)&dk, (size_t)nn * 8));
    k_cells<<<nblocks(nn), kBlock, 0, s->stream>>>(s->node_ref, s->node_level, s->t_hi, nn, s->hilbert ? 1 : 0, dl, dk);
    hipError_t e1 = hipMemcpyAsync(level, dl, (size_t)nn * 4, hipMemcpyDeviceToHost, s->stream);
    hipError_t e2 = hipMemcpyAsync(key, dk, (size_t)nn * 8, hipMemcpyDeviceToHost, s->stream);
    hipError_t e3 = hipStreamSynchronize(s->stream);
    (void)hipFree(dl);
    (void)hipFree(dk);
    NBMI_HIP_CHECK(e1);
    NBMI_HIP_CHECK(e2);
    NBMI_HIP_CHECK(e3);
    return 0;
}

int nbmi_enable_timers(nbmi_sim *s, int enable) {
    if (int rc = check_handle(s)) return rc;
    s->timers = enable != 0;
    return 0;
}

int nbmi_get_timers(nbmi_sim *s, double *ms5, int64_t *count, int reset) {
    if (int rc = check_handle(s)) return rc;
    if (ms5) memcpy(ms5, s->ms, sizeof(s->ms));
    if (count) *count = s->timed_steps;
    if (reset) {
        memset(s->ms, 0, sizeof(s->ms));
        s->timed_steps = 0;
    }
    return 0;
}

int nbmi_walk_counters(nbmi_sim *s, int64_t *out8 /* 17 entries */) {
    if (int rc = check_handle(s)) return rc;
    int64_t *out3 = out8;
    if (!out3) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    TreeInfo h;
    NBMI_HIP_CHECK(hipMemcpyAsync(&h, s->info, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    out3[0] = (int64_t)h.wave_visits; out3[1] = (int64_t)h.lane_visits; out3[2] = (int64_t)h.lane_accepts;
    for (int w = 0; w < 4; w++) out8[3 + w] = (int64_t)h.win_miss[w];
    out8[7] = (int64_t)h.jumps;
    for (int x = 0; x < 8; x++) out8[8 + x] = (int64_t)h.xcd_visits[x];
    out8[16] = (int64_t)h.band_visits;
    return 0;
}

int nbmi_set_shard(nbmi_sim *s, int64_t begin, int64_t end) {
    if (int rc = check_handle(s)) return rc;
    if (begin < 0 || end < begin || end > s->n) {
        nbmi::set_error("nbmi_set_shard: bad range [%lld,%lld) for n=%lld", (long long)begin, (long long)end, (long long)s->n);
        return NBMI_ERR_ARG;
    }
    s->shard_begin = begin;
    s->shard_end = end;
    s->maxabs_fused = false;
    return 0;
}

int nbmi_export_shard(nbmi_sim *s, void *dev_rows) {
    if (int rc = check_handle(s)) return rc;
    const int64_t c = s->shard_end - s->shard_begin;
    if (c <= 0) return 0;
    if (!dev_rows) { nbmi::set_error("null device buffer"); return NBMI_ERR_ARG; }
    k_pack_rows<<<nblocks(c), kBlock, 0, s->stream>>>(s->buf[s->curbuf], s->shard_begin, s->shard_end, (double *)dev_rows);
    NBMI_HIP_CHECK(hipGetLastError());
    if (s->exchange_sync) NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_import_ranks(nbmi_sim *s, const void *dev_rows, int64_t begin, int64_t end) {
    if (int rc = check_handle(s)) return rc;
    if (begin < 0 || end < begin || end > s->n) { nbmi::set_error("nbmi_import_ranks: bad range"); return NBMI_ERR_ARG; }
    const int64_t c = end - begin;
    if (c == 0) return 0;
    if (!dev_rows) { nbmi::set_error("null device buffer"); return NBMI_ERR_ARG; }
    k_unpack_rows<<<nblocks(c), kBlock, 0, s->stream>>>(s->buf[s->curbuf], begin, end, (const double *)dev_rows);
    NBMI_HIP_CHECK(hipGetLastError());
    if (s->exchange_sync) NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->maxabs_fused = false;
    s->tree_valid = false;
    return 0;
}

// =========================================================================================
// Owner mode (multi-GPU stage 2).  See include/nbmi.h for the per-step protocol.
// =========================================================================================
namespace {
int owner_check(nbmi_sim *s, const char *what) {
    if (int rc = check_handle(s)) return rc;
    if (!s->owner) {
        nbmi::set_error("%s: not an owner-mode handle (nbmi_create_owner)", what);
        return NBMI_ERR_ARG;
    }
    return 0;
}
// `batch` arrays of n (+1 output) entries, `stride` apart
int enqueue_iscan(nbmi_sim *s, const int32_t *in, int64_t n, int32_t *out, int batch = 1, int64_t stride = 0) {
    const int64_t ntiles = (n + 1 + kScanTile - 1) / kScanTile;
    const int64_t tstride = s->let_tile_stride;
    const dim3 grid((unsigned)ntiles, (unsigned)batch);
    k_iscan_reduce<<<grid, kBlock, 0, s->stream>>>(in, n, stride, s->let_tiles, tstride);
    k_iscan_tiles<<<dim3(1, (unsigned)batch), kBlock, 0, s->stream>>>(s->let_tiles, ntiles, tstride);
    k_iscan_apply<<<grid, kBlock, 0, s->stream>>>(in, n, stride, s->let_tiles, tstride, out);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}
}  // namespace

nbmi_sim *nbmi_create_owner(int64_t n, const double *pos, const double *vel, const double *mass, const int32_t *ids,
                            int64_t capacity, int64_t let_capacity, int world, int rank, double G, double softening,
                            double damping, double theta, int device) {
    nbmi::clear_error();
    if (n < 0 || capacity < n || capacity < 1 || capacity > kMaxBodies || (n > 0 && (!pos || !vel || !mass || !ids)) ||
        world < 1 || world > kMaxWorld || rank < 0 || rank >= world || let_capacity < 0 || !(softening >= 0.0) ||
        !(theta >= 0.0)) {
        nbmi::set_error("nbmi_create_owner: bad arguments (n=%lld, capacity=%lld, world=%d, rank=%d)", (long long)n,
                        (long long)capacity, world, rank);
        return nullptr;
    }
    const int count = nbmi_device_count();
    if (count <= 0 || device < 0 || device >= count) {
        nbmi::set_error("nbmi_create_owner: no HIP device %d (have %d)", device, count);
        return nullptr;
    }
    nbmi_sim *s = new nbmi_sim();
    s->n = n; s->cap = capacity; s->method = NBMI_METHOD_BARNES_HUT; s->device = device;
    s->G = G; s->softening = softening; s->damping = damping; s->theta = theta;
    s->owner = true; s->world = world; s->rank = rank;
    s->let_capacity = let_capacity;
    // the own tree sits in the MIDDLE of the walk array: room for received pieces in front of it (lower ranks) and
    // behind it (higher ranks), a jump node at row 0
    s->node_extra = world > 1 ? 2 * let_capacity + 2 : 0;
    s->own_base = world > 1 ? let_capacity + 1 : 0;
    read_env_knobs(s);
    int rc = create_impl(s, pos, vel, mass);
    if (rc == 0) {
        const int64_t c = s->cap;
        // per-destination work arrays over the OWN tree's rows (diff / scan / keep), one row set per rank
        s->let_stride = s->own_node_rows + 8;
        s->let_tile_stride = s->let_stride / kScanTile + 4;
        const size_t all = (size_t)s->let_stride * world;
        if (dev_alloc(s, &s->let_split, kMaxWorld) || dev_alloc(s, &s->let_dest, c) ||
            dev_alloc(s, &s->let_counts, 2 * kMaxWorld) || dev_alloc(s, &s->let_dead, c) ||
            dev_alloc(s, &s->let_diff, all) || dev_alloc(s, &s->let_scan, all) || dev_alloc(s, &s->let_keep, all) ||
            dev_alloc(s, &s->let_tiles, (size_t)s->let_tile_stride * world) || dev_alloc(s, &s->let_ranges, 2 * kBoxesPerRank) ||
            dev_alloc(s, &s->let_supers, (size_t)6 * kSupersPerRank * kMaxWorld) ||
            dev_alloc(s, &s->let_megas, (size_t)6 * kMegasPerRank * kMaxWorld) || dev_alloc(s, &s->let_rankbox, 6 * kMaxWorld) ||
            dev_alloc(s, &s->chain_r, kChainLevels) || dev_alloc(s, &s->own_links, kOwnLinks))
            rc = -2;
        if (rc == 0 && (hipHostMalloc((void **)&s->h_info, sizeof(TreeInfo)) != hipSuccess || hipEventCreate(&s->ev_info) != hipSuccess ||
                        hipMemsetAsync(s->let_dead, 0, (size_t)c, s->stream) != hipSuccess)) {
            nbmi::set_error("nbmi_create_owner: host buffer / event allocation failed");
            rc = -2;
        }
    }
    if (rc == 0 && n > 0) {  // global ids instead of the row numbers k_split_state wrote
        hipError_t e = hipMemcpyAsync(s->stage, ids, (size_t)n * 4, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            k_copy_ids<<<nblocks(n), kBlock, 0, s->stream>>>((const int32_t *)s->stage, s->buf[0].id, n);
            e = hipStreamSynchronize(s->stream);
        }
        if (e != hipSuccess) { nbmi::set_error("nbmi_create_owner: id upload failed: %s", hipGetErrorString(e)); rc = -2; }
    }
    if (rc != 0) {
        std::string keep = nbmi::get_error();
        nbmi_destroy(s);
        nbmi::set_error("%s", keep.c_str());
        return nullptr;
    }
    return s;
}

int64_t nbmi_owner_count(nbmi_sim *s) { return s ? s->n : -1; }
int nbmi_owner_boxes_per_rank(void) { return kBoxesPerRank; }

int nbmi_owner_get_ids(nbmi_sim *s, int32_t *out) {
    if (int rc = owner_check(s, "nbmi_owner_get_ids")) return rc;
    if (s->n == 0) return 0;
    if (!out) { nbmi::set_error("null output"); return NBMI_ERR_ARG; }
    NBMI_HIP_CHECK(hipMemcpyAsync(out, s->buf[s->curbuf].id, (size_t)s->n * 4, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    return 0;
}

int nbmi_owner_maxabs(nbmi_sim *s, void *dev_maxabs) {
    if (int rc = owner_check(s, "nbmi_owner_maxabs")) return rc;
    if (!dev_maxabs) { nbmi::set_error("nbmi_owner_maxabs: null buffer"); return NBMI_ERR_ARG; }
    if (int rc = enqueue_maxabs(s)) return rc;  // also clears the per-step tree header
    // a non-negative double and its bit pattern order the same way: the word IS the double
    NBMI_HIP_CHECK(hipMemcpyAsync(dev_maxabs, &s->info->maxabs_bits, 8, hipMemcpyDeviceToDevice, s->stream));
    if (s->exchange_sync) NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));  // (stream-ordered callers: nbmi_set_exchange_sync(0))
    return 0;
}

int nbmi_owner_sample(nbmi_sim *s, const void *dev_maxabs, void *dev_samples, int nsamples, int nvalid) {
    if (int rc = owner_check(s, "nbmi_owner_sample")) return rc;
    if (!dev_maxabs || !dev_samples || nsamples < 1 || (int64_t)nsamples * s->world > kSampleCap) {
        nbmi::set_error("nbmi_owner_sample: bad arguments (at most %d samples over all ranks)", kSampleCap);
        return NBMI_ERR_ARG;
    }
    hipStream_t st = s->stream;
    NBMI_HIP_CHECK(hipMemcpyAsync(&s->info->maxabs_bits, dev_maxabs, 8, hipMemcpyDeviceToDevice, st));
    if (s->world == 1) return 0;  // one owner: no splitters to find (nbmi_owner_adopt computes the keys it sorts by)
    Bodies cur = s->buf[s->curbuf];
    if (s->n > 0 && s->hilbert) k_keys<true><<<nblocks(s->n), kBlock, 0, st>>>(cur.x, cur.y, cur.z, s->n, s->info, s->key_hi, s->key_lo, s->idx);
    else if (s->n > 0) k_keys<false><<<nblocks(s->n), kBlock, 0, st>>>(cur.x, cur.y, cur.z, s->n, s->info, s->key_hi, s->key_lo, s->idx);
    if (nvalid < 1 || nvalid > nsamples) nvalid = nsamples;
    k_key_samples<<<(nsamples + kBlock - 1) / kBlock, kBlock, 0, st>>>(s->key_hi, s->n, nsamples, nvalid, (uint64_t *)dev_samples);
    NBMI_HIP_CHECK(hipGetLastError());
    if (s->exchange_sync) NBMI_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

int nbmi_owner_partition(nbmi_sim *s, const void *dev_all_samples, int total_samples, void *dev_send_rows,
                         int64_t *counts /* world, host */) {
    if (int rc = owner_check(s, "nbmi_owner_partition")) return rc;
    if (!dev_all_samples || !counts || total_samples < 1 || total_samples > kSampleCap || (s->n > 0 && !dev_send_rows)) {
        nbmi::set_error("nbmi_owner_partition: bad arguments");
        return NBMI_ERR_ARG;
    }
    hipStream_t st = s->stream;
    const int64_t n = s->n, stride = s->let_stride;
    const int W = s->world;
    for (int j = 0; j < W; j++) counts[j] = 0;
    if (W == 1) {  // nobody to hand bodies to (the dead flags stay clear)
        s->n_leaving = 0;
        return 0;
    }
    k_splitters<<<(total_samples + kBlock - 1) / kBlock, kBlock, 0, st>>>((const uint64_t *)dev_all_samples, total_samples, W, s->let_split);
    if (n > 0) {
        // only the bodies whose key has left this rank's range travel; everybody else stays where it is
        k_dest<<<nblocks(n), kBlock, 0, st>>>(s->key_hi, n, s->let_split, W, s->let_dest, s->idx);
        k_emigrant_flags<<<nblocks(n), kBlock, 0, st>>>(s->let_dest, n, W, s->rank, stride, s->let_keep, s->let_dead);
        if (int rc = enqueue_iscan(s, s->let_keep, n, s->let_scan, W, stride)) return rc;
        k_let_counts<<<1, 64, 0, st>>>(s->let_scan, n, stride, W, s->rank, s->let_counts);
        k_pack_emigrants<<<nblocks(n), kBlock, 0, st>>>(s->buf[s->curbuf], s->let_dest, n, s->rank, stride, s->let_scan,
                                                       s->let_counts + W, (double *)dev_send_rows);
        NBMI_HIP_CHECK(hipGetLastError());
        NBMI_HIP_CHECK(hipMemcpyAsync(counts, s->let_counts, (size_t)W * 8, hipMemcpyDeviceToHost, st));
    }
    NBMI_HIP_CHECK(hipStreamSynchronize(st));
    int64_t left = 0;
    for (int j = 0; j < W; j++) left += counts[j];
    s->n_leaving = left;
    return 0;
}

int nbmi_owner_chain_doubles(void) { return kChainDoubles; }

int nbmi_owner_adopt(nbmi_sim *s, const void *dev_recv_rows, int64_t n_recv, const void *dev_maxabs, void *dev_boxes,
                     void *dev_chain) {
    if (int rc = owner_check(s, "nbmi_owner_adopt")) return rc;
    const int64_t n_old = s->n, n_work = n_old + n_recv, n_new = n_old - s->n_leaving + n_recv;
    if (n_recv < 0 || n_work > s->cap) {
        nbmi::set_error("nbmi_owner_adopt: %lld + %lld bodies do not fit the capacity of %lld (raise the head room)",
                        (long long)n_old, (long long)n_recv, (long long)s->cap);
        return NBMI_ERR_CAPACITY;
    }
    if ((n_recv > 0 && !dev_recv_rows) || !dev_maxabs || !dev_boxes || (s->world > 1 && !dev_chain)) {
        nbmi::set_error("nbmi_owner_adopt: null buffer");
        return NBMI_ERR_ARG;
    }
    hipStream_t st = s->stream;
    Bodies cur = s->buf[s->curbuf];
    // immigrants go behind the rows already here; the emigrants' rows stay in place, flagged dead
    if (n_recv > 0) {
        k_unpack_rows<<<nblocks(n_recv), kBlock, 0, st>>>(cur, n_old, n_work, (const double *)dev_recv_rows);
        NBMI_HIP_CHECK(hipMemsetAsync(s->let_dead + n_old, 0, (size_t)n_recv, st));
    }
    s->n = n_new; s->nt = n_new; s->shard_begin = 0; s->shard_end = n_new;
    s->n_leaving = 0;
    // the tree header of this step: cleared, then the GLOBAL extent (every rank builds inside the same root cube)
    NBMI_HIP_CHECK(hipMemsetAsync(s->info, 0, offsetof(TreeInfo, wave_visits), st));
    NBMI_HIP_CHECK(hipMemcpyAsync(&s->info->maxabs_bits, dev_maxabs, 8, hipMemcpyDeviceToDevice, st));
    s->step_dt = s->owner_dt;  // "auto" force precision is decided during the build
    struct DtScope { nbmi_sim *s; ~DtScope() { s->step_dt = 0.0; } } dt_scope{s};
    if (n_new > 0 && s->world == 1) {
        // one owner: the plain build (no dead rows, no boxes to publish)
        if (int rc = enqueue_local_sort(s, -1)) return rc;
        if (int rc = enqueue_global_tree(s, false)) return rc;
    } else if (n_new > 0) {
        // rank of every dead row among the dead rows (exclusive scan of the flags): their place behind the live ones
        k_dead_flags<<<nblocks(n_work), kBlock, 0, st>>>(s->let_dead, n_work, s->let_keep);
        if (int rc = enqueue_iscan(s, s->let_keep, n_work, s->let_scan)) return rc;
        if (int rc = enqueue_local_sort(s, -1, n_work, n_new, s->let_dead)) return rc;
        if (int rc = enqueue_global_tree(s)) return rc;
        // where this rank's bodies are: boxes of the cells of its tree (see k_box_flags).  The node count lives on
        // the device: launch for the row budget, the kernels stop at num_nodes themselves.
        const int64_t rows = s->own_node_rows, ob = s->own_base;
        NBMI_HIP_CHECK(hipMemsetAsync(s->chain_r, 0xff, sizeof(int32_t) * kChainLevels, st));
        k_box_flags<<<nblocks(rows), kBlock, 0, st>>>(s->nodes + ob, s->node_level, s->node_ref, s->hi_s, s->lo_s, s->info, n_new, rows, s->let_keep,
                                                     s->chain_r, ob);
        if (int rc = enqueue_iscan(s, s->let_keep, rows, s->let_scan)) return rc;
        NBMI_HIP_CHECK(hipMemsetAsync(s->let_ranges, 0, sizeof(int32_t) * 2 * kBoxesPerRank, st));
        k_box_ranges<<<nblocks(rows), kBlock, 0, st>>>(s->nodes + ob, s->node_ref, s->let_keep, s->let_scan, s->info, rows, n_new, s->let_ranges, ob);
        // what the other ranks need to know about this rank's two ends (travels with the boxes)
        k_chain_table<<<1, 64, 0, st>>>(s->nodes + ob, s->node_ref, s->delta, s->S, s->T, s->hi_s, s->lo_s, n_new, s->info, s->chain_r, ob,
                                        (ChainTable *)dev_chain);
        k_boxes_init<<<(6 * kBoxesPerRank + kBlock - 1) / kBlock, kBlock, 0, st>>>((double *)dev_boxes);
        k_range_boxes<<<(unsigned)((n_new + (int64_t)kBoxChunk * (kBlock / 64) - 1) / ((int64_t)kBoxChunk * (kBlock / 64))), kBlock, 0, st>>>(
            s->p64_s, s->let_ranges, s->let_scan + rows, n_new, (double *)dev_boxes);
    } else {
        std::vector<double> empty(6 * kBoxesPerRank);
        for (int k = 0; k < 6 * kBoxesPerRank; k++) empty[k] = (k % 6) < 3 ? INFINITY : -INFINITY;
        NBMI_HIP_CHECK(hipMemcpyAsync(dev_boxes, empty.data(), empty.size() * 8, hipMemcpyHostToDevice, st));
        if (dev_chain) NBMI_HIP_CHECK(hipMemsetAsync(dev_chain, 0, sizeof(ChainTable), st));  // n = 0: no bodies, no part in the tree
        NBMI_HIP_CHECK(hipStreamSynchronize(st));  // `empty` dies here
    }
    NBMI_HIP_CHECK(hipGetLastError());
    // the tree header travels to the host behind the build; nbmi_owner_export_let / _step read it after their own wait
    NBMI_HIP_CHECK(hipMemcpyAsync(s->h_info, s->info, sizeof(TreeInfo), hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipEventRecord(s->ev_info, st));
    if (s->exchange_sync) NBMI_HIP_CHECK(hipStreamSynchronize(st));
    s->tree_valid = n_new > 0 && s->world > 1;
    return 0;
}

// the float64 records as the exchange sees them: present whenever the handle's trees carry them
static inline NodeD *let_nodesd(const nbmi_sim *s) { return s->force_prec != 1 ? s->nodesd : nullptr; }
int nbmi_owner_let_row_bytes(void) { return kLetRow; }

int nbmi_owner_set_dt(nbmi_sim *s, double dt) {
    if (int rc = owner_check(s, "nbmi_owner_set_dt")) return rc;
    if (!(dt >= 0.0)) { nbmi::set_error("nbmi_owner_set_dt: dt must be >= 0"); return NBMI_ERR_ARG; }
    s->owner_dt = dt;
    return 0;
}

int nbmi_owner_export_let(nbmi_sim *s, const void *dev_boxes, const void *dev_chains, void *dev_let, int64_t *counts /* world, host */) {
    if (int rc = owner_check(s, "nbmi_owner_export_let")) return rc;
    if (!dev_boxes || !dev_let || !counts || (s->world > 1 && !dev_chains)) { nbmi::set_error("nbmi_owner_export_let: null buffer"); return NBMI_ERR_ARG; }
    s->chains = (const ChainTable *)dev_chains;
    for (int j = 0; j < s->world; j++) counts[j] = 0;
    if (s->n == 0 || s->world == 1) return 0;
    hipStream_t st = s->stream;
    const int64_t stride = s->let_stride, ob = s->own_base;
    const int W = s->world;
    const ChainTable *T = s->chains;
    // the own tree becomes this rank's piece of the global pre-order array (a few nodes change; up to 43 are added)
    k_chain_fix<<<1, 64, 0, st>>>(T, W, s->rank, s->nodes + ob, s->nodes64 + ob, let_nodesd(s) ? let_nodesd(s) + ob : nullptr, s->node_level,
                                  s->node_ref, s->chain_r, s->theta > 0.0 ? 1.0 / (s->theta * s->theta) : INFINITY, s->own_node_rows,
                                  s->own_links, s->info);
    NBMI_HIP_CHECK(hipMemcpyAsync(s->h_info, s->info, sizeof(TreeInfo), hipMemcpyDeviceToHost, st));  // (read after the wait below)
    // How many nodes the own tree has is known on the device; the host needs a launch bound.  The previous step's
    // count + 2 % is one (a tree changes by a few nodes in a thousand per step): no wait for this step's header
    // before the kernels are enqueued, ONE wait at the end for the counts - and the header, which says whether the
    // bound held (if not: once more with the exact count).
    int64_t nn = -1;
    if (s->last_nodes > 0 && !s->exchange_sync) {
        nn = s->last_nodes + s->last_nodes / 50 + 4096;  // (covers the handful of cells k_chain_fix adds)
        if (nn > s->own_node_rows) nn = s->own_node_rows;
        if (getenv("NBMI_LET_UNDERESTIMATE")) nn = s->last_nodes / 2;  // test hook: forces the "bound did not hold" repeat
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        if (nn < 0) {
            NBMI_HIP_CHECK(hipStreamSynchronize(st));  // this step's header, behind k_chain_fix
            if (s->h_info->error) return check_device_error(s);
            nn = s->h_info->num_nodes;
        }
        NBMI_HIP_CHECK(hipMemsetAsync(s->let_diff, 0, (size_t)stride * W * 4, st));
        k_super_boxes<<<(W * kSupersPerRank + 63) / 64, 64, 0, st>>>((const double *)dev_boxes, W * kSupersPerRank, s->let_supers);
        k_rank_boxes<<<1, kMaxWorld * kMegasPerRank, 0, st>>>(s->let_supers, W, s->let_megas, s->let_rankbox);
        k_let_mark<<<nblocks(nn), kBlock, 0, st>>>(s->nodes + ob, s->nodes64 + ob, nn, (const double *)dev_boxes, s->let_supers, s->let_megas,
                                                  s->let_rankbox, W, s->rank, s->theta, s->softening * s->softening, s->let_diff, stride,
                                                  s->info, ob);
        k_let_mark_left<<<W, 64, 0, st>>>(T, W, s->rank, (const double *)dev_boxes, s->let_supers, s->let_megas, s->let_rankbox, s->theta,
                                          s->softening * s->softening, s->let_diff, stride, s->info);
        if (int rc = enqueue_iscan(s, s->let_diff, nn, s->let_scan, W, stride)) return rc;
        k_let_keep<<<dim3((unsigned)nblocks(nn), (unsigned)W), kBlock, 0, st>>>(s->let_diff, s->let_scan, nn, stride, s->let_keep, s->info);
        if (int rc = enqueue_iscan(s, s->let_keep, nn, s->let_scan, W, stride)) return rc;
        k_let_counts<<<1, 64, 0, st>>>(s->let_scan, nn, stride, W, s->rank, s->let_counts);
        k_let_compact<<<dim3((unsigned)nblocks(nn), (unsigned)W), kBlock, 0, st>>>(s->nodes + ob, s->nodes64 + ob, let_nodesd(s) ? let_nodesd(s) + ob : nullptr,
                                                                                s->node_level, s->own_links, T, s->let_keep, s->let_scan, nn, stride,
                                                                                s->let_capacity, s->rank, s->let_counts + W, (char *)dev_let,
                                                                                s->info, ob);
        NBMI_HIP_CHECK(hipGetLastError());
        NBMI_HIP_CHECK(hipMemcpyAsync(counts, s->let_counts, (size_t)W * 8, hipMemcpyDeviceToHost, st));
        NBMI_HIP_CHECK(hipStreamSynchronize(st));  // the one wait: counts + (long since) this step's header
        if (s->h_info->error) return check_device_error(s);
        if (s->h_info->num_nodes <= nn) break;
        nn = s->h_info->num_nodes;  // the estimate was too small: the exact count now
    }
    s->last_nodes = s->h_info->num_nodes;
    int64_t total = 0;
    for (int j = 0; j < W; j++) total += counts[j];
    if (total > s->let_capacity) {
        nbmi::set_error("nbmi_owner_export_let: the trees for the other ranks have %lld nodes, more than the %lld rows reserved",
                        (long long)total, (long long)s->let_capacity);
        return NBMI_ERR_CAPACITY;
    }
    return 0;
}

int nbmi_owner_step_facts(nbmi_sim *s, int64_t *out4) {
    if (int rc = owner_check(s, "nbmi_owner_step_facts")) return rc;
    if (!out4) { nbmi::set_error("nbmi_owner_step_facts: null output"); return NBMI_ERR_ARG; }
    out4[0] = out4[1] = 0;
    out4[2] = out4[3] = (int64_t)1 << 62;
    if (s->n == 0 || s->world == 1 || !s->h_info) return 0;
    NBMI_HIP_CHECK(hipEventSynchronize(s->ev_info));  // (long complete: nbmi_owner_export_let waited behind it)
    const TreeInfo &h = *s->h_info;
    if (s->force_prec == 0 && s->nodesd && s->owner_dt > 0.0) { out4[0] = h.ask_waves; out4[1] = h.n_waves; }
    out4[2] = s->own_base + h.own_skip - 1;                        // tree rows that fit in front of the own piece
    out4[3] = s->node_capacity - (s->own_base + h.num_nodes) - 1;  // ... and behind it
    return 0;
}

int nbmi_owner_set_all64(nbmi_sim *s, int verdict) {
    if (int rc = owner_check(s, "nbmi_owner_set_all64")) return rc;
    s->owner_all64 = verdict < 0 ? -1 : (verdict ? 1 : 0);
    return 0;
}

int nbmi_owner_step(nbmi_sim *s, const void *dev_recv, const int64_t *counts, double dt) {
    if (int rc = owner_check(s, "nbmi_owner_step")) return rc;
    if (!counts || (s->world > 1 && !dev_recv)) { nbmi::set_error("nbmi_owner_step: null buffer"); return NBMI_ERR_ARG; }
    if (s->n == 0) return 0;
    hipStream_t st = s->stream;
    if (s->owner_all64 >= 0) k_set_all64<<<1, 1, 0, st>>>(s->info, s->owner_all64);  // the ranks' common verdict
    if (s->world == 1) {  // nothing received: the walk array is the own tree; an overflow freezes the walk and is reported at the next sync
        k_let_finish_alone<<<1, 1, 0, st>>>(s->info);
        NBMI_HIP_CHECK(hipGetLastError());
        if (int rc = enqueue_walk(s, true, dt, nullptr)) return rc;
        s->curbuf ^= 1;
        s->tree_valid = false;
        s->maxabs_fused = s->fuse_maxabs;  // the walk left max |coordinate| of what it wrote
        return 0;
    }
    if (!s->chains) { nbmi::set_error("nbmi_owner_step: call nbmi_owner_export_let first (it cuts the own tree to its piece of the global one)"); return NBMI_ERR_ARG; }
    NBMI_HIP_CHECK(hipEventSynchronize(s->ev_info));  // (long complete: nbmi_owner_export_let waited behind it)
    const TreeInfo &h = *s->h_info;  // as nbmi_owner_export_let left it: the own piece in its global form
    if (h.error) return check_device_error(s);
    // the pieces in rank order: lower ranks packed up against the own piece, higher ranks behind it (see Pieces)
    Pieces P;
    const int W = s->world, me = s->rank;
    int64_t lower = 0, upper = 0, seen = 0;
    for (int j = 0; j < W; j++) {
        if (j == me || counts[j] <= 0) continue;
        if (j < me) lower += counts[j]; else upper += counts[j];
    }
    const int64_t own_first = s->own_base + h.own_skip, own_end = s->own_base + h.num_nodes;
    if (lower + 1 > own_first || own_end + upper + 1 > s->node_capacity) {
        nbmi::set_error("nbmi_owner_step: received trees do not fit (%lld rows in front of / %lld behind the own %lld of %lld rows)",
                        (long long)lower, (long long)upper, (long long)h.num_nodes, (long long)s->node_capacity);
        return NBMI_ERR_CAPACITY;
    }
    const int64_t walk_first = own_first - lower;
    int64_t at_lo = walk_first, at_hi = own_end;
    for (int j = 0; j < kMaxWorld; j++) { P.src[j] = nullptr; P.count[j] = 0; P.base[j] = 0; }
    for (int j = 0; j < W; j++) {
        if (j == me) { P.base[j] = own_first; continue; }
        const int64_t c = counts[j] > 0 ? counts[j] : 0;
        P.src[j] = (const char *)dev_recv + seen * kLetRow;
        P.count[j] = c;
        if (j < me) { P.base[j] = at_lo; at_lo += c; } else { P.base[j] = at_hi; at_hi += c; }
        seen += c;
    }
    P.total = at_hi;
    P.me = me;
    NodeD *nd = let_nodesd(s);
    for (int j = 0; j < W; j++) {
        if (j == me || P.count[j] == 0) continue;
        k_let_append<<<nblocks(P.count[j]), kBlock, 0, st>>>(P, j, s->nodes, s->nodes64, nd, s->info);
    }
    k_let_finish<<<1, 64, 0, st>>>(P, s->own_links, s->chains, s->own_base, walk_first, s->nodes, nd, s->info);
    NBMI_HIP_CHECK(hipGetLastError());
    if (int rc = enqueue_walk(s, true, dt, nullptr)) return rc;
    s->curbuf ^= 1;
    s->tree_valid = false;
    s->maxabs_fused = s->fuse_maxabs;  // the walk left max |coordinate| of the rows it wrote (all of this rank's live bodies)
    return 0;
}

// ---- frame codec ----------------------------------------------------------------
namespace {
int frame_current(nbmi_sim *s, float **d_pos) {  // float32 positions in the caller's order, in `stage`
    const int64_t n = s->n;
    Bodies cur = s->buf[s->curbuf];
    float *dp = (float *)s->stage;
    k_unperm3_f32<<<nblocks(n), kBlock, 0, s->stream>>>(cur.x, cur.y, cur.z, s->owner ? nullptr : cur.id, n, dp);
    NBMI_HIP_CHECK(hipGetLastError());
    *d_pos = dp;
    return 0;
}
int frame_alloc(nbmi_sim *s) {
    if (s->frame_prev) return 0;
    const int64_t c = s->cap > s->n ? s->cap : s->n;
    if (dev_alloc(s, &s->frame_prev, (size_t)6 * (c ? c : 1)) || dev_alloc(s, &s->frame_q, (size_t)6 * (c ? c : 1))) return NBMI_ERR_HIP;
    return 0;
}
}  // namespace

int nbmi_frame_keyframe(nbmi_sim *s, float *out_pos, float *out_col) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!out_pos || !out_col) { nbmi::set_error("nbmi_frame_keyframe: null output"); return NBMI_ERR_ARG; }
    if (int rc = frame_alloc(s)) return rc;
    float *dp = nullptr;
    if (int rc = frame_current(s, &dp)) return rc;
    hipStream_t st = s->stream;
    NBMI_HIP_CHECK(hipMemcpyAsync(s->frame_prev, dp, (size_t)n * 12, hipMemcpyDeviceToDevice, st));
    NBMI_HIP_CHECK(hipMemcpyAsync(s->frame_prev + 3 * n, s->colors, (size_t)n * 12, hipMemcpyDeviceToDevice, st));
    NBMI_HIP_CHECK(hipMemcpyAsync(out_pos, dp, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipMemcpyAsync(out_col, s->colors, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipStreamSynchronize(st));
    s->frame_have_prev = true;
    if (s->method == NBMI_METHOD_BARNES_HUT) return check_device_error(s);
    return 0;
}

int nbmi_frame_delta_i16(nbmi_sim *s, int16_t *out_dpos, int16_t *out_dcol) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!out_dpos || !out_dcol) { nbmi::set_error("nbmi_frame_delta_i16: null output"); return NBMI_ERR_ARG; }
    if (!s->frame_have_prev) {
        nbmi::set_error("nbmi_frame_delta_i16: no previous frame on the device (call nbmi_frame_keyframe or "
                        "nbmi_frame_set_previous first)");
        return NBMI_ERR_ARG;
    }
    float *dp = nullptr;
    if (int rc = frame_current(s, &dp)) return rc;
    hipStream_t st = s->stream;
    k_frame_delta<<<nblocks(3 * n), kBlock, 0, st>>>(dp, s->frame_prev, 3 * n, s->frame_q);
    k_frame_delta<<<nblocks(3 * n), kBlock, 0, st>>>(s->colors, s->frame_prev + 3 * n, 3 * n, s->frame_q + 3 * n);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemcpyAsync(out_dpos, s->frame_q, (size_t)n * 6, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipMemcpyAsync(out_dcol, s->frame_q + 3 * n, (size_t)n * 6, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipStreamSynchronize(st));
    if (s->method == NBMI_METHOD_BARNES_HUT) return check_device_error(s);
    return 0;
}

int nbmi_frame_set_previous(nbmi_sim *s, const float *pos, const float *col) {
    if (int rc = check_handle(s)) return rc;
    const int64_t n = s->n;
    if (n == 0) return 0;
    if (!pos || !col) { nbmi::set_error("nbmi_frame_set_previous: null input"); return NBMI_ERR_ARG; }
    if (int rc = frame_alloc(s)) return rc;
    NBMI_HIP_CHECK(hipMemcpyAsync(s->frame_prev, pos, (size_t)n * 12, hipMemcpyHostToDevice, s->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(s->frame_prev + 3 * n, col, (size_t)n * 12, hipMemcpyHostToDevice, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->frame_have_prev = true;
    return 0;
}

namespace {
struct EmitPoints {
    const double *x, *y, *z;
    const float *colors;  // (N,3), original order
    float *out_pos, *out_col;
    __device__ void operator()(int64_t k, int64_t i, uint32_t slot) const {
        out_pos[3 * k] = (float)x[slot];
        out_pos[3 * k + 1] = (float)y[slot];
        out_pos[3 * k + 2] = (float)z[slot];
        out_col[3 * k] = colors[3 * i];
        out_col[3 * k + 1] = colors[3 * i + 1];
        out_col[3 * k + 2] = colors[3 * i + 2];
    }
};
}  // namespace

int nbmi_visible_points(nbmi_sim *s, const double *cam12, double tan_h, double tan_v, double far_dist, float *out_pos,
                        float *out_col, int64_t capacity, int64_t *count) {
    if (int rc = check_handle(s)) return rc;
    if (s->owner) { nbmi::set_error("nbmi_visible_points: not available on an owner-mode handle"); return NBMI_ERR_ARG; }
    if (!cam12 || !count || capacity < 0 || (capacity > 0 && (!out_pos || !out_col))) {
        nbmi::set_error("nbmi_visible_points: null argument");
        return NBMI_ERR_ARG;
    }
    const int64_t n = s->n;
    *count = 0;
    if (n == 0) return 0;
    const int64_t ntiles = vis::tiles_for(n);
    if (!s->vis_flag) {
        if (dev_alloc(s, &s->vis_flag, vis::flag_bytes(n)) || dev_alloc(s, &s->vis_slot, (size_t)ntiles * vis::kTile) ||
            dev_alloc(s, &s->vis_tiles, ntiles + 1))
            return NBMI_ERR_HIP;
        NBMI_HIP_CHECK(hipMemsetAsync(s->vis_flag, 0, vis::flag_bytes(n), s->stream));
    }
    vis::Camera c;
    for (int k = 0; k < 3; k++) { c.p[k] = cam12[k]; c.f[k] = cam12[3 + k]; c.r[k] = cam12[6 + k]; c.u[k] = cam12[9 + k]; }
    c.tan_h = tan_h; c.tan_v = tan_v; c.z_near = 0.1; c.z_far = far_dist; c.margin = 1.2;
    Bodies cur = s->buf[s->curbuf];
    hipStream_t st = s->stream;
    float *d_pos = (float *)s->stage, *d_col = d_pos + 3 * n;  // stage holds 7 N doubles
    vis::k_mark<<<nblocks(n), kBlock, 0, st>>>(cur.x, cur.y, cur.z, cur.id, n, c, s->vis_flag, s->vis_slot);
    vis::k_count<<<(int)ntiles, vis::kBlock, 0, st>>>(s->vis_flag, n, s->vis_tiles);
    vis::k_scan_tiles<<<1, vis::kBlock, 0, st>>>(s->vis_tiles, ntiles);
    EmitPoints e{cur.x, cur.y, cur.z, s->colors, d_pos, d_col};
    vis::k_emit<<<(int)ntiles, vis::kBlock, 0, st>>>(s->vis_flag, s->vis_slot, s->vis_tiles, n, n, e);
    NBMI_HIP_CHECK(hipGetLastError());
    uint32_t total = 0;
    NBMI_HIP_CHECK(hipMemcpyAsync(&total, s->vis_tiles + ntiles, 4, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipStreamSynchronize(st));
    *count = total;
    const int64_t rows = (int64_t)total < capacity ? (int64_t)total : capacity;
    if (rows > 0) {
        NBMI_HIP_CHECK(hipMemcpyAsync(out_pos, d_pos, (size_t)rows * 12, hipMemcpyDeviceToHost, st));
        NBMI_HIP_CHECK(hipMemcpyAsync(out_col, d_col, (size_t)rows * 12, hipMemcpyDeviceToHost, st));
        NBMI_HIP_CHECK(hipStreamSynchronize(st));
    }
    return 0;
}

int nbmi_set_force_precision(nbmi_sim *s, int mode, double tau) {
    if (int rc = check_handle(s)) return rc;
    if (mode < 0 || mode > 2 || (mode == 0 && !(tau >= 0.0))) {
        nbmi::set_error("nbmi_set_force_precision: mode must be 0 (auto), 1 (fp32) or 2 (float64), tau >= 0");
        return NBMI_ERR_ARG;
    }
    if (s->method != NBMI_METHOD_BARNES_HUT) { nbmi::set_error("not a Barnes-Hut handle"); return NBMI_ERR_ARG; }
    if (mode != 1 && !s->nodesd) {
        if (!(s->softening > 1e-12) || s->node_capacity + 2 > kMaxNodeDRows) {
            nbmi::set_error("nbmi_set_force_precision: float64 forces need softening > 0 and at most %lld node rows",
                            (long long)kMaxNodeDRows);
            return NBMI_ERR_ARG;
        }
        if (dev_alloc(s, &s->nodesd, s->node_capacity + 2)) return NBMI_ERR_HIP;
        if (int rc = upload_walk_table(s)) return rc;
        s->tree_valid = false;
    }
    s->force_prec = mode;
    if (mode == 0 && tau > 0.0) s->prec_tau = tau;
    return 0;
}

int nbmi_force_precision_share(nbmi_sim *s, double *share, int *all64) {
    if (int rc = check_handle(s)) return rc;
    if (!share || !all64) { nbmi::set_error("nbmi_force_precision_share: null output"); return NBMI_ERR_ARG; }
    *share = 0.0;
    *all64 = 0;
    if (s->method != NBMI_METHOD_BARNES_HUT || !s->nodesd || s->n == 0) return 0;
    if (s->force_prec == 2) { *share = 1.0; *all64 = 1; return 0; }
    if (s->force_prec == 1) return 0;
    const int64_t nw = (s->n + 63) / 64;
    std::vector<unsigned char> h((size_t)nw);
    TreeInfo ti;
    NBMI_HIP_CHECK(hipMemcpyAsync(h.data(), s->wave_flag, (size_t)nw, hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(&ti, s->info, sizeof(ti), hipMemcpyDeviceToHost, s->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(s->stream));
    int64_t c = 0;
    for (int64_t i = 0; i < nw; i++) c += h[i] ? 1 : 0;
    *share = (double)c / (double)nw;
    *all64 = ti.force_all64;
    return 0;
}

int nbmi_set_exchange_sync(nbmi_sim *s, int sync) {
    if (int rc = check_handle(s)) return rc;
    s->exchange_sync = sync != 0;
    return 0;
}

void *nbmi_stream(nbmi_sim *s) { return s ? (void *)s->stream : nullptr; }

}  // extern "C"
