// Boids fixed-radius neighbour sweep on MI355X (gfx950).  C ABI in include/bdmi.h.
//
// One bdmi_step == one Flock.update(dt) of the reference (boids/flock.py:627-678):
//   assign_cells (:30-44) -> argsort + build_cell_lists (:610-625, :47-65) ->
//   compute_flocking_spatial (:68-238) -> update_physics_numba (:241-308).
// float64 state and arithmetic like the reference (this path is bandwidth bound; MI355X
// float64 vector rate is far above what ~7 candidates per boid need).
//
// HBM layout: master state SoA float64 {p, v, c} x 3 + int32 id, kept in CELL-SORTED order
// between steps (buffer A).  Per step: cell keys from A -> radix sort (cell, rank) -> gather A into
// the read-side copy B (array of 32-byte {x,y,z,-} records per quantity: one cache line per
// neighbour position) -> 1-bit-per-cell occupancy map + {start,end} pair per non-empty cell ->
// sweep kernel tests the occupancy bit (L2 resident) before touching the pair table, reads
// neighbours from B and writes the updated boid back to A at the same rank, physics fused.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/bdmi.h"
#include "common.h"
#include "visible.h"

namespace {

constexpr int kBlock = 256;

struct Boids {
    double *px, *py, *pz, *vx, *vy, *vz, *cr, *cg, *cb;
    int32_t *id;
};

// read-side copy of the boids for the sweep: array of structures, one 32-byte record per boid and
// quantity, so that a neighbour's position is ONE cache line (the SoA master touches three)
struct BoidsAoS {
    double4 *p, *v, *c;  // {x,y,z,-}, {vx,vy,vz,-}, {r,g,b,-}
    int32_t *id;
};

struct GridP {
    double cell_size, offset;
    int dim, range;
};

struct FlockP {
    double perception_sq, separation_sq, sep_w, ali_w, coh_w, max_speed, max_force;
    double bounds, margin, wall_force, blend, dt;
};

// get_cell_index (flock.py:16-27): int() truncation toward zero, then clamp
__device__ __forceinline__ int cell_coord(double v, const GridP &g) {
    int c = (int)((v + g.offset) / g.cell_size);
    c = c < 0 ? 0 : c;
    c = c > g.dim - 1 ? g.dim - 1 : c;
    return c;
}

__global__ __launch_bounds__(kBlock) void k_assign(Boids a, int64_t n, GridP g, uint32_t *__restrict__ keys,
                                                   uint32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_coord(a.px[i], g), cy = cell_coord(a.py[i], g), cz = cell_coord(a.pz[i], g);
    keys[i] = (uint32_t)(cx + cy * g.dim + cz * g.dim * g.dim);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void k_reorder(Boids a, BoidsAoS b, const uint32_t *__restrict__ perm, int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const uint32_t j = perm[r];
    b.p[r] = make_double4(a.px[j], a.py[j], a.pz[j], 0.0);
    b.v[r] = make_double4(a.vx[j], a.vy[j], a.vz[j], 0.0);
    b.c[r] = make_double4(a.cr[j], a.cg[j], a.cb[j], 0.0);
    b.id[r] = a.id[j];
}

// build_cell_lists (flock.py:47-65) from sorted keys, as a COMPACT table: the non-empty cells are
// numbered in cell order (rank k), cell_start[k] = first sorted boid of the k-th non-empty cell
// (its end is cell_start[k + 1]); occ[w] = {bits, prefix}: one bit per cell of the 32 cells of word w
// (non-empty or not) and the rank of the word's first non-empty cell, so that one 8-byte load gives
//     rank(cell) = occ[cell >> 5].prefix + popcount(occ[cell >> 5].bits & ((1 << (cell & 31)) - 1)).
// 2 MB + 4 B per non-empty cell, all cache resident; the table indexed by cell (66 MB at the
// reference grid, one random 128-byte line per lookup) was where the sweep's HBM traffic came from.
// Two passes over the sorted keys (count "first boid of a cell" per tile, scan, emit).
__device__ __forceinline__ bool first_of_cell(const uint32_t *keys_s, int64_t r, int64_t n) {
    return r < n && (r == 0 || keys_s[r - 1] != keys_s[r]);
}

__global__ __launch_bounds__(kBlock) void k_first_count(const uint32_t *__restrict__ keys_s, int64_t n,
                                                        uint32_t *__restrict__ tile_cnt) {
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * vis::kItems;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) c += first_of_cell(keys_s, base + k, n) ? 1u : 0u;
    unsigned total;
    (void)vis::block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void k_table(const uint32_t *__restrict__ keys_s, int64_t n,
                                                  const uint32_t *__restrict__ tile_cnt, int64_t ntiles,
                                                  uint2 *__restrict__ occ, int32_t *__restrict__ cell_start) {
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * vis::kItems;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) m |= (first_of_cell(keys_s, base + k, n) ? 1u : 0u) << k;
    unsigned total;
    unsigned rank = tile_cnt[blockIdx.x] + vis::block_exclusive_scan(__popc(m), &total);
    // a thread's 8 consecutive boids usually fall into one or two occupancy words: OR their bits
    // locally and issue one atomic per word
    uint32_t pend_word = 0xffffffffu, pend_bits = 0u;
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) {
        if (!((m >> k) & 1u)) continue;
        const int64_t r = base + k;
        const uint32_t c = keys_s[r];
        cell_start[rank] = (int32_t)r;
        if ((c >> 5) != pend_word) {
            if (pend_bits) atomicOr(&occ[pend_word].x, pend_bits);
            pend_word = c >> 5;
            pend_bits = 0u;
        }
        pend_bits |= 1u << (c & 31);
        if (r == 0 || (keys_s[r - 1] >> 5) != (c >> 5)) occ[c >> 5].y = rank;
        rank++;
    }
    if (pend_bits) atomicOr(&occ[pend_word].x, pend_bits);
    if (blockIdx.x == 0 && threadIdx.x == 0) cell_start[tile_cnt[ntiles]] = (int32_t)n;  // end of the last cell
}

// number of non-empty cells of the last grid; on demand only (bdmi_grid_info), one atomic per block
__global__ __launch_bounds__(kBlock) void k_count_cells(const uint32_t *__restrict__ keys_s, int64_t n,
                                                        unsigned long long *occupied) {
    __shared__ unsigned red[kBlock / 64];
    unsigned cnt = 0;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock)
        cnt += (r == 0 || keys_s[r - 1] != keys_s[r]) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) cnt += red[w];
        atomicAdd(occupied, (unsigned long long)cnt);
    }
}

// steer(): mean -> normalise * max_speed - v -> clamp to max_force -> * weight
// (the repeated block of flock.py:174-234); returns false if the magnitude is 0.
__device__ __forceinline__ bool steer(double &x, double &y, double &z, double vx, double vy, double vz,
                                      double max_speed, double max_force, double w) {
    double mag = sqrt(x * x + y * y + z * z);
    if (!(mag > 0)) return false;
    x = (x / mag) * max_speed - vx;
    y = (y / mag) * max_speed - vy;
    z = (z / mag) * max_speed - vz;
    mag = sqrt(x * x + y * y + z * z);
    if (mag > max_force) {
        x = (x / mag) * max_force;
        y = (y / mag) * max_force;
        z = (z / mag) * max_force;
    }
    x *= w; y *= w; z *= w;
    return true;
}

// XCD-aware block remap (same idea as the tree walk's): hardware deals consecutive blocks round-robin
// over the 8 XCDs, each with its own L2.  A boid's candidates sit up to ~grid_dim^2 * density ranks
// away in the cell-sorted order (the cz +- 1 planes), so each XCD gets one CONTIGUOUS eighth of the
// blocks and finds them in its own L2 instead of re-fetching them from HBM.  xcd_contiguous = 0
// keeps the hardware order (measurement knob BDMI_XCD=0).
__device__ __forceinline__ int64_t logical_block(int b, int nb, int xcd_contiguous) {
    if (!xcd_contiguous) return b;
    const int xcd = b & 7, j = b >> 3;
    const int q = nb >> 3, rem = nb & 7;
    return (int64_t)xcd * q + (xcd < rem ? xcd : rem) + j;
}

// compute_flocking_spatial (flock.py:68-238) + update_physics_numba (flock.py:241-308).
// kPhysics=false: write the four force arrays (caller's boid order) instead of integrating.
template <bool kPhysics>
__global__ __launch_bounds__(kBlock) void k_flock(BoidsAoS b, Boids a, const uint2 *__restrict__ occ,
                                                  const int32_t *__restrict__ cell_start, int64_t n, GridP g, FlockP P,
                                                  double *__restrict__ o_sep, double *__restrict__ o_ali,
                                                  double *__restrict__ o_coh, double *__restrict__ o_avg,
                                                  int xcd_contiguous) {
    __shared__ int32_t runs[18][kBlock];  // [2 row, 2 row + 1][thread]: candidate run of the thread's boid in each of its nine rows
    constexpr int kHitCap = 32;
    __shared__ unsigned short hits[kHitCap][kBlock];  // [k][thread]: the boid's neighbours found so far, (row << 12) | offset in the row's run
    const int64_t r = logical_block(blockIdx.x, gridDim.x, xcd_contiguous) * kBlock + threadIdx.x;
    if (r >= n) return;
    const double4 pi4 = b.p[r], vi4 = b.v[r], ci4 = b.c[r];
    if (kPhysics && b.id[r] < 0) {
        // a ghost (slab mode: a neighbour slab's boid inside the halo): visible to the others, owned elsewhere
        a.px[r] = pi4.x; a.py[r] = pi4.y; a.pz[r] = pi4.z;
        a.vx[r] = vi4.x; a.vy[r] = vi4.y; a.vz[r] = vi4.z;
        a.cr[r] = ci4.x; a.cg[r] = ci4.y; a.cb[r] = ci4.z;
        a.id[r] = b.id[r];
        return;
    }
    const double pix = pi4.x, piy = pi4.y, piz = pi4.z;
    const double vix = vi4.x, viy = vi4.y, viz = vi4.z;
    const double cir = ci4.x, cig = ci4.y, cib = ci4.z;
    const int cx = cell_coord(pix, g), cy = cell_coord(piy, g), cz = cell_coord(piz, g);
    double sx = 0, sy = 0, sz = 0, alx = 0, aly = 0, alz = 0, cox = 0, coy = 0, coz = 0, clr = 0, clg = 0, clb = 0;
    int sep_count = 0, nb_count = 0;
    // The reference visits the (2 range + 1)^3 neighbour cells one by one (flock.py:117-132).  Cells that
    // differ only in x have consecutive indices, and boids are sorted by cell index, so the candidates
    // of one (y, z) row of cells are ONE contiguous run of sorted boids: (2 range + 1)^2 runs instead of
    // (2 range + 1)^3 cell visits, found with two rank lookups in the compact table.  Same candidate
    // set as the reference; the floating-point sums run in a different order (the reference's own
    // order inside a cell is unspecified anyway: np.argsort, flock.py:618).
    const int x_lo = cx - g.range < 0 ? 0 : cx - g.range;
    const int x_hi = cx + g.range > g.dim - 1 ? g.dim - 1 : cx + g.range;
    // one candidate: the reference's test and sums (flock.py:134-172)
    auto candidate = [&](int32_t q) {
        if (q == r) return;
        const double4 qp = b.p[q];
        const double dx = pix - qp.x, dy = piy - qp.y, dz = piz - qp.z;
        const double dist_sq = dx * dx + dy * dy + dz * dz;
        if (dist_sq < P.perception_sq && dist_sq > 0.0001) {
            const double dist = sqrt(dist_sq);
            if (dist_sq < P.separation_sq) {
                const double inv_dist = 1.0 / dist;
                sx += dx * inv_dist / dist;
                sy += dy * inv_dist / dist;
                sz += dz * inv_dist / dist;
                sep_count++;
            }
            const double4 qv = b.v[q], qc = b.c[q];
            alx += qv.x; aly += qv.y; alz += qv.z;
            cox += qp.x; coy += qp.y; coz += qp.z;
            clr += qc.x; clg += qc.y; clb += qc.z;
            nb_count++;
        }
    };
    if (g.range == 1) {
        // The usual grid (cell = perception radius, flock.py:478-481).  A row of cells is 3 cells = at most two
        // words of the occupancy table, and what a row costs is a chain of dependent loads (table word -> run
        // bounds -> candidates), not arithmetic: the three rows of a z plane go through each stage TOGETHER, so
        // that their loads are in flight at the same time.  Same rows, same order, same sums as the loop below.
        bool big = false;
        int nr = 0;  // non-empty runs noted so far
        for (int dcz = -1; dcz <= 1; dcz++) {
            const int ncz = cz + dcz;
            const bool zok = ncz >= 0 && ncz < g.dim;
            uint2 wa[3], wb[3];
            int lo_bit[3], hi_bit[3];
            bool two[3], rok[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int ncy = cy + j - 1;
                rok[j] = zok && ncy >= 0 && ncy < g.dim;
                const int64_t row = (int64_t)ncy * g.dim + (int64_t)ncz * g.dim * g.dim;
                const int64_t c_lo = row + x_lo, c_hi = row + x_hi;
                lo_bit[j] = (int)(c_lo & 31);
                hi_bit[j] = (int)(c_hi & 31);
                two[j] = (c_lo >> 5) != (c_hi >> 5);
                wa[j] = rok[j] ? occ[c_lo >> 5] : make_uint2(0u, 0u);
                wb[j] = (rok[j] && two[j]) ? occ[c_hi >> 5] : make_uint2(0u, 0u);
            }
            int32_t qb[3], qe[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                uint32_t ba = wa[j].x & (~0u << lo_bit[j]);
                uint32_t bb = wb[j].x & (~0u >> (31 - hi_bit[j]));
                if (!two[j]) { ba &= ~0u >> (31 - hi_bit[j]); bb = 0u; }
                qb[j] = 0; qe[j] = 0;
                if (rok[j] && (ba | bb)) {
                    // first and last non-empty cell of the row
                    const uint2 fw = ba ? wa[j] : wb[j];
                    const int first = ba ? __ffs(ba) - 1 : __ffs(bb) - 1;
                    const uint2 lw = bb ? wb[j] : wa[j];
                    const int last = bb ? 31 - __clz(bb) : 31 - __clz(ba);
                    const uint32_t k_lo = fw.y + __popc(fw.x & ((1u << first) - 1u));
                    const uint32_t k_hi = lw.y + __popc(lw.x & ((1u << last) - 1u)) + 1u;
                    qb[j] = cell_start[k_lo];
                    qe[j] = cell_start[k_hi];
                }
            }
            // [r4] the runs are only noted here; ONE loop behind the three planes walks all nine (below)
            // (only the NON-EMPTY runs are noted, in row order: the loop below then never has to step over an empty one)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (qe[j] > qb[j]) {
                    runs[2 * nr][threadIdx.x] = qb[j];
                    runs[2 * nr + 1][threadIdx.x] = qe[j];
                    nr++;
                }
                big = big || qe[j] - qb[j] > 4095;
            }
        }
        // [r4] One flattened candidate loop per boid instead of nine, in two phases.  In the state flocks reach (53
        // candidates per boid instead of 9, two thirds of the cells empty, the rest holding more) the nine per-row
        // loops cost a wave the SUM of the rows' longest runs among its 64 boids; one loop over a lane's nine runs back
        // to back costs the longest TOTAL.  And only one candidate in six is a neighbour (a sphere of one cell size in
        // a cube of three), yet nearly every trip of the loop had SOME lane with one, so the wave ran the expensive
        // half of the body - a float64 square root, four divisions, two dependent 32-byte loads - on every trip with
        // a sixth of its lanes (66 % of the sweep's wave cycles were waits, 85 vector instructions per trip:
        // profiles/r04_boids_steady_state_ab.txt).  Now phase 1 only tests distances, four candidates per trip
        // with their position records requested together, and notes the neighbours (16 bits each: row and offset in
        // the row's run) in a per-lane list in LDS; phase 2 walks that list - every lane busy with a real neighbour,
        // two at a time with all six records in flight.  Same neighbours in the same order: results unchanged bit
        // for bit.  Sweep at 2 M boids: 0.275 -> 0.205 ms on the initial state, 0.783 -> 0.579 ms after 1000 steps.
        // (Measured and dropped: phase 1 on an fp32 copy of the positions relative to the cell centres, eight per trip,
        // float64 re-test inside the fp32 error band - same sets, 0.67 ms: the sweep waits for its gathers, it is not
        // short of arithmetic or bytes.)
        int nh = 0;
        auto neighbour = [&](const double4 qp, const double4 qv, const double4 qc) {  // flock.py:150-172
            const double dx = pix - qp.x, dy = piy - qp.y, dz = piz - qp.z;
            const double dist_sq = dx * dx + dy * dy + dz * dz;
            if (dist_sq < P.separation_sq) {
                const double dist = sqrt(dist_sq);
                const double inv_dist = 1.0 / dist;
                sx += dx * inv_dist / dist;
                sy += dy * inv_dist / dist;
                sz += dz * inv_dist / dist;
                sep_count++;
            }
            alx += qv.x; aly += qv.y; alz += qv.z;
            cox += qp.x; coy += qp.y; coz += qp.z;
            clr += qc.x; clg += qc.y; clb += qc.z;
            nb_count++;
        };
        auto flush = [&]() {
            for (int h = 0; h < nh; h += 2) {
                const unsigned e0 = hits[h][threadIdx.x], e1 = hits[h + 1 < nh ? h + 1 : h][threadIdx.x];
                const int32_t q0 = runs[2 * (e0 >> 12)][threadIdx.x] + (int32_t)(e0 & 4095u);
                const int32_t q1 = runs[2 * (e1 >> 12)][threadIdx.x] + (int32_t)(e1 & 4095u);
                const double4 p0 = b.p[q0], v0 = b.v[q0], c0 = b.c[q0], p1 = b.p[q1], v1 = b.v[q1], c1 = b.c[q1];
                neighbour(p0, v0, c0);
                if (h + 1 < nh) neighbour(p1, v1, c1);
            }
            nh = 0;
        };
        int row = 0;
        int32_t q = 0, e = 0, q_first = 0;
        if (nr > 0) { q = q_first = runs[0][threadIdx.x]; e = runs[1][threadIdx.x]; }
        auto next = [&](unsigned &code) -> int32_t {
            if (q >= e) {  // the run is used up: on to the next noted one (none of them is empty)
                if (row + 1 >= nr) { row = nr; return -1; }
                row++;
                q = q_first = runs[2 * row][threadIdx.x];
                e = runs[2 * row + 1][threadIdx.x];
            }
            code = ((unsigned)row << 12) | (unsigned)(q - q_first);
            return q++;
        };
        auto test = [&](int32_t c, unsigned code, const double4 qp) {
            if (c < 0 || c == r) return;
            const double dx = pix - qp.x, dy = piy - qp.y, dz = piz - qp.z;
            const double dist_sq = dx * dx + dy * dy + dz * dz;
            if (dist_sq < P.perception_sq && dist_sq > 0.0001) hits[nh++][threadIdx.x] = (unsigned short)code;
        };
        if (big) {
            // a run of more than 4095 boids (three cells!): its offsets do not fit the 16-bit notes - such a boid
            // takes its candidates one by one, as the loop for other grids below does
            for (int k = 0; k < nr; k++)
                for (int32_t c = runs[2 * k][threadIdx.x]; c < runs[2 * k + 1][threadIdx.x]; c++) candidate(c);
        } else
        for (;;) {
            if (nh > kHitCap - 4) flush();
            unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
            const int32_t c0 = next(k0);
            if (c0 < 0) break;
            const int32_t c1 = next(k1), c2 = next(k2), c3 = next(k3);
            const double4 p0 = b.p[c0], p1 = b.p[c1 < 0 ? c0 : c1], p2 = b.p[c2 < 0 ? c0 : c2], p3 = b.p[c3 < 0 ? c0 : c3];
            test(c0, k0, p0);
            test(c1, k1, p1);
            test(c2, k2, p2);
            test(c3, k3, p3);
        }
        flush();
    } else {
    for (int dcz = -g.range; dcz <= g.range; dcz++) {
        const int ncz = cz + dcz;
        if (ncz < 0 || ncz >= g.dim) continue;
        for (int dcy = -g.range; dcy <= g.range; dcy++) {
            const int ncy = cy + dcy;
            if (ncy < 0 || ncy >= g.dim) continue;
            const int64_t row = (int64_t)ncy * g.dim + (int64_t)ncz * g.dim * g.dim;
            const int64_t c_lo = row + x_lo, c_hi = row + x_hi;
            // first and last non-empty cell of [c_lo, c_hi]
            int first = -1, last = -1;  // bit positions inside their words
            uint2 first_w = make_uint2(0u, 0u), last_w = make_uint2(0u, 0u);
            for (int64_t w = c_lo >> 5; w <= (c_hi >> 5); w++) {
                const uint2 e = occ[w];
                uint32_t bits = e.x;
                if (w == (c_lo >> 5)) bits &= ~0u << (c_lo & 31);
                if (w == (c_hi >> 5)) bits &= ~0u >> (31 - (c_hi & 31));
                if (bits) {
                    if (first < 0) { first = __ffs(bits) - 1; first_w = e; }
                    last = 31 - __clz(bits);
                    last_w = e;
                }
            }
            if (first < 0) continue;  // the whole row is empty: no table read
            const uint32_t k_lo = first_w.y + __popc(first_w.x & ((1u << first) - 1u));
            const uint32_t k_hi = last_w.y + __popc(last_w.x & ((1u << last) - 1u)) + 1u;
            const int32_t q_begin = cell_start[k_lo], q_end = cell_start[k_hi];
            for (int32_t q = q_begin; q < q_end; q++) candidate(q);
        }
    }
    }
    double fsx = 0, fsy = 0, fsz = 0, fax = 0, fay = 0, faz = 0, fcx = 0, fcy = 0, fcz = 0;
    double avr = cir, avg = cig, avb = cib;  // caller pre-fill: avg_colors <- colors (flock.py:636)
    if (sep_count > 0) {
        sx /= sep_count; sy /= sep_count; sz /= sep_count;
        if (steer(sx, sy, sz, vix, viy, viz, P.max_speed, P.max_force, P.sep_w)) { fsx = sx; fsy = sy; fsz = sz; }
    }
    if (nb_count > 0) {
        alx /= nb_count; aly /= nb_count; alz /= nb_count;
        if (steer(alx, aly, alz, vix, viy, viz, P.max_speed, P.max_force, P.ali_w)) { fax = alx; fay = aly; faz = alz; }
        cox = cox / nb_count - pix; coy = coy / nb_count - piy; coz = coz / nb_count - piz;
        if (steer(cox, coy, coz, vix, viy, viz, P.max_speed, P.max_force, P.coh_w)) { fcx = cox; fcy = coy; fcz = coz; }
        avr = (clr + cir) / (nb_count + 1);
        avg = (clg + cig) / (nb_count + 1);
        avb = (clb + cib) / (nb_count + 1);
    }
    if (!kPhysics) {
        const int64_t o = 3 * (int64_t)b.id[r];
        o_sep[o] = fsx; o_sep[o + 1] = fsy; o_sep[o + 2] = fsz;
        o_ali[o] = fax; o_ali[o + 1] = fay; o_ali[o + 2] = faz;
        o_coh[o] = fcx; o_coh[o + 1] = fcy; o_coh[o + 2] = fcz;
        o_avg[o] = avr; o_avg[o + 1] = avg; o_avg[o + 2] = avb;
        return;
    }
    double acc[3] = {fsx + fax + fcx, fsy + fay + fcy, fsz + faz + fcz};
    const double pos[3] = {pix, piy, piz};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const double dist_pos = pos[d] - (P.bounds - P.margin);
        if (dist_pos > 0) acc[d] -= fmin(dist_pos / P.margin * 2.0, 1.0) * P.wall_force;
        const double dist_neg = (-P.bounds + P.margin) - pos[d];
        if (dist_neg > 0) acc[d] += fmin(dist_neg / P.margin * 2.0, 1.0) * P.wall_force;
    }
    double nvx = vix + acc[0] * P.dt, nvy = viy + acc[1] * P.dt, nvz = viz + acc[2] * P.dt;
    const double speed = sqrt(nvx * nvx + nvy * nvy + nvz * nvz);
    if (speed > P.max_speed) {
        const double scale = P.max_speed / speed;
        nvx *= scale; nvy *= scale; nvz *= scale;
    }
    a.vx[r] = nvx; a.vy[r] = nvy; a.vz[r] = nvz;
    a.px[r] = pix + nvx * P.dt; a.py[r] = piy + nvy * P.dt; a.pz[r] = piz + nvz * P.dt;
    a.cr[r] = cir + (avr - cir) * P.blend;
    a.cg[r] = cig + (avg - cig) * P.blend;
    a.cb[r] = cib + (avb - cib) * P.blend;
    a.id[r] = b.id[r];
}

__global__ __launch_bounds__(kBlock) void k_split(const double *__restrict__ p, const double *__restrict__ v,
                                                  const double *__restrict__ c, Boids a, int64_t n, int by_id) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t i = by_id ? (int64_t)a.id[r] : r;
    if (p) { a.px[r] = p[3 * i]; a.py[r] = p[3 * i + 1]; a.pz[r] = p[3 * i + 2]; }
    if (v) { a.vx[r] = v[3 * i]; a.vy[r] = v[3 * i + 1]; a.vz[r] = v[3 * i + 2]; }
    if (c) { a.cr[r] = c[3 * i]; a.cg[r] = c[3 * i + 1]; a.cb[r] = c[3 * i + 2]; }
    if (!by_id) a.id[r] = (int32_t)r;
}

__global__ __launch_bounds__(kBlock) void k_join(const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, const int32_t *__restrict__ id, int64_t n,
                                                 double *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int64_t o = 3 * (int64_t)id[r];
    out[o] = x[r]; out[o + 1] = y[r]; out[o + 2] = z[r];
}

__global__ __launch_bounds__(kBlock) void k_cells_out(Boids a, int64_t n, GridP g, int32_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const int cx = cell_coord(a.px[r], g), cy = cell_coord(a.py[r], g), cz = cell_coord(a.pz[r], g);
    out[a.id[r]] = cx + cy * g.dim + cz * g.dim * g.dim;
}

// ---- slab mode (multi-GPU): selections of rows by a predicate, in row order ------------------------------
// what = 0: owned rows (id >= 0); 1: owned rows in the LEFT halo zone (x < x_lo + cell); 2: RIGHT (x >= x_hi - cell)
struct SlabSel {
    const double *px;
    const int32_t *id;
    int what;
    double lo_edge, hi_edge;
    __device__ bool operator()(int64_t r) const {
        if (id[r] < 0) return false;
        if (what == 0) return true;
        return what == 1 ? px[r] < lo_edge : px[r] >= hi_edge;
    }
};
__global__ __launch_bounds__(kBlock) void k_sel_count(SlabSel sel, int64_t n, uint32_t *__restrict__ tile_cnt) {
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * vis::kItems;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) c += (base + k < n && sel(base + k)) ? 1u : 0u;
    unsigned total;
    (void)vis::block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}
// selected rows -> packed rows {p, v, c, id} of 10 doubles (dst_rows) or -> the SoA arrays `dst` (compaction)
__global__ __launch_bounds__(kBlock) void k_sel_emit(SlabSel sel, Boids src, int64_t n, const uint32_t *__restrict__ tile_cnt,
                                                     Boids dst, double *__restrict__ dst_rows) {
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * vis::kItems;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) m |= ((base + k < n && sel(base + k)) ? 1u : 0u) << k;
    unsigned total;
    int64_t slot = (int64_t)tile_cnt[blockIdx.x] + vis::block_exclusive_scan(__popc(m), &total);
#pragma unroll
    for (int k = 0; k < vis::kItems; k++) {
        if (!((m >> k) & 1u)) continue;
        const int64_t r = base + k;
        if (dst_rows) {
            double *o = dst_rows + 10 * slot;
            o[0] = src.px[r]; o[1] = src.py[r]; o[2] = src.pz[r];
            o[3] = src.vx[r]; o[4] = src.vy[r]; o[5] = src.vz[r];
            o[6] = src.cr[r]; o[7] = src.cg[r]; o[8] = src.cb[r];
            o[9] = (double)src.id[r];
        } else {
            dst.px[slot] = src.px[r]; dst.py[slot] = src.py[r]; dst.pz[slot] = src.pz[r];
            dst.vx[slot] = src.vx[r]; dst.vy[slot] = src.vy[r]; dst.vz[slot] = src.vz[r];
            dst.cr[slot] = src.cr[r]; dst.cg[slot] = src.cg[r]; dst.cb[slot] = src.cb[r];
            dst.id[slot] = src.id[r];
        }
        slot++;
    }
}
// owned rows that have left the slab stay for this step as ghosts (their new owner got a copy)
__global__ __launch_bounds__(kBlock) void k_slab_demote(Boids a, int64_t n, double x_lo, double x_hi) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= n) return;
    const double x = a.px[r];
    if (a.id[r] >= 0 && !(x >= x_lo && x < x_hi)) a.id[r] = -1 - a.id[r];
}
// received rows: inside the slab -> owned, else a ghost
__global__ __launch_bounds__(kBlock) void k_slab_append(Boids a, int64_t at, const double *__restrict__ rows, int64_t count,
                                                        double x_lo, double x_hi) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const double *o = rows + 10 * k;
    const int64_t r = at + k;
    a.px[r] = o[0]; a.py[r] = o[1]; a.pz[r] = o[2];
    a.vx[r] = o[3]; a.vy[r] = o[4]; a.vz[r] = o[5];
    a.cr[r] = o[6]; a.cg[r] = o[7]; a.cb[r] = o[8];
    const int32_t gid = (int32_t)o[9];
    a.id[r] = (o[0] >= x_lo && o[0] < x_hi) ? gid : -1 - gid;
}
__global__ __launch_bounds__(kBlock) void k_set_ids(const int32_t *__restrict__ ids, int32_t *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = ids[i];
}

inline int nblocks(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

struct bdmi_flock {
    int64_t n = 0;
    int device = 0;
    double params[11] = {};
    GridP grid = {};
    int64_t num_cells = 0;
    int key_bits = 0;
    hipStream_t stream = nullptr;
    Boids A = {};
    BoidsAoS B = {};
    // slab mode (multi-GPU): rows [0, n) = owned boids and this step's ghosts; T = scratch for compactions
    bool slab = false;
    int64_t cap = 0;
    double x_lo = 0, x_hi = 0;
    int has_left = 0, has_right = 0;
    Boids T = {};
    uint32_t *keys = nullptr, *keys_s = nullptr, *idx = nullptr, *perm = nullptr;
    uint2 *occ = nullptr;          // per 32 cells: {occupancy bits, rank of the first non-empty cell}
    int32_t *cell_start = nullptr;    // first sorted boid of the k-th non-empty cell; [count] = n
    uint32_t *tile_cnt = nullptr;     // scan scratch
    int64_t occ_words = 0;
    unsigned long long *occupied = nullptr;
    void *tmp_sort = nullptr;
    size_t tmp_sort_bytes = 0;
    double *stage = nullptr;  // 12 N doubles
    int xcd_contiguous = 1;   // sweep block -> XCD mapping, see logical_block()
    // render-side reduction scratch (bdmi_visible_vertices), allocated on first use
    uint8_t *vis_flag = nullptr;
    uint32_t *vis_slot = nullptr, *vis_tiles = nullptr;
    float *vis_verts = nullptr, *vis_cols = nullptr;  // 18 floats per boid each
    bool timers = false;
    hipEvent_t ev[4] = {};
    double ms[3] = {0, 0, 0};
    int64_t timed = 0;
    std::vector<void *> allocs;
};

namespace {

template <typename T>
int dev_alloc(bdmi_flock *f, T **p, size_t count) {
    void *q = nullptr;
    NBMI_HIP_CHECK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    f->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

int alloc_boids(bdmi_flock *f, Boids *b, int64_t n) {
    if (dev_alloc(f, &b->px, n) || dev_alloc(f, &b->py, n) || dev_alloc(f, &b->pz, n) || dev_alloc(f, &b->vx, n) ||
        dev_alloc(f, &b->vy, n) || dev_alloc(f, &b->vz, n) || dev_alloc(f, &b->cr, n) || dev_alloc(f, &b->cg, n) ||
        dev_alloc(f, &b->cb, n) || dev_alloc(f, &b->id, n))
        return -2;
    return 0;
}

int check(bdmi_flock *f) {
    if (!f) { nbmi::set_error("null bdmi_flock handle"); return -1; }
    if (hipSetDevice(f->device) != hipSuccess) { nbmi::set_error("hipSetDevice(%d) failed", f->device); return -2; }
    return 0;
}

FlockP make_params(const bdmi_flock *f, double dt) {
    const double *p = f->params;
    FlockP P;
    P.bounds = p[0]; P.margin = p[1];
    P.max_speed = p[3]; P.max_force = p[4];
    P.wall_force = p[4] * p[2];  // max_force * wall_weight (flock.py:673)
    P.perception_sq = p[5] * p[5];
    P.separation_sq = p[6] * p[6];
    P.sep_w = p[7]; P.ali_w = p[8]; P.coh_w = p[9];
    const double blend = p[10] * dt;  // min(1, color_blend_rate * dt) (flock.py:662)
    P.blend = blend < 1.0 ? blend : 1.0;
    P.dt = dt;
    return P;
}

// cells -> sort -> reorder A->B -> table
int enqueue_grid(bdmi_flock *f, bool timed) {
    const int64_t n = f->n;
    hipStream_t st = f->stream;
    if (timed) NBMI_HIP_CHECK(hipEventRecord(f->ev[0], st));
    k_assign<<<nblocks(n), kBlock, 0, st>>>(f->A, n, f->grid, f->keys, f->idx);
    NBMI_HIP_CHECK(nbmi::sort_pairs_u32_u32(f->tmp_sort, f->tmp_sort_bytes, f->keys, f->keys_s, f->idx, f->perm,
                                            (size_t)n, 0, f->key_bits, st));
    if (timed) NBMI_HIP_CHECK(hipEventRecord(f->ev[1], st));
    k_reorder<<<nblocks(n), kBlock, 0, st>>>(f->A, f->B, f->perm, n);
    NBMI_HIP_CHECK(hipMemsetAsync(f->occ, 0, (size_t)f->occ_words * sizeof(uint2), st));
    {
        const int64_t ntiles = vis::tiles_for(n);
        k_first_count<<<(int)ntiles, kBlock, 0, st>>>(f->keys_s, n, f->tile_cnt);
        vis::k_scan_tiles<<<1, vis::kBlock, 0, st>>>(f->tile_cnt, ntiles);
        k_table<<<(int)ntiles, kBlock, 0, st>>>(f->keys_s, n, f->tile_cnt, ntiles, f->occ, f->cell_start);
    }
    if (timed) NBMI_HIP_CHECK(hipEventRecord(f->ev[2], st));
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char *bdmi_last_error(void) { return nbmi::get_error(); }

void bdmi_destroy(bdmi_flock *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    for (void *p : f->allocs) (void)hipFree(p);
    for (auto &e : f->ev)
        if (e) (void)hipEventDestroy(e);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

static int bd_create_impl(bdmi_flock *f, const double *pos, const double *vel, const double *col) {
    const int64_t n = f->n;
    const int64_t c = f->cap > n ? f->cap : n;  // rows allocated (slab mode keeps head room for ghosts / immigrants)
    NBMI_HIP_CHECK(hipSetDevice(f->device));
    NBMI_HIP_CHECK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    for (auto &e : f->ev) NBMI_HIP_CHECK(hipEventCreate(&e));
    f->occ_words = (f->num_cells + 31) / 32;
    if (alloc_boids(f, &f->A, c)) return -2;
    if (f->slab && alloc_boids(f, &f->T, c)) return -2;
    if (dev_alloc(f, &f->B.p, c) || dev_alloc(f, &f->B.v, c) || dev_alloc(f, &f->B.c, c) || dev_alloc(f, &f->B.id, c))
        return -2;
    if (dev_alloc(f, &f->keys, c) || dev_alloc(f, &f->keys_s, c) || dev_alloc(f, &f->idx, c) ||
        dev_alloc(f, &f->perm, c) || dev_alloc(f, &f->occ, (size_t)f->occ_words) ||
        dev_alloc(f, &f->cell_start, (size_t)c + 2) ||
        dev_alloc(f, &f->tile_cnt, (size_t)vis::tiles_for(c) + 2) || dev_alloc(f, &f->occupied, 1) ||
        dev_alloc(f, &f->stage, (size_t)12 * (c ? c : 1)))
        return -2;
    f->tmp_sort_bytes = nbmi::sort_pairs32_temp_bytes((size_t)c, 0, f->key_bits);
    char *t = nullptr;
    if (dev_alloc(f, &t, f->tmp_sort_bytes + 256)) return -2;
    f->tmp_sort = t;
    NBMI_HIP_CHECK(nbmi::sort_init_temp(t, f->stream));
    if (n > 0) {
        double *dp = f->stage, *dv = dp + 3 * n, *dc = dv + 3 * n;
        NBMI_HIP_CHECK(hipMemcpyAsync(dp, pos, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
        NBMI_HIP_CHECK(hipMemcpyAsync(dv, vel, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
        NBMI_HIP_CHECK(hipMemcpyAsync(dc, col, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
        k_split<<<nblocks(n), kBlock, 0, f->stream>>>(dp, dv, dc, f->A, n, 0);
        NBMI_HIP_CHECK(hipGetLastError());
    }
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

static bdmi_flock *bd_create(int64_t n, const double *pos, const double *vel, const double *col, const double *params,
                             int device, bool slab, int64_t capacity, const int32_t *ids, double x_lo, double x_hi,
                             int has_left, int has_right) {
    nbmi::clear_error();
    if (n < 0 || n > 1000000000 || !params || (n > 0 && (!pos || !vel || !col))) {
        nbmi::set_error("bdmi_create: bad arguments (n=%lld)", (long long)n);
        return nullptr;
    }
    const double bounds = params[0], perception = params[5], margin = params[1];
    if (!(bounds > 0) || !(perception > 0) || !(margin > 0)) {
        nbmi::set_error("bdmi_create: bounds, perception_radius and wall_margin must be > 0");
        return nullptr;
    }
    if (slab && (capacity < n || capacity < 1 || (n > 0 && !ids) || !(x_hi - x_lo >= 2 * perception))) {
        nbmi::set_error("bdmi_create_slab: capacity < n, missing ids, or a slab narrower than two cells");
        return nullptr;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        nbmi::set_error("bdmi_create: no HIP device available");
        return nullptr;
    }
    if (device < 0 || device >= count) {
        nbmi::set_error("bdmi_create: device %d out of range (have %d)", device, count);
        return nullptr;
    }
    bdmi_flock *f = new bdmi_flock();
    if (const char *e = getenv("BDMI_XCD")) f->xcd_contiguous = atoi(e) != 0;  // measurement knob
    f->n = n;
    f->device = device;
    f->slab = slab; f->cap = slab ? capacity : 0;
    f->x_lo = x_lo; f->x_hi = x_hi; f->has_left = has_left; f->has_right = has_right;
    memcpy(f->params, params, sizeof(f->params));
    // Flock.__init__ grid (flock.py:478-481)
    f->grid.cell_size = perception;
    const double dimf = ceil(bounds * 2 / perception) + 2;
    if (!(dimf >= 1) || dimf > 1290) {  // dim^3 must fit int32 cell ids
        nbmi::set_error("bdmi_create: grid dimension %.0f out of range", dimf);
        delete f;
        return nullptr;
    }
    f->grid.dim = (int)dimf;
    f->grid.offset = bounds + perception;
    f->grid.range = (int)ceil(perception / f->grid.cell_size);
    f->num_cells = (int64_t)f->grid.dim * f->grid.dim * f->grid.dim;
    f->key_bits = 1;
    while (((int64_t)1 << f->key_bits) < f->num_cells) f->key_bits++;
    int rc = bd_create_impl(f, pos, vel, col);
    if (rc == 0 && slab && n > 0) {  // global ids instead of row numbers
        hipError_t e = hipMemcpyAsync(f->stage, ids, (size_t)n * 4, hipMemcpyHostToDevice, f->stream);
        if (e == hipSuccess) {
            k_set_ids<<<nblocks(n), kBlock, 0, f->stream>>>((const int32_t *)f->stage, f->A.id, n);
            e = hipStreamSynchronize(f->stream);
        }
        if (e != hipSuccess) { nbmi::set_error("bdmi_create_slab: id upload failed: %s", hipGetErrorString(e)); rc = -2; }
    }
    if (rc != 0) {
        std::string keep = nbmi::get_error();
        bdmi_destroy(f);
        nbmi::set_error("%s", keep.c_str());
        return nullptr;
    }
    return f;
}

bdmi_flock *bdmi_create(int64_t n, const double *pos, const double *vel, const double *col, const double *params,
                        int device) {
    return bd_create(n, pos, vel, col, params, device, false, 0, nullptr, 0.0, 0.0, 0, 0);
}

// ---- slab mode --------------------------------------------------------------------------
bdmi_flock *bdmi_create_slab(int64_t n, const double *pos, const double *vel, const double *col, const int32_t *ids,
                             int64_t capacity, const double *params, double x_lo, double x_hi, int has_left, int has_right,
                             int device) {
    return bd_create(n, pos, vel, col, params, device, true, capacity, ids, x_lo, x_hi, has_left, has_right);
}

namespace {
// rows selected by `sel`, in row order, to packed rows or to the SoA arrays `dst`; the count stays on the device
int enqueue_select(bdmi_flock *f, const SlabSel &sel, int64_t n, Boids dst, double *dst_rows) {
    const int64_t ntiles = vis::tiles_for(n);
    k_sel_count<<<(int)ntiles, kBlock, 0, f->stream>>>(sel, n, f->tile_cnt);
    vis::k_scan_tiles<<<1, vis::kBlock, 0, f->stream>>>(f->tile_cnt, ntiles);
    k_sel_emit<<<(int)ntiles, kBlock, 0, f->stream>>>(sel, f->A, n, f->tile_cnt, dst, dst_rows);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}
int slab_check(bdmi_flock *f, const char *what) {
    if (int rc = check(f)) return rc;
    if (!f->slab) { nbmi::set_error("%s: not a slab-mode handle (bdmi_create_slab)", what); return -1; }
    return 0;
}
// drop the ghosts: owned rows to the front, in row order
int slab_compact_owned(bdmi_flock *f, int64_t *n_owned) {
    uint32_t total = 0;
    if (f->n > 0) {
        const SlabSel own{f->A.px, f->A.id, 0, 0.0, 0.0};
        if (int rc = enqueue_select(f, own, f->n, f->T, nullptr)) return rc;
        NBMI_HIP_CHECK(hipMemcpyAsync(&total, f->tile_cnt + vis::tiles_for(f->n), 4, hipMemcpyDeviceToHost, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
        std::swap(f->A, f->T);
    }
    f->n = total;
    *n_owned = total;
    return 0;
}
}  // namespace

int64_t bdmi_slab_count(bdmi_flock *f) {
    if (!f || !f->slab) return -1;
    int64_t n = 0;
    if (slab_compact_owned(f, &n)) return -1;
    return n;
}

int bdmi_slab_export(bdmi_flock *f, void *dev_left, int64_t *n_left, void *dev_right, int64_t *n_right) {
    if (int rc = slab_check(f, "bdmi_slab_export")) return rc;
    if (!n_left || !n_right || (f->has_left && !dev_left) || (f->has_right && !dev_right)) {
        nbmi::set_error("bdmi_slab_export: null buffer");
        return -1;
    }
    *n_left = *n_right = 0;
    int64_t owned = 0;
    if (int rc = slab_compact_owned(f, &owned)) return rc;  // last step's ghosts are gone
    if (owned == 0) return 0;
    const double cell = f->grid.cell_size;
    uint32_t cl = 0, cr = 0;
    const int64_t ntiles = vis::tiles_for(owned);
    if (f->has_left) {
        const SlabSel sel{f->A.px, f->A.id, 1, f->x_lo + cell, 0.0};
        if (int rc = enqueue_select(f, sel, owned, Boids{}, (double *)dev_left)) return rc;
        NBMI_HIP_CHECK(hipMemcpyAsync(&cl, f->tile_cnt + ntiles, 4, hipMemcpyDeviceToHost, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    }
    if (f->has_right) {
        const SlabSel sel{f->A.px, f->A.id, 2, 0.0, f->x_hi - cell};
        if (int rc = enqueue_select(f, sel, owned, Boids{}, (double *)dev_right)) return rc;
        NBMI_HIP_CHECK(hipMemcpyAsync(&cr, f->tile_cnt + ntiles, 4, hipMemcpyDeviceToHost, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    }
    // boids that have left the slab were just handed over: here they are ghosts for this step.  (Slabs at the
    // two ends of the domain keep their outer side: the walls turn those boids back.)
    k_slab_demote<<<nblocks(owned), kBlock, 0, f->stream>>>(f->A, owned, f->has_left ? f->x_lo : -INFINITY,
                                                           f->has_right ? f->x_hi : INFINITY);
    NBMI_HIP_CHECK(hipGetLastError());
    *n_left = cl;
    *n_right = cr;
    return 0;
}

int bdmi_slab_import(bdmi_flock *f, const void *dev_rows, int64_t count) {
    if (int rc = slab_check(f, "bdmi_slab_import")) return rc;
    if (count < 0 || (count > 0 && !dev_rows)) { nbmi::set_error("bdmi_slab_import: bad arguments"); return -1; }
    if (f->n + count > f->cap) {
        nbmi::set_error("bdmi_slab_import: %lld + %lld boids exceed the capacity %lld", (long long)f->n, (long long)count,
                        (long long)f->cap);
        return -4;
    }
    if (count == 0) return 0;
    k_slab_append<<<nblocks(count), kBlock, 0, f->stream>>>(f->A, f->n, (const double *)dev_rows, count,
                                                           f->has_left ? f->x_lo : -INFINITY, f->has_right ? f->x_hi : INFINITY);
    NBMI_HIP_CHECK(hipGetLastError());
    f->n += count;
    return 0;
}

int bdmi_slab_get(bdmi_flock *f, double *rows10, int64_t capacity, int64_t *count) {
    if (int rc = slab_check(f, "bdmi_slab_get")) return rc;
    if (!count) { nbmi::set_error("bdmi_slab_get: null count"); return -1; }
    *count = 0;
    if (f->n == 0) return 0;
    // owned rows {p, v, c, id}, packed, through the staging buffer (12 doubles per row are reserved)
    const SlabSel own{f->A.px, f->A.id, 0, 0.0, 0.0};
    if (int rc = enqueue_select(f, own, f->n, Boids{}, f->stage)) return rc;
    uint32_t total = 0;
    NBMI_HIP_CHECK(hipMemcpyAsync(&total, f->tile_cnt + vis::tiles_for(f->n), 4, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    *count = total;
    const int64_t rows = (int64_t)total < capacity ? (int64_t)total : capacity;
    if (rows > 0) {
        if (!rows10) { nbmi::set_error("bdmi_slab_get: null output"); return -1; }
        NBMI_HIP_CHECK(hipMemcpyAsync(rows10, f->stage, (size_t)rows * 80, hipMemcpyDeviceToHost, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    }
    return 0;
}

int bdmi_step(bdmi_flock *f, double dt, int substeps) {
    if (int rc = check(f)) return rc;
    if (substeps < 0) { nbmi::set_error("bdmi_step: substeps < 0"); return -1; }
    if (f->n == 0) return 0;
    const FlockP P = make_params(f, dt);
    for (int k = 0; k < substeps; k++) {
        if (int rc = enqueue_grid(f, f->timers)) return rc;
        k_flock<true><<<nblocks(f->n), kBlock, 0, f->stream>>>(f->B, f->A, f->occ, f->cell_start, f->n, f->grid, P,
                                                               nullptr, nullptr, nullptr, nullptr, f->xcd_contiguous);
        NBMI_HIP_CHECK(hipGetLastError());
        if (f->timers) {
            NBMI_HIP_CHECK(hipEventRecord(f->ev[3], f->stream));
            NBMI_HIP_CHECK(hipEventSynchronize(f->ev[3]));
            for (int p = 0; p < 3; p++) {
                float ms = 0.f;
                NBMI_HIP_CHECK(hipEventElapsedTime(&ms, f->ev[p], f->ev[p + 1]));
                f->ms[p] += ms;
            }
            f->timed++;
        }
    }
    return 0;
}

// synchronise, then look at the radix sort's sticky error word (a timed-out look-back = a corrupt cell order)
static int sync_checked(bdmi_flock *f) {
    unsigned sort_err = 0u;
    if (f->tmp_sort) NBMI_HIP_CHECK(nbmi::sort_error_word(f->tmp_sort, &sort_err, f->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    if (sort_err) {
        NBMI_HIP_CHECK(nbmi::sort_init_temp(f->tmp_sort, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
        nbmi::set_error("device radix sort: a look-back spin timed out; the steps since the last synchronisation are invalid");
        return -2;
    }
    return 0;
}

int bdmi_sync(bdmi_flock *f) {
    if (int rc = check(f)) return rc;
    return sync_checked(f);
}

int bdmi_get_state(bdmi_flock *f, double *pos, double *vel, double *col) {
    if (int rc = check(f)) return rc;
    const int64_t n = f->n;
    if (n == 0) return 0;
    double *dp = f->stage, *dv = dp + 3 * n, *dc = dv + 3 * n;
    const Boids &a = f->A;
    if (pos) {
        k_join<<<nblocks(n), kBlock, 0, f->stream>>>(a.px, a.py, a.pz, a.id, n, dp);
        NBMI_HIP_CHECK(hipMemcpyAsync(pos, dp, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    }
    if (vel) {
        k_join<<<nblocks(n), kBlock, 0, f->stream>>>(a.vx, a.vy, a.vz, a.id, n, dv);
        NBMI_HIP_CHECK(hipMemcpyAsync(vel, dv, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    }
    if (col) {
        k_join<<<nblocks(n), kBlock, 0, f->stream>>>(a.cr, a.cg, a.cb, a.id, n, dc);
        NBMI_HIP_CHECK(hipMemcpyAsync(col, dc, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    }
    NBMI_HIP_CHECK(hipGetLastError());
    return sync_checked(f);
}

int bdmi_set_state(bdmi_flock *f, const double *pos, const double *vel, const double *col) {
    if (int rc = check(f)) return rc;
    const int64_t n = f->n;
    if (n == 0) return 0;
    double *dp = f->stage, *dv = dp + 3 * n, *dc = dv + 3 * n;
    if (pos) NBMI_HIP_CHECK(hipMemcpyAsync(dp, pos, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
    if (vel) NBMI_HIP_CHECK(hipMemcpyAsync(dv, vel, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
    if (col) NBMI_HIP_CHECK(hipMemcpyAsync(dc, col, (size_t)n * 24, hipMemcpyHostToDevice, f->stream));
    k_split<<<nblocks(n), kBlock, 0, f->stream>>>(pos ? dp : nullptr, vel ? dv : nullptr, col ? dc : nullptr, f->A, n, 1);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

int bdmi_get_cell_indices(bdmi_flock *f, int32_t *out) {
    if (int rc = check(f)) return rc;
    const int64_t n = f->n;
    if (n == 0) return 0;
    if (!out) { nbmi::set_error("null output"); return -1; }
    int32_t *d = (int32_t *)f->stage;
    k_cells_out<<<nblocks(n), kBlock, 0, f->stream>>>(f->A, n, f->grid, d);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemcpyAsync(out, d, (size_t)n * 4, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

int bdmi_get_forces(bdmi_flock *f, double *sep, double *ali, double *coh, double *avg) {
    if (int rc = check(f)) return rc;
    const int64_t n = f->n;
    if (n == 0) return 0;
    if (!sep || !ali || !coh || !avg) { nbmi::set_error("null output"); return -1; }
    if (int rc = enqueue_grid(f, false)) return rc;
    double *d0 = f->stage, *d1 = d0 + 3 * n, *d2 = d1 + 3 * n, *d3 = d2 + 3 * n;
    const FlockP P = make_params(f, 0.0);
    k_flock<false><<<nblocks(n), kBlock, 0, f->stream>>>(f->B, f->A, f->occ, f->cell_start, n, f->grid, P, d0, d1,
                                                         d2, d3, f->xcd_contiguous);
    NBMI_HIP_CHECK(hipGetLastError());
    NBMI_HIP_CHECK(hipMemcpyAsync(sep, d0, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(ali, d1, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(coh, d2, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipMemcpyAsync(avg, d3, (size_t)n * 24, hipMemcpyDeviceToHost, f->stream));
    NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
    return 0;
}

int bdmi_grid_info(bdmi_flock *f, int32_t *grid_dim, int64_t *num_cells, int64_t *occupied) {
    if (int rc = check(f)) return rc;
    if (grid_dim) *grid_dim = f->grid.dim;
    if (num_cells) *num_cells = f->num_cells;
    if (occupied) {
        unsigned long long h = 0;
        if (f->n > 0) {
            NBMI_HIP_CHECK(hipMemsetAsync(f->occupied, 0, sizeof(unsigned long long), f->stream));
            int gb = nblocks(f->n);
            if (gb > 64) gb = 64;
            k_count_cells<<<gb, kBlock, 0, f->stream>>>(f->keys_s, f->n, f->occupied);
            NBMI_HIP_CHECK(hipGetLastError());
        }
        NBMI_HIP_CHECK(hipMemcpyAsync(&h, f->occupied, sizeof(h), hipMemcpyDeviceToHost, f->stream));
        NBMI_HIP_CHECK(hipStreamSynchronize(f->stream));
        *occupied = (int64_t)h;
    }
    return 0;
}

int bdmi_enable_timers(bdmi_flock *f, int enable) {
    if (int rc = check(f)) return rc;
    f->timers = enable != 0;
    return 0;
}

int bdmi_get_timers(bdmi_flock *f, double *ms3, int64_t *count, int reset) {
    if (int rc = check(f)) return rc;
    if (ms3) memcpy(ms3, f->ms, sizeof(f->ms));
    if (count) *count = f->timed;
    if (reset) { memset(f->ms, 0, sizeof(f->ms)); f->timed = 0; }
    return 0;
}

namespace {
// build_vertices_numba (flock.py:351-447): two triangles (tip/right/left, tip/up/down) per boid,
// float64 arithmetic in the reference's order, one rounding to float32 on store.
struct EmitCones {
    Boids a;
    double cone_length, cone_radius;
    float *verts, *cols;
    __device__ void operator()(int64_t k, int64_t, uint32_t slot) const {
        const double wux = 0.0, wuy = 1.0, wuz = 0.0, wrx = 1.0, wry = 0.0, wrz = 0.0;
        const double px = a.px[slot], py = a.py[slot], pz = a.pz[slot];
        const double vx = a.vx[slot], vy = a.vy[slot], vz = a.vz[slot];
        double speed = sqrt(vx * vx + vy * vy + vz * vz);
        if (speed < 0.0001) speed = 0.0001;
        const double fx = vx / speed, fy = vy / speed, fz = vz / speed;
        double rx = fy * wuz - fz * wuy;
        double ry = fz * wux - fx * wuz;
        double rz = fx * wuy - fy * wux;
        double r_len = sqrt(rx * rx + ry * ry + rz * rz);
        if (r_len < 0.1) {
            rx = fy * wrz - fz * wry;
            ry = fz * wrx - fx * wrz;
            rz = fx * wry - fy * wrx;
            r_len = sqrt(rx * rx + ry * ry + rz * rz);
        }
        if (r_len > 0.0001) { rx /= r_len; ry /= r_len; rz /= r_len; }
        const double ux = ry * fz - rz * fy;
        const double uy = rz * fx - rx * fz;
        const double uz = rx * fy - ry * fx;
        const double r = cone_radius;
        const float tip[3] = {(float)(px + fx * cone_length), (float)(py + fy * cone_length), (float)(pz + fz * cone_length)};
        float *o = verts + 18 * k;
        o[0] = tip[0]; o[1] = tip[1]; o[2] = tip[2];
        o[3] = (float)(px + rx * r); o[4] = (float)(py + ry * r); o[5] = (float)(pz + rz * r);
        o[6] = (float)(px - rx * r); o[7] = (float)(py - ry * r); o[8] = (float)(pz - rz * r);
        o[9] = tip[0]; o[10] = tip[1]; o[11] = tip[2];
        o[12] = (float)(px + ux * r); o[13] = (float)(py + uy * r); o[14] = (float)(pz + uz * r);
        o[15] = (float)(px - ux * r); o[16] = (float)(py - uy * r); o[17] = (float)(pz - uz * r);
        const float cr = (float)a.cr[slot], cg = (float)a.cg[slot], cb = (float)a.cb[slot];
        float *c = cols + 18 * k;
#pragma unroll
        for (int v = 0; v < 6; v++) { c[3 * v] = cr; c[3 * v + 1] = cg; c[3 * v + 2] = cb; }
    }
};
}  // namespace

int bdmi_visible_vertices(bdmi_flock *f, const double *cam12, double tan_h, double tan_v, double fog_end,
                          double cone_length, double cone_radius, float *out_vertices, float *out_colors,
                          int64_t capacity_boids, int64_t *count) {
    if (int rc = check(f)) return rc;
    if (!cam12 || !count || capacity_boids < 0 || (capacity_boids > 0 && (!out_vertices || !out_colors))) {
        nbmi::set_error("bdmi_visible_vertices: null argument");
        return -1;
    }
    const int64_t n = f->n;
    *count = 0;
    if (n == 0) return 0;
    const int64_t ntiles = vis::tiles_for(n);
    if (!f->vis_flag) {
        if (dev_alloc(f, &f->vis_flag, vis::flag_bytes(n)) || dev_alloc(f, &f->vis_slot, (size_t)ntiles * vis::kTile) ||
            dev_alloc(f, &f->vis_tiles, ntiles + 1) || dev_alloc(f, &f->vis_verts, (size_t)n * 18) ||
            dev_alloc(f, &f->vis_cols, (size_t)n * 18))
            return -2;
        NBMI_HIP_CHECK(hipMemsetAsync(f->vis_flag, 0, vis::flag_bytes(n), f->stream));
    }
    vis::Camera c;
    for (int k = 0; k < 3; k++) { c.p[k] = cam12[k]; c.f[k] = cam12[3 + k]; c.r[k] = cam12[6 + k]; c.u[k] = cam12[9 + k]; }
    c.tan_h = tan_h; c.tan_v = tan_v; c.z_near = 0.5; c.z_far = fog_end; c.margin = 1.0;
    const Boids &a = f->A;
    hipStream_t st = f->stream;
    vis::k_mark<<<nblocks(n), kBlock, 0, st>>>(a.px, a.py, a.pz, a.id, n, c, f->vis_flag, f->vis_slot);
    vis::k_count<<<(int)ntiles, vis::kBlock, 0, st>>>(f->vis_flag, n, f->vis_tiles);
    vis::k_scan_tiles<<<1, vis::kBlock, 0, st>>>(f->vis_tiles, ntiles);
    EmitCones e{a, cone_length, cone_radius, f->vis_verts, f->vis_cols};
    vis::k_emit<<<(int)ntiles, vis::kBlock, 0, st>>>(f->vis_flag, f->vis_slot, f->vis_tiles, n, n, e);
    NBMI_HIP_CHECK(hipGetLastError());
    uint32_t total = 0;
    NBMI_HIP_CHECK(hipMemcpyAsync(&total, f->vis_tiles + ntiles, 4, hipMemcpyDeviceToHost, st));
    NBMI_HIP_CHECK(hipStreamSynchronize(st));
    *count = total;
    const int64_t rows = (int64_t)total < capacity_boids ? (int64_t)total : capacity_boids;
    if (rows > 0) {
        NBMI_HIP_CHECK(hipMemcpyAsync(out_vertices, f->vis_verts, (size_t)rows * 72, hipMemcpyDeviceToHost, st));
        NBMI_HIP_CHECK(hipMemcpyAsync(out_colors, f->vis_cols, (size_t)rows * 72, hipMemcpyDeviceToHost, st));
        NBMI_HIP_CHECK(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
