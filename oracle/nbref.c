/*
 * nbref.c - CPU restatement of the reference's N-body hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle and the "port" CPU baseline of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (libnbmi.so + the Python package) never does and fails loudly without a GPU.
 *
 * Pinned against golden vectors produced by the reference's own functions
 * (oracle/gen_golden.py -> tests/golden/, checked by tests/test_oracle_*.py).
 *
 * Each function restates one reference function, float64, same operation order so that a
 * strict-IEEE build (-O2 -ffp-contract=off, no -ffast-math) reproduces the reference run
 * under CPython bit for bit.  Reference = /root/reference (Keshav-Madhav/3d-spatial-sim-...):
 *
 *   nbref_compute_bounds          nbody/simulation.py:308-317   compute_bounds
 *   nbref_build_octree            nbody/simulation.py:63-198    build_octree (serial insertion)
 *   nbref_compute_forces_bh       nbody/simulation.py:201-278   compute_forces_barnes_hut
 *   nbref_update                  nbody/simulation.py:281-305   update_positions_velocities
 *   nbref_colors                  nbody/simulation.py:320-400   compute_colors_by_velocity
 *   nbref_direct_forces           nbody/gpu_backend.py:145-174  compute_forces_brute_cuda (f64)
 *   nbref_step                    tools/record.py:833-858       one CPU substep of record()
 *
 * Helpers that have no reference twin (they describe the reference tree so that a sort-based
 * build can be compared with it):
 *   nbref_body_keys     replays get_octant/get_octant_center (simulation.py:38-60) per body
 *   nbref_tree_cells    (level, octant-path key) of every node of a built tree
 *   nbref_group_walk_stats  union-of-visits statistics for 64-body groups (design analysis)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NBREF_MAX_TREE_NODES 8000000 /* nbody/simulation.py:35 */
#define NBREF_STACK 64               /* nbody/simulation.py:233 */

int nbref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void nbref_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* nbody/simulation.py:308-317 */
double nbref_compute_bounds(const double *pos, int64_t n) {
    double max_extent = 0.0;
    for (int64_t i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            double ext = fabs(pos[3 * i + d]);
            if (ext > max_extent) max_extent = ext;
        }
    return max_extent * 1.1 + 10.0;
}

/* nbody/simulation.py:38-49 */
static inline int get_octant(double px, double py, double pz, double cx, double cy, double cz) {
    int o = 0;
    if (px >= cx) o |= 1;
    if (py >= cy) o |= 2;
    if (pz >= cz) o |= 4;
    return o;
}

/* nbody/simulation.py:63-198.  `cap` is the reference's MAX_TREE_NODES constant (pass
 * NBREF_MAX_TREE_NODES for faithful behaviour, or a huge value for the uncapped tree);
 * `rows` is the number of rows actually allocated - the reference never checks it ([quirk],
 * SURVEY 8a row 4); here running out of rows returns -1 instead of writing out of bounds. */
int64_t nbref_build_octree(const double *pos, const double *masses, int64_t n, double bounds,
                           double *centers, double *half, double *nmass, double *com,
                           int32_t *children, int32_t *body_idx, uint8_t *is_leaf,
                           int64_t rows, int64_t cap) {
    if (rows < 1) return -1;
    centers[0] = centers[1] = centers[2] = 0.0;
    half[0] = bounds;
    nmass[0] = 0.0;
    com[0] = com[1] = com[2] = 0.0;
    body_idx[0] = -1;
    is_leaf[0] = 1;
    for (int c = 0; c < 8; c++) children[c] = -1;
    int64_t num_nodes = 1;

    for (int64_t i = 0; i < n; i++) {
        double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        double m = masses[i];
        int64_t cur = 0;
        for (;;) {
            double cx = centers[3 * cur], cy = centers[3 * cur + 1], cz = centers[3 * cur + 2];
            double hs = half[cur];
            if (is_leaf[cur]) {
                if (body_idx[cur] == -1) {
                    body_idx[cur] = (int32_t)i;
                    nmass[cur] = m;
                    com[3 * cur] = px; com[3 * cur + 1] = py; com[3 * cur + 2] = pz;
                    break;
                } else {
                    int32_t old = body_idx[cur];
                    double opx = pos[3 * (int64_t)old], opy = pos[3 * (int64_t)old + 1], opz = pos[3 * (int64_t)old + 2];
                    double om = masses[old];
                    is_leaf[cur] = 0;
                    body_idx[cur] = -1;
                    int oct = get_octant(opx, opy, opz, cx, cy, cz);
                    if (children[8 * cur + oct] == -1) {
                        int64_t ch = num_nodes;
                        num_nodes += 1;
                        if (num_nodes >= cap) goto body_done; /* `break` out of the while loop */
                        if (ch >= rows) return -1;
                        children[8 * cur + oct] = (int32_t)ch;
                        double q = hs * 0.5;
                        centers[3 * ch] = (oct & 1) ? cx + q : cx - q;
                        centers[3 * ch + 1] = (oct & 2) ? cy + q : cy - q;
                        centers[3 * ch + 2] = (oct & 4) ? cz + q : cz - q;
                        half[ch] = hs * 0.5;
                        nmass[ch] = om;
                        com[3 * ch] = opx; com[3 * ch + 1] = opy; com[3 * ch + 2] = opz;
                        body_idx[ch] = old;
                        is_leaf[ch] = 1;
                        for (int c = 0; c < 8; c++) children[8 * ch + c] = -1;
                    }
                    /* loop again: `cur` is now internal */
                }
            } else {
                double total = nmass[cur] + m;
                if (total > 0) {
                    com[3 * cur] = (com[3 * cur] * nmass[cur] + px * m) / total;
                    com[3 * cur + 1] = (com[3 * cur + 1] * nmass[cur] + py * m) / total;
                    com[3 * cur + 2] = (com[3 * cur + 2] * nmass[cur] + pz * m) / total;
                }
                nmass[cur] = total;
                int oct = get_octant(px, py, pz, cx, cy, cz);
                if (children[8 * cur + oct] == -1) {
                    int64_t ch = num_nodes;
                    num_nodes += 1;
                    if (num_nodes >= cap) goto body_done;
                    if (ch >= rows) return -1;
                    children[8 * cur + oct] = (int32_t)ch;
                    double q = hs * 0.5;
                    centers[3 * ch] = (oct & 1) ? cx + q : cx - q;
                    centers[3 * ch + 1] = (oct & 2) ? cy + q : cy - q;
                    centers[3 * ch + 2] = (oct & 4) ? cz + q : cz - q;
                    half[ch] = hs * 0.5;
                    nmass[ch] = m;
                    com[3 * ch] = px; com[3 * ch + 1] = py; com[3 * ch + 2] = pz;
                    body_idx[ch] = (int32_t)i;
                    is_leaf[ch] = 1;
                    for (int c = 0; c < 8; c++) children[8 * ch + c] = -1;
                    break;
                } else {
                    cur = children[8 * cur + oct];
                }
            }
        }
    body_done:;
    }
    return num_nodes;
}

/* stats[0]=total visits, [1]=accepted-with-force, [2]=dropped pushes (stack full),
 * [3]=peak stack occupancy, [4]=opened nodes, [5]=visits of single-child internal nodes
 * ("chain" cells; design analysis). May be NULL (6 entries otherwise). */
void nbref_compute_forces_bh(const double *pos, const double *masses, double *acc,
                             const double *centers, const double *half, const double *nmass,
                             const double *com, const int32_t *children, const int32_t *body_idx,
                             const uint8_t *is_leaf, int64_t num_nodes, int64_t n, double theta,
                             double G, double softening, int64_t *stats) {
    (void)masses; (void)centers;
    const double eps2 = softening * softening;
    int64_t visits = 0, accepted = 0, dropped = 0, opened = 0, chain = 0;
    int peak = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : visits, accepted, dropped, opened, chain) reduction(max : peak)
    for (int64_t i = 0; i < n; i++) {
        double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        double ax = 0.0, ay = 0.0, az = 0.0;
        int32_t stack[NBREF_STACK];
        stack[0] = 0;
        int sp = 1;
        while (sp > 0) {
            sp -= 1;
            int64_t node = stack[sp];
            if (node < 0 || node >= num_nodes) continue;
            visits++;
            if (!is_leaf[node]) {
                int nch = 0;
                for (int c = 0; c < 8; c++) nch += children[8 * node + c] >= 0;
                chain += nch == 1;
            }
            if (is_leaf[node] && body_idx[node] == i) continue;
            double dx = com[3 * node] - px;
            double dy = com[3 * node + 1] - py;
            double dz = com[3 * node + 2] - pz;
            double dist_sq = dx * dx + dy * dy + dz * dz + eps2;
            double dist = sqrt(dist_sq);
            double node_size = half[node] * 2.0;
            if (is_leaf[node] || (node_size / dist < theta)) {
                if (nmass[node] > 0 && dist_sq > eps2) {
                    double inv_dist3 = 1.0 / (dist * dist_sq);
                    double fm = G * nmass[node] * inv_dist3;
                    ax += dx * fm;
                    ay += dy * fm;
                    az += dz * fm;
                    accepted++;
                }
            } else {
                opened++;
                for (int c = 0; c < 8; c++) {
                    int32_t ch = children[8 * node + c];
                    if (ch >= 0) {
                        if (sp < NBREF_STACK) {
                            stack[sp++] = ch;
                            if (sp > peak) peak = sp;
                        } else {
                            dropped++;
                        }
                    }
                }
            }
        }
        acc[3 * i] = ax; acc[3 * i + 1] = ay; acc[3 * i + 2] = az;
    }
    if (stats) {
        stats[0] = visits; stats[1] = accepted; stats[2] = dropped; stats[3] = peak; stats[4] = opened; stats[5] = chain;
    }
}

/* nbody/simulation.py:281-305 */
void nbref_update(double *pos, double *vel, const double *acc, double damping, double dt, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) {
            double v = vel[3 * i + d];
            v += acc[3 * i + d] * dt;
            v *= damping;
            vel[3 * i + d] = v;
            pos[3 * i + d] += v * dt;
        }
    }
}

/* nbody/simulation.py:320-400 (colours are float32 stores of float64 expressions) */
void nbref_colors(const double *vel, float *colors, int64_t n, double max_speed) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
        double speed = sqrt(vx * vx + vy * vy + vz * vz);
        double t = speed / max_speed;
        if (t > 1.0) t = 1.0;
        double r, g, b, s, s2;
        if (t < 0.55) {
            if (t < 0.15) {
                s = t / 0.15;
                r = 0.4 - 0.2 * s; g = 0.2 + 0.2 * s; b = 0.8 + 0.1 * s;
            } else if (t < 0.30) {
                s = (t - 0.15) / 0.15;
                r = 0.2 + 0.1 * s; g = 0.4 + 0.1 * s; b = 0.9 + 0.05 * s;
            } else {
                s = (t - 0.30) / 0.25;
                if (s < 0.6) {
                    s2 = s / 0.6;
                    r = 0.3 - 0.1 * s2; g = 0.5 + 0.3 * s2; b = 0.95 + 0.05 * s2;
                } else {
                    s2 = (s - 0.6) / 0.4;
                    r = 0.2 + 0.8 * s2; g = 0.8 + 0.2 * s2; b = 1.0;
                }
            }
        } else if (t < 0.90) {
            r = 1.0; g = 1.0; b = 1.0;
        } else if (t < 0.95) {
            s = (t - 0.90) / 0.05;
            r = 1.0; g = 1.0 - 0.05 * s; b = 1.0 - 1.0 * s;
        } else if (t < 0.99) {
            s = (t - 0.95) / 0.04;
            r = 1.0; g = 0.95 - 0.45 * s; b = 0.0;
        } else {
            s = (t - 0.99) / 0.01;
            r = 1.0; g = 0.5 - 0.5 * s; b = 0.0;
        }
        colors[3 * i] = (float)r; colors[3 * i + 1] = (float)g; colors[3 * i + 2] = (float)b;
    }
}

/* nbody/gpu_backend.py:145-174 (brute) == :179-240 (tiled): identical sum, j ascending */
void nbref_direct_forces(const double *pos, const double *masses, double *acc, double G,
                         double softening, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (int64_t j = 0; j < n; j++) {
            if (i == j) continue;
            double dx = pos[3 * j] - px, dy = pos[3 * j + 1] - py, dz = pos[3 * j + 2] - pz;
            double dist_sq = dx * dx + dy * dy + dz * dz + softening * softening;
            double inv = 1.0 / sqrt(dist_sq);
            double inv3 = inv * inv * inv;
            double f = G * masses[j] * inv3;
            ax += f * dx; ay += f * dy; az += f * dz;
        }
        acc[3 * i] = ax; acc[3 * i + 1] = ay; acc[3 * i + 2] = az;
    }
}

/* Same sum for a SUBSET of the bodies (rows idx[0..k-1]) against all n: lets a test check the all-pairs
 * kernel at BASELINE config 3's full size on a sample (the full N^2 in float64 would take hours). */
void nbref_direct_forces_subset(const double *pos, const double *masses, const int64_t *idx, int64_t k, double *acc,
                                double G, double softening, int64_t n) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t s = 0; s < k; s++) {
        int64_t i = idx[s];
        double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (int64_t j = 0; j < n; j++) {
            if (i == j) continue;
            double dx = pos[3 * j] - px, dy = pos[3 * j + 1] - py, dz = pos[3 * j + 2] - pz;
            double dist_sq = dx * dx + dy * dy + dz * dz + softening * softening;
            double inv = 1.0 / sqrt(dist_sq);
            double inv3 = inv * inv * inv;
            double f = G * masses[j] * inv3;
            ax += f * dx; ay += f * dy; az += f * dz;
        }
        acc[3 * s] = ax; acc[3 * s + 1] = ay; acc[3 * s + 2] = az;
    }
}

/* nbody/gpu_backend.py:243-257 update_bodies_cuda: v=(v+a dt)*damping; x+=v dt */
void nbref_direct_update(double *pos, double *vel, const double *acc, double dt, double damping, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < 3 * n; i++) {
        vel[i] = (vel[i] + acc[i] * dt) * damping;
        pos[i] += vel[i] * dt;
    }
}

/* One CPU substep exactly as tools/record.py:835-858 (bounds, three fills, build, walk,
 * update).  Work arrays are caller-allocated with `rows` rows.  Returns num_nodes (or -1).
 * phase_s (may be NULL) receives wall seconds for [bounds, fill, build, walk, update]. */
static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

int64_t nbref_step(double *pos, double *vel, const double *masses, double *acc, int64_t n,
                   double theta, double G, double softening, double damping, double dt,
                   double *centers, double *half, double *nmass, double *com, int32_t *children,
                   int32_t *body_idx, uint8_t *is_leaf, int64_t rows, int64_t cap, int64_t *stats,
                   double *phase_s) {
    double t0 = now_s();
    double bounds = nbref_compute_bounds(pos, n);
    double t1 = now_s();
    memset(children, 0xff, (size_t)rows * 8 * sizeof(int32_t));
    memset(body_idx, 0xff, (size_t)rows * sizeof(int32_t));
    memset(is_leaf, 1, (size_t)rows);
    double t2 = now_s();
    int64_t nn = nbref_build_octree(pos, masses, n, bounds, centers, half, nmass, com, children,
                                    body_idx, is_leaf, rows, cap);
    if (nn < 0) return nn;
    double t3 = now_s();
    nbref_compute_forces_bh(pos, masses, acc, centers, half, nmass, com, children, body_idx, is_leaf,
                            nn, n, theta, G, softening, stats);
    double t4 = now_s();
    nbref_update(pos, vel, acc, damping, dt, n);
    double t5 = now_s();
    if (phase_s) {
        phase_s[0] += t1 - t0; phase_s[1] += t2 - t1; phase_s[2] += t3 - t2;
        phase_s[3] += t4 - t3; phase_s[4] += t5 - t4;
    }
    return nn;
}

/* ---- helpers describing the reference tree ------------------------------------------- */

/* Replays the reference's compare/halve recurrence (get_octant + get_octant_center,
 * simulation.py:38-60) from the root cube [-bounds,bounds]^3 for 42 levels.  Octal digit
 * per level = x>=cx | (y>=cy)<<1 | (z>=cz)<<2, most significant digit = level 1.
 * hi = levels 1..21 (63 bits), lo = levels 22..42. */
void nbref_body_keys(const double *pos, int64_t n, double bounds, uint64_t *hi, uint64_t *lo) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        double cx = 0.0, cy = 0.0, cz = 0.0, hs = bounds;
        uint64_t k[2] = {0, 0};
        for (int w = 0; w < 2; w++)
            for (int l = 0; l < 21; l++) {
                int o = get_octant(px, py, pz, cx, cy, cz);
                double q = hs * 0.5;
                cx = (o & 1) ? cx + q : cx - q;
                cy = (o & 2) ? cy + q : cy - q;
                cz = (o & 4) ? cz + q : cz - q;
                hs = hs * 0.5;
                k[w] = (k[w] << 3) | (uint64_t)o;
            }
        hi[i] = k[0];
        lo[i] = k[1];
    }
}

/* (level, path key) of every node of a tree built by nbref_build_octree; key is the octal
 * path from the root (levels <= 21 fit 63 bits; deeper nodes get key = UINT64_MAX). */
int nbref_tree_cells(const int32_t *children, int64_t num_nodes, int32_t *level, uint64_t *key) {
    int32_t *queue = (int32_t *)malloc((size_t)num_nodes * sizeof(int32_t));
    if (!queue) return -1;
    int64_t head = 0, tail = 0;
    level[0] = 0; key[0] = 0;
    queue[tail++] = 0;
    while (head < tail) {
        int32_t u = queue[head++];
        for (int c = 0; c < 8; c++) {
            int32_t v = children[8 * (int64_t)u + c];
            if (v >= 0 && v < num_nodes) {
                level[v] = level[u] + 1;
                key[v] = (level[v] <= 21 && key[u] != UINT64_MAX) ? ((key[u] << 3) | (uint64_t)c) : UINT64_MAX;
                queue[tail++] = v;
            }
        }
    }
    free(queue);
    return (tail == num_nodes) ? 0 : 1;
}

/* Design analysis: for groups of `gs` bodies taken in the order `order` (e.g. key-sorted),
 * count the union of nodes visited by the group when each body applies the reference's own
 * per-body opening test (group descends into a node if ANY member opens it).
 * out[0]=sum over groups of union visits, out[1]=sum of per-body visits, out[2]=groups,
 * out[3]=max union visits of one group. */
void nbref_group_walk_stats(const double *pos, const int64_t *order, int64_t n, int gs,
                            const double *half, const double *com, const int32_t *children,
                            const int32_t *body_idx, const uint8_t *is_leaf, double theta,
                            double softening, int64_t *out) {
    const double eps2 = softening * softening;
    int64_t ngroups = (n + gs - 1) / gs;
    int64_t sum_union = 0, sum_body = 0, max_union = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : sum_union, sum_body) reduction(max : max_union)
    for (int64_t g = 0; g < ngroups; g++) {
        int64_t lo = g * gs, hi = lo + gs > n ? n : lo + gs;
        int cnt = (int)(hi - lo);
        /* explicit stack of (node, active mask) - masks up to 128 members */
        typedef unsigned __int128 mask_t;
        int cap = 4096, sp = 0;
        int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
        mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
        sn[0] = 0; sm[0] = (cnt == 128) ? ~(mask_t)0 : ((((mask_t)1) << cnt) - 1); sp = 1;
        int64_t uni = 0;
        while (sp > 0) {
            sp--;
            int32_t node = sn[sp];
            mask_t mask = sm[sp];
            uni++;
            mask_t open = 0;
            for (int l = 0; l < cnt; l++) {
                if (!((mask >> l) & 1)) continue;
                sum_body++;
                int64_t i = order[lo + l];
                if (is_leaf[node]) continue;
                double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1],
                       dz = com[3 * node + 2] - pos[3 * i + 2];
                double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                if (!(half[node] * 2.0 / dist < theta)) open |= ((mask_t)1) << l;
            }
            if (open) {
                for (int c = 0; c < 8; c++) {
                    int32_t ch = children[8 * (int64_t)node + c];
                    if (ch >= 0) {
                        if (sp == cap) {
                            cap *= 2;
                            sn = (int32_t *)realloc(sn, sizeof(int32_t) * cap);
                            sm = (mask_t *)realloc(sm, sizeof(mask_t) * cap);
                        }
                        sn[sp] = ch; sm[sp] = open; sp++;
                    }
                }
            }
        }
        free(sn); free(sm);
        sum_union += uni;
        if (uni > max_union) max_union = uni;
    }
    (void)body_idx;
    out[0] = sum_union; out[1] = sum_body; out[2] = ngroups; out[3] = max_union;
}

/* Design analysis (round 2): like nbref_group_walk_stats, plus what the lock-step walk of a group of
 * `gs` bodies looks like visit by visit.  For every node the group visits:
 *   a = members taking part (all their ancestors were opened by them), t = of those, members that
 *   accept the node (leaf or opening test passed), o = a - t members that open it.
 * hist_active[a] += 1 (a = 1..gs);  cls[0..2] = visits with o == 0 (unanimous accept) / t == 0
 * (unanimous open) / mixed;  cls[3..5] = sum of a over those classes;  cls[6] = sum of t (accepts);
 * cls[7] = visits at which a == gs. */
void nbref_group_walk_hist(const double *pos, const int64_t *order, int64_t n, int gs,
                           const double *half, const double *com, const int32_t *children,
                           const uint8_t *is_leaf, double theta, double softening, int64_t *hist_active,
                           int64_t *cls) {
    const double eps2 = softening * softening;
    int64_t ngroups = (n + gs - 1) / gs;
    for (int i = 0; i <= gs; i++) hist_active[i] = 0;
    for (int i = 0; i < 8; i++) cls[i] = 0;
#pragma omp parallel
    {
        int64_t *h = (int64_t *)calloc((size_t)gs + 1, sizeof(int64_t));
        int64_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 16)
        for (int64_t g = 0; g < ngroups; g++) {
            int64_t lo = g * gs, hi = lo + gs > n ? n : lo + gs;
            int cnt = (int)(hi - lo);
            typedef unsigned __int128 mask_t;
            int cap = 4096, sp = 0;
            int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
            mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
            sn[0] = 0; sm[0] = (cnt == 128) ? ~(mask_t)0 : ((((mask_t)1) << cnt) - 1); sp = 1;
            while (sp > 0) {
                sp--;
                int32_t node = sn[sp];
                mask_t mask = sm[sp];
                mask_t open = 0;
                int a = 0, o = 0;
                for (int l = 0; l < cnt; l++) {
                    if (!((mask >> l) & 1)) continue;
                    a++;
                    int64_t i = order[lo + l];
                    if (is_leaf[node]) continue;
                    double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1],
                           dz = com[3 * node + 2] - pos[3 * i + 2];
                    double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                    if (!(half[node] * 2.0 / dist < theta)) { open |= ((mask_t)1) << l; o++; }
                }
                h[a]++;
                int k = (o == 0) ? 0 : (o == a ? 1 : 2);
                c[k]++; c[3 + k] += a; c[6] += a - o;
                if (a == gs) c[7]++;
                if (open) {
                    for (int ch8 = 0; ch8 < 8; ch8++) {
                        int32_t ch = children[8 * (int64_t)node + ch8];
                        if (ch >= 0) {
                            if (sp == cap) {
                                cap *= 2;
                                sn = (int32_t *)realloc(sn, sizeof(int32_t) * cap);
                                sm = (mask_t *)realloc(sm, sizeof(mask_t) * cap);
                            }
                            sn[sp] = ch; sm[sp] = open; sp++;
                        }
                    }
                }
            }
            free(sn); free(sm);
        }
#pragma omp critical
        {
            for (int i = 0; i <= gs; i++) hist_active[i] += h[i];
            for (int i = 0; i < 8; i++) cls[i] += c[i];
        }
        free(h);
    }
}

/* ---------------------------------------------------------------------------------------------
 * Render-side reduction: compute_visibility_points, nbody/simulation.py:403-434.
 * z < 0.1 or z > far_dist -> hidden; else |x| < z*tan_h*1.2 and |y| < z*tan_v*1.2.
 * cam = {pos[3], forward[3], right[3], up[3]}.
 * ------------------------------------------------------------------------------------------- */
void nbref_visibility_points(const double *pos, const double *cam, double tan_h, double tan_v, double far_dist,
                             uint8_t *visible_mask, int64_t n) {
    const double *cp = cam, *cf = cam + 3, *cr = cam + 6, *cu = cam + 9;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const double dx = pos[3 * i] - cp[0], dy = pos[3 * i + 1] - cp[1], dz = pos[3 * i + 2] - cp[2];
        const double z = dx * cf[0] + dy * cf[1] + dz * cf[2];
        if (z < 0.1 || z > far_dist) {
            visible_mask[i] = 0;
            continue;
        }
        const double x = dx * cr[0] + dy * cr[1] + dz * cr[2];
        const double y = dx * cu[0] + dy * cu[1] + dz * cu[2];
        const double half_width = z * tan_h * 1.2;
        const double half_height = z * tan_v * 1.2;
        visible_mask[i] = (fabs(x) < half_width && fabs(y) < half_height) ? 1 : 0;
    }
}
