/*
 * bdref.c - CPU restatement of the reference's boids hot path.  TEST INFRASTRUCTURE ONLY
 * (see the header of nbref.c for the rules: tests/, smoke() and bench.py's cpu_baseline only).
 *
 * float64, same operation order as the reference functions in /root/reference/boids/flock.py:
 *   bdref_assign_cells              flock.py:16-44    get_cell_index / assign_cells
 *   bdref_build_cell_lists          flock.py:47-65    build_cell_lists (serial)
 *   bdref_compute_flocking_spatial  flock.py:68-238   compute_flocking_spatial
 *   bdref_update_physics            flock.py:241-308  update_physics_numba
 *   bdref_step                      flock.py:627-678  Flock.update (sort supplied by caller or
 *                                                     done here with a stable counting sort)
 * The reference orders boids inside a cell by np.argsort(cell_indices) (introsort, unstable,
 * flock.py:618); callers that want bit-identical sums pass that permutation in.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* flock.py:16-27: int() truncates toward zero, then clamp to [0, grid_dim-1] */
static inline int32_t cell_index(double x, double y, double z, double cell_size, int grid_dim, double offset) {
    int cx = (int)((x + offset) / cell_size);
    int cy = (int)((y + offset) / cell_size);
    int cz = (int)((z + offset) / cell_size);
    cx = clampi(cx, 0, grid_dim - 1);
    cy = clampi(cy, 0, grid_dim - 1);
    cz = clampi(cz, 0, grid_dim - 1);
    return cx + cy * grid_dim + cz * grid_dim * grid_dim;
}

void bdref_assign_cells(const double *pos, int32_t *cell_indices, double cell_size, int grid_dim,
                        double offset, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        cell_indices[i] = cell_index(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], cell_size, grid_dim, offset);
}

void bdref_build_cell_lists(const int32_t *cell_indices, const int32_t *sorted_indices,
                            int32_t *cell_starts, int32_t *cell_counts, int64_t n, int64_t num_cells) {
    for (int64_t i = 0; i < num_cells; i++) {
        cell_starts[i] = -1;
        cell_counts[i] = 0;
    }
    for (int64_t i = 0; i < n; i++) {
        int32_t cell = cell_indices[sorted_indices[i]];
        if (cell_starts[cell] == -1) cell_starts[cell] = (int32_t)i;
        cell_counts[cell] += 1;
    }
}

/* stable counting sort of boid indices by cell (a valid instance of the reference's
 * unspecified within-cell order) */
void bdref_sort_by_cell(const int32_t *cell_indices, int32_t *sorted_indices, int64_t n, int64_t num_cells) {
    int64_t *cnt = (int64_t *)calloc((size_t)num_cells + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) cnt[cell_indices[i] + 1]++;
    for (int64_t c = 0; c < num_cells; c++) cnt[c + 1] += cnt[c];
    for (int64_t i = 0; i < n; i++) sorted_indices[cnt[cell_indices[i]]++] = (int32_t)i;
    free(cnt);
}

void bdref_compute_flocking_spatial(
    const double *pos, const double *vel, const double *col, const int32_t *sorted_indices,
    const int32_t *cell_starts, const int32_t *cell_counts, double *sepf, double *alif, double *cohf,
    double *avg_colors, double cell_size, int grid_dim, double offset, double perception_radius,
    double separation_radius, double separation_weight, double alignment_weight,
    double cohesion_weight, double max_speed, double max_force, int64_t n) {
    const double perception_sq = perception_radius * perception_radius;
    const double separation_sq = separation_radius * separation_radius;
    const int cell_range = (int)ceil(perception_radius / cell_size);
#pragma omp parallel for schedule(dynamic, 512)
    for (int64_t i = 0; i < n; i++) {
        const double *pi = pos + 3 * i, *vi = vel + 3 * i;
        int cx = clampi((int)((pi[0] + offset) / cell_size), 0, grid_dim - 1);
        int cy = clampi((int)((pi[1] + offset) / cell_size), 0, grid_dim - 1);
        int cz = clampi((int)((pi[2] + offset) / cell_size), 0, grid_dim - 1);
        double sep_x = 0, sep_y = 0, sep_z = 0, ali_x = 0, ali_y = 0, ali_z = 0;
        double coh_x = 0, coh_y = 0, coh_z = 0, col_r = 0, col_g = 0, col_b = 0;
        int64_t sep_count = 0, neighbor_count = 0;
        for (int dcx = -cell_range; dcx <= cell_range; dcx++) {
            int ncx = cx + dcx;
            if (ncx < 0 || ncx >= grid_dim) continue;
            for (int dcy = -cell_range; dcy <= cell_range; dcy++) {
                int ncy = cy + dcy;
                if (ncy < 0 || ncy >= grid_dim) continue;
                for (int dcz = -cell_range; dcz <= cell_range; dcz++) {
                    int ncz = cz + dcz;
                    if (ncz < 0 || ncz >= grid_dim) continue;
                    int64_t cell = ncx + (int64_t)ncy * grid_dim + (int64_t)ncz * grid_dim * grid_dim;
                    int32_t start = cell_starts[cell];
                    if (start == -1) continue;
                    int32_t count = cell_counts[cell];
                    for (int32_t k = 0; k < count; k++) {
                        int64_t j = sorted_indices[start + k];
                        if (i == j) continue;
                        double dx = pi[0] - pos[3 * j], dy = pi[1] - pos[3 * j + 1], dz = pi[2] - pos[3 * j + 2];
                        double dist_sq = dx * dx + dy * dy + dz * dz;
                        if (dist_sq < perception_sq && dist_sq > 0.0001) {
                            double dist = sqrt(dist_sq);
                            if (dist_sq < separation_sq) {
                                double inv_dist = 1.0 / dist;
                                sep_x += dx * inv_dist / dist;
                                sep_y += dy * inv_dist / dist;
                                sep_z += dz * inv_dist / dist;
                                sep_count += 1;
                            }
                            ali_x += vel[3 * j]; ali_y += vel[3 * j + 1]; ali_z += vel[3 * j + 2];
                            coh_x += pos[3 * j]; coh_y += pos[3 * j + 1]; coh_z += pos[3 * j + 2];
                            col_r += col[3 * j]; col_g += col[3 * j + 1]; col_b += col[3 * j + 2];
                            neighbor_count += 1;
                        }
                    }
                }
            }
        }
        if (sep_count > 0) {
            sep_x /= sep_count; sep_y /= sep_count; sep_z /= sep_count;
            double mag = sqrt(sep_x * sep_x + sep_y * sep_y + sep_z * sep_z);
            if (mag > 0) {
                sep_x = (sep_x / mag) * max_speed - vi[0];
                sep_y = (sep_y / mag) * max_speed - vi[1];
                sep_z = (sep_z / mag) * max_speed - vi[2];
                mag = sqrt(sep_x * sep_x + sep_y * sep_y + sep_z * sep_z);
                if (mag > max_force) {
                    sep_x = (sep_x / mag) * max_force;
                    sep_y = (sep_y / mag) * max_force;
                    sep_z = (sep_z / mag) * max_force;
                }
                sepf[3 * i] = sep_x * separation_weight;
                sepf[3 * i + 1] = sep_y * separation_weight;
                sepf[3 * i + 2] = sep_z * separation_weight;
            }
        }
        if (neighbor_count > 0) {
            ali_x /= neighbor_count; ali_y /= neighbor_count; ali_z /= neighbor_count;
            double mag = sqrt(ali_x * ali_x + ali_y * ali_y + ali_z * ali_z);
            if (mag > 0) {
                ali_x = (ali_x / mag) * max_speed - vi[0];
                ali_y = (ali_y / mag) * max_speed - vi[1];
                ali_z = (ali_z / mag) * max_speed - vi[2];
                mag = sqrt(ali_x * ali_x + ali_y * ali_y + ali_z * ali_z);
                if (mag > max_force) {
                    ali_x = (ali_x / mag) * max_force;
                    ali_y = (ali_y / mag) * max_force;
                    ali_z = (ali_z / mag) * max_force;
                }
                alif[3 * i] = ali_x * alignment_weight;
                alif[3 * i + 1] = ali_y * alignment_weight;
                alif[3 * i + 2] = ali_z * alignment_weight;
            }
            coh_x = coh_x / neighbor_count - pi[0];
            coh_y = coh_y / neighbor_count - pi[1];
            coh_z = coh_z / neighbor_count - pi[2];
            mag = sqrt(coh_x * coh_x + coh_y * coh_y + coh_z * coh_z);
            if (mag > 0) {
                coh_x = (coh_x / mag) * max_speed - vi[0];
                coh_y = (coh_y / mag) * max_speed - vi[1];
                coh_z = (coh_z / mag) * max_speed - vi[2];
                mag = sqrt(coh_x * coh_x + coh_y * coh_y + coh_z * coh_z);
                if (mag > max_force) {
                    coh_x = (coh_x / mag) * max_force;
                    coh_y = (coh_y / mag) * max_force;
                    coh_z = (coh_z / mag) * max_force;
                }
                cohf[3 * i] = coh_x * cohesion_weight;
                cohf[3 * i + 1] = coh_y * cohesion_weight;
                cohf[3 * i + 2] = coh_z * cohesion_weight;
            }
            avg_colors[3 * i] = (col_r + col[3 * i]) / (neighbor_count + 1);
            avg_colors[3 * i + 1] = (col_g + col[3 * i + 1]) / (neighbor_count + 1);
            avg_colors[3 * i + 2] = (col_b + col[3 * i + 2]) / (neighbor_count + 1);
        }
    }
}

void bdref_update_physics(double *pos, double *vel, double *col, const double *sepf, const double *alif,
                          const double *cohf, const double *avg_colors, double bounds, double margin,
                          double wall_force, double max_speed, double color_blend, double dt, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double a[3];
        for (int d = 0; d < 3; d++) a[d] = sepf[3 * i + d] + alif[3 * i + d] + cohf[3 * i + d];
        for (int d = 0; d < 3; d++) {
            double p = pos[3 * i + d];
            double dist_pos = p - (bounds - margin);
            if (dist_pos > 0) {
                double strength = fmin(dist_pos / margin * 2.0, 1.0);
                a[d] -= strength * wall_force;
            }
            double dist_neg = (-bounds + margin) - p;
            if (dist_neg > 0) {
                double strength = fmin(dist_neg / margin * 2.0, 1.0);
                a[d] += strength * wall_force;
            }
        }
        double *v = vel + 3 * i;
        v[0] += a[0] * dt; v[1] += a[1] * dt; v[2] += a[2] * dt;
        double speed = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (speed > max_speed) {
            double scale = max_speed / speed;
            v[0] *= scale; v[1] *= scale; v[2] *= scale;
        }
        for (int d = 0; d < 3; d++) {
            pos[3 * i + d] += v[d] * dt;
            col[3 * i + d] += (avg_colors[3 * i + d] - col[3 * i + d]) * color_blend;
        }
    }
}

/* params: [bounds, wall_margin, wall_weight, max_speed, max_force, perception_radius,
 *          separation_radius, separation_weight, alignment_weight, cohesion_weight,
 *          color_blend_rate]  (config/boids.py:30-46)
 * Flock.update (flock.py:627-678).  If sorted_in != NULL it is used as the argsort result.
 * work arrays: cell_indices[n], sorted_indices[n], cell_starts[C], cell_counts[C],
 * f[4][3n] (sep, align, coh, avg). */
void bdref_step(double *pos, double *vel, double *col, int64_t n, const double *params, double dt,
                const int32_t *sorted_in, int32_t *cell_indices, int32_t *sorted_indices,
                int32_t *cell_starts, int32_t *cell_counts, double *f) {
    double bounds = params[0], margin = params[1], wall_weight = params[2], max_speed = params[3],
           max_force = params[4], perception = params[5], sep_r = params[6], sep_w = params[7],
           ali_w = params[8], coh_w = params[9], blend_rate = params[10];
    double cell_size = perception;
    int grid_dim = (int)ceil(bounds * 2 / cell_size) + 2; /* flock.py:479 */
    int64_t num_cells = (int64_t)grid_dim * grid_dim * grid_dim;
    double offset = bounds + cell_size;
    bdref_assign_cells(pos, cell_indices, cell_size, grid_dim, offset, n);
    if (sorted_in) memcpy(sorted_indices, sorted_in, (size_t)n * sizeof(int32_t));
    else bdref_sort_by_cell(cell_indices, sorted_indices, n, num_cells);
    bdref_build_cell_lists(cell_indices, sorted_indices, cell_starts, cell_counts, n, num_cells);
    double *sepf = f, *alif = f + 3 * n, *cohf = f + 6 * n, *avg = f + 9 * n;
    memset(f, 0, (size_t)9 * n * sizeof(double));
    memcpy(avg, col, (size_t)3 * n * sizeof(double));
    bdref_compute_flocking_spatial(pos, vel, col, sorted_indices, cell_starts, cell_counts, sepf, alif, cohf,
                                   avg, cell_size, grid_dim, offset, perception, sep_r, sep_w, ali_w, coh_w,
                                   max_speed, max_force, n);
    double blend = fmin(1.0, blend_rate * dt);
    bdref_update_physics(pos, vel, col, sepf, alif, cohf, avg, bounds, margin, max_force * wall_weight,
                         max_speed, blend, dt, n);
}

/* ---------------------------------------------------------------------------------------------
 * Render-side reductions.
 * bdref_visibility      flock.py:311-348  compute_visibility_numba (z < 0.5 or z > fog_end hidden,
 *                                         no margin factor)
 * bdref_build_vertices  flock.py:351-447  build_vertices_numba: 6 float32 vertices (two triangles:
 *                                         tip/right/left, tip/up/down) + 6 colours per visible boid
 * cam = {pos[3], forward[3], right[3], up[3]}.
 * ------------------------------------------------------------------------------------------- */
void bdref_visibility(const double *pos, const double *cam, double tan_h, double tan_v, double fog_end,
                      uint8_t *visible_mask, int64_t n) {
    const double *cp = cam, *cf = cam + 3, *cr = cam + 6, *cu = cam + 9;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const double dx = pos[3 * i] - cp[0], dy = pos[3 * i + 1] - cp[1], dz = pos[3 * i + 2] - cp[2];
        const double z = dx * cf[0] + dy * cf[1] + dz * cf[2];
        if (z < 0.5 || z > fog_end) {
            visible_mask[i] = 0;
            continue;
        }
        const double x = dx * cr[0] + dy * cr[1] + dz * cr[2];
        const double y = dx * cu[0] + dy * cu[1] + dz * cu[2];
        const double half_width = z * tan_h;
        const double half_height = z * tan_v;
        visible_mask[i] = (fabs(x) < half_width && fabs(y) < half_height) ? 1 : 0;
    }
}

void bdref_build_vertices(const double *pos, const double *vel, const double *col, const int32_t *visible_indices,
                          float *vertices, float *vert_colors, double cone_length, double cone_radius,
                          int64_t num_visible) {
    const double wux = 0.0, wuy = 1.0, wuz = 0.0;  /* world up */
    const double wrx = 1.0, wry = 0.0, wrz = 0.0;  /* world right */
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < num_visible; idx++) {
        const int64_t i = visible_indices[idx];
        const double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
        const double vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
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
        if (r_len > 0.0001) {
            rx /= r_len; ry /= r_len; rz /= r_len;
        }
        const double ux = ry * fz - rz * fy;
        const double uy = rz * fx - rx * fz;
        const double uz = rx * fy - ry * fx;
        const double r = cone_radius;
        const double v[6][3] = {
            {px + fx * cone_length, py + fy * cone_length, pz + fz * cone_length},
            {px + rx * r, py + ry * r, pz + rz * r},
            {px - rx * r, py - ry * r, pz - rz * r},
            {px + fx * cone_length, py + fy * cone_length, pz + fz * cone_length},
            {px + ux * r, py + uy * r, pz + uz * r},
            {px - ux * r, py - uy * r, pz - uz * r},
        };
        float *o = vertices + 18 * idx, *c = vert_colors + 18 * idx;
        for (int k = 0; k < 6; k++) {
            o[3 * k] = (float)v[k][0]; o[3 * k + 1] = (float)v[k][1]; o[3 * k + 2] = (float)v[k][2];
            c[3 * k] = (float)col[3 * i]; c[3 * k + 1] = (float)col[3 * i + 1]; c[3 * k + 2] = (float)col[3 * i + 2];
        }
    }
}
