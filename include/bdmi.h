/*
 * bdmi.h - C ABI of the MI355X-native boids neighbour sweep (part of libnbmi.so).
 *
 * The reference has no backend seam for boids; its operator boundary is the five flat-array
 * @njit kernels driven by Flock.update (reference boids/flock.py:627-678):
 *   assign_cells              boids/flock.py:30-44   (get_cell_index :16-27)
 *   np.argsort + build_cell_lists   boids/flock.py:610-625, :47-65
 *   compute_flocking_spatial  boids/flock.py:68-238
 *   update_physics_numba      boids/flock.py:241-308
 * bdmi_step() is one Flock.update(dt): all four stages on the device, float64 state and
 * arithmetic as in the reference, state resident in HBM between steps.
 * Arrays crossing the boundary are C-order (N,3) float64, rows in the caller's boid order.
 * Return codes / error string as in nbmi.h (bdmi_last_error == nbmi_last_error).
 */
#ifndef BDMI_H
#define BDMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bdmi_flock bdmi_flock;

/* params[11] in the order of reference config/boids.py:30-46:
 * bounds, wall_margin, wall_weight, max_speed, max_force, perception_radius,
 * separation_radius, separation_weight, alignment_weight, cohesion_weight, color_blend_rate.
 * Grid as Flock.__init__ (flock.py:478-481): cell = perception_radius,
 * dim = ceil(2*bounds/cell) + 2, offset = bounds + cell. */
bdmi_flock *bdmi_create(int64_t n, const double *positions_xyz, const double *velocities_xyz,
                        const double *colors_rgb, const double *params11, int device);
void bdmi_destroy(bdmi_flock *f);
const char *bdmi_last_error(void);

/* `substeps` x Flock.update(dt) (flock.py:627-678), enqueued without host synchronisation. */
int bdmi_step(bdmi_flock *f, double dt, int substeps);
int bdmi_sync(bdmi_flock *f);

/* positions / velocities / colors (N,3) float64 each; any pointer may be NULL. */
int bdmi_get_state(bdmi_flock *f, double *positions_xyz, double *velocities_xyz, double *colors_rgb);
int bdmi_set_state(bdmi_flock *f, const double *positions_xyz, const double *velocities_xyz,
                   const double *colors_rgb);

/* ---- parity hooks ------------------------------------------------------------------- */
/* assign_cells output for the current positions: (N,) int32, caller's boid order. */
int bdmi_get_cell_indices(bdmi_flock *f, int32_t *cell_indices);
/* compute_flocking_spatial outputs for the current state (no physics applied):
 * separation / alignment / cohesion forces and avg_colors, (N,3) float64 each, with the
 * caller-side pre-fill of Flock.update (forces 0, avg_colors = colors; flock.py:633-636). */
int bdmi_get_forces(bdmi_flock *f, double *sep, double *ali, double *coh, double *avg_colors);
/* grid facts: dim, num_cells, non-empty cells of the last built grid. */
int bdmi_grid_info(bdmi_flock *f, int32_t *grid_dim, int64_t *num_cells, int64_t *occupied);
/* device ms accumulated per phase [cells+sort, reorder+table, sweep+physics], steps counted */
int bdmi_enable_timers(bdmi_flock *f, int enable);
int bdmi_get_timers(bdmi_flock *f, double *ms3, int64_t *count, int reset);


/* ---- multi-GPU: x-slabs with a one-cell halo -----------------------------------------
 * A rank owns the boids with x_lo <= x < x_hi (a whole number of cell planes, at least two cells wide); the
 * neighbour sweep needs the boids within one cell (= the perception radius, flock.py:479) on the other side
 * of each face.  The handle holds the owned boids plus this step's ghosts (read-only copies of the
 * neighbours' boundary boids).  A boid moves at most max_speed * dt << one cell per step, so it can only
 * ever change to an adjacent slab, and ONE exchange per step carries migrants and halo alike:
 *
 *   bdmi_slab_export(h, left, &nl, right, &nr)   drop last step's ghosts; rows {p, v, c, global id} (10 doubles)
 *                                                of the owned boids with x < x_lo + cell -> `left`, with
 *                                                x >= x_hi - cell -> `right` (device buffers, counts on the
 *                                                host); owned boids now outside the slab stay as ghosts
 *        exchange with the two neighbours        (send/recv or all-to-all-v on the host framework)
 *   bdmi_slab_import(h, rows, count)             received rows: inside the slab -> owned, else ghosts
 *   bdmi_step(h, dt, 1)                          Flock.update for the owned boids; ghosts only take part as
 *                                                neighbours
 *
 * The candidate set of every owned boid equals the single-GPU one, so forces agree to float64 summation
 * order.  bdmi_slab_get returns the owned rows (10 doubles each, any order). */
bdmi_flock *bdmi_create_slab(int64_t n, const double *positions_xyz, const double *velocities_xyz, const double *colors_rgb,
                             const int32_t *global_ids, int64_t capacity, const double *params11, double x_lo, double x_hi,
                             int has_left_neighbour, int has_right_neighbour, int device);
int64_t bdmi_slab_count(bdmi_flock *f);
int bdmi_slab_export(bdmi_flock *f, void *dev_left_rows, int64_t *n_left, void *dev_right_rows, int64_t *n_right);
int bdmi_slab_import(bdmi_flock *f, const void *dev_rows, int64_t count);
int bdmi_slab_get(bdmi_flock *f, double *rows10, int64_t capacity, int64_t *count);

/* ---- render-side reduction ------------------------------------------ */
/* Flock._compute_visibility + _build_vertices on the device (flock.py:680-728): frustum test of
 * compute_visibility_numba (:311-348; z < 0.5 or z > fog_end hidden), np.where order (ascending
 * boid index), then build_vertices_numba (:351-447): 6 float32 vertices + 6 float32 colours per
 * visible boid.  cam12 = {cam_pos, cam_forward, cam_right, cam_up}; tan_h / tan_v as the caller
 * derives them from (fov, aspect, fov_margin).  *count = visible boids; at most capacity_boids of
 * them are copied out ((count*6, 3) rows each).  Only the visible part crosses PCIe. */
int bdmi_visible_vertices(bdmi_flock *f, const double *cam12, double tan_h, double tan_v, double fog_end,
                          double cone_length, double cone_radius, float *out_vertices, float *out_colors,
                          int64_t capacity_boids, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* BDMI_H */
