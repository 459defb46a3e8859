/*
 * nbmi.h - C ABI of the MI355X-native N-body backend (libnbmi.so).
 *
 * This is the drop-in boundary for the reference's N-body *backend protocol*: the duck-typed
 * object returned by create_gpu_simulation() (reference nbody/gpu_backend.py:623-679) whose
 * methods are step / compute_colors / get_positions / get_velocities / get_colors / sync
 * (reference class CUDASimulation, nbody/gpu_backend.py:336-409; Metal twin
 * nbody/metal/metal_backend.py:246, 487-599).  One entry point per protocol method, plain
 * pointers and sizes only.  All functions return 0 on success and a negative code on failure;
 * nbmi_last_error() gives the message (thread-local).  A handle is used from one host thread
 * at a time and owns one HIP stream.
 *
 * Body arrays crossing the boundary are the reference's layouts: positions / velocities are
 * C-order (N,3), masses (N,), float64 in (as the reference constructors take them,
 * gpu_backend.py:339-356), float32 positions / colours and float64 velocities out
 * (gpu_backend.py:394-404).  Rows are always in the caller's original body order.
 */
#ifndef NBMI_H
#define NBMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbmi_sim nbmi_sim;

#define NBMI_METHOD_BARNES_HUT 0 /* octree build + tree walk: nbody/simulation.py:63-305        */
#define NBMI_METHOD_DIRECT 1     /* all-pairs O(N^2): nbody/gpu_backend.py:145-257               */

#define NBMI_OK 0
#define NBMI_ERR_ARG (-1)
#define NBMI_ERR_HIP (-2)
#define NBMI_ERR_NODEV (-3)
#define NBMI_ERR_CAPACITY (-4) /* octree needs more node rows than allocated (4N, at most 178 M) */

/* Number of visible HIP devices (0 if none / no driver).  Replaces the probe in
 * detect_backend()/_check_cuda(), nbody/gpu_backend.py:36-70. */
int nbmi_device_count(void);

/* Message of the last failing call on this thread ("" if none). */
const char *nbmi_last_error(void);

/* Constructor: CUDASimulation.__init__ (gpu_backend.py:339-366) /
 * MetalBarnesHutSimulation.__init__(…, theta) (metal_backend.py:252-254).  Copies the three
 * host arrays to the device; the caller keeps ownership.  Returns NULL on failure. */
nbmi_sim *nbmi_create(int64_t n, const double *positions_xyz, const double *velocities_xyz,
                      const double *masses, double G, double softening, double damping,
                      double theta, int method, int device);

/* Constructor with DEVICE-SIDE initial conditions: the bodies are drawn on the
 * GPU from generate_distribution's formulas (tools/presets.py:104-232 "galaxy" / "collision",
 * :350-397 "cluster"; unit masses) with a counter-based Philox4x32-10 stream keyed by `seed`, so a
 * 10 M-body start needs no host generator and no 640 MB upload.  Statistical, not bit, parity with
 * the NumPy generator; the same (seed, n, radius, G) always gives the same bodies.  G is used both
 * for the rotation curve / dispersions and for the simulation, as record() does
 * (tools/record.py:747-758).  Body i of the getters is the i-th generated body. */
#define NBMI_IC_GALAXY 0
#define NBMI_IC_COLLISION 1
#define NBMI_IC_CLUSTER 2
nbmi_sim *nbmi_create_generated(int distribution, int64_t n, double spawn_radius, uint64_t seed, double G,
                                double softening, double damping, double theta, int method, int device);
/* The generator's random function, computed on the host (known-answer tests). */
void nbmi_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* Masses (N,) float64 in the caller's order (state checkpoints of generated systems). */
int nbmi_get_masses_f64(nbmi_sim *sim, double *out);

void nbmi_destroy(nbmi_sim *sim);

/* step(dt): gpu_backend.py:368-386.  `substeps` consecutive steps of size dt are enqueued on
 * the handle's stream without host synchronisation (record() calls step() `substeps` times per
 * frame, tools/record.py:823-824).  Barnes-Hut: bounds -> keys -> sort -> octree -> walk with the
 * reference's kick-drift update fused in (simulation.py:308-317, 63-198, 201-278, 281-305). */
int nbmi_step(nbmi_sim *sim, double dt, int substeps);
/* Number of steps enqueued on this handle since it was created (every substep of nbmi_step counts).  The
 * recorder's Ctrl-C path asks the library, not its own bookkeeping, which frame the device has reached: an
 * interrupt is delivered when nbmi_step returns, before the caller can note that the step was taken
 * (tools/record.py:916-935 writes "state at the frame the device is at"). */
int64_t nbmi_step_count(nbmi_sim *sim);

/* compute_colors(max_speed): gpu_backend.py:388-392 (ramp of simulation.py:320-400). */
int nbmi_compute_colors(nbmi_sim *sim, double max_speed);

/* get_positions() -> (N,3) float32 (gpu_backend.py:394-396). */
int nbmi_get_positions_f32(nbmi_sim *sim, float *out_xyz);
/* get_velocities() -> (N,3) float64 (gpu_backend.py:398-400). */
int nbmi_get_velocities_f64(nbmi_sim *sim, double *out_xyz);
/* get_colors() -> (N,3) float32 (gpu_backend.py:402-404). */
int nbmi_get_colors_f32(nbmi_sim *sim, float *out_rgb);
/* sync(): gpu_backend.py:406-409.  Also reports deferred device-side errors.  NBMI_ERR_CAPACITY is
 * sticky on the device: from the substep whose octree did not fit, the bodies are no longer advanced
 * (they stay at the last completed step) until a sync / getter has reported the error once. */
int nbmi_sync(nbmi_sim *sim);

/* ---- supersets of the protocol (parity / measurement hooks) --------------------------- */

/* Full-precision state (float64 master copy kept on the device). */
int nbmi_get_positions_f64(nbmi_sim *sim, double *out_xyz);
/* Replace positions+velocities (resume from a state_%04d.npz, tools/record.py:728-733). */
int nbmi_set_state(nbmi_sim *sim, const double *positions_xyz, const double *velocities_xyz);

/* Build the octree for the CURRENT positions without advancing time (bounds, keys, sort,
 * node emission).  After it the tree queries below describe that tree. */
int nbmi_build_tree(nbmi_sim *sim);
/* Accelerations (N,3) float64 of the current positions (build + walk, no integration):
 * compute_forces_barnes_hut / compute_forces_*_cuda output. */
int nbmi_get_accelerations_f64(nbmi_sim *sim, double *out_xyz);
/* num_nodes as build_octree returns it (simulation.py:198), deepest level, root half-size
 * (compute_bounds, simulation.py:317) of the most recently built tree. */
int nbmi_tree_stats(nbmi_sim *sim, int64_t *num_nodes, int32_t *max_depth, double *bounds);
/* Octant-path keys of every body for the most recently built tree, original body order:
 * key_hi = levels 1..21 (3 bits per level, level 1 most significant, digit =
 * x>=cx | (y>=cy)<<1 | (z>=cz)<<2 as get_octant, simulation.py:38-49), key_lo = levels 22..42. */
int nbmi_get_keys(nbmi_sim *sim, uint64_t *key_hi, uint64_t *key_lo);
/* The keys the device actually sorts by, same layout: the octant digits relabelled along the 3-D Hilbert curve
 * (csrc/hilbert.h: the digit of a child depends on its octant and on the orientation of its cell; a bijection per
 * cell, so two bodies share exactly as many leading digits as with the octant digits, and nbmi_get_keys is the
 * decoded form of these).  Only the order of a cell's eight children differs from the octant order. */
int nbmi_get_sort_keys(nbmi_sim *sim, uint64_t *key_hi, uint64_t *key_lo);
/* Body indices in sort-key (octree DFS, children along the Hilbert curve) order for the most recently built tree:
 * order[r] = index, in the caller's numbering, of the r-th body along that order. */
int nbmi_get_order(nbmi_sim *sim, int32_t *order);
/* (level, path key) of every node of the most recently built tree (num_nodes entries,
 * unspecified order).  Keys of levels > 21 are reported as UINT64_MAX. */
int nbmi_get_cells(nbmi_sim *sim, int32_t *level, uint64_t *key, int64_t capacity);
/* Per-phase device time in ms accumulated since the last reset:
 * [bounds+keys, sort, tree build, walk+integrate, other]; count = steps accumulated.
 * Enabling timers adds hipEvent records to every step. */
int nbmi_enable_timers(nbmi_sim *sim, int enable);
int nbmi_get_timers(nbmi_sim *sim, double *ms5, int64_t *count, int reset);
/* Work counters of the last counted walk (nbmi_get_accelerations_f64): [wave-level node visits,
 * lane-level visits, lane accepts, node-window misses for windows of 8/16/32/64 nodes,
 * non-sequential cursor moves, wave-level visits executed on each of the 8 XCDs, lane visits whose
 * opening test was a near-tie in fp32 and was re-decided in float64 like the reference
 * (simulation.py:252-258)] (17 values). */
int nbmi_walk_counters(nbmi_sim *sim, int64_t *out17);

/* Multi-GPU (one process per GPU).  A handle created with nbmi_create holds ALL bodies; with a
 * shard set, step() integrates only the key-sorted ranks [begin,end) (direct method: the body
 * indices [begin,end), its state is never re-ordered) and leaves the others untouched until
 * nbmi_import_ranks() supplies them.  Packed row = 8 doubles
 * {x,y,z,vx,vy,vz,m,id}.  Pointers are DEVICE pointers (e.g. torch tensors' data_ptr()) so the
 * exchange itself can be an RCCL all-gather issued by the host framework.  Results equal the unsharded
 * handle's bit for bit when `begin` is a multiple of 64 (a wave's 64 bodies, and with them the order of its
 * fp32 sums, are then the same however the ranks are cut; nbody/sharded.py::shard_bounds does that).  The rows carry
 * every body's OWN mass: masses are fixed at creation (a direct-N^2 handle whose bodies all have the same mass takes
 * G m out of its pair loop, decided once at nbmi_create). */
int nbmi_set_shard(nbmi_sim *sim, int64_t begin, int64_t end);
int nbmi_export_shard(nbmi_sim *sim, void *dev_rows);                              /* (end-begin, 8) f64 */
int nbmi_import_ranks(nbmi_sim *sim, const void *dev_rows, int64_t begin, int64_t end);

/* Multi-GPU stage 2, "owner mode" (BASELINE north_star: bodies shard by Morton range, exchange of the upper /
 * locally essential octree cells).  A rank's handle holds only the bodies whose octant key falls into the
 * rank's key range (rebalanced every step), builds the octree of THOSE bodies inside the GLOBAL root cube
 * (compute_bounds over the whole system, simulation.py:308-317) and walks its own tree plus, behind it in the
 * same node array, the part of every other rank's tree that some body of this rank can open (the other ranks
 * prune their trees against this rank's bounding boxes with the reference's own opening test, made conservative
 * by 1e-9).  Per-rank sort / build / memory no longer grow with the number of ranks.  [r3] The ranks' trees are
 * pieces of ONE global octree: every rank publishes a small table about its first and last bodies
 * (nbmi_owner_chain_doubles() doubles, written by nbmi_owner_adopt, all-gathered like the boxes), and
 * nbmi_owner_export_let first turns the own tree into the rank's piece of the global pre-order node array - cells
 * that reach into higher ranks get the global moments, cells that exist only across a boundary are inserted, the
 * copies of cells that begin on a lower rank are dropped.  nbmi_owner_step puts the received pieces around the own
 * one in rank order.  Every body then visits the single-GPU run's accepted nodes in the single-GPU order: with
 * float64 forces the result equals the single handle's to rounding (tests: 1e-13 after 4 steps), in the default
 * "auto" precision 1 M bodies on 8 ranks stay within 1.4e-5 of the float64 reference after 100 steps.
 * The replicated-tree exchange above (nbmi_set_shard) stays as the bit-for-bit mode.
 *
 * One step, host side (buffers are DEVICE pointers; the collectives are the host framework's):
 *   nbmi_owner_maxabs(h, m)                         m <- max |coordinate| of the owned bodies (1 double)
 *        all-reduce MAX of m
 *   nbmi_owner_sample(h, m, samples, S, V)          keys of the owned bodies in the global cube; V <= S regular samples
 *                                                   (the other slots all ones = ignored).  V in proportion to the rank's
 *                                                   body count, e.g. 0.8 S n_rank world / n_total, keeps the ranks balanced
 *        all-gather of the samples                  (world x S keys)
 *   nbmi_owner_partition(h, all, world*S, send, counts)   splitters at equal quantiles; rows {x,y,z,vx,vy,vz,m,id} of
 *                                                   the bodies that now belong to ANOTHER rank, grouped by destination
 *                                                   in `send`; counts[world] on the host (counts[own rank] = 0)
 *        all-to-all of the counts, all-to-all-v of the rows  (only bodies that crossed a splitter travel)
 *   nbmi_owner_adopt(h, recv, n_recv, m, box, chain)   the n_recv received rows join the bodies that stayed: keys, sort, octree;
 *                                                   box <- B = nbmi_owner_boxes_per_rank() bounding boxes (6 doubles
 *                                                   each: lo xyz, hi xyz) of the bodies inside cells of the own
 *                                                   tree (level 4, refined to level 11 along the two boundary
 *                                                   chains), in key order; unused boxes are empty (lo > hi)
 *                                                   chain <- the rank's boundary table (nbmi_owner_chain_doubles() doubles)
 *        all-gather of the boxes, all-gather of the tables   (world x B x 6 doubles, world x C doubles)
 *   nbmi_owner_export_let(h, boxes, chains, let, counts)   the own tree becomes the rank's piece of the global tree, then is
 *                                                   pruned against EACH other rank's boxes: counts[j] rows
 *                                                   for rank j, packed one destination after the other in `let`
 *                                                   (rows of nbmi_owner_let_row_bytes() = 48 bytes, float64
 *                                                   throughout: {cx, cy, cz, G m, float s2t, link, node index, level};
 *                                                   let_capacity rows in all); counts[world] on the host
 *        all-to-all of the counts, all-to-all-v of the rows
 *   nbmi_owner_step(h, recv, recv_counts, dt)       put the received pieces (packed in rank order, at most
 *                                                   let_capacity rows) around the own one, walk, kick-drift
 *
 * Host waits [r3]: with nbmi_set_exchange_sync(h, 0) and the collectives enqueued ON the handle's stream
 * (nbmi_stream; torch.cuda.ExternalStream), nbmi_owner_maxabs / _sample / _adopt / _step only enqueue; the two
 * calls that hand counts to the host wait once each (nbmi_owner_partition, nbmi_owner_export_let).  With the
 * default sync = 1 every call waits for its own work, as in round 2.  World size 1 skips the splitter, dead-row
 * and box work altogether.
 *
 * Force precision [r3]: an owner handle computes forces like a plain one (nbmi_set_force_precision; default
 * "auto") - the exchanged rows carry the float64 centre of mass and G m of every node, and the receiver builds
 * both walk records from them.  "auto" is decided while nbmi_owner_adopt builds the tree, so the dt of the
 * step has to be known by then: nbmi_owner_set_dt(h, dt) before nbmi_owner_adopt (no dt set: fp32 forces, as
 * "auto" does for any build that is not part of a step).  Each rank decides for its own waves by their density;
 * the "most of the system asks => every wave" half of the rule is taken SYSTEM-WIDE [r4]: after
 * nbmi_owner_export_let, nbmi_owner_step_facts gives the rank's votes (asking waves, waves), the caller sums them
 * over the ranks (they ride in the exchange of the tree counts), applies the single handle's rule (enter above a third,
 * leave below a quarter) and hands the verdict to every rank with nbmi_owner_set_all64 before nbmi_owner_step - the
 * arithmetic no longer depends on the world size or on where the splitters fall.  Without a verdict (-1, default) a
 * rank applies the rule to its own waves.
 *
 * Getters of an owner handle return the owned bodies in their current (key) order; nbmi_owner_get_ids gives the
 * global body ids of those rows. */
nbmi_sim *nbmi_create_owner(int64_t n, const double *positions_xyz, const double *velocities_xyz, const double *masses,
                            const int32_t *global_ids, int64_t capacity, int64_t let_capacity, int world, int rank,
                            double G, double softening, double damping, double theta, int device);
int64_t nbmi_owner_count(nbmi_sim *sim);
int nbmi_owner_boxes_per_rank(void);
int nbmi_owner_let_row_bytes(void);
int nbmi_owner_set_dt(nbmi_sim *sim, double dt);
int nbmi_owner_get_ids(nbmi_sim *sim, int32_t *out);
int nbmi_owner_maxabs(nbmi_sim *sim, void *dev_maxabs);
int nbmi_owner_sample(nbmi_sim *sim, const void *dev_maxabs, void *dev_samples, int nsamples, int nvalid);
int nbmi_owner_partition(nbmi_sim *sim, const void *dev_all_samples, int total_samples, void *dev_send_rows,
                         int64_t *counts_host);
int nbmi_owner_chain_doubles(void);
int nbmi_owner_adopt(nbmi_sim *sim, const void *dev_recv_rows, int64_t n_recv, const void *dev_maxabs, void *dev_boxes,
                     void *dev_chain);
int nbmi_owner_export_let(nbmi_sim *sim, const void *dev_boxes, const void *dev_chains, void *dev_let, int64_t *counts_host);
int nbmi_owner_step(nbmi_sim *sim, const void *dev_recv_let, const int64_t *recv_counts_host, double dt);
/* After nbmi_owner_export_let: out4 = {waves of this rank whose own density asks for float64 forces, waves of this
 * rank (both 0 unless the mode is "auto" and a dt is set), tree rows that still fit in front of the own piece of the
 * walk array, tree rows that fit behind it} - what every rank needs to know of every other rank to take the
 * system-wide precision decision and to evaluate nbmi_owner_step's "received trees do not fit" for all ranks
 * together (a rank raising alone leaves the others in the next collective). */
int nbmi_owner_step_facts(nbmi_sim *sim, int64_t *out4_host);
/* verdict 1 / 0: the next nbmi_owner_step computes every wave's forces in float64 / leaves the choice to the waves;
 * -1: back to the rank's own rule. */
int nbmi_owner_set_all64(nbmi_sim *sim, int verdict);

/* Render-side reduction: NBodySimulation._compute_visibility + the gather of
 * draw() on the device (nbody/simulation.py:880-903, 927-928).  Frustum test of
 * compute_visibility_points (:403-434; z < 0.1 or z > far_dist hidden, 20 % margin) on the float64
 * device positions, then positions[mask].astype(float32) and colors[mask] (colours of the last
 * nbmi_compute_colors) in the caller's body order.  cam12 = {cam_pos, cam_forward, cam_right,
 * cam_up}.  *count = visible bodies; at most `capacity` rows are copied out.  Only the visible
 * part crosses PCIe. */
int nbmi_visible_points(nbmi_sim *sim, const double *cam12, double tan_h, double tan_v, double far_dist,
                        float *out_positions_xyz, float *out_colors_rgb, int64_t capacity, int64_t *count);
/* nbmi_export_shard / nbmi_import_ranks and the nbmi_owner_* calls synchronise the handle's stream by default.
 * With sync = 0 they only enqueue (except where a call returns counts to the host): for callers that issue
 * their collective ON the handle's stream (nbmi_stream; e.g. torch.cuda.ExternalStream). */
int nbmi_set_exchange_sync(nbmi_sim *sim, int sync);

/* Arithmetic of the Barnes-Hut pair forces (the accepted (body, node) sets are the reference's in every mode):
 *   0  per wave of 64 key-adjacent bodies: float64 where G rho dt^2 of the wave's densest quarter (16 bodies: sum of
 *      G m over the volume of their bounding box, every edge at least one softening length) exceeds `tau`
 *      (default 5e-5; tau = 0 keeps the current value), fp32 elsewhere - and float64 for EVERY wave while most of
 *      the system qualifies (entered when more than a third of a step's waves do, left below a quarter).  Default.  The reference computes in float64 throughout
 *      (nbody/simulation.py:246-268); in the dense part of a system fp32's systematic roundings are amplified to
 *      > 1e-4 of the largest coordinate within 100 steps, and where most of the system is that dense the rest
 *      follows.
 *   1  fp32 everywhere (float64 sums): fastest.
 *   2  float64 everywhere: follows the reference to ~1e-13 over 100 steps at 1 M bodies.
 * Environment NBMI_FORCE_PREC / NBMI_PREC_TAU set the initial values. */
int nbmi_set_force_precision(nbmi_sim *sim, int mode, double tau);
/* Mode 0, after a step: the share of the waves whose own density asked for float64, and whether the step ran every
 * wave in float64 (the system-wide rule above: entered when more than a third asked, left below a quarter). */
int nbmi_force_precision_share(nbmi_sim *sim, double *share, int *all_float64);
/* Native HIP stream of the handle (for ordering against framework streams). */
void *nbmi_stream(nbmi_sim *sim);

/* Frame codec on the device (tools/record.py:231-326).  A recording's .zstd frame holds either
 * absolute float32 positions + colours (format 1) or int16((cur - prev) * 1000) against the previous DECODED
 * frame (format 2, :254-262, decoder :313-322; lossy, wraps beyond +-32.767 like the reference's cast).  The
 * previous decoded frame stays in HBM, the quantisation runs on the device, and a delta frame costs 12 bytes per
 * body over PCIe instead of 24; zstd itself stays on the host.
 *   nbmi_frame_keyframe      current positions (as nbmi_get_positions_f32) and colours (of the last
 *                            nbmi_compute_colors), float32 (N,3) each; they become the previous frame
 *   nbmi_frame_delta_i16     int16 (N,3) position and colour deltas against the previous decoded frame, which is
 *                            advanced to prev + int16 / 1000 (what load_frame() will reconstruct)
 *   nbmi_frame_set_previous  restore the previous decoded frame after a resume (decoded by the host codec) */
int nbmi_frame_keyframe(nbmi_sim *sim, float *out_positions_xyz, float *out_colors_rgb);
int nbmi_frame_delta_i16(nbmi_sim *sim, int16_t *out_dpos, int16_t *out_dcol);
int nbmi_frame_set_previous(nbmi_sim *sim, const float *positions_xyz, const float *colors_rgb);

/* Test / measurement hook for the device sort behind the octree build ("Morton-code octree build via
 * device radix sort"; it replaces np.argsort of boids/flock.py:618 as well): sorts n (key, value) pairs
 * given as HOST arrays by the low `bits` bits of the key (key_bytes 4 or 8), stable.  impl must be 0 = the
 * hand-written gfx950 radix sort of csrc/radix.hip, the product path (the rocPRIM cross-check is a library of
 * the tests' own since round 4: tests/native/rocprim_check.hip).
 * *ms_per_sort = mean device time of `repeats` sorts after one untimed run. */
int nbmi_debug_sort_pairs(int key_bytes, int64_t n, const void *keys, const uint32_t *values, void *keys_out,
                          uint32_t *values_out, int bits, int impl, int repeats, double *ms_per_sort);

#ifdef __cplusplus
}
#endif
#endif /* NBMI_H */
