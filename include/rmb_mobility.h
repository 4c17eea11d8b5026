/*
 * rmb_mobility.h -- C ABI of the MI355X blob-mobility engine (librmb_mobility.so).
 *
 * This is the drop-in boundary for the reference's blob-level pairwise operators.  The reference
 * reaches its GPU backend through Python wrappers that allocate, upload, launch and download on
 * every call (mobility/mobility_pycuda.py:2235-2267 and siblings; multi_bodies/forces_pycuda.py:
 * 148-180).  Here the same operators are exported as plain-C entry points (pointers + sizes, no
 * torch / numpy types) that a ctypes / cffi / pybind stub can bind; INTEGRATION.md shows the stub.
 *
 * All matrices are never formed: every entry point computes  out = M(kind) . vec  (or the pair
 * forces) by an O(N^2) sweep in fp64 on the device.
 *
 * Conventions
 *   - positions, vectors, outputs: double, C-contiguous (N,3) row-major == flat 3N, exactly the
 *     layout the reference wrappers pass (mobility/mobility_numba.py:132-134).
 *   - `L` = periodic_length[3]; a component <= 0 means open in that direction; > 0 means
 *     pseudo-periodic (nearest image + the 3^d first neighbour boxes, mobility_numba.py:140-197).
 *   - `wall` != 0 selects the single-wall kernels AND the wrapper-level regularisation of the
 *     reference: z_eff = max(z, a) (mobility.py:52-64) and u = B M(z_eff) B v with
 *     B_ii = z_i/a for z_i < a (mobility.py:67-84, :1150-1163).  wall == 0: unbounded RPY, no clamp
 *     (mobility.py:1119-1129).
 *   - every function returns 0 on success, a negative rmb_status otherwise; rmb_last_error()
 *     gives a message for the calling thread.  The reference defines no error codes (failures
 *     surface as Python exceptions from pycuda); the Python shim raises RuntimeError on non-zero.
 *   - host entry points are synchronous (result complete on return), as the reference's are.
 *     *_device entry points enqueue on the context's stream and return immediately.
 */
#ifndef RMB_MOBILITY_H
#define RMB_MOBILITY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmb_ctx rmb_ctx;

enum rmb_status {
  RMB_OK = 0,
  RMB_ERR_ARG = -1,      /* null pointer, bad kind, negative size ...            */
  RMB_ERR_STATE = -2,    /* matvec before set_positions, bad target range ...    */
  RMB_ERR_HIP = -3,      /* a HIP runtime call failed (message has the HIP text) */
  RMB_ERR_NO_DEVICE = -4 /* no gfx950 device visible                             */
};

/* Which block of the grand mobility.  Each row names the reference function it replaces
 * (numba CPU twin / pycuda GPU twin, both in mobility/):
 *   TT     mobility_numba.py:124 (wall) :13 (no wall)   / mobility_pycuda.py:150, :371
 *   TR     mobility_numba.py:548        :440            / mobility_pycuda.py:1516, :1733
 *   RT     mobility_numba.py:938        :832            / mobility_pycuda.py:926, :1034
 *   RR     mobility_numba.py:1189       :1077           / mobility_pycuda.py:593, :703
 *   TT_TR  fused u = M_tt f + M_tr tau, pycuda only     / mobility_pycuda.py:1266, :1394
 *   TT_FREE_SURFACE  u = [RPY(d) + RPY(image) P] f above a stress-free surface at z = 0
 *          mobility_numba.py:1770 / mobility_pycuda.py:2081; needs set_positions(wall = 0)   */
enum rmb_kind { RMB_TT = 0, RMB_TR = 1, RMB_RT = 2, RMB_RR = 3, RMB_TT_TR = 4, RMB_TT_FREE_SURFACE = 5 };

/* ---- library / device ------------------------------------------------------------------- */
const char* rmb_version(void);
const char* rmb_last_error(void);
int rmb_device_count(void);

/* ---- persistent context: positions stay resident across the matvecs of one solve ----------
 * (GMRES / Lanczos call M.f many times with fixed r_vectors, multi_bodies.py:445, :599;
 *  the reference re-uploads positions on every call.) */
/* device: index into the visible devices, or -1 = the default device: the environment variable RMB_DEVICE when it
 * is set (an index; anything else -> RMB_ERR_ARG), else 0. */
int rmb_ctx_create(int device, rmb_ctx** ctx);
int rmb_ctx_destroy(rmb_ctx* ctx);
/* hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream.
 * A context is SINGLE-STREAM at a time (its accumulators, workspaces and packed positions are re-used from call to
 * call): when the handle changes, the new stream is made to wait (event) for everything already queued on the
 * previous one.  Unchanged handle: no cost.
 * LIFETIME: the stream a context is bound to must be alive while calls are enqueued on it AND at the moment the
 * handle is changed (the switch records an event on it; HIP does not validate stream handles, a destroyed one is a
 * use-after-free).  A host that destroys its streams (one stream per time step, say) calls rmb_ctx_release_stream()
 * before hipStreamDestroy: it waits for the context's work on that stream and forgets the handle (the context is then
 * on the default stream until the next rmb_ctx_set_stream).  If the record on the previous stream fails with an error
 * code, the switch falls back to a device-wide synchronisation and adopts the new handle all the same: a context
 * never stays bound to a stream it could not fence. */
int rmb_ctx_set_stream(rmb_ctx* ctx, void* hip_stream);
int rmb_ctx_release_stream(rmb_ctx* ctx);
/* Options (unknown key -> RMB_ERR_ARG).  Defaults in [].
 *   "timing"          [0]  n >= 1 = bracket every n-th pair-sweep launch with HIP events (rmb_timing_collect); an
 *                          event pair serialises ~4-8 us around the launch, so throughput runs sample (n = 4)
 *   "symmetric"       [1]  1 = evaluate each unordered pair once and update both blobs (sym_kernels.h /
 *                          symx_kernels.h) whenever the full target range is resident and n >= 128;
 *                          0 = always the one-sided sweep (every ordered pair, atomic-free)
 *   "deterministic"   [0]  bit-reproducible results (for a given N and device).  1 = the one-sided sweep (every
 *                          ordered pair, fixed summation order, no workspace; ~1.5x the default time);
 *                          2 = the symmetric pass with per-unit partials summed in a fixed order instead of
 *                          atomics (1.06-1.18x; workspace min(0.19 N^2 B, "det_workspace_mb")).  Pair shards
 *                          (nshards > 1) honour 2 -- a shard is then a range of WHOLE tile pairs, so a G-rank run
 *                          is bit-reproducible whenever its all-reduce is -- and ignore 1 (a slice of the unordered
 *                          pairs has no one-sided form).  Forces: 1 and 2 both take the one-sided sweep.
 *   "precision"       [64] 32 = single precision on the default symmetric path with open boundaries (the
 *                          reference's `precision = 'single'` build, mobility_pycuda.py:7-19) for tt / tr / rt / rr,
 *                          RMB_TT_TR, RMB_TT_FREE_SURFACE, the in-plane products, the grand / force-column / k-vector operations and the
 *                          blob-blob forces (the reference's GPU force kernel is always single precision,
 *                          forces_pycuda.py:14-21) (sym32_kernels.h, symx32_kernels.h): pair arithmetic in fp32
 *                          (~1e-6 relative, separations from a head / tail split of the fp64 positions), partial sums,
 *                          self terms and scaling in fp64; 1.5-1.6x faster.
 *                          Pseudo-periodic domains, the one-sided sweep, the deterministic modes,
 *                           the per-blob-radii product with sources != targets, pair shards of tr / rt / rr and
 *                          the source->target operators compute in fp64 whatever this says.  Other values:
 *                          RMB_ERR_ARG.  The "wave_clock" / "skip_pairs" diagnostics exist in the fp64 kernels only:
 *                          a product that would run an fp32 kernel with one of them set returns RMB_ERR_STATE.
 *   "force_cull"      [1]  blob-blob forces (uniform radius, symmetric path; open or pseudo-periodic): skip tile pairs whose
 *                          bounding boxes are further apart than 2a + 750 b, where exp(-(r - 2a)/b) underflows to
 *                          exactly 0 in double precision (110 b for the float kernel): no bit of the result changes
 *   "force_sort"      [1]  with "force_cull", from 2048 blobs on: sort the blobs along a Morton curve once per configuration
 *                          (on the device) and run the force kernel on the sorted copy, results written back to the caller's
 *                          indices -- the culling then skips the same tile pairs whatever order the caller lists the blobs in
 *                          (262 144 rollers listed at random: 48 -> 4 ms).  Same pair terms in another summation order:
 *                          equal to rounding; 0 = keep the caller's order
 *   "force_precision" [0]  blob-blob forces: 0 = follow "precision", 32 / 64 = pinned whatever "precision" says
 *                          (products in single precision with double-precision forces, or the reverse)
 *   "det_workspace_mb" [8192]  workspace of mode 2; the unit list is processed in chunks that fit
 *   "fused_symmetric" [1]  RMB_TT_TR: 1 = one symmetric pass sharing the pair geometry between both blocks,
 *                          2 = two symmetric passes (tt, then tr accumulated), 0 = the one-sided fused sweep
 *   "symx_single"     [0]  1 = run tt / tr / rt / rr (and the two-vector product) through the generic multi-block
 *                          skeleton instead of the dedicated kernels (A/B measurements)
 *   "chunks"          [0]  one-sided sweep: number of source chunks (blockIdx.y); 0 = chosen from the occupancy
 *   "sym_oversub"     [8]  symmetric kernels: launch up to this many times the resident workgroup count
 *   "sym_min_steps"   [64] symmetric kernels: floor on rotation steps per wave (one tile pair = 64 steps)
 *   "sym_fine_steps"  [0]  symmetric kernels: floor on rotation steps per wave when the launch is smaller than one
 *                          resident round (small suspensions, one rank's pair shard); 0 = 16 while 16-step waves do
 *                          not fill the chip, 32 beyond
 *   "sym_coop"        [1]  tt / tr / rt / rr on the dedicated symmetric kernel: the workgroup-cooperative variant
 *                          (the four waves of a workgroup share one staged tile and one flush per tile instead of each
 *                          staging and flushing its own): 0 = never, 1 = launches of at most four resident rounds of
 *                          workgroups (small suspensions, one rank's pair shard, products up to ~1e4 blobs: faster below
 *                          one round, same time with half the atomic flush traffic up to four), 2 = always
 *   "sym_two_targets" [1]  tt / tr / rt / rr with open boundaries: two target blobs per lane (sym2t_kernels.h: a lane keeps
 *                          blob `lane` of two tile rows, so one record read and one set of LDS adds serve two pairs; +3-4.5 %
 *                          from 1e4 to 1e6 blobs, half the atomic flush traffic): 0 = never, 1 = launches of at least half a
 *                          resident round of workgroups (~6000 blobs; smaller ones stay with the cooperative kernel), 2 = always
 *   "sym_order"       [1]  symmetric kernels: order in which the tile pairs are visited: 1 = blocked (super-blocks of 32 x 32
 *                          tiles, so that neighbouring step ranges re-use the same 64 tiles), 0 = row-major over the tile
 *                          triangle.  The deterministic symmetric mode always runs row-major
 *   "sym_xcd"         [1]  symmetric kernels: XCD-aware workgroup numbering -- every XCD (own L2) gets one contiguous eighth
 *                          of the step range; with "sym_order" = 1 the tile loads of a launch mostly hit L2 (HBM fetch
 *                          traffic / 10 at >= 1e5 blobs; the kernels are VALU-bound, so time moves little)
 *   "sym_chunk_steps" [1024]  symmetric kernels: a wave's share of the rotation steps is cut into equal strided chunks of about
 *                          this many steps (wave w takes chunks w, w + W, w + 2 W, ...), so that the waves running at the same
 *                          time work on neighbouring tile pairs at any problem size; 0 = one contiguous range per wave
 *   "gmres_fuse_pc"   [1]  rmb_rigid_gmres_device: the last launch of an Arnoldi step (normalisation, workgroup = body) also applies
 *                          the block-diagonal preconditioner to the vector it has just normalised, so every step but the first
 *                          of a restart cycle is six launches instead of seven; 0 = separate launches (same arithmetic)
 *                          Also governs rmb_rigid_lanczos_device (its normalisation launch applies L_b^-T for the next step)
 *   "gmres_fuse_dots" [1]  rmb_rigid_gmres_device, decks of up to 256 bodies: the operator's finishing launch (workgroup = body) also
 *                          takes the first Gram-Schmidt pass's dots of its slices with every basis row (one partial per body, summed
 *                          by the first update launch): five launches per iteration; 0 = a separate dots launch.  The Lanczos
 *                          step's finishing launch ("lanczos_fuse_finish") does the same
 *   "krylov_low_sync" [1]  steps of the native GMRES / Lanczos entries: the launch that subtracts the second Gram-Schmidt projection
 *                          also normalises, with |w2|^2 = |w1|^2 - |h2|^2 from partials of the first update launch (orthonormal
 *                          basis; h2 is rounding-sized): one launch less per iteration; 0 = separate norm / normalisation launch
 *   "lanczos_fuse_finish" [1]  rmb_rigid_lanczos_step_device / rmb_rigid_lanczos_device: the sweep leaves its raw sums and ONE
 *                          launch (workgroup = body) finishes them and multiplies by L_b^-1; 0 = finalize and block product as
 *                          two launches (same arithmetic)
 *   "sym_wps"         [0]  symmetric kernels: cap on resident workgroups per CU (0 = occupancy limit)
 *   "sym_pin"         [1]  symmetric kernels: pad dynamic LDS so that residency is exactly that number
 *   "wave_clock"      [0]  1 = stamp every wave's start / end (rmb_wave_clock_collect); schedule diagnostics
 *   "skip_pairs"      [0]  diagnostics, results are WRONG: bit 0 = no pair arithmetic, bit 1 = no flush of the
 *                          per-wave LDS accumulators (tools/experiments/exp_prewarm.py prices the atomics with it)          */
int rmb_ctx_set_option(rmb_ctx* ctx, const char* key, long value);
/* Current value of an option (same keys); lets a caller switch one temporarily and restore what was there.  Read-only
 * key "last_path": the kernel family of the last product (0 one-sided sweep, 1 symmetric per wave, 2 deterministic
 * symmetric, 3 symmetric workgroup-cooperative, 4 symmetric with two target blobs per lane). */
int rmb_ctx_get_option(rmb_ctx* ctx, const char* key, long* value);

/* Upload / pack positions: fuses shift_heights + damping_matrix_B (mobility.py:52-84).
 * r: (n,3).  The *_device variant reads a device pointer and is asynchronous. */
int rmb_set_positions(rmb_ctx* ctx, const double* r_host, long n, double a, const double* L, int wall);
int rmb_set_positions_device(rmb_ctx* ctx, const double* r_dev, long n, double a, const double* L, int wall);

/* Multi-GPU: this context only produces targets [begin, end) of the n blobs (all n are sources).
 * Default after set_positions is [0, n).  Output arrays then hold 3*(end-begin) doubles. */
int rmb_set_target_range(rmb_ctx* ctx, long begin, long end);

/* out = M(kind) . vec.  vec2 = torque for RMB_TT_TR, NULL otherwise.  in_plane != 0 gives the
 * in_plane_* variants (mobility_numba.py:291, :690; only meaningful for TT / TR with a wall).
 * vec/vec2: 3n doubles (all sources); out: 3*(end-begin) doubles. */
int rmb_matvec(rmb_ctx* ctx, int kind, int in_plane, const double* vec_host, const double* vec2_host,
               double eta, double* out_host);
int rmb_matvec_device(rmb_ctx* ctx, int kind, int in_plane, const double* vec_dev, const double* vec2_dev,
                      double eta, double* out_dev);

/* Multi-GPU, symmetric pair sharding (RMB_TT / TR / RT / RR / TT_FREE_SURFACE): the unordered blob pairs are cut into
 * `nshards` equal parts; this call evaluates part `shard` (each pair once, applied to both blobs) and
 * writes its contribution to ALL n targets (3n doubles).  The sum over shards is the full product; the
 * self term of target i is added by the shard that owns i in the contiguous block partition.  The
 * caller all-reduces (or reduce-scatters) the outputs.  No reference counterpart (single device). */
int rmb_matvec_pairshard_device(rmb_ctx* ctx, int kind, const double* vec_dev, double eta, double* out_dev,
                                long shard, long nshards);
/* Two source vectors in ONE pass over the pairs (tt only): out_a = M vec_a, out_b = M vec_b.  The vector-independent
 * part of every pair (geometry, both inverse square roots, RPY and wall coefficients) is evaluated once, so the call
 * costs ~0.66 of two rmb_matvec_device calls.  Serves solvers that advance two right-hand sides in lockstep with the
 * same mobility (the Brownian-slip and RFD solves of quaternion_integrator_multi_bodies.py:985-996).  Falls back to two
 * single products where the symmetric kernel does not apply (n < 128, "deterministic", target sub-ranges); a pair
 * shard (nshards > 1) always runs the symmetric kernel, for any n. */
int rmb_matvec2_device(rmb_ctx* ctx, int kind, const double* vec_a_dev, const double* vec_b_dev, double eta,
                       double* out_a_dev, double* out_b_dev);
int rmb_matvec2_pairshard_device(rmb_ctx* ctx, int kind, const double* vec_a_dev, const double* vec_b_dev, double eta,
                                 double* out_a_dev, double* out_b_dev, long shard, long nshards);

/* Several blocks of the grand mobility from ONE pass over the unordered pairs: differences, both inverse square
 * roots, the wall polynomials and the heights are evaluated once per pair and shared by all blocks.
 *   RMB_OP_VELOCITY_FROM_FORCE_TORQUE  in: f, tau   out: u = M_tt f + M_tr tau          (= RMB_TT_TR;
 *       mobility_pycuda.py:1266-1391 / :1394-1512)
 *   RMB_OP_GRAND                       in: f, tau   out: u, w = [[M_tt, M_tr], [M_rt, M_rr]] [f; tau]
 *       (quaternion_integrator/quaternion_integrator_rollers.py:1114-1121 applies the four blocks separately)
 *   RMB_OP_FORCE_COLUMN                in: f        out: u = M_tt f, w = M_rt f   (the two random-finite-difference
 *       products of one draw, quaternion_integrator_rollers.py:1138-1160)
 *   RMB_OP_TT_MULTI (TR_ / RT_ / RR_)  in: k vectors (1..4)   out: the block applied to each (solves and Lanczos
 *       recursions advanced in lockstep on the same configuration: geometry and coefficients once per pair)
 * in_dev / out_dev: arrays of n_in / n_out device pointers to 3n doubles (3*(end-begin) for outputs under a target
 * range).  in_plane != 0 zeroes the z component of every input and output (mobility_numba.py:291, :690).
 * Wall / no-wall / pseudo-periodic follow rmb_set_positions.  The *_pairshard variant evaluates pair shard `shard`
 * of `nshards` into full-length partial outputs (sum over shards = product), as rmb_matvec_pairshard_device. */
enum rmb_op { RMB_OP_VELOCITY_FROM_FORCE_TORQUE = 0, RMB_OP_GRAND = 1, RMB_OP_FORCE_COLUMN = 2, RMB_OP_TT_MULTI = 3,
              RMB_OP_TR_MULTI = 4, RMB_OP_RT_MULTI = 5, RMB_OP_RR_MULTI = 6 };
int rmb_matvec_op_device(rmb_ctx* ctx, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                         double* const* out_dev, double eta);
int rmb_matvec_op_pairshard_device(rmb_ctx* ctx, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                                   double* const* out_dev, double eta, long shard, long nshards);

/* Dense translation-translation mobility of each rigid body's own blobs (building block of the
 * block-diagonal preconditioner, multi_bodies/multi_bodies.py:516-531; replaces body/body.py:186-191 ->
 * mobility/mobility.py:1018-1116 / :967-1013 called once per body in Python).  All listed bodies have
 * n_b contiguous blobs starting at first_blob[k]; out[k] is the (3 n_b x 3 n_b) row-major block
 * B M(z_eff) B of body k.  Device pointers, asynchronous.  Non-periodic. */
int rmb_body_mobility_dense_device(rmb_ctx* ctx, const long* first_blob_dev, long n_bodies, int n_b, double eta,
                                   double* out_dev);

/* The O(N) pieces of the rigid-body saddle-point solve that sit between two sweeps (csrc/rmb_krylov.hip).  Device
 * pointers, asynchronous on the context's stream, no positions needed, every reduction in a fixed order.
 *
 * rmb_block_apply_device: for every batch entry b (a rigid body)
 *     y1_b = beta1 y1_b + alpha (A11_b x1_b + A12_b x2_b)       y1_b: r1 values at y1 + b r1,  x1_b: c1 values at x1 + b c1
 *     y2_b = beta2 y2_b + alpha (A21_b x1_b + A22_b x2_b)       y2_b: r2 values at y2 + b r2,  x2_b: c2 values at x2 + b c2
 * in ONE launch.  A block is addressed as p[b batch_stride + row row_stride + col col_stride] (a transposed block is
 * the same memory with the two strides exchanged); a NULL rmb_block* or p == NULL is a zero block; beta == 0 does not
 * read y.  x and y must not overlap.  c1 + c2 + r1 + r2 <= 8192.  Replaces the four batched products of the block-diagonal
 * preconditioner (multi_bodies/multi_bodies.py:548-560: A = the blocks of [[M, -K], [-K^T, 0]]^-1 per body) and the
 * K U / K^T lambda products of the operator (multi_bodies/multi_bodies.py:327-375 called from :424-471).
 *
 * rmb_krylov_orthogonalize_device: one Arnoldi step's orthogonalisation against the `rows` basis vectors V[0..rows)
 * (row r at V + r ldv, n values), two passes of classical Gram-Schmidt:
 *     h = V w;  w -= V^T h;  h2 = V w;  w -= V^T h2;   col[0..rows) = h + h2,  col[rows] = |w|,  v_next = w / |w|
 * (w is overwritten with the orthogonalised, un-normalised vector; |w| == 0 leaves inf / nan in v_next, as the division
 * would: test col[rows]).  rows <= 256.  Four launches; what scipy.sparse.linalg.gmres does internally for the
 * reference (general_application_utils.py:608-627). */
/* Per-body geometry and the per-body factors of the block-diagonal preconditioner, one launch each (csrc/rmb_rigid.hip).
 * Bodies of one call have n_b blobs each, stored body after body.
 *
 * rmb_rigid_configuration_device: r = R(q) ref + x for every blob (body/body.py:64-78; rotation matrix of the unit
 * quaternion (s, p) as quaternion_integrator/quaternion.py:41-51), the body-frame offsets rel = R(q) ref and
 * K = [I, -(rel x)] per blob (body/body.py:81-115).  ref (n_bodies, n_b, 3), loc (n_bodies, 3), quat (n_bodies, 4),
 * r (n_bodies n_b, 3), rel (n_bodies, n_b, 3) or NULL, K (n_bodies, 3 n_b, 6) row-major or NULL.
 *
 * rmb_rigid_preconditioner_device: from each body's dense blob mobility Mb (n_bodies, 3 n_b, 3 n_b; symmetrised on
 * load; rmb_body_mobility_dense_device builds it) and K: Lchol (Mb = L L^T, lower, zeros above), Linv = L^-1,
 * Minv = Mb^-1, Nbody = (K^T Mb^-1 K)^-1 (6 x 6), and the blocks of [[Mb, -K], [-K^T, 0]]^-1:
 *   A12 = -Mb^-1 K N (3 n_b x 6), A11 = Mb^-1 + A12 (Mb^-1 K)^T, A21 = A12^T (6 x 3 n_b), A22 = -N
 * (multi_bodies/multi_bodies.py:516-531 builds L and N, :548-560 applies them).  *info_dev (one int, device) is set to
 * 0 and then to 1 by any body whose Mb is not positive definite or whose 6 x 6 resistance K^T Mb^-1 K has no accurate
 * inverse (single blobs, collinear rods: the reference takes the pseudo-inverse there, the caller must too).
 * n_b <= 42, i.e. the reference's 12- and 42-blob shells (one workgroup per body, the factors of a body live in LDS: two
 * n x n matrices up to 16 blobs, ONE matrix worked on in place -- Cholesky, triangular inverse, M^-1 formed on the way out --
 * from 17 to 42 blobs: 126 x 126 doubles = 127 KB of the 160 KB).
 *
 * rmb_rigid_advance_device: loc_out = loc + U[:, 0:3] dt, quat_out = quaternion(U[:, 3:6] dt) * quat for every body
 * (quaternion_integrator/quaternion_integrator_multi_bodies.py:86-91; quaternion.py:17-39: the rotation quaternion is
 * (cos |phi|/2, sin(|phi|/2) phi / |phi|), multiplied from the LEFT).  U (n_bodies, 6); dt_body_dev: NULL, or one step
 * per body that replaces dt (the random finite difference scales the displacement by the body length).  Outputs may
 * alias the inputs. */
/* The rigid-body saddle-point operator in one call (multi_bodies/multi_bodies.py:424-471, all bodies free, one body shape):
 *   out[0 .. 3N)        = M_tt lambda - K U          (N = n_bodies n_b blobs, lambda = x[0 .. 3N), U = x[3N .. 3N + 6 n_bodies))
 *   out[3N .. 3N + 6nb) = -K^T lambda
 * on the context's resident configuration (n = n_bodies n_b).  K (n_bodies, 3 n_b, 6) row-major as
 * rmb_rigid_configuration_device writes it.  With the symmetric kernels (open boundaries, double precision, default
 * modes) this is the pair sweep + ONE finishing launch (workgroup = body: self term, scaling, - K U, and -K^T lambda by
 * a reduction over the body's blobs); otherwise the product and the two K products as separate launches, same result. */
int rmb_rigid_operator_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* K_dev, const double* x_dev, double eta,
                              double* out_dev);
/* One Arnoldi step of the right-preconditioned GMRES of the rigid-body problem (quaternion_integrator_multi_bodies.py:
 * 1441-1547 -> general_application_utils.py:608-627; all bodies free, one body shape), enqueued by ONE call:
 *   z = P^-1 v_j   (the four blocks of rmb_rigid_preconditioner_device, one launch),
 *   w = [M z_lambda - K z_U; -K^T z_lambda]   (rmb_rigid_operator_device: pair sweep + one finishing launch),
 *   two classical Gram-Schmidt passes of w against v_0 .. v_j, column j of the Hessenberg matrix to col_dev[0 .. j + 1]
 *   (and to col_mapped_dev when not NULL), |w| to col_dev[j + 1], v_{j+1} = w / |w|   (rmb_krylov_orthogonalize2_device).
 * V_dev: (restart + 1) rows of ldv doubles, n = 3 n_bodies n_b + 6 n_bodies unknowns each; z_dev, w_dev: n doubles of
 * scratch.  Seven launches from one host call, no copy command -- what a small deck's iteration costs is the number of
 * launches (~4.5 us each however little they do) and of host calls, not their work (profiles/r5_gmres_step.txt). */
int rmb_rigid_arnoldi_step_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev,
                                  const double* A21_dev, const double* A22_dev, const double* K_dev, double* V_dev, long ldv, long j,
                                  double eta, double* z_dev, double* w_dev, double* col_dev, double* col_mapped_dev);
/* One step of the preconditioned Lanczos recursion for the Brownian forcing (P^T M P)^{1/2} z, P = blockdiag(L_b^-T)
 * (stochastic_forcing/stochastic_forcing.py:112-264 driven by multi_bodies/multi_bodies.py:590-614), enqueued by ONE call:
 *   y = P v_i,  w = M_tt y,  w <- P^T w   (two block launches around the pair sweep),
 *   two classical Gram-Schmidt passes of w against v_0 .. v_i: col_dev[i] = h_ii, col_dev[i + 1] = h_{i+1,i} (also to
 *   col_mapped_dev when not NULL), v_{i+1} = w / |w|.
 * Linv_dev: (n_bodies, 3 n_b, 3 n_b) = L_b^-1 as rmb_rigid_preconditioner_device writes it; V_dev rows of ldv >= 3 N doubles;
 * y_dev, w_dev: 3 N doubles of scratch. */
int rmb_rigid_lanczos_step_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* Linv_dev, double* V_dev, long ldv, long i,
                                  double eta, double* y_dev, double* w_dev, double* col_dev, double* col_mapped_dev);
/* The whole right-preconditioned GMRES(restart) of [[M, -K], [-K^T, 0]] x = b (quaternion_integrator_multi_bodies.py:1441-1547
 * -> general_application_utils.py:608-627 -> scipy; all bodies free, one body shape) as ONE call: per iteration one
 * rmb_rigid_arnoldi_step_device and an event; the Givens rotations and the convergence test run on the host INSIDE the
 * library, one iteration behind the device, on the Hessenberg column the Gram-Schmidt kernel stored into mapped host
 * memory.  Stops when |b - A x| <= tol |b| by the rotated residual (scipy's tol, atol = 0), on an exact breakdown, or
 * after maxiter INNER iterations; the true residual is formed at every restart.  b_dev: n = 3 n_bodies n_b + 6 n_bodies
 * doubles (the caller normalises it as the reference does, :1518-1521); x_dev: the solution P^-1 y.  *iterations, *residual
 * (relative), *discarded (steps enqueued for nothing: at most one per restart cycle), *products (operator applications),
 * history[0 .. min(iterations, history_cap)) = the relative residual after every iteration (NULL: not wanted).
 * rhs_norm: NULL = b_dev is used as given; otherwise b_dev is the RAW right-hand side: it is scaled to unit norm inside,
 * the solution scaled back, and *rhs_norm = |b| (0: x = 0, no iteration).
 * Synchronous: returns when x_dev is enqueued on the context's stream and every scalar is final. */
int rmb_rigid_gmres_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* A11_dev, const double* A12_dev, const double* A21_dev,
                           const double* A22_dev, const double* K_dev, const double* b_dev, double tol, long restart, long maxiter,
                           double eta, double* x_dev, long* iterations, double* residual, long* discarded, long* products,
                           double* history, long history_cap, double* rhs_norm);
/* The whole preconditioned Lanczos forcing  noise = factor * blockdiag(L_b) (P^T M P)^{1/2} z,  P = blockdiag(L_b^-T), of the
 * Brownian rigid-body schemes (quaternion_integrator_multi_bodies.py:966-973 -> stochastic_forcing/stochastic_forcing.py:112-264
 * with the preconditioner of multi_bodies.py:590-614; covariance factor^2 M) as ONE call: per iteration one
 * rmb_rigid_lanczos_step_device and an event; the small tridiagonal eigenproblem (QL sweeps inside the library) and the
 * reference's stopping rule (:239-255: relative change of the noise estimate below tol) run on the host one iteration
 * behind the device.  Linv_dev / Lchol_dev: (n_bodies, 3 n_b, 3 n_b) contiguous factors of the body mobilities
 * (rmb_rigid_preconditioner_device); z_dev: 3 N standard normals; noise_dev: 3 N doubles out; max_rows: basis vectors the
 * workspace may hold (2 .. 254).  *status: 0 = noise_dev written, *iterations as the reference counts them; 1 = exact
 * breakdown / eigen-solve failure, 2 = more than max_rows basis vectors needed -- then nothing is written, the stream is
 * drained and the caller runs its general loop.  *products = pair sweeps enqueued (iterations + 1, + 1 discarded).
 * Returns when noise_dev is enqueued on the context's stream and every scalar is final. */
int rmb_rigid_lanczos_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* Linv_dev, const double* Lchol_dev,
                             const double* z_dev, double factor, double tol, long max_iter, long max_rows, double eta,
                             double* noise_dev, long* iterations, long* products, int* status);
/* The unpreconditioned forcing of the single-blob (roller) schemes as one call, same loop:  noise = factor * M^{1/2} z  with
 * product 0: M = M_tt over 3 N unknowns (in_plane != 0: its in-plane variant), product 1: the 6 N x 6 N grand mobility
 * [[M_tt, M_tr], [M_rt, M_rr]], z = [z_f; z_tau] (quaternion_integrator_rollers.py:1082-1121, :1203-1260, :1315-1353 ->
 * stochastic_forcing_lanczos without preconditioner: 40-50 iterations at tol 1e-6).  Arguments and status as above. */
int rmb_lanczos_device(rmb_ctx* ctx, int product, int in_plane, const double* z_dev, double factor, double tol, long max_iter,
                       long max_rows, double eta, double* noise_dev, long* iterations, long* products, int* status);
/* HOST function, no GPU work: coef = scale * Q sqrt(max(lambda, 0)) Q^T e_1 of the k x k symmetric tridiagonal matrix with
 * diagonal h_diag[0 .. k) and off-diagonal h_sup[0 .. k-1) -- the coordinates of the Lanczos noise estimate in the Krylov
 * basis after k iterations (stochastic_forcing.py:215-229), as rmb_rigid_lanczos_device computes them. */
int rmb_lanczos_noise_coefficients(long k, const double* h_diag, const double* h_sup, double scale, double* coef_out);
int rmb_rigid_configuration_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* ref_dev, const double* loc_dev,
                                   const double* quat_dev, double* r_dev, double* rel_dev, double* K_dev);
int rmb_rigid_advance_device(rmb_ctx* ctx, long n_bodies, const double* loc_dev, const double* quat_dev, const double* U_dev,
                             double dt, const double* dt_body_dev, double* loc_out_dev, double* quat_out_dev);
int rmb_rigid_preconditioner_device(rmb_ctx* ctx, long n_bodies, long n_b, const double* Mb_dev, const double* K_dev,
                                    double* Lchol_dev, double* Linv_dev, double* Minv_dev, double* Nbody_dev, double* A11_dev,
                                    double* A12_dev, double* A21_dev, double* A22_dev, int* info_dev);

typedef struct rmb_block { const double* p; long batch_stride, row_stride, col_stride; } rmb_block;
int rmb_block_apply_device(rmb_ctx* ctx, long n_batch, long r1, long c1, long r2, long c2, const rmb_block* a11,
                           const rmb_block* a12, const rmb_block* a21, const rmb_block* a22, const double* x1_dev,
                           const double* x2_dev, double alpha, double beta1, double* y1_dev, double beta2, double* y2_dev);
int rmb_krylov_orthogonalize_device(rmb_ctx* ctx, long n, long rows, const double* V_dev, long ldv, double* w_dev,
                                    double* col_dev, double* v_next_dev);
/* The same step with the new column written once more to col_mapped_dev (rows + 1 doubles of page-locked, device-mapped
 * host memory from rmb_host_mapped_alloc; NULL = rmb_krylov_orthogonalize_device): the host reads it after one event wait,
 * no copy command. */
int rmb_krylov_orthogonalize2_device(rmb_ctx* ctx, long n, long rows, const double* V_dev, long ldv, double* w_dev,
                                     double* col_dev, double* v_next_dev, double* col_mapped_dev);
/* Page-locked host memory mapped into the device's address space (hipHostMallocMapped), zero-filled: *host = the address
 * the host reads / writes, *dev = the address kernels use.  Freed with rmb_host_mapped_free(host). */
int rmb_host_mapped_alloc(size_t bytes, void** host, void** dev);
int rmb_host_mapped_free(void* host);

/* Blob-blob soft repulsion on the resident positions (multi_bodies/forces_numba.py:12-55,
 * forces_pycuda.py:66-118): out (n_targets,3).  Uses the UNCLAMPED positions: call
 * rmb_set_positions with wall = 0 first (the reference passes raw r_vectors). */
int rmb_blob_blob_force(rmb_ctx* ctx, double repulsion_strength, double debye_length, double blob_radius,
                        double* out_host);
int rmb_blob_blob_force_device(rmb_ctx* ctx, double repulsion_strength, double debye_length,
                               double blob_radius, double* out_dev);
/* One-blob forces of the rigid-multiblob driver (multi_bodies/multi_bodies_functions.py:153-188, `blob_external_force`):
 * f = (0, 0, -weight + wall repulsion), wall repulsion = (eps_wall / debye_wall) exp(-(h - a) / debye_wall) above contact
 * (h > a), eps_wall / debye_wall below; r_dev: n x 3 raw coordinates (the caller's, not the resident ones); accumulate != 0
 * adds to out_dev's z entries (e.g. on top of rmb_blob_blob_force_device's result), 0 overwrites all 3 n entries. */
int rmb_one_blob_force_device(rmb_ctx* ctx, long n, const double* r_dev, double blob_radius, double weight, double eps_wall,
                              double debye_wall, int accumulate, double* out_dev);
/* Pair shard `shard` of `nshards` of the forces (each unordered pair once, F_ji = -F_ij) into a full-length partial
 * (n,3): the sum over shards is rmb_blob_blob_force_device's result (all-reduce on several GPUs).  Atomic flushes
 * whatever "deterministic" says. */
int rmb_blob_blob_force_pairshard_device(rmb_ctx* ctx, double repulsion_strength, double debye_length,
                                         double blob_radius, double* out_dev, long shard, long nshards);
/* Same with one radius per blob: contact distance a_i + a_j instead of 2a (multi_bodies/forces_numba.py:73-137,
 * `blob_blob_force_implementation radii_numba`).  radii: double[n], host / device. */
int rmb_blob_blob_force_radii(rmb_ctx* ctx, const double* radii, double repulsion_strength, double debye_length,
                              double* out);
int rmb_blob_blob_force_radii_device(rmb_ctx* ctx, const double* radii_dev, double repulsion_strength,
                                     double debye_length, double* out_dev);

/* Timing of the dominant (sweep) kernel, measured with HIP events on the context's stream when the
 * "timing" option is on.  Copies up to max_n most recent durations (ms) into ms[], returns count. */
int rmb_timing_collect(rmb_ctx* ctx, double* ms, int max_n);
int rmb_timing_reset(rmb_ctx* ctx);
/* Measurement aid: chip-wide issue rate of independent v_fma_f64 (G wave-instructions per second, 4 waves per SIMD on
 * every CU, no memory traffic) over `launches` back-to-back launches on the context's stream -- the ceiling bench.py
 * prices the VALU-bound pair sweeps against, measured in the same process and clock state.  Synchronous. */
int rmb_ubench_fp64_issue(rmb_ctx* ctx, int launches, double* g_wave_instr_per_s);
/* Schedule diagnostics: with option "wave_clock" = 1 the symmetric kernel stamps every wave's start and end
 * (100 MHz wall clock); copies (start, end) pairs of the last launch into stamps[2*max_waves], returns count. */
int rmb_wave_clock_collect(rmb_ctx* ctx, long long* stamps, long max_waves);
/* Host wall clock of the last rmb_matvec call on this context, microseconds: us4 = {upload of the vector(s) (pageable
 * host memory: the copy is staged, the call returns when it is done), enqueue of the kernels, wait for them + download
 * of the result, the whole call}.  What the reference's synchronous call shape costs beyond the kernel. */
int rmb_last_host_timing(rmb_ctx* ctx, double* us4);
/* launch geometry of the last sweep: target tiles, source chunks, workgroups */
int rmb_last_launch(rmb_ctx* ctx, long* tiles, long* chunks, long* workgroups);
int rmb_ctx_synchronize(rmb_ctx* ctx);

/* Options of the library's default context, the one the stateless entry points below run on (rmb_mobility_oneshot,
 * rmb_forces_oneshot, rmb_mobility_source_target, rmb_pressure_stokeslet, rmb_double_layer): same keys as
 * rmb_ctx_set_option -- e.g. "precision" = 32 selects the single-precision twins for one-shot callers too. */
int rmb_default_ctx_set_option(const char* key, long value);
/* Device of the default context: an index, or -1 = RMB_DEVICE / 0 (the initial state).  An existing default context
 * on another device is destroyed (its resident positions go with it) and re-created on the next stateless call. */
int rmb_default_ctx_set_device(int device);

/* ---- single-process multi-device engine ------------------------------------------------------------------------
 * The reference's callers are ONE Python process that calls a module-level function (multi_bodies/multi_bodies.py:
 * 233-287 selects it, :445 / :599 call it; mobility/mobility.py:222-252 is the call shape): to give that call the
 * whole node the sharding has to sit behind it.  An engine owns one context per listed device, each with its own
 * stream.  Per product: inputs go to every device (host entry: one pinned staging copy + G uploads; device entry: every
 * shard pulls from devices[0] over xGMI with hipMemcpyPeerAsync), device g evaluates pair shard g of G (each unordered
 * pair once, both blobs updated) into a full-length partial, then device g sums slice g of the G partials in FIXED order
 * through peer-mapped reads of the other devices' partials and hands the slice over (peer copy to devices[0] / download).
 * Caller-owned memory is only ever touched by runtime copies; peer-mapped loads are confined to buffers the engine
 * allocated after it enabled peer access.  With option
 * "deterministic" = 2 the mobility products are bit-reproducible for a given device list (the forces' pair shards
 * flush with atomics whatever the option says).  No reference counterpart (single
 * device); the contract is "equal to the one-context result to rounding" (<= 1e-13).
 *   - devices: 1..16 indices; the same device may be listed several times (rehearsal of the G-device path on one GPU).
 *   - without peer access between two listed devices (or with RMB_MULTI_NO_PEER=1) slices travel by hipMemcpyPeerAsync.
 *   - rmb_multi_set_option: "reduce" 0 [default] = the fixed-order slice reduction, 1 = RCCL all-reduce in place
 *     (ncclCommInitAll; librccl.so is dlopen()ed on first use; distinct devices only; not bit-reproducible); every other
 *     key is forwarded to all shard contexts (rmb_ctx_set_option).  rmb_multi_get_option also answers "peer" (1 / 0)
 *     and "threads".
 *   - with several shards every shard has a worker thread that issues its HIP calls (~25 us per shard and product from
 *     one thread otherwise); RMB_MULTI_THREADS=0 keeps everything on the calling thread.  An engine is used from one
 *     thread at a time and does not survive fork() (its worker threads do not exist in the child: create it there).
 *   - host entry points are synchronous.  *_device entry points take pointers on devices[0], are ordered after the work
 *     already queued on the engine's primary stream (rmb_multi_set_stream; NULL = default stream of devices[0]) and
 *     order that stream after their own completion; the primary stream must be alive when a call is made, and may be
 *     destroyed between calls (a switch never touches the previous handle).
 *   - products: kinds as rmb_matvec (in_plane for TT / TR / RT / RR / TT_TR), operations as rmb_matvec_op_device,
 *     uniform-radius blob-blob forces.  Target ranges do not apply (every call produces all n targets). */
typedef struct rmb_multi rmb_multi;
int rmb_multi_create(const int* devices, int n_devices, rmb_multi** engine);
int rmb_multi_destroy(rmb_multi* engine);
int rmb_multi_n_shards(rmb_multi* engine);
/* the context of shard g (options, rmb_timing_collect, rmb_last_launch of one device); owned by the engine */
int rmb_multi_shard_ctx(rmb_multi* engine, int shard, rmb_ctx** ctx);
int rmb_multi_set_stream(rmb_multi* engine, void* hip_stream);
int rmb_multi_set_option(rmb_multi* engine, const char* key, long value);
int rmb_multi_get_option(rmb_multi* engine, const char* key, long* value);
int rmb_multi_set_positions(rmb_multi* engine, const double* r_host, long n, double a, const double* L, int wall);
int rmb_multi_set_positions_device(rmb_multi* engine, const double* r_dev, long n, double a, const double* L, int wall);
int rmb_multi_matvec(rmb_multi* engine, int kind, int in_plane, const double* vec_host, const double* vec2_host, double eta,
                     double* out_host);
int rmb_multi_matvec_device(rmb_multi* engine, int kind, int in_plane, const double* vec_dev, const double* vec2_dev,
                            double eta, double* out_dev);
int rmb_multi_matvec_op_device(rmb_multi* engine, int op, int in_plane, int n_in, const double* const* in_dev, int n_out,
                               double* const* out_dev, double eta);
int rmb_multi_blob_blob_force(rmb_multi* engine, double repulsion_strength, double debye_length, double blob_radius,
                              double* out_host);
int rmb_multi_blob_blob_force_device(rmb_multi* engine, double repulsion_strength, double debye_length, double blob_radius,
                                     double* out_dev);
int rmb_multi_synchronize(rmb_multi* engine);

/* ---- stateless one-shot calls: exactly the reference wrapper signature ----------------------
 * r, vec (, vec2) host (n,3); out host (n,3).  Uses a process-wide context on the default device (RMB_DEVICE, or 0;
 * rmb_default_ctx_set_device). */
int rmb_mobility_oneshot(int kind, int wall, int in_plane, long n, const double* r, const double* vec,
                         const double* vec2, double eta, double a, const double* L, double* out);
int rmb_forces_oneshot(long n, const double* r, const double* L, double repulsion_strength,
                       double debye_length, double blob_radius, double* out);

/* ---- source -> target products with per-blob radii (K13) --------------------------------------
 * u_t = sum_s M(x_t, a_t; y_s, a_s) f_s for nt targets and ns sources, each blob with its own radius
 * (mobility/mobility_numba.py:1480-1658; CUDA twin mobility_pycuda.py:1974-2078; wrappers
 * mobility.py:494-615: per-blob height clamp + B on both sides when wall == 1).  wall == 2 selects the
 * stress-free surface at z = 0 instead (mobility_numba.py:1941-2091, wrapper mobility.py:1409-1429: mirror image
 * with the z column negated, raw heights, no clamp).  Stateless; the host variant is synchronous, the device
 * variant enqueues on the context's stream. */
int rmb_mobility_source_target(long ns, const double* src, const double* radius_src, long nt, const double* tgt,
                               const double* radius_tgt, const double* force, double eta, const double* L, int wall,
                               double* out);
int rmb_mobility_source_target_device(rmb_ctx* ctx, long ns, const double* src_dev, const double* radius_src_dev,
                                      long nt, const double* tgt_dev, const double* radius_tgt_dev,
                                      const double* force_dev, double eta, const double* L, int wall, double* out_dev);

/* ---- Stokeslet pressure and Stokes double layer, source -> target ---------------------------------------------
 * The remaining O(N_s N_t) operators of mobility/mobility_numba.py (wrappers mobility/mobility.py:1345-1366,
 * :1376-1387, :1432-1442).  Plain positions (no height clamp, no radii), stateless like the K13 entry points; the
 * host variants are synchronous, the device variants enqueue on the context's stream.  Atomic-free, bit-reproducible.
 *
 * rmb_pressure_stokeslet: out[nt], p_t = 1/(4 pi) sum_s f_s . r/|r|^3, wall = 1 adds Blake's image system
 *   (mobility_numba.py:1332-1396, :1399-1476).  The reference's wall routine rescales its running sum inside the
 *   source loop (:1474); the factor is applied once here (= the reference for one source, = superposition of its
 *   single-source results).  L must be NULL or zero: the reference's pseudo-periodic branch divides by the unwrapped
 *   distance (:1374-1375), which is not reproduced -- RMB_ERR_ARG otherwise.
 * rmb_double_layer: out[3 nt], u_t = -3/(4 pi) sum_s w_s r (r.n_s)(r.v_s)/|r|^5, pairs with r <= 1e-14 skipped;
 *   wall = 1 adds the image terms of mobility_numba.py:1725-1759 (evaluated for r = 0 too); blob_radius >= 0 selects
 *   the RPY-regularised unbounded operator (:2095-2168; wall must be 0), blob_radius < 0 the plain one. */
int rmb_pressure_stokeslet(long ns, const double* src, long nt, const double* tgt, const double* force, const double* L,
                           int wall, double* out);
int rmb_pressure_stokeslet_device(rmb_ctx* ctx, long ns, const double* src_dev, long nt, const double* tgt_dev,
                                  const double* force_dev, const double* L, int wall, double* out_dev);
int rmb_double_layer(long ns, const double* src, long nt, const double* tgt, const double* normals, const double* vector,
                     const double* weights, int wall, double blob_radius, double* out);
int rmb_double_layer_device(rmb_ctx* ctx, long ns, const double* src_dev, long nt, const double* tgt_dev,
                            const double* normals_dev, const double* vector_dev, const double* weights_dev, int wall,
                            double blob_radius, double* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* RMB_MOBILITY_H */
