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
int rmb_ctx_create(int device, rmb_ctx** ctx);
int rmb_ctx_destroy(rmb_ctx* ctx);
/* hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int rmb_ctx_set_stream(rmb_ctx* ctx, void* hip_stream);
/* options: "chunks" (0 = auto source-chunk count), "timing" (1 = record HIP events per sweep) */
int rmb_ctx_set_option(rmb_ctx* ctx, const char* key, long value);

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

/* Multi-GPU, symmetric pair sharding (RMB_TT / TR / RT / RR): the unordered blob pairs are cut into
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
 * single products where the symmetric kernel does not apply (n < 128, "deterministic", target sub-ranges). */
int rmb_matvec2_device(rmb_ctx* ctx, int kind, const double* vec_a_dev, const double* vec_b_dev, double eta,
                       double* out_a_dev, double* out_b_dev);
int rmb_matvec2_pairshard_device(rmb_ctx* ctx, int kind, const double* vec_a_dev, const double* vec_b_dev, double eta,
                                 double* out_a_dev, double* out_b_dev, long shard, long nshards);

/* Dense translation-translation mobility of each rigid body's own blobs (building block of the
 * block-diagonal preconditioner, multi_bodies/multi_bodies.py:516-531; replaces body/body.py:186-191 ->
 * mobility/mobility.py:1018-1116 / :967-1013 called once per body in Python).  All listed bodies have
 * n_b contiguous blobs starting at first_blob[k]; out[k] is the (3 n_b x 3 n_b) row-major block
 * B M(z_eff) B of body k.  Device pointers, asynchronous.  Non-periodic. */
int rmb_body_mobility_dense_device(rmb_ctx* ctx, const long* first_blob_dev, long n_bodies, int n_b, double eta,
                                   double* out_dev);

/* Blob-blob soft repulsion on the resident positions (multi_bodies/forces_numba.py:12-55,
 * forces_pycuda.py:66-118): out (n_targets,3).  Uses the UNCLAMPED positions: call
 * rmb_set_positions with wall = 0 first (the reference passes raw r_vectors). */
int rmb_blob_blob_force(rmb_ctx* ctx, double repulsion_strength, double debye_length, double blob_radius,
                        double* out_host);
int rmb_blob_blob_force_device(rmb_ctx* ctx, double repulsion_strength, double debye_length,
                               double blob_radius, double* out_dev);
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
/* Schedule diagnostics: with option "wave_clock" = 1 the symmetric kernel stamps every wave's start and end
 * (100 MHz wall clock); copies (start, end) pairs of the last launch into stamps[2*max_waves], returns count. */
int rmb_wave_clock_collect(rmb_ctx* ctx, long long* stamps, long max_waves);
/* launch geometry of the last sweep: target tiles, source chunks, workgroups */
int rmb_last_launch(rmb_ctx* ctx, long* tiles, long* chunks, long* workgroups);
int rmb_ctx_synchronize(rmb_ctx* ctx);

/* ---- stateless one-shot calls: exactly the reference wrapper signature ----------------------
 * r, vec (, vec2) host (n,3); out host (n,3).  Uses a process-wide context on device 0. */
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

#ifdef __cplusplus
}
#endif
#endif /* RMB_MOBILITY_H */
