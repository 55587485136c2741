/*
 * fluid.h — C-ABI of the MI355X-native 2-D stable-fluids grid solver.
 *
 * This is the drop-in boundary for the per-timestep solver of
 * deaklajos/Fluid-Simulation, FluidSimulation/kernel.cu.  The reference has no
 * FFI: the path is a set of `void f(void)` host functions over file-scope
 * globals (kernel.cu:435-479) called from GLUT callbacks.  Each export below
 * names the reference host function it replaces (file:line under
 * FluidSimulation/).  Plain pointers and sizes only; no C++/torch types.
 *
 * The header is implemented twice with identical semantics:
 *   - fluid-simulation_amd/csrc/  -> libfluid_hip.so   (the product, gfx950 HIP)
 *   - oracle/fluid_ref.c          -> liboracle_fluid.so (test infrastructure only)
 *
 * Field layout handed across the boundary (fluid_read / fluid_write): dense
 * row-major, x fastest, index = x + y*n (kernel.cu:79, :398); velocity is
 * interleaved (u,v) float2, density interleaved (r,g,b,a) float4, pressure and
 * divergence plain float.  All arithmetic is fp32.
 *
 * Threading: one thread per context at a time (the reference is single
 * threaded).  Several contexts may coexist.  Every call returns 0 on success
 * or a negative FLUID_E_* code; nothing ever calls exit().
 */
#ifndef FLUID_H
#define FLUID_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUID_API __attribute__((visibility("default")))

/* ---- error codes ------------------------------------------------------- */
enum {
  FLUID_OK = 0,
  FLUID_E_INVALID = -1,  /* bad argument (NULL, n < 4, unknown field, ...)    */
  FLUID_E_NOMEM = -2,    /* host or device allocation failed                   */
  FLUID_E_DEVICE = -3,   /* HIP runtime / kernel launch error (see fluid_last_error) */
  FLUID_E_SIZE = -4,     /* caller buffer too small / larger than a stated maximum   */
  FLUID_E_COMM = -5,     /* halo transport error                               */
  FLUID_E_HALO = -6,     /* advection backtrace left the slab halo (multi-GPU) */
  FLUID_E_UNSUPPORTED = -7
};

/* ---- fields (fluid_read / fluid_write) ---------------------------------- */
enum {
  FLUID_VELOCITY = 0,   /* float2 * n*n : velocityBuffer[inputVelocityBuffer]  kernel.cu:442-444 */
  FLUID_DENSITY = 1,    /* float4 * n*n : densityBuffer[inputDensityBuffer]    kernel.cu:446-448 */
  FLUID_PRESSURE = 2,   /* float  * n*n : pressureBuffer[inputPressureBuffer]  kernel.cu:451-453 */
  FLUID_DIVERGENCE = 3, /* float  * n*n : divergenceBuffer                     kernel.cu:455-456 */
  FLUID_VORTICITY = 4   /* float  * n*n : vorticityBuffer (only written when
                           FLUID_FLAG_VORTICITY is set)                        kernel.cu:458-459 */
};

/* ---- flags --------------------------------------------------------------- */
enum {
  FLUID_FLAG_VORTICITY = 1u << 0, /* run simulateVorticity() between diffusion and projection
                                     (kernel.cu:705, commented out in the reference => default off) */
  FLUID_FLAG_NO_GRAPH = 1u << 1,  /* launch kernels directly instead of replaying a hipGraph   */
  FLUID_FLAG_UNFUSED = 1u << 2    /* one kernel launch per reference kernel launch (25/step),
                                     no temporal blocking; the straight mapping of kernel.cu   */
};

/*
 * Parameters.  fluid_default_params() fills in the reference's literals:
 *   n=512 (kernel.cu:438), diffusion_iters=10 (:584), pressure_iters=10 (:620),
 *   dt=0.1f (:31), viscosity=0.01f (:181), splat_radius=0.001f (:368).
 * The reference additionally needs n % 32 == 0 (kernel.cu:439-440, unguarded
 * kernels); this library accepts any n >= 4.
 *
 * Slab decomposition (one process per GPU): the n x n domain is cut into
 * slab_count row slabs; this context owns rows [n*slab_rank/slab_count,
 * n*(slab_rank+1)/slab_count).  slab_count <= 1 means the whole domain.
 * Halo rows travel through the caller (fluid_halo_* below): one exchange per
 * step, see fluid_slab_plan.
 */
typedef struct fluid_params {
  int n;
  int diffusion_iters;
  int pressure_iters;
  float dt;
  float viscosity;
  float splat_radius;
  int device;          /* HIP device ordinal this context lives on            */
  unsigned flags;      /* FLUID_FLAG_*                                        */
  int slab_rank;       /* 0 .. slab_count-1                                   */
  int slab_count;      /* 0 or 1 = no decomposition                           */
  int halo_rows;       /* A: advection reach in rows when slab-decomposed (0 = 8) */
  int pressure_groups; /* slab contexts: split the J sweeps into this many groups with a
                          pressure-halo exchange between them (0/1 = one group); see below */
  int ny;              /* rows of the grid; 0 = n (square, the reference's only case:
                          kernel.cu:438).  n is then the number of columns: index = x + y*n,
                          x < n, y < ny.  Every rule keeps its form with the extent taken per
                          axis (x tests and clamps against n, y tests and clamps against ny;
                          fluid_add_source normalises x by n and y by ny)                 */
  int inner_slabs;     /* whole-domain contexts: 0/1 = one context on one stream (default).  G >= 2
                          = run the domain as G row slabs of THIS GPU side by side, one HIP
                          stream each ("inner slabs", see below)                          */
  int reserved[2];     /* must be zero                                        */
  int n_devices;       /* 0: one context on `device` (above).  G >= 1: ONE process drives G GPUs, as
                          the reference's main() drives its one (kernel.cu:887-946): the domain is cut
                          into G row slabs, slab g lives on device_ids[g], and this context answers for
                          the whole grid ("several GPUs under one context", below).  Not together with
                          inner_slabs or slab_count.                                              */
  int device_ids[8];   /* HIP device ordinals, slab 0 (rows from 0) first; an ordinal may repeat */
} fluid_params;

/*
 * Several GPUs under one context (fluid_params.n_devices = G, device_ids[]; SURVEY.md 8b's
 * single-process contract; no counterpart in the reference, which is single-GPU).  The context is the
 * parent of G slab contexts, one per entry of device_ids, each with its own stream on its own device;
 * fluid_step enqueues every slab's launches from the calling thread and moves the halo rows between
 * neighbouring slabs with device-to-device copies on the sending slab's stream (peer copies over xGMI
 * when the two devices differ), ordered by events — the scheme of the inner slabs above, with the
 * children placed on different devices.  fluid_read returns the whole grid, slabs gathered in y order;
 * fluid_write / fluid_add_source / fluid_reset / fluid_diagnostics fan out.  Same limits as inner slabs.
 * With every ordinal the same the G slabs share that GPU (that is what the builder's one-GPU boxes
 * could run: bit-identical to the plain context, tests/test_inner_slabs.py); with G different GPUs
 * nothing has been measured, and steps are enqueued directly (no hipGraph across devices).  One process
 * per GPU with fluid_comm_* (RCCL) is the other multi-GPU form.
 */

/*
 * Inner slabs (fluid_params.inner_slabs = G >= 2; no counterpart in the reference).  The launches of a
 * mid-size grid (2048^2 .. 8192^2) each fill the GPU for 30-100 us and lose a good part of that to
 * their ramp-up and drain, and consecutive launches of one step depend on each other.  Two slabs of
 * the same domain are independent for a whole step (the slab decomposition's one exchange per step,
 * fluid_slab_plan), so their launches overlap on the GPU and fill each other's gaps: 4096^2, J = 40:
 * 0.53 -> 0.50 ms per step with G = 2 on MI355X (0.48 with 3); nothing to gain at 16384^2.  The context behaves as a
 * whole-domain context and gives the same bits (fluid_read / fluid_write / fluid_add_source /
 * fluid_diagnostics / fluid_geometry see the whole grid; halo rows move between the slabs by
 * device-side copies inside fluid_step), with these limits: the advection reach of slab contexts
 * applies (halo_rows, FLUID_E_HALO if a backtrace is longer); FLUID_FLAG_UNFUSED is
 * refused (FLUID_E_UNSUPPORTED), and so are fluid_step_noalloc,
 * fluid_profile_steps, the single-stage functions, fluid_step_part, fluid_halo_* and fluid_comm_*
 * on such a context.  The shortest slab must hold a halo (fluid_slab_plan's rule).
 */

typedef struct fluid_ctx fluid_ctx;

/* Fill *p with the reference literals (see above).  Never fails on non-NULL. */
FLUID_API int fluid_default_params(fluid_params *p);

/* initBuffers() + resetSimulation()  kernel.cu:481-528.
 * Allocates every field on p->device and zeroes them.  p == NULL = defaults.
 * The context is an opaque heap object; no globals. */
FLUID_API int fluid_init(fluid_ctx **out, const fluid_params *p);

/* Teardown, kernel.cu:914-934.  NULL is a no-op. */
FLUID_API void fluid_destroy(fluid_ctx *ctx);

/* resetSimulation()  kernel.cu:518-528 (the 'r' key, :819-821): zero the
 * current velocity, pressure and density. */
FLUID_API int fluid_reset(fluid_ctx *ctx);

/* addForce() -> addForceCUDA  kernel.cu:687-699, :357-383.
 * (sx,sy) are the NORMALIZED coordinates addForceCUDA receives (x/width,
 * y/height, :689-690); (fx,fy) the raw force; rgba = densityColor (:507),
 * caller-owned.  Applied to every cell, border included; ordered with respect
 * to steps (same stream). */
FLUID_API int fluid_add_source(fluid_ctx *ctx, float sx, float sy, float fx, float fy,
                               const float rgba[4]);

/* simulationStep()  kernel.cu:701-708, nsteps times.  Asynchronous: work is
 * enqueued on the context's stream (a replayed hipGraph unless
 * FLUID_FLAG_NO_GRAPH); no host synchronisation is implied. */
FLUID_API int fluid_step(fluid_ctx *ctx, int nsteps);

/* Build (capture and instantiate) the graphs fluid_step(ctx, nsteps) will replay, without running
 * anything: whole steps are captured in power-of-two batches per graph (21 steps = 16 + 4 + 1, at
 * most FLUID_GRAPH_STEPS = 16 per graph), one graph per batch size and ping-pong state, built on
 * first use otherwise.  No counterpart in the reference (it launches kernel by kernel); for
 * callers that want the first fluid_step of a given length to cost what the later ones do. */
FLUID_API int fluid_prepare_steps(fluid_ctx *ctx, int nsteps);

/* fluid_step for callers that cannot allocate: code under HIP graph capture (with the context on
 * the capturing stream, fluid_set_stream), tight serving loops, harnesses that forbid hipMalloc.
 * No counterpart in the reference.  Same kernels, same results as fluid_step, bit for bit; it
 * launches them directly (fluid_step captures and instantiates hipGraphs on first use, which
 * allocates) and gives the temporally blocked kernels their dump row (the sink for rows a wave
 * computes but must not store) in a file-scope device array of FLUID_NOALLOC_MAX_N * 16 + 256
 * bytes (256.25 KiB, one copy per GPU, held until the process ends) instead of the context's own.
 * Calls no hipMalloc / hipFree / allocating routine.  n > FLUID_NOALLOC_MAX_N: FLUID_E_SIZE; slab
 * contexts: FLUID_E_UNSUPPORTED.  Every call on a GPU shares that GPU's array: its contents are
 * never read, so nothing is cleared between calls and calls of different contexts may overlap.
 * (Since the end of round 3 most launches drop those rows in hardware instead; the sink is still
 * written by the divergence + pressure launch of plans whose last diffusion launch does not write
 * the divergence itself: FLUID_FLAG_VORTICITY, diffusion counts not ending in 8 or 10 sweeps.) */
#define FLUID_NOALLOC_MAX_N 16384
FLUID_API int fluid_step_noalloc(fluid_ctx *ctx, int nsteps);

/* Test hook for the above; no counterpart in the reference and not part of the solver.  The sink
 * row is write-only, so results cannot show which one the kernels were handed: the context's own
 * (which = 0) or this GPU's static scratch (which = 1; its first n * 16 + 256 bytes, the part a
 * grid of width n uses).  fill != 0: every 32-bit word of that row := pattern, on the context's
 * stream.  differing != NULL: synchronise and store the number of words that are not `pattern`.
 * (Allocates host memory: it is a checker, not a no-allocation entry point.) */
FLUID_API int fluid_debug_sink(fluid_ctx *ctx, int which, int fill, unsigned pattern, long long *differing);

/* Diagnostics of the temporally blocked launches; no counterpart in the reference.  A context made
 * with FLUID_TB_TRACE=1 in the environment keeps four time stamps per wave of its last such launch
 * (device-wide 100 MHz counter: start, pipeline filled, tail begins, end; record (chunk-row, strip)
 * at out[4 * (chunk-row * strips + strip)], the third stamp only from waves that ran the tail of a
 * chained chunk, FLUID_TB_CHAIN=1 — DESIGN.md 5.7).  Synchronises; *words = words written, at
 * most max_words.  FLUID_E_UNSUPPORTED on a context made without FLUID_TB_TRACE. */
FLUID_API int fluid_debug_chain_trace(fluid_ctx *ctx, unsigned long long *out, size_t max_words, size_t *words);

/* The four stage functions simulationStep() is made of, individually
 * (kernel.cu:542-555, :582-602, :604-662, :664-685) and the optional
 * simulateVorticity() (:557-580).  fluid_step == advect_velocity, diffuse,
 * [vorticity], project, advect_density in that order. */
FLUID_API int fluid_advect_velocity(fluid_ctx *ctx);
FLUID_API int fluid_diffuse(fluid_ctx *ctx);
FLUID_API int fluid_vorticity(fluid_ctx *ctx);
FLUID_API int fluid_project(fluid_ctx *ctx);
FLUID_API int fluid_advect_density(fluid_ctx *ctx);

/* D2H half of visualizationStep()  kernel.cu:753.  Copies the CURRENT
 * ping-pong half of `field` (kernel.cu:715, :727, :740) into dst (dense,
 * row-major, owned rows only when slab-decomposed) and synchronises the stream
 * (the reference's only sync point is this copy).  dst_bytes must be at least
 * rows*n*sizeof(element). */
FLUID_API int fluid_read(fluid_ctx *ctx, int field, void *dst, size_t dst_bytes);

/* Inverse of fluid_read (checkpoint restore / test state injection); no
 * counterpart in the reference, whose state lives only in device arrays. */
FLUID_API int fluid_write(fluid_ctx *ctx, int field, const void *src, size_t src_bytes);

/* cudaDeviceSynchronize analogue for this context's stream, kernel.cu:914. */
FLUID_API int fluid_sync(fluid_ctx *ctx);

/*
 * Diagnostics of the current state over the owned rows.  No counterpart in the reference
 * (kernel.cu has no residual or diagnostic; its only read-back is the picture, :710-756);
 * computed on the device by wavefront reductions in a fixed order (same state, same bits) and
 * defined with the reference's own formulas so that each value can be checked exactly:
 *   max_abs_u/v           max |u|, max |v| of velocityBuffer[inputVelocityBuffer]; dt * max is the
 *                         advection reach in cells (cf. fluid_params.halo_rows)
 *   max_abs_div           max |divergenceBuffer|: right-hand side of the last projection
 *   max_abs_p             max |pressureBuffer[inputPressureBuffer]|
 *   max_jacobi_update     max over interior cells of |p' - p|, p' = one more pressureJacobi sweep
 *                         (kernel.cu:313): how far the J sweeps stopped from their fixed point
 *   max_abs_div_velocity  max over interior cells of the divergence formula (kernel.cu:282) applied
 *                         to the current velocity: what the projection left behind
 *   dye[4]                sum of each density channel (fp32 per block, double across blocks)
 * A maximum is NaN if any value it covers is.  Synchronises the stream.  Slab contexts: the two
 * stencil quantities read one halo row of velocity on each side, so refresh the velocity halos
 * first (fluid_halo_pack_part / fluid_halo_unpack_part with FLUID_PART_VELOCITY) — the pressure
 * rows they need are already there; combine ranks with max / sum.
 */
typedef struct fluid_diag {
  float max_abs_u, max_abs_v;
  float max_abs_div;
  float max_abs_p;
  float max_jacobi_update;
  float max_abs_div_velocity;
  double dye[4];
} fluid_diag;
FLUID_API int fluid_diagnostics(fluid_ctx *ctx, fluid_diag *out);

/* Checksum of rows of the CURRENT half of `field`, computed on the device: the sum of their 32-bit
 * words modulo 2^64 (equal checksums of equal-length blocks: almost surely equal bits).  row0 / rows
 * are GLOBAL rows inside the rows this context owns; rows == 0: all owned rows.  No counterpart in the
 * reference; for comparing a slab's rows with the same rows of a whole-domain context without moving
 * gigabytes through the host (bench.py --gpus N).  Synchronises the stream. */
FLUID_API int fluid_checksum(fluid_ctx *ctx, int field, int row0, int rows, unsigned long long *out);

/* Geometry of this context: grid width (columns), first owned row, number of owned rows (for a
 * whole-domain context: 0 and the grid height ny). */
FLUID_API int fluid_geometry(const fluid_ctx *ctx, int *n, int *row0, int *rows);

/* Human-readable text for a FLUID_E_* code; never NULL. */
FLUID_API const char *fluid_strerror(int code);
/* Detail of the last FLUID_E_DEVICE / FLUID_E_COMM on this context ("" if none). */
FLUID_API const char *fluid_last_error(const fluid_ctx *ctx);

/* ---- measurement hooks (bench.py; replaces nothing in the reference, which
 *      has no timers: SURVEY.md section 5) ----------------------------------- */

/* Enqueue nsteps steps bracketed by HIP events on the context's own stream,
 * wait for them, and return the elapsed device time in milliseconds. */
FLUID_API int fluid_time_steps(fluid_ctx *ctx, int nsteps, float *ms_out);

#define FLUID_MAX_KERNEL_STATS 16
typedef struct fluid_kernel_stat {
  char name[48];        /* kernel name as it appears in a rocprofv3 trace       */
  int launches;         /* launches in the profiled steps                       */
  float total_ms;       /* sum of launch durations (HIP events, own stream)     */
  double alg_bytes;     /* algorithmic bytes per launch: the
                           bytes of the reference launches this launch stands for   */
  double moved_bytes;   /* bytes the launch itself must move (each input read once,
                           each output written once): < alg_bytes where launches are
                           fused (K sweeps per launch; projection + density advection) */
} fluid_kernel_stat;

/* Run nsteps steps with direct launches, each launch bracketed by HIP events
 * on the context's stream; fill stats[0..*count).  Results are identical to
 * fluid_step. */
FLUID_API int fluid_profile_steps(fluid_ctx *ctx, int nsteps, fluid_kernel_stat *stats,
                                  int max_stats, int *count);

/* ---- slab decomposition: one process per GPU, one halo exchange per step ----
 *
 * No counterpart in the reference (single GPU).  The step's dependency cone is
 * covered by ONE exchange before the step: each slab receives
 *   halo_vel  = J + 1 + D + A   rows of velocity (Jg instead of J with pressure groups; two more
 *                               with FLUID_FLAG_VORTICITY: the confinement force on a row reads the
 *                               curl of the rows either side, and that the velocity one row further) and
 *   halo_dens = A               rows of density
 * from each neighbour (J = pressure_iters, D = diffusion_iters, A = halo_rows,
 * the advection reach in rows, default 8) and recomputes the shrinking cone in
 * its halo instead of exchanging a row per sweep (xGMI halo traffic is latency
 * bound: SURVEY.md 8e).  Results are bit-identical to the whole-domain run.
 * A backtrace longer than A rows raises a device flag; fluid_sync / fluid_read
 * then return FLUID_E_HALO.
 *
 * Per step the caller does, on every slab:
 *   fluid_halo_pack(ctx, -1, send_lo); fluid_halo_pack(ctx, +1, send_hi);
 *   <move send_lo to rank-1's recv_hi, send_hi to rank+1's recv_lo>
 *   fluid_halo_unpack(ctx, -1, recv_lo); fluid_halo_unpack(ctx, +1, recv_hi);
 *   fluid_step(ctx, 1);
 * Buffers are device memory of fluid_halo_bytes(ctx, dir) bytes; pack/unpack are
 * asynchronous copies on the context's stream.  dir = -1 is the neighbour that
 * owns the rows below (slab_rank - 1), +1 the one above.
 */
typedef struct fluid_slab_info {
  int n, row0, rows;            /* owned rows [row0, row0 + rows)                  */
  int halo_vel, halo_dens;      /* rows exchanged per neighbour per step            */
  int store_row0, store_rows;   /* rows kept in the local buffers (owned + halos)   */
  int has_lo, has_hi;           /* neighbour below / above exists                   */
  int pressure_groups;          /* groups the J sweeps are split into (>= 1)        */
  int halo_prs;                 /* pressure rows exchanged between groups (0 if one) */
} fluid_slab_info;

/* Pure host arithmetic (needs no GPU): the geometry fluid_init will use for p. */
FLUID_API int fluid_slab_plan(const fluid_params *p, fluid_slab_info *out);
FLUID_API size_t fluid_halo_bytes(const fluid_ctx *ctx, int dir);

/* The same, one part at a time, so the exchange can hide behind compute:
 *   FLUID_PART_VELOCITY = simulateAdvection + simulateDiffusion + projection  (kernel.cu:703-706)
 *   FLUID_PART_DENSITY  = simulateDensityAdvection                            (kernel.cu:707)
 * fluid_step(ctx, 1) == fluid_step_part(VELOCITY) then fluid_step_part(DENSITY).  The velocity
 * part needs only velocity halos and the density part only density halos (and the projected
 * velocity on the owned rows), so the velocity halos for step s+1 can travel while step s's
 * density advection runs, and the density halos while step s+1's velocity part runs
 * (fluid-simulation_amd/slab.py).  pack/unpack_part move one field's halo block; `stream` is the
 * hipStream_t to enqueue the copy on (NULL = the context's stream).  Works on whole-domain
 * contexts too (fluid_step_part only).
 *
 * Deep pressure loops on short slabs (J = 200 on 1024-row slabs: 219 halo rows) can be cut into
 * fluid_params.pressure_groups groups of Jg = ceil(J / groups) sweeps with an exchange of
 * halo_prs = Jg + 1 pressure rows between groups; the velocity halo shrinks to Jg + 1 + D + A.
 * The velocity part is then driven as
 *   fluid_step_part(VELOCITY_HEAD)    advection, diffusion, divergence, sweeps of group 0
 *   { exchange FLUID_PART_PRESSURE_GROUP halos; fluid_step_part(PRESSURE_GROUP) }  x (groups - 1)
 *   fluid_step_part(VELOCITY_TAIL)    gradient subtraction (kernel.cu:326-355)
 * (fluid_step and FLUID_PART_VELOCITY return FLUID_E_UNSUPPORTED on such a context: the exchanges
 * are the caller's.)  With one group HEAD + TAIL == VELOCITY.  Same bits as the whole-domain run. */
enum {
  FLUID_PART_VELOCITY = 1,
  FLUID_PART_DENSITY = 2,
  FLUID_PART_VELOCITY_HEAD = 4,
  FLUID_PART_PRESSURE_GROUP = 8,
  FLUID_PART_VELOCITY_TAIL = 16,
  /* VELOCITY_TAIL + DENSITY as the one fused kernel, in two parts: first the owned rows that
   * the next exchange sends (within halo_vel rows of a neighbouring slab), then the rest, so that
   * both halo messages travel while the interior is computed.  EDGES flips the current buffers:
   * after it fluid_halo_pack_part sends the new edge rows.  INTERIOR is enqueued as up to 4
   * launches over consecutive row ranges (env FLUID_INTERIOR_SPLIT, 1..8) so that work queued on
   * another stream in the meantime is dispatched at a launch boundary at the latest. */
  FLUID_PART_TAIL_DENSITY_EDGES = 32,
  FLUID_PART_TAIL_DENSITY_INTERIOR = 64,
  /* Slab contexts, between two steps: the next step's advection (kernel.cu:114-140) on the same
   * interior rows, which read no halo row, so it can run before that step's velocity halos have
   * arrived (more cover for the message); the following VELOCITY / VELOCITY_HEAD then advects only
   * the rows either side.  Same kernel, inputs and bits.  A backtrace from those rows that leaves
   * the owned rows (longer than halo_vel rows) is reported as FLUID_E_HALO.  fluid_add_source,
   * fluid_reset and fluid_write cancel it (the full advection runs). */
  FLUID_PART_ADVECT_INTERIOR = 128
};
FLUID_API int fluid_step_part(fluid_ctx *ctx, int part);
FLUID_API size_t fluid_halo_bytes_part(const fluid_ctx *ctx, int dir, int parts);
FLUID_API int fluid_halo_pack_part(fluid_ctx *ctx, int dir, int part, void *dev_dst, void *stream);
FLUID_API int fluid_halo_unpack_part(fluid_ctx *ctx, int dir, int part, const void *dev_src,
                                     void *stream);
FLUID_API int fluid_halo_pack(fluid_ctx *ctx, int dir, void *dev_dst);
FLUID_API int fluid_halo_unpack(fluid_ctx *ctx, int dir, const void *dev_src);

/* ---- halo transport under the C ABI ------------------------------------------
 *
 * No counterpart in the reference (single GPU).  With a transport attached to a slab context,
 * fluid_step(ctx, nsteps) IS the multi-GPU step: before each step the library exchanges the halo
 * blocks itself (one group per step: halo_vel velocity rows and halo_dens density rows to and
 * from each neighbour, sent from and received into the field rows directly — whole rows are
 * contiguous, nothing is staged), on a second HIP stream, overlapped with compute: the last
 * kernel of a step runs on the edge rows first, the exchange for the next step is posted, and
 * the interior rows (and the interior rows of the next step's advection) are computed while the
 * messages travel.  With a capturable transport (RCCL) and FLUID_COMM_GRAPH=1 in the environment
 * whole steps — both streams and the send/recv calls — are captured as hipGraphs of up to 16 steps
 * and replayed; if the capture is refused the steps are enqueued directly instead (fluid_comm_info
 * reports which).  Results are bit-identical to the whole-domain run wherever that could be checked:
 * 2-4 processes sharing one GPU with a host-staged transport, and the RCCL calls with a one-rank
 * communicator.  NO run with two or more RCCL ranks has happened yet (the builder had one GPU at a
 * time), so multi-GPU parity is UNPINNED on real links and graph replay is opt-in; bench.py --gpus N
 * checks every rank's rows against a whole-domain run (fluid_checksum) before it times anything.
 * Contexts with pressure_groups exchange pressure rows between groups as well.
 *
 * The calls are collective in the SPMD sense: every slab makes the same fluid_* calls in the same
 * order (fluid_add_source, fluid_reset and fluid_write invalidate halos already exchanged; the
 * next fluid_step exchanges again).  Slab k talks to slabs k-1 and k+1 as ranks k-1 / k+1 of the
 * transport.
 *
 *   fluid_comm_unique_id   ncclGetUniqueId: 128 bytes for rank 0 to hand to the others (any way)
 *   fluid_comm_init_rccl   ncclCommInitRank(slab_count ranks, id, slab_rank) on the context's
 *                          device; collective; the communicator is destroyed with the context
 *   fluid_comm_attach_rccl an existing ncclComm_t whose rank / size are the slab's (caller-owned)
 *   fluid_comm_attach      any other transport: four callbacks, each returning 0 or an error;
 *                          one exchange = group_start, send + recv per neighbour and field,
 *                          group_end; pointers are device memory; `hip_stream` is the stream the
 *                          transfers must be ordered on (work enqueued on it before the call
 *                          precedes them, work enqueued after group_end follows them); peers are
 *                          slab ranks.  capturable != 0: the calls may be made under
 *                          hipStreamBeginCapture and become graph nodes
 *   fluid_comm_detach      drains both streams and drops the transport
 *   fluid_comm_refresh     halo rows of the current state in place without stepping
 * librccl is bound at run time (dlopen), so the library itself has no RCCL dependency. */
#define FLUID_COMM_ID_BYTES 128
typedef struct fluid_transport {
  void *user;
  int (*group_start)(void *user);
  int (*send)(void *user, const void *dev_src, size_t bytes, int peer, void *hip_stream);
  int (*recv)(void *user, void *dev_dst, size_t bytes, int peer, void *hip_stream);
  int (*group_end)(void *user, void *hip_stream);
  int capturable;
} fluid_transport;
typedef struct fluid_comm_stats {
  int attached;          /* a transport is attached                                      */
  int capturable;        /* it may be captured                                           */
  int graph;             /* whole steps are replayed as hipGraphs (0: enqueued directly) */
  int graphs_built;      /* graphs captured so far                                       */
  long long exchanges;   /* halo exchanges posted so far                                 */
  long long bytes_sent;  /* bytes sent so far                                            */
  char transport[16];    /* "rccl" / "caller"                                            */
  char note[160];        /* why graph capture was given up, if it was                    */
  int ranks;             /* ranks of the communicator as the transport itself reports them
                            (RCCL: ncclCommCount); 0 = the transport cannot say              */
  int reserved;
} fluid_comm_stats;
FLUID_API int fluid_comm_unique_id(void *id_out, size_t id_bytes);
FLUID_API int fluid_comm_init_rccl(fluid_ctx *ctx, const void *id, size_t id_bytes);
FLUID_API int fluid_comm_attach_rccl(fluid_ctx *ctx, void *nccl_comm);
FLUID_API int fluid_comm_attach(fluid_ctx *ctx, const fluid_transport *t);
FLUID_API int fluid_comm_detach(fluid_ctx *ctx);
/* Collective: the halo rows of the CURRENT state in place when the context's stream gets to
 * whatever is enqueued next (fluid_diagnostics across the seams, a caller's own kernels); posts
 * the exchange unless the last fluid_step already did, and makes the stream wait for it. */
FLUID_API int fluid_comm_refresh(fluid_ctx *ctx);
FLUID_API int fluid_comm_info(const fluid_ctx *ctx, fluid_comm_stats *out);
/* One-GPU check of the RCCL binding (a slab context on a single GPU has no neighbour to talk to):
 * a one-rank communicator on `device` sends `bytes` bytes to itself through the same calls the
 * slab step makes — eagerly, then (use_graph) captured as a hipGraph and replayed twice — and the
 * received bytes are verified on the host.  msg (may be NULL) receives what was done / what failed. */
FLUID_API int fluid_comm_selftest_rccl(int device, size_t bytes, int use_graph, char *msg, size_t msg_bytes);

/* ---- caller-provided stream ------------------------------------------------ */

/* The context's hipStream_t (as void*).  fluid_set_stream makes the context
 * enqueue on a caller-owned stream instead (e.g. torch's current stream, so
 * torch.distributed halo transfers are ordered with the kernels); NULL restores
 * the context's own stream.  Cached graphs are dropped. */
FLUID_API void *fluid_stream(fluid_ctx *ctx);
FLUID_API int fluid_set_stream(fluid_ctx *ctx, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_H */
