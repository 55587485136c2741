// fluid_internal.h — shared between the kernel TU and the C-ABI host TU.
// Product code (gfx950 only).  Nothing here touches oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// Geometry of one context's local buffers.  The domain is n x n; the context
// stores rows [ybase, ybase + L) (its owned rows plus halo rows); local row r
// is global row ybase + r.  Whole-domain contexts have ybase = 0, L = n.
struct Geom {
  int n;      // grid width: columns, and the row pitch of every field (the reference's gridResolution)
  int ybase;  // global y of local row 0
  int L;      // local rows stored
  int ny;     // grid height: rows of the whole domain (== n for the reference's square grids)
};

// Kernel ids for per-kernel timing (fluid_profile_steps) and rocprof matching.
enum KernelId {
  K_ADVECT_VEL = 0,
  K_DIFFUSE,
  K_DIVERGENCE,
  K_JACOBI,
  K_PROJECT,
  K_ADVECT_DENS,
  K_ADD_FORCE,
  K_VORTICITY,
  K_ADD_VORTICITY,
  K_DIFFUSE_TB,   // temporally blocked diffusion
  K_JACOBI_TB,    // temporally blocked Jacobi
  K_JACOBI_DIV_TB,  // divergence + pressure clear + first Jacobi launch
  K_PROJECT_ADVECT_DENS,  // gradient subtraction + density advection in one pass
  K_DIFFUSE_TILE,         // LDS-tiled sweeps for small grids (fluid_kernels_tile.hip)
  K_JACOBI_TILE,
  K_JACOBI_DIV_TILE,
  K_DIFFUSE_DIV_TB,  // the last diffusion launch of a whole step, also writing the divergence
  K_JACOBI0_TB,      // the first pressure launch after it: from p = 0, the divergence read from memory
  K_COUNT
};
extern const char *const kKernelNames[K_COUNT];

// ---- straight (one launch per reference launch) kernels ---------------------
// All take local row range [r0, r1) to compute; reads reach one row beyond
// (stencils) or up to the advection halo (gathers).
// [vlo, vhi): local rows of the gathered field that hold current data; a
// backtrace outside raises *halo_flag.
// nt: non-temporal stores of the output (grids far beyond the caches, see fluid_kernels.hip)
// xcd_run: 0, or the run of consecutive tiles each XCD takes (tile_of_block in fluid_kernels.hip)
void launch_copy_block(hipStream_t s, void *dst, const void *src, size_t bytes);  // bytes % 4 == 0
void launch_advect_vel(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                       const float2 *vin, float2 *vout, int *halo_flag, bool nt, int xcd_run = 0, int r0b = 0, int r1b = 0);
void launch_diffuse(hipStream_t s, Geom g, int r0, int r1, float alpha, float beta,
                    const float2 *vin, float2 *vout);
void launch_divergence(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, float *div);
void launch_jacobi(hipStream_t s, Geom g, int r0, int r1, const float *pin, const float *div,
                   float *pout);
void launch_project(hipStream_t s, Geom g, int r0, int r1, const float2 *vin, const float *p,
                    float2 *vout);
void launch_project_advect_dens(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                                const float2 *vin, const float *p, float2 *vout, const float4 *din,
                                float4 *dout, int *flag, bool nt, int xcd_run = 0, int r0b = 0, int r1b = 0);
void launch_advect_dens(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                        const float2 *vel, const float4 *din, float4 *dout, int *halo_flag);
void launch_add_force(hipStream_t s, Geom g, int r0, int r1, float dt, float radius, float sx,
                      float sy, float fx, float fy, float4 rgba, float2 *vel, float4 *dens);
void launch_vorticity(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, float *vort);
void launch_add_vorticity(hipStream_t s, Geom g, int r0, int r1, float dt, const float *vort,
                          float2 *vel);

// ---- temporally blocked sweeps (fluid_kernels_tb.hip) -----------------------
// k sweeps per launch; cols = columns per lane (2 or 4); chunk = rows a wave
// streams.  Return false when the shape is unsupported (caller falls back to
// the straight kernels): n % cols != 0 or k not instantiated.  dump: one row of
// n * 16 bytes of device memory that receives the rows a wave computes but must not store.
// Chained chunks: the rows a chunk hands to the chunk above it through memory instead of both
// computing them (fluid_kernels_tb.hip, Pipe::run).  Device memory of the context: side records
// (8-byte words), one flag per record (zero between launches), a status word (non-zero: a wave gave up
// waiting for its neighbour's rows).
// trace (FLUID_TB_TRACE, diagnostics): null, or 4 shader-clock stamps per record — start, fill done,
// tail begins, end — of the last chained launch.
// sync: the four waves (strips) of a workgroup meet at a barrier every three rows (Pipe::run)
struct TbChain { unsigned long long *side; int *flags; int *status; unsigned long long *trace; int sync; };
bool tb_chain_chunk_ok(int k, int cols, int chunk, int rows);
int tb_chain_min_chunk(int k);
int tb_chain_chunk_multiple();
size_t tb_chain_words(int n, int k, int cols, bool float2_field, int chunks);
size_t tb_chain_flags(int n, int k, int cols, int chunks);
bool tb_supported(int n, int cols);
int tb_resident_blocks(int kind, int k, int cols);  // kind: 0 pressure, 1 divergence + pressure, 2 diffusion, 3 diffusion + divergence, 4 pressure from p = 0
int tb_blocks_per_row(int n, int k, int cols);
bool tb_has_k(int k, int cols, bool float2_field);
bool launch_jacobi_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                      const float *pin, const float *div, float *pout, void *dump, const TbChain *chained = nullptr);
// divergence + p^0 = 0 + the first k Jacobi sweeps in one launch (writes div and p)
bool launch_jacobi_div_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                          const float2 *vel, float *div_out, float *pout, void *dump, const TbChain *chained = nullptr);
bool launch_diffuse_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                       float alpha, float beta, const float2 *vin, float2 *vout, void *dump,
                       const TbChain *chained = nullptr);

// the last diffusion launch of a whole step, also writing the divergence of the field it stores, and the first
// pressure launch after it (k sweeps from p^0 == 0, the divergence read from memory); strips of the former are
// laid out for k + 1 sweeps (tb_blocks_per_row(n, k + 1, cols))
bool tb_has_div_k(int k, int cols);
bool launch_diffuse_div_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk, float alpha, float beta,
                           const float2 *vin, float2 *vout, float *div_out, void *dump, const TbChain *extra = nullptr);
bool launch_jacobi0_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk, const float *div,
                       float *pout, void *dump, const TbChain *extra = nullptr);

// ---- LDS-tiled sweeps for small grids (fluid_kernels_tile.hip) ---------------
// k sweeps per launch on local rows [r0, r1); false when k is not instantiated.
// ty: rows a workgroup stores (16 or 32, tile_has_ty).
bool tile_has_k(int k, bool float2_field);
bool tile_has_ty(int ty);
bool launch_jacobi_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, const float *pin, const float *div,
                        float *pout);
bool launch_jacobi_div_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, const float2 *vel, float *div_out,
                            float *pout);
// launch_jacobi_tile with pin == nullptr: the first launch of a projection whose divergence is in
// memory already (p^0 == 0).  launch_diffuse_tile with div_out: also writes the divergence of the
// field it stores (the last diffusion launch of a whole step).
bool launch_diffuse_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, float alpha, float beta,
                         const float2 *vin, float2 *vout, float *div_out = nullptr);

// ---- diagnostics (fluid_kernels_diag.hip) -----------------------------------
// Maxima (absolute values) and sums over local rows [r0, r1); rows r0-1 and r1 must hold current
// velocity and pressure where they are inside the domain.  partial: DIAG_MAX_BLOCKS * DIAG_NREC
// floats of device scratch; out: device record, valid when the stream has drained.
constexpr int DIAG_NMAX = 6;  // |u|, |v|, |div buffer|, |p|, Jacobi update |p' - p|, |div of the velocity|
constexpr int DIAG_NSUM = 4;  // dye channels
constexpr int DIAG_NREC = DIAG_NMAX + DIAG_NSUM;
constexpr int DIAG_MAX_BLOCKS = 2048;
struct DiagResult { float mx[DIAG_NMAX]; double sm[DIAG_NSUM]; };
void launch_diagnostics(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, const float4 *dens,
                        const float *p, const float *div, float *partial, DiagResult *out);

void launch_checksum(hipStream_t s, const void *base, size_t words, unsigned long long *out);

// ---- halo transport (fluid_comm.hip; include/fluid.h fluid_comm_*) -----------------------------
// One exchange = group_start, then a send and a recv per neighbour and field, then group_end, all
// on one stream; every call returns 0 or a transport error (text from `error`).  Buffers are device
// memory (the field rows themselves: halo blocks are whole rows, hence contiguous).
struct Transport {
  void *user = nullptr;
  int (*group_start)(void *user) = nullptr;
  int (*send)(void *user, const void *dev_src, size_t bytes, int peer, void *hip_stream) = nullptr;
  int (*recv)(void *user, void *dev_dst, size_t bytes, int peer, void *hip_stream) = nullptr;
  int (*group_end)(void *user, void *hip_stream) = nullptr;
  const char *(*error)(void *user) = nullptr;   // may be null
  int (*ranks)(void *user) = nullptr;           // may be null: ranks of the communicator behind it
  void (*destroy)(void *user) = nullptr;        // may be null
  bool capturable = false;                      // the calls may be made under stream capture
  const char *name = "";
};
bool rccl_unique_id(void *out, size_t bytes, char *err, size_t errlen);
bool rccl_transport_init(const void *id_bytes, size_t bytes, int rank, int nranks, Transport *out, char *err,
                         size_t errlen);
bool rccl_transport_from_comm(void *comm, int rank, int nranks, Transport *out, char *err, size_t errlen);
bool rccl_selftest(int device, size_t bytes, bool use_graph, char *msg, size_t msglen);
