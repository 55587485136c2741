// fluid_kernels.hip — straight gfx950 kernels: one launch per launch of the
// reference's simulationStep() (/root/reference/FluidSimulation/kernel.cu:701-708).
//
// These are the parity baseline (FLUID_FLAG_UNFUSED) and the building blocks
// the temporally blocked kernels in fluid_kernels_tb.hip are checked against.
// Written for wave64: blocks are 64 lanes wide in x so every wave reads and
// writes whole contiguous row segments (float2: 512 B, float4: 1 KiB per wave
// instruction), four rows per block.
//
// Arithmetic contract: fp32 only, no FMA contraction, the
// reference's written summation order, bilinear in the author's mix() form
// x + (y - x) * a (kernel.cu:60-71), texture clamp-to-edge restated as index
// clamps.  With that contract every kernel here is bit-identical to the CPU
// oracle on the same input; only expf (add_force) may differ in the last ulp.
#include "fluid_internal.h"

#pragma clang fp contract(off)

const char *const kKernelNames[K_COUNT] = {
    "k_advect_vel", "k_diffuse",   "k_divergence",    "k_jacobi",     "k_project", "k_advect_dens",
    "k_add_force",  "k_vorticity", "k_add_vorticity", "k_diffuse_tb", "k_jacobi_tb", "k_jacobi_div_tb",
    "k_project_advect_dens", "k_diffuse_tile", "k_jacobi_tile", "k_jacobi_div_tile", "k_diffuse_div_tb", "k_jacobi0_tb"};

namespace {

constexpr int BX = 64;  // one wave wide
constexpr int BY = 4;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float2 neg(float2 v) { return make_float2(-v.x, -v.y); }

// Stores of a field that is written once and next read a whole pass later.  On a grid far beyond
// the caches (16384^2: 2-4 GB per field against 4 MB of L2 per XCD and 256 MB of MALL) a
// non-temporal store keeps those lines from evicting the p / velocity / density lines the
// neighbouring blocks are about to gather: k_project_advect_dens 2.47 -> 2.28 ms.  On grids whose
// fields fit the MALL the next kernel finds them there, so NT is chosen by size (fluid_api.hip).
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void store_out(float2 *q, float2 v) {
  if constexpr (NT) {
    f2v t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(q));
  } else {
    *q = v;
  }
}
template <bool NT> __device__ __forceinline__ void store_out(float4 *q, float4 v) {
  if constexpr (NT) {
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(q));
  } else {
    *q = v;
  }
}
// kernel.cu:60-71
__device__ __forceinline__ float mixf(float x, float y, float a) { return x + (y - x) * a; }
__device__ __forceinline__ float2 mix(float2 x, float2 y, float a) {
  return make_float2(mixf(x.x, y.x, a), mixf(x.y, y.y, a));
}
__device__ __forceinline__ float4 mix(float4 x, float4 y, float a) {
  return make_float4(mixf(x.x, y.x, a), mixf(x.y, y.y, a), mixf(x.z, y.z, a), mixf(x.w, y.w, a));
}

// Clamp-to-edge texel fetch in GLOBAL coordinates from a local (slab) buffer.
// A row outside the stored rows raises *flag (advection left the halo) and is
// clamped so the access stays in bounds.
struct Valid { int lo, hi; };  // local rows [lo, hi) that hold current data
struct Rows { int r0, r1; };   // a second row range for the blocks with blockIdx.z == 1

template <typename T>
__device__ __forceinline__ T texel(const T *t, Geom g, int i, int j, Valid v, int *flag) {
  i = clampi(i, 0, g.n - 1);
  int r = clampi(j, 0, g.ny - 1) - g.ybase;
  if (r < v.lo || r >= v.hi) {
    if (flag) *flag = 1;
    r = clampi(r, 0, g.L - 1);
  }
  return t[(size_t)r * g.n + i];
}

// LINEAR fetch at texture coordinate (bx + .5, by + .5): SURVEY.md Appendix B.2.  Split in two so
// that a wave can decide between the gathers below once the texel indices are known.
struct Tap { int i, j; float a, b; };
__device__ __forceinline__ Tap tap_of(float bx, float by) {
  const float fi = floorf(bx), fj = floorf(by);
  return Tap{(int)fi, (int)fj, bx - fi, by - fj};
}
template <typename T>
__device__ __forceinline__ T bilinear(const T *t, Geom g, Tap p, Valid v, int *flag) {
  T v0 = mix(texel(t, g, p.i, p.j, v, flag), texel(t, g, p.i + 1, p.j, v, flag), p.a);
  T v1 = mix(texel(t, g, p.i, p.j + 1, v, flag), texel(t, g, p.i + 1, p.j + 1, v, flag), p.a);
  return mix(v0, v1, p.b);
}
// All four texels inside the domain and inside the rows that hold current data: no clamps, no
// validity test, one address; the same four values, so the same result bit for bit.
__device__ __forceinline__ bool tap_inside(Tap p, Geom g, Valid v) {
  const int r = p.j - g.ybase;
  return p.i >= 0 && p.i <= g.n - 2 && p.j >= 0 && p.j <= g.ny - 2 && r >= v.lo && r + 1 < v.hi;
}
typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ float2 bilinear_inside(const float2 *t, Geom g, Tap p) {
  const float2 *q = t + ((size_t)(p.j - g.ybase) * g.n + p.i);
  const f4a8 lo = *reinterpret_cast<const f4a8 *>(q);         // texels (i, j), (i+1, j)
  const f4a8 hi = *reinterpret_cast<const f4a8 *>(q + g.n);   // texels (i, j+1), (i+1, j+1)
  const float2 v0 = mix(make_float2(lo.x, lo.y), make_float2(lo.z, lo.w), p.a);
  const float2 v1 = mix(make_float2(hi.x, hi.y), make_float2(hi.z, hi.w), p.a);
  return mix(v0, v1, p.b);
}
__device__ __forceinline__ float4 bilinear_inside(const float4 *t, Geom g, Tap p) {
  const float4 *q = t + ((size_t)(p.j - g.ybase) * g.n + p.i);
  const float4 t00 = q[0], t10 = q[1], t01 = q[g.n], t11 = q[g.n + 1];
  return mix(mix(t00, t10, p.a), mix(t01, t11, p.a), p.b);
}
// LINEAR fetch at INTEGER texture coordinates: Appendix B.3 (2x2 mean).
__device__ __forceinline__ float2 linear_at_int(const float2 *t, Geom g, int m, int k) {
  const Valid all{0, g.L};  // border rules reach at most two rows inward of a domain edge
  float2 v0 = mix(texel(t, g, m - 1, k - 1, all, nullptr), texel(t, g, m, k - 1, all, nullptr), 0.5f);
  float2 v1 = mix(texel(t, g, m - 1, k, all, nullptr), texel(t, g, m, k, all, nullptr), 0.5f);
  return mix(v0, v1, 0.5f);
}

#define CELL_PROLOGUE                                     \
  const int x = blockIdx.x * BX + threadIdx.x;            \
  const int r = r0 + blockIdx.y * BY + threadIdx.y;       \
  if (x >= g.n || r >= r1) return;                        \
  const int y = g.ybase + r;                              \
  const size_t id = (size_t)r * g.n + x;                  \
  const bool interior = x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1;

// Rows per thread in the two advection kernels.  A backtrace is two dependent memory round
// trips (own velocity, then the four texels), so a one-cell-per-thread kernel is bound by the
// number of waves in flight, not by bandwidth (measured 3.2 TB/s); each thread therefore
// issues the loads of ADV_ROWS cells before it consumes any of them.
#ifndef ADV_RV
#define ADV_RV 4
#endif
#ifndef ADV_RD
#define ADV_RD 1
#endif
// (Round 3: a thread walking down four groups of ADV_ROWS rows with the NEXT group's velocities loaded
// before it gathers for the current one — the first round trip of a backtrace overlapped with the second —
// is slower everywhere: 16384^2 872 -> 1014 us, 8192^2 246 -> 272, 4096^2 80 -> 88
// (profiles/r03/r3_adv_groups.log; bit-identical).  Four times fewer, four times longer workgroups lose
// more than the overlap gives.)
constexpr int ADV_ROWS_VEL = ADV_RV;   // measured: 1.33 -> 1.23 ms at 16384^2
constexpr int ADV_ROWS_DENS = ADV_RD;  // float4 texels: more rows cost occupancy (1.82 -> 2.08 ms with 4)

// Which tile a workgroup takes.  Workgroups go to the 8 XCDs round-robin by linear id, each XCD with an L2 of
// its own, so with the plain order every XCD touches every 512-byte piece of a row's tiles in turn and a tile's x
// neighbours (whose gathers overlap its own) are always in other L2s.  run > 0: every XCD takes runs of `run`
// consecutive tiles of the row-major walk instead — 64 KB of a row at run = 128 — within blocks of 8 * run ids, so
// the XCDs still walk the grid together (run < 0: one run each, eight bands); what is left over at the end keeps
// the plain order.  (Runs shaped as 2-D patches of tiles — y neighbours in the same L2 too — were measured: worse
// than the plain order at every size, profiles/r03/r3_xcd_patch1.log; it is the long contiguous pieces that pay.)
__device__ __forceinline__ uint2 tile_of_block(int run) {
  if (run == 0) return make_uint2(blockIdx.x, blockIdx.y);
  const unsigned gx = gridDim.x, total = gx * gridDim.y, id = blockIdx.y * gx + blockIdx.x;
  if (run < 0) run = total / 8 > 0 ? total / 8 : 1;  // one run per XCD: eight contiguous bands of the grid
  const unsigned span = 8u * run;
  if (id >= total / span * span) return make_uint2(blockIdx.x, blockIdx.y);
  const unsigned in = id % span, nid = id - in + (in % 8) * run + in / 8;
  return make_uint2(nid % gx, nid / gx);
}

#define ADVECT_PROLOGUE                                        \
  const uint2 tile = tile_of_block(xcd_run);                   \
  const int x = tile.x * BX + threadIdx.x;                     \
  const int rbase = r0 + (tile.y * BY + threadIdx.y) * ADV_ROWS; \
  if (x >= g.n) return;

// kernel.cu:114-140
// ADV_ROWS: rows per thread.  Four on the grids that matter for bandwidth; one on the smallest
// (up to ~1100^2: 768^2 8.9 -> 7.1 us, 1024^2 10.2 -> 9.6), where a launch is a wave or two of workgroups and what counts is how many
// lanes share the two dependent round trips (512^2, J = 10: 26.4 -> 25.0 us per step).
template <bool NT, int ADV_ROWS>
__global__ __launch_bounds__(BX *BY) void k_advect_vel(Geom g, int r0, int r1, Rows second, Valid valid, float dt,
                                                        const float2 *__restrict__ vin,
                                                        float2 *__restrict__ vout, int *flag, int xcd_run) {
  if (blockIdx.z) { r0 = second.r0; r1 = second.r1; }  // two row ranges in one launch (slab edges)
  ADVECT_PROLOGUE
  float2 v[ADV_ROWS];
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int r = min(rbase + k, r1 - 1);  // clamped duplicate rows are computed but not stored
    v[k] = vin[(size_t)r * g.n + x];
  }
  // Branch-free main path for every cell (the gather is in bounds for any cell thanks to the
  // clamps), so the texel loads of all ADV_ROWS cells are in flight together; border cells are
  // overwritten afterwards on a rarely taken path.  When every lane's four texels lie inside
  // the domain and the valid rows (all waves but the ones at a border or with a backtrace that
  // leaves the halo) the gather needs no clamps and two 16-byte loads per cell.
  float2 o[ADV_ROWS];
  Tap tp[ADV_ROWS];
  bool inside = true;
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int y = g.ybase + min(rbase + k, r1 - 1);
    float px = (float)x - dt * v[k].x;
    float py = (float)y - dt * v[k].y;
    px = fmaxf(0.0f, fminf(px, (float)g.n));
    py = fmaxf(0.0f, fminf(py, (float)g.ny));
    tp[k] = tap_of(px, py);
    inside = inside && tap_inside(tp[k], g, valid);
  }
  if (__all(inside)) {
#pragma unroll
    for (int k = 0; k < ADV_ROWS; ++k) o[k] = bilinear_inside(vin, g, tp[k]);
  } else {
#pragma unroll
    for (int k = 0; k < ADV_ROWS; ++k) {
      const int y = g.ybase + min(rbase + k, r1 - 1);
      const bool interior = x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1;
      o[k] = bilinear(vin, g, tp[k], valid, interior ? flag : nullptr);  // border cells: discarded
    }
  }
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int y = g.ybase + min(rbase + k, r1 - 1);
    const bool interior = x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1;
    if (!interior) {
      float2 b = make_float2(0.0f, 0.0f);
      if (x == 0) b = neg(linear_at_int(vin, g, x + 1, y));
      if (x == g.n - 1) b = neg(linear_at_int(vin, g, x - 1, y));
      if (y == 0) b = neg(linear_at_int(vin, g, x + 1, y + 1));
      if (y == g.ny - 1) b = neg(linear_at_int(vin, g, x + 1, y - 1));
      o[k] = b;
    }
  }
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k)
    if (rbase + k < r1) store_out<NT>(vout + (size_t)(rbase + k) * g.n + x, o[k]);
}

// kernel.cu:142-165
__global__ __launch_bounds__(BX *BY) void k_advect_dens(Geom g, int r0, int r1, Valid valid, float dt,
                                                         const float2 *__restrict__ vel,
                                                         const float4 *__restrict__ din,
                                                         float4 *__restrict__ dout, int *flag) {
  constexpr int ADV_ROWS = ADV_ROWS_DENS;
  constexpr int xcd_run = 0;
  ADVECT_PROLOGUE
  float2 v[ADV_ROWS];
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int r = min(rbase + k, r1 - 1);
    v[k] = vel[(size_t)r * g.n + x];
  }
  float4 o[ADV_ROWS];
  Tap tp[ADV_ROWS];
  bool inside = true;
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int y = g.ybase + min(rbase + k, r1 - 1);
    float px = (float)x - dt * v[k].x;
    float py = (float)y - dt * v[k].y;
    px = fmaxf(0.0f, fminf(px, (float)g.n));
    py = fmaxf(0.0f, fminf(py, (float)g.ny));
    tp[k] = tap_of(px, py);
    inside = inside && tap_inside(tp[k], g, valid);
  }
  if (__all(inside)) {
#pragma unroll
    for (int k = 0; k < ADV_ROWS; ++k) o[k] = bilinear_inside(din, g, tp[k]);
  } else {
#pragma unroll
    for (int k = 0; k < ADV_ROWS; ++k) {
      const int y = g.ybase + min(rbase + k, r1 - 1);
      const bool interior = x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1;
      o[k] = bilinear(din, g, tp[k], valid, interior ? flag : nullptr);
    }
  }
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k) {
    const int y = g.ybase + min(rbase + k, r1 - 1);
    if (!(x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1)) o[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
#pragma unroll
  for (int k = 0; k < ADV_ROWS; ++k)
    if (rbase + k < r1) dout[(size_t)(rbase + k) * g.n + x] = o[k];
}

// kernel.cu:173-205 (one sweep)
__global__ __launch_bounds__(BX *BY) void k_diffuse(Geom g, int r0, int r1, float alpha, float beta,
                                                     const float2 *__restrict__ vin,
                                                     float2 *__restrict__ vout) {
  CELL_PROLOGUE
  float2 v = vin[id];
  if (interior) {
    float2 vL = vin[id - 1], vR = vin[id + 1], vB = vin[id - g.n], vT = vin[id + g.n];
    float2 o;
    o.x = ((((vL.x + vR.x) + vB.x) + vT.x) + alpha * v.x) * beta;
    o.y = ((((vL.y + vR.y) + vB.y) + vT.y) + alpha * v.y) * beta;
    vout[id] = o;
  } else {
    vout[id] = v;
  }
}

// kernel.cu:265-289
__global__ __launch_bounds__(BX *BY) void k_divergence(Geom g, int r0, int r1,
                                                        const float2 *__restrict__ vel,
                                                        float *__restrict__ div) {
  CELL_PROLOGUE
  if (interior) {
    float2 vL = vel[id - 1], vR = vel[id + 1], vB = vel[id - g.n], vT = vel[id + g.n];
    div[id] = 0.5f * ((vR.x - vL.x) + (vT.y - vB.y));
  } else {
    div[id] = 0.0f;
  }
}

// kernel.cu:291-324 (one sweep)
__global__ __launch_bounds__(BX *BY) void k_jacobi(Geom g, int r0, int r1,
                                                    const float *__restrict__ pin,
                                                    const float *__restrict__ div,
                                                    float *__restrict__ pout) {
  CELL_PROLOGUE
  if (interior) {
    float vL = pin[id - 1], vR = pin[id + 1], vB = pin[id - g.n], vT = pin[id + g.n];
    float d = div[id];
    pout[id] = ((((vL + vR) + vB) + vT) + (-1.0f) * d) * 0.25f;
  } else {
    float o = 0.0f;
    if (x == 0) o = -pin[id + 1];
    if (x == g.n - 1) o = -pin[id - 1];
    if (y == 0) o = -pin[id + g.n];
    if (y == g.ny - 1) o = -pin[id - g.n];
    pout[id] = o;
  }
}

// kernel.cu:326-355
__device__ __forceinline__ float2 project_cell(Geom g, int x, int y, int r, size_t id, bool interior,
                                               const float2 *__restrict__ vin,
                                               const float *__restrict__ p) {
  if (interior) {
    float pL = p[id - 1], pR = p[id + 1], pB = p[id - g.n], pT = p[id + g.n];
    float2 v = vin[id];
    return make_float2(v.x - (pR - pL), v.y - (pT - pB));
  }
  float2 o = make_float2(0.0f, 0.0f);
  const int xp = min(x + 1, g.n - 1);  // clamp-to-edge of x+1 (kernel.cu:352-353)
  if (x == 0) o = neg(vin[id + 1]);
  if (x == g.n - 1) o = neg(vin[id - 1]);
  if (y == 0) o = neg(vin[(size_t)(r + 1) * g.n + xp]);
  if (y == g.ny - 1) o = neg(vin[(size_t)(r - 1) * g.n + xp]);
  return o;
}
__global__ __launch_bounds__(BX *BY) void k_project(Geom g, int r0, int r1,
                                                     const float2 *__restrict__ vin,
                                                     const float *__restrict__ p,
                                                     float2 *__restrict__ vout) {
  CELL_PROLOGUE
  vout[id] = project_cell(g, x, y, r, id, interior, vin, p);
}

// projection (kernel.cu:326-355) and density advection (kernel.cu:142-165) of the same cell in one
// pass: the density backtrace starts from the projected velocity of its own cell only, so the
// value just computed is used instead of being written out and read back (8 B/cell less).
#ifndef FUSED_BY
#define FUSED_BY 4
#endif
constexpr int TAIL_BY = FUSED_BY;  // rows per block of the fused kernel
template <bool NT>
__global__ __launch_bounds__(BX *TAIL_BY) void k_project_advect_dens(
    Geom g, int r0, int r1, Rows second, Valid valid, float dt, const float2 *__restrict__ vin,
    const float *__restrict__ p, float2 *__restrict__ vout, const float4 *__restrict__ din,
    float4 *__restrict__ dout, int *flag, int xcd_run) {
  if (blockIdx.z) { r0 = second.r0; r1 = second.r1; }  // two row ranges in one launch (slab edges)
  const uint2 tile = tile_of_block(xcd_run);
  const int x = tile.x * BX + threadIdx.x;
  const int r = r0 + tile.y * TAIL_BY + threadIdx.y;
  if (x >= g.n || r >= r1) return;
  const int y = g.ybase + r;
  const size_t id = (size_t)r * g.n + x;
  const bool interior = x > 0 && x < g.n - 1 && y > 0 && y < g.ny - 1;
  const float2 v = project_cell(g, x, y, r, id, interior, vin, p);
  store_out<NT>(vout + id, v);
  float px = (float)x - dt * v.x;
  float py = (float)y - dt * v.y;
  px = fmaxf(0.0f, fminf(px, (float)g.n));
  py = fmaxf(0.0f, fminf(py, (float)g.ny));
  const Tap tp = tap_of(px, py);
  float4 o;
  if (__all(tap_inside(tp, g, valid))) o = bilinear_inside(din, g, tp);
  else o = bilinear(din, g, tp, valid, interior ? flag : nullptr);
  if (!interior) o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  store_out<NT>(dout + id, o);
}

// kernel.cu:357-383 (every cell, border included)
__global__ __launch_bounds__(BX *BY) void k_add_force(Geom g, int r0, int r1, float dt, float radius,
                                                       float sx, float sy, float fx, float fy,
                                                       float4 rgba, float2 *__restrict__ vel,
                                                       float4 *__restrict__ dens) {
  CELL_PROLOGUE
  (void)interior;
  float dx = ((float)x / (float)g.n) - sx;
  float dy = ((float)y / (float)g.ny) - sy;
  float c = expf(-(dx * dx + dy * dy) / radius) * dt;
  float2 v = vel[id];
  vel[id] = make_float2(v.x + c * fx, v.y + c * fy);
  float4 d = dens[id];
  dens[id] = make_float4(d.x + c * rgba.x, d.y + c * rgba.y, d.z + c * rgba.z, d.w + c * rgba.w);
}

// kernel.cu:207-230
__global__ __launch_bounds__(BX *BY) void k_vorticity(Geom g, int r0, int r1,
                                                       const float2 *__restrict__ vel,
                                                       float *__restrict__ vort) {
  CELL_PROLOGUE
  if (interior) {
    float2 vL = vel[id - 1], vR = vel[id + 1], vB = vel[id - g.n], vT = vel[id + g.n];
    vort[id] = (vR.y - vL.y) - (vT.x - vB.x);
  } else {
    vort[id] = 0.0f;
  }
}

// kernel.cu:232-263
__global__ __launch_bounds__(BX *BY) void k_add_vorticity(Geom g, int r0, int r1, float dt,
                                                           const float *__restrict__ vort,
                                                           float2 *__restrict__ vel) {
  CELL_PROLOGUE
  if (!interior) return;
  float vL = vort[id - 1], vR = vort[id + 1], vB = vort[id - g.n], vT = vort[id + g.n];
  float gx = vR - vL, gy = vT - vB;
  float d = ((gx * gx + gy * gy) + 0.0f * 0.0f) + 0.0f * 0.0f;
  if (d) {
    float cx = gy * 1.0f - 0.0f * 0.0f;
    float cy = 0.0f * 0.0f - gx * 1.0f;
    float fx = 0.2f * cx, fy = 0.2f * cy;
    float2 v = vel[id];
    vel[id] = make_float2(v.x + fx * dt, v.y + fy * dt);
  }
}

// Halo blocks (whole rows, contiguous) to and from the staging buffers.  Not hipMemcpyAsync: its
// blit kernel has few, large workgroups, which are not dispatched while a long launch on another
// stream keeps every CU refilled (the pack then waited for the end of the interior
// launch it is meant to overlap).  Many 256-thread workgroups get in at once.
template <typename T>
__global__ __launch_bounds__(256) void k_copy_block(const T *__restrict__ src, T *__restrict__ dst, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = src[i];
}

inline dim3 grid_for(Geom g, int r0, int r1) {
  return dim3((g.n + BX - 1) / BX, (r1 - r0 + BY - 1) / BY);
}
inline dim3 grid_adv(Geom g, int r0, int r1, int rows_per_thread) {
  const int per_block = BY * rows_per_thread;
  return dim3((g.n + BX - 1) / BX, (r1 - r0 + per_block - 1) / per_block);
}
const dim3 kBlock(BX, BY);

}  // namespace

void launch_copy_block(hipStream_t s, void *dst, const void *src, size_t bytes) {
  if (!bytes) return;
  if (((uintptr_t)dst | (uintptr_t)src | bytes) % 16 == 0) {
    const size_t count = bytes / 16;
    hipLaunchKernelGGL(k_copy_block<uint4>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                       (const uint4 *)src, (uint4 *)dst, count);
  } else {  // every field element is a multiple of 4 bytes
    const size_t count = bytes / 4;
    hipLaunchKernelGGL(k_copy_block<unsigned>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                       (const unsigned *)src, (unsigned *)dst, count);
  }
}
// r0b..r1b: an optional second row range handled by the same launch (the two edges of a slab:
// two launches of ~10 us each are mostly ramp-up and drain).
void launch_advect_vel(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                       const float2 *vin, float2 *vout, int *flag, bool nt, int xcd_run, int r0b, int r1b) {
  if (r1 <= r0) { r0 = r0b; r1 = r1b; r0b = r1b = 0; }
  if (r1 <= r0) return;
  const bool two = r1b > r0b;
  const bool small = (long long)g.n * g.L <= 5 * (1ll << 18);  // see k_advect_vel
  const int rows_per_thread = small ? 1 : ADV_ROWS_VEL;
  dim3 grid = grid_adv(g, r0, r1, rows_per_thread);
  if (two) {
    const dim3 gb = grid_adv(g, r0b, r1b, rows_per_thread);
    grid.y = grid.y > gb.y ? grid.y : gb.y;
    grid.z = 2;
  }
  if (small)
    hipLaunchKernelGGL((k_advect_vel<false, 1>), grid, kBlock, 0, s, g, r0, r1, Rows{r0b, r1b}, Valid{vlo, vhi}, dt,
                       vin, vout, flag, xcd_run);
  else if (nt)
    hipLaunchKernelGGL((k_advect_vel<true, ADV_ROWS_VEL>), grid, kBlock, 0, s, g, r0, r1, Rows{r0b, r1b},
                       Valid{vlo, vhi}, dt, vin, vout, flag, xcd_run);
  else
    hipLaunchKernelGGL((k_advect_vel<false, ADV_ROWS_VEL>), grid, kBlock, 0, s, g, r0, r1, Rows{r0b, r1b},
                       Valid{vlo, vhi}, dt, vin, vout, flag, xcd_run);
}
void launch_diffuse(hipStream_t s, Geom g, int r0, int r1, float alpha, float beta,
                    const float2 *vin, float2 *vout) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_diffuse, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, alpha, beta, vin, vout);
}
void launch_divergence(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, float *div) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_divergence, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, vel, div);
}
void launch_jacobi(hipStream_t s, Geom g, int r0, int r1, const float *pin, const float *div,
                   float *pout) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_jacobi, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, pin, div, pout);
}
void launch_project(hipStream_t s, Geom g, int r0, int r1, const float2 *vin, const float *p,
                    float2 *vout) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_project, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, vin, p, vout);
}
void launch_project_advect_dens(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                                const float2 *vin, const float *p, float2 *vout, const float4 *din,
                                float4 *dout, int *flag, bool nt, int xcd_run, int r0b, int r1b) {
  if (r1 <= r0) { r0 = r0b; r1 = r1b; r0b = r1b = 0; }
  if (r1 <= r0) return;
  const int rows = r1b - r0b > r1 - r0 ? r1b - r0b : r1 - r0;
  const dim3 grid((g.n + BX - 1) / BX, (rows + TAIL_BY - 1) / TAIL_BY, r1b > r0b ? 2 : 1), block(BX, TAIL_BY);
  if (nt)
    hipLaunchKernelGGL(k_project_advect_dens<true>, grid, block, 0, s, g, r0, r1, Rows{r0b, r1b}, Valid{vlo, vhi}, dt,
                       vin, p, vout, din, dout, flag, xcd_run);
  else
    hipLaunchKernelGGL(k_project_advect_dens<false>, grid, block, 0, s, g, r0, r1, Rows{r0b, r1b}, Valid{vlo, vhi}, dt,
                       vin, p, vout, din, dout, flag, xcd_run);
}
void launch_advect_dens(hipStream_t s, Geom g, int r0, int r1, int vlo, int vhi, float dt,
                        const float2 *vel, const float4 *din, float4 *dout, int *flag) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_advect_dens, grid_adv(g, r0, r1, ADV_ROWS_DENS), kBlock, 0, s, g, r0, r1, Valid{vlo, vhi}, dt,
                     vel, din, dout, flag);
}
void launch_add_force(hipStream_t s, Geom g, int r0, int r1, float dt, float radius, float sx,
                      float sy, float fx, float fy, float4 rgba, float2 *vel, float4 *dens) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_add_force, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, dt, radius, sx, sy,
                     fx, fy, rgba, vel, dens);
}
void launch_vorticity(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, float *vort) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_vorticity, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, vel, vort);
}
void launch_add_vorticity(hipStream_t s, Geom g, int r0, int r1, float dt, const float *vort,
                          float2 *vel) {
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_add_vorticity, grid_for(g, r0, r1), kBlock, 0, s, g, r0, r1, dt, vort, vel);
}
