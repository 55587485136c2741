// fluid_kernels_tb.hip — temporally blocked 5-point sweeps for gfx950.
//
// The reference runs its two relaxation loops as 10 + 10 separate full-grid
// launches (/root/reference/FluidSimulation/kernel.cu:584-601 diffusion,
// :620-641 pressureJacobi), i.e. one HBM round trip per sweep.  Here K sweeps
// are done per round trip.
//
// Scheme (per wave64, no LDS, no barriers): a wave owns a vertical strip of
// 64*C columns (C consecutive columns per lane) and streams down a chunk of
// rows.  Level 0 is the input field; level l+1 is the field after l+1 sweeps.
// When level-l row t-l arrives, level l+1 row t-l-1 can be produced from the
// level-l rows (t-l-2, t-l-1, t-l) kept in registers — a K-deep software
// pipeline in registers.  x-neighbours that live in the adjacent lane come
// through DPP wave shifts (v_*_dpp wave_shr:1 / wave_shl:1), so a sweep costs
// VALU only.  Cells within K of a strip/chunk edge that is not a domain border
// are recomputed by the neighbouring strip/chunk (halo overlap); at a domain
// border the reference's boundary rule reads inward only, so nothing is lost.
//
// Per-cell operation order is the reference's, and FMA contraction is off, so
// K fused sweeps are bit-identical to K launches of the straight kernels.
#include "fluid_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;

// value held by lane-1 / lane+1 (DPP wave shift; every lane must be active)
__device__ __forceinline__ float from_left(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));
}

template <int C> struct Row { float v[C]; };

template <int C>
__device__ __forceinline__ Row<C> zero_row() {
  Row<C> r;
#pragma unroll
  for (int c = 0; c < C; ++c) r.v[c] = 0.0f;
  return r;
}

// Vector load/store of C consecutive floats (16-byte for C=4, 8-byte for C=2).
template <int C>
__device__ __forceinline__ Row<C> load_row(const float *p) {
  Row<C> r;
  if constexpr (C == 4) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (C == 2) {
    float2 t = *reinterpret_cast<const float2 *>(p);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int C>
__device__ __forceinline__ void store_row(float *p, const Row<C> &r) {
  if constexpr (C == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else if constexpr (C == 2) {
    *reinterpret_cast<float2 *>(p) = make_float2(r.v[0], r.v[1]);
  } else {
    *p = r.v[0];
  }
}

template <int K, int C> struct StripGeom {
  static constexpr int HX = ((K + C - 1) / C) * C;  // x halo, rounded to the vector width
  static constexpr int VW = WAVE * C - 2 * HX;      // columns a strip produces
  static_assert(VW > 0, "strip too narrow for this K");
};

// ---------------------------------------------------------------------------
// pressureJacobi x K   (kernel.cu:291-324)
// ---------------------------------------------------------------------------
template <int C, bool XEDGE>
__device__ __forceinline__ Row<C> jacobi_row(const Row<C> &a, const Row<C> &b, const Row<C> &n,
                                             const Row<C> &d, int r, int x0, int N) {
  Row<C> o;
  if (r <= 0 || r >= N - 1) {  // wave-uniform
#pragma unroll
    for (int c = 0; c < C; ++c) o.v[c] = (r == 0) ? -n.v[c] : -a.v[c];  // kernel.cu:321-322
    return o;
  }
  const float lft = from_left(b.v[C - 1]);
  const float rgt = from_right(b.v[0]);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float vL = c > 0 ? b.v[c - 1] : lft;
    const float vR = c < C - 1 ? b.v[c + 1] : rgt;
    float s = ((((vL + vR) + a.v[c]) + n.v[c]) + (-1.0f) * d.v[c]) * 0.25f;  // kernel.cu:313
    if (XEDGE) {
      const int x = x0 + c;
      if (x == 0) s = -vR;      // kernel.cu:319
      if (x == N - 1) s = -vL;  // kernel.cu:320
    }
    o.v[c] = s;
  }
  return o;
}

template <int K, int C, bool XEDGE>
__device__ __forceinline__ void jacobi_stream(Geom g, int ya, int yb, int x0, bool in_dom,
                                              bool st_ok, const float *__restrict__ pin,
                                              const float *__restrict__ div,
                                              float *__restrict__ pout) {
  Row<C> a[K], b[K], dv[K];
#pragma unroll
  for (int l = 0; l < K; ++l) { a[l] = zero_row<C>(); b[l] = zero_row<C>(); dv[l] = zero_row<C>(); }
  const int N = g.n;
  auto row_ok = [&](int t) { return in_dom && t >= 0 && t < N && t - g.ybase >= 0 && t - g.ybase < g.L; };
  auto ld = [&](const float *f, int t) {
    return row_ok(t) ? load_row<C>(f + (size_t)(t - g.ybase) * N + x0) : zero_row<C>();
  };
  const int t0 = ya - K, t1 = yb + K;
  Row<C> cur_p = ld(pin, t0), cur_d = ld(div, t0);
  Row<C> nxt_p = ld(pin, t0 + 1), nxt_d = ld(div, t0 + 1);
  Row<C> prev_d = zero_row<C>();
  for (int t = t0; t < t1; ++t) {
    Row<C> ld_p = ld(pin, t + 2), ld_d = ld(div, t + 2);  // prefetch two rows ahead
#pragma unroll
    for (int l = K - 1; l > 0; --l) dv[l] = dv[l - 1];
    dv[0] = prev_d;  // div row t-1
    Row<C> n = cur_p;  // level 0, row t
#pragma unroll
    for (int l = 0; l < K; ++l) {
      Row<C> o = jacobi_row<C, XEDGE>(a[l], b[l], n, dv[l], t - l - 1, x0, N);
      a[l] = b[l];
      b[l] = n;
      n = o;
    }
    const int ro = t - K;  // level K row just produced
    if (st_ok && ro >= ya && ro < yb) store_row<C>(pout + (size_t)(ro - g.ybase) * N + x0, n);
    prev_d = cur_d;
    cur_p = nxt_p; cur_d = nxt_d;
    nxt_p = ld_p; nxt_d = ld_d;
  }
}

template <int K, int C>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_jacobi_tb(
    Geom g, int r0, int r1, int chunk, const float *__restrict__ pin,
    const float *__restrict__ div, float *__restrict__ pout) {
  using SG = StripGeom<K, C>;
  const int lane = threadIdx.x & (WAVE - 1);
  const int strip = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int xs = strip * SG::VW;
  if (xs >= g.n) return;  // whole wave
  const int x0 = xs - SG::HX + lane * C;
  const bool in_dom = x0 >= 0 && x0 < g.n;
  const bool st_ok = in_dom && lane * C >= SG::HX && lane * C < SG::HX + SG::VW;
  const int ya = g.ybase + r0 + blockIdx.y * chunk;
  const int yb = min(ya + chunk, g.ybase + r1);
  const bool xedge = xs == 0 || xs + SG::VW + SG::HX >= g.n;  // wave-uniform
  if (xedge)
    jacobi_stream<K, C, true>(g, ya, yb, x0, in_dom, st_ok, pin, div, pout);
  else
    jacobi_stream<K, C, false>(g, ya, yb, x0, in_dom, st_ok, pin, div, pout);
}

// ---------------------------------------------------------------------------
// diffusion x K   (kernel.cu:173-205), float2 field: two planes per column
// ---------------------------------------------------------------------------
template <int C> struct Row2 { float u[C], w[C]; };  // (x, y) components of C columns

template <int C>
__device__ __forceinline__ Row2<C> zero_row2() {
  Row2<C> r;
#pragma unroll
  for (int c = 0; c < C; ++c) { r.u[c] = 0.0f; r.w[c] = 0.0f; }
  return r;
}
template <int C>
__device__ __forceinline__ Row2<C> load_row2(const float2 *p) {
  Row2<C> r;
  if constexpr (C == 2) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    r.u[0] = t.x; r.w[0] = t.y; r.u[1] = t.z; r.w[1] = t.w;
  } else {
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      float4 t = reinterpret_cast<const float4 *>(p)[c / 2];
      r.u[c] = t.x; r.w[c] = t.y; r.u[c + 1] = t.z; r.w[c + 1] = t.w;
    }
  }
  return r;
}
template <int C>
__device__ __forceinline__ void store_row2(float2 *p, const Row2<C> &r) {
#pragma unroll
  for (int c = 0; c < C; c += 2)
    reinterpret_cast<float4 *>(p)[c / 2] = make_float4(r.u[c], r.w[c], r.u[c + 1], r.w[c + 1]);
}

template <int C, bool XEDGE>
__device__ __forceinline__ Row2<C> diffuse_row(const Row2<C> &a, const Row2<C> &b, const Row2<C> &n,
                                               int r, int x0, int N, float alpha, float beta) {
  if (r <= 0 || r >= N - 1) return b;  // wave-uniform; border copies through, kernel.cu:201-203
  Row2<C> o;
  const float lu = from_left(b.u[C - 1]), lw = from_left(b.w[C - 1]);
  const float ru = from_right(b.u[0]), rw = from_right(b.w[0]);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float uL = c > 0 ? b.u[c - 1] : lu, wL = c > 0 ? b.w[c - 1] : lw;
    const float uR = c < C - 1 ? b.u[c + 1] : ru, wR = c < C - 1 ? b.w[c + 1] : rw;
    float su = ((((uL + uR) + a.u[c]) + n.u[c]) + alpha * b.u[c]) * beta;  // kernel.cu:195
    float sw = ((((wL + wR) + a.w[c]) + n.w[c]) + alpha * b.w[c]) * beta;
    if (XEDGE) {
      const int x = x0 + c;
      if (x == 0 || x == N - 1) { su = b.u[c]; sw = b.w[c]; }
    }
    o.u[c] = su;
    o.w[c] = sw;
  }
  return o;
}

template <int K, int C, bool XEDGE>
__device__ __forceinline__ void diffuse_stream(Geom g, int ya, int yb, int x0, bool in_dom,
                                               bool st_ok, float alpha, float beta,
                                               const float2 *__restrict__ vin,
                                               float2 *__restrict__ vout) {
  Row2<C> a[K], b[K];
#pragma unroll
  for (int l = 0; l < K; ++l) { a[l] = zero_row2<C>(); b[l] = zero_row2<C>(); }
  const int N = g.n;
  auto ld = [&](int t) {
    const bool ok = in_dom && t >= 0 && t < N && t - g.ybase >= 0 && t - g.ybase < g.L;
    return ok ? load_row2<C>(vin + (size_t)(t - g.ybase) * N + x0) : zero_row2<C>();
  };
  const int t0 = ya - K, t1 = yb + K;
  Row2<C> cur = ld(t0), nxt = ld(t0 + 1);
  for (int t = t0; t < t1; ++t) {
    Row2<C> pre = ld(t + 2);
    Row2<C> n = cur;
#pragma unroll
    for (int l = 0; l < K; ++l) {
      Row2<C> o = diffuse_row<C, XEDGE>(a[l], b[l], n, t - l - 1, x0, N, alpha, beta);
      a[l] = b[l];
      b[l] = n;
      n = o;
    }
    const int ro = t - K;
    if (st_ok && ro >= ya && ro < yb) store_row2<C>(vout + (size_t)(ro - g.ybase) * N + x0, n);
    cur = nxt;
    nxt = pre;
  }
}

template <int K, int C>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_diffuse_tb(
    Geom g, int r0, int r1, int chunk, float alpha, float beta, const float2 *__restrict__ vin,
    float2 *__restrict__ vout) {
  using SG = StripGeom<K, C>;
  const int lane = threadIdx.x & (WAVE - 1);
  const int strip = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int xs = strip * SG::VW;
  if (xs >= g.n) return;
  const int x0 = xs - SG::HX + lane * C;
  const bool in_dom = x0 >= 0 && x0 < g.n;
  const bool st_ok = in_dom && lane * C >= SG::HX && lane * C < SG::HX + SG::VW;
  const int ya = g.ybase + r0 + blockIdx.y * chunk;
  const int yb = min(ya + chunk, g.ybase + r1);
  const bool xedge = xs == 0 || xs + SG::VW + SG::HX >= g.n;
  if (xedge)
    diffuse_stream<K, C, true>(g, ya, yb, x0, in_dom, st_ok, alpha, beta, vin, vout);
  else
    diffuse_stream<K, C, false>(g, ya, yb, x0, in_dom, st_ok, alpha, beta, vin, vout);
}

template <int K, int C>
dim3 tb_grid(Geom g, int r0, int r1, int chunk) {
  using SG = StripGeom<K, C>;
  const int strips = (g.n + SG::VW - 1) / SG::VW;
  return dim3((strips + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, (r1 - r0 + chunk - 1) / chunk);
}

template <int K, int C>
void jacobi_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, const float *pin,
                      const float *div, float *pout) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk);
  k_jacobi_tb<K, C><<<grid, dim3(WAVE * WAVES_PER_BLOCK), 0, s>>>(g, r0, r1, chunk, pin, div, pout);
}
template <int K, int C>
void diffuse_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, float alpha, float beta,
                       const float2 *vin, float2 *vout) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk);
  k_diffuse_tb<K, C><<<grid, dim3(WAVE * WAVES_PER_BLOCK), 0, s>>>(g, r0, r1, chunk, alpha, beta, vin,
                                                                vout);
}

}  // namespace

#define TB_DISPATCH_K(FN, C_, ...)                    \
  switch (k) {                                        \
    case 1: FN<1, C_>(__VA_ARGS__); return true;      \
    case 2: FN<2, C_>(__VA_ARGS__); return true;      \
    case 4: FN<4, C_>(__VA_ARGS__); return true;      \
    case 5: FN<5, C_>(__VA_ARGS__); return true;      \
    case 8: FN<8, C_>(__VA_ARGS__); return true;      \
    case 10: FN<10, C_>(__VA_ARGS__); return true;    \
    default: return false;                            \
  }

bool tb_supported(int n, int cols) { return (cols == 2 || cols == 4) && n % cols == 0 && n >= 8; }

bool launch_jacobi_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                      const float *pin, const float *div, float *pout) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1) return false;
  if (cols == 4) { TB_DISPATCH_K(jacobi_tb_launch, 4, s, g, r0, r1, chunk, pin, div, pout) }
  TB_DISPATCH_K(jacobi_tb_launch, 2, s, g, r0, r1, chunk, pin, div, pout)
}

bool launch_diffuse_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                       float alpha, float beta, const float2 *vin, float2 *vout) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1) return false;
  if (cols == 4) { TB_DISPATCH_K(diffuse_tb_launch, 4, s, g, r0, r1, chunk, alpha, beta, vin, vout) }
  TB_DISPATCH_K(diffuse_tb_launch, 2, s, g, r0, r1, chunk, alpha, beta, vin, vout)
}
