// fluid_kernels_tile.hip — LDS-tiled multi-sweep relaxation for SMALL grids (gfx950).
//
// The register pipeline of fluid_kernels_tb.hip streams rows serially inside a wave; on a grid of
// a million cells there is too little work to hide that serial chain and a launch of K sweeps
// costs about 2.8 us per sweep whatever K is.  Here a 256-thread workgroup keeps a (64 + 2K) x
// (32 + 2K) tile of the field in LDS (160 KB per CU: three or four such workgroups fit), every
// thread owns a few 4-cell row segments of it for all K sweeps (16-byte LDS reads of the rows
// above and below, one barrier per sweep), and the 64 x 32 centre — the cells no tile-edge error
// has reached — is written back.  Same per-cell rules, operation order and
// boundary handling as the straight kernels (kernel.cu:173-205 diffusion, :265-289 divergence,
// :291-324 pressureJacobi), FMA contraction off: K tile sweeps are bit-identical to K launches.
#include "fluid_internal.h"

#pragma clang fp contract(off)

namespace {

#ifndef TILE_THREADS
#define TILE_THREADS 1024
#endif
constexpr int TX = 64, TY = 32, THREADS = TILE_THREADS;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Tile geometry: K halo rows above and below, HX = K rounded up to even halo columns left and
// right, so that a tile row is a whole number of 4-float segments (16-byte LDS accesses).
template <int K> struct TileGeom {
  static constexpr int HX = (K + 1) & ~1;
  static constexpr int W = TX + 2 * HX, H = TY + 2 * K, W4 = W / 4;
  static constexpr int SEGS = (H * W4 + THREADS - 1) / THREADS;  // segments a thread owns
  static_assert(W % 4 == 0, "tile rows are made of float4 segments");
};

// What a thread knows about one of the 4-cell row segments it owns for all K sweeps.
struct Seg {
  int c;       // LDS index of its first cell (ly * W + 4 * sx); -1: not a segment / tile edge row
  int x, y;    // global coordinates of its first cell
};

// The interior rule of one plane on a segment: rm / rp = rows y-1 / y+1, ce = the segment,
// l / r = the cells left and right of it.
struct F4 { float v[4]; };
__device__ __forceinline__ F4 ld4(const float *p) {
  const float4 t = *reinterpret_cast<const float4 *>(p);
  return F4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void st4(float *p, const F4 &f) {
  *reinterpret_cast<float4 *>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
}

// pressureJacobi, kernel.cu:291-324: interior rule :313, border rules :319-322 (y rules last)
struct JacobiRule {
  static __device__ __forceinline__ float interior(float L, float R, float B, float T, float, float d, float, float) {
    return ((((L + R) + B) + T) + (-1.0f) * d) * 0.25f;
  }
  static __device__ __forceinline__ float border(float L, float R, float B, float T, float, int x, int y, int n) {
    float o = 0.0f;
    if (x == 0) o = -R;
    if (x == n - 1) o = -L;
    if (y == 0) o = -T;
    if (y == n - 1) o = -B;
    return o;
  }
};
// diffusion, kernel.cu:173-205: interior rule :195, the border copies through :201-203
struct DiffuseRule {
  static __device__ __forceinline__ float interior(float L, float R, float B, float T, float c, float, float alpha, float beta) {
    return ((((L + R) + B) + T) + alpha * c) * beta;
  }
  static __device__ __forceinline__ float border(float, float, float, float, float c, int, int, int) { return c; }
};

// One sweep of one plane over the segments this thread owns: a -> b.  `aux` = per-segment
// constant (the divergence), `edge`: the tile touches a domain border (block-uniform).
template <int K, class Rule>
__device__ __forceinline__ void plane_sweep(const float *__restrict__ a, float *__restrict__ b,
                                            const Seg (&seg)[TileGeom<K>::SEGS], const F4 (&aux)[TileGeom<K>::SEGS],
                                            bool edge, int n, float alpha, float beta) {
  using G = TileGeom<K>;
#pragma unroll
  for (int j = 0; j < G::SEGS; ++j) {
    const int c = seg[j].c;
    if (c < 0) continue;
    const F4 ce = ld4(a + c), rm = ld4(a + c - G::W), rp = ld4(a + c + G::W);
    const float l = a[c - 1], r = a[c + 4];
    F4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float L = i > 0 ? ce.v[i - 1] : l, R = i < 3 ? ce.v[i + 1] : r;
      o.v[i] = Rule::interior(L, R, rm.v[i], rp.v[i], ce.v[i], aux[j].v[i], alpha, beta);
    }
    if (edge) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = seg[j].x + i, y = seg[j].y;
        if (x <= 0 || x >= n - 1 || y <= 0 || y >= n - 1) {
          const float L = i > 0 ? ce.v[i - 1] : l, R = i < 3 ? ce.v[i + 1] : r;
          o.v[i] = Rule::border(L, R, rm.v[i], rp.v[i], ce.v[i], x, y, n);
        }
      }
    }
    st4(b + c, o);
  }
}

template <int K>
__device__ __forceinline__ void own_segments(Seg (&seg)[TileGeom<K>::SEGS], int x_org, int y_org) {
  using G = TileGeom<K>;
#pragma unroll
  for (int j = 0; j < G::SEGS; ++j) {
    const int i = threadIdx.x + j * THREADS;
    const int ly = i / G::W4, sx = i - ly * G::W4;
    // rows 0 and H-1 have no row above / below in the tile: never computed (they are the first
    // ring the tile edge spoils anyway)
    seg[j].c = (i < G::H * G::W4 && ly >= 1 && ly <= G::H - 2) ? ly * G::W + 4 * sx : -1;
    seg[j].x = x_org + 4 * sx;
    seg[j].y = y_org + ly;
  }
}

// K pressure sweeps.  WITH_DIV: first launch of a projection — the divergence (kernel.cu:265-289)
// is computed from the velocity while the tile is filled (and written out for the later
// launches and fluid_read), p^0 == 0 is a constant (resetPressure, kernel.cu:530-540).
// Every sweep recomputes the whole tile but its edge rows; a cell at distance d from the tile
// edge is right for the first d sweeps, the stored centre is K or more away.  Cells outside the
// domain hold garbage that nothing reads: interior cells have all four neighbours inside the
// domain and the border rules read inward only.
template <int K, bool WITH_DIV>
__global__ __launch_bounds__(THREADS) void k_jacobi_tile(Geom g, int r0, int r1,
                                                         const float *__restrict__ pin,
                                                         const float2 *__restrict__ vel,
                                                         const float *__restrict__ div_in,
                                                         float *__restrict__ div_out,
                                                         float *__restrict__ pout) {
  using G = TileGeom<K>;
  constexpr int W = G::W, H = G::H;
  __shared__ __attribute__((aligned(16))) float A[H * W + 4], B[H * W + 4], Dv[H * W];
  const int n = g.n;
  const int x_org = blockIdx.x * TX - G::HX;        // global x of tile column 0
  const int r_org = r0 + blockIdx.y * TY - K;       // local row of tile row 0
  const int y_org = g.ybase + r_org;
  // Fill.  Rows/columns outside the stored data are read from a clamped address: they only ever
  // feed cells beyond the dependency cone of what this tile stores.
  for (int i = threadIdx.x; i < H * W; i += THREADS) {
    const int ly = i / W, lx = i - ly * W;
    const int x = clampi(x_org + lx, 0, n - 1), r = clampi(r_org + ly, 0, g.L - 1);
    const size_t id = (size_t)r * n + x;
    if (WITH_DIV) {
      const int y = g.ybase + r;
      float d = 0.0f;
      if (x > 0 && x < n - 1 && y > 0 && y < n - 1) {
        const size_t up = (size_t)min(r + 1, g.L - 1) * n + x, dn = (size_t)max(r - 1, 0) * n + x;
        const float2 vL = vel[id - 1], vR = vel[id + 1], vB = vel[dn], vT = vel[up];
        d = 0.5f * ((vR.x - vL.x) + (vT.y - vB.y));  // kernel.cu:282
      }
      Dv[i] = d;
      A[i] = 0.0f;
      // this tile's own cells of the launch range
      if (lx >= G::HX && lx < G::HX + TX && ly >= K && ly < K + TY && r_org + ly < r1 && x_org + lx < n) div_out[id] = d;
    } else {
      A[i] = pin[id];
      Dv[i] = div_in[id];
    }
  }
  Seg seg[G::SEGS];
  own_segments<K>(seg, x_org, y_org);
  __syncthreads();
  F4 dv[G::SEGS];
#pragma unroll
  for (int j = 0; j < G::SEGS; ++j) dv[j] = ld4(Dv + (seg[j].c < 0 ? 0 : seg[j].c));
  const bool edge = x_org <= 0 || x_org + W >= n || y_org <= 0 || y_org + H >= n;  // block-uniform
  float *a = A, *b = B;
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    plane_sweep<K, JacobiRule>(a, b, seg, dv, edge, n, 0.0f, 0.0f);
    __syncthreads();
    float *t = a; a = b; b = t;
  }
  for (int i = threadIdx.x; i < TX * TY; i += THREADS) {
    const int ly = K + i / TX, lx = G::HX + (i & (TX - 1));
    const int x = x_org + lx, r = r_org + ly;
    if (r < r1 && x < n) pout[(size_t)r * n + x] = a[ly * W + lx];
  }
}

// K diffusion sweeps; the two velocity components are two planes in LDS.
template <int K>
__global__ __launch_bounds__(THREADS) void k_diffuse_tile(Geom g, int r0, int r1, float alpha, float beta,
                                                          const float2 *__restrict__ vin,
                                                          float2 *__restrict__ vout) {
  using G = TileGeom<K>;
  constexpr int W = G::W, H = G::H;
  // four planes of H * W floats; dynamic because 10 sweeps need 70 KB (static LDS stops at 64 KB)
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  constexpr int PLANE = H * W + 4;
  float *Au = tile_lds, *Aw = Au + PLANE, *Bu = Aw + PLANE, *Bw = Bu + PLANE;
  const int n = g.n;
  const int x_org = blockIdx.x * TX - G::HX;
  const int r_org = r0 + blockIdx.y * TY - K;
  const int y_org = g.ybase + r_org;
  for (int i = threadIdx.x; i < H * W; i += THREADS) {
    const int ly = i / W, lx = i - ly * W;
    const int x = clampi(x_org + lx, 0, n - 1), r = clampi(r_org + ly, 0, g.L - 1);
    const float2 v = vin[(size_t)r * n + x];
    Au[i] = v.x;
    Aw[i] = v.y;
  }
  Seg seg[G::SEGS];
  own_segments<K>(seg, x_org, y_org);
  F4 none[G::SEGS];
#pragma unroll
  for (int j = 0; j < G::SEGS; ++j) none[j] = F4{{0.0f, 0.0f, 0.0f, 0.0f}};
  __syncthreads();
  const bool edge = x_org <= 0 || x_org + W >= n || y_org <= 0 || y_org + H >= n;
  float *au = Au, *aw = Aw, *bu = Bu, *bw = Bw;
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    plane_sweep<K, DiffuseRule>(au, bu, seg, none, edge, n, alpha, beta);
    plane_sweep<K, DiffuseRule>(aw, bw, seg, none, edge, n, alpha, beta);
    __syncthreads();
    float *t = au; au = bu; bu = t;
    t = aw; aw = bw; bw = t;
  }
  for (int i = threadIdx.x; i < TX * TY; i += THREADS) {
    const int ly = K + i / TX, lx = G::HX + (i & (TX - 1));
    const int x = x_org + lx, r = r_org + ly;
    if (r < r1 && x < n) vout[(size_t)r * n + x] = make_float2(au[ly * W + lx], aw[ly * W + lx]);
  }
}

inline dim3 tile_grid(Geom g, int r0, int r1) { return dim3((g.n + TX - 1) / TX, (r1 - r0 + TY - 1) / TY); }

}  // namespace

// Sweep counts instantiated: the LDS tile grows with K (pressure: three float planes, 52 KB at
// K = 10; diffusion: four, 70 KB at K = 10, dynamic LDS above the 64 KB static limit).
bool tile_has_k(int k, bool) {
  switch (k) {
    case 1: case 2: case 4: case 5: case 8: case 10: return true;
    default: return false;
  }
}

#define TILE_JACOBI_CASE(K_, DIV_)                                                                        \
  case K_:                                                                                               \
    k_jacobi_tile<K_, DIV_><<<tile_grid(g, r0, r1), THREADS, 0, s>>>(g, r0, r1, pin, vel, div_in, div_out, pout); \
    return true;

bool launch_jacobi_tile(hipStream_t s, Geom g, int r0, int r1, int k, const float *pin, const float *div_in,
                        float *pout) {
  if (r1 <= r0) return true;
  const float2 *vel = nullptr;
  float *div_out = nullptr;
  switch (k) {
    TILE_JACOBI_CASE(1, false) TILE_JACOBI_CASE(2, false) TILE_JACOBI_CASE(4, false)
    TILE_JACOBI_CASE(5, false) TILE_JACOBI_CASE(8, false) TILE_JACOBI_CASE(10, false)
    default: return false;
  }
}

bool launch_jacobi_div_tile(hipStream_t s, Geom g, int r0, int r1, int k, const float2 *vel, float *div_out,
                            float *pout) {
  if (r1 <= r0) return true;
  const float *pin = nullptr, *div_in = nullptr;
  switch (k) {
    TILE_JACOBI_CASE(1, true) TILE_JACOBI_CASE(2, true) TILE_JACOBI_CASE(4, true)
    TILE_JACOBI_CASE(5, true) TILE_JACOBI_CASE(8, true) TILE_JACOBI_CASE(10, true)
    default: return false;
  }
}

template <int K>
bool diffuse_tile_launch(hipStream_t s, Geom g, int r0, int r1, float alpha, float beta, const float2 *vin,
                         float2 *vout) {
  using G = TileGeom<K>;
  constexpr size_t bytes = 4 * (size_t)(G::H * G::W + 4) * sizeof(float);
  static bool raised = false;  // more than 64 KB of LDS per workgroup is an opt-in, once per kernel
  if (bytes > 64 * 1024 && !raised) {
    if (hipFuncSetAttribute((const void *)k_diffuse_tile<K>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return false;
    raised = true;
  }
  k_diffuse_tile<K><<<tile_grid(g, r0, r1), THREADS, bytes, s>>>(g, r0, r1, alpha, beta, vin, vout);
  return true;
}

bool launch_diffuse_tile(hipStream_t s, Geom g, int r0, int r1, int k, float alpha, float beta,
                         const float2 *vin, float2 *vout) {
  if (r1 <= r0) return true;
  switch (k) {
    case 1: return diffuse_tile_launch<1>(s, g, r0, r1, alpha, beta, vin, vout);
    case 2: return diffuse_tile_launch<2>(s, g, r0, r1, alpha, beta, vin, vout);
    case 4: return diffuse_tile_launch<4>(s, g, r0, r1, alpha, beta, vin, vout);
    case 5: return diffuse_tile_launch<5>(s, g, r0, r1, alpha, beta, vin, vout);
    case 8: return diffuse_tile_launch<8>(s, g, r0, r1, alpha, beta, vin, vout);
    case 10: return diffuse_tile_launch<10>(s, g, r0, r1, alpha, beta, vin, vout);
    default: return false;
  }
}
