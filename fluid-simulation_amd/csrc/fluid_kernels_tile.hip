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

#ifndef TILE_TX
#define TILE_TX 64
#endif
constexpr int TX = TILE_TX;  // columns a workgroup stores; it stores TY rows (32, or 16 on the smallest grids:
                             // a tile's sweep is a latency chain of about 1.3 us whatever the grid, so when 64 x 32
                             // tiles do not even give every CU one, shorter tiles are what shortens the launch)
constexpr int R = 2;                   // rows of a thread's run

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Tile geometry: K halo rows above and below, HX = K rounded up to a multiple of four halo
// columns left and right, so that the tile and its stored centre are whole 4-float segments
// (16-byte LDS and global accesses).  A thread owns a run of R vertically adjacent segments.
template <int K, int TY> struct TileGeom {
  static constexpr int HX = (K + 3) & ~3;
  static constexpr int W = TX + 2 * HX, H = TY + 2 * K, W4 = W / 4;
  static constexpr int RUNS = (H / R) * W4;
  static constexpr int THREADS = (RUNS + 63) / 64 * 64;
  static constexpr int PLANE = H * W + 8;  // 4 floats of padding before and after
  static_assert(H % R == 0 && THREADS <= 1024, "tile shape");
};

struct F4 { float v[4]; };
__device__ __forceinline__ F4 ld4(const float *p) {  // 16-byte aligned (LDS, or global with n % 4 == 0)
  const float4 t = *reinterpret_cast<const float4 *>(p);
  return F4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void st4(float *p, const F4 &f) {
  *reinterpret_cast<float4 *>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
}

// pressureJacobi, kernel.cu:291-324: interior rule :313, border rules :319-322 (y rules last)
struct JacobiRule {
  static __device__ __forceinline__ float interior(float L, float Rt, float B, float T, float, float d, float, float) {
    return ((((L + Rt) + B) + T) + (-1.0f) * d) * 0.25f;
  }
  static __device__ __forceinline__ float border(float L, float Rt, float B, float T, float, int x, int y, int n, int ny) {
    float o = 0.0f;
    if (x == 0) o = -Rt;
    if (x == n - 1) o = -L;
    if (y == 0) o = -T;
    if (y == ny - 1) o = -B;
    return o;
  }
  // the same rules one side at a time (border tiles: the side is known per run, see Run)
  static __device__ __forceinline__ float x0(float, float Rt, float) { return -Rt; }
  static __device__ __forceinline__ float x1(float L, float, float) { return -L; }
  static __device__ __forceinline__ float y0(float, float T, float) { return -T; }
  static __device__ __forceinline__ float y1(float B, float, float) { return -B; }
};
// diffusion, kernel.cu:173-205: interior rule :195, the border copies through :201-203
struct DiffuseRule {
  static __device__ __forceinline__ float interior(float L, float Rt, float B, float T, float c, float, float alpha, float beta) {
    return ((((L + Rt) + B) + T) + alpha * c) * beta;
  }
  static __device__ __forceinline__ float border(float, float, float, float, float c, int, int, int, int) { return c; }
  static __device__ __forceinline__ float x0(float, float, float c) { return c; }
  static __device__ __forceinline__ float x1(float, float, float c) { return c; }
  static __device__ __forceinline__ float y0(float, float, float c) { return c; }
  static __device__ __forceinline__ float y1(float, float, float c) { return c; }
};

// How a workgroup treats its tile (block-uniform).
enum TileMode {
  TILE_INSIDE = 0,  // tile inside the domain and the stored rows: whole segments, no border rule
  TILE_BORDER = 1,  // touches a domain border or rows that are not stored, n % 4 == 0: whole segments
                    // (a segment is inside the domain or outside it), border rules by side
  TILE_CELLS = 2    // n % 4 != 0: element-wise clamped loads, per-cell border tests, element-wise stores
};

// Where a thread's run sits.
template <int K, int TY> struct Run {
  int c0;      // LDS index of the run's first segment
  int cm, cp;  // LDS index of the segment above the run / below it (clamped into the tile)
  int x, y;    // global coordinates of the run's first cell
  int ly;      // tile row of the run's first row
  bool active;
  // TILE_BORDER: the run's segment holds column 0 / column n-1 (its first / last cell: n % 4 == 0);
  // row j of the run is row 0 / row n-1; the segment lies inside the domain in x
  bool bx0, bx1, by0[R], by1[R], yb, in_x;
  __device__ __forceinline__ void locate(int x_org, int y_org, int n, int ny) {
    using G = TileGeom<K, TY>;
    const int run = threadIdx.x;
    const int rr = run / G::W4, sx = run - rr * G::W4, ly0 = rr * R;
    active = run < G::RUNS;
    c0 = active ? ly0 * G::W + 4 * sx : 0;
    // the tile's first and last row have no row above / below: any value does, they are the
    // ring the tile edge spoils in the first sweep anyway
    cm = ly0 > 0 ? c0 - G::W : c0;
    cp = ly0 + R < G::H ? c0 + R * G::W : c0 + (R - 1) * G::W;
    x = x_org + 4 * sx;
    y = y_org + ly0;
    ly = ly0;
    bx0 = x == 0;
    bx1 = x + 3 == n - 1;
    in_x = x >= 0 && x < n;
    yb = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      by0[j] = y + j == 0;
      by1[j] = y + j == ny - 1;
      yb = yb || by0[j] || by1[j];
    }
  }
  // Does anything that is stored depend on what sweep s (0-based) makes of this run?  The cells a
  // sweep must get right shrink by one ring per sweep towards the stored centre: sweep s produces
  // iterate s + 1, which is needed on the tile rows [s + 1, H - 2 - s] only (the centre plus the
  // K - 1 - s rings the remaining sweeps still reach across).  Runs wholly above or below that are
  // skipped: their cells go stale, and only cells that are skipped themselves ever read them.
  // Runs are numbered row by row, so whole waves fall outside — the first and the last wave of a
  // 64 x 32 tile in half of its ten sweeps.  (The same holds for columns, but a wave spans every
  // column: nothing to skip there.)
  __device__ __forceinline__ bool need(int s) const {
    return ly + R - 1 >= s + 1 && ly <= TileGeom<K, TY>::H - 2 - s;
  }
};

// One sweep of one plane: the run's cells stay in registers (ce) from sweep to sweep; the rows
// above and below the run and the cells left and right of it come from LDS (a), the new values
// go to LDS (b) for the neighbours.
template <int K, int TY, class Rule>
__device__ __forceinline__ void run_sweep(const float *__restrict__ a, float *__restrict__ b, const Run<K, TY> &rn,
                                          F4 (&ce)[R], const F4 (&aux)[R], int mode, int n, int ny, float alpha,
                                          float beta) {
  using G = TileGeom<K, TY>;
  const F4 rm = ld4(a + rn.cm), rp = ld4(a + rn.cp);
  float l[R], r[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    l[j] = a[rn.c0 + j * G::W - 1];
    r[j] = a[rn.c0 + j * G::W + 4];
  }
  F4 o[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const F4 &up = j > 0 ? ce[j - 1] : rm, &dn = j < R - 1 ? ce[j + 1] : rp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float L = i > 0 ? ce[j].v[i - 1] : l[j], Rt = i < 3 ? ce[j].v[i + 1] : r[j];
      float v = Rule::interior(L, Rt, up.v[i], dn.v[i], ce[j].v[i], aux[j].v[i], alpha, beta);
      if (mode == TILE_CELLS) {
        const int x = rn.x + i, y = rn.y + j;
        if (x <= 0 || x >= n - 1 || y <= 0 || y >= ny - 1)
          v = Rule::border(L, Rt, up.v[i], dn.v[i], ce[j].v[i], x, y, n, ny);
      }
      o[j].v[i] = v;
    }
  }
  if (mode == TILE_BORDER) {
    // x rules first, y rules over them (kernel.cu:319-322: four independent ifs, the last wins).
    // Column 0 / n-1 is the first / last cell of one segment per tile row: two selects per row in
    // every wave of a tile at an x border.  Rows 0 / n-1 are one run-row of the tile: a branch only
    // the wave or two that hold it take.
#pragma unroll
    for (int j = 0; j < R; ++j) {
      o[j].v[0] = rn.bx0 ? Rule::x0(l[j], ce[j].v[1], ce[j].v[0]) : o[j].v[0];
      o[j].v[3] = rn.bx1 ? Rule::x1(ce[j].v[2], r[j], ce[j].v[3]) : o[j].v[3];
    }
    if (rn.yb) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const F4 &up = j > 0 ? ce[j - 1] : rm, &dn = j < R - 1 ? ce[j + 1] : rp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = o[j].v[i];
          v = rn.by0[j] ? Rule::y0(up.v[i], dn.v[i], ce[j].v[i]) : v;
          v = rn.by1[j] ? Rule::y1(up.v[i], dn.v[i], ce[j].v[i]) : v;
          o[j].v[i] = v;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    ce[j] = o[j];
    st4(b + rn.c0 + j * G::W, o[j]);
  }
}

// Is this one of the cells the workgroup stores?  (whole segments: HX and TX are multiples of 4)
template <int K, int TY>
__device__ __forceinline__ bool stored_segment(const Run<K, TY> &rn, int j, int x_org, int y_org, int y1, int n) {
  using G = TileGeom<K, TY>;
  const int lx = rn.x - x_org, ly = rn.y + j - y_org;
  return rn.active && lx >= G::HX && lx < G::HX + TX && ly >= K && ly < K + TY && rn.y + j < y1 && rn.x < n;
}

// K pressure sweeps.  WITH_DIV: first launch of a projection — the divergence (kernel.cu:265-289)
// is computed from the velocity while the tile is filled (and written out for the later
// launches and fluid_read), p^0 == 0 is a constant (resetPressure, kernel.cu:530-540).
// Every sweep recomputes the whole tile; a cell at distance d from the tile edge is right for the
// first d sweeps, the stored centre is K or more away.  Cells outside the domain hold garbage
// that nothing reads: interior cells have all four neighbours inside the domain and the border
// rules read inward only.  A tile that touches a domain border or rows that are not stored
// (slab) still moves whole segments and applies the border rules by side (TileMode); only a grid
// that is not a multiple of four wide takes the element-wise path.
// SRC: where the first iterate and the divergence come from
//   0  both from memory (a later launch of a projection)
//   1  WITH_DIV (below): p^0 == 0, divergence computed from the velocity
//   2  p^0 == 0, divergence from memory (first launch when the diffusion launch wrote it, k_diffuse_tile)
template <int K, int SRC, int TY>
__global__ __launch_bounds__((TileGeom<K, TY>::THREADS)) void k_jacobi_tile(Geom g, int r0, int r1,
                                                                      const float *__restrict__ pin,
                                                                      const float2 *__restrict__ vel,
                                                                      const float *__restrict__ div_in,
                                                                      float *__restrict__ div_out,
                                                                      float *__restrict__ pout) {
  using G = TileGeom<K, TY>;
  constexpr bool WITH_DIV = SRC == 1;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::PLANE];
  float *a = lds + 4, *b = lds + G::PLANE + 4;
  const int n = g.n;
  const int x_org = blockIdx.x * TX - G::HX;        // global x of tile column 0
  const int r_org = r0 + blockIdx.y * TY - K;       // local row of tile row 0
  const int y_org = g.ybase + r_org;
  // WITH_DIV reads the velocity one row above and below every tile row: the tile must then lie a
  // row inside the stored rows to take the unclamped path (a slab's tile can start at its local
  // row 0 without touching the domain border)
  constexpr int RM = WITH_DIV ? 1 : 0;
  const bool edge = x_org <= 0 || x_org + G::W >= n || y_org <= 0 || y_org + G::H >= g.ny || r_org < RM ||
                    r_org + G::H > g.L - RM;
  const int mode = (n & 3) ? TILE_CELLS : edge ? TILE_BORDER : TILE_INSIDE;
  Run<K, TY> rn;
  rn.locate(x_org, y_org, n, g.ny);
  F4 ce[R], dv[R];
  const F4 zero = F4{{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int rl = rn.y + j - g.ybase;  // local row
    // TILE_BORDER: a segment is inside the domain in x or outside it; rows are clamped into the
    // stored rows (a row outside them holds values nothing stored depends on)
    const int rc = clampi(rl, 0, g.L - 1);
    const bool whole = mode == TILE_INSIDE || (mode == TILE_BORDER && rn.in_x);
    if (!rn.active || (mode == TILE_BORDER && !rn.in_x)) {
      // the padding of the workgroup, or a segment beyond the x border: no memory accesses
      ce[j] = dv[j] = zero;
    } else if (WITH_DIV && whole) {
      // divergence of the segment from whole-segment loads: rows rl-1, rl, rl+1 of the velocity
      // (u0 w0 u1 w1 | u2 w2 u3 w3) and the cells left and right of it
      const int rb = mode == TILE_INSIDE ? rl - 1 : max(rc - 1, 0), rt = mode == TILE_INSIDE ? rl + 1 : min(rc + 1, g.L - 1);
      const float *q = reinterpret_cast<const float *>(vel + (size_t)rc * n + rn.x);
      const float *qb = reinterpret_cast<const float *>(vel + (size_t)rb * n + rn.x);
      const float *qt = reinterpret_cast<const float *>(vel + (size_t)rt * n + rn.x);
      const F4 c0 = ld4(q), c1 = ld4(q + 4);
      const F4 b0 = ld4(qb), b1 = ld4(qb + 4), t0 = ld4(qt), t1 = ld4(qt + 4);
      const float uL = rn.bx0 ? 0.0f : q[rn.bx0 ? 0 : -2], uR = rn.bx1 ? 0.0f : q[rn.bx1 ? 0 : 8];
      const float u[6] = {uL, c0.v[0], c0.v[2], c1.v[0], c1.v[2], uR};
      const float wB[4] = {b0.v[1], b0.v[3], b1.v[1], b1.v[3]}, wT[4] = {t0.v[1], t0.v[3], t1.v[1], t1.v[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float d = 0.5f * ((u[i + 2] - u[i]) + (wT[i] - wB[i]));  // kernel.cu:282
        // border cells: 0 (kernel.cu:287); the flags are all false in TILE_INSIDE
        if ((i == 0 && rn.bx0) || (i == 3 && rn.bx1) || rn.by0[j] || rn.by1[j]) d = 0.0f;
        dv[j].v[i] = d;
        ce[j].v[i] = 0.0f;
      }
    } else if (WITH_DIV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = clampi(rn.x + i, 0, n - 1), r = rc, y = g.ybase + r;
        float d = 0.0f;
        if (x > 0 && x < n - 1 && y > 0 && y < g.ny - 1) {
          const size_t id = (size_t)r * n + x;
          const size_t up = (size_t)min(r + 1, g.L - 1) * n + x, dn = (size_t)max(r - 1, 0) * n + x;
          const float2 vL = vel[id - 1], vR = vel[id + 1], vB = vel[dn], vT = vel[up];
          d = 0.5f * ((vR.x - vL.x) + (vT.y - vB.y));  // kernel.cu:282
        }
        dv[j].v[i] = d;
        ce[j].v[i] = 0.0f;
      }
    } else if (whole) {
      const size_t id = (size_t)rc * n + rn.x;
      ce[j] = SRC == 2 ? zero : ld4(pin + id);
      dv[j] = ld4(div_in + id);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t id = (size_t)rc * n + clampi(rn.x + i, 0, n - 1);
        ce[j].v[i] = SRC == 2 ? 0.0f : pin[id];
        dv[j].v[i] = div_in[id];
      }
    }
    if (rn.active) st4(a + rn.c0 + j * G::W, ce[j]);
    if (WITH_DIV && stored_segment<K, TY>(rn, j, x_org, y_org, g.ybase + r1, n)) {
      if (mode != TILE_CELLS) {
        st4(div_out + (size_t)rl * n + rn.x, dv[j]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rn.x + i < n) div_out[(size_t)rl * n + rn.x + i] = dv[j].v[i];
      }
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    if (rn.active && rn.need(s)) run_sweep<K, TY, JacobiRule>(a, b, rn, ce, dv, mode, n, g.ny, 0.0f, 0.0f);
    __syncthreads();
    float *t = a; a = b; b = t;
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (!stored_segment<K, TY>(rn, j, x_org, y_org, g.ybase + r1, n)) continue;
    const size_t id = (size_t)(rn.y + j - g.ybase) * n + rn.x;
    if (mode != TILE_CELLS) {
      st4(pout + id, ce[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rn.x + i < n) pout[id + i] = ce[j].v[i];
    }
  }
}

// K diffusion sweeps; the two velocity components are two planes.  DIV: the launch that finishes
// the diffusion also writes the divergence (kernel.cu:265-289) of the field it stores, so that
// the first pressure launch reads 4 bytes per cell of divergence instead of three rows of
// velocity for every cell of ITS tile (halo included: 2.2 times the cells).  The divergence of a
// stored cell needs the final velocity one cell beyond the stored centre, hence a tile with one
// more ring of halo than sweeps (geometry of K + 1, K sweeps).
template <int K, int TY, bool DIV>
__global__ __launch_bounds__((TileGeom<K + (DIV ? 1 : 0), TY>::THREADS)) void k_diffuse_tile(
    Geom g, int r0, int r1, float alpha, float beta, const float2 *__restrict__ vin, float2 *__restrict__ vout,
    float *__restrict__ div_out) {
  constexpr int HALO = K + (DIV ? 1 : 0);
  using G = TileGeom<HALO, TY>;
  // four planes; dynamic because the deep tiles need more than the 64 KB static LDS limit
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  float *au = tile_lds + 4, *aw = au + G::PLANE, *bu = aw + G::PLANE, *bw = bu + G::PLANE;
  const int n = g.n;
  const int x_org = blockIdx.x * TX - G::HX;
  const int r_org = r0 + blockIdx.y * TY - HALO;
  const int y_org = g.ybase + r_org;
  const bool edge = x_org <= 0 || x_org + G::W >= n || y_org <= 0 || y_org + G::H >= g.ny || r_org < 0 ||
                    r_org + G::H > g.L;
  const int mode = (n & 3) ? TILE_CELLS : edge ? TILE_BORDER : TILE_INSIDE;
  Run<HALO, TY> rn;
  rn.locate(x_org, y_org, n, g.ny);
  F4 cu[R], cw[R], none[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int rl = rn.y + j - g.ybase;
    const int rc = clampi(rl, 0, g.L - 1);
    none[j] = F4{{0.0f, 0.0f, 0.0f, 0.0f}};
    if (!rn.active || (mode == TILE_BORDER && !rn.in_x)) {
      cu[j] = cw[j] = none[j];
    } else if (mode != TILE_CELLS) {
      const float *q = reinterpret_cast<const float *>(vin + (size_t)rc * n + rn.x);
      const F4 lo = ld4(q), hi = ld4(q + 4);  // u0 w0 u1 w1 | u2 w2 u3 w3
      cu[j] = F4{{lo.v[0], lo.v[2], hi.v[0], hi.v[2]}};
      cw[j] = F4{{lo.v[1], lo.v[3], hi.v[1], hi.v[3]}};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 v = vin[(size_t)rc * n + clampi(rn.x + i, 0, n - 1)];
        cu[j].v[i] = v.x;
        cw[j].v[i] = v.y;
      }
    }
    if (rn.active) {
      st4(au + rn.c0 + j * G::W, cu[j]);
      st4(aw + rn.c0 + j * G::W, cw[j]);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    if (rn.active && rn.need(s)) {
      run_sweep<HALO, TY, DiffuseRule>(au, bu, rn, cu, none, mode, n, g.ny, alpha, beta);
      run_sweep<HALO, TY, DiffuseRule>(aw, bw, rn, cw, none, mode, n, g.ny, alpha, beta);
    }
    __syncthreads();
    float *t = au; au = bu; bu = t;
    t = aw; aw = bw; bw = t;
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (!stored_segment<HALO, TY>(rn, j, x_org, y_org, g.ybase + r1, n)) continue;
    const size_t id = (size_t)(rn.y + j - g.ybase) * n + rn.x;
    if (mode != TILE_CELLS) {
      float *q = reinterpret_cast<float *>(vout + id);
      st4(q, F4{{cu[j].v[0], cw[j].v[0], cu[j].v[1], cw[j].v[1]}});
      st4(q + 4, F4{{cu[j].v[2], cw[j].v[2], cu[j].v[3], cw[j].v[3]}});
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rn.x + i < n) vout[id + i] = make_float2(cu[j].v[i], cw[j].v[i]);
    }
    if constexpr (DIV) {
      // the final iterate: this run's cells are in cu / cw, everything around them in the planes
      // the last sweep wrote (au / aw after the swap); a stored cell is K + 1 or more from the
      // tile edge, its four neighbours K or more: right after K sweeps
      const F4 wB = j > 0 ? cw[j - 1] : ld4(aw + rn.cm), wT = j < R - 1 ? cw[j + 1] : ld4(aw + rn.cp);
      const float uL = au[rn.c0 + j * G::W - 1], uR = au[rn.c0 + j * G::W + 4];
      F4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float l = i > 0 ? cu[j].v[i - 1] : uL, r = i < 3 ? cu[j].v[i + 1] : uR;
        const int x = rn.x + i, y = rn.y + j;
        float v = 0.5f * ((r - l) + (wT.v[i] - wB.v[i]));                       // kernel.cu:282
        if (x <= 0 || x >= n - 1 || y <= 0 || y >= g.ny - 1) v = 0.0f;         // kernel.cu:287
        d.v[i] = v;
      }
      if (mode != TILE_CELLS) {
        st4(div_out + id, d);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rn.x + i < n) div_out[id + i] = d.v[i];
      }
    }
  }
}

inline dim3 tile_grid(Geom g, int r0, int r1, int ty) { return dim3((g.n + TX - 1) / TX, (r1 - r0 + ty - 1) / ty); }

}  // namespace

// Sweep counts instantiated: the LDS tile grows with K (two planes of (TX + 2 HX) x (TY + 2K)
// floats for the pressure, four for the velocity).  Tile heights: 32 and 16 rows.
bool tile_has_k(int k, bool float2_field) {
  switch (k) {
    case 1: case 2: case 4: case 5: case 8: case 10: return true;
    case 13: case 14: case 20: return !float2_field;  // pressure only: two planes of a 104 x 72 tile are 60 KB
    default: return false;
  }
}
bool tile_has_ty(int ty) { return ty == 16 || ty == 32; }

template <int K, int DIV, int TY>
void jacobi_tile_launch(hipStream_t s, Geom g, int r0, int r1, const float *pin, const float2 *vel, const float *div_in,
                        float *div_out, float *pout) {
  k_jacobi_tile<K, DIV, TY><<<tile_grid(g, r0, r1, TY), TileGeom<K, TY>::THREADS, 0, s>>>(g, r0, r1, pin, vel, div_in,
                                                                                          div_out, pout);
}
template <int K, int DIV>
bool jacobi_tile_ty(hipStream_t s, Geom g, int r0, int r1, int ty, const float *pin, const float2 *vel,
                    const float *div_in, float *div_out, float *pout) {
  if (ty == 16) jacobi_tile_launch<K, DIV, 16>(s, g, r0, r1, pin, vel, div_in, div_out, pout);
  else if (ty == 32) jacobi_tile_launch<K, DIV, 32>(s, g, r0, r1, pin, vel, div_in, div_out, pout);
  else return false;
  return true;
}
template <int DIV>
bool jacobi_tile_k(hipStream_t s, Geom g, int r0, int r1, int k, int ty, const float *pin, const float2 *vel,
                   const float *div_in, float *div_out, float *pout) {
  switch (k) {
    case 1: return jacobi_tile_ty<1, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 2: return jacobi_tile_ty<2, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 4: return jacobi_tile_ty<4, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 5: return jacobi_tile_ty<5, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 8: return jacobi_tile_ty<8, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 10: return jacobi_tile_ty<10, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 13: return jacobi_tile_ty<13, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 14: return jacobi_tile_ty<14, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    case 20: return jacobi_tile_ty<20, DIV>(s, g, r0, r1, ty, pin, vel, div_in, div_out, pout);
    default: return false;
  }
}

bool launch_jacobi_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, const float *pin, const float *div_in,
                        float *pout) {
  if (r1 <= r0) return true;
  if (!pin) return jacobi_tile_k<2>(s, g, r0, r1, k, ty, nullptr, nullptr, div_in, nullptr, pout);  // p^0 == 0
  return jacobi_tile_k<0>(s, g, r0, r1, k, ty, pin, nullptr, div_in, nullptr, pout);
}

bool launch_jacobi_div_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, const float2 *vel, float *div_out,
                            float *pout) {
  if (r1 <= r0) return true;
  return jacobi_tile_k<1>(s, g, r0, r1, k, ty, nullptr, vel, nullptr, div_out, pout);
}

template <int K, int TY, bool DIV>
bool diffuse_tile_launch(hipStream_t s, Geom g, int r0, int r1, float alpha, float beta, const float2 *vin,
                         float2 *vout, float *div_out) {
  using G = TileGeom<K + (DIV ? 1 : 0), TY>;
  constexpr size_t bytes = 4 * (size_t)G::PLANE * sizeof(float);
  // more than 64 KB of LDS per workgroup is an opt-in, once per kernel AND device (a context on
  // another device, fluid_params.device, has its own copy of the code object)
  static bool raised[64] = {false};
  int dev = 0;
  if (bytes > 64 * 1024 && (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !raised[dev])) {
    if (hipFuncSetAttribute((const void *)k_diffuse_tile<K, TY, DIV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return false;
    if (dev >= 0 && dev < 64) raised[dev] = true;
  }
  k_diffuse_tile<K, TY, DIV><<<tile_grid(g, r0, r1, TY), G::THREADS, bytes, s>>>(g, r0, r1, alpha, beta, vin, vout,
                                                                                div_out);
  return true;
}
template <int K>
bool diffuse_tile_ty(hipStream_t s, Geom g, int r0, int r1, int ty, float alpha, float beta, const float2 *vin,
                     float2 *vout, float *div_out) {
  if (div_out) {
    if (ty == 16) return diffuse_tile_launch<K, 16, true>(s, g, r0, r1, alpha, beta, vin, vout, div_out);
    if (ty == 32) return diffuse_tile_launch<K, 32, true>(s, g, r0, r1, alpha, beta, vin, vout, div_out);
    return false;
  }
  if (ty == 16) return diffuse_tile_launch<K, 16, false>(s, g, r0, r1, alpha, beta, vin, vout, nullptr);
  if (ty == 32) return diffuse_tile_launch<K, 32, false>(s, g, r0, r1, alpha, beta, vin, vout, nullptr);
  return false;
}

// div_out != nullptr: also write the divergence of the stored field (whole rows [r0, r1))
bool launch_diffuse_tile(hipStream_t s, Geom g, int r0, int r1, int k, int ty, float alpha, float beta,
                         const float2 *vin, float2 *vout, float *div_out) {
  if (r1 <= r0) return true;
  switch (k) {
    case 1: return diffuse_tile_ty<1>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    case 2: return diffuse_tile_ty<2>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    case 4: return diffuse_tile_ty<4>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    case 5: return diffuse_tile_ty<5>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    case 8: return diffuse_tile_ty<8>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    case 10: return diffuse_tile_ty<10>(s, g, r0, r1, ty, alpha, beta, vin, vout, div_out);
    default: return false;
  }
}
