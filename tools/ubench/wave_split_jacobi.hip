// wave_split_jacobi.hip (round 3 experiment, NOT built into libfluid_hip.so; it was
// csrc/fluid_kernels_ws.hip in the commit before this file moved here, wired in as FLUID_WS=K;
// measurements: profiles/r03/r3_ws*.log) — pressureJacobi sweeps (/root/reference/FluidSimulation/kernel.cu:291-324,
// loop :620-641), temporally blocked with the K levels SPLIT OVER THE WAVES OF A WORKGROUP.
//
// fluid_kernels_tb.hip keeps all K levels x 3 rows of a strip in ONE wave's registers: 108-160
// VGPRs, i.e. 3-4 waves per SIMD (the kernels are bound by instruction issue at that occupancy),
// 2 columns per lane (a strip loses 2K of its 128 columns to halo recompute) and one chunk of rows
// per wave (41-row chunks at 4096^2: 2K = 20 halo rows redone per chunk).  Here a workgroup of NW
// waves owns one strip of 256 columns (4 per lane) and is a pipeline of NW STAGES: wave w holds
// only levels [w LW, (w+1) LW), LW = K / NW, takes level-(w LW) rows from wave w-1 and hands
// level-((w+1) LW) rows to wave w+1 through a double-buffered LDS slot (one 16-byte write and read
// per lane and row), one s_barrier per row step.  The divergence row each level needs travels down
// the stages with the pressure rows: a stage keeps a window of LW divergence rows in registers and
// passes on the row that leaves it.  Wave 0 streams p and div from memory (three rows of register
// prefetch), wave NW-1 stores.  Same per-cell operations in the same order as the straight kernel:
// bit-identical (tests/test_parity_gpu.py, test_wave_split_*).
//
//   step t, wave w:  rin = t - w DL, DL = LW + 1 (+ 1, WS_LATE)   row of level w LW that arrives
//     local level l:  row rin - l - 1 of level w LW + l + 1  from rows rin-l-2 .. rin-l of the level below
//     hands on:       row rin - LW of level (w+1) LW, and divergence row rin - LW - 1
//   wave NW-1 stores row t - (NW - 1) DL - LW of level K.
#include "fluid_internal.h"

#include <mutex>

#pragma clang fp contract(off)

namespace {

constexpr int WAVE = 64;
constexpr int C = 4;   // columns per lane
constexpr int NW = 4;  // waves per workgroup = pipeline stages
// A stage uses the rows it reads from LDS one step AFTER the read is issued (the read of step t + 1's
// input is in flight while step t computes), at the price of one more row of lag per stage.
#ifndef WS_LATE
#define WS_LATE 1
#endif

struct Row4 { float v[C]; };

__device__ __forceinline__ Row4 load_row(const float *p) {
  const float4 t = *reinterpret_cast<const float4 *>(p);
  return Row4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void store_row(float *p, const Row4 &r) {
  *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
__device__ __forceinline__ float neg_opaque(float x) {  // see fluid_kernels_tb.hip
  float r;
  asm("v_xor_b32 %0, 0x80000000, %1" : "=v"(r) : "v"(x));
  return r;
}
struct XBorder { unsigned long long lo, hi; };
__device__ __forceinline__ float select_neg(float o, float v, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, -%2, %3" : "=v"(r) : "v"(o), "v"(v), "s"(mask));
  return r;
}

// kernel.cu:313 for a lane's four columns, hand-placed (the reason is in fluid_kernels_tb.hip,
// jacobi_row2_asm: DPP operands that need no wait states, the four chains interleaved).
//   column 0: left neighbour = the lane below's column 3 (DPP wave_shr), right = own column 1
//   column 3: left = own column 2, right = the lane above's column 0 (DPP wave_shl)
__device__ __forceinline__ void jacobi_row4_asm(Row4 &o, const Row4 &a, const Row4 &b, const Row4 &n, const Row4 &d) {
  asm("v_add_f32_dpp %0, %7, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  // left(b3) + b1
      "v_add_f32_dpp %3, %4, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  // right(b0) + b2
      "v_add_f32 %1, %4, %6\n\t"                                                         // b0 + b2
      "v_add_f32 %2, %5, %7\n\t"                                                         // b1 + b3
      "v_add_f32 %0, %0, %8\n\t"
      "v_add_f32 %1, %1, %9\n\t"
      "v_add_f32 %2, %2, %10\n\t"
      "v_add_f32 %3, %3, %11\n\t"
      "v_add_f32 %0, %0, %12\n\t"
      "v_add_f32 %1, %1, %13\n\t"
      "v_add_f32 %2, %2, %14\n\t"
      "v_add_f32 %3, %3, %15\n\t"
      "v_sub_f32 %0, %0, %16\n\t"
      "v_sub_f32 %1, %1, %17\n\t"
      "v_sub_f32 %2, %2, %18\n\t"
      "v_sub_f32 %3, %3, %19\n\t"
      "v_mul_f32 %0, 0x3e800000, %0\n\t"
      "v_mul_f32 %1, 0x3e800000, %1\n\t"
      "v_mul_f32 %2, 0x3e800000, %2\n\t"
      "v_mul_f32 %3, 0x3e800000, %3"
      : "=&v"(o.v[0]), "=&v"(o.v[1]), "=&v"(o.v[2]), "=&v"(o.v[3])
      : "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]),
        "v"(n.v[0]), "v"(n.v[1]), "v"(n.v[2]), "v"(n.v[3]), "v"(d.v[0]), "v"(d.v[1]), "v"(d.v[2]), "v"(d.v[3]));
}

// Row r of the next level from rows r-1 (a), r (b), r+1 (n) of this one and divergence row r (d).
template <bool XEDGE, bool YCHK>
__device__ __forceinline__ Row4 jacobi_rule(const Row4 &a, const Row4 &b, const Row4 &n, const Row4 &d, int r, int NY,
                                            XBorder xb) {
  Row4 o;
  if (YCHK && (r <= 0 || r >= NY - 1)) {  // wave-uniform: y == 0: -p[x,1]; y == N-1: -p[x,N-2]  (kernel.cu:321-322)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float nv = n.v[c], av = a.v[c];
      o.v[c] = neg_opaque(r <= 0 ? nv : av);
    }
    return o;
  }
  jacobi_row4_asm(o, a, b, n, d);
  if constexpr (XEDGE) {  // n % 4 == 0: column 0 is a lane's first column, column N-1 a lane's last
    o.v[0] = select_neg(o.v[0], b.v[1], xb.lo);  // kernel.cu:319: -p[1,y]
    o.v[3] = select_neg(o.v[3], b.v[2], xb.hi);  // kernel.cu:320: -p[N-2,y]
  }
  return o;
}

template <int K> struct WsGeom {
  static_assert(K % NW == 0, "the levels are dealt evenly to the stages");
  static constexpr int LW = K / NW;            // levels per stage
  static constexpr int DL = LW + 1 + WS_LATE;  // rows a stage lags the one before it
  static constexpr int HX = (K + C - 1) / C * C;
  static constexpr int VW = WAVE * C - 2 * HX;  // columns a strip produces
  static constexpr int U = LW % 2 == 0 ? (LW % 3 == 0 ? LW : 3 * LW) : (LW % 3 == 0 ? 2 * LW : 6 * LW);  // lcm(2, 3, LW)
  static constexpr int LAG = (NW - 1) * DL;     // steps the last stage lags the first
};

// LDS: hand-off h (from wave h to wave h + 1), slot (step parity), part (0 pressure row, 1 divergence row),
// lane: 16 bytes each.
constexpr int kPartBytes = WAVE * 16;
constexpr int kSlotBytes = 2 * kPartBytes;
constexpr int kHandBytes = 2 * kSlotBytes;
constexpr int kLdsBytes = (NW - 1) * kHandBytes;

struct WsArgs {
  Geom g;
  const float *in, *aux;
  float *out, *dump;
};

// ROLE 0: first stage (loads), 1: middle, 2: last (stores)
template <int K, int ROLE, bool XEDGE> struct Stage {
  using G = WsGeom<K>;
  static constexpr int LW = G::LW;
  Row4 s[LW][3];  // s[l]: ring of the rows of local level l; in phase J slots (J, J+1, J+2) % 3 hold rows (r-1, r, r+1)
  Row4 dw[LW];    // divergence window: in phase J slot (J - l) % LW holds row rin - 1 - l
  Row4 pf[3], apf[3];  // ROLE 0: prefetch rings, pf[J % 3] = pressure row t, apf[J % 3] = divergence row t - 1
  Row4 hp, hd;         // ROLE 1, 2 with WS_LATE: the rows read from LDS in the previous step
  WsArgs p;
  int w, xc, ya, yb;
  bool st_ok;
  XBorder xb;
  char *lds_rd, *lds_wr;  // this lane's 16 bytes in hand-off w-1 / w, slot 0, part 0

  __device__ __forceinline__ size_t off(int t) const {
    const int r = min(max(t - p.g.ybase, 0), p.g.L - 1);
    return (size_t)r * p.g.n + xc;
  }

  __device__ __forceinline__ void init(int t0) {
#pragma unroll
    for (int l = 0; l < LW; ++l) {
#pragma unroll
      for (int j = 0; j < 3; ++j) s[l][j] = Row4{{0.f, 0.f, 0.f, 0.f}};
      dw[l] = Row4{{0.f, 0.f, 0.f, 0.f}};
    }
    hp = hd = Row4{{0.f, 0.f, 0.f, 0.f}};
    if constexpr (ROLE == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        pf[i] = load_row(p.in + off(t0 + i));
        apf[i] = load_row(p.aux + off(t0 - 1 + i));
      }
    }
  }

  template <int J, bool YCHK> __device__ __forceinline__ Row4 levels(int rin) {
    Row4 fin;
#pragma unroll
    for (int l = 0; l < LW; ++l) {
      const Row4 d = dw[((J - l) % LW + LW) % LW];
      const Row4 o = jacobi_rule<XEDGE, YCHK>(s[l][J % 3], s[l][(J + 1) % 3], s[l][(J + 2) % 3], d, rin - l - 1, p.g.ny, xb);
      if (l + 1 < LW) s[l + 1][(J + 2) % 3] = o; else fin = o;
    }
    return fin;
  }

  // One row step; J = (t - t0) mod U, so ring slots, window slots and LDS slot parity are static.
  template <int J> __device__ __forceinline__ void step(int t) {
    const int rin = t - w * G::DL;
    Row4 pin, din;
    if constexpr (ROLE == 0) {
      pin = pf[J % 3];
      pf[J % 3] = load_row(p.in + off(t + 3));
      din = apf[J % 3];
      apf[J % 3] = load_row(p.aux + off(t - 1 + 3));
    } else {  // what the stage before wrote in the previous step
      const char *q = lds_rd + ((J + 1) % 2) * kSlotBytes;
      const float4 a = *reinterpret_cast<const float4 *>(q);
      const float4 b = *reinterpret_cast<const float4 *>(q + kPartBytes);
      if constexpr (WS_LATE) {
        pin = hp;
        din = hd;
        hp = Row4{{a.x, a.y, a.z, a.w}};
        hd = Row4{{b.x, b.y, b.z, b.w}};
      } else {
        pin = Row4{{a.x, a.y, a.z, a.w}};
        din = Row4{{b.x, b.y, b.z, b.w}};
      }
    }
    const Row4 old = dw[J % LW];  // divergence row rin - 1 - LW: the next stage's newest
    dw[J % LW] = din;
    s[0][(J + 2) % 3] = pin;
    Row4 fin;
    if (rin - LW >= 1 && rin - 1 <= p.g.ny - 2) fin = levels<J, false>(rin);  // every row of this step is y-interior
    else fin = levels<J, true>(rin);
    if constexpr (ROLE == 2) {
      const int ro = rin - LW;
      const bool real = st_ok && ro >= ya && ro < yb;
      store_row(real ? p.out + off(ro) : p.dump + xc, fin);  // unconditional, see fluid_kernels_tb.hip
    } else {
      char *q = lds_wr + (J % 2) * kSlotBytes;
      *reinterpret_cast<float4 *>(q) = make_float4(fin.v[0], fin.v[1], fin.v[2], fin.v[3]);
      *reinterpret_cast<float4 *>(q + kPartBytes) = make_float4(old.v[0], old.v[1], old.v[2], old.v[3]);
    }
    // LDS writes landed, then the workgroup's barrier.  Raw: __syncthreads()'s fences would also be
    // correct here, this form promises not to touch the vector-memory counter (the prefetch rings).
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  template <int J> __device__ __forceinline__ void steps(int t) {
    if constexpr (J < G::U) {
      step<J>(t + J);
      steps<J + 1>(t);
    }
  }

  __device__ __forceinline__ void run() {
    // level-0 row ya - K is the first one the stored rows depend on; the last stage stores row
    // t - LAG - LW in step t, so row yb - 1 in step yb - 1 + LAG + LW.  Every wave of the workgroup runs the same steps (barriers).
    const int t0 = ya - K, t1 = yb + G::LAG + LW;
    init(t0);
#pragma nounroll
    for (int t = t0; t < t1; t += G::U) steps<0>(t);
  }
};

template <int K, int ROLE, bool XEDGE>
__device__ __forceinline__ void run_stage(const WsArgs &a, int w, int lane, int x0, int xc, int ya, int yb, bool st_ok,
                                          char *lds) {
  Stage<K, ROLE, XEDGE> st;
  st.p = a;
  st.w = w;
  st.xc = xc;
  st.ya = ya;
  st.yb = yb;
  st.st_ok = st_ok;
  st.xb = XBorder{__builtin_amdgcn_ballot_w64(x0 == 0), __builtin_amdgcn_ballot_w64(x0 + C - 1 == a.g.n - 1)};
  st.lds_rd = lds + (w > 0 ? w - 1 : 0) * kHandBytes + lane * 16;
  st.lds_wr = lds + (w < NW - 1 ? w : 0) * kHandBytes + lane * 16;
  st.run();
}

template <int K>
__global__ __launch_bounds__(WAVE *NW) void k_jacobi_ws(Geom g, int r0, int r1, int chunk, const float *__restrict__ pin,
                                                        const float *__restrict__ div, float *__restrict__ pout,
                                                        float *__restrict__ dump) {
  using G = WsGeom<K>;
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  const int lane = threadIdx.x & (WAVE - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and known to be
  const int strips = (g.n + G::VW - 1) / G::VW;
  const int nb = strips * ((r1 - r0 + chunk - 1) / chunk);
  const int id = blockIdx.x;
  if (id >= nb) return;  // whole workgroup
  const int bx = id % strips, by = id / strips;
  const int xs = bx * G::VW;
  const int x0 = xs - G::HX + lane * C;
  const bool in_dom = x0 >= 0 && x0 < g.n;  // n % 4 == 0: a lane's columns are all in or all out
  const bool st_ok = in_dom && lane * C >= G::HX && lane * C < G::HX + G::VW;
  const int xc = min(max(x0, 0), g.n - C);
  const int ya = g.ybase + r0 + by * chunk;
  const int yb = min(ya + chunk, g.ybase + r1);
  const bool xedge = xs == 0 || xs + G::VW + G::HX >= g.n;  // workgroup-uniform
  const WsArgs a{g, pin, div, pout, dump};
  // every wave of the workgroup takes the same XEDGE branch and then one role: the same number of
  // barriers in every wave
  if (xedge) {
    if (w == 0) run_stage<K, 0, true>(a, 0, lane, x0, xc, ya, yb, st_ok, lds);
    else if (w == NW - 1) run_stage<K, 2, true>(a, NW - 1, lane, x0, xc, ya, yb, st_ok, lds);
    else run_stage<K, 1, true>(a, w, lane, x0, xc, ya, yb, st_ok, lds);
  } else {
    if (w == 0) run_stage<K, 0, false>(a, 0, lane, x0, xc, ya, yb, st_ok, lds);
    else if (w == NW - 1) run_stage<K, 2, false>(a, NW - 1, lane, x0, xc, ya, yb, st_ok, lds);
    else run_stage<K, 1, false>(a, w, lane, x0, xc, ya, yb, st_ok, lds);
  }
}

template <int K> int ws_strips(int n) { return (n + WsGeom<K>::VW - 1) / WsGeom<K>::VW; }

template <int K>
void jacobi_ws_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, const float *pin, const float *div, float *pout,
                      void *dump) {
  const int nb = ws_strips<K>(g.n) * ((r1 - r0 + chunk - 1) / chunk);
  k_jacobi_ws<K><<<dim3(nb), dim3(WAVE * NW), 0, s>>>(g, r0, r1, chunk, pin, div, pout, (float *)dump);
}

template <int K> int jacobi_ws_resident() {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_jacobi_ws<K>, WAVE * NW, 0) != hipSuccess ||
      hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return per_cu * cus;
}

}  // namespace

bool ws_has_k(int k) { return k == 8 || k == 12 || k == 16; }
bool ws_supported(int n) { return n % C == 0 && n >= 64; }
int ws_strips_of(int n, int k) {
  switch (k) {
    case 8: return ws_strips<8>(n);
    case 12: return ws_strips<12>(n);
    case 16: return ws_strips<16>(n);
  }
  return 0;
}
// workgroups of the k-sweep kernel resident on the current device at once (0 = unknown)
int ws_resident_blocks(int k) {
  constexpr int kDevs = 64;
  static int cache[kDevs][3];
  static std::mutex mu;
  int dev = 0;
  if (!ws_has_k(k) || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) return 0;
  std::lock_guard<std::mutex> lock(mu);
  int &slot = cache[dev][k / 4 - 2];
  if (slot == 0) {
    const int v = k == 8 ? jacobi_ws_resident<8>() : k == 12 ? jacobi_ws_resident<12>() : jacobi_ws_resident<16>();
    slot = v > 0 ? v : -1;
  }
  return slot > 0 ? slot : 0;
}

bool launch_jacobi_ws(hipStream_t s, Geom g, int r0, int r1, int k, int chunk, const float *pin, const float *div,
                      float *pout, void *dump) {
  if (r1 <= r0) return true;
  if (!ws_supported(g.n) || chunk < 1 || !ws_has_k(k)) return false;
  switch (k) {
    case 8: jacobi_ws_launch<8>(s, g, r0, r1, chunk, pin, div, pout, dump); return true;
    case 12: jacobi_ws_launch<12>(s, g, r0, r1, chunk, pin, div, pout, dump); return true;
    case 16: jacobi_ws_launch<16>(s, g, r0, r1, chunk, pin, div, pout, dump); return true;
  }
  return false;
}
