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

#include <mutex>

#pragma clang fp contract(off)

namespace {

constexpr int WAVE = 64;
// Strips (waves) per workgroup.  6 and 12 re-measured in round 3 with and without the row-loop barrier
// (profiles/r03/r3_tb_wpb.log): 6 breaks the occupancy (4096^2 0.497 -> 0.595 ms), 12 — a whole CU's waves in
// one workgroup — costs 7 % at 16384^2 and 1 % at 4096^2.
#ifndef TB_WPB
#define TB_WPB 4
#endif
constexpr int WAVES_PER_BLOCK = TB_WPB;
// Prefetch ring depth = unroll factor of the time loop (a multiple of 3, the period of the
// level slots).  Measured on MI355X (tools/ubench/valu_rate.hip): one wave issues a VALU op
// only every ~8 cycles, and it takes 4-8 waves per SIMD to approach 2.5-3; these pipelines run
// at 2-4 waves per SIMD, so a wave parked on a load costs issue rate directly.  Six rows of
// prefetch were tried against three: no measurable change (0.967 ms either way for the 13-sweep
// Jacobi launch at 16384^2) at +18 VGPRs and twice the code, so the ring stays at three: the
// kernels are bound by VALU issue at their occupancy, not by load latency.
#ifndef TB_RING
#define TB_RING 3
#endif
constexpr int RING = TB_RING;
#ifndef TB_ASM
#define TB_ASM 1
#endif
#ifndef TB_XCD
#define TB_XCD 0  // see tb_kernel_body
#endif
#ifndef TB_STEADY
#define TB_STEADY 1
#endif
#ifndef TB_RAMP
#define TB_RAMP 1  // unrolled pipeline fill with only the levels each step needs (Pipe::ramp)
#endif
#ifndef TB_RAMP_MINK
#define TB_RAMP_MINK 8
#endif
// k_diffuse_div_tb<10, 2> holds 172 VGPRs: two waves per SIMD.  Held to three (168 VGPRs, 6 spilled) or without
// its steady-state loop (158 VGPRs) it is not faster — 16384^2: 1236 us as it is, 1299 / 1293 us those ways,
// profiles/r03/r3_tb_div_time*.log: the launch takes what its 20 bytes per cell take on this access pattern
// (the plain diffusion launch: 16 bytes, 963 us), whatever the occupancy.
#ifndef TB_DIV_WAVES
#define TB_DIV_WAVES 1  // waves per SIMD the diffusion + divergence launch is held to at least
#endif
#ifndef TB_DIV_STEADY
#define TB_DIV_STEADY 1  // its steady-state loop (20 VGPRs)
#endif
// The pipelines store their output rows through a buffer resource (end of round 3): the lanes and rows that must
// not be stored (x halo of a strip, y halo of a chunk) get an offset beyond the resource and the hardware drops
// them — still an unconditional instruction, so the counted waits stay — instead of writing a dump row (28 % of
// a 16384^2 pressure launch's store requests, 43 % at 4096^2, all of them L2 traffic).  16384^2: k_jacobi_tb
// 754 -> 738 us, k_jacobi0_tb 689 -> 673, k_diffuse_div_tb 1231 -> 1186, the step 6.51 -> 6.45 ms; 4096^2: 67.6 ->
// 65.2, 70.1 -> 69.0, 98.1 -> 93.6.  Non-temporal on top (TB_BUFST_AUX=2): slower again (766 / 679 / 1209 us).
// 0: pointers and the dump row, as before (the divergence rows of k_jacobi_div_tb still go that way).
#ifndef TB_BUFST
#define TB_BUFST 1
#endif
#ifndef TB_ZERO0
#define TB_ZERO0 1  // the first sweep from p^0 == 0 in its short form (JacobiOp::apply_from_zero)
#endif
#ifndef TB_CHAIN_MINK
#define TB_CHAIN_MINK 10  // chained chunks (Pipe::run) are instantiated from this sweep count up
#endif



// value held by lane-1 / lane+1 (DPP wave shift; every lane must be active)
__device__ __forceinline__ float from_left(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// Exact negate (sign-bit flip) the optimiser cannot see through.  ROCm 7.2 hipcc
// miscompiles fneg(select(uniform, rowA, rowB)) on 2-wide float vectors in this
// kernel: one half of the packed negate is dropped (seen in the .s as
// v_cndmask, v_cndmask, v_xor on one register only) and the y-border rows come
// out with the wrong sign in one of the three unrolled phases.  Border paths
// only, so the opaque form costs nothing measurable.
__device__ __forceinline__ float neg_opaque(float x) {
  float r;
  asm("v_xor_b32 %0, 0x80000000, %1" : "=v"(r) : "v"(x));
  return r;
}

// Lane masks of the two border columns of an x-border strip (n even: column 0 is its lane's
// first column, column N-1 its lane's second).  Wave-uniform values, computed once per wave.
struct XBorder { unsigned long long lo, hi; };
__device__ __forceinline__ XBorder x_border(int x0, int N) {
  return XBorder{__builtin_amdgcn_ballot_w64(x0 == 0), __builtin_amdgcn_ballot_w64(x0 + 1 == N - 1)};
}
// lanes in `mask`: -v (an exact sign flip, on the way into the select); others: o.  One instruction.
__device__ __forceinline__ float select_neg(float o, float v, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, -%2, %3" : "=v"(r) : "v"(o), "v"(v), "s"(mask));
  return r;
}
__device__ __forceinline__ float select_val(float o, float v, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(o), "v"(v), "s"(mask));
  return r;
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
// (Non-temporal loads/stores were measured and rejected: the pure access pattern gains 5-10 %
// in tools/ubench/strip_stream.hip, but the real kernels re-read their x/y halos from L2 and
// write the dump row over and over: 0.88 -> 1.05 ms with nt loads, 2.2 ms with nt stores too.)
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

// x halo of a strip: the sweep count rounded to the vector width.  (Rounded up to 16 columns, so that every strip
// starts on a 128-byte line and stores whole lines: slower everywhere — 16384^2 6.49 -> 6.64 ms, 2048^2 0.169 ->
// 0.199 — the extra strips cost more than the split lines; profiles/r03/r3_tb_hxalign.log.)
constexpr int tb_hx(int k, int c) { return (k + c - 1) / c * c; }
template <int K, int C> struct StripGeom {
  static constexpr int HX = tb_hx(K, C);  // x halo, rounded to the vector width
  static constexpr int VW = WAVE * C - 2 * HX;      // columns a strip produces
  static_assert(VW > 0, "strip too narrow for this K");
};

struct NoAux {};

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
#pragma unroll
  for (int c = 0; c < C; c += 2) {
    const float4 t = reinterpret_cast<const float4 *>(p)[c / 2];
    r.u[c] = t.x; r.w[c] = t.y; r.u[c + 1] = t.z; r.w[c + 1] = t.w;
  }
  return r;
}
template <int C>
__device__ __forceinline__ void store_row2(float2 *p, const Row2<C> &r) {
#pragma unroll
  for (int c = 0; c < C; c += 2)
    reinterpret_cast<float4 *>(p)[c / 2] = make_float4(r.u[c], r.w[c], r.u[c + 1], r.w[c + 1]);
}

// Stores through a raw buffer resource (TB_BUFST): an offset at or beyond num_records is dropped by the hardware,
// which is how the lanes and rows that must not be stored are kept out of memory without a branch, an EXEC mask
// or a dump row.  kOobOffset is beyond any resource made here (tb_rsrc: at most 0xFFFFFF00 bytes), and so are the
// up to 32 bytes a lane stores from it (no wrap-around to offset 0).
constexpr unsigned kOobOffset = 0xFFFFFF80u;
#ifndef TB_BUFST_AUX
#define TB_BUFST_AUX 0  // cache policy bits of the stores (2: non-temporal)
#endif
template <int C>
__device__ __forceinline__ void buf_store_row(__amdgpu_buffer_rsrc_t rs, unsigned boff, const Row<C> &r) {
  if constexpr (C == 4) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i d; d.x = __float_as_int(r.v[0]); d.y = __float_as_int(r.v[1]); d.z = __float_as_int(r.v[2]); d.w = __float_as_int(r.v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)boff, 0, TB_BUFST_AUX);
  } else if constexpr (C == 2) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i d; d.x = __float_as_int(r.v[0]); d.y = __float_as_int(r.v[1]);
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)boff, 0, TB_BUFST_AUX);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(r.v[0]), rs, (int)boff, 0, TB_BUFST_AUX);
  }
}
template <int C>
__device__ __forceinline__ void buf_store_row2(__amdgpu_buffer_rsrc_t rs, unsigned boff, const Row2<C> &r) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  static_assert(C == 2 || C == 4, "two or four columns per lane");
  const v4i d0 = {__float_as_int(r.u[0]), __float_as_int(r.w[0]), __float_as_int(r.u[1]), __float_as_int(r.w[1])};
  __builtin_amdgcn_raw_buffer_store_b128(d0, rs, (int)boff, 0, TB_BUFST_AUX);
  if constexpr (C == 4) {
    const v4i d1 = {__float_as_int(r.u[2]), __float_as_int(r.w[2]), __float_as_int(r.u[3]), __float_as_int(r.w[3])};
    __builtin_amdgcn_raw_buffer_store_b128(d1, rs, (int)(boff + 16u), 0, TB_BUFST_AUX);
  }
}
// where a row of the stage after the last level goes (DiffuseDivOp::Tail::push): a pointer, or a buffer offset
template <int C> struct PtrDst {
  float *q;
  __device__ __forceinline__ void put(const Row<C> &d) const { store_row<C>(q, d); }
};
template <int C> struct BufDst {
  __amdgpu_buffer_rsrc_t rs;
  unsigned boff;
  __device__ __forceinline__ void put(const Row<C> &d) const { buf_store_row<C>(rs, boff, d); }
};

// Rows handed from one chunk to the chunk above it (chained chunks, see Pipe::run): 8-byte words,
// word w of a lane at q[w * WAVE].  The two chunks run on any two CUs, possibly of two XCDs whose
// L2s do not snoop each other, so both sides use agent-scope atomics: the store is written through
// and the load is served from the device's coherence point, neither leaves a line behind in an L2
// or L1 that the other side cannot see.  (Vector memory instructions; nothing here goes through
// the scalar cache.)
typedef unsigned long long u64;
#ifndef TB_SIDE_PLAIN
#define TB_SIDE_PLAIN 0  // timing experiments only: 1 = plain stores and loads, 2 = plain stores, 3 = plain loads
#endif
__device__ __forceinline__ void side_put(u64 *q, float a, float b) {
  const u64 v = ((u64)__float_as_uint(b) << 32) | (u64)__float_as_uint(a);
  if (TB_SIDE_PLAIN == 1 || TB_SIDE_PLAIN == 2) *q = v;
  else __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void side_get(const u64 *q, float &a, float &b) {
  const u64 v = (TB_SIDE_PLAIN == 1 || TB_SIDE_PLAIN == 3) ? *q : __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = __uint_as_float((unsigned)v);
  b = __uint_as_float((unsigned)(v >> 32));
}

// ---------------------------------------------------------------------------
// Per-row update rules.  apply<XEDGE, YCHK>(a, b, n, d, r, ...) produces row r
// of the next level from rows r-1 (a), r (b), r+1 (n) of the current level.
// XEDGE: the strip touches x == 0 or x == N-1.  YCHK: r may be <= 0 or >= N-1
// (wave-uniform test); the steady-state instantiation has neither.
// ---------------------------------------------------------------------------

// Row rule in hand-placed ISA (2 columns per lane; every row that is not a y-border row.  A strip at
// an x border runs it too and then overwrites its border column, see JacobiOp::apply).
// Same operations in the same order as the C++ expression in apply(); what the asm buys is
// instruction placement: hipcc folds the lane shifts into v_add_f32_dpp whose tied `old`
// operand lands on a just-written temporary, so nearly every one is preceded by an s_nop (24
// per 13-level row), and a wave issues only one instruction of any kind per ~7 cycles at the
// occupancy these kernels run at.  Here the two columns' chains are interleaved and the DPP
// destination is the (long dead) output slot: no wait states needed.  (DPP hazard: a VALU
// write of the DPP source in the previous two instructions — the source here is a row produced
// a whole step earlier; the tied old/vdst is never consumed with bound_ctrl:1 and full masks.
// tests/test_host_and_build.py scans the emitted ISA for violations.)
__device__ __forceinline__ void jacobi_row2_asm(float &o0, float &o1, float b0, float b1, float a0,
                                                float a1, float n0, float n1, float d0, float d1) {
  asm("v_add_f32_dpp %0, %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  // left(b1) + b1
      "v_add_f32_dpp %1, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  // right(b0) + b0
      "v_add_f32 %0, %0, %4\n\t"
      "v_add_f32 %1, %1, %5\n\t"
      "v_add_f32 %0, %0, %6\n\t"
      "v_add_f32 %1, %1, %7\n\t"
      "v_sub_f32 %0, %0, %8\n\t"
      "v_sub_f32 %1, %1, %9\n\t"
      "v_mul_f32 %0, 0x3e800000, %0\n\t"
      "v_mul_f32 %1, 0x3e800000, %1"
      : "=&v"(o0), "=&v"(o1)
      : "v"(b0), "v"(b1), "v"(a0), "v"(a1), "v"(n0), "v"(n1), "v"(d0), "v"(d1));
}
// One plane (u or w) of the diffusion rule, kernel.cu:195.
__device__ __forceinline__ void diffuse_row2_asm(float &o0, float &o1, float b0, float b1, float a0,
                                                 float a1, float n0, float n1, float alpha, float beta) {
  float t0, t1;
  asm("v_add_f32_dpp %0, %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_mul_f32 %2, %10, %4\n\t"
      "v_mul_f32 %3, %10, %5\n\t"
      "v_add_f32 %0, %0, %6\n\t"
      "v_add_f32 %1, %1, %7\n\t"
      "v_add_f32 %0, %0, %8\n\t"
      "v_add_f32 %1, %1, %9\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32 %1, %1, %3\n\t"
      "v_mul_f32 %0, %11, %0\n\t"
      "v_mul_f32 %1, %11, %1"
      : "=&v"(o0), "=&v"(o1), "=&v"(t0), "=&v"(t1)
      : "v"(b0), "v"(b1), "v"(a0), "v"(a1), "v"(n0), "v"(n1), "s"(alpha), "s"(beta));
}

// pressureJacobi, kernel.cu:291-324
template <int C> struct JacobiOp {
  static constexpr int kCols = C;
  using R = Row<C>;
  using A = Row<C>;  // divergence row
  using Elem = float;
  struct P { const float *in; const float *aux; float *out; float *dump; };
  static __device__ __forceinline__ R zero() { return zero_row<C>(); }
  static __device__ __forceinline__ A zero_aux() { return zero_row<C>(); }
  static __device__ __forceinline__ R load(const P &p, size_t off) { return load_row<C>(p.in + off); }
  static __device__ __forceinline__ A load_aux(const P &p, size_t off) { return load_row<C>(p.aux + off); }
  static __device__ __forceinline__ Elem *out_ptr(const P &p, bool real, size_t off, size_t doff) {
    return real ? p.out + off : p.dump + doff;
  }
  static __device__ __forceinline__ void store_at(Elem *q, const R &r) { store_row<C>(q, r); }
  static constexpr bool kBufStore = TB_BUFST != 0;
  static __device__ __forceinline__ void store_buf(__amdgpu_buffer_rsrc_t rs, unsigned boff, const R &r) {
    buf_store_row<C>(rs, boff, r);
  }
  static constexpr bool kLoadsField = true;   // level 0 is read from memory
  static constexpr bool kStoresAux = false;   // the auxiliary rows are only read
  static constexpr int kTailRows = 0;  // no stage after the last level (DiffuseDivOp has one)
  struct Tail {};
  static constexpr int kSideWords = (C + 1) / 2;  // 8-byte words per lane of a row handed to the chunk above
  static __device__ __forceinline__ void side_store(u64 *q, const R &r) {
#pragma unroll
    for (int w = 0; w < kSideWords; ++w) side_put(q + w * WAVE, r.v[2 * w], r.v[C > 1 ? 2 * w + 1 : 0]);
  }
  static __device__ __forceinline__ R side_load(const u64 *q) {
    R r;
#pragma unroll
    for (int w = 0; w < kSideWords; ++w) side_get(q + w * WAVE, r.v[2 * w], r.v[C > 1 ? 2 * w + 1 : 0]);
    return r;
  }
  // The separate steady-state loop (Pipe::steady_step) costs ~20 VGPRs.  The pressure launches are
  // bound by their memory pattern, not by instruction issue (measured: 0.876 ms with it, 0.866
  // without at 16384^2; 0.086 vs 0.077 at 4096^2 where the registers cost a wave per SIMD).
  static constexpr bool kSteady = false;

  // Divergence rows come from memory through a three-slot ring indexed by the phase; the row a
  // step consumes is also copied into the pipeline's delay line for the deeper levels.
  template <bool XEDGE> struct Feed {
    A apf[RING];  // in phase J: apf[J + i] = row t - 1 + i, i = 0 .. RING-1
    template <class PipeT> __device__ __forceinline__ void init(PipeT &pp, const P &p, int t0) {
#pragma unroll
      for (int i = 0; i < RING; ++i) apf[i] = load_aux(p, pp.off(t0 - 1 + i));
    }
    // aux row t-1 (what level 0 needs in step t); `aoff` addresses row t-1+RING
    template <int J, class PipeT> __device__ __forceinline__ A cur(PipeT &pp, const P &p, size_t aoff) {
      const A r = apf[J % RING];
      apf[J % RING] = load_aux(p, aoff);  // refill the freed slot
      pp.hh(1 - J % 3) = r;
      return r;
    }
    // the same without the refill (the last RING steps that still need a divergence row)
    template <int J, class PipeT> __device__ __forceinline__ A cur_last(PipeT &pp) {
      const A r = apf[J % RING];
      pp.hh(1 - J % 3) = r;
      return r;
    }
    template <int J, bool YCHK, bool LOAD = true, class PipeT>
    __device__ __forceinline__ void produce(PipeT &, const P &, int, size_t, float *) {}
  };

  // The first sweep from p^0 == 0 (JacobiZeroOp, JacobiDivOp): ((((0 + 0) + 0) + 0) - d) * 0.25 is (0 - d) * 0.25
  // bit for bit (the three sums of +0 are +0, and +0 - d is what the general rule's last subtraction computes),
  // so that level costs 2 instructions per column instead of 5 and needs no rows of zeros: 6 of the ~153 VALU
  // instructions of a 14-sweep step.  The border rules give -(+0) = -0, as the general rule does on rows of +0.
  static constexpr bool kZeroShortcut = TB_ASM && C == 2 && TB_ZERO0;
  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply_from_zero(const A &d, int r, int NY, XBorder xb, float zero) {
    R o;
    if (YCHK && (r <= 0 || r >= NY - 1)) {  // wave-uniform: -p^0 of the row inward
#pragma unroll
      for (int c = 0; c < C; ++c) o.v[c] = neg_opaque(zero);
      return o;
    }
    asm("v_sub_f32 %0, %2, %3\n\t"
        "v_sub_f32 %1, %2, %4\n\t"
        "v_mul_f32 %0, 0x3e800000, %0\n\t"
        "v_mul_f32 %1, 0x3e800000, %1"
        : "=&v"(o.v[0]), "=&v"(o.v[C > 1 ? 1 : 0])
        : "v"(zero), "v"(d.v[0]), "v"(d.v[C > 1 ? 1 : 0]));
    if constexpr (XEDGE) {  // kernel.cu:319-320 on a field of +0
      o.v[0] = select_neg(o.v[0], zero, xb.lo);
      o.v[C > 1 ? 1 : 0] = select_neg(o.v[C > 1 ? 1 : 0], zero, xb.hi);
    }
    return o;
  }

  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply(const R &a, const R &b, const R &n, const A &d, int r,
                                            int x0, int N, int NY, const P &, XBorder xb) {
    R o;
    if (YCHK && (r <= 0 || r >= NY - 1)) {  // wave-uniform
#pragma unroll
      for (int c = 0; c < C; ++c) {  // y == 0: -p[x,1]; y == N-1: -p[x,N-2]  (kernel.cu:321-322)
        const float nv = n.v[c], av = a.v[c];  // values, not a select of addresses (keeps SROA)
        o.v[c] = neg_opaque(r <= 0 ? nv : av);
      }
      return o;
    }
    if constexpr (TB_ASM && C == 2) {
      jacobi_row2_asm(o.v[0], o.v[1], b.v[0], b.v[1], a.v[0], a.v[1], n.v[0], n.v[1], d.v[0], d.v[1]);
      if constexpr (XEDGE) {
        // A strip at the x border runs the same rule and overwrites its one border column per
        // side (n even: column 0 is a lane's first column, column N-1 a lane's second).  What
        // the lanes beyond the border compute never gets past that column.  The single-round
        // launches (4096^2, slabs) end when their slowest wave does, and these strips were it.
        o.v[0] = select_neg(o.v[0], b.v[1], xb.lo);  // kernel.cu:319: -p[1,y]
        o.v[1] = select_neg(o.v[1], b.v[0], xb.hi);  // kernel.cu:320: -p[N-2,y]
      }
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
        if (x == 0) s = neg_opaque(vR);      // kernel.cu:319
        if (x == N - 1) s = neg_opaque(vL);  // kernel.cu:320
      }
      o.v[c] = s;
    }
    return o;
  }
};

// diffusion, kernel.cu:173-205 (float2 field: two planes per column)
template <int C> struct DiffuseOp {
  static constexpr int kCols = C;
  using R = Row2<C>;
  using A = NoAux;
  using Elem = float2;
  struct P { const float2 *in; float2 *out; float2 *dump; float alpha, beta; };
  static __device__ __forceinline__ R zero() { return zero_row2<C>(); }
  static __device__ __forceinline__ A zero_aux() { return NoAux{}; }
  static __device__ __forceinline__ R load(const P &p, size_t off) { return load_row2<C>(p.in + off); }
  static __device__ __forceinline__ A load_aux(const P &, size_t) { return NoAux{}; }
  static __device__ __forceinline__ Elem *out_ptr(const P &p, bool real, size_t off, size_t doff) {
    return real ? p.out + off : p.dump + doff;
  }
  static __device__ __forceinline__ void store_at(Elem *q, const R &r) { store_row2<C>(q, r); }
  // (two columns per lane only: the compiler assembles the second 16 bytes of a four-column row through scratch
  // memory — scratch_store 24 bytes, scratch_load 16 at offset 8 — whichever way the vector is written down; the
  // four-column plans are not a default anywhere and keep the pointer stores)
  static constexpr bool kZeroShortcut = false;
  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply_from_zero(const A &, int, int, XBorder, float) { return zero(); }
  static constexpr bool kBufStore = TB_BUFST != 0 && C == 2;
  static __device__ __forceinline__ void store_buf(__amdgpu_buffer_rsrc_t rs, unsigned boff, const R &r) {
    buf_store_row2<C>(rs, boff, r);
  }
  static constexpr bool kLoadsField = true;
  static constexpr bool kStoresAux = false;
  static constexpr bool kSteady = true;  // two planes per column: issue-bound, 1.14 -> 1.06 ms at 16384^2
  static constexpr int kTailRows = 0;
  struct Tail {};
  static constexpr int kSideWords = C;  // (u, w) of one column per 8-byte word
  static __device__ __forceinline__ void side_store(u64 *q, const R &r) {
#pragma unroll
    for (int w = 0; w < C; ++w) side_put(q + w * WAVE, r.u[w], r.w[w]);
  }
  static __device__ __forceinline__ R side_load(const u64 *q) {
    R r;
#pragma unroll
    for (int w = 0; w < C; ++w) side_get(q + w * WAVE, r.u[w], r.w[w]);
    return r;
  }
  template <bool XEDGE> struct Feed {  // no auxiliary field
    template <class PipeT> __device__ __forceinline__ void init(PipeT &, const P &, int) {}
    template <int J, class PipeT> __device__ __forceinline__ A cur(PipeT &, const P &, size_t) { return NoAux{}; }
    template <int J, class PipeT> __device__ __forceinline__ A cur_last(PipeT &) { return NoAux{}; }
    template <int J, bool YCHK, bool LOAD = true, class PipeT>
    __device__ __forceinline__ void produce(PipeT &, const P &, int, size_t, float2 *) {}
  };

  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply(const R &a, const R &b, const R &n, const A &, int r,
                                            int x0, int N, int NY, const P &p, XBorder xb) {
    if (YCHK && (r <= 0 || r >= NY - 1)) return b;  // border copies through, kernel.cu:201-203
    R o;
    const float alpha = p.alpha, beta = p.beta;
    if constexpr (TB_ASM && C == 2) {
      diffuse_row2_asm(o.u[0], o.u[1], b.u[0], b.u[1], a.u[0], a.u[1], n.u[0], n.u[1], alpha, beta);
      diffuse_row2_asm(o.w[0], o.w[1], b.w[0], b.w[1], a.w[0], a.w[1], n.w[0], n.w[1], alpha, beta);
      if constexpr (XEDGE) {  // border columns copy through, kernel.cu:201-203 (see JacobiOp::apply)
        o.u[0] = select_val(o.u[0], b.u[0], xb.lo); o.w[0] = select_val(o.w[0], b.w[0], xb.lo);
        o.u[1] = select_val(o.u[1], b.u[1], xb.hi); o.w[1] = select_val(o.w[1], b.w[1], xb.hi);
      }
      return o;
    }
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
};

// The last diffusion launch of a whole step, which also writes the divergence of the field it stores
// (kernel.cu:265-289) — one stage more after the last level: divergence row r - 1 from final rows r - 2, r - 1, r
// as they leave the pipeline.  The first pressure launch then reads 4 bytes of divergence per cell instead of 8
// of velocity and writes no divergence (JacobiZeroOp): 4 bytes per cell and step less, and the velocity ring of
// the divergence + pressure launch goes.  The chunk computes one final row more at either end, the strip one
// column more at either side (kTailRows).
template <int C> struct DiffuseDivOp : DiffuseOp<C> {
  using R = Row2<C>;
  using A = NoAux;
  using Elem = float2;
  struct P { const float2 *in; float2 *out; float2 *dump; float alpha, beta; float *div_out; };
  static __device__ __forceinline__ R load(const P &p, size_t off) { return load_row2<C>(p.in + off); }
  static __device__ __forceinline__ A load_aux(const P &, size_t) { return NoAux{}; }
  static __device__ __forceinline__ Elem *out_ptr(const P &p, bool real, size_t off, size_t doff) {
    return real ? p.out + off : p.dump + doff;
  }
  static constexpr int kTailRows = 1;
  static constexpr bool kSteady = TB_DIV_STEADY;
  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply(const R &a, const R &b, const R &n, const A &d, int r,
                                            int x0, int N, int NY, const P &p, XBorder xb) {
    return DiffuseOp<C>::template apply<XEDGE, YCHK>(a, b, n, d, r, x0, N, NY,
                                                     typename DiffuseOp<C>::P{nullptr, nullptr, nullptr, p.alpha, p.beta}, xb);
  }
  template <bool XEDGE> struct Feed {
    template <class PipeT> __device__ __forceinline__ void init(PipeT &, const P &, int) {}
    template <int J, class PipeT> __device__ __forceinline__ A cur(PipeT &, const P &, size_t) { return NoAux{}; }
    template <int J, class PipeT> __device__ __forceinline__ A cur_last(PipeT &) { return NoAux{}; }
    template <int J, bool YCHK, bool LOAD = true, class PipeT>
    __device__ __forceinline__ void produce(PipeT &, const P &, int, size_t, float2 *) {}
  };
  struct Tail {
    R f1;         // final row r - 1
    float w2[C];  // y-velocity of final row r - 2
    __device__ __forceinline__ void init() {
      f1 = zero_row2<C>();
#pragma unroll
      for (int c = 0; c < C; ++c) w2[c] = 0.0f;
    }
    // fin = final row r; stores the divergence of row r - 1 at dst (its real place or the dump row)
    template <bool XEDGE, bool YCHK, class Dst>
    __device__ __forceinline__ void push(const R &fin, int r, int x0, int N, int NY, Dst dst) {
      Row<C> d;
      if (YCHK && (r - 1 <= 0 || r - 1 >= NY - 1)) {  // wave-uniform: border rows, kernel.cu:287
        d = zero_row<C>();
      } else {
        const float lu = from_left(f1.u[C - 1]);
        const float ru = from_right(f1.u[0]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float uL = c > 0 ? f1.u[c - 1] : lu;
          const float uR = c < C - 1 ? f1.u[c + 1] : ru;
          float v = 0.5f * ((uR - uL) + (fin.w[c] - w2[c]));  // kernel.cu:282
          if (XEDGE) {
            const int x = x0 + c;
            if (x == 0 || x == N - 1) v = 0.0f;  // kernel.cu:287
          }
          d.v[c] = v;
        }
      }
      dst.put(d);
#pragma unroll
      for (int c = 0; c < C; ++c) w2[c] = f1.w[c];
      f1 = fin;
    }
  };
};

// pressureJacobi from p^0 == 0 with the divergence in memory (written by DiffuseDivOp's launch): JacobiOp whose
// level 0 is the constant zero (as in JacobiDivOp: values the optimiser cannot fold, for the sign of zero).
template <int C> struct JacobiZeroOp : JacobiOp<C> {
  using R = Row<C>;
  struct P { const float *in; const float *aux; float *out; float *dump; };
  static __device__ __forceinline__ R load(const P &, size_t) {
    R r;
#pragma unroll
    for (int c = 0; c < C; ++c) asm("v_mov_b32 %0, 0" : "=v"(r.v[c]));
    return r;
  }
  static __device__ __forceinline__ Row<C> load_aux(const P &p, size_t off) { return load_row<C>(p.aux + off); }
  static __device__ __forceinline__ float *out_ptr(const P &p, bool real, size_t off, size_t doff) {
    return real ? p.out + off : p.dump + doff;
  }
  static constexpr bool kLoadsField = false;
  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply(const R &a, const R &b, const R &n, const Row<C> &d, int r,
                                            int x0, int N, int NY, const P &, XBorder xb) {
    return JacobiOp<C>::template apply<XEDGE, YCHK>(a, b, n, d, r, x0, N, NY, typename JacobiOp<C>::P{}, xb);
  }
  template <bool XEDGE> struct Feed {  // JacobiOp's, on this P
    Row<C> apf[RING];
    template <class PipeT> __device__ __forceinline__ void init(PipeT &pp, const P &p, int t0) {
#pragma unroll
      for (int i = 0; i < RING; ++i) apf[i] = load_aux(p, pp.off(t0 - 1 + i));
    }
    template <int J, class PipeT> __device__ __forceinline__ Row<C> cur(PipeT &pp, const P &p, size_t aoff) {
      const Row<C> r = apf[J % RING];
      apf[J % RING] = load_aux(p, aoff);
      pp.hh(1 - J % 3) = r;
      return r;
    }
    template <int J, class PipeT> __device__ __forceinline__ Row<C> cur_last(PipeT &pp) {
      const Row<C> r = apf[J % RING];
      pp.hh(1 - J % 3) = r;
      return r;
    }
    template <int J, bool YCHK, bool LOAD = true, class PipeT>
    __device__ __forceinline__ void produce(PipeT &, const P &, int, size_t, float *) {}
  };
};

// First projection launch: divergence (kernel.cu:265-289) + resetPressure (kernel.cu:530-540,
// p^0 == 0 is never read from memory) + the first K pressureJacobi sweeps, in one pass over the
// velocity.  Same sweep rule as JacobiOp; only where level 0 and the divergence come from differs:
// level 0 is the constant 0 and divergence row t is computed from velocity rows t-1, t, t+1 as
// they stream by (and written out for the later launches and for fluid_read).
template <int C> struct JacobiDivOp : JacobiOp<C> {
  using R = Row<C>;
  using A = Row<C>;
  using Elem = float;
  struct P { const float2 *vel; float *div_out; float *out; float *dump; int n; };
  // p^0 == 0, as a value the optimiser cannot see through.  With a literal zero row the first sweep
  // of the columns whose neighbours are all in this lane (four columns per lane: the middle two)
  // folds to (0 - d) * 0.25, and the AMDGPU instruction selector turns fsub(+0, d) feeding a
  // multiply into a negate modifier: -0 where the reference's sum gives +0 (d == +0 away from any
  // splat).  Found by tools/fuzz_parity.py on the non-default four-column plans; bit-exactness
  // includes the sign of zero.
  static __device__ __forceinline__ R load(const P &, size_t) {
    R r;
#pragma unroll
    for (int c = 0; c < C; ++c) asm("v_mov_b32 %0, 0" : "=v"(r.v[c]));
    return r;
  }
  static __device__ __forceinline__ Elem *out_ptr(const P &p, bool real, size_t off, size_t doff) {
    return real ? p.out + off : p.dump + doff;
  }
  static constexpr bool kLoadsField = false;
  static constexpr bool kStoresAux = true;  // divergence row t is stored in step t
  static constexpr bool kSteady = false;

  template <bool XEDGE, bool YCHK>
  static __device__ __forceinline__ R apply(const R &a, const R &b, const R &n, const A &d, int r,
                                            int x0, int N, int NY, const P &, XBorder xb) {
    return JacobiOp<C>::template apply<XEDGE, YCHK>(a, b, n, d, r, x0, N, NY, typename JacobiOp<C>::P{}, xb);
  }

  template <bool XEDGE> struct Feed {
    Row2<C> vpf[RING];  // velocity ring: in phase J slot J + i holds row t + 1 + i
    float wa[C];        // y-velocity of row t-1
    Row2<C> vb;         // velocity row t
    template <class PipeT> __device__ __forceinline__ void init(PipeT &pp, const P &p, int t0) {
#pragma unroll
      for (int c = 0; c < C; ++c) wa[c] = 0.0f;
      vb = load_row2<C>(p.vel + pp.off(t0));
#pragma unroll
      for (int i = 0; i < RING; ++i) vpf[i] = load_row2<C>(p.vel + pp.off(t0 + 1 + i));
    }
    // divergence row t-1 was produced by the previous step straight into the slot read here
    template <int J, class PipeT> __device__ __forceinline__ A cur(PipeT &pp, const P &, size_t) {
      return pp.hh(1 - J % 3);
    }
    template <int J, class PipeT> __device__ __forceinline__ A cur_last(PipeT &pp) { return pp.hh(1 - J % 3); }
    // divergence row t from velocity rows t-1, t, t+1; `voff` addresses velocity row t+1+RING,
    // `dst` is where divergence row t is stored (its real place or the dump row)
    // (LOAD false: no refill — the tail of a chained chunk, whose last divergence rows are in flight)
    template <int J, bool YCHK, bool LOAD = true, class PipeT>
    __device__ __forceinline__ void produce(PipeT &pp, const P &p, int t, size_t voff, float *dst) {
      const Row2<C> vc = vpf[J % RING];                 // velocity row t+1
      if constexpr (LOAD) vpf[J % RING] = load_row2<C>(p.vel + voff);  // refill the freed slot
      A d;
      const int N = pp.g.n;
      if (YCHK && (t <= 0 || t >= pp.g.ny - 1)) {  // wave-uniform: border rows, kernel.cu:287
        d = zero_row<C>();
      } else {
        const float lu = from_left(vb.u[C - 1]);
        const float ru = from_right(vb.u[0]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float uL = c > 0 ? vb.u[c - 1] : lu;
          const float uR = c < C - 1 ? vb.u[c + 1] : ru;
          float v = 0.5f * ((uR - uL) + (vc.w[c] - wa[c]));  // kernel.cu:282
          if (XEDGE) {
            const int x = pp.x0 + c;
            if (x == 0 || x == N - 1) v = 0.0f;  // kernel.cu:287
          }
          d.v[c] = v;
        }
      }
      store_row<C>(dst, d);
#pragma unroll
      for (int c = 0; c < C; ++c) wa[c] = vb.w[c];
      vb = vc;
      pp.hh(J % 3 == 2 ? -2 : -(J % 3)) = d;  // where the next step's cur() reads it (after the shift if J == 2)
    }
  };
};

// ---------------------------------------------------------------------------
// The register pipeline.  Level l keeps three row slots s[l][0..2] used as a
// ring: in a step of phase J the slots (J, J+1, J+2) mod 3 hold rows
// (r-1, r, r+1); the producer (the load for level 0, level l-1 otherwise)
// writes the new row straight into slot (J+2) mod 3, so rows never move
// between registers.  The time loop is unrolled by 3 so every slot index is a
// compile-time constant.
//
// The auxiliary field (divergence) is needed by level l at row t-l-1: a delay
// line.  h[] holds it with hh(j) = row tg - j (tg = first step of the current
// group of three), so level l >= 1 reads hh(l+1-J) in phase J — static indices
// again — and the line shifts by three rows once per group instead of by one
// row per step (a third of the register moves).
// ---------------------------------------------------------------------------
template <class Op, int K, bool XEDGE, bool CHAIN = false> struct Pipe {
  using R = typename Op::R;
  using A = typename Op::A;
  using P = typename Op::P;
  using Elem = typename Op::Elem;
  R s[K + 1][3];
  A h[K + 3];   // hh(j), j = -2 .. K
  R pf[RING];   // prefetch ring: in phase J, pf[J + i] = level-0 row t + i
  float zero0;  // +0 the optimiser cannot see through (kZeroShortcut: all there is of a level 0 that is p^0 == 0)
  typename Op::template Feed<XEDGE> feed;  // where the auxiliary rows come from
  Geom g;
  int x0, xc, ya, yb;  // xc: x0 clamped into the domain (address of this lane's loads)
  int st_a, st_b;      // rows this chunk stores: [ya, yb) but for kTailRows rows at either end
  static constexpr int T = Op::kTailRows;
  static_assert(!(CHAIN && T > 0), "no chained form of the launch with a stage after the last level");
  typename Op::Tail tail;
  bool st_ok;
  // Op::kBufStore: rows [st_a, st_b) of the output field (and of the field the stage after the last level writes)
  // as buffer resources; a stored row's offset is relative to row st_a
  __amdgpu_buffer_rsrc_t obuf, dbuf;
  XBorder xb;  // meaningful in x-border strips only
  // chained chunks (CHAIN): this wave's record of side rows and the record of the chunk below it
  // (same strip), as per-lane pointers; the flags that say a record is complete; which ends of
  // this chunk are chained (wave-uniform)
  u64 *side_mine;
  const u64 *side_below;
  int *flag_mine, *flag_below, *status;
  u64 *trace;  // null, or this wave's four time stamps (FLUID_TB_TRACE)
  bool sync;   // wave-uniform: barriers in the row loops (TbChain::sync)
  bool hands_up, takes_from_below;
  static constexpr int SIDE_ROW = Op::kSideWords * WAVE;   // 8-byte words of one side row of a strip
  static constexpr int SIDE_ROWS = 2 * K - 2;              // levels 1 .. K-1, two rows each
  R sb[CHAIN ? SIDE_ROWS : 1];                             // side rows on their way in (down_step)

  __device__ __forceinline__ A &hh(int j) { return h[j + 2]; }

  // Every load and store below is UNCONDITIONAL.  A predicated or branched-around memory
  // operation makes hipcc's s_waitcnt placement assume it may not have been issued, and the
  // wait before a prefetched row degrades to vmcnt(0/1), i.e. no prefetch at all.  So rows and
  // columns outside the stored data are read from a clamped address (the values only ever feed
  // cells that are not stored: a strip/chunk edge loses one ring of cells per sweep, and at a
  // domain border the boundary rule never reads outward), and rows that must not be stored go
  // to a dump row instead of being skipped.
  __device__ __forceinline__ size_t off(int t) const {
#ifdef TB_FAKE_NOHALO  // timing experiment only (wrong results): the y-halo rows of a chunk cost no new traffic
    t = min(max(t, ya), yb - 1);
#endif
    const int r = min(max(t - g.ybase, 0), g.L - 1);
    return (size_t)r * g.n + xc;
  }

  // store row `ro` of the output field if `real`; else nowhere (buffer stores) or into the dump row
  __device__ __forceinline__ void put(const P &p, bool real, int ro, const R &fin) {
    if constexpr (Op::kBufStore) {
      Op::store_buf(obuf, real ? boff(ro, sizeof(Elem)) : kOobOffset, fin);
    } else {
      Op::store_at(Op::out_ptr(p, real, off(ro), (size_t)xc), fin);
    }
  }
  // byte offset of this lane's columns of row r (st_a <= r < st_b) from the start of row st_a
  __device__ __forceinline__ unsigned boff(int r, size_t elem) const {
    return ((unsigned)(r - st_a) * (unsigned)g.n + (unsigned)xc) * (unsigned)elem;
  }

  __device__ __forceinline__ void init(const P &p, int t0) {
#pragma unroll
    for (int l = 0; l < K; ++l) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // a level 0 that is not read from memory (p^0 == 0) stays in its slots for the whole run:
        // Op::load gives it as values the optimiser cannot fold (JacobiDivOp::load)
        if (l == 0 && !Op::kLoadsField) s[l][j] = Op::load(p, 0);
        else s[l][j] = Op::zero();
      }
    }
#pragma unroll
    for (int j = 0; j < K + 3; ++j) h[j] = Op::zero_aux();
    if constexpr (Op::kLoadsField) {
#pragma unroll
      for (int i = 0; i < RING; ++i) pf[i] = Op::load(p, off(t0 + i));
    }
    feed.init(*this, p, t0);
    if constexpr (T > 0) tail.init();
    asm("v_mov_b32 %0, 0" : "=v"(zero0));
  }
  // the stage after the last level: fin is final row ro; its output is row ro - 1
  template <bool YCHK>
  __device__ __forceinline__ void tail_push(const P &p, const R &fin, int ro) {
    const int rd = ro - 1;
    const bool real = st_ok && rd >= st_a && rd < st_b;
    if constexpr (Op::kBufStore) {
      tail.template push<XEDGE, YCHK>(fin, ro, x0, g.n, g.ny, BufDst<Op::kCols>{dbuf, real ? boff(rd, sizeof(float)) : kOobOffset});
    } else {
      tail.template push<XEDGE, YCHK>(fin, ro, x0, g.n, g.ny,
                                      PtrDst<Op::kCols>{real ? p.div_out + off(rd) : (float *)p.dump + 2 * (size_t)g.n + xc});
    }
  }

  __device__ __forceinline__ void shift_aux() {
#pragma unroll
    for (int j = K; j >= 1; --j) hh(j) = hh(j - 3);  // (as v_mov_b64 pairs, half the instructions: no faster, 0.648 vs 0.643 ms)
  }

  // LV: levels evaluated (K in the general step; fewer while the pipeline fills, see ramp())
  template <int J, bool YCHK, int LV = K, int LO = 0>
  __device__ __forceinline__ R levels(const P &p, int t, const A &cur) {
    R fin = Op::zero();
#pragma unroll
    for (int l = LO; l < LV; ++l) {
      A d = cur;  // a copy of the value (never a select between two rows' addresses)
      if (l > 0) d = hh(l + 1 - J % 3);
      R o;
      if constexpr (!Op::kLoadsField && Op::kZeroShortcut) {
        if (l == 0) o = Op::template apply_from_zero<XEDGE, YCHK>(d, t - 1, g.ny, xb, zero0);
        else o = Op::template apply<XEDGE, YCHK>(s[l][J % 3], s[l][(J + 1) % 3], s[l][(J + 2) % 3], d,
                                                 t - l - 1, x0, g.n, g.ny, p, xb);
      } else {
        o = Op::template apply<XEDGE, YCHK>(s[l][J % 3], s[l][(J + 1) % 3], s[l][(J + 2) % 3], d,
                                            t - l - 1, x0, g.n, g.ny, p, xb);
      }
      if (l + 1 < K) s[l + 1][(J + 2) % 3] = o; else fin = o;
    }
    return fin;
  }

  // One time step, any row.  Nothing loop-carried changes register: the rings are indexed by the
  // phase, so the loads issued here are first touched RING steps later (counted vmcnt).
  template <int J>
  __device__ __forceinline__ void step(const P &p, int t) {
    if (J % 3 == 0) shift_aux();
    const A cur = feed.template cur<J>(*this, p, off(t - 1 + RING));  // aux row t-1
    if constexpr (Op::kLoadsField) {
      s[0][(J + 2) % 3] = pf[J % RING];          // level 0, row t
      pf[J % RING] = Op::load(p, off(t + RING));  // refill the freed slot: RING rows of prefetch
    }
    if constexpr (Op::kStoresAux) {
      const bool areal = st_ok && t >= ya && t < yb;
      feed.template produce<J, true>(*this, p, t, off(t + 1 + RING),
                                     areal ? p.div_out + off(t) : p.dump + (size_t)g.n + xc);
    }
    R fin;
    if (t - K >= 1 && t - 1 <= g.ny - 2) fin = levels<J, false>(p, t, cur);  // every level's row is y-interior
    else fin = levels<J, true>(p, t, cur);
    const int ro = t - K;  // level K row just produced
    put(p, st_ok && ro >= st_a && ro < st_b, ro, fin);
    if constexpr (T > 0) {
      if (ro - 1 >= 1 && ro - 1 <= g.ny - 2) tail_push<false>(p, fin, ro);
      else tail_push<true>(p, fin, ro);
    }
  }

  // The first 2K steps of a chunk fill the pipeline: level L is first needed for row ya - (K - L),
  // which is produced in step t0 + 2L, so step t0 + j needs only the levels up to j / 2 (everything
  // deeper would be computed from rows outside the dependency cone of the rows this chunk stores,
  // and nothing that is stored ever reads it).  The general step evaluates all K levels every time:
  // K^2 of a chunk's (chunk + 2K) * K level evaluations, 16 % at 4096^2 (41-row chunks, K = 10).
  // These steps are therefore unrolled with a compile-time level count — straight-line code, same
  // register slots, no conditional slot writes (a run-time bound inside the general step cost
  // 25-60 VGPRs).  Nothing is stored before step t0 + 2K (row t0 + 2K - K = ya).
  static constexpr int RAMP = TB_RAMP && K >= TB_RAMP_MINK ? (2 * K + RING - 1) / RING * RING : 0;
  template <int JJ, bool YCHK>
  __device__ __forceinline__ void ramp_step(const P &p, int t) {
    constexpr int J = JJ % RING;
    constexpr int LV = JJ / 2 < K ? JJ / 2 : K;
    if (J % 3 == 0) shift_aux();
    const A cur = feed.template cur<J>(*this, p, off(t - 1 + RING));
    if constexpr (Op::kLoadsField) {
      s[0][(J + 2) % 3] = pf[J % RING];
      pf[J % RING] = Op::load(p, off(t + RING));
    }
    if constexpr (Op::kStoresAux) {
      const bool areal = st_ok && t >= ya && t < yb;
      feed.template produce<J, true>(*this, p, t, off(t + 1 + RING),
                                     areal ? p.div_out + off(t) : p.dump + (size_t)g.n + xc);
    }
    if constexpr (LV > 0) {
      const R fin = levels<J, YCHK, LV>(p, t, cur);
      if constexpr (LV == K) {
        const int ro = t - K;
        put(p, st_ok && ro >= st_a && ro < st_b, ro, fin);
        if constexpr (T > 0) tail_push<YCHK>(p, fin, ro);
      }
    }
    // chained chunks: level L = JJ / 2 has just produced the first (JJ even) or second (JJ odd) row
    // of its range; the chunk above ends its level-L range one row short of these two
    if constexpr (CHAIN && JJ >= 2 && JJ <= 2 * K - 1)
      Op::side_store(side_mine + (size_t)(JJ - 2) * SIDE_ROW, s[JJ / 2][(J + 2) % 3]);
  }
  template <int JJ, bool YCHK>
  __device__ __forceinline__ void ramp(const P &p, int t0) {
    if constexpr (JJ < RAMP) {
      ramp_step<JJ, YCHK>(p, t0 + JJ);
      ramp<JJ + 1, YCHK>(p, t0);
    }
  }

  // Chained chunks, the end of a chunk that has a chunk below it.  Level L of this chunk covers rows
  // [ya - K + L, yb - K + L): every level R rows, none computed twice.  Its last row needs level
  // L - 1 two rows further down than level L - 1's own range goes; those are the first two rows of
  // the range of the chunk below (produced in ITS first steps, ramp_step above) and come in through
  // memory.  With e = t - (yb - K): level 1 .. K compute while 2L > e, level L = e / 2 takes its row
  // from the record (e = 2L, 2L + 1: side row e - 2), levels below e / 2 are finished — the mirror
  // image of ramp().  DOWN0 ordinary steps come first, so that the unrolled tail starts a multiple of
  // RING steps after t0 when the chunk length is one.
  //
  // A side row is a load from the device's coherence point (a microsecond), and the steps get
  // shorter as levels finish (the last ones are one level: ~100 cycles), so a fixed prefetch
  // distance in steps would leave the end of every chunk waiting for memory.  The rows are
  // therefore fetched as early as registers are free for them: a finished level gives back its
  // three row slots, so B(e) = 3 + 3 * (finished levels - 1) rows may be in flight in step e, and
  // step e issues every row consumed up to step e + B(e) (all 2K - 2 rows are on their way by
  // e = K - 2, when more than half the levels still compute).  Straight-line code with compile-time
  // indices: the allocator puts sb[] into the registers the finished levels leave.
  static constexpr int DOWN0 = RING;
  static constexpr int DOWN = DOWN0 + 2 * K;
  static constexpr int side_issued(int e) {  // side rows fetched by the end of step e
    if (e < -1) return 0;
    const int fin = e / 2 - 1 > 0 ? e / 2 - 1 : 0;
    const int cnt = e + (3 + 3 * fin) - 1;
    return cnt < 0 ? 0 : cnt > SIDE_ROWS ? SIDE_ROWS : cnt;
  }
  template <int M, int M1>
  __device__ __forceinline__ void side_fetch() {
    if constexpr (M < M1) {
      sb[M] = Op::side_load(side_below + (size_t)M * SIDE_ROW);
      side_fetch<M + 1, M1>();
    }
  }
  template <int EE>
  __device__ __forceinline__ void down_step(const P &p, int t) {
    constexpr int J = EE % RING;
    constexpr int e = EE - DOWN0;
    constexpr int LO = e < 2 ? 0 : e / 2;   // the level whose row comes in from outside this step
    if (J % 3 == 0) shift_aux();
    // The steps shrink as levels finish, and a load that is RING short steps ahead is not ahead at
    // all: what nobody reads any more is not fetched.  Level l reads the divergence row that came
    // in l steps earlier, and the levels still computing are l >= e / 2: rows from steps up to
    // e = K are read (the one of step e - 1 is produced there when the launch computes the
    // divergence itself), nothing after.
    A cur = Op::zero_aux();
    if constexpr (e + RING <= K) cur = feed.template cur<J>(*this, p, off(t - 1 + RING));
    else if constexpr (e <= K) cur = feed.template cur_last<J>(*this);
    if constexpr (LO > 0) {
      s[LO][(J + 2) % 3] = sb[e - 2];
    } else if constexpr (Op::kLoadsField) {
      s[0][(J + 2) % 3] = pf[J % RING];
      if constexpr (e + RING < 2) pf[J % RING] = Op::load(p, off(t + RING));
    }
    side_fetch<side_issued(e - 1), side_issued(e)>();
    if constexpr (Op::kStoresAux && e <= K - 1) {  // divergence rows up to yb - 1
      const bool areal = st_ok && t >= ya && t < yb;
      feed.template produce<J, true, (e + RING <= K - 1)>(*this, p, t, off(t + 1 + RING),
                                                          areal ? p.div_out + off(t) : p.dump + (size_t)g.n + xc);
    }
    // every row computed here is y-interior and one this chunk stores (the host only chains
    // chunks of at least RAMP + 2 * RING rows, tb_chain_chunk_ok)
    const R fin = levels<J, false, K, LO>(p, t, cur);
    put(p, st_ok, t - K, fin);
  }
  template <int EE>
  __device__ __forceinline__ void down(const P &p, int td) {
    if constexpr (EE < DOWN) {
      down_step<EE>(p, td + EE);
      down<EE + 1>(p, td);
    }
  }

  // Steady state: every level's row is y-interior, the produced row (and divergence row) is one
  // this wave stores, and no load needs clamping — the row addresses are a uniform offset that
  // advances by one row per step plus this lane's column, the store pointers advance by a
  // per-lane stride (zero for the lanes that write the dump row).
  struct Steady {
    size_t ro;       // (t - ybase) * n, uniform
    Elem *sp;        // where this lane stores row t-K
    float *dp;       // where it stores auxiliary row t (kStoresAux)
    size_t sstride;  // n for storing lanes, 0 for dump lanes
    unsigned so, sinc, dpo, dinc;  // Op::kBufStore: the same as byte offsets into obuf / dbuf (dropped lanes stay out of range)
  };
  __device__ __forceinline__ void steady_init(const P &p, int t, Steady &st) const {
    st.ro = (size_t)(t - g.ybase) * g.n;
    st.sstride = st_ok ? (size_t)g.n : 0;
    st.sp = Op::out_ptr(p, st_ok, (size_t)(t - K - g.ybase) * g.n + xc, (size_t)xc);
    st.so = st_ok ? boff(t - K, sizeof(Elem)) : kOobOffset;
    st.sinc = st_ok ? (unsigned)g.n * (unsigned)sizeof(Elem) : 0u;
    st.dpo = st_ok ? boff(t - K - 1, sizeof(float)) : kOobOffset;
    st.dinc = st_ok ? (unsigned)g.n * (unsigned)sizeof(float) : 0u;
    st.dp = nullptr;
    if constexpr (Op::kStoresAux) st.dp = st_ok ? p.div_out + st.ro + xc : p.dump + (size_t)g.n + xc;
    if constexpr (T > 0)  // divergence row t - K - 1
      st.dp = st_ok ? p.div_out + (size_t)(t - K - 1 - g.ybase) * g.n + xc : (float *)p.dump + 2 * (size_t)g.n + xc;
  }
  template <int J>
  __device__ __forceinline__ void steady_step(const P &p, int t, Steady &st) {
    if (J % 3 == 0) shift_aux();
    const size_t n = g.n;
    const A cur = feed.template cur<J>(*this, p, st.ro + (RING - 1) * n + (unsigned)xc);
    if constexpr (Op::kLoadsField) {
      s[0][(J + 2) % 3] = pf[J % RING];
      pf[J % RING] = Op::load(p, st.ro + RING * n + (unsigned)xc);
    }
    if constexpr (Op::kStoresAux)
      feed.template produce<J, false>(*this, p, t, st.ro + (RING + 1) * n + (unsigned)xc, st.dp);
    const R fin = levels<J, false>(p, t, cur);
    if constexpr (Op::kBufStore) {
      Op::store_buf(obuf, st.so, fin);
      if constexpr (T > 0) {
        tail.template push<XEDGE, false>(fin, t - K, x0, g.n, g.ny, BufDst<Op::kCols>{dbuf, st.dpo});
        st.dpo += st.dinc;
      }
      st.so += st.sinc;
      if constexpr (Op::kStoresAux) st.dp += st.sstride;
    } else {
      Op::store_at(st.sp, fin);
      if constexpr (T > 0) tail.template push<XEDGE, false>(fin, t - K, x0, g.n, g.ny, PtrDst<Op::kCols>{st.dp});
      st.sp += st.sstride;
      if constexpr (Op::kStoresAux || T > 0) st.dp += st.sstride;
    }
    st.ro += n;
  }

  __device__ __forceinline__ void stamp(int i) {
    if (trace) trace[i] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, one counter for the whole device
  }
  __device__ __forceinline__ void run(const P &p) {
    const int t0 = ya - K, t1 = yb + K;
    stamp(0);
    init(p, t0);
    static_assert(RING == 3 || RING == 6, "the time loop is unrolled by a multiple of the slot period");
    // steady window [lo, hi] of steps, then the groups of three that lie inside it
    // chained at the bottom: the loops end DOWN0 steps before the first step that reads a side row,
    // a multiple of RING steps after t0 (chunk lengths are multiples of RING then), down() does the rest
    const bool tail = CHAIN && takes_from_below;
    const int tend = tail ? yb - K - DOWN0 : t1;
    // (steady steps store unconditionally: the produced row, and the row of the stage after it, lie in [st_a, st_b))
    const int lo = max(st_a + T + K, K + 1 + T);
    int hi = min(min((Op::kStoresAux ? st_b : st_b + K) - 1, g.ny - 1), g.ybase + g.L - 2 - RING);
    if (tail) hi = min(hi, tend - 1);
    const int tsa = t0 + RING * ((lo - t0 + RING - 1) / RING);
    const int tea = hi - (RING - 1) >= tsa ? t0 + RING * ((hi - (RING - 1) - t0) / RING) : tsa - RING;
    Steady st;
    steady_init(p, tsa, st);
    // general steps up to the steady window, the window, general steps to the end: one copy of
    // the general loop's code, run twice (up to RING-1 surplus steps at the end: loads are
    // clamped, stores dumped)
    // (one instantiation, with the y-border tests: a second one without them for the chunks that
    // touch no border was measured — nothing at 4096^2 and up, 178 -> 190 us per step at 2048^2,
    // where the extra code costs more than the tests)
    ramp<0, true>(p, t0);
    if constexpr (CHAIN) {
      if (hands_up) {  // the record is complete once the stores above have been acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag_mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    stamp(1);
    int t = t0 + RAMP;  // a multiple of RING: the phase of the general loop is unchanged
    int gend = TB_STEADY && Op::kSteady && tea >= tsa ? min(tsa, tend) : tend;
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
      for (; t < gend; t += RING) {
        // The four strips of a workgroup kept within RING rows of each other (TbChain::sync): their 4 x 0.5-1 KB
        // row pieces are adjacent in memory, and on the grids whose launches run for many rounds the waves drift
        // apart otherwise.  Pays in the divergence + pressure launch, which streams the most bytes per row
        // (16384^2 1070 -> 1009 us, 8192^2 293 -> 272), a little in the pressure launches at 6144^2-8192^2, costs
        // the diffusion launch 5 % at 8192^2, nothing anywhere below 6144^2; a barrier every row instead: 1 % worse
        // (profiles/r03/r3_tb_sync*.log).  Waves that left before the loop (padding strips) are not waited for.
        if (sync) __builtin_amdgcn_s_barrier();
        step<0>(p, t);
        step<1>(p, t + 1);
        step<2>(p, t + 2);
        if constexpr (RING == 6) {
          step<3>(p, t + 3);
          step<4>(p, t + 4);
          step<5>(p, t + 5);
        }
      }
      if (TB_STEADY && Op::kSteady && pass == 0 && gend < tend) {
        for (; t <= tea; t += RING) {
          if (sync) __builtin_amdgcn_s_barrier();
          steady_step<0>(p, t, st);
          steady_step<1>(p, t + 1, st);
          steady_step<2>(p, t + 2, st);
          if constexpr (RING == 6) {
            steady_step<3>(p, t + 3, st);
            steady_step<4>(p, t + 4, st);
            steady_step<5>(p, t + 5, st);
          }
        }
      }
      gend = tend;
    }
    if constexpr (CHAIN) {
      if (tail) {
        // The chunk below has a lower workgroup id (tb_kernel_body) and sets the flag after its
        // first RAMP steps without waiting for anything, so this loop ends; it is bounded anyway
        // (about a second), because a wave that never ends takes the machine with it: giving up
        // raises the status word (wrong rows, reported by the host as FLUID_E_DEVICE).  The flag
        // goes back to 0 for the next launch (one reader per flag).
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(
                   __hip_atomic_load(flag_below, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
          if (++spins > (1 << 20)) {
            __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        __hip_atomic_store(flag_below, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(2);
        down<0>(p, tend);
      }
    }
    stamp(3);
  }
};

// Where a wave's side records and flags are: record (chunk-row by, strip), nstr strips per chunk-row.
template <class PipeT>
__device__ __forceinline__ void chain_setup(PipeT &pipe, TbChain ch, int nstr, int by, int strip, int lane,
                                            bool chunk_below) {
  const size_t rec = (size_t)PipeT::SIDE_ROWS * PipeT::SIDE_ROW;  // 8-byte words of one record
  const int mine = by * nstr + strip;
  pipe.side_mine = ch.side + (size_t)mine * rec + lane;
  pipe.side_below = ch.side + (size_t)(mine + nstr) * rec + lane;
  pipe.flag_mine = ch.flags + mine;
  pipe.flag_below = ch.flags + mine + nstr;
  pipe.status = ch.status;
  pipe.hands_up = by > 0;
  pipe.takes_from_below = chunk_below;
}

// The rows [ya, yb) a chunk stores of a field, as a raw buffer resource (stride 0, 32-bit data format: dword 3 as
// gfx942 / gfx950 want it).  The host keeps a chunk's rows under 0xFFFFFF00 bytes (tb_chunk in fluid_api.hip
// shortens the chunk, tb_chunk_fits_buffer refuses what is still longer).
template <class T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tb_rsrc(T *field, Geom g, int ya, int yb) {
  T *base = field + (size_t)(ya - g.ybase) * g.n;
  const size_t bytes = (size_t)(yb - ya) * g.n * sizeof(T);
  return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(unsigned)bytes, 0x00020000);
}
template <class Op, class PipeT>
__device__ __forceinline__ void tb_set_buffers(PipeT &pipe, const typename Op::P &p, Geom g, int ya, int yb) {
  if constexpr (Op::kBufStore) {
    pipe.obuf = tb_rsrc(p.out, g, ya, yb);
    if constexpr (Op::kTailRows > 0) pipe.dbuf = tb_rsrc(p.div_out, g, ya, yb);
    else pipe.dbuf = pipe.obuf;
  }
}

template <class Op, int K, int C, bool CHAIN = false>
__device__ __forceinline__ void tb_kernel_body(Geom g, int r0, int r1, int chunk,
                                               const typename Op::P &p, TbChain ch = TbChain{}) {
  using SG = StripGeom<K + Op::kTailRows, C>;
  const int lane = threadIdx.x & (WAVE - 1);
  // Workgroups are handed to the 8 XCDs round-robin by linear id.  TB_XCD=1 remaps them so that
  // each XCD (own L2) works on a contiguous run of blocks (neighbouring strips share halo columns,
  // consecutive chunks share halo rows).  Measured: no difference from 4096^2 up, and 6 % slower
  // at 2048^2 (0.230 vs 0.217 ms per step), where the run of an XCD then holds all of a border's
  // slower waves; off.  (1-D grid of 8 * ceil(blocks / 8) workgroups either way, tb_grid below.)
  // Round 3: runs of 1 .. 64 consecutive workgroups per XCD, which pay in the gather kernels (tile_of_block in
  // fluid_kernels.hip): +-0.5 % here at every size (profiles/r03/r3_tb_xcd_run.log).
  const int gx = ((g.n + SG::VW - 1) / SG::VW + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int nb = gx * ((r1 - r0 + chunk - 1) / chunk);
  int id = blockIdx.x;
  if constexpr (CHAIN) {
    // Chained chunks: a chunk waits (once, near its end) for the first steps of the chunk below
    // it, so the bottom chunk-row takes the lowest workgroup ids: workgroups are handed out in
    // increasing id (round-robin over the XCDs, in order on each), so whatever a workgroup waits
    // for belongs to one with a lower id — dispatched before it on its XCD's queue, running or
    // finished, and itself waiting only for ids lower still: the lowest unfinished id never waits,
    // so every wait ends.  (A ticket from an atomic counter instead of blockIdx would not need
    // the dispatch order, and was the first form: 4096 same-address device-scope atomics cost
    // 36-92 us per launch at 16384^2, more than chaining saves.)  The wait is bounded all the
    // same (Pipe::run): a wave that gives up raises *ch.status and the host reports it.
    static_assert(K >= TB_RAMP_MINK && TB_RAMP && C == 2, "chained chunks need the unrolled fill");
    id = nb - 1 - id;
    if (id < 0) return;  // whole block: the padding of the grid
  } else {
    if (TB_XCD) id = (id % 8) * ((nb + 7) / 8) + id / 8;
    if (id >= nb) return;  // whole block: the padding of the grid
  }
  const int bx = id % gx, by = id / gx;
  const int strip = bx * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int xs = strip * SG::VW;
  if (xs >= g.n) return;  // whole wave
  const int x0 = xs - SG::HX + lane * C;
  const bool in_dom = x0 >= 0 && x0 < g.n;  // n % C == 0: a lane's columns are all in or all out
  const bool st_ok = in_dom && lane * C >= SG::HX && lane * C < SG::HX + SG::VW;
  const int xc = min(max(x0, 0), g.n - C);
  const int ya = g.ybase + r0 + by * chunk;
  const int yb = min(ya + chunk, g.ybase + r1);
  const bool xedge = xs == 0 || xs + SG::VW + SG::HX >= g.n;  // wave-uniform
  if (xedge) {
    Pipe<Op, K, true, CHAIN> pipe;
    pipe.g = g; pipe.x0 = x0; pipe.xc = xc; pipe.st_ok = st_ok;
    pipe.st_a = ya; pipe.st_b = yb; pipe.ya = ya - Op::kTailRows; pipe.yb = yb + Op::kTailRows;
    pipe.trace = ch.trace ? ch.trace + (size_t)(by * gx * WAVES_PER_BLOCK + strip) * 4 : nullptr;
    pipe.sync = ch.sync != 0;
    pipe.xb = x_border(x0, g.n);
    tb_set_buffers<Op>(pipe, p, g, ya, yb);
    if constexpr (CHAIN) chain_setup(pipe, ch, gx * WAVES_PER_BLOCK, by, strip, lane, yb < g.ybase + r1);
    pipe.run(p);
  } else {
    Pipe<Op, K, false, CHAIN> pipe;
    pipe.g = g; pipe.x0 = x0; pipe.xc = xc; pipe.st_ok = st_ok;
    pipe.st_a = ya; pipe.st_b = yb; pipe.ya = ya - Op::kTailRows; pipe.yb = yb + Op::kTailRows;
    pipe.trace = ch.trace ? ch.trace + (size_t)(by * gx * WAVES_PER_BLOCK + strip) * 4 : nullptr;
    pipe.sync = ch.sync != 0;
    pipe.xb = XBorder{0, 0};
    tb_set_buffers<Op>(pipe, p, g, ya, yb);
    if constexpr (CHAIN) chain_setup(pipe, ch, gx * WAVES_PER_BLOCK, by, strip, lane, yb < g.ybase + r1);
    pipe.run(p);
  }
}

template <int K, int C, bool CHAIN = false>
// (Holding the 13-sweep form to 128 VGPRs for a fourth wave per SIMD was tried twice, with
// __launch_bounds__(256, 4) and amdgpu_waves_per_eu: 7-8 spilled registers, 0.967 -> 1.133 ms
// per launch at 16384^2.  Left to the register allocator: 132 VGPRs.)
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_jacobi_tb(
    Geom g, int r0, int r1, int chunk, const float *__restrict__ pin,
    const float *__restrict__ div, float *__restrict__ pout, float *__restrict__ dump, TbChain ch) {
  typename JacobiOp<C>::P p{pin, div, pout, dump};
  tb_kernel_body<JacobiOp<C>, K, C, CHAIN>(g, r0, r1, chunk, p, ch);
}

template <int K, int C, bool CHAIN = false>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_jacobi_div_tb(
    Geom g, int r0, int r1, int chunk, const float2 *__restrict__ vel, float *__restrict__ div_out,
    float *__restrict__ pout, float *__restrict__ dump, TbChain ch) {
  typename JacobiDivOp<C>::P p{vel, div_out, pout, dump, g.n};
  tb_kernel_body<JacobiDivOp<C>, K, C, CHAIN>(g, r0, r1, chunk, p, ch);
}

template <int K, int C, bool CHAIN = false>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_diffuse_tb(
    Geom g, int r0, int r1, int chunk, float alpha, float beta, const float2 *__restrict__ vin,
    float2 *__restrict__ vout, float2 *__restrict__ dump, TbChain ch) {
  typename DiffuseOp<C>::P p{vin, vout, dump, alpha, beta};
  tb_kernel_body<DiffuseOp<C>, K, C, CHAIN>(g, r0, r1, chunk, p, ch);
}

template <int K, int C>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK, TB_DIV_WAVES) void k_diffuse_div_tb(
    Geom g, int r0, int r1, int chunk, float alpha, float beta, const float2 *__restrict__ vin,
    float2 *__restrict__ vout, float *__restrict__ div_out, float2 *__restrict__ dump, TbChain ch) {
  typename DiffuseDivOp<C>::P p{vin, vout, dump, alpha, beta, div_out};
  tb_kernel_body<DiffuseDivOp<C>, K, C, false>(g, r0, r1, chunk, p, ch);
}

template <int K, int C>
__global__ __launch_bounds__(WAVE *WAVES_PER_BLOCK) void k_jacobi0_tb(
    Geom g, int r0, int r1, int chunk, const float *__restrict__ div, float *__restrict__ pout,
    float *__restrict__ dump, TbChain ch) {
  typename JacobiZeroOp<C>::P p{nullptr, div, pout, dump};
  tb_kernel_body<JacobiZeroOp<C>, K, C, false>(g, r0, r1, chunk, p, ch);
}

// A chunk's stored rows are addressed through a buffer resource with 32-bit offsets (TB_BUFST): refused beyond
// that (a launch plan of very few chunk-rows on a grid of tens of thousands of columns; never a default plan).
inline bool tb_chunk_fits_buffer(Geom g, int chunk, size_t elem) {
  return !TB_BUFST || (unsigned long long)chunk * (unsigned long long)g.n * elem <= 0xFFFFFF00ull;
}

template <int K, int C>
dim3 tb_grid(Geom g, int r0, int r1, int chunk) {
  using SG = StripGeom<K, C>;
  const int strips = (g.n + SG::VW - 1) / SG::VW;
  const int nb = ((strips + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) * ((r1 - r0 + chunk - 1) / chunk);
  return dim3((nb + 7) / 8 * 8);  // see tb_kernel_body: padded so the XCD remap is a bijection
}

// chained: nullptr, or where the side records, flags and ticket counter of this context are (the
// caller has checked that the launch fits them, tb_chain_fits)
template <int K, int C>
void jacobi_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, const float *pin,
                      const float *div, float *pout, void *dump, const TbChain *chained) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk), block(WAVE * WAVES_PER_BLOCK);
  if constexpr (C == 2 && K >= TB_CHAIN_MINK && K >= TB_RAMP_MINK) {
    if (chained && chained->side) {
      k_jacobi_tb<K, C, true><<<grid, block, 0, s>>>(g, r0, r1, chunk, pin, div, pout, (float *)dump, *chained);
      return;
    }
  }
  k_jacobi_tb<K, C><<<grid, block, 0, s>>>(g, r0, r1, chunk, pin, div, pout, (float *)dump, chained ? *chained : TbChain{});
}
template <int K, int C>
void jacobi_div_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, const float2 *vel,
                          float *div_out, float *pout, void *dump, const TbChain *chained) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk), block(WAVE * WAVES_PER_BLOCK);
  if constexpr (C == 2 && K >= TB_CHAIN_MINK && K >= TB_RAMP_MINK) {
    if (chained && chained->side) {
      k_jacobi_div_tb<K, C, true><<<grid, block, 0, s>>>(g, r0, r1, chunk, vel, div_out, pout, (float *)dump, *chained);
      return;
    }
  }
  k_jacobi_div_tb<K, C><<<grid, block, 0, s>>>(g, r0, r1, chunk, vel, div_out, pout, (float *)dump, chained ? *chained : TbChain{});
}
template <int K, int C>
void diffuse_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, float alpha, float beta,
                       const float2 *vin, float2 *vout, void *dump, const TbChain *chained) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk), block(WAVE * WAVES_PER_BLOCK);
  if constexpr (C == 2 && K >= TB_CHAIN_MINK && K >= TB_RAMP_MINK) {
    if (chained && chained->side) {
      k_diffuse_tb<K, C, true><<<grid, block, 0, s>>>(g, r0, r1, chunk, alpha, beta, vin, vout, (float2 *)dump, *chained);
      return;
    }
  }
  k_diffuse_tb<K, C><<<grid, block, 0, s>>>(g, r0, r1, chunk, alpha, beta, vin, vout, (float2 *)dump, chained ? *chained : TbChain{});
}

template <int K, int C>
void diffuse_div_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, float alpha, float beta,
                           const float2 *vin, float2 *vout, float *div_out, void *dump, const TbChain *extra) {
  const dim3 grid = tb_grid<K + 1, C>(g, r0, r1, chunk), block(WAVE * WAVES_PER_BLOCK);  // strips one column narrower at either side
  k_diffuse_div_tb<K, C><<<grid, block, 0, s>>>(g, r0, r1, chunk, alpha, beta, vin, vout, div_out, (float2 *)dump,
                                                extra ? *extra : TbChain{});
}
template <int K, int C>
void jacobi0_tb_launch(hipStream_t s, Geom g, int r0, int r1, int chunk, const float *div, float *pout, void *dump,
                       const TbChain *extra) {
  const dim3 grid = tb_grid<K, C>(g, r0, r1, chunk), block(WAVE * WAVES_PER_BLOCK);
  k_jacobi0_tb<K, C><<<grid, block, 0, s>>>(g, r0, r1, chunk, div, pout, (float *)dump, extra ? *extra : TbChain{});
}

// Workgroups of one kernel that are resident on the whole device at once (occupancy from the
// runtime: registers decide it — 108-112 VGPRs: 4 waves per SIMD, 132-160: 3).
template <class KernelT>
void resident_blocks_of(KernelT kernel, int *out) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE * WAVES_PER_BLOCK, 0) != hipSuccess ||
      hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    *out = 0;
    return;
  }
  *out = per_cu * cus;
}
template <int K, int C> void jacobi_tb_occ(int *out) { resident_blocks_of(k_jacobi_tb<K, C>, out); }
template <int K, int C> void jacobi_div_tb_occ(int *out) { resident_blocks_of(k_jacobi_div_tb<K, C>, out); }
template <int K, int C> void diffuse_tb_occ(int *out) { resident_blocks_of(k_diffuse_tb<K, C>, out); }
template <int K, int C> void diffuse_div_tb_occ(int *out) { resident_blocks_of(k_diffuse_div_tb<K, C>, out); }
template <int K, int C> void jacobi0_tb_occ(int *out) { resident_blocks_of(k_jacobi0_tb<K, C>, out); }

}  // namespace

// Instantiated sweep counts.  Four columns per lane stop at 5 sweeps, two columns at 10
// (diffusion, two planes) / 14 (pressure): beyond that the pipeline no longer fits 256 VGPRs.
#define TB_DISPATCH_K4(FN, ...)                      \
  switch (k) {                                       \
    case 1: FN<1, 4>(__VA_ARGS__); return true;      \
    case 2: FN<2, 4>(__VA_ARGS__); return true;      \
    case 4: FN<4, 4>(__VA_ARGS__); return true;      \
    case 5: FN<5, 4>(__VA_ARGS__); return true;      \
    default: return false;                           \
  }
#define TB_DISPATCH_K2(FN, ...)                      \
  switch (k) {                                       \
    case 1: FN<1, 2>(__VA_ARGS__); return true;      \
    case 2: FN<2, 2>(__VA_ARGS__); return true;      \
    case 4: FN<4, 2>(__VA_ARGS__); return true;      \
    case 5: FN<5, 2>(__VA_ARGS__); return true;      \
    case 8: FN<8, 2>(__VA_ARGS__); return true;      \
    case 10: FN<10, 2>(__VA_ARGS__); return true;    \
    default: break;                                  \
  }
#define TB_DISPATCH_K2_DEEP(FN, ...)                 \
  switch (k) {                                       \
    case 13: FN<13, 2>(__VA_ARGS__); return true;    \
    case 14: FN<14, 2>(__VA_ARGS__); return true;    \
    default: break;                                  \
  }

// The instantiations are split over three translation units so that they compile side by side (the Makefile
// builds this file three times, -DTB_PART=0 / 1 / 2; without TB_PART everything is emitted, as the ISA test wants):
// part 0 the pressure launches and the helpers, part 1 the two first-pressure-launch forms, part 2 the diffusion ones.
#ifdef TB_PART
#define TB_IN_PART(n) (TB_PART == (n))
#else
#define TB_IN_PART(n) 1
#endif
bool tb_occ_part1(int kind, int k, int cols, int *out);
bool tb_occ_part2(int kind, int k, int cols, int *out);

#if TB_IN_PART(0)
bool tb_supported(int n, int cols) { return (cols == 2 || cols == 4) && n % cols == 0 && n >= 8; }
bool tb_has_k(int k, int cols, bool float2_field) {
  switch (k) {
    case 1: case 2: case 4: case 5: return true;
    case 8: case 10: return cols == 2;
    case 13: case 14: return cols == 2 && !float2_field;
    default: return false;
  }
}

// Chained chunks (Pipe::run): what a launch must look like, and what it needs in the context's
// side buffer.  The unrolled tail assumes a chunk length that is a multiple of RING (its phase), a
// chunk long enough for the fill and the tail not to overlap and for every row of the tail to be
// one the chunk stores, and somebody to chain to.
bool tb_chain_chunk_ok(int k, int cols, int chunk, int rows) {
  const int ramp = (2 * k + RING - 1) / RING * RING;
  return TB_RAMP && cols == 2 && k >= TB_RAMP_MINK && k >= TB_CHAIN_MINK && chunk % RING == 0 && chunk >= ramp + 2 * RING && rows > chunk;
}
int tb_chain_min_chunk(int k) { return (2 * k + RING - 1) / RING * RING + 2 * RING; }
int tb_chain_chunk_multiple() { return RING; }
// 8-byte words of side records / flags a launch of `chunks` chunk-rows uses
size_t tb_chain_words(int n, int k, int cols, bool float2_field, int chunks) {
  const int nstr = tb_blocks_per_row(n, k, cols) * WAVES_PER_BLOCK;
  const size_t row = (size_t)(float2_field ? cols : (cols + 1) / 2) * WAVE;
  return (size_t)chunks * nstr * (2 * k - 2) * row;
}
size_t tb_chain_flags(int n, int k, int cols, int chunks) {
  return (size_t)chunks * tb_blocks_per_row(n, k, cols) * WAVES_PER_BLOCK;
}

bool launch_jacobi_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                      const float *pin, const float *div, float *pout, void *dump, const TbChain *chained) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1 || !tb_has_k(k, cols, false) || !tb_chunk_fits_buffer(g, chunk, sizeof(float))) return false;
  TbChain plain{};  // not chainable: only the trace pointer goes along
  if (chained && (!chained->side || !tb_chain_chunk_ok(k, cols, chunk, r1 - r0))) { plain.trace = chained->trace; plain.sync = chained->sync; chained = &plain; }
  if (cols == 4) TB_DISPATCH_K4(jacobi_tb_launch, s, g, r0, r1, chunk, pin, div, pout, dump, chained)
  TB_DISPATCH_K2(jacobi_tb_launch, s, g, r0, r1, chunk, pin, div, pout, dump, chained)
  TB_DISPATCH_K2_DEEP(jacobi_tb_launch, s, g, r0, r1, chunk, pin, div, pout, dump, chained)
  return false;
}

#endif  // part 0
#if TB_IN_PART(1)
bool launch_jacobi_div_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                          const float2 *vel, float *div_out, float *pout, void *dump, const TbChain *chained) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1 || !tb_has_k(k, cols, false) || !tb_chunk_fits_buffer(g, chunk, sizeof(float))) return false;
  TbChain plain{};  // not chainable: only the trace pointer goes along
  if (chained && (!chained->side || !tb_chain_chunk_ok(k, cols, chunk, r1 - r0))) { plain.trace = chained->trace; plain.sync = chained->sync; chained = &plain; }
  if (cols == 4) TB_DISPATCH_K4(jacobi_div_tb_launch, s, g, r0, r1, chunk, vel, div_out, pout, dump, chained)
  TB_DISPATCH_K2(jacobi_div_tb_launch, s, g, r0, r1, chunk, vel, div_out, pout, dump, chained)
  TB_DISPATCH_K2_DEEP(jacobi_div_tb_launch, s, g, r0, r1, chunk, vel, div_out, pout, dump, chained)
  return false;
}

#endif  // part 1
#if TB_IN_PART(2)
bool launch_diffuse_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk,
                       float alpha, float beta, const float2 *vin, float2 *vout, void *dump, const TbChain *chained) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1 || !tb_has_k(k, cols, true) || !tb_chunk_fits_buffer(g, chunk, sizeof(float2))) return false;
  TbChain plain{};  // not chainable: only the trace pointer goes along
  if (chained && (!chained->side || !tb_chain_chunk_ok(k, cols, chunk, r1 - r0))) { plain.trace = chained->trace; plain.sync = chained->sync; chained = &plain; }
  if (cols == 4) TB_DISPATCH_K4(diffuse_tb_launch, s, g, r0, r1, chunk, alpha, beta, vin, vout, dump, chained)
  TB_DISPATCH_K2(diffuse_tb_launch, s, g, r0, r1, chunk, alpha, beta, vin, vout, dump, chained)
  return false;
}

// The last diffusion launch of a whole step, also writing the divergence of its result (DiffuseDivOp); the two
// sweep counts a diffusion plan ends with on the grids that take this path.
bool tb_has_div_k(int k, int cols) { return cols == 2 && (k == 8 || k == 10); }
bool launch_diffuse_div_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk, float alpha, float beta,
                           const float2 *vin, float2 *vout, float *div_out, void *dump, const TbChain *extra) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1 || !tb_has_div_k(k, cols) || !tb_chunk_fits_buffer(g, chunk, sizeof(float2))) return false;
  TbChain plain{};  // never chained: the trace pointer and the barrier flag go along
  if (extra) { plain.trace = extra->trace; plain.sync = extra->sync; }
  if (k == 8) diffuse_div_tb_launch<8, 2>(s, g, r0, r1, chunk, alpha, beta, vin, vout, div_out, dump, &plain);
  else diffuse_div_tb_launch<10, 2>(s, g, r0, r1, chunk, alpha, beta, vin, vout, div_out, dump, &plain);
  return true;
}
bool tb_occ_part2(int kind, int k, int cols, int *out) {
  if (kind == 3) {
    if (!tb_has_div_k(k, cols)) return false;
    if (k == 8) diffuse_div_tb_occ<8, 2>(out); else diffuse_div_tb_occ<10, 2>(out);
    return true;
  }
  if (cols == 4) TB_DISPATCH_K4(diffuse_tb_occ, out)
  TB_DISPATCH_K2(diffuse_tb_occ, out)
  return false;
}
#endif  // part 2
#if TB_IN_PART(1)
// The first pressure launch after it: k sweeps from p^0 == 0 with the divergence read from memory (JacobiZeroOp).
bool launch_jacobi0_tb(hipStream_t s, Geom g, int r0, int r1, int k, int cols, int chunk, const float *div,
                       float *pout, void *dump, const TbChain *extra) {
  if (r1 <= r0) return true;
  if (!tb_supported(g.n, cols) || chunk < 1 || !tb_has_k(k, cols, false) || !tb_chunk_fits_buffer(g, chunk, sizeof(float))) return false;
  TbChain plain{};
  if (extra) { plain.trace = extra->trace; plain.sync = extra->sync; }
  extra = &plain;
  if (cols == 4) TB_DISPATCH_K4(jacobi0_tb_launch, s, g, r0, r1, chunk, div, pout, dump, extra)
  TB_DISPATCH_K2(jacobi0_tb_launch, s, g, r0, r1, chunk, div, pout, dump, extra)
  TB_DISPATCH_K2_DEEP(jacobi0_tb_launch, s, g, r0, r1, chunk, div, pout, dump, extra)
  return false;
}

bool tb_occ_part1(int kind, int k, int cols, int *out) {
  if (kind == 4) {
    if (cols == 4) TB_DISPATCH_K4(jacobi0_tb_occ, out)
    TB_DISPATCH_K2(jacobi0_tb_occ, out)
    TB_DISPATCH_K2_DEEP(jacobi0_tb_occ, out)
    return false;
  }
  if (cols == 4) TB_DISPATCH_K4(jacobi_div_tb_occ, out)
  TB_DISPATCH_K2(jacobi_div_tb_occ, out)
  TB_DISPATCH_K2_DEEP(jacobi_div_tb_occ, out)
  return false;
}
#endif  // part 1
#if TB_IN_PART(0)
// Resident workgroups (4 waves each) of the k-sweep kernel of `kind` (0 pressure, 1 divergence + pressure,
// 2 diffusion, 3 diffusion + divergence, 4 pressure from p = 0) on the current device; 0 if unknown.  Asked
// once per kernel.
namespace {
bool occ_dispatch(int kind, int k, int cols, int *out) {
  if (kind == 2 || kind == 3) return tb_occ_part2(kind, k, cols, out);
  if (kind == 1 || kind == 4) return tb_occ_part1(kind, k, cols, out);
  if (cols == 4) TB_DISPATCH_K4(jacobi_tb_occ, out)
  TB_DISPATCH_K2(jacobi_tb_occ, out)
  TB_DISPATCH_K2_DEEP(jacobi_tb_occ, out)
  return false;
}
}  // namespace
int tb_resident_blocks(int kind, int k, int cols) {
  // per device (the answer is the current device's: callers have set the context's), one lock for all
  constexpr int kDevs = 64;
  static int cache[kDevs][5][16][5];  // 0 = not asked yet, -1 = unknown
  static std::mutex mu;
  int dev = 0;
  if (kind < 0 || kind > 4 || k < 1 || k > 15 || cols < 1 || cols > 4) return 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) return 0;
  std::lock_guard<std::mutex> lock(mu);
  int &slot = cache[dev][kind][k][cols];
  if (slot == 0) {
    int v = 0;
    slot = occ_dispatch(kind, k, cols, &v) && v > 0 ? v : -1;
  }
  return slot > 0 ? slot : 0;
}
// blocks one chunk-row of the launch has (a workgroup is WAVES_PER_BLOCK strips wide)
int tb_blocks_per_row(int n, int k, int cols) {
  const int hx = tb_hx(k, cols), vw = WAVE * cols - 2 * hx;
  const int strips = (n + vw - 1) / vw;
  return (strips + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
}
#endif  // part 0
