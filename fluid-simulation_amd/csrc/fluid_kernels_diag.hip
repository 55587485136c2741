// fluid_kernels_diag.hip — diagnostics of the current state by wavefront reductions.
//
// No counterpart in the reference (kernel.cu has no residual or diagnostic of any kind; its only
// read-back is the picture, kernel.cu:710-756).  What is measured is defined with the
// reference's own formulas so that it can be checked bit for bit:
//   residual   |p' - p| where p' is one more pressureJacobi sweep of p (kernel.cu:313)
//   div(u)     the `divergence` kernel's formula (kernel.cu:282) applied to the current velocity
// plus plain maxima of |u|, |v|, |p|, |div| and the sum of each dye channel.
//
// Reduction: every lane keeps its own running values over the cells it visits; a wave64 combines
// them with a DPP butterfly (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: 16 lanes
// without touching LDS) and two cross-row exchanges; the four waves of a block meet in LDS; one
// partial record per block goes to memory and a second, single-block launch folds the records
// in index order.  No atomics, fixed order: the same state gives the same bits every time.
#include "fluid_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int DIAG_THREADS = 256;
constexpr int NWAVES = DIAG_THREADS / 64;

template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// maximum that keeps a NaN once it has seen one (a diagnostic must not hide a blown-up field)
__device__ __forceinline__ float max_nan(float a, float b) { return (b > a || b != b) ? b : a; }
struct MaxOp { __device__ __forceinline__ float operator()(float a, float b) const { return max_nan(a, b); } };
struct AddOp { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };

// All 64 lanes must be active.  Every lane ends up with the wave's result (both partners of an
// exchange compute the same commutative combination).
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  v = op(v, dpp_perm<0xB1>(v));   // quad_perm:[1,0,3,2]  lane ^ 1
  v = op(v, dpp_perm<0x4E>(v));   // quad_perm:[2,3,0,1]  lane ^ 2
  v = op(v, dpp_perm<0x141>(v));  // row_half_mirror      quad <-> quad within 8 lanes
  v = op(v, dpp_perm<0x140>(v));  // row_mirror           half <-> half within 16 lanes
  v = op(v, __shfl_xor(v, 16));
  v = op(v, __shfl_xor(v, 32));
  return v;
}

// kernel.cu:313 (interior rule of one more sweep) minus the current iterate
__device__ __forceinline__ float jacobi_update(const float *__restrict__ p, const float *__restrict__ div,
                                               size_t id, int n) {
  const float vL = p[id - 1], vR = p[id + 1], vB = p[id - n], vT = p[id + n];
  return ((((vL + vR) + vB) + vT) + (-1.0f) * div[id]) * 0.25f - p[id];
}

__global__ __launch_bounds__(DIAG_THREADS) void k_diagnostics(Geom g, int r0, int r1,
                                                              const float2 *__restrict__ vel,
                                                              const float4 *__restrict__ dens,
                                                              const float *__restrict__ p,
                                                              const float *__restrict__ div,
                                                              float *__restrict__ partial) {
  float mx[DIAG_NMAX], sm[DIAG_NSUM];
#pragma unroll
  for (int k = 0; k < DIAG_NMAX; ++k) mx[k] = 0.0f;  // every quantity is an absolute value
#pragma unroll
  for (int k = 0; k < DIAG_NSUM; ++k) sm[k] = 0.0f;
  for (int r = r0 + (int)blockIdx.x; r < r1; r += (int)gridDim.x) {
    const int y = g.ybase + r;
    const bool yin = y >= 1 && y <= g.ny - 2;
    for (int x = (int)threadIdx.x; x < g.n; x += DIAG_THREADS) {
      const size_t id = (size_t)r * g.n + x;
      const float2 u = vel[id];
      const float4 d = dens[id];
      mx[0] = max_nan(mx[0], fabsf(u.x));
      mx[1] = max_nan(mx[1], fabsf(u.y));
      mx[2] = max_nan(mx[2], fabsf(div[id]));
      mx[3] = max_nan(mx[3], fabsf(p[id]));
      if (yin && x >= 1 && x <= g.n - 2) {
        mx[4] = max_nan(mx[4], fabsf(jacobi_update(p, div, id, g.n)));
        const float2 vL = vel[id - 1], vR = vel[id + 1], vB = vel[id - g.n], vT = vel[id + g.n];
        mx[5] = max_nan(mx[5], fabsf(0.5f * ((vR.x - vL.x) + (vT.y - vB.y))));  // kernel.cu:282
      }
      sm[0] += d.x; sm[1] += d.y; sm[2] += d.z; sm[3] += d.w;
    }
  }
  // wave level (all lanes are here: no early exits above)
#pragma unroll
  for (int k = 0; k < DIAG_NMAX; ++k) mx[k] = wave_reduce(mx[k], MaxOp());
#pragma unroll
  for (int k = 0; k < DIAG_NSUM; ++k) sm[k] = wave_reduce(sm[k], AddOp());
  // block level
  __shared__ float acc[NWAVES][DIAG_NREC];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < DIAG_NMAX; ++k) acc[wave][k] = mx[k];
#pragma unroll
    for (int k = 0; k < DIAG_NSUM; ++k) acc[wave][DIAG_NMAX + k] = sm[k];
  }
  __syncthreads();
  if (threadIdx.x < DIAG_NREC) {
    const int k = (int)threadIdx.x;
    float v = acc[0][k];
    for (int w = 1; w < NWAVES; ++w) v = k < DIAG_NMAX ? max_nan(v, acc[w][k]) : v + acc[w][k];
    partial[(size_t)blockIdx.x * DIAG_NREC + k] = v;
  }
}

// Folds the per-block records in a fixed order: maxima as floats, sums in double.
__global__ __launch_bounds__(DIAG_THREADS) void k_diagnostics_fold(const float *__restrict__ partial, int nrec,
                                                                   DiagResult *__restrict__ out) {
  __shared__ float fmx[DIAG_THREADS][DIAG_NMAX];
  __shared__ double fsm[DIAG_THREADS][DIAG_NSUM];
  const int t = (int)threadIdx.x;
  float mx[DIAG_NMAX];
  double sm[DIAG_NSUM];
#pragma unroll
  for (int k = 0; k < DIAG_NMAX; ++k) mx[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < DIAG_NSUM; ++k) sm[k] = 0.0;
  for (int b = t; b < nrec; b += DIAG_THREADS) {
    const float *rec = partial + (size_t)b * DIAG_NREC;
#pragma unroll
    for (int k = 0; k < DIAG_NMAX; ++k) mx[k] = max_nan(mx[k], rec[k]);
#pragma unroll
    for (int k = 0; k < DIAG_NSUM; ++k) sm[k] += (double)rec[DIAG_NMAX + k];
  }
#pragma unroll
  for (int k = 0; k < DIAG_NMAX; ++k) fmx[t][k] = mx[k];
#pragma unroll
  for (int k = 0; k < DIAG_NSUM; ++k) fsm[t][k] = sm[k];
  __syncthreads();
  if (t < DIAG_NMAX) {
    float v = fmx[0][t];
    for (int i = 1; i < DIAG_THREADS; ++i) v = max_nan(v, fmx[i][t]);
    out->mx[t] = v;
  } else if (t < DIAG_NREC) {
    double v = fsm[0][t - DIAG_NMAX];
    for (int i = 1; i < DIAG_THREADS; ++i) v += fsm[i][t - DIAG_NMAX];
    out->sm[t - DIAG_NMAX] = v;
  }
}

// Sum of the 32-bit words of a block of rows, modulo 2^64: a checksum that does not leave the
// device (bench_multi.py compares slab rows with whole-domain rows by it instead of reading 7 GB per
// rank through the host).  Integer addition: any order gives the same value, so one 64-bit atomic per
// workgroup is deterministic here (the float reductions above use a fixed order instead).
__global__ __launch_bounds__(256) void k_checksum(const uint4 *__restrict__ v, size_t n16,
                                                  const unsigned *__restrict__ tail, int ntail,
                                                  unsigned long long *__restrict__ out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 t = v[i];
    acc += (unsigned long long)t.x + t.y + t.z + t.w;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) acc += tail[threadIdx.x];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

// words: 32-bit words from base (16-byte aligned: whole rows of a field); *out must be zero
void launch_checksum(hipStream_t s, const void *base, size_t words, unsigned long long *out) {
  const size_t n16 = words / 4;
  const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_checksum, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, (const uint4 *)base, n16,
                     (const unsigned *)base + n16 * 4, (int)(words - n16 * 4), out);
}

static int diag_blocks(int rows) { return rows < DIAG_MAX_BLOCKS ? rows : DIAG_MAX_BLOCKS; }

void launch_diagnostics(hipStream_t s, Geom g, int r0, int r1, const float2 *vel, const float4 *dens,
                        const float *p, const float *div, float *partial, DiagResult *out) {
  const int nb = diag_blocks(r1 - r0);
  hipLaunchKernelGGL(k_diagnostics, dim3(nb), dim3(DIAG_THREADS), 0, s, g, r0, r1, vel, dens, p, div, partial);
  hipLaunchKernelGGL(k_diagnostics_fold, dim3(1), dim3(DIAG_THREADS), 0, s, partial, nb, out);
}
