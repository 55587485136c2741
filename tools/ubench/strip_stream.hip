// Micro-benchmark: what HBM bandwidth does the temporal-blocking kernels' access pattern reach
// by itself?  A wave64 owns a vertical strip of 64*C float columns and streams down a chunk of
// rows with a RING-deep register prefetch, reading NIN input fields and writing one — the memory
// side of k_jacobi_tb without the sweeps.  Variants: x-halo overlap (H columns re-read on each
// side, stores of halo lanes sent to a dump row or masked off), XCD-aware block order, C = 2 / 4,
// and a plain row-major copy as the ceiling.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int C> struct Vec;
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };
__device__ inline float2 add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float4 add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// MODE 0: halo lanes store to a dump row; 1: halo lanes masked off (exec);
template <int C, int NIN, int RING, int MODE, bool XCD>
__global__ __launch_bounds__(256) void k_strip(int n, int H, int chunk, int nbx, const float *__restrict__ a,
                                               const float *__restrict__ b, float *__restrict__ out,
                                               float *__restrict__ dump) {
  using V = typename Vec<C>::T;
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {  // consecutive workgroup ids go to consecutive XCDs: give each XCD a contiguous run
    const int nb = gridDim.x * gridDim.y, id = by * gridDim.x + bx;
    const int per = (nb + 7) / 8;
    const int nid = (id % 8) * per + id / 8;
    if (nid >= nb) return;
    bx = nid % gridDim.x; by = nid / gridDim.x;
  }
  const int lane = threadIdx.x & 63;
  const int VW = 64 * C - 2 * H;
  const int strip = bx * 4 + (threadIdx.x >> 6);
  const int xs = strip * VW;
  if (xs >= n) return;
  const int x0 = xs - H + lane * C;
  const bool st_ok = x0 >= 0 && x0 < n && lane * C >= H && lane * C < H + VW;
  const int xc = min(max(x0, 0), n - C);
  const int ya = by * chunk, yb = min(ya + chunk, n);
  V pa[RING], pb[RING];
  auto rowoff = [&](int t) { return (size_t)min(t, n - 1) * n + xc; };
#pragma unroll
  for (int i = 0; i < RING; ++i) {
    pa[i] = *(const V *)(a + rowoff(ya + i));
    if (NIN > 1) pb[i] = *(const V *)(b + rowoff(ya + i));
  }
  for (int t = ya; t < yb; t += RING) {
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      V v = pa[j];
      if (NIN > 1) v = add(v, pb[j]);
      pa[j] = *(const V *)(a + rowoff(t + j + RING));
      if (NIN > 1) pb[j] = *(const V *)(b + rowoff(t + j + RING));
      const bool real = t + j < yb;
      if (MODE == 0) {
        float *q = (st_ok && real) ? out + (size_t)(t + j) * n + xc : dump + xc;
        *(V *)q = v;
      } else {
        if (st_ok && real) *(V *)(out + (size_t)(t + j) * n + xc) = v;
      }
    }
  }
}


// Burst variant: rows are fetched B at a time, back to back (the B rows of one group), one group
// ahead; WPB waves per block (adjacent strips); NT: non-temporal loads and stores.
template <int B, int WPB, bool NT, bool XCD>
__global__ __launch_bounds__(WPB * 64) void k_burst(int n, int pitch, int chunk, const float *__restrict__ a,
                                                   const float *__restrict__ b, float *__restrict__ out) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {
    const int nb = gridDim.x * gridDim.y, id = by * gridDim.x + bx;
    const int per = (nb + 7) / 8;
    const int nid = (id % 8) * per + id / 8;
    if (nid >= nb) return;
    bx = nid % gridDim.x; by = nid / gridDim.x;
  }
  const int lane = threadIdx.x & 63;
  const int strip = bx * WPB + (threadIdx.x >> 6);
  const int xc = strip * 128 + lane * 2;
  if (xc >= n) return;
  const int ya = by * chunk, yb = min(ya + chunk, n);
  float2 pa[B], pb[B];
  auto ld = [&](const float *f, int t) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v *q = pitch > 0 ? (const f2v *)(f + (size_t)min(t, n - 1) * pitch + xc)
                             : (const f2v *)(f + ((size_t)strip * n + min(t, n - 1)) * 128 + lane * 2);
    const f2v r = NT ? __builtin_nontemporal_load(q) : *q;
    return make_float2(r.x, r.y);
  };
#pragma unroll
  for (int i = 0; i < B; ++i) { pa[i] = ld(a, ya + i); pb[i] = ld(b, ya + i); }
  for (int t = ya; t < yb; t += B) {
    float2 v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = add(pa[j], pb[j]);
#pragma unroll
    for (int j = 0; j < B; ++j) { pa[j] = ld(a, t + B + j); pb[j] = ld(b, t + B + j); }
#pragma unroll
    for (int j = 0; j < B; ++j)
      if (t + j < yb) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v *q = pitch > 0 ? (f2v *)(out + (size_t)(t + j) * pitch + xc)
                           : (f2v *)(out + ((size_t)strip * n + t + j) * 128 + lane * 2);
        const f2v w = {v[j].x, v[j].y};
        if (NT) __builtin_nontemporal_store(w, q); else *q = w;
      }
  }
}
template <int B, int WPB, bool NT, bool XCD>
void run_burst(const char *name, int n, int pitch, int waves, float *a, float *b, float *out) {
  const int strips = n / 128;
  int chunks = waves / strips; if (chunks < 1) chunks = 1;
  const int chunk = (n + chunks - 1) / chunks;
  dim3 grid((strips + WPB - 1) / WPB, (n + chunk - 1) / chunk);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) k_burst<B, WPB, NT, XCD><<<grid, WPB * 64>>>(n, pitch, chunk, a, b, out);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) k_burst<B, WPB, NT, XCD><<<grid, WPB * 64>>>(n, pitch, chunk, a, b, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-12s pitch=n+%-4d burst=%d wpb=%2d nt=%d xcd=%d waves=%6d chunk=%4d : %.3f ms  %.0f GB/s\n", name, pitch - n, B, WPB, (int)NT, (int)XCD,
         strips * (int)grid.y, chunk, ms, (double)n * n * 12 / ms / 1e6);
}

// MODE 0: loads only (sum kept in registers, one store at the end); 1: stores only; 2: copy with
// stores issued in bursts of B rows after the next burst of loads.
template <int B, int MODE>
__global__ __launch_bounds__(256) void k_dir(int n, int chunk, const float *__restrict__ a,
                                             const float *__restrict__ b, float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int xc = strip * 128 + lane * 2;
  if (xc >= n) return;
  const int ya = blockIdx.y * chunk, yb = min(ya + chunk, n);
  float2 pa[B], pb[B], acc = make_float2(0.f, 0.f);
  auto ld = [&](const float *f, int t) { return *(const float2 *)(f + (size_t)min(t, n - 1) * n + xc); };
  if (MODE != 1) {
#pragma unroll
    for (int i = 0; i < B; ++i) { pa[i] = ld(a, ya + i); pb[i] = ld(b, ya + i); }
  }
  for (int t = ya; t < yb; t += B) {
    float2 v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = MODE == 1 ? make_float2((float)t, (float)j) : add(pa[j], pb[j]);
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < B; ++j) { pa[j] = ld(a, t + B + j); pb[j] = ld(b, t + B + j); }
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      if (MODE == 0) acc = add(acc, v[j]);
      else if (t + j < yb) *(float2 *)(out + (size_t)(t + j) * n + xc) = v[j];
    }
  }
  if (MODE == 0) *(float2 *)(out + (size_t)ya * n + xc) = acc;
}
template <int B, int MODE>
void run_dir(const char *name, int n, int waves, float *a, float *b, float *out) {
  const int strips = n / 128;
  int chunks = waves / strips; if (chunks < 1) chunks = 1;
  const int chunk = (n + chunks - 1) / chunks;
  dim3 grid((strips + 3) / 4, (n + chunk - 1) / chunk);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) k_dir<B, MODE><<<grid, 256>>>(n, chunk, a, b, out);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) k_dir<B, MODE><<<grid, 256>>>(n, chunk, a, b, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double bytes = (double)n * n * 4 * (MODE == 0 ? 2 : MODE == 1 ? 1 : 3);
  printf("%-12s B=%d waves=%6d chunk=%4d : %.3f ms  %.0f GB/s\n", name, B, strips * (int)grid.y, chunk, ms, bytes / ms / 1e6);
}

__global__ void k_rowmajor(size_t n4, const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ out, int nin) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 v = a[i];
    if (nin > 1) v = add(v, b[i]);
    out[i] = v;
  }
}

template <int C, int NIN, int RING, int MODE, bool XCD>
void run(const char *name, int n, int H, int waves, float *a, float *b, float *out, float *dump) {
  const int VW = 64 * C - 2 * H;
  const int strips = (n + VW - 1) / VW;
  const int nbx = (strips + 3) / 4;
  int chunks = waves / strips; if (chunks < 1) chunks = 1;
  const int chunk = (n + chunks - 1) / chunks;
  dim3 grid(nbx, (n + chunk - 1) / chunk);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) k_strip<C, NIN, RING, MODE, XCD><<<grid, 256>>>(n, H, chunk, nbx, a, b, out, dump);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) k_strip<C, NIN, RING, MODE, XCD><<<grid, 256>>>(n, H, chunk, nbx, a, b, out, dump);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)n * n * 4 * (NIN + 1);
  printf("%-34s C=%d nin=%d ring=%d H=%2d waves=%6d chunk=%4d : %.3f ms  %.0f GB/s (ideal bytes)\n", name, C, NIN, RING, H,
         (int)(grid.x * grid.y * 4), chunk, ms, bytes / ms / 1e6);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16384;
  const size_t cells = (size_t)n * n;
  float *a, *b, *out, *dump;
  const size_t padded = (size_t)n * (n + 8192) * 4;
  CHECK(hipMalloc(&a, padded)); CHECK(hipMalloc(&b, padded)); CHECK(hipMalloc(&out, padded));
  CHECK(hipMalloc(&dump, (size_t)n * 4 * 2));
  CHECK(hipMemset(a, 0, cells * 4)); CHECK(hipMemset(b, 0, cells * 4));
  {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int nin = 1; nin <= 2; ++nin) {
      const size_t n4 = cells / 4;
      k_rowmajor<<<(n4 + 255) / 256, 256>>>(n4, (float4 *)a, (float4 *)b, (float4 *)out, nin);
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) k_rowmajor<<<(n4 + 255) / 256, 256>>>(n4, (float4 *)a, (float4 *)b, (float4 *)out, nin);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("row-major copy nin=%d : %.3f ms  %.0f GB/s\n", nin, ms, (double)cells * 4 * (nin + 1) / ms / 1e6);
    }
  }
  for (int waves : {4096, 8192, 16384}) {
    run_dir<1, 0>("loads-only", n, waves, a, b, out);
    run_dir<3, 0>("loads-only", n, waves, a, b, out);
    run_dir<8, 0>("loads-only", n, waves, a, b, out);
    run_dir<1, 1>("stores-only", n, waves, a, b, out);
    run_dir<3, 1>("stores-only", n, waves, a, b, out);
    run_dir<1, 2>("copy", n, waves, a, b, out);
    run_dir<3, 2>("copy-burst", n, waves, a, b, out);
    run_dir<8, 2>("copy-burst", n, waves, a, b, out);
  }
  if (argc > 2)
  for (int waves : {8192, 16384, 32768}) {
    run<2, 1, 3, 0, false>("strip copy", n, 0, waves, a, b, out, dump);
    run<2, 2, 3, 0, false>("strip 2-in", n, 0, waves, a, b, out, dump);
    run<2, 2, 3, 0, false>("strip 2-in halo dump", n, 6, waves, a, b, out, dump);
    run<2, 2, 3, 0, false>("strip 2-in halo dump", n, 14, waves, a, b, out, dump);
    run<2, 2, 3, 1, false>("strip 2-in halo masked", n, 14, waves, a, b, out, dump);
    run<2, 2, 3, 1, true>("strip 2-in halo masked xcd", n, 14, waves, a, b, out, dump);
    run<2, 2, 3, 0, true>("strip 2-in halo dump xcd", n, 14, waves, a, b, out, dump);
    run<2, 2, 6, 1, false>("strip 2-in halo masked ring6", n, 14, waves, a, b, out, dump);
    run<4, 2, 3, 1, false>("strip 2-in halo masked C4", n, 14, waves, a, b, out, dump);
    run<4, 2, 3, 1, true>("strip 2-in halo masked C4 xcd", n, 14, waves, a, b, out, dump);
    run<4, 2, 3, 0, false>("strip 2-in C4 nohalo", n, 0, waves, a, b, out, dump);
  }
  return 0;
}
