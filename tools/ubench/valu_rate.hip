// Micro-benchmark: VALU issue rate on gfx950 for scalar vs packed fp32 adds, as a function of
// waves per SIMD and of dependent vs independent chains.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c = out[0];
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 1) {  // 8 dependent v_add_f32
      asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                   : "+v"(a0) : "v"(c));
    } else if (MODE == 2) {  // 4 independent v_pk_add_f32 (8 adds)
      float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, cc = {c, c};
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (MODE == 3) {  // 8 independent v_add_f32 with dpp wave_shr
      asm volatile("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 4) {  // 8 independent v_add_f32 with row_shr dpp
      asm volatile("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 5) {  // 7 adds + s_nop 0
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n s_nop 0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x + 1] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name, float *d, int waves_per_simd) {
  const int iters = 20000, blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100);
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // instructions per SIMD = waves_per_simd * iters * 8 ; cycles at assumed 2.4 GHz
  double ns_per_instr = ms * 1e6 / ((double)waves_per_simd * iters * 8);
  printf("%-28s waves/SIMD %d : %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name,
         waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4);
}

int main() {
  float *d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_f32 independent", d, w);
    run<1>("v_add_f32 dependent", d, w);
    run<2>("v_pk_add_f32 (per pk instr)", d, w);
    run<3>("v_add_f32_dpp wave_shr", d, w);
    run<4>("v_add_f32_dpp row_shr", d, w);
    run<5>("7 v_add + s_nop 0", d, w);
  }
  return 0;
}
