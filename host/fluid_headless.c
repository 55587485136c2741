/*
 * fluid_headless.c — headless C host driver over the C-ABI (include/fluid.h).
 *
 * Replaces the reference's GLUT main loop (kernel.cu:887-946: display() = one
 * simulationStep() + one visualizationStep() per frame, :786-801) with a
 * scripted run: seeded splats instead of mouseMove (:870-878),
 * fluid_read instead of glDrawPixels (:753-755).  Pure C; links libfluid_hip.so.
 *
 *   fluid_headless [--n N] [--diffusion D] [--jacobi J] [--steps S] [--warmup W]
 *                  [--windows K] [--unfused] [--no-graph] [--dump PREFIX] [--splats]
 */
#include "../include/fluid.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* splitmix64; uniform float = (next() >> 40) * 2^-24 */
static uint64_t sm_state = 0x5EEDF1D5ull;
static uint64_t sm_next(void) {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float sm_uniform(void) { return (float)(sm_next() >> 40) * (1.0f / 16777216.0f); }

/* kernel.cu:833-850 */
static const float kPresets[4][4] = {
    {1.f, 1.f, 1.f, 1.f}, {1.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f}};

static int g_print_splats = 0;

static int emit_splat(fluid_ctx *ctx) {
  float sx = 0.25f + 0.5f * sm_uniform();
  float sy = 0.25f + 0.5f * sm_uniform();
  float fx = -50.0f + 100.0f * sm_uniform();
  float fy = -50.0f + 100.0f * sm_uniform();
  int preset = (int)(sm_next() % 4);
  if (g_print_splats)
    printf("splat %.9g %.9g %.9g %.9g %d\n", sx, sy, fx, fy, preset + 1);
  return ctx ? fluid_add_source(ctx, sx, sy, fx, fy, kPresets[preset]) : 0;
}

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_) {                                                                            \
      fprintf(stderr, "%s failed: %s (%s)\n", #call, fluid_strerror(rc_),                 \
              ctx ? fluid_last_error(ctx) : "");                                          \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

static int dump_field(fluid_ctx *ctx, const char *prefix, const char *name, int field, size_t elem,
                      int n) {
  size_t bytes = (size_t)n * n * elem;
  void *buf = malloc(bytes);
  if (!buf) return 1;
  int rc = fluid_read(ctx, field, buf, bytes);
  if (!rc) {
    char path[512];
    snprintf(path, sizeof path, "%s.%s.f32", prefix, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(buf, 1, bytes, f) != bytes) rc = 1;
    if (f) fclose(f);
  }
  free(buf);
  return rc;
}

static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

int main(int argc, char **argv) {
  fluid_params prm;
  fluid_default_params(&prm);
  int steps = 100, warmup = 10, windows = 5, splats_only = 0;
  const char *dump = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--n") && i + 1 < argc) prm.n = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--diffusion") && i + 1 < argc) prm.diffusion_iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--jacobi") && i + 1 < argc) prm.pressure_iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--windows") && i + 1 < argc) windows = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) prm.device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--unfused")) prm.flags |= FLUID_FLAG_UNFUSED;
    else if (!strcmp(argv[i], "--no-graph")) prm.flags |= FLUID_FLAG_NO_GRAPH;
    else if (!strcmp(argv[i], "--vorticity")) prm.flags |= FLUID_FLAG_VORTICITY;
    else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    else if (!strcmp(argv[i], "--splats")) splats_only = 1;
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  fluid_ctx *ctx = NULL;
  if (splats_only) { /* print the first `steps` splats of the seeded stream; no GPU needed */
    g_print_splats = 1;
    for (int i = 0; i < steps; ++i) emit_splat(NULL);
    return 0;
  }
  if (windows < 1) windows = 1;
  if (windows > 64) windows = 64;
  CHECK(fluid_init(&ctx, &prm));

  /* seeded splats: one before step 0 and before every 10th step */
  int done = 0;
  while (done < warmup) {
    CHECK(emit_splat(ctx));
    int chunk = warmup - done < 10 ? warmup - done : 10;
    CHECK(fluid_step(ctx, chunk));
    done += chunk;
  }
  CHECK(fluid_sync(ctx));

  /* timed windows: no fluid_read / fluid_add_source inside */
  float ms[64];
  for (int w = 0; w < windows; ++w) CHECK(fluid_time_steps(ctx, steps, &ms[w]));
  qsort(ms, windows, sizeof(float), cmp_float);
  double t_step = ms[windows / 2] / (steps > 0 ? steps : 1) * 1e-3;
  double cells = (double)prm.n * prm.n;
  double bpc = 88.0 + 16.0 * prm.diffusion_iters + 12.0 * prm.pressure_iters;
  fluid_diag dg; /* state after the timed steps: wavefront reductions on the device */
  CHECK(fluid_diagnostics(ctx, &dg));
  printf("{\"n\": %d, \"diffusion_iters\": %d, \"pressure_iters\": %d, \"steps\": %d, "
         "\"ms_per_step\": %.6f, \"gcups\": %.4f, \"alg_bytes_per_cell\": %.0f, "
         "\"effective_GBps\": %.1f, \"frac_of_8TBps\": %.4f, "
         "\"max_abs_u\": %.9g, \"max_abs_v\": %.9g, \"max_abs_div\": %.9g, \"max_abs_p\": %.9g, "
         "\"max_jacobi_update\": %.9g, \"max_abs_div_velocity\": %.9g, \"dye\": [%.9g, %.9g, %.9g, %.9g]}\n",
         prm.n, prm.diffusion_iters, prm.pressure_iters, steps, t_step * 1e3,
         cells / t_step / 1e9, bpc, bpc * cells / t_step / 1e9, bpc * cells / t_step / 8e12,
         dg.max_abs_u, dg.max_abs_v, dg.max_abs_div, dg.max_abs_p, dg.max_jacobi_update,
         dg.max_abs_div_velocity, dg.dye[0], dg.dye[1], dg.dye[2], dg.dye[3]);

  if (dump) {
    if (dump_field(ctx, dump, "velocity", FLUID_VELOCITY, 8, prm.n) ||
        dump_field(ctx, dump, "density", FLUID_DENSITY, 16, prm.n) ||
        dump_field(ctx, dump, "pressure", FLUID_PRESSURE, 4, prm.n)) {
      fprintf(stderr, "dump failed\n");
      return 1;
    }
  }
  fluid_destroy(ctx);
  return 0;
}
