/*
 * fluid_ref.c — CPU oracle: serial host-C restatement of the per-timestep
 * solver of /root/reference/FluidSimulation/kernel.cu.
 *
 * TEST INFRASTRUCTURE ONLY (see fluid_ref.h).  It implements include/fluid.h so
 * the same harness drives the oracle and the HIP library, plus the
 * kernel-level ref_* functions for known-answer tests.
 *
 * Parity status: "parity unpinned" — the reference ships no tests or golden
 * vectors and cannot be compiled here (CUDA + un-vendored CUDA-Samples headers
 * + texture-reference API).  Pinned only by the hand-derivable
 * KATs of SURVEY.md section 4 (tests/test_oracle_kat.py).
 *
 * Rules of the restatement:
 *  - double-buffer exactly like the reference (read array A, write array B),
 *    in place only where the reference RMWs through a surface;
 *  - fp32 only; build with -ffp-contract=off -fno-fast-math; keep the written
 *    left-to-right summation order (kernel.cu:195, :282, :313, :344);
 *  - texture fetches restated per SURVEY.md Appendix B: unnormalized
 *    coordinates, clamp-to-edge, POINT = texel [floor x, floor y], LINEAR =
 *    bilinear at (x-0.5, y-0.5) with exact fp32 weights, evaluated in the
 *    author's own mix() form  x + (y - x) * a  (kernel.cu:60-71), x first then
 *    y (kernel.cu:88-91);
 *  - filter modes as written on the line after each bind (hazard H1);
 *  - border writes are four independent ifs, last true one wins (H4).
 */
#include "fluid_ref.h"
#include "../include/fluid.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Row loops of the kernels: every output row depends only on the input buffers (read array A,
 * write array B; add_force touches each cell once), so the rows can be shared out over threads
 * without changing a bit.  Built serial (liboracle_fluid.so, the checker and the 1-core baseline)
 * and with -fopenmp (liboracle_fluid_omp.so, the multi-core CPU baseline of bench.py). */
#ifdef _OPENMP
#define OMP_ROWS _Pragma("omp parallel for schedule(static)")
#else
#define OMP_ROWS
#endif

typedef struct { float x, y; } f2;
typedef struct { float x, y, z, w; } f4;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* helper_math.h clamp(float, float, float) = fmaxf(a, fminf(f, b))  (Appendix C) */
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

/* kernel.cu:60-64  mix(float2, float2, float) = x + (y - x) * a */
static inline f2 mix2(f2 x, f2 y, float a) {
  f2 r;
  r.x = x.x + (y.x - x.x) * a;
  r.y = x.y + (y.y - x.y) * a;
  return r;
}
/* kernel.cu:66-71 */
static inline f4 mix4(f4 x, f4 y, float a) {
  f4 r;
  r.x = x.x + (y.x - x.x) * a;
  r.y = x.y + (y.y - x.y) * a;
  r.z = x.z + (y.z - x.z) * a;
  r.w = x.w + (y.w - x.w) * a;
  return r;
}
static inline f2 neg2(f2 v) { f2 r = { -v.x, -v.y }; return r; }

/* POINT fetch, clamp-to-edge (Appendix B.1). */
static inline f2 texel2(const f2 *t, int nx, int ny, int i, int j) {
  return t[clampi(i, 0, nx - 1) + (size_t)clampi(j, 0, ny - 1) * nx];
}
static inline f4 texel4(const f4 *t, int nx, int ny, int i, int j) {
  return t[clampi(i, 0, nx - 1) + (size_t)clampi(j, 0, ny - 1) * nx];
}
static inline float texel1(const float *t, int nx, int ny, int i, int j) {
  return t[clampi(i, 0, nx - 1) + (size_t)clampi(j, 0, ny - 1) * nx];
}

/* LINEAR fetch at texture coordinate (bx + 0.5, by + 0.5), i.e. bilinear with
 * base texel floor(bx), weight frac(bx) computed from bx directly (Appendix
 * B.2: do not round-trip through +0.5-0.5). */
static inline f2 bilinear2(const f2 *t, int nx, int ny, float bx, float by) {
  float fi = floorf(bx), fj = floorf(by);
  float a = bx - fi, b = by - fj;
  int i = (int)fi, j = (int)fj;
  f2 v0 = mix2(texel2(t, nx, ny, i, j), texel2(t, nx, ny, i + 1, j), a);
  f2 v1 = mix2(texel2(t, nx, ny, i, j + 1), texel2(t, nx, ny, i + 1, j + 1), a);
  return mix2(v0, v1, b);
}
static inline f4 bilinear4(const f4 *t, int nx, int ny, float bx, float by) {
  float fi = floorf(bx), fj = floorf(by);
  float a = bx - fi, b = by - fj;
  int i = (int)fi, j = (int)fj;
  f4 v0 = mix4(texel4(t, nx, ny, i, j), texel4(t, nx, ny, i + 1, j), a);
  f4 v1 = mix4(texel4(t, nx, ny, i, j + 1), texel4(t, nx, ny, i + 1, j + 1), a);
  return mix4(v0, v1, b);
}
/* LINEAR fetch at INTEGER texture coordinates (m, n) (Appendix B.3): base
 * texel (m-1, n-1), both weights exactly 0.5 => 2x2 mean in mix() form. */
static inline f2 linear_at_int2(const f2 *t, int nx, int ny, int m, int k) {
  f2 v0 = mix2(texel2(t, nx, ny, m - 1, k - 1), texel2(t, nx, ny, m, k - 1), 0.5f);
  f2 v1 = mix2(texel2(t, nx, ny, m - 1, k), texel2(t, nx, ny, m, k), 0.5f);
  return mix2(v0, v1, 0.5f);
}

static inline int interior(int x, int y, int nx, int ny) {
  return x > 0 && x < nx - 1 && y > 0 && y < ny - 1;
}

/* ------------------------------------------------------------------------- */
/* kernels                                                                    */
/* ------------------------------------------------------------------------- */

/* kernel.cu:114-140 */
void ref_advection_rect(int nx, int ny, float dt, const float *vel_in, float *vel_out) {
  const f2 *vin = (const f2 *)vel_in;
  f2 *vout = (f2 *)vel_out;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        f2 v = vin[id]; /* :125 LINEAR at texel centre = exact texel */
        float px = (float)x - dt * v.x; /* :126 */
        float py = (float)y - dt * v.y;
        px = clampf(px, 0.0f, (float)nx); /* :130 */
        py = clampf(py, 0.0f, (float)ny);
        vout[id] = bilinear2(vin, nx, ny, px, py); /* :131 */
      } else {
        f2 o = { 0.0f, 0.0f }; /* unreachable default: one of the ifs is always true */
        if (x == 0) o = neg2(linear_at_int2(vin, nx, ny, x + 1, y));         /* :135 */
        if (x == nx - 1) o = neg2(linear_at_int2(vin, nx, ny, x - 1, y));     /* :136 */
        if (y == 0) o = neg2(linear_at_int2(vin, nx, ny, x + 1, y + 1));     /* :137 */
        if (y == ny - 1) o = neg2(linear_at_int2(vin, nx, ny, x + 1, y - 1)); /* :138 */
        vout[id] = o;
      }
    }
}

/* kernel.cu:142-165 */
void ref_advection_density_rect(int nx, int ny, float dt, const float *vel, const float *dens_in,
                           float *dens_out) {
  const f2 *v2 = (const f2 *)vel;
  const f4 *din = (const f4 *)dens_in;
  f4 *dout = (f4 *)dens_out;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        f2 v = v2[id]; /* :154 POINT */
        float px = (float)x - dt * v.x; /* :156 */
        float py = (float)y - dt * v.y;
        px = clampf(px, 0.0f, (float)nx); /* :158 */
        py = clampf(py, 0.0f, (float)ny);
        dout[id] = bilinear4(din, nx, ny, px, py); /* :159 */
      } else {
        f4 z = { 0.0f, 0.0f, 0.0f, 0.0f }; /* :163 */
        dout[id] = z;
      }
    }
}

/* kernel.cu:173-205 */
void ref_diffusion_rect(int nx, int ny, float dt, float viscosity, const float *vel_in, float *vel_out) {
  const f2 *vin = (const f2 *)vel_in;
  f2 *vout = (f2 *)vel_out;
  float alpha = 1.0f / (viscosity * dt); /* :182 */
  float beta = 1.0f / (4.0f + alpha);    /* :183 */
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        f2 vL = vin[id - 1], vR = vin[id + 1], vB = vin[id - nx], vT = vin[id + nx];
        f2 v = vin[id];
        f2 o; /* :195  (vL + vR + vB + vT + alpha * velocity) * beta */
        o.x = ((((vL.x + vR.x) + vB.x) + vT.x) + alpha * v.x) * beta;
        o.y = ((((vL.y + vR.y) + vB.y) + vT.y) + alpha * v.y) * beta;
        vout[id] = o;
      } else {
        vout[id] = vin[id]; /* :201-203 */
      }
    }
}

/* kernel.cu:207-230 */
void ref_vorticity_rect(int nx, int ny, const float *vel, float *vort) {
  const f2 *v = (const f2 *)vel;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        f2 vL = v[id - 1], vR = v[id + 1], vB = v[id - nx], vT = v[id + nx];
        vort[id] = (vR.y - vL.y) - (vT.x - vB.x); /* :222 */
      } else {
        vort[id] = 0.0f; /* :228 */
      }
    }
}

/* kernel.cu:232-263 */
void ref_add_vorticity_rect(int nx, int ny, float dt, const float *vort, float *vel) {
  f2 *v = (f2 *)vel;
  const float scale = 0.2f; /* :239 */
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (!interior(x, y, nx, ny)) continue;
      float vL = vort[id - 1], vR = vort[id + 1], vB = vort[id - nx], vT = vort[id + nx];
      float gx = vR - vL, gy = vT - vB; /* :249 gradV = (gx, gy, 0, 0) */
      /* :252 dot(float4,float4) = x*x + y*y + z*z + w*w, left to right */
      float d = ((gx * gx + gy * gy) + 0.0f * 0.0f) + 0.0f * 0.0f;
      if (d) {
        /* :254 scale * cross((gx,gy,0),(0,0,1)):
         * cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x) */
        float cx = gy * 1.0f - 0.0f * 0.0f;
        float cy = 0.0f * 0.0f - gx * 1.0f;
        float fx = scale * cx, fy = scale * cy;
        v[id].x = v[id].x + fx * dt; /* :259 */
        v[id].y = v[id].y + fy * dt;
      }
    }
}

/* kernel.cu:265-289 */
void ref_divergence_rect(int nx, int ny, const float *vel, float *div) {
  const f2 *v = (const f2 *)vel;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        f2 vL = v[id - 1], vR = v[id + 1], vB = v[id - nx], vT = v[id + nx];
        div[id] = 0.5f * ((vR.x - vL.x) + (vT.y - vB.y)); /* :282 */
      } else {
        div[id] = 0.0f; /* :287 */
      }
    }
}

/* kernel.cu:291-324 */
void ref_pressure_jacobi_rect(int nx, int ny, const float *p_in, const float *div, float *p_out) {
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        float alpha = -1.0f, beta = 0.25f; /* :303-304 */
        float vL = p_in[id - 1], vR = p_in[id + 1], vB = p_in[id - nx], vT = p_in[id + nx];
        float d = div[id];
        p_out[id] = ((((vL + vR) + vB) + vT) + alpha * d) * beta; /* :313 */
      } else {
        float o = 0.0f;
        if (x == 0) o = -texel1(p_in, nx, ny, x + 1, y);     /* :319 */
        if (x == nx - 1) o = -texel1(p_in, nx, ny, x - 1, y); /* :320 */
        if (y == 0) o = -texel1(p_in, nx, ny, x, y + 1);     /* :321 */
        if (y == ny - 1) o = -texel1(p_in, nx, ny, x, y - 1); /* :322 */
        p_out[id] = o;
      }
    }
}

/* kernel.cu:326-355 */
void ref_projection_rect(int nx, int ny, const float *vel_in, const float *p, float *vel_out) {
  const f2 *vin = (const f2 *)vel_in;
  f2 *vout = (f2 *)vel_out;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      if (interior(x, y, nx, ny)) {
        float pL = p[id - 1], pR = p[id + 1], pB = p[id - nx], pT = p[id + nx];
        f2 v = vin[id];
        f2 o = { v.x - (pR - pL), v.y - (pT - pB) }; /* :344 */
        vout[id] = o;
      } else {
        f2 o = { 0.0f, 0.0f };
        if (x == 0) o = neg2(texel2(vin, nx, ny, x + 1, y));         /* :350 */
        if (x == nx - 1) o = neg2(texel2(vin, nx, ny, x - 1, y));     /* :351 */
        if (y == 0) o = neg2(texel2(vin, nx, ny, x + 1, y + 1));     /* :352 */
        if (y == ny - 1) o = neg2(texel2(vin, nx, ny, x + 1, y - 1)); /* :353 */
        vout[id] = o;
      }
    }
}

/* kernel.cu:357-383 */
void ref_add_force_rect(int nx, int ny, float dt, float radius, float sx, float sy, float fx, float fy,
                   const float rgba[4], float *vel, float *dens) {
  f2 *v = (f2 *)vel;
  f4 *d = (f4 *)dens;
  OMP_ROWS
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      size_t id = x + (size_t)y * nx;
      float dx = ((float)x / (float)nx) - sx; /* :365 */
      float dy = ((float)y / (float)ny) - sy; /* :366 */
      float c = expf(-(dx * dx + dy * dy) / radius) * dt; /* :370 (float overload of exp) */
      v[id].x = v[id].x + c * fx; /* :378 */
      v[id].y = v[id].y + c * fy;
      d[id].x = d[id].x + c * rgba[0]; /* :382 */
      d[id].y = d[id].y + c * rgba[1];
      d[id].z = d[id].z + c * rgba[2];
      d[id].w = d[id].w + c * rgba[3];
    }
}

/* The reference's kernels take ONE gridResolution for both axes (kernel.cu:115, :143, :174, :266,
 * :292, :327, :358, :438): these are the restatements proper.  The *_rect forms above are the
 * same rules with the extent taken per axis (x tests and clamps against nx, y tests and clamps
 * against ny, row pitch nx, add_force normalises x by nx and y by ny) - the generalisation
 * SURVEY.md 8f-4 asks for, with no behaviour of the reference to be identical to when nx != ny. */
void ref_advection(int n, float dt, const float *vel_in, float *vel_out) { ref_advection_rect(n, n, dt, vel_in, vel_out); }
void ref_advection_density(int n, float dt, const float *vel, const float *dens_in, float *dens_out) {
  ref_advection_density_rect(n, n, dt, vel, dens_in, dens_out);
}
void ref_diffusion(int n, float dt, float viscosity, const float *vel_in, float *vel_out) {
  ref_diffusion_rect(n, n, dt, viscosity, vel_in, vel_out);
}
void ref_vorticity(int n, const float *vel, float *vort) { ref_vorticity_rect(n, n, vel, vort); }
void ref_add_vorticity(int n, float dt, const float *vort, float *vel) { ref_add_vorticity_rect(n, n, dt, vort, vel); }
void ref_divergence(int n, const float *vel, float *div) { ref_divergence_rect(n, n, vel, div); }
void ref_pressure_jacobi(int n, const float *p_in, const float *div, float *p_out) {
  ref_pressure_jacobi_rect(n, n, p_in, div, p_out);
}
void ref_projection(int n, const float *vel_in, const float *p, float *vel_out) { ref_projection_rect(n, n, vel_in, p, vel_out); }
void ref_add_force(int n, float dt, float radius, float sx, float sy, float fx, float fy, const float rgba[4],
                   float *vel, float *dens) {
  ref_add_force_rect(n, n, dt, radius, sx, sy, fx, fy, rgba, vel, dens);
}

/* ------------------------------------------------------------------------- */
/* host driver: the reference's L2 layer (kernel.cu:435-708) behind fluid.h   */
/* ------------------------------------------------------------------------- */

struct fluid_ctx {
  fluid_params prm;
  int nx, ny; /* grid extent: prm.n columns, prm.ny rows (0 = square, the reference's case) */
  size_t cells;
  float *vel[2];  /* velocityBufferArray[2]  kernel.cu:443-444 */
  float *dens[2]; /* densityBufferArray[2]   :447-448 */
  float *prs[2];  /* pressureBufferArray[2]  :452-453 */
  float *div;     /* :455-456 */
  float *vort;    /* :458-459 */
  int vin, din, pin; /* inputVelocityBuffer, inputDensityBuffer, inputPressureBuffer :442,:446,:451 */
};

int fluid_default_params(fluid_params *p) {
  if (!p) return FLUID_E_INVALID;
  memset(p, 0, sizeof *p);
  p->n = 512;              /* kernel.cu:438 */
  p->diffusion_iters = 10; /* :584 */
  p->pressure_iters = 10;  /* :620 */
  p->dt = 0.1f;            /* :31 */
  p->viscosity = 0.01f;    /* :181 */
  p->splat_radius = 0.001f;/* :368 */
  return FLUID_OK;
}

int fluid_init(fluid_ctx **out, const fluid_params *p) {
  if (!out) return FLUID_E_INVALID;
  fluid_params prm;
  if (p) prm = *p; else fluid_default_params(&prm);
  if (prm.n < 4 || prm.diffusion_iters < 0 || prm.pressure_iters < 0) return FLUID_E_INVALID;
  if (prm.ny != 0 && prm.ny < 4) return FLUID_E_INVALID;
  if (prm.slab_count > 1) return FLUID_E_UNSUPPORTED; /* the oracle is whole-domain only */
  fluid_ctx *c = (fluid_ctx *)calloc(1, sizeof *c);
  if (!c) return FLUID_E_NOMEM;
  c->prm = prm;
  c->nx = prm.n;
  c->ny = prm.ny ? prm.ny : prm.n;
  c->cells = (size_t)c->nx * c->ny;
  int ok = 1;
  for (int i = 0; i < 2; ++i) {
    ok &= (c->vel[i] = (float *)calloc(c->cells, 2 * sizeof(float))) != NULL;
    ok &= (c->dens[i] = (float *)calloc(c->cells, 4 * sizeof(float))) != NULL;
    ok &= (c->prs[i] = (float *)calloc(c->cells, sizeof(float))) != NULL;
  }
  ok &= (c->div = (float *)calloc(c->cells, sizeof(float))) != NULL;
  ok &= (c->vort = (float *)calloc(c->cells, sizeof(float))) != NULL;
  if (!ok) { fluid_destroy(c); return FLUID_E_NOMEM; }
  *out = c;
  return fluid_reset(c);
}

void fluid_destroy(fluid_ctx *c) {
  if (!c) return;
  for (int i = 0; i < 2; ++i) { free(c->vel[i]); free(c->dens[i]); free(c->prs[i]); }
  free(c->div); free(c->vort); free(c);
}

/* resetSimulationCUDA, kernel.cu:34-49, on three chosen buffers */
static void reset3(fluid_ctx *c, float *v, float *p, float *d) {
  memset(v, 0, c->cells * 2 * sizeof(float));
  memset(p, 0, c->cells * sizeof(float));
  memset(d, 0, c->cells * 4 * sizeof(float));
}

/* kernel.cu:518-528 */
int fluid_reset(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  reset3(c, c->vel[c->vin], c->prs[c->pin], c->dens[c->din]);
  return FLUID_OK;
}

/* kernel.cu:530-540 */
static void reset_pressure(fluid_ctx *c) {
  reset3(c, c->vel[(c->vin + 1) % 2], c->prs[c->pin], c->dens[(c->din + 1) % 2]);
}

/* kernel.cu:542-555 */
int fluid_advect_velocity(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  int nxt = (c->vin + 1) % 2;
  ref_advection_rect(c->nx, c->ny, c->prm.dt, c->vel[c->vin], c->vel[nxt]);
  c->vin = nxt;
  return FLUID_OK;
}

/* kernel.cu:582-602 */
int fluid_diffuse(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  for (int i = 0; i < c->prm.diffusion_iters; ++i) {
    int nxt = (c->vin + 1) % 2;
    ref_diffusion_rect(c->nx, c->ny, c->prm.dt, c->prm.viscosity, c->vel[c->vin], c->vel[nxt]);
    c->vin = nxt;
  }
  return FLUID_OK;
}

/* kernel.cu:557-580 */
int fluid_vorticity(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  ref_vorticity_rect(c->nx, c->ny, c->vel[c->vin], c->vort);
  ref_add_vorticity_rect(c->nx, c->ny, c->prm.dt, c->vort, c->vel[c->vin]);
  return FLUID_OK;
}

/* kernel.cu:604-662 */
int fluid_project(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  ref_divergence_rect(c->nx, c->ny, c->vel[c->vin], c->div); /* :607-616 */
  reset_pressure(c);                                /* :618 */
  for (int i = 0; i < c->prm.pressure_iters; ++i) { /* :620-641 */
    int nxt = (c->pin + 1) % 2;
    ref_pressure_jacobi_rect(c->nx, c->ny, c->prs[c->pin], c->div, c->prs[nxt]);
    c->pin = nxt;
  }
  int nxt = (c->vin + 1) % 2; /* :643-661 */
  ref_projection_rect(c->nx, c->ny, c->vel[c->vin], c->prs[c->pin], c->vel[nxt]);
  c->vin = nxt;
  return FLUID_OK;
}

/* kernel.cu:664-685 */
int fluid_advect_density(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  int nxt = (c->din + 1) % 2;
  ref_advection_density_rect(c->nx, c->ny, c->prm.dt, c->vel[c->vin], c->dens[c->din], c->dens[nxt]);
  c->din = nxt;
  return FLUID_OK;
}

/* kernel.cu:701-708 */
int fluid_step(fluid_ctx *c, int nsteps) {
  if (!c || nsteps < 0) return FLUID_E_INVALID;
  for (int s = 0; s < nsteps; ++s) {
    fluid_advect_velocity(c);
    fluid_diffuse(c);
    if (c->prm.flags & FLUID_FLAG_VORTICITY) fluid_vorticity(c); /* :705 (commented out) */
    fluid_project(c);
    fluid_advect_density(c);
  }
  return FLUID_OK;
}

/* kernel.cu:687-699 */
int fluid_prepare_steps(fluid_ctx *c, int nsteps) { return c && nsteps >= 0 ? FLUID_OK : FLUID_E_INVALID; }

int fluid_add_source(fluid_ctx *c, float sx, float sy, float fx, float fy, const float rgba[4]) {
  if (!c || !rgba) return FLUID_E_INVALID;
  ref_add_force_rect(c->nx, c->ny, c->prm.dt, c->prm.splat_radius, sx, sy, fx, fy, rgba, c->vel[c->vin],
                c->dens[c->din]);
  return FLUID_OK;
}

static float *field_ptr(fluid_ctx *c, int field, size_t *elem) {
  switch (field) {
    case FLUID_VELOCITY: *elem = 2 * sizeof(float); return c->vel[c->vin];
    case FLUID_DENSITY: *elem = 4 * sizeof(float); return c->dens[c->din];
    case FLUID_PRESSURE: *elem = sizeof(float); return c->prs[c->pin];
    case FLUID_DIVERGENCE: *elem = sizeof(float); return c->div;
    case FLUID_VORTICITY: *elem = sizeof(float); return c->vort;
  }
  return NULL;
}

int fluid_read(fluid_ctx *c, int field, void *dst, size_t dst_bytes) {
  size_t elem;
  if (!c || !dst) return FLUID_E_INVALID;
  float *p = field_ptr(c, field, &elem);
  if (!p) return FLUID_E_INVALID;
  if (dst_bytes < c->cells * elem) return FLUID_E_SIZE;
  memcpy(dst, p, c->cells * elem);
  return FLUID_OK;
}

int fluid_write(fluid_ctx *c, int field, const void *src, size_t src_bytes) {
  size_t elem;
  if (!c || !src) return FLUID_E_INVALID;
  float *p = field_ptr(c, field, &elem);
  if (!p) return FLUID_E_INVALID;
  if (src_bytes != c->cells * elem) return FLUID_E_SIZE;
  memcpy(p, src, c->cells * elem);
  return FLUID_OK;
}

int fluid_sync(fluid_ctx *c) { return c ? FLUID_OK : FLUID_E_INVALID; }

/* fluid_diagnostics: no counterpart in the reference; a serial statement of the definitions in
 * include/fluid.h, with the formulas of kernel.cu:313 (one more pressureJacobi sweep) and :282
 * (divergence) in their written order.  Checker for the HIP library's wavefront reductions. */
static float max_nan(float a, float b) { return (b > a || b != b) ? b : a; }
int fluid_diagnostics(fluid_ctx *c, fluid_diag *out) {
  if (!c || !out) return FLUID_E_INVALID;
  const int n = c->nx, ny = c->ny;
  const float *v = c->vel[c->vin], *d = c->dens[c->din], *p = c->prs[c->pin], *dv = c->div;
  float mx[6] = {0, 0, 0, 0, 0, 0};
  double sm[4] = {0, 0, 0, 0};
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < n; ++x) {
      const size_t id = (size_t)y * n + x;
      mx[0] = max_nan(mx[0], fabsf(v[2 * id]));
      mx[1] = max_nan(mx[1], fabsf(v[2 * id + 1]));
      mx[2] = max_nan(mx[2], fabsf(dv[id]));
      mx[3] = max_nan(mx[3], fabsf(p[id]));
      if (x >= 1 && x <= n - 2 && y >= 1 && y <= ny - 2) {
        const float vL = p[id - 1], vR = p[id + 1], vB = p[id - n], vT = p[id + n];
        const float next = ((((vL + vR) + vB) + vT) + (-1.0f) * dv[id]) * 0.25f; /* kernel.cu:313 */
        mx[4] = max_nan(mx[4], fabsf(next - p[id]));
        const float uL = v[2 * (id - 1)], uR = v[2 * (id + 1)];
        const float wB = v[2 * (id - n) + 1], wT = v[2 * (id + n) + 1];
        mx[5] = max_nan(mx[5], fabsf(0.5f * ((uR - uL) + (wT - wB)))); /* kernel.cu:282 */
      }
      for (int k = 0; k < 4; ++k) sm[k] += (double)d[4 * id + k];
    }
  out->max_abs_u = mx[0];
  out->max_abs_v = mx[1];
  out->max_abs_div = mx[2];
  out->max_abs_p = mx[3];
  out->max_jacobi_update = mx[4];
  out->max_abs_div_velocity = mx[5];
  for (int k = 0; k < 4; ++k) out->dye[k] = sm[k];
  return FLUID_OK;
}

int fluid_geometry(const fluid_ctx *c, int *n, int *row0, int *rows) {
  if (!c) return FLUID_E_INVALID;
  if (n) *n = c->nx;
  if (row0) *row0 = 0;
  if (rows) *rows = c->ny;
  return FLUID_OK;
}

const char *fluid_strerror(int code) {
  switch (code) {
    case FLUID_OK: return "ok";
    case FLUID_E_INVALID: return "invalid argument";
    case FLUID_E_NOMEM: return "out of memory";
    case FLUID_E_DEVICE: return "device error";
    case FLUID_E_SIZE: return "buffer size mismatch";
    case FLUID_E_COMM: return "halo transport error";
    case FLUID_E_HALO: return "advection left the slab halo";
    case FLUID_E_UNSUPPORTED: return "unsupported";
  }
  return "unknown error";
}
const char *fluid_last_error(const fluid_ctx *c) { (void)c; return ""; }
