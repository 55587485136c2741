/*
 * fluid_headless.c — headless C host driver over the C-ABI (include/fluid.h).
 *
 * Replaces the reference's GLUT main loop (kernel.cu:887-946: display() = one
 * simulationStep() + one visualizationStep() per frame, :786-801) with a
 * scripted run: seeded splats or a recorded input session
 * instead of mouseMove (:870-878), fluid_read and an image file instead of
 * glDrawPixels (:753-755).  Pure C; links libfluid_hip.so.
 *
 *   fluid_headless [--n N] [--ny ROWS] [--diffusion D] [--jacobi J] [--steps S] [--warmup W]
 *                  [--windows K] [--unfused] [--no-graph] [--vorticity] [--device I]
 *                  [--dump PREFIX] [--splats] [--inner-slabs G] [--devices I,J,...]
 *       timed run on the seeded splat script; prints one JSON line (--inner-slabs: the domain as G
 *       slabs side by side on the one GPU, fluid_params.inner_slabs; --devices: ONE process, the
 *       domain as one row slab per listed GPU, fluid_params.n_devices / device_ids — the shape of
 *       the reference's main(), kernel.cu:887-946, over several GPUs)
 *
 *   ... --replay FILE [--frames F] [--window W H]
 *       replays a recorded session instead (kernel.cu:858-878, :813-856): events of frame f are
 *       applied before that frame's simulationStep, converted exactly like mouseClick / mouseMove /
 *       keyUp + addForce.  FILE is either lines "FRAME down X Y" / "FRAME move X Y" / "FRAME key C"
 *       ('#' starts a comment) or the JSON list fluid-simulation_amd/replay.py reads
 *       ([{"frame": 0, "type": "move", "x": 110, "y": 195}, {"frame": 2, "type": "key", "key": "3"}]).
 *
 *   ... --view d|v|p [--ppm FILE] [--pfm FILE]
 *       the frame visualizationStep() would draw (kernel.cu:389-431, :710-756; keys 'd' 'v' 'p',
 *       :823-831) from the final state: W x H pixels (the window, default n x n), pixel (x, y) =
 *       the mapping of texel (x, y); PPM is 8-bit, top row first as on
 *       screen, PFM is the float image bottom row first as glDrawPixels receives it.  A 'd' / 'v' /
 *       'p' key event in a replayed session selects the view too.  (The reference launches
 *       n/32 x n/32 blocks whatever the window, :439-440: pixels beyond the grid are never written
 *       there; here they are zero.)
 *
 *   ... --slab-rank R --slab-count K --comm-file PATH [--halo-rows A] [--pressure-groups G]
 *       one process per GPU (device R unless --device): this process owns slab R of K and
 *       exchanges halo rows with its neighbours over RCCL from inside fluid_step
 *       (fluid_comm_init_rccl).  Rank 0 writes the communicator id to PATH, the others wait for it.
 *       --dump writes the owned rows to PREFIX.rankR.<field>.f32.
 */
#define _POSIX_C_SOURCE 200809L
#include "../include/fluid.h"

#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

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

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int dump_field(fluid_ctx *ctx, const char *prefix, const char *name, int field, size_t elem,
                      int n, int rows) {
  size_t bytes = (size_t)n * rows * elem;
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

/* ---- recorded sessions (kernel.cu:813-878) ------------------------------------------------- */
enum { EV_DOWN, EV_MOVE, EV_KEY };
typedef struct { int frame, type, x, y; char key; } event_t;

/* value of "name": in the JSON object [obj, end); NULL if absent */
static const char *json_value(const char *obj, const char *end, const char *name) {
  char pat[32];
  snprintf(pat, sizeof pat, "\"%s\"", name);
  const char *p = obj;
  while ((p = strstr(p, pat)) && p < end) {
    p += strlen(pat);
    while (p < end && isspace((unsigned char)*p)) ++p;
    if (p < end && *p == ':') {
      ++p;
      while (p < end && isspace((unsigned char)*p)) ++p;
      return p;
    }
  }
  return NULL;
}

static int parse_events(const char *path, event_t **out, int *count) {
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *text = (char *)malloc((size_t)len + 1);
  if (!text || fread(text, 1, (size_t)len, f) != (size_t)len) { fclose(f); free(text); return 1; }
  fclose(f);
  text[len] = 0;
  int cap = 64, n = 0;
  event_t *ev = (event_t *)malloc(sizeof(event_t) * cap);
  const char *p = text;
  while (*p && isspace((unsigned char)*p)) ++p;
  int bad = 0;
  if (*p == '[') { /* the JSON list replay.py reads */
    while ((p = strchr(p, '{'))) {
      const char *end = strchr(p, '}');
      if (!end) { bad = 1; break; }
      event_t e = {0, 0, 0, 0, 0};
      const char *v;
      if (!(v = json_value(p, end, "frame"))) { bad = 1; break; }
      e.frame = atoi(v);
      if (!(v = json_value(p, end, "type")) || *v != '"') { bad = 1; break; }
      if (!strncmp(v + 1, "down", 4)) e.type = EV_DOWN;
      else if (!strncmp(v + 1, "move", 4)) e.type = EV_MOVE;
      else if (!strncmp(v + 1, "key", 3)) e.type = EV_KEY;
      else { bad = 1; break; }
      if (e.type == EV_KEY) {
        if (!(v = json_value(p, end, "key"))) { bad = 1; break; }
        e.key = *v == '"' ? v[1] : *v;   /* "3" or 3 */
      } else {
        const char *vx = json_value(p, end, "x"), *vy = json_value(p, end, "y");
        if (!vx || !vy) { bad = 1; break; }
        e.x = atoi(vx);
        e.y = atoi(vy);
      }
      if (n == cap) ev = (event_t *)realloc(ev, sizeof(event_t) * (cap *= 2));
      ev[n++] = e;
      p = end + 1;
    }
  } else { /* "FRAME down X Y" / "FRAME move X Y" / "FRAME key C" */
    char *save = NULL;
    for (char *line = strtok_r(text, "\n", &save); line && !bad; line = strtok_r(NULL, "\n", &save)) {
      char *hash = strchr(line, '#');
      if (hash) *hash = 0;
      event_t e = {0, 0, 0, 0, 0};
      char type[16] = {0}, key[8] = {0};
      int got = sscanf(line, "%d %15s", &e.frame, type);
      if (got <= 0) continue; /* blank */
      if (got < 2) { bad = 1; break; }
      if (!strcmp(type, "down") || !strcmp(type, "move")) {
        e.type = type[0] == 'd' ? EV_DOWN : EV_MOVE;
        if (sscanf(line, "%*d %*s %d %d", &e.x, &e.y) != 2) bad = 1;
      } else if (!strcmp(type, "key")) {
        e.type = EV_KEY;
        if (sscanf(line, "%*d %*s %7s", key) != 1) bad = 1;
        e.key = key[0];
      } else bad = 1;
      if (bad) break;
      if (n == cap) ev = (event_t *)realloc(ev, sizeof(event_t) * (cap *= 2));
      ev[n++] = e;
    }
  }
  free(text);
  if (bad) { free(ev); return 1; }
  /* stable by frame (events of one frame keep their file order): insertion sort, lists are short */
  for (int i = 1; i < n; ++i) {
    event_t e = ev[i];
    int j = i - 1;
    while (j >= 0 && ev[j].frame > e.frame) { ev[j + 1] = ev[j]; --j; }
    ev[j + 1] = e;
  }
  *out = ev;
  *count = n;
  return 0;
}

/* display() for `frames` frames (kernel.cu:786-801): the input events of a frame, then one
 * simulationStep().  mouseClick :860-868, mouseMove :870-878, keyUp :813-856. */
static int replay(fluid_ctx *ctx, const event_t *ev, int nev, int frames, int width, int height, int *view) {
  float color[4] = {1.f, 1.f, 1.f, 1.f}; /* densityColor, kernel.cu:507 */
  int mX = 0, mY = 0;                      /* kernel.cu:858: zero-initialised globals */
  int i = 0;
  for (int f = 0; f < frames; ++f) {
    for (; i < nev && ev[i].frame <= f; ++i) {
      const event_t *e = &ev[i];
      if (e->type == EV_DOWN) {
        mX = e->x;
        mY = e->y;
      } else if (e->type == EV_MOVE) {
        const float fx = (float)(e->x - mX), fy = -(float)(e->y - mY); /* :872-873 */
        const int px = height / 2, py = width / 2;                     /* :875: always the centre (x and y swapped) */
        CHECK(fluid_add_source(ctx, (float)px / width, (float)py / height, fx, fy, color)); /* :689-690 */
        mX = e->x;
        mY = e->y;
      } else {
        switch (e->key) {
          case 'r': CHECK(fluid_reset(ctx)); break;                    /* :819-821 */
          case 'd': *view = 0; break;                                  /* :823-831 */
          case 'v': *view = 1; break;
          case 'p': *view = 2; break;
          case '1': case '2': case '3': case '4':                      /* :833-850 */
            memcpy(color, kPresets[e->key - '1'], sizeof color);
            break;
          default: break;                                              /* other keys do nothing */
        }
      }
    }
    CHECK(fluid_step(ctx, 1));
  }
  return 0;
}

/* ---- visualizationStep() without the window (kernel.cu:389-431, :710-756) -------------------- */
static float clamp01(float v) { return v < 0.f ? 0.f : v > 1.f ? 1.f : v; }

static int write_view(fluid_ctx *ctx, int n, int rows, int view, int width, int height, const char *ppm, const char *pfm) {
  static const int field_of[3] = {FLUID_DENSITY, FLUID_VELOCITY, FLUID_PRESSURE};
  static const int comps_of[3] = {4, 2, 1};
  const int comps = comps_of[view];
  float *fld = (float *)malloc(sizeof(float) * comps * (size_t)n * rows);
  float *img = (float *)calloc(4 * (size_t)width * height, sizeof(float));
  if (!fld || !img) { free(fld); free(img); return 1; }
  {
    const int rrc = fluid_read(ctx, field_of[view], fld, sizeof(float) * comps * (size_t)n * rows);
    if (rrc) {
      fprintf(stderr, "fluid_read failed: %s (%s)\n", fluid_strerror(rrc), fluid_last_error(ctx));
      free(fld);
      free(img);
      return 1;
    }
  }
  for (int y = 0; y < height && y < rows; ++y)    /* threads exist for id < n only, kernel.cu:439-440 */
    for (int x = 0; x < width && x < n; ++x) {
      const size_t t = (size_t)y * n + x;         /* tex2D(tex, id.x, id.y), point filter: texel (x, y) */
      float *o = img + 4 * ((size_t)y * width + x); /* visualizationBuffer[id.x + id.y * width] */
      if (view == 0) {                               /* :397-398 passthrough */
        memcpy(o, fld + 4 * t, 4 * sizeof(float));
      } else if (view == 1) {                        /* :411-414 */
        o[0] = (1.0f + fld[2 * t]) / 2.0f;
        o[1] = (1.0f + fld[2 * t + 1]) / 2.0f;
        o[2] = 0.0f;
        o[3] = 0.0f;
      } else {                                       /* :427-429 */
        const float v = (1.0f + fld[t]) / 2.0f;
        o[0] = o[1] = o[2] = o[3] = v;
      }
    }
  int rc = 0;
  if (ppm) { /* what the window shows: glDrawPixels puts row 0 at the bottom, GL clamps to [0, 1] */
    FILE *f = fopen(ppm, "wb");
    if (!f) rc = 1;
    else {
      fprintf(f, "P6\n%d %d\n255\n", width, height);
      for (int y = height - 1; y >= 0 && !rc; --y)
        for (int x = 0; x < width; ++x) {
          const float *o = img + 4 * ((size_t)y * width + x);
          unsigned char px[3];
          for (int c = 0; c < 3; ++c) px[c] = (unsigned char)(clamp01(o[c]) * 255.0f + 0.5f);
          if (fwrite(px, 1, 3, f) != 3) { rc = 1; break; }
        }
      fclose(f);
    }
  }
  if (pfm && !rc) { /* the float image itself, bottom row first (little endian) */
    FILE *f = fopen(pfm, "wb");
    if (!f) rc = 1;
    else {
      fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
      for (size_t i = 0; i < (size_t)width * height && !rc; ++i)
        if (fwrite(img + 4 * i, sizeof(float), 3, f) != 3) rc = 1;
      fclose(f);
    }
  }
  free(fld);
  free(img);
  return rc;
}

/* ---- communicator id through a file (one process per GPU, no launcher needed) --------------- */
static int share_comm_id(const char *path, int rank, unsigned char id[FLUID_COMM_ID_BYTES]) {
  if (rank == 0) {
    if (fluid_comm_unique_id(id, FLUID_COMM_ID_BYTES)) return 1;
    char tmp[600];
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, FLUID_COMM_ID_BYTES, f) != FLUID_COMM_ID_BYTES) { if (f) fclose(f); return 1; }
    fclose(f);
    return rename(tmp, path) != 0; /* appears whole or not at all */
  }
  for (int tries = 0; tries < 1200; ++tries) { /* up to two minutes */
    FILE *f = fopen(path, "rb");
    if (f) {
      size_t got = fread(id, 1, FLUID_COMM_ID_BYTES, f);
      fclose(f);
      if (got == FLUID_COMM_ID_BYTES) return 0;
    }
    { struct timespec ts = {0, 100000000}; nanosleep(&ts, NULL); }
  }
  return 1;
}

int main(int argc, char **argv) {
  fluid_params prm;
  fluid_default_params(&prm);
  int steps = 100, warmup = 10, windows = 5, splats_only = 0, device = -1;
  int frames = -1, win_w = 0, win_h = 0, view = -1;
  const char *dump = NULL, *replay_path = NULL, *ppm = NULL, *pfm = NULL, *comm_file = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--n") && i + 1 < argc) prm.n = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ny") && i + 1 < argc) prm.ny = atoi(argv[++i]); /* rows; default = n (square) */
    else if (!strcmp(argv[i], "--diffusion") && i + 1 < argc) prm.diffusion_iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--jacobi") && i + 1 < argc) prm.pressure_iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--windows") && i + 1 < argc) windows = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--unfused")) prm.flags |= FLUID_FLAG_UNFUSED;
    else if (!strcmp(argv[i], "--no-graph")) prm.flags |= FLUID_FLAG_NO_GRAPH;
    else if (!strcmp(argv[i], "--vorticity")) prm.flags |= FLUID_FLAG_VORTICITY;
    else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    else if (!strcmp(argv[i], "--splats")) splats_only = 1;
    else if (!strcmp(argv[i], "--replay") && i + 1 < argc) replay_path = argv[++i];
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--window") && i + 2 < argc) { win_w = atoi(argv[++i]); win_h = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--view") && i + 1 < argc) {
      const char *v = argv[++i];
      view = v[0] == 'd' ? 0 : v[0] == 'v' ? 1 : v[0] == 'p' ? 2 : -2;
      if (view == -2) { fprintf(stderr, "--view takes d, v or p\n"); return 2; }
    }
    else if (!strcmp(argv[i], "--ppm") && i + 1 < argc) ppm = argv[++i];
    else if (!strcmp(argv[i], "--pfm") && i + 1 < argc) pfm = argv[++i];
    else if (!strcmp(argv[i], "--slab-rank") && i + 1 < argc) prm.slab_rank = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--slab-count") && i + 1 < argc) prm.slab_count = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--halo-rows") && i + 1 < argc) prm.halo_rows = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pressure-groups") && i + 1 < argc) prm.pressure_groups = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--comm-file") && i + 1 < argc) comm_file = argv[++i];
    else if (!strcmp(argv[i], "--inner-slabs") && i + 1 < argc) prm.inner_slabs = atoi(argv[++i]); /* fluid.h: G slabs side by side on this GPU */
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) { /* fluid.h: several GPUs under one context */
      const char *q = argv[++i];
      prm.n_devices = 0;
      while (*q && prm.n_devices < 8) {
        prm.device_ids[prm.n_devices++] = (int)strtol(q, (char **)&q, 10);
        if (*q == ',') ++q;
      }
      if (*q || prm.n_devices < 1) { fprintf(stderr, "--devices takes 1 to 8 comma-separated ordinals\n"); return 2; }
    }
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  fluid_ctx *ctx = NULL;
  if (splats_only) { /* print the first `steps` splats of the seeded stream; no GPU needed */
    g_print_splats = 1;
    for (int i = 0; i < steps; ++i) emit_splat(NULL);
    return 0;
  }
  const int slabs = prm.slab_count > 1;
  if (slabs && !comm_file) { fprintf(stderr, "--slab-count needs --comm-file PATH (shared by all ranks)\n"); return 2; }
  if (slabs && (replay_path || ppm || pfm)) { fprintf(stderr, "--replay / --ppm / --pfm are whole-domain options\n"); return 2; }
  prm.device = device >= 0 ? device : (slabs ? prm.slab_rank : 0);
  if (windows < 1) windows = 1;
  if (windows > 64) windows = 64;
  if (win_w <= 0 || win_h <= 0) { win_w = prm.n; win_h = prm.ny ? prm.ny : prm.n; } /* the reference's window is the grid, kernel.cu:438, :466-467 */
  CHECK(fluid_init(&ctx, &prm));
  int n = 0, row0 = 0, rows = 0;
  CHECK(fluid_geometry(ctx, &n, &row0, &rows));
  if (slabs) {
    unsigned char id[FLUID_COMM_ID_BYTES];
    if (share_comm_id(comm_file, prm.slab_rank, id)) { fprintf(stderr, "could not share the communicator id through %s\n", comm_file); return 1; }
    CHECK(fluid_comm_init_rccl(ctx, id, sizeof id));
  }

  if (replay_path) { /* a recorded session: no timing, the result is the state (and the frame) */
    event_t *ev = NULL;
    int nev = 0;
    if (parse_events(replay_path, &ev, &nev)) { fprintf(stderr, "cannot read events from %s\n", replay_path); return 1; }
    if (frames < 0) frames = nev ? ev[nev - 1].frame + 1 : 0;
    int sel = view < 0 ? 0 : view;
    if (replay(ctx, ev, nev, frames, win_w, win_h, &sel)) return 1;
    free(ev);
    if (view < 0) view = sel; /* --view wins over the session's d / v / p keys */
    CHECK(fluid_sync(ctx));
    printf("{\"n\": %d, \"frames\": %d, \"events\": %d, \"view\": \"%c\"}\n", n, frames, nev, "dvp"[view]);
  } else {
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
    for (int w = 0; w < windows; ++w) {
      if (slabs) { /* two streams and a transport: wall clock around the enqueue and the drain */
        const double t0 = now_s();
        CHECK(fluid_step(ctx, steps));
        CHECK(fluid_sync(ctx));
        ms[w] = (float)((now_s() - t0) * 1e3);
      } else {
        CHECK(fluid_time_steps(ctx, steps, &ms[w]));
      }
    }
    qsort(ms, windows, sizeof(float), cmp_float);
    double t_step = ms[windows / 2] / (steps > 0 ? steps : 1) * 1e-3;
    double cells = (double)prm.n * (prm.ny ? prm.ny : prm.n);
    double bpc = 88.0 + 16.0 * prm.diffusion_iters + 12.0 * prm.pressure_iters;
    fluid_diag dg; /* state after the timed steps: wavefront reductions on the device (own rows) */
    if (slabs) CHECK(fluid_comm_refresh(ctx));
    CHECK(fluid_diagnostics(ctx, &dg));
    fluid_comm_stats cs;
    CHECK(fluid_comm_info(ctx, &cs));
    printf("{\"n\": %d, \"diffusion_iters\": %d, \"pressure_iters\": %d, \"steps\": %d, "
           "\"slab_rank\": %d, \"slab_count\": %d, \"row0\": %d, \"rows\": %d, \"transport\": \"%s\", \"graph\": %d, "
           "\"ms_per_step\": %.6f, \"gcups\": %.4f, \"alg_bytes_per_cell\": %.0f, "
           "\"effective_GBps\": %.1f, \"frac_of_8TBps\": %.4f, "
           "\"max_abs_u\": %.9g, \"max_abs_v\": %.9g, \"max_abs_div\": %.9g, \"max_abs_p\": %.9g, "
           "\"max_jacobi_update\": %.9g, \"max_abs_div_velocity\": %.9g, \"dye\": [%.9g, %.9g, %.9g, %.9g]}\n",
           prm.n, prm.diffusion_iters, prm.pressure_iters, steps, prm.slab_rank, slabs ? prm.slab_count : 1, row0, rows,
           cs.transport, cs.graph, t_step * 1e3,
           cells / t_step / 1e9, bpc, bpc * cells / t_step / 1e9, bpc * cells / t_step / 8e12,
           dg.max_abs_u, dg.max_abs_v, dg.max_abs_div, dg.max_abs_p, dg.max_jacobi_update,
           dg.max_abs_div_velocity, dg.dye[0], dg.dye[1], dg.dye[2], dg.dye[3]);
  }

  if (ppm || pfm) {
    if (write_view(ctx, n, rows, view < 0 ? 0 : view, win_w, win_h, ppm, pfm)) { fprintf(stderr, "writing the frame failed\n"); return 1; }
  }
  if (dump) {
    char prefix[600];
    if (slabs) snprintf(prefix, sizeof prefix, "%s.rank%d", dump, prm.slab_rank);
    else snprintf(prefix, sizeof prefix, "%s", dump);
    if (dump_field(ctx, prefix, "velocity", FLUID_VELOCITY, 8, n, rows) ||
        dump_field(ctx, prefix, "density", FLUID_DENSITY, 16, n, rows) ||
        dump_field(ctx, prefix, "pressure", FLUID_PRESSURE, 4, n, rows)) {
      fprintf(stderr, "dump failed\n");
      return 1;
    }
  }
  fluid_destroy(ctx);
  return 0;
}
