// fluid_api.hip — host side of libfluid_hip.so: the C-ABI of include/fluid.h.
//
// Replaces the reference's L2 host layer (/root/reference/FluidSimulation/
// kernel.cu:435-708): buffer allocation, ping-pong indices, stage ordering and
// iteration counts.  The reference's ~99 texture/surface bind calls and 25
// null-stream launches per step become one hipGraph replay on the context's
// stream.  Product code: no CPU fallback, nothing from oracle/.
#include "fluid_internal.h"
#include "../../include/fluid.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

// ---- scratch of the entry point that allocates nothing (fluid_step_noalloc) ----------------
// The temporally blocked kernels write the rows a wave computes but must not store to a dump row
// (fluid_kernels_tb.hip, "Every load and store below is UNCONDITIONAL"): n * 8 bytes are used, the
// contexts allocate n * 16 + 256.  fluid_step_noalloc hands the kernels this file-scope array
// instead, sized for its stated maximum grid edge.  One copy per GPU the library runs on, held
// until the process ends.  Host code never uses the array's NAME as an address (on the host it is
// a stand-in, not device memory): the device address is taken once per device with
// hipGetSymbolAddress (scratch_setup, called by fluid_init with that device current) and kept in
// g_scratch_ptr[device]; kernels get that pointer as their `dump` argument.
#define FLUID_NOALLOC_SCRATCH_BYTES ((size_t)FLUID_NOALLOC_MAX_N * 16 + 256)
static __device__ __align__(256) unsigned char scratch[FLUID_NOALLOC_SCRATCH_BYTES];
static constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices] = {nullptr};
static std::mutex g_scratch_mutex;

// With `device` current.  hipSuccess and a non-null pointer, or the error of the lookup.
static hipError_t scratch_setup(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return hipSuccess;
  void *ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!ptr) return hipErrorInvalidSymbol;
  g_scratch_ptr[device] = ptr;
  return hipSuccess;
}

namespace {

struct GraphEntry {
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  int next_state = 0;
  long long exchanges = 0, bytes = 0;  // halo exchanges / bytes sent per launch (slab graphs with a transport)
  int halo_after = 0;                  // halo_state the captured steps leave behind
};

struct ProfRec {
  int kid;
  int e0, e1;
  double bytes;  // algorithmic bytes of the reference launches this launch stands for
  double moved;  // bytes the launch itself has to move: every input once, every output once
};

}  // namespace

struct fluid_ctx {
  fluid_params prm;
  Geom g;
  int row0 = 0, rows = 0;  // owned global rows [row0, row0 + rows)
  int own_r0 = 0;          // local row of the first owned row
  int halo_vel = 0, halo_dens = 0, adv_rows = 0;  // slab halos (fluid_slab_plan); 0 when whole-domain
  int groups = 1, jg = 0, halo_prs = 0;  // pressure groups: count, sweeps per group, rows exchanged
  int pgroup = 0;                        // next pressure group to run in the current step
  bool tail_edges_done = false;          // FLUID_PART_TAIL_DENSITY_EDGES ran, INTERIOR pending
  bool advect_early = false;             // FLUID_PART_ADVECT_INTERIOR ran for the step about to start
  bool slab = false, has_lo = false, has_hi = false;
  float2 *vel[2] = {nullptr, nullptr};   // velocityBufferArray[2]  kernel.cu:443-444
  float4 *dens[2] = {nullptr, nullptr};  // densityBufferArray[2]   kernel.cu:447-448
  float *prs[2] = {nullptr, nullptr};    // pressureBufferArray[2]  kernel.cu:452-453
  float *div = nullptr;                  // kernel.cu:455-456
  float *vort = nullptr;                 // kernel.cu:458-459 (allocated on first use)
  int vin = 0, din = 0, pin = 0;  // inputVelocityBuffer / inputDensityBuffer / inputPressureBuffer
  int *halo_flag = nullptr;       // device word: advection left the stored rows
  void *dump = nullptr;           // one row (n * 16 B): sink for rows a TB wave must not store
  unsigned long long *cksum = nullptr;  // fluid_checksum: one device word (allocated on first use)
  float *diag_partial = nullptr;  // fluid_diagnostics: per-block records + the result (allocated on first use)
  DiagResult *diag_out = nullptr;
  char *arena = nullptr;          // the seven field buffers live in one allocation (fluid_init)
  float alpha = 0.f, beta = 0.f;  // diffusion constants, kernel.cu:182-183
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::map<int, GraphEntry> graphs;
  // temporal-blocking plan (fused path): sweeps per launch, columns per lane, target waves
  int tb_kj = 10, tb_kd = 5, tb_cj = 2, tb_cd = 2, tb_waves = 8192, tb_minchunk = 4;
  float tb_rounds = 0.f;  // > 0: size the temporally blocked launches to this many resident rounds of workgroups (tb_chunk)
  int tb_waves_d = 0;  // target waves of the diffusion launches (0 = tb_waves): they hold more registers, so fewer are resident
  // chained chunks (fluid_kernels_tb.hip, Pipe::run): vertically adjacent chunks hand each other their
  // boundary rows through this buffer instead of both computing K rows of halo
  bool tb_chain = false;
  int tb_chain_kinds = 1;  // bit per kind of launch that is chained: 1 pressure (the only kind that gains), 2 divergence + pressure, 4 diffusion
  TbChain chain{};
  TbChain trace_only{};  // FLUID_TB_TRACE without chaining: only the trace pointer
  void *chain_mem = nullptr;
  size_t chain_words = 0, chain_flags = 0;  // capacity: 8-byte words of side records, flags
  bool tile_div = true;    // tile path: the last diffusion launch also writes the divergence (FLUID_TILE_DIV)
  bool tb_div = true;      // the same on the register-pipeline path (FLUID_TB_DIV): launch_diffuse_div_tb / launch_jacobi0_tb
  bool div_ready = false;  // ... and did, for the projection about to be enqueued
  int tile_ty = 32;   // rows an LDS tile stores (16 on the smallest grids, see fluid_init)
  bool tile = false;  // small grids: LDS-tiled sweeps (fluid_kernels_tile.hip) instead of the register pipeline
  bool nt_stores = false;  // grids far beyond the caches: non-temporal stores in the two streaming kernels
  float tb_j0_share = 1.f;  // share of a resident round the first pressure launch from p = 0 is sized to (it holds four waves per SIMD)
  int tb_sync = 0;  // bit per kind of temporally blocked launch (1 pressure, 2 divergence + pressure, 4 diffusion): barriers in the row loops
  int xcd_run_adv = 0, xcd_run_tail = 0;  // tiles per XCD run in k_advect_vel / k_project_advect_dens (tile_of_block); 0: plain order
  int graph_steps = 1;     // whole steps captured per graph (fluid_step)
  int force_k = 0;           // FLUID_TEST_FORCE_K, see next_pass
  std::vector<int> pass_d, pass_j;  // sweeps of the next launch when `left` diffusion / pressure sweeps remain
  bool direct_parts = true;  // step parts of one or two kernels are launched directly, not as graphs (FLUID_DIRECT_PARTS)
  int interior_split = 4;  // launches FLUID_PART_TAIL_DENSITY_INTERIOR is cut into (at most)
  int interior_piece_min = 1 << 21;  // ... each of at least this many cells
  // halo transport under the C ABI (fluid_comm_attach*): with one attached, fluid_step on a slab
  // context exchanges the halos itself, on comm_stream, overlapped with the interior launches
  Transport tr;
  bool has_comm = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_halo = nullptr;  // compute -> comm ("rows are final"), comm -> compute ("halos landed")
  int halo_state = 0;        // HALO_NONE / HALO_POSTED / HALO_JOINED, see comm_step
  bool comm_graph = true;    // capture whole slab steps (both streams + the transport's calls) as hipGraphs
  bool early_advect = true;  // FLUID_PART_ADVECT_INTERIOR before the halos are waited for (FLUID_SLAB_EARLY_ADVECT)
  long long comm_exchanges = 0, comm_bytes = 0;
  char comm_note[160] = {0};  // why graph capture was given up, if it was
  std::map<int, GraphEntry> comm_graphs;
  // inner slabs (fluid_params.inner_slabs >= 2, "inner_*" below): this context owns no field; `subs`
  // are the G slab contexts of the same GPU, stepped side by side on their own streams
  std::vector<fluid_ctx *> subs;
  hipEvent_t ev_fork = nullptr;     // this context's stream -> the slabs' streams
  std::vector<hipEvent_t> ev_done;  // ... and back
  bool inner_posted = false;        // the halo copies for the current state are enqueued
  bool peer_copies = false;         // halo rows move by hipMemcpyAsync device to device (slabs on different GPUs)
  int inner_of = 0;                 // > 0: this context is one of that many inner slabs
  bool prof = false;
  int enq_rc = 0;          // first failure while enqueueing (a launch_* that refused, a failed memset); see check_launch
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<ProfRec> recs;
  char err[256] = {0};
};

namespace {

int fail(fluid_ctx *c, int code, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIPCHK(c, call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail((c), e_ == hipErrorOutOfMemory ? FLUID_E_NOMEM : FLUID_E_DEVICE,       \
                  "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
  } while (0)

// The enq_* functions below are void (they also run under stream capture, where a launch cannot
// be undone); the first thing that goes wrong is kept in c->enq_rc and returned by check_launch
// after the enqueue, so that no caller ever sees FLUID_OK for a step that was not fully enqueued.
// launch_*_tb / launch_*_tile return false for a shape or sweep count this build does not hold.
#define LAUNCH(c, call)                                                                              \
  do {                                                                                               \
    if (!(call) && !(c)->enq_rc)                                                                      \
      (c)->enq_rc = fail((c), FLUID_E_UNSUPPORTED, "%s refused (shape or sweep count not in this build; %s:%d)", \
                         #call, __FILE__, __LINE__);                                                 \
  } while (0)
#define HIPNOTE(c, call)                                                                             \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess && !(c)->enq_rc)                                                             \
      (c)->enq_rc = fail((c), FLUID_E_DEVICE, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

inline size_t cells_local(const fluid_ctx *c) { return (size_t)c->g.L * c->g.n; }
inline size_t cells_owned(const fluid_ctx *c) { return (size_t)c->rows * c->g.n; }
inline int state_of(const fluid_ctx *c) { return c->vin | (c->din << 1) | (c->pin << 2); }
inline void set_state(fluid_ctx *c, int s) { c->vin = s & 1; c->din = (s >> 1) & 1; c->pin = (s >> 2) & 1; }
inline int st_delta(int before, int after) { return (before ^ after) & 7; }  // which ping-pong indices a batch of steps flips

void drop_graphs(fluid_ctx *c) {
  for (auto *m : {&c->graphs, &c->comm_graphs}) {
    for (auto &kv : *m) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    m->clear();
  }
}

// ---- per-launch profiling (fluid_profile_steps) -----------------------------
int prof_event(fluid_ctx *c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    c->ev_pool.push_back(e);
  }
  return (int)c->ev_used++;
}
struct ProfScope {
  fluid_ctx *c;
  int kid, e0 = -1;
  double bytes, moved;
  // moved_ < 0: the launch is one reference launch, its compulsory bytes are the algorithmic ones
  ProfScope(fluid_ctx *c_, int kid_, double bytes_, double moved_ = -1.0)
      : c(c_), kid(kid_), bytes(bytes_), moved(moved_ < 0 ? bytes_ : moved_) {
    if (c->prof) {
      e0 = prof_event(c);
      if (e0 >= 0) (void)hipEventRecord(c->ev_pool[e0], c->stream);
    }
  }
  ~ProfScope() {
    if (c->prof && e0 >= 0) {
      int e1 = prof_event(c);
      if (e1 >= 0) {
        (void)hipEventRecord(c->ev_pool[e1], c->stream);
        c->recs.push_back({kid, e0, e1, bytes, moved});
      }
    }
  }
};

// ---- stage enqueue: the reference's host stage functions --------------------
// Row ranges are local rows; for a whole-domain context [0, n).

// Local row range: the owned rows widened by `m` rows into each halo (clipped to
// what is stored; at a domain edge nothing is stored beyond it).  Slab contexts
// recompute the step's dependency cone in their halos (fluid.h, fluid_slab_plan);
// whole-domain contexts have no halo, so every range is the owned rows.
struct Range { int r0, r1; };
Range widened(const fluid_ctx *c, int m) {
  const int lo = c->own_r0, hi = c->g.L - c->own_r0 - c->rows;  // halo rows stored below / above
  return Range{c->own_r0 - (m < lo ? m : lo), c->own_r0 + c->rows + (m < hi ? m : hi)};
}

// The owned rows at least halo_vel rows away from a neighbouring slab: no neighbour waits for
// them at the end of a step (FLUID_PART_TAIL_DENSITY_INTERIOR) and they need no halo row at the
// start of the next (FLUID_PART_ADVECT_INTERIOR).  Empty on a slab shorter than two halos.
Range interior_band(const fluid_ctx *c) {
  const Range own = widened(c, 0);
  const int h = c->halo_vel < own.r1 - own.r0 ? c->halo_vel : own.r1 - own.r0;
  const int lo_end = c->has_lo ? own.r0 + h : own.r0;
  const int hi_beg = c->has_hi ? (own.r1 - h > lo_end ? own.r1 - h : lo_end) : own.r1;
  return Range{lo_end, hi_beg};
}

// simulateAdvection()  kernel.cu:542-555.  out_m: halo margin the result must cover.
void enq_advect_velocity(fluid_ctx *c, int out_m) {
  const double cells = (double)cells_owned(c);
  const Range r = widened(c, out_m);
  // after FLUID_PART_ADVECT_INTERIOR only the rows on either side of the interior band are left
  const Range band = c->advect_early ? interior_band(c) : Range{r.r1, r.r1};
  {  // [r.r0, band.r0) and [band.r1, r.r1) in one launch; the second range is empty without the band
    ProfScope p(c, K_ADVECT_VEL, 16.0 * cells * ((band.r0 - r.r0) + (r.r1 - band.r1)) / (r.r1 - r.r0));
    launch_advect_vel(c->stream, c->g, r.r0, band.r0, 0, c->g.L, c->prm.dt, c->vel[c->vin],
                      c->vel[c->vin ^ 1], c->halo_flag, c->nt_stores, c->xcd_run_adv, band.r1, r.r1);
  }
  c->advect_early = false;
  c->vin ^= 1;
}

// The interior band's share of the next step's advection, before that step's velocity halos have
// arrived: same kernel, same inputs (owned rows of the current velocity), same outputs.  A
// backtrace that leaves the owned rows (more than halo_vel rows long) raises the halo flag.
void enq_advect_interior(fluid_ctx *c) {
  const Range band = interior_band(c), own = widened(c, 0);
  if (band.r1 > band.r0) {
    ProfScope p(c, K_ADVECT_VEL, 16.0 * (double)c->g.n * (band.r1 - band.r0));
    launch_advect_vel(c->stream, c->g, band.r0, band.r1, own.r0, own.r1, c->prm.dt, c->vel[c->vin],
                      c->vel[c->vin ^ 1], c->halo_flag, c->nt_stores, c->xcd_run_adv);
  }
  c->advect_early = true;
}

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Next launch's sweep count: split `left` sweeps into the fewest launches the instantiated
// counts (fluid_kernels_tb.hip tb_has_k) allow with none above kmax, and among those the most
// even split (40 with kmax 14 -> 14 + 13 + 13, not 14 + 14 + 10 + 2).
int plan_pass(int left, int kmax, int cols, bool float2_field, bool tile) {
  if (left <= 1) return 1;
  std::vector<int> cnt(left + 1, 1 << 20), big(left + 1, 0), pick(left + 1, 1);
  cnt[0] = 0;
  for (int j = 1; j <= left; ++j)
    for (int k = 1; k <= kmax && k <= j; ++k) {
      if (!(tile ? tile_has_k(k, float2_field) : tb_has_k(k, cols, float2_field))) continue;
      const int c = cnt[j - k] + 1, m = big[j - k] > k ? big[j - k] : k;
      if (c < cnt[j] || (c == cnt[j] && m < big[j])) { cnt[j] = c; big[j] = m; pick[j] = k; }
    }
  // launch the largest pass first (the order of the passes does not change the result)
  int best = pick[left];
  for (int j = left; j > 0; j -= pick[j]) best = pick[j] > best ? pick[j] : best;
  return best;
}

// The same, from the tables fluid_init fills for this context's plan (pass_d / pass_j: the answer
// for every `left` up to D / J): enqueueing a step does no planning and no host allocation.
int next_pass(const fluid_ctx *c, int left, bool float2_field) {
  // FLUID_TEST_FORCE_K (tests only): every pass asks for this sweep count, instantiated or not
  if (c->force_k > 0) return c->force_k < left ? c->force_k : left;
  const std::vector<int> &t = float2_field ? c->pass_d : c->pass_j;
  return left >= 0 && (size_t)left < t.size() ? t[left] : 1;
}
void fill_pass_tables(fluid_ctx *c) {
  c->pass_d.assign(c->prm.diffusion_iters + 1, 1);
  c->pass_j.assign(c->prm.pressure_iters + 1, 1);
  for (int l = 1; l <= c->prm.diffusion_iters; ++l) c->pass_d[l] = plan_pass(l, c->tb_kd, c->tile ? 0 : c->tb_cd, true, c->tile);
  for (int l = 1; l <= c->prm.pressure_iters; ++l) c->pass_j[l] = plan_pass(l, c->tb_kj, c->tile ? 0 : c->tb_cj, false, c->tile);
  if (!c->tile && c->tb_rounds > 0.f) {  // ask the runtime for the occupancies now, not while a step is enqueued or captured
    for (int k : c->pass_d) { (void)tb_resident_blocks(2, k, c->tb_cd); (void)tb_resident_blocks(3, k, c->tb_cd); }
    for (int k : c->pass_j) {
      (void)tb_resident_blocks(0, k, c->tb_cj); (void)tb_resident_blocks(1, k, c->tb_cj); (void)tb_resident_blocks(4, k, c->tb_cj);
    }
  }
}

// Rows per wave (chunk) of a temporally blocked launch of `kind` (0 pressure, 1 divergence +
// pressure, 2 diffusion).  With tb_rounds > 0 the launch is sized to that many RESIDENT ROUNDS of
// workgroups: resident = what the runtime says fits the device at this kernel's register count, a
// workgroup is four strips wide, so chunks = floor(rounds * resident / workgroups per chunk-row) —
// a launch a few workgroups over a round pays a whole extra round for them (8192^2, J = 200:
// 3034 waves asked for = 37 x 21 workgroups = 777 > 768 resident: 5.26 ms against 4.45 ms at 35 x 21).
// Otherwise about tb_waves waves (the largest grids: many rounds, shorter chunks even out the end).
int tb_chunk_count(const fluid_ctx *c, int kind, int k, int cols) {
  int chunks = 0;
  const int resident = c->tb_rounds > 0.f ? tb_resident_blocks(kind, k, cols) : 0;
  const int kg = kind == 3 ? k + 1 : k;  // strips of the diffusion + divergence launch are laid out for one stage more
  if (resident > 0) {
    chunks = (int)(c->tb_rounds * resident * (kind == 4 ? c->tb_j0_share : 1.f)) / tb_blocks_per_row(c->g.n, kg, cols);
  } else {
    const int waves = (kind == 2 || kind == 3) && c->tb_waves_d > 0 ? c->tb_waves_d : c->tb_waves;
    const int hx = (kg + cols - 1) / cols * cols;
    const int vw = 64 * cols - 2 * hx;
    const int strips = (c->g.n + vw - 1) / vw;
    chunks = waves / strips;
  }
  return chunks < 1 ? 1 : chunks;
}
int tb_chunk(const fluid_ctx *c, int kind, int k, int cols, int rows) {
  const int chunks = tb_chunk_count(c, kind, k, cols);
  int chunk = (rows + chunks - 1) / chunks;
  if (chunk < c->tb_minchunk * k) chunk = c->tb_minchunk * k;  // bound the y-halo recompute (2k rows per chunk)
  // a chunk's stored rows are one buffer resource with 32-bit offsets (fluid_kernels_tb.hip, tb_chunk_fits_buffer;
  // 8 bytes per cell at most): only a plan of a handful of chunk-rows on a grid of tens of thousands of columns gets here
  const long long fit = 0xFFFFFF00ll / ((long long)c->g.n * 8);
  if (chunk > fit && fit >= 1) chunk = (int)fit;
  if (c->tb_chain && ((c->tb_chain_kinds >> kind) & 1) && tb_has_k(k, cols, kind == 2) && cols == 2) {
    // chained chunks: a multiple of the pipeline's slot period, long enough for fill and tail; a
    // launch whose rows then fit one chunk (or whose k has no unrolled fill) runs unchained
    // The bottom chunk has nobody to take rows from and ends the old way, k full steps later than
    // the others; a launch that is one resident round ends with its slowest wave, so the bottom
    // chunk is made k rows shorter: (chunks - 1) * chunk + (chunk - k) = rows.
    const int m = tb_chain_chunk_multiple();
    if (chunks > 1) chunk = (rows + k + chunks - 1) / chunks;
    chunk = (chunk + m - 1) / m * m;
    if (chunk < tb_chain_min_chunk(k)) chunk = tb_chain_min_chunk(k);
  }
  return chunk;
}
// What goes along with a temporally blocked launch besides its arguments: the context's side buffer
// if the launch is to run chained and fits it, and the trace buffer (FLUID_TB_TRACE).
TbChain tb_extra(const fluid_ctx *c, int kind, int k, int cols, int chunk, int rows) {
  TbChain x{};
  const int chunks = (rows + chunk - 1) / chunk;
  const bool fits = tb_chain_flags(c->g.n, kind == 3 ? k + 1 : k, cols, chunks) <= c->chain_flags;
  if (c->trace_only.trace && fits) x.trace = c->trace_only.trace;
  x.sync = (c->tb_sync >> (kind == 3 ? 2 : kind == 4 ? 1 : kind)) & 1;  // 3, 4: as the launches they stand in for
  if (kind >= 3) return x;
  if (c->tb_chain && ((c->tb_chain_kinds >> kind) & 1) && c->chain_mem && fits && tb_chain_chunk_ok(k, cols, chunk, rows) &&
      tb_chain_words(c->g.n, k, cols, kind == 2, chunks) <= c->chain_words) {
    x.side = c->chain.side;
    x.flags = c->chain.flags;
    x.status = c->chain.status;
  }
  return x;
}
// What the plan's launches can need at most (a launch has at most tb_chunk_count chunk-rows, and
// never more than its rows over the shortest chained chunk).
void tb_chain_capacity(const fluid_ctx *c, size_t *words, size_t *flags) {
  *words = *flags = 0;
  auto one = [&](int kind, int k, int cols) {
    if (k < 1 || cols != 2 || !tb_has_k(k, cols, kind == 2)) return;
    int chunks = tb_chunk_count(c, kind, k, cols);
    const int most = c->g.L / tb_chain_min_chunk(k) + 1;
    if (chunks > most) chunks = most;
    const size_t w = tb_chain_words(c->g.n, k, cols, kind == 2, chunks), f = tb_chain_flags(c->g.n, k, cols, chunks);
    if (w > *words) *words = w;
    if (f > *flags) *flags = f;
  };
  // (every kind, whichever of them FLUID_TB_CHAIN_KINDS chains: the trace buffer is sized by the flags)
  for (int k : c->pass_d) one(2, k, c->tb_cd);
  for (int k : c->pass_j) { one(0, k, c->tb_cj); one(1, k, c->tb_cj); }
}

bool use_tb(const fluid_ctx *c, int cols) {
  return !(c->prm.flags & FLUID_FLAG_UNFUSED) && tb_supported(c->g.n, cols);
}

// simulateDiffusion()  kernel.cu:582-602.  out_m: margin the final iterate must cover; a
// sweep with `left` sweeps still to come must cover out_m + left.
// with_div (whole steps of a whole-domain context on the tile path): the launch that finishes the
// diffusion also writes the divergence of its result, and the first pressure launch reads it
// (c->div_ready) instead of recomputing it from the velocity for every cell of its tiles.
void enq_diffuse(fluid_ctx *c, int out_m, bool with_div = false) {
  const double cells = (double)cells_owned(c);
  if (c->tile && !(c->prm.flags & FLUID_FLAG_UNFUSED)) {
    for (int left = c->prm.diffusion_iters; left > 0;) {
      const int k = next_pass(c, left, true);
      const bool div = with_div && left == k;  // the last pass
      ProfScope p(c, K_DIFFUSE_TILE, 16.0 * k * cells + (div ? 12.0 * cells : 0.0), (div ? 20.0 : 16.0) * cells);
      const Range r = widened(c, out_m + left - k);
      LAUNCH(c, launch_diffuse_tile(c->stream, c->g, r.r0, r.r1, k, c->tile_ty, c->alpha, c->beta, c->vel[c->vin],
                                    c->vel[c->vin ^ 1], div ? c->div : nullptr));
      if (div) c->div_ready = true;
      c->vin ^= 1;
      left -= k;
    }
    return;
  }
  if (use_tb(c, c->tb_cd)) {
    for (int left = c->prm.diffusion_iters; left > 0;) {
      const int k = next_pass(c, left, true);
      const Range r = widened(c, out_m + left - k);
      if (with_div && c->tb_div && left == k && tb_has_div_k(k, c->tb_cd)) {
        // the last pass of a whole step also writes the divergence of its result; the first pressure launch reads
        // that (c->div_ready) instead of the velocity
        ProfScope p(c, K_DIFFUSE_DIV_TB, 16.0 * k * cells + 12.0 * cells, 20.0 * cells);
        const int chunk = tb_chunk(c, 3, k, c->tb_cd, r.r1 - r.r0);
        const TbChain extra = tb_extra(c, 3, k, c->tb_cd, chunk, r.r1 - r.r0);
        LAUNCH(c, launch_diffuse_div_tb(c->stream, c->g, r.r0, r.r1, k, c->tb_cd, chunk, c->alpha, c->beta, c->vel[c->vin],
                                     c->vel[c->vin ^ 1], c->div, c->dump, &extra));
        c->div_ready = true;
        c->vin ^= 1;
        left -= k;
        continue;
      }
      ProfScope p(c, K_DIFFUSE_TB, 16.0 * k * cells, 16.0 * cells);
      const int chunk = tb_chunk(c, 2, k, c->tb_cd, r.r1 - r.r0);
      const TbChain extra = tb_extra(c, 2, k, c->tb_cd, chunk, r.r1 - r.r0);
      LAUNCH(c, launch_diffuse_tb(c->stream, c->g, r.r0, r.r1, k, c->tb_cd, chunk, c->alpha, c->beta, c->vel[c->vin],
                               c->vel[c->vin ^ 1], c->dump, &extra));
      c->vin ^= 1;
      left -= k;
    }
    // (the reference flips buffers once per sweep, kernel.cu:600; which physical buffer holds
    // the current iterate is not observable through the ABI)
    return;
  }
  for (int i = 0; i < c->prm.diffusion_iters; ++i) {
    ProfScope p(c, K_DIFFUSE, 16.0 * cells);
    const Range r = widened(c, out_m + c->prm.diffusion_iters - 1 - i);
    launch_diffuse(c->stream, c->g, r.r0, r.r1, c->alpha, c->beta, c->vel[c->vin],
                   c->vel[c->vin ^ 1]);
    c->vin ^= 1;
  }
}

// simulateVorticity()  kernel.cu:557-580
// out_m: halo margin the forced velocity must cover (slabs; the curl is needed one row beyond it, and
// reads the diffused velocity one row beyond that)
void enq_vorticity(fluid_ctx *c, int out_m = 0) {
  const double cells = (double)cells_owned(c);
  {
    ProfScope p(c, K_VORTICITY, 12.0 * cells);
    const Range r = widened(c, out_m + 1);
    launch_vorticity(c->stream, c->g, r.r0, r.r1, c->vel[c->vin], c->vort);
  }
  {
    ProfScope p(c, K_ADD_VORTICITY, 20.0 * cells);
    const Range r = widened(c, out_m);
    launch_add_vorticity(c->stream, c->g, r.r0, r.r1, c->prm.dt, c->vort, c->vel[c->vin]);
  }
}

// projection()  kernel.cu:604-662, first half: divergence, pressure clear and the Jacobi sweeps,
// in `groups` groups (one unless the context was made with pressure_groups; then the caller
// exchanges halo_prs pressure rows between groups).  The projected velocity is produced on the
// owned rows; it needs p^J one row beyond, hence within a group p on 1 + left rows beyond when
// `left` sweeps of the group are still to come, and the divergence on (group length) rows.
void enq_pressure_group(fluid_ctx *c, int g) {
  const double cells = (double)cells_owned(c);
  const int J = c->prm.pressure_iters;
  const int glen = c->groups > 1 ? c->jg : J;
  const int first = g * glen, last = first + glen < J ? first + glen : J;
  int left = last - first;  // sweeps of this group
  const bool tb = use_tb(c, c->tb_cj);
  const bool tile = c->tile && !(c->prm.flags & FLUID_FLAG_UNFUSED);
  if (tile) {  // small grids: LDS-tiled sweeps, same ranges as the register pipeline below
    const int div_m = c->groups > 1 ? glen : -1;
    bool first = g == 0;
    while (left > 0) {
      const int k = next_pass(c, left, false);
      if (first && c->div_ready) {  // the diffusion launch wrote the divergence: p^0 == 0, 4 + 4 bytes per cell
        ProfScope p(c, K_JACOBI_TILE, 12.0 * k * cells, 8.0 * cells);
        const Range r = widened(c, 1 + left - k);
        LAUNCH(c, launch_jacobi_tile(c->stream, c->g, r.r0, r.r1, k, c->tile_ty, nullptr, c->div, c->prs[c->pin ^ 1]));
      } else if (first) {
        ProfScope p(c, K_JACOBI_DIV_TILE, 12.0 * cells + 12.0 * k * cells, 16.0 * cells);  // velocity in, divergence and p out
        const Range r = widened(c, div_m >= 0 ? div_m : 1 + left - k);
        LAUNCH(c, launch_jacobi_div_tile(c->stream, c->g, r.r0, r.r1, k, c->tile_ty, c->vel[c->vin], c->div, c->prs[c->pin ^ 1]));
      } else {
        ProfScope p(c, K_JACOBI_TILE, 12.0 * k * cells, 12.0 * cells);
        const Range r = widened(c, 1 + left - k);
        LAUNCH(c, launch_jacobi_tile(c->stream, c->g, r.r0, r.r1, k, c->tile_ty, c->prs[c->pin], c->div, c->prs[c->pin ^ 1]));
      }
      first = false;
      c->pin ^= 1;
      left -= k;
    }
    if (first) {  // J == 0: still the divergence and the pressure clear
      if (!c->div_ready) {
        ProfScope p(c, K_DIVERGENCE, 12.0 * cells);
        const Range r = widened(c, div_m >= 0 ? div_m : J);
        launch_divergence(c->stream, c->g, r.r0, r.r1, c->vel[c->vin], c->div);
      }
      HIPNOTE(c, hipMemsetAsync(c->prs[c->pin], 0, cells_local(c) * sizeof(float), c->stream));
    }
    c->div_ready = false;
    return;
  }
  if (g == 0) {
    // the divergence must cover what every later group's first sweep reads: glen rows beyond
    const int div_m = c->groups > 1 ? glen : -1;
    if (tb && left > 0 && c->div_ready) {
      // the divergence is in memory (written by the last diffusion launch): resetPressure + the first sweeps
      const int k = next_pass(c, left, false);
      ProfScope p(c, K_JACOBI0_TB, 12.0 * k * cells, 8.0 * cells);  // divergence in, p out
      const Range r = widened(c, 1 + left - k);
      const int chunk = tb_chunk(c, 4, k, c->tb_cj, r.r1 - r.r0);
      const TbChain extra = tb_extra(c, 4, k, c->tb_cj, chunk, r.r1 - r.r0);
      LAUNCH(c, launch_jacobi0_tb(c->stream, c->g, r.r0, r.r1, k, c->tb_cj, chunk, c->div, c->prs[c->pin ^ 1], c->dump, &extra));
      c->pin ^= 1;
      left -= k;
    } else if (tb && left > 0) {
      // divergence + resetPressure (p^0 == 0 is a constant, not a buffer) + the first sweeps
      const int k = next_pass(c, left, false);
      ProfScope p(c, K_JACOBI_DIV_TB, 12.0 * cells + 12.0 * k * cells, 16.0 * cells);  // velocity in, divergence and p out
      const Range r = widened(c, div_m >= 0 ? div_m : 1 + left - k);
      const int chunk = tb_chunk(c, 1, k, c->tb_cj, r.r1 - r.r0);
      const TbChain extra = tb_extra(c, 1, k, c->tb_cj, chunk, r.r1 - r.r0);
      LAUNCH(c, launch_jacobi_div_tb(c->stream, c->g, r.r0, r.r1, k, c->tb_cj, chunk, c->vel[c->vin], c->div,
                                  c->prs[c->pin ^ 1], c->dump, &extra));
      c->pin ^= 1;
      left -= k;
    } else {
      if (!c->div_ready) {
        ProfScope p(c, K_DIVERGENCE, 12.0 * cells);
        const Range r = widened(c, div_m >= 0 ? div_m : J);
        launch_divergence(c->stream, c->g, r.r0, r.r1, c->vel[c->vin], c->div);
      }
      // resetPressure()  kernel.cu:530-540: only the pressure clear is live (the
      // other two targets are overwritten before they are read).
      HIPNOTE(c, hipMemsetAsync(c->prs[c->pin], 0, cells_local(c) * sizeof(float), c->stream));
    }
    c->div_ready = false;
  }
  if (tb) {
    while (left > 0) {
      const int k = next_pass(c, left, false);
      ProfScope p(c, K_JACOBI_TB, 12.0 * k * cells, 12.0 * cells);
      const Range r = widened(c, 1 + left - k);
      const int chunk = tb_chunk(c, 0, k, c->tb_cj, r.r1 - r.r0);
      const TbChain extra = tb_extra(c, 0, k, c->tb_cj, chunk, r.r1 - r.r0);
      LAUNCH(c, launch_jacobi_tb(c->stream, c->g, r.r0, r.r1, k, c->tb_cj, chunk, c->prs[c->pin], c->div,
                              c->prs[c->pin ^ 1], c->dump, &extra));
      c->pin ^= 1;
      left -= k;
    }
  } else {
    for (; left > 0; --left) {
      ProfScope p(c, K_JACOBI, 12.0 * cells);
      const Range r = widened(c, left);
      launch_jacobi(c->stream, c->g, r.r0, r.r1, c->prs[c->pin], c->div, c->prs[c->pin ^ 1]);
      c->pin ^= 1;
    }
  }
}

// projection(), second half: the gradient subtraction, kernel.cu:655-661.
void enq_gradient(fluid_ctx *c, bool with_density) {
  const double cells = (double)cells_owned(c);
  if (with_density) {
    // + simulateDensityAdvection() kernel.cu:664-685 on the same rows (see enq_step)
    ProfScope p(c, K_PROJECT_ADVECT_DENS, 60.0 * cells, 52.0 * cells);  // the projected velocity stays in registers
    const Range r = widened(c, 0);
    const Range valid = c->slab ? widened(c, c->halo_dens) : Range{0, c->g.L};
    launch_project_advect_dens(c->stream, c->g, r.r0, r.r1, valid.r0, valid.r1, c->prm.dt,
                               c->vel[c->vin], c->prs[c->pin], c->vel[c->vin ^ 1], c->dens[c->din],
                               c->dens[c->din ^ 1], c->halo_flag, c->nt_stores, c->xcd_run_tail);
    c->vin ^= 1;
    c->din ^= 1;
    return;
  }
  ProfScope p(c, K_PROJECT, 20.0 * cells);
  const Range r = widened(c, 0);
  launch_project(c->stream, c->g, r.r0, r.r1, c->vel[c->vin], c->prs[c->pin], c->vel[c->vin ^ 1]);
  c->vin ^= 1;
}

// The fused gradient subtraction + density advection on a sub-range of the owned rows: the rows
// within halo_vel of a neighbouring slab (edges: what the next exchange sends) or the rest.
void enq_gradient_density_rows(fluid_ctx *c, bool edges) {
  const Range own = widened(c, 0), band = interior_band(c);
  const int lo_end = band.r0, hi_beg = band.r1;  // edges: [own.r0, lo_end) and [hi_beg, own.r1)
  const Range valid = c->slab ? widened(c, c->halo_dens) : Range{0, c->g.L};
  // EDGES reads the current buffers and writes the other halves, then flips; INTERIOR runs after
  // the flip, so its input is the other half and its output the current one.
  const int vi = edges ? c->vin : c->vin ^ 1, di = edges ? c->din : c->din ^ 1;
  if (edges) {  // both edges in one launch (two launches of ~11 us each are mostly ramp-up and drain)
    const int rows = (lo_end - own.r0) + (own.r1 - hi_beg);
    if (rows > 0) {
      ProfScope p(c, K_PROJECT_ADVECT_DENS, 60.0 * (double)c->g.n * rows, 52.0 * (double)c->g.n * rows);
      launch_project_advect_dens(c->stream, c->g, own.r0, lo_end, valid.r0, valid.r1, c->prm.dt, c->vel[vi],
                                 c->prs[c->pin], c->vel[vi ^ 1], c->dens[di], c->dens[di ^ 1], c->halo_flag,
                                 false, c->xcd_run_tail, hi_beg, own.r1);  // regular stores: the edge rows are packed next
    }
  } else {
    // The interior goes out as a few launches, not one: work queued on another stream meanwhile
    // (the halo messages) is sure of a way in at each boundary (a kernel of few, large
    // workgroups did not get in while one long launch kept every CU refilled).
    const int rows = hi_beg - lo_end;
    int k = c->interior_split;
    while (k > 1 && (double)c->g.n * rows / k < (double)c->interior_piece_min) --k;  // keep every piece a full launch
    for (int i = 0; i < k; ++i) {
      const Range r{lo_end + (int)((long)rows * i / k), lo_end + (int)((long)rows * (i + 1) / k)};
      if (r.r1 <= r.r0) continue;
      ProfScope p(c, K_PROJECT_ADVECT_DENS, 60.0 * (double)c->g.n * (r.r1 - r.r0), 52.0 * (double)c->g.n * (r.r1 - r.r0));
      launch_project_advect_dens(c->stream, c->g, r.r0, r.r1, valid.r0, valid.r1, c->prm.dt, c->vel[vi],
                                 c->prs[c->pin], c->vel[vi ^ 1], c->dens[di], c->dens[di ^ 1], c->halo_flag,
                                 c->nt_stores, c->xcd_run_tail);
    }
  }
  if (edges) {
    c->vin ^= 1;
    c->din ^= 1;
  }
}

void enq_project(fluid_ctx *c, bool with_density = false) {  // one group: the whole projection()
  enq_pressure_group(c, 0);
  enq_gradient(c, with_density);
}

// simulateDensityAdvection()  kernel.cu:664-685
void enq_advect_density(fluid_ctx *c) {
  const double cells = (double)cells_owned(c);
  ProfScope p(c, K_ADVECT_DENS, 40.0 * cells);
  const Range r = widened(c, 0);
  const Range valid = c->slab ? widened(c, c->halo_dens) : Range{0, c->g.L};  // density halo is A rows
  launch_advect_dens(c->stream, c->g, r.r0, r.r1, valid.r0, valid.r1, c->prm.dt, c->vel[c->vin],
                     c->dens[c->din], c->dens[c->din ^ 1], c->halo_flag);
  c->din ^= 1;
}

// simulationStep()  kernel.cu:701-708
void enq_velocity_head(fluid_ctx *c) {
  // margins of the dependency cone (zero-width for whole-domain contexts, see widened())
  const int glen = c->groups > 1 ? c->jg : c->prm.pressure_iters;
  const int m_diff = glen + 1;  // diffused velocity feeds divergence on glen rows
  const bool vrt = (c->prm.flags & FLUID_FLAG_VORTICITY) != 0;
  const int m_vrt = vrt ? 2 : 0;  // the confinement force on those rows reads the diffused velocity two rows further
  enq_advect_velocity(c, m_diff + m_vrt + c->prm.diffusion_iters);
  // (vorticity confinement changes the velocity between the two stages.  Slabs: the pipeline path's fused launch
  // covers widened(glen + 1) and its divergence is good one row inside that — widened(glen), all any pressure
  // launch reads; the tile path keeps the separate form there)
  enq_diffuse(c, m_diff + m_vrt, (c->tile ? c->tile_div && !c->slab : c->tb_div) && !vrt);
  if (vrt) enq_vorticity(c, m_diff);  // kernel.cu:705
  enq_pressure_group(c, 0);
}
void enq_step(fluid_ctx *c, int parts) {
  if (parts & FLUID_PART_ADVECT_INTERIOR) enq_advect_interior(c);
  if (parts & FLUID_PART_VELOCITY) {  // one pressure group (checked by the callers)
    enq_velocity_head(c);
    // When the whole step is enqueued at once, the gradient subtraction and the density
    // advection (both over the owned rows) run as one kernel.
    const bool fuse = (parts & FLUID_PART_DENSITY) && !(c->prm.flags & FLUID_FLAG_UNFUSED);
    enq_gradient(c, fuse);
    if (fuse) return;
  }
  if (parts & FLUID_PART_VELOCITY_HEAD) enq_velocity_head(c);
  if (parts & FLUID_PART_PRESSURE_GROUP) enq_pressure_group(c, c->pgroup);
  if (parts & FLUID_PART_VELOCITY_TAIL) enq_gradient(c, false);
  if (parts & FLUID_PART_TAIL_DENSITY_EDGES) enq_gradient_density_rows(c, true);
  if (parts & FLUID_PART_TAIL_DENSITY_INTERIOR) enq_gradient_density_rows(c, false);
  if (parts & FLUID_PART_DENSITY) enq_advect_density(c);
}

int check_launch(fluid_ctx *c) {
  if (c->enq_rc) {  // a launch_* refused or a memset failed while enqueueing (message already in c->err)
    const int rc = c->enq_rc;
    c->enq_rc = 0;
    (void)hipGetLastError();
    return rc;
  }
  hipError_t e = hipGetLastError();  // cudaPeekAtLastError after every launch, e.g. kernel.cu:552
  if (e != hipSuccess) return fail(c, FLUID_E_DEVICE, "kernel launch: %s", hipGetErrorString(e));
  return FLUID_OK;
}

// launch == false: only build (capture and instantiate) the graphs the call would replay.
int run_steps(fluid_ctx *c, int nsteps, int parts = FLUID_PART_VELOCITY | FLUID_PART_DENSITY, bool launch = true) {
  // Parts of one or two kernels are launched directly: a graph launch costs more than the
  // launches it replaces (measured: 2048-row slab of 16384^2 driven in parts).
  const int small_parts = FLUID_PART_DENSITY | FLUID_PART_VELOCITY_TAIL | FLUID_PART_TAIL_DENSITY_EDGES |
                          FLUID_PART_TAIL_DENSITY_INTERIOR | FLUID_PART_ADVECT_INTERIOR;
  const bool direct = (c->prm.flags & FLUID_FLAG_NO_GRAPH) || c->prof || (c->direct_parts && (parts & ~small_parts) == 0);
  // Whole steps of a whole-domain context are captured several to a graph, in power-of-two batches
  // (21 steps = 16 + 4 + 1): a graph launch costs 4-5 us of its own on this runtime, 12 % of the
  // reference's default step (512^2, J = 10: 40.5 -> 35.6 us per step) and what separated graph
  // replay from back-to-back launches at every size up to 4096^2.
  const int whole = FLUID_PART_VELOCITY | FLUID_PART_DENSITY;
  const int per_graph = (!direct && parts == whole && !c->slab) ? c->graph_steps : 1;
  if (direct && !launch) return FLUID_OK;
  const int state0 = state_of(c);
  for (int s = 0; s < nsteps;) {
    if (direct) {
      enq_step(c, parts);
      int rc = check_launch(c);
      if (rc) return rc;
      ++s;
      continue;
    }
    int batch = 1;
    while (batch * 2 <= per_graph && batch * 2 <= nsteps - s) batch *= 2;
    const int key = state_of(c) | (parts << 3) | (c->advect_early ? 1 << 11 : 0) | ((c->pgroup & 0xfff) << 12) |
                    (batch << 24);
    auto it = c->graphs.find(key);
    s += batch;
    if (it == c->graphs.end()) {
      GraphEntry ge;
      const bool early0 = c->advect_early;  // consumed by the enqueue; the launch below clears it
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      for (int b = 0; b < batch; ++b) enq_step(c, parts);
      hipError_t e = hipStreamEndCapture(c->stream, &ge.graph);
      ge.next_state = state_of(c);
      set_state(c, key & 7);
      c->advect_early = early0;
      if (c->enq_rc) {  // something refused to enqueue: no graph, and the caller hears about it
        if (e == hipSuccess && ge.graph) (void)hipGraphDestroy(ge.graph);
        return check_launch(c);
      }
      if (e != hipSuccess) return fail(c, FLUID_E_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
      HIPCHK(c, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
      it = c->graphs.emplace(key, ge).first;
    }
    if (launch) HIPCHK(c, hipGraphLaunch(it->second.exec, c->stream));
    set_state(c, it->second.next_state);
    if (launch && (parts & (FLUID_PART_VELOCITY | FLUID_PART_VELOCITY_HEAD))) c->advect_early = false;
  }
  if (!launch) set_state(c, state0);
  return FLUID_OK;
}

int ensure_vort(fluid_ctx *c) {
  if (!c->vort) {
    HIPCHK(c, hipMalloc(&c->vort, cells_local(c) * sizeof(float)));
    HIPCHK(c, hipMemsetAsync(c->vort, 0, cells_local(c) * sizeof(float), c->stream));
  }
  return FLUID_OK;
}

// After the stream has drained: did an advection gather leave the rows it may read?
int check_halo_flag(fluid_ctx *c) {
  if (c->chain_mem) {  // chained chunks: did a wave give up waiting for its neighbour's rows?  (never seen)
    int st = 0;
    HIPCHK(c, hipMemcpy(&st, c->chain.status, sizeof st, hipMemcpyDeviceToHost));
    if (st) return fail(c, FLUID_E_DEVICE, "a chained chunk gave up waiting for the rows of the chunk below it");
  }
  if (!c->slab) return FLUID_OK;
  int flag = 0;
  HIPCHK(c, hipMemcpy(&flag, c->halo_flag, sizeof flag, hipMemcpyDeviceToHost));
  if (flag) return fail(c, FLUID_E_HALO, "a backtrace was longer than halo_rows=%d rows", c->halo_dens);
  return FLUID_OK;
}

void *field_ptr(fluid_ctx *c, int field, size_t *elem) {
  switch (field) {
    case FLUID_VELOCITY: *elem = sizeof(float2); return c->vel[c->vin];
    case FLUID_DENSITY: *elem = sizeof(float4); return c->dens[c->din];
    case FLUID_PRESSURE: *elem = sizeof(float); return c->prs[c->pin];
    case FLUID_DIVERGENCE: *elem = sizeof(float); return c->div;
    case FLUID_VORTICITY: *elem = sizeof(float); return c->vort;
  }
  return nullptr;
}

}  // namespace

extern "C" {

// inner slabs (fluid_params.inner_slabs >= 2): the whole-domain entry points of such a context
static int inner_reset(fluid_ctx *P);
static int inner_add_source(fluid_ctx *P, float sx, float sy, float fx, float fy, const float rgba[4]);
static int inner_steps(fluid_ctx *P, int nsteps, bool launch);
static int inner_read(fluid_ctx *P, int field, void *dst, size_t dst_bytes);
static int inner_write(fluid_ctx *P, int field, const void *src, size_t src_bytes);
static int inner_sync(fluid_ctx *P);
static int inner_diagnostics(fluid_ctx *P, fluid_diag *out);
static int inner_time_steps(fluid_ctx *P, int nsteps, float *ms_out);
static inline bool is_inner(const fluid_ctx *c) { return !c->subs.empty(); }
#define NOT_ON_INNER(c, what)                                                                        \
  do {                                                                                               \
    if (is_inner(c)) return fail((c), FLUID_E_UNSUPPORTED, what " is not available with inner_slabs"); \
  } while (0)

int fluid_default_params(fluid_params *p) {
  if (!p) return FLUID_E_INVALID;
  memset(p, 0, sizeof *p);
  p->n = 512;               // kernel.cu:438
  p->diffusion_iters = 10;  // kernel.cu:584
  p->pressure_iters = 10;   // kernel.cu:620
  p->dt = 0.1f;             // kernel.cu:31
  p->viscosity = 0.01f;     // kernel.cu:181
  p->splat_radius = 0.001f; // kernel.cu:368
  return FLUID_OK;
}

int fluid_slab_plan(const fluid_params *p, fluid_slab_info *out) {
  if (!p || !out) return FLUID_E_INVALID;
  if (p->n < 4 || p->diffusion_iters < 0 || p->pressure_iters < 0) return FLUID_E_INVALID;
  if (p->ny != 0 && p->ny < 4) return FLUID_E_INVALID;
  const int ny = p->ny ? p->ny : p->n;  // rows of the whole grid (square unless fluid_params.ny says otherwise)
  memset(out, 0, sizeof *out);
  out->n = p->n;
  if (p->slab_count <= 1) {
    out->rows = ny;
    out->store_rows = ny;
    out->pressure_groups = 1;
    return FLUID_OK;
  }
  if (p->slab_rank < 0 || p->slab_rank >= p->slab_count) return FLUID_E_INVALID;
  const int a = p->halo_rows > 0 ? p->halo_rows : 8;
  out->row0 = (int)((long long)ny * p->slab_rank / p->slab_count);
  const int row1 = (int)((long long)ny * (p->slab_rank + 1) / p->slab_count);
  out->rows = row1 - out->row0;
  out->halo_dens = a;
  out->pressure_groups = p->pressure_groups > 1 && p->pressure_iters > 1
                             ? (p->pressure_groups < p->pressure_iters ? p->pressure_groups : p->pressure_iters) : 1;
  const int jg = (p->pressure_iters + out->pressure_groups - 1) / out->pressure_groups;
  out->pressure_groups = jg > 0 ? (p->pressure_iters + jg - 1) / jg : 1;  // no empty last group
  out->halo_prs = out->pressure_groups > 1 ? jg + 1 : 0;
  // vorticity confinement (FLUID_FLAG_VORTICITY) sits between diffusion and divergence and reads two rows further:
  // the force on row r needs the curl of rows r - 1 .. r + 1, the curl of row r the velocity of rows r - 1 .. r + 1
  const int vrt = (p->flags & FLUID_FLAG_VORTICITY) ? 2 : 0;
  out->halo_vel = jg + 1 + p->diffusion_iters + a + vrt;
  out->has_lo = p->slab_rank > 0;
  out->has_hi = p->slab_rank < p->slab_count - 1;
  // a halo must come from the adjacent slab alone, and the y-border rules read two rows inward
  const int min_rows = ny / p->slab_count;  // the shortest slab
  if (min_rows < out->halo_vel || min_rows < 4) return FLUID_E_UNSUPPORTED;
  const int lo = out->has_lo ? out->halo_vel : 0, hi = out->has_hi ? out->halo_vel : 0;
  out->store_row0 = out->row0 - lo;
  out->store_rows = out->rows + lo + hi;
  return FLUID_OK;
}

static int inner_init(fluid_ctx **out, const fluid_params &prm);

// Launch plan of the register pipelines by owned cells, from sweeps on MI355X; results do not depend
// on it (tests run seven plans), FLUID_TB_* / FLUID_TILE override it.  One row per size class, with
// the measurement the row came from.  (Up to 2^20 cells the LDS tiles run instead: init_ctx.)
struct PlanRow {
  long long below;   // owned cells < below
  int kj, kd;        // sweeps per pressure / diffusion launch (at most)
  int waves, waves_d;  // target waves per launch (diffusion: 0 = the same)
  float rounds;      // > 0: size launches to this many resident rounds of workgroups instead (tb_chunk)
  const char *measured;
};
static const PlanRow kPlan[] = {
    {9ll << 19, 8, 10, 2048, 0, 0.f,
     "up to 2048^2: 8 sweeps per pressure launch, half a resident round: longer chunks, fewer halo rows redone "
     "(tools/plan_sweep.py r1; 0.177 ms r2; r3 re-swept: within 1 % of the best of 48 plans from 1280^2 to 2048^2)"},
    {1ll << 23, 10, 10, 2048, 0, 0.5f,
     "r3 (profiles/r03/r3_sweep*.log): 10 sweeps and HALF a resident round of workgroups (counted per kernel, tb_chunk) "
     "instead of 8 sweeps and 2048 waves: 2304^2 0.245 -> 0.217 ms, 2560^2 0.250 -> 0.240, 2816^2 0.318 -> 0.293"},
    {1ll << 24, 14, 10, 2048, 2900, 1.f,
     "r3 (profiles/r03/r3_plan_*.log, r3_sweep2.log): 14 sweeps per pressure launch, 10 per diffusion launch and ONE "
     "resident round instead of 8 / 5 sweeps and 2048 / 2900 waves: 3072^2 0.347 -> 0.334 ms (J=200 0.900 -> 0.887), 3584^2 "
     "0.435 -> 0.400, a slab of 8 of 8192^2 0.343 -> 0.317 (J=200 0.914 -> 0.881), a slab of 2 of 4096^2 0.351 -> 0.322"},
    {1ll << 25, 14, 10, 3900, 2900, 1.f,
     "4096^2, a slab of 4 of 8192^2: ONE resident round (tools/waves_sweep2.py r2); r3: 14 sweeps per pressure launch "
     "instead of 10 (a launch fewer): 4096^2 0.531 -> 0.517 ms, J=200 1.330 -> 1.291, the slab 0.547 -> 0.529 (J=200 1.435 -> 1.351)"},
    {(1ll << 26) + 1, 14, 10, 6100, 0, 1.f,
     "6144^2 1.197 -> 1.085 ms, 8192^2 1.903 -> 1.826, 8192^2 J=200 4.83 -> 4.43 with one resident round (tools/waves_sweep2.py r2)"},
    {(1ll << 27) + 1, 14, 10, 6100, 0, 0.f, "a slab of 2 of 16384^2: 6100 waves, two rounds (tools/slab_time.py r1)"},
    {0x7fffffffffffffffll, 14, 10, 8192, 0, 0.f,
     "12288^2, 16384^2: r1 16384 waves (shorter chunks even out the end of a launch); r3, with the workgroups of the pressure launches "
     "kept in step: 8192 waves, 12288^2 3.896 -> 3.843 ms, 16384^2 6.560 -> 6.550 (profiles/r03/r3_waves_resweep.log)"},
};

// inner_of > 0: the context is one of that many inner slabs of a whole-domain context (inner_init)
static int init_ctx(fluid_ctx **out, const fluid_params *p, int inner_of) {
  if (!out) return FLUID_E_INVALID;
  *out = nullptr;
  fluid_params prm;
  if (p) prm = *p; else fluid_default_params(&prm);
  if (prm.n < 4 || prm.diffusion_iters < 0 || prm.pressure_iters < 0) return FLUID_E_INVALID;
  if (prm.ny != 0 && prm.ny < 4) return FLUID_E_INVALID;
  if (prm.inner_slabs < 0 || prm.reserved[0] || prm.reserved[1]) return FLUID_E_INVALID;
  fluid_slab_info si;
  int prc = fluid_slab_plan(&prm, &si);
  if (prc) return prc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FLUID_E_DEVICE;  // no CPU fallback
  if (prm.device < 0 || prm.device >= ndev) return FLUID_E_INVALID;
  if (prm.n_devices < 0 || prm.n_devices > 8) return FLUID_E_INVALID;
  if (prm.n_devices > 0 && !inner_of) {  // several GPUs under one context: slab g on device_ids[g]
    if (prm.inner_slabs > 1 || prm.slab_count > 1) return FLUID_E_INVALID;
    for (int g = 0; g < prm.n_devices; ++g)
      if (prm.device_ids[g] < 0 || prm.device_ids[g] >= ndev) return FLUID_E_INVALID;
    prm.device = prm.device_ids[0];
    if (prm.n_devices == 1) prm.n_devices = 0;  // one slab: the plain context on that device
    else return inner_init(out, prm);
  }
  if (prm.inner_slabs >= 2 && !inner_of) return inner_init(out, prm);
  fluid_ctx *c = new (std::nothrow) fluid_ctx();
  if (!c) return FLUID_E_NOMEM;
  c->prm = prm;
  c->inner_of = inner_of < 0 ? 0 : inner_of;  // < 0: a slab of a several-GPU context: it has its GPU to itself
  c->g = Geom{prm.n, si.store_row0, si.store_rows, prm.ny ? prm.ny : prm.n};
  c->row0 = si.row0;
  c->rows = si.rows;
  c->own_r0 = si.row0 - si.store_row0;
  c->slab = prm.slab_count > 1;
  c->halo_vel = si.halo_vel;
  c->halo_dens = si.halo_dens;
  c->groups = si.pressure_groups;
  c->halo_prs = si.halo_prs;
  c->jg = c->groups > 1 ? si.halo_prs - 1 : prm.pressure_iters;
  c->has_lo = si.has_lo;
  c->has_hi = si.has_hi;
  // Launch plan by owned cells: LDS tiles up to 2^20 (1024^2: launch/latency bound; one resident round of
  // 64 x 32 tiles ends there, 1088^2: 0.131 ms against 0.123 for the pipeline), then kPlan's rows.
  const long long owned = (long long)prm.n * si.rows;
  c->tile = env_int("FLUID_TILE", owned <= (1ll << 20) ? 1 : 0) != 0;
  c->tb_minchunk = 1;
  // 64 x 32 tiles of a 512^2 grid are 128 workgroups on 256 CUs, and a tile's 10 sweeps are a latency
  // chain of about 13 us whatever the grid: 16-row tiles there (512^2, J = 10: 33.4 -> 29.6 us per step,
  // J = 40: 57.9 -> 51.5; slower from 768^2 up, where 32-row tiles already give every CU work)
  c->tile_ty = env_int("FLUID_TILE_TY", owned <= 3 * (1ll << 17) ? 16 : 32);
  if (!tile_has_ty(c->tile_ty)) c->tile_ty = 32;
  c->tile_div = env_int("FLUID_TILE_DIV", 1) != 0;
  c->tb_div = env_int("FLUID_TB_DIV", 1) != 0;
  // (pressure: 14 + 13 + 13 sweeps for J = 40 instead of 4 x 10 — a launch fewer; the taller tiles cost
  // 16.4 us per launch instead of 13.4: 1024^2 81.6 -> 79.2 us per step, 768^2 74.9 -> 73.5, 512^2 J = 40
  // 45.4 -> 43.7.  Two launches of 20 on 104 x 72 tiles lose from 768^2 up: 30.2 us each, 1024^2 92.3.)
  if (c->tile) { c->tb_kj = 14; c->tb_kd = 10; }
  // Inner slabs share the GPU with their siblings: each launch is sized to its share of a resident
  // round, and with the chunks that long the deep pipelines pay (tools/inner_slabs_probe.py).
  else if (inner_of > 0 && owned < (1ll << 22)) { c->tb_kj = 10; c->tb_kd = 10; c->tb_waves = 2048 / inner_of; }
  else if (inner_of > 0) { c->tb_kj = 14; c->tb_kd = 10; c->tb_waves = 4096 / inner_of; c->tb_waves_d = 3072 / inner_of; }
  else {
    for (const PlanRow &r : kPlan) {
      if (owned >= r.below) continue;
      c->tb_kj = r.kj; c->tb_kd = r.kd; c->tb_waves = r.waves; c->tb_waves_d = r.waves_d; c->tb_rounds = r.rounds;
      break;
    }
    // the reference's own J = 10 (and anything up to it) in ONE pressure launch on the smallest pipeline
    // grids too, where J = 40 prefers five launches of 8: 2048^2 J=10 0.127 -> 0.116 ms, 1536^2 0.0847 -> 0.0808
    if (owned < (9ll << 19) && prm.pressure_iters <= 10) c->tb_kj = 10;
    // (the fallback wave target when the runtime cannot tell the occupancy: slabs of 16384^2)
    if (owned >= (1ll << 25) && owned <= (1ll << 26) && prm.n >= 12288) c->tb_waves = 3072;
  }
  // non-temporal stores in the two streaming kernels (fluid_kernels.hip).  Break-even re-measured in round 3
  // (profiles/r03/r3_nt*.log) between 2048^2 (0.1765 -> 0.1829 ms with them) and 2304^2 (0.2154 -> 0.2100):
  // on from 4.7 M stored cells (round 2: from 2^24; 2560^2 0.240 -> 0.233, 3072^2 0.332 -> 0.326, 3584^2
  // 0.400 -> 0.393; slabs of that size 0-3 % faster).
  // (inner slabs: by the whole domain's size, what the GPU's caches see — 4096^2 as two slabs: 0.525 -> 0.515 ms)
  // Tile order of the two gather kernels (tile_of_block in fluid_kernels.hip): runs of 128 consecutive tiles per
  // XCD instead of one (64 KB of a row per XCD at a time).  profiles/r03/r3_xcd_sweep*.log, us per launch,
  // plain order -> runs of 128 (runs of 32 .. 256 are within 2 % of each other; eight whole bands, one per XCD,
  // win as much up to 4096^2 and lose 6-11 % at 16384^2):
  //   k_advect_vel            2048^2 24.1 -> 18.5, 3072^2 45.1 -> 36.8, 4096^2 78.3 -> 58.9, 6144^2 150.6 -> 122.7,
  //                           8192^2 247.6 -> 210.1, 12288^2 517 -> 470, 16384^2 876 -> 807; nothing below 2048^2
  //   k_project_advect_dens   1280^2 18.0 -> 16.2, 1792^2 31.5 -> 26.4, 2048^2 36.1 -> 33.5, 3072^2 80.5 -> 77.1,
  //                           4096^2 144.4 -> 139.9; +-1 % from 6144^2 up (its own 16-byte texels dominate): off there
  {
    const long long stored = (long long)prm.n * si.store_rows * (inner_of > 0 ? inner_of : 1);
    c->xcd_run_adv = stored >= (7ll << 19) ? 128 : 0;
    c->xcd_run_tail = stored >= (3ll << 19) && prm.n <= 5120 ? 128 : 0;  // by width: a slab of a wide grid does not gain either (one of 8 of 8192^2, J = 200: 0.871 -> 0.882 ms with it)
  }
  // barriers in the row loops (fluid_kernels_tb.hip, Pipe::run): the divergence + pressure launch from 2^25 owned
  // cells, the pressure launches between 2^25 and 2^26 (6144^2 .. 8192^2); never the diffusion launch
  if (inner_of == 0 && owned >= (1ll << 25)) c->tb_sync = owned <= (1ll << 26) ? 3 : 2;
  // the first pressure launch from p = 0 (k_jacobi0_tb) holds four waves per SIMD where the other pressure launches
  // hold three; at 3072^2 .. 4096^2 it does better at three quarters of its resident round (chunks as long as
  // theirs): 4096^2 74.5 -> 70.7 us, 3072^2 52.2 -> 49.3; a full round from 6144^2 up (profiles/r03/r3_j0_share.log)
  if (inner_of == 0 && owned >= (1ll << 23) && owned <= (1ll << 24)) c->tb_j0_share = 0.75f;
  c->tb_j0_share = env_int("FLUID_TB_J0_PCT", (int)(c->tb_j0_share * 100.f + 0.5f)) / 100.f;
  c->tb_sync = env_int("FLUID_TB_SYNC", c->tb_sync);
  c->xcd_run_adv = env_int("FLUID_XCD_RUN_ADV", c->xcd_run_adv);
  c->xcd_run_tail = env_int("FLUID_XCD_RUN_TAIL", c->xcd_run_tail);
  c->nt_stores = env_int("FLUID_NT", (long long)prm.n * si.store_rows * (inner_of > 0 ? inner_of : 1) >= (9ll << 19) ? 1 : 0) != 0;
  c->graph_steps = env_int("FLUID_GRAPH_STEPS", 16);
  if (c->graph_steps < 1) c->graph_steps = 1;
  c->direct_parts = env_int("FLUID_DIRECT_PARTS", 1) != 0;
  c->force_k = env_int("FLUID_TEST_FORCE_K", 0);
  c->interior_split = env_int("FLUID_INTERIOR_SPLIT", 4);
  c->interior_split = c->interior_split < 1 ? 1 : c->interior_split > 8 ? 8 : c->interior_split;
  c->interior_piece_min = env_int("FLUID_INTERIOR_PIECE_MIN", 1 << 21);
  c->tb_kj = env_int("FLUID_TB_KJ", c->tb_kj);
  c->tb_kd = env_int("FLUID_TB_KD", c->tb_kd);
  c->tb_cj = env_int("FLUID_TB_CJ", c->tb_cj);
  c->tb_cd = env_int("FLUID_TB_CD", c->tb_cd);
  c->tb_waves = env_int("FLUID_TB_WAVES", c->tb_waves);
  c->tb_waves_d = env_int("FLUID_TB_WAVES_D", c->tb_waves_d);
  if (env_int("FLUID_TB_WAVES", 0) > 0) c->tb_rounds = 0.f;  // an explicit wave target wins over the table's rounds
  c->tb_rounds = env_int("FLUID_TB_ROUNDS_PCT", (int)(c->tb_rounds * 100.f + 0.5f)) / 100.f;
  c->tb_minchunk = env_int("FLUID_TB_MINCHUNK", c->tb_minchunk);
  c->tb_chain = env_int("FLUID_TB_CHAIN", c->tb_chain ? 1 : 0) != 0 && !c->tile;
  c->tb_chain_kinds = env_int("FLUID_TB_CHAIN_KINDS", c->tb_chain_kinds);
  c->alpha = 1.0f / (prm.viscosity * prm.dt);  // kernel.cu:182, fp32 at run time
  c->beta = 1.0f / (4.0f + c->alpha);          // kernel.cu:183
  int rc = [&]() -> int {
    HIPCHK(c, hipSetDevice(prm.device));
    fill_pass_tables(c);  // asks the runtime for occupancies: on this context's device, hence after hipSetDevice
    // this device's copy of the static scratch (fluid_step_noalloc).  A failed lookup (or a device
    // ordinal beyond the table) only disables that one entry point: it reports the missing address itself.
    (void)scratch_setup(prm.device);
    (void)hipGetLastError();
    HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    const size_t nc = cells_local(c);
    // One arena for the seven fields.  FLUID_ARENA_SKEW (bytes, experiments only) offsets each
    // field's base against the others; swept from 256 B to 3 MiB on MI355X (tools/skew_sweep.py)
    // it has no reproducible effect — what does move the 16384^2 step by about 3 % from one
    // context to the next is where the driver places the arena physically.
    {
      const size_t skew = (size_t)env_int("FLUID_ARENA_SKEW", 0) & ~(size_t)255;
      const size_t sizes[7] = {nc * sizeof(float2), nc * sizeof(float2), nc * sizeof(float4), nc * sizeof(float4),
                               nc * sizeof(float), nc * sizeof(float), nc * sizeof(float)};
      size_t off[7], total = 0;
      for (int k = 0; k < 7; ++k) {
        off[k] = total + k * skew;
        total += (sizes[k] + 255) & ~(size_t)255;
      }
      total += 7 * skew;
      HIPCHK(c, hipMalloc(&c->arena, total));
      // The reference zeroes only the current halves (kernel.cu:518-528); the
      // others are fully overwritten before first read.  Zero all: it is free.
      HIPCHK(c, hipMemsetAsync(c->arena, 0, total, c->stream));
      c->vel[0] = (float2 *)(c->arena + off[0]);
      c->vel[1] = (float2 *)(c->arena + off[1]);
      c->dens[0] = (float4 *)(c->arena + off[2]);
      c->dens[1] = (float4 *)(c->arena + off[3]);
      c->prs[0] = (float *)(c->arena + off[4]);
      c->prs[1] = (float *)(c->arena + off[5]);
      c->div = (float *)(c->arena + off[6]);
    }
    HIPCHK(c, hipMalloc(&c->dump, (size_t)prm.n * 16 + 256));
    HIPCHK(c, hipMalloc(&c->halo_flag, 256));
    HIPCHK(c, hipMemsetAsync(c->halo_flag, 0, 256, c->stream));
    if (c->tb_chain || env_int("FLUID_TB_TRACE", 0)) {
      tb_chain_capacity(c, &c->chain_words, &c->chain_flags);
      if (c->chain_words > 0) {
        const size_t side = (c->chain_words * 8 + 255) & ~(size_t)255, ctl = ((c->chain_flags + 64) * sizeof(int) + 255) & ~(size_t)255;
        const size_t trace = env_int("FLUID_TB_TRACE", 0) ? c->chain_flags * 4 * sizeof(unsigned long long) : 0;
        HIPCHK(c, hipMalloc(&c->chain_mem, side + ctl + trace));
        if (trace) {
          c->chain.trace = (unsigned long long *)((char *)c->chain_mem + side + ctl);
          HIPCHK(c, hipMemsetAsync(c->chain.trace, 0, trace, c->stream));
          c->trace_only.trace = c->chain.trace;
        }
        HIPCHK(c, hipMemsetAsync((char *)c->chain_mem + side, 0, ctl, c->stream));  // status word and flags
        c->chain.side = (unsigned long long *)c->chain_mem;
        c->chain.status = (int *)((char *)c->chain_mem + side);
        c->chain.flags = c->chain.status + 64;
      }
    }
    if (prm.flags & FLUID_FLAG_VORTICITY) {
      int r = ensure_vort(c);
      if (r) return r;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
  }();
  if (rc != FLUID_OK) {
    fluid_destroy(c);
    return rc;
  }
  *out = c;
  return FLUID_OK;
}

int fluid_init(fluid_ctx **out, const fluid_params *p) { return init_ctx(out, p, 0); }

void fluid_destroy(fluid_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->prm.device);
  for (fluid_ctx *s : c->subs) fluid_destroy(s);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (hipEvent_t e : c->ev_done) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  drop_graphs(c);
  if (c->has_comm && c->tr.destroy) c->tr.destroy(c->tr.user);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->arena) (void)hipFree(c->arena);
  if (c->vort) (void)hipFree(c->vort);
  if (c->halo_flag) (void)hipFree(c->halo_flag);
  if (c->dump) (void)hipFree(c->dump);
  if (c->chain_mem) (void)hipFree(c->chain_mem);
  if (c->diag_partial) (void)hipFree(c->diag_partial);
  if (c->cksum) (void)hipFree(c->cksum);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int fluid_reset(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_reset(c);
  c->advect_early = false;  // rows advected ahead of the step are stale now (recomputed by the full launch)
  c->halo_state = 0;        // and so are halos already exchanged or on their way (fluid_comm_*)
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t nc = cells_local(c);
  HIPCHK(c, hipMemsetAsync(c->vel[c->vin], 0, nc * sizeof(float2), c->stream));
  HIPCHK(c, hipMemsetAsync(c->prs[c->pin], 0, nc * sizeof(float), c->stream));
  HIPCHK(c, hipMemsetAsync(c->dens[c->din], 0, nc * sizeof(float4), c->stream));
  return FLUID_OK;
}

int fluid_add_source(fluid_ctx *c, float sx, float sy, float fx, float fy, const float rgba[4]) {
  if (!c || !rgba) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_add_source(c, sx, sy, fx, fy, rgba);
  c->advect_early = false;  // rows advected ahead of the step are stale now (recomputed by the full launch)
  c->halo_state = 0;
  HIPCHK(c, hipSetDevice(c->prm.device));
  ProfScope p(c, K_ADD_FORCE, 48.0 * (double)cells_owned(c));
  launch_add_force(c->stream, c->g, c->own_r0, c->own_r0 + c->rows, c->prm.dt, c->prm.splat_radius,
                   sx, sy, fx, fy, make_float4(rgba[0], rgba[1], rgba[2], rgba[3]), c->vel[c->vin],
                   c->dens[c->din]);
  return check_launch(c);
}

static int run_comm_steps(fluid_ctx *c, int nsteps);

int fluid_step(fluid_ctx *c, int nsteps) {
  if (!c || nsteps < 0) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_steps(c, nsteps, true);
  if (c->slab && c->has_comm) {  // the library exchanges the halos itself (fluid_comm_attach*)
    HIPCHK(c, hipSetDevice(c->prm.device));
    return run_comm_steps(c, nsteps);
  }
  if (c->slab && nsteps > 1) return FLUID_E_UNSUPPORTED;  // halos must be refreshed before every step
  if (c->groups > 1 && nsteps > 0) return FLUID_E_UNSUPPORTED;  // pressure halos between groups: fluid_step_part
  HIPCHK(c, hipSetDevice(c->prm.device));
  return run_steps(c, nsteps);
}

// fluid_step for callers that cannot allocate (include/fluid.h): direct launches (no graph is
// captured or instantiated), the pass plan from the tables fluid_init filled, and the dump row of
// the temporally blocked kernels in this GPU's static scratch instead of the context's hipMalloc'ed
// row.  Calls no hipMalloc / hipFree / allocating routine.  The dump row is write-only garbage
// (never read), so nothing has to be cleared between calls; it is shared by every call on the GPU.
int fluid_step_noalloc(fluid_ctx *c, int nsteps) {
  if (!c || nsteps < 0) return FLUID_E_INVALID;
  NOT_ON_INNER(c, "fluid_step_noalloc");
  if (c->slab) return FLUID_E_UNSUPPORTED;  // halos must be refreshed before every step (fluid_step_part)
  if (c->g.n > FLUID_NOALLOC_MAX_N)
    return fail(c, FLUID_E_SIZE, "fluid_step_noalloc: n = %d is beyond the static scratch (n <= %d, %zu bytes)",
                c->g.n, FLUID_NOALLOC_MAX_N, FLUID_NOALLOC_SCRATCH_BYTES);
  if ((size_t)c->g.n * 16 + 256 > FLUID_NOALLOC_SCRATCH_BYTES) return FLUID_E_SIZE;
  void *ptr = nullptr;  // set by fluid_init on this device
  if (c->prm.device >= 0 && c->prm.device < kMaxDevices) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ptr = g_scratch_ptr[c->prm.device];
  }
  if (!ptr) return fail(c, FLUID_E_DEVICE, "fluid_step_noalloc: no scratch address for device %d", c->prm.device);
  HIPCHK(c, hipSetDevice(c->prm.device));
  void *const own_dump = c->dump;
  c->dump = ptr;
  int rc = FLUID_OK;
  for (int s = 0; s < nsteps && rc == FLUID_OK; ++s) {
    enq_step(c, FLUID_PART_VELOCITY | FLUID_PART_DENSITY);
    rc = check_launch(c);
  }
  c->dump = own_dump;
  return rc;
}

// Test hook (include/fluid.h): lets a test see WHICH sink row the temporally blocked kernels were
// handed — the context's own hipMalloc'ed dump row (which = 0) or this GPU's static scratch
// (which = 1, its first n * 16 + 256 bytes).  fill: every 32-bit word := pattern, on the context's
// stream.  differing: after a synchronise, the number of words that are not `pattern`.
int fluid_debug_sink(fluid_ctx *c, int which, int fill, unsigned pattern, long long *differing) {
  if (!c || which < 0 || which > 1) return FLUID_E_INVALID;
  NOT_ON_INNER(c, "fluid_debug_sink");
  void *ptr = c->dump;
  if (which == 1) {
    ptr = nullptr;
    if (c->prm.device >= 0 && c->prm.device < kMaxDevices) {
      std::lock_guard<std::mutex> lock(g_scratch_mutex);
      ptr = g_scratch_ptr[c->prm.device];
    }
    if ((size_t)c->g.n * 16 + 256 > FLUID_NOALLOC_SCRATCH_BYTES) return FLUID_E_SIZE;
  }
  if (!ptr) return fail(c, FLUID_E_DEVICE, "fluid_debug_sink: no such row on device %d", c->prm.device);
  const size_t words = ((size_t)c->g.n * 16 + 256) / 4;
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (fill) HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)ptr, (int)pattern, words, c->stream));
  if (differing) {
    std::vector<unsigned> host(words);
    HIPCHK(c, hipMemcpyAsync(host.data(), ptr, words * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    long long d = 0;
    for (unsigned w : host) d += w != pattern;
    *differing = d;
  }
  return FLUID_OK;
}

// Diagnostics (include/fluid.h; FLUID_TB_TRACE=1 at fluid_init): four time stamps per wave of the last
// temporally blocked launch, record (chunk-row, strip) at out[4 * record]; *words = words written.
int fluid_debug_chain_trace(fluid_ctx *c, unsigned long long *out, size_t max_words, size_t *words) {
  if (!c || !out || !words) return FLUID_E_INVALID;
  if (!c->chain.trace) return FLUID_E_UNSUPPORTED;
  const size_t n = c->chain_flags * 4 < max_words ? c->chain_flags * 4 : max_words;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->chain.trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  *words = n;
  return FLUID_OK;
}

// Single stages are whole-domain only: in a slab the row range a stage must cover depends on
// what follows it in the step.
#define STAGE_FN(name, call, pre)                          \
  int name(fluid_ctx *c) {                                 \
    if (!c) return FLUID_E_INVALID;                        \
    if (c->slab || is_inner(c)) return FLUID_E_UNSUPPORTED; \
    HIPCHK(c, hipSetDevice(c->prm.device));                \
    pre;                                                   \
    call;                                                  \
    return check_launch(c);                                \
  }
STAGE_FN(fluid_advect_velocity, enq_advect_velocity(c, 0), (void)0)
STAGE_FN(fluid_diffuse, enq_diffuse(c, 0), (void)0)
STAGE_FN(fluid_vorticity, enq_vorticity(c), { int r_ = ensure_vort(c); if (r_) return r_; })
STAGE_FN(fluid_project, enq_project(c), (void)0)
STAGE_FN(fluid_advect_density, enq_advect_density(c), (void)0)

int fluid_read(fluid_ctx *c, int field, void *dst, size_t dst_bytes) {
  if (!c || !dst) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_read(c, field, dst, dst_bytes);
  size_t elem = 0;
  char *src = (char *)field_ptr(c, field, &elem);
  if (!elem) return FLUID_E_INVALID;
  const size_t bytes = cells_owned(c) * elem;
  if (dst_bytes < bytes) return FLUID_E_SIZE;
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (!src) {  // vorticity never computed: reads as zeros
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memset(dst, 0, bytes);
    return FLUID_OK;
  }
  src += (size_t)c->own_r0 * c->g.n * elem;
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_halo_flag(c);
}

int fluid_write(fluid_ctx *c, int field, const void *srcv, size_t src_bytes) {
  if (!c || !srcv) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_write(c, field, srcv, src_bytes);
  c->advect_early = false;  // rows advected ahead of the step are stale now (recomputed by the full launch)
  c->halo_state = 0;
  if (field == FLUID_VORTICITY) {
    int r = ensure_vort(c);
    if (r) return r;
  }
  size_t elem = 0;
  char *dst = (char *)field_ptr(c, field, &elem);
  if (!dst) return FLUID_E_INVALID;
  const size_t bytes = cells_owned(c) * elem;
  if (src_bytes != bytes) return FLUID_E_SIZE;
  HIPCHK(c, hipSetDevice(c->prm.device));
  dst += (size_t)c->own_r0 * c->g.n * elem;
  HIPCHK(c, hipMemcpyAsync(dst, srcv, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FLUID_OK;
}

int fluid_sync(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_sync(c);
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));  // halos posted for the next step
  return check_halo_flag(c);
}

int fluid_diagnostics(fluid_ctx *c, fluid_diag *out) {
  if (!c || !out) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_diagnostics(c, out);
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (!c->diag_partial) {
    const size_t rec_bytes = (size_t)DIAG_MAX_BLOCKS * DIAG_NREC * sizeof(float);
    HIPCHK(c, hipMalloc(&c->diag_partial, rec_bytes + sizeof(DiagResult)));
    c->diag_out = (DiagResult *)((char *)c->diag_partial + rec_bytes);
  }
  launch_diagnostics(c->stream, c->g, c->own_r0, c->own_r0 + c->rows, c->vel[c->vin], c->dens[c->din],
                     c->prs[c->pin], c->div, c->diag_partial, c->diag_out);
  int rc = check_launch(c);
  if (rc) return rc;
  DiagResult r;
  HIPCHK(c, hipMemcpyAsync(&r, c->diag_out, sizeof r, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  out->max_abs_u = r.mx[0];
  out->max_abs_v = r.mx[1];
  out->max_abs_div = r.mx[2];
  out->max_abs_p = r.mx[3];
  out->max_jacobi_update = r.mx[4];
  out->max_abs_div_velocity = r.mx[5];
  for (int k = 0; k < 4; ++k) out->dye[k] = r.sm[k];
  return check_halo_flag(c);
}

int fluid_checksum(fluid_ctx *c, int field, int row0, int rows, unsigned long long *out) {
  if (!c || !out || row0 < 0 || rows < 0) return FLUID_E_INVALID;
  NOT_ON_INNER(c, "fluid_checksum");
  if (rows == 0) { row0 = c->row0; rows = c->rows; }
  if (row0 < c->row0 || row0 + rows > c->row0 + c->rows) return FLUID_E_INVALID;
  size_t elem = 0;
  const char *src = (const char *)field_ptr(c, field, &elem);
  if (!elem || !src) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (!c->cksum) HIPCHK(c, hipMalloc(&c->cksum, 256));
  HIPCHK(c, hipMemsetAsync(c->cksum, 0, sizeof(unsigned long long), c->stream));
  src += (size_t)(row0 - c->row0 + c->own_r0) * c->g.n * elem;
  launch_checksum(c->stream, src, (size_t)rows * c->g.n * elem / 4, c->cksum);
  int rc = check_launch(c);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(out, c->cksum, sizeof *out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_halo_flag(c);
}

int fluid_geometry(const fluid_ctx *c, int *n, int *row0, int *rows) {
  if (!c) return FLUID_E_INVALID;
  if (n) *n = c->g.n;
  if (row0) *row0 = c->row0;
  if (rows) *rows = c->rows;
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

const char *fluid_last_error(const fluid_ctx *c) { return c ? c->err : ""; }

int fluid_prepare_steps(fluid_ctx *c, int nsteps) {
  if (!c || nsteps < 0) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_steps(c, nsteps, false);
  if (c->slab && nsteps > 1) return FLUID_E_UNSUPPORTED;
  if (c->groups > 1 && nsteps > 0) return FLUID_E_UNSUPPORTED;
  HIPCHK(c, hipSetDevice(c->prm.device));
  return run_steps(c, nsteps, FLUID_PART_VELOCITY | FLUID_PART_DENSITY, false);
}

int fluid_time_steps(fluid_ctx *c, int nsteps, float *ms_out) {
  if (!c || nsteps < 0 || !ms_out) return FLUID_E_INVALID;
  if (is_inner(c)) return inner_time_steps(c, nsteps, ms_out);
  if (c->slab && c->has_comm) return FLUID_E_UNSUPPORTED;  // time fluid_step from outside (two streams)
  if (c->slab && nsteps > 1) return FLUID_E_UNSUPPORTED;
  if (c->groups > 1 && nsteps > 0) return FLUID_E_UNSUPPORTED;  // as fluid_step: the exchanges between groups are the caller's
  HIPCHK(c, hipSetDevice(c->prm.device));
  {
    int prc = run_steps(c, nsteps, FLUID_PART_VELOCITY | FLUID_PART_DENSITY, false);  // graphs built outside the timed region
    if (prc) return prc;
  }
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  int rc = [&]() -> int {
    HIPCHK(c, hipEventRecord(e0, c->stream));
    int r = run_steps(c, nsteps);
    if (r) return r;
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    HIPCHK(c, hipEventElapsedTime(ms_out, e0, e1));
    return FLUID_OK;
  }();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

int fluid_profile_steps(fluid_ctx *c, int nsteps, fluid_kernel_stat *stats, int max_stats,
                        int *count) {
  if (!c || nsteps < 0 || !stats || !count || max_stats <= 0) return FLUID_E_INVALID;
  NOT_ON_INNER(c, "fluid_profile_steps (per-launch events; the slabs' launches overlap)");
  if (c->slab && c->has_comm) return FLUID_E_UNSUPPORTED;  // per-launch events on one stream; the step runs on two
  if (c->slab && nsteps > 1) return FLUID_E_UNSUPPORTED;
  if (c->groups > 1 && nsteps > 0) return FLUID_E_UNSUPPORTED;
  HIPCHK(c, hipSetDevice(c->prm.device));
  c->prof = true;
  c->ev_used = 0;
  c->recs.clear();
  int rc = run_steps(c, nsteps);
  c->prof = false;
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  fluid_kernel_stat acc[K_COUNT];
  memset(acc, 0, sizeof acc);
  for (const ProfRec &r : c->recs) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e1]));
    acc[r.kid].launches += 1;
    acc[r.kid].total_ms += ms;
    acc[r.kid].alg_bytes += r.bytes;
    acc[r.kid].moved_bytes += r.moved;
  }
  for (int i = 0; i < K_COUNT; ++i)
    if (acc[i].launches) {  // mean per launch
      acc[i].alg_bytes /= acc[i].launches;
      acc[i].moved_bytes /= acc[i].launches;
    }
  int k = 0;
  for (int i = 0; i < K_COUNT && k < max_stats; ++i) {
    if (!acc[i].launches) continue;
    snprintf(acc[i].name, sizeof acc[i].name, "%s", kKernelNames[i]);
    stats[k++] = acc[i];
  }
  *count = k;
  return FLUID_OK;
}

int fluid_step_part(fluid_ctx *c, int part) {
  if (!c) return FLUID_E_INVALID;
  NOT_ON_INNER(c, "fluid_step_part");
  switch (part) {
    case FLUID_PART_VELOCITY:
      if (c->groups > 1) return FLUID_E_UNSUPPORTED;
      break;
    case FLUID_PART_DENSITY:
      break;
    case FLUID_PART_VELOCITY_HEAD:
      c->pgroup = 0;
      break;
    case FLUID_PART_PRESSURE_GROUP:  // groups 1 .. groups-1, in order, after the head
      if (c->pgroup < 1 || c->pgroup >= c->groups) return fail(c, FLUID_E_INVALID, "no pressure group pending");
      break;
    case FLUID_PART_VELOCITY_TAIL:
    case FLUID_PART_TAIL_DENSITY_EDGES:
      if (c->pgroup != c->groups) return fail(c, FLUID_E_INVALID, "pressure groups not finished");
      break;
    case FLUID_PART_TAIL_DENSITY_INTERIOR:
      if (!c->tail_edges_done) return fail(c, FLUID_E_INVALID, "FLUID_PART_TAIL_DENSITY_EDGES must run first");
      break;
    case FLUID_PART_ADVECT_INTERIOR:  // between two steps of a slab context, once
      if (!c->slab) return FLUID_E_UNSUPPORTED;
      if (c->advect_early || c->tail_edges_done || (c->pgroup != 0 && c->pgroup != c->groups))
        return fail(c, FLUID_E_INVALID, "FLUID_PART_ADVECT_INTERIOR belongs between two steps");
      break;
    default:
      return FLUID_E_INVALID;
  }
  HIPCHK(c, hipSetDevice(c->prm.device));
  c->halo_state = 0;  // driven in parts: the halos are the caller's business from here on
  const int rc = run_steps(c, 1, part);
  if (rc) return rc;
  if (part == FLUID_PART_VELOCITY_HEAD) c->pgroup = 1;
  else if (part == FLUID_PART_PRESSURE_GROUP) ++c->pgroup;
  else if (part == FLUID_PART_VELOCITY_TAIL) c->pgroup = 0;
  else if (part == FLUID_PART_TAIL_DENSITY_EDGES) { c->pgroup = 0; c->tail_edges_done = true; }
  else if (part == FLUID_PART_TAIL_DENSITY_INTERIOR) c->tail_edges_done = false;
  return FLUID_OK;
}

// One field's halo block: which buffer, how many rows, bytes per cell.
struct HaloField { char *base; int rows; size_t esize; };
static bool halo_field(const fluid_ctx *c, int part, HaloField *f) {
  switch (part) {
    case FLUID_PART_VELOCITY: *f = HaloField{(char *)c->vel[c->vin], c->halo_vel, sizeof(float2)}; return true;
    case FLUID_PART_DENSITY: *f = HaloField{(char *)c->dens[c->din], c->halo_dens, sizeof(float4)}; return true;
    case FLUID_PART_PRESSURE_GROUP: *f = HaloField{(char *)c->prs[c->pin], c->halo_prs, sizeof(float)}; return c->halo_prs > 0;
    default: return false;
  }
}

size_t fluid_halo_bytes_part(const fluid_ctx *c, int dir, int parts) {
  if (!c || !c->slab) return 0;
  if ((dir < 0 && !c->has_lo) || (dir > 0 && !c->has_hi) || dir == 0) return 0;
  size_t b = 0;
  if (parts & FLUID_PART_VELOCITY) b += (size_t)c->g.n * c->halo_vel * sizeof(float2);
  if (parts & FLUID_PART_DENSITY) b += (size_t)c->g.n * c->halo_dens * sizeof(float4);
  if (parts & FLUID_PART_PRESSURE_GROUP) b += (size_t)c->g.n * c->halo_prs * sizeof(float);
  return b;
}

size_t fluid_halo_bytes(const fluid_ctx *c, int dir) {
  return fluid_halo_bytes_part(c, dir, FLUID_PART_VELOCITY | FLUID_PART_DENSITY);
}

// One field's halo block; stream NULL = the context's stream.
int fluid_halo_pack_part(fluid_ctx *c, int dir, int part, void *dst, void *stream) {
  HaloField f;
  if (!c || !dst || !halo_field(c, part, &f)) return FLUID_E_INVALID;
  if (!fluid_halo_bytes_part(c, dir, part)) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const size_t row = (size_t)c->g.n * f.esize;
  const size_t r = dir < 0 ? c->own_r0 : c->own_r0 + c->rows - f.rows;  // owned rows next to that neighbour
  launch_copy_block(st, dst, f.base + r * row, f.rows * row);
  return check_launch(c);
}

int fluid_halo_unpack_part(fluid_ctx *c, int dir, int part, const void *src, void *stream) {
  HaloField f;
  if (!c || !src || !halo_field(c, part, &f)) return FLUID_E_INVALID;
  if (!fluid_halo_bytes_part(c, dir, part)) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const size_t row = (size_t)c->g.n * f.esize;
  const size_t r = dir < 0 ? c->own_r0 - f.rows : c->own_r0 + c->rows;  // halo rows just outside the owned rows
  launch_copy_block(st, f.base + r * row, src, f.rows * row);
  return check_launch(c);
}

// Message layout: [halo_vel rows of float2][halo_dens rows of float4], rows in ascending y.
int fluid_halo_pack(fluid_ctx *c, int dir, void *dst) {
  if (!c || !dst) return FLUID_E_INVALID;
  if (!fluid_halo_bytes(c, dir)) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t n = c->g.n;
  const size_t vb = (size_t)c->halo_vel * n * sizeof(float2), db = (size_t)c->halo_dens * n * sizeof(float4);
  // the owned rows adjacent to that neighbour
  const size_t rv = dir < 0 ? c->own_r0 : c->own_r0 + c->rows - c->halo_vel;
  const size_t rd = dir < 0 ? c->own_r0 : c->own_r0 + c->rows - c->halo_dens;
  launch_copy_block(c->stream, dst, c->vel[c->vin] + rv * n, vb);
  launch_copy_block(c->stream, (char *)dst + vb, c->dens[c->din] + rd * n, db);
  return check_launch(c);
}

int fluid_halo_unpack(fluid_ctx *c, int dir, const void *src) {
  if (!c || !src) return FLUID_E_INVALID;
  if (!fluid_halo_bytes(c, dir)) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  const size_t n = c->g.n;
  const size_t vb = (size_t)c->halo_vel * n * sizeof(float2), db = (size_t)c->halo_dens * n * sizeof(float4);
  // the halo rows just outside the owned rows on that side
  const size_t rv = dir < 0 ? c->own_r0 - c->halo_vel : c->own_r0 + c->rows;
  const size_t rd = dir < 0 ? c->own_r0 - c->halo_dens : c->own_r0 + c->rows;
  launch_copy_block(c->stream, c->vel[c->vin] + rv * n, src, vb);
  launch_copy_block(c->stream, c->dens[c->din] + rd * n, (const char *)src + vb, db);
  return check_launch(c);
}


// ---- halo exchange under the C ABI ------------------------------------------------------------
// No counterpart in the reference (single GPU).  With a transport attached, fluid_step on a slab
// context is the whole multi-GPU step: it posts the halo exchange itself (one group per step:
// velocity and density blocks to and from both neighbours), straight out of and into the field
// rows (a halo block is whole rows, hence contiguous: no staging copies), on a second stream, and
// overlaps it with the interior launches exactly as fluid_step_part lets a caller do by hand:
//
//   [interior rows of the advection]  wait for the halos  advection of the rest, diffusion,
//   divergence + pressure sweeps  gradient subtraction + density advection on the EDGE rows
//   -> post the exchange for the next step ->  the same on the INTERIOR rows
//
// halo_state: HALO_NONE    no exchange of the current state has been posted (fresh context, or
//                          fluid_add_source / fluid_reset / fluid_write changed the owned rows)
//             HALO_POSTED  posted on comm_stream, ev_halo recorded behind it
//             HALO_JOINED  the compute stream has waited for it: the halo rows are in place
// Every rank must make the same calls in the same order (SPMD), as with any collective.
enum { HALO_NONE = 0, HALO_POSTED = 1, HALO_JOINED = 2 };

static int post_exchange(fluid_ctx *c, int parts) {
  Transport &T = c->tr;
  HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));           // everything enqueued so far ...
  HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));  // ... precedes the transfers
  int rc = T.group_start(T.user);
  if (rc) return fail(c, FLUID_E_COMM, "%s group start: %s", T.name, T.error ? T.error(T.user) : "");
  for (int dir = -1; dir <= 1; dir += 2) {
    if (dir < 0 ? !c->has_lo : !c->has_hi) continue;
    const int peer = c->prm.slab_rank + dir;
    for (int part : {FLUID_PART_VELOCITY, FLUID_PART_DENSITY, FLUID_PART_PRESSURE_GROUP}) {
      HaloField f;
      if (!(parts & part) || !halo_field(c, part, &f) || f.rows <= 0) continue;
      const size_t row = (size_t)c->g.n * f.esize, bytes = f.rows * row;
      const size_t rs = dir < 0 ? c->own_r0 : c->own_r0 + c->rows - f.rows;  // owned rows next to that neighbour
      const size_t rr = dir < 0 ? c->own_r0 - f.rows : c->own_r0 + c->rows;  // halo rows just outside the owned rows
      if (!rc) rc = T.send(T.user, f.base + rs * row, bytes, peer, c->comm_stream);
      if (!rc) rc = T.recv(T.user, f.base + rr * row, bytes, peer, c->comm_stream);
      c->comm_bytes += (long long)bytes;
    }
  }
  const int rc2 = T.group_end(T.user, c->comm_stream);
  if (rc || rc2) return fail(c, FLUID_E_COMM, "%s exchange: %s", T.name, T.error ? T.error(T.user) : "");
  HIPCHK(c, hipEventRecord(c->ev_halo, c->comm_stream));
  ++c->comm_exchanges;
  return FLUID_OK;
}

static int wait_exchange(fluid_ctx *c) {
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  return FLUID_OK;
}

// One step, enqueue only: the same code runs eagerly and under stream capture.
static int comm_step(fluid_ctx *c) {
  const int VD = FLUID_PART_VELOCITY | FLUID_PART_DENSITY;
  const bool split = !(c->prm.flags & FLUID_FLAG_UNFUSED);  // the fused last kernel in two parts
  int rc;
  if (c->halo_state == HALO_NONE) {
    if ((rc = post_exchange(c, VD))) return rc;
    c->halo_state = HALO_POSTED;
  }
  if (c->halo_state == HALO_POSTED) {
    // the interior rows of this step's advection read no halo row: more cover for the message
    if (c->early_advect && !c->advect_early) enq_advect_interior(c);
    if ((rc = wait_exchange(c))) return rc;
    c->halo_state = HALO_JOINED;
  }
  enq_velocity_head(c);
  for (int g = 1; g < c->groups; ++g) {  // deep pressure loops on short slabs: exposed exchanges between groups
    if ((rc = post_exchange(c, FLUID_PART_PRESSURE_GROUP))) return rc;
    if ((rc = wait_exchange(c))) return rc;
    enq_pressure_group(c, g);
  }
  if (split) {
    enq_gradient_density_rows(c, true);   // the rows the neighbours wait for; flips to the new buffers
    if ((rc = post_exchange(c, VD))) return rc;  // they travel while the interior rows are computed
    c->halo_state = HALO_POSTED;
    enq_gradient_density_rows(c, false);
  } else {
    enq_gradient(c, false);
    enq_advect_density(c);
    c->halo_state = HALO_NONE;
  }
  c->pgroup = 0;
  c->tail_edges_done = false;
  return check_launch(c);
}

static int run_comm_steps(fluid_ctx *c, int nsteps) {
  for (int s = 0; s < nsteps;) {
    const bool graph = c->comm_graph && c->tr.capturable && !(c->prm.flags & FLUID_FLAG_NO_GRAPH) && !c->prof;
    if (!graph) {
      const int rc = comm_step(c);
      if (rc) return rc;
      ++s;
      continue;
    }
    // A graph starts and ends with the halos in place (an event recorded outside a capture cannot
    // be waited for inside it): bring the state there eagerly.  This also makes the transport's
    // first exchange (connection set-up, lazy allocations) happen outside any capture.
    int rc;
    if (c->halo_state == HALO_NONE) {
      if ((rc = post_exchange(c, FLUID_PART_VELOCITY | FLUID_PART_DENSITY))) return rc;
      c->halo_state = HALO_POSTED;
    }
    if (c->halo_state == HALO_POSTED) {
      if ((rc = wait_exchange(c))) return rc;
      c->halo_state = HALO_JOINED;
    }
    int batch = 1;
    while (batch * 2 <= c->graph_steps && batch * 2 <= nsteps - s) batch *= 2;
    const int key = state_of(c) | (c->advect_early ? 8 : 0) | (batch << 8);
    auto it = c->comm_graphs.find(key);
    if (it == c->comm_graphs.end()) {
      GraphEntry ge;
      const int state0 = state_of(c);
      const bool early0 = c->advect_early;
      const long long ex0 = c->comm_exchanges, by0 = c->comm_bytes;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
      rc = FLUID_OK;
      for (int b = 0; b < batch && !rc; ++b) rc = comm_step(c);
      if (!rc && c->halo_state == HALO_POSTED) rc = wait_exchange(c);  // joins comm_stream back into the capture
      const hipError_t e = hipStreamEndCapture(c->stream, &ge.graph);
      ge.next_state = state_of(c);
      ge.halo_after = c->halo_state;  // joined (split last kernel) or none (FLUID_FLAG_UNFUSED: no exchange posted yet)
      ge.exchanges = c->comm_exchanges - ex0;
      ge.bytes = c->comm_bytes - by0;
      set_state(c, state0);
      c->advect_early = early0;
      c->halo_state = HALO_JOINED;
      c->comm_exchanges = ex0;
      c->comm_bytes = by0;
      c->enq_rc = 0;
      hipError_t ei = hipSuccess;
      if (!rc && e == hipSuccess) ei = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
      if (rc || e != hipSuccess || ei != hipSuccess) {
        // this transport's calls (or this runtime) do not capture: launch everything eagerly from
        // now on — nothing of the failed capture has run
        snprintf(c->comm_note, sizeof c->comm_note, "graph capture given up (%s): %.100s",
                 rc ? "enqueue" : e != hipSuccess ? hipGetErrorString(e) : hipGetErrorString(ei), rc ? c->err : "");
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        (void)hipGetLastError();
        c->comm_graph = false;
        continue;
      }
      it = c->comm_graphs.emplace(key, ge).first;
    }
    HIPCHK(c, hipGraphLaunch(it->second.exec, c->stream));
    set_state(c, it->second.next_state);
    c->advect_early = false;
    c->halo_state = it->second.halo_after;
    c->comm_exchanges += it->second.exchanges;
    c->comm_bytes += it->second.bytes;
    s += batch;
  }
  return FLUID_OK;
}

static int comm_attach_common(fluid_ctx *c, const Transport &t) {
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (!c->comm_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  if (!c->ev_ready) HIPCHK(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  if (!c->ev_halo) HIPCHK(c, hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  c->tr = t;
  c->has_comm = true;
  c->halo_state = HALO_NONE;
  // Whole slab steps (both streams + the transport's calls) as hipGraphs: opt-in (FLUID_COMM_GRAPH=1)
  // until a run with two or more RCCL ranks exists — a capture that fails is noticed and falls back,
  // a replay that hangs or moves wrong data on real links would not be.  bench.py --gpus N
  // asks for it explicitly, verifies a replayed batch against the whole-domain run and falls back itself.
  c->comm_graph = env_int("FLUID_COMM_GRAPH", 0) != 0;
  c->early_advect = env_int("FLUID_SLAB_EARLY_ADVECT", 1) != 0;
  c->comm_note[0] = 0;
  return FLUID_OK;
}

static int comm_precheck(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  if (!c->slab) return fail(c, FLUID_E_UNSUPPORTED, "a whole-domain context has no halos to exchange");
  if (c->has_comm) return fail(c, FLUID_E_INVALID, "a transport is already attached (fluid_comm_detach first)");
  return FLUID_OK;
}

int fluid_comm_unique_id(void *id_out, size_t id_bytes) {
  char err[160];
  if (!id_out) return FLUID_E_INVALID;
  return rccl_unique_id(id_out, id_bytes, err, sizeof err) ? FLUID_OK : FLUID_E_COMM;
}

int fluid_comm_init_rccl(fluid_ctx *c, const void *id, size_t id_bytes) {
  int rc = comm_precheck(c);
  if (rc) return rc;
  if (!id) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  Transport t;
  char err[200];
  if (!rccl_transport_init(id, id_bytes, c->prm.slab_rank, c->prm.slab_count, &t, err, sizeof err))
    return fail(c, FLUID_E_COMM, "%s", err);
  rc = comm_attach_common(c, t);
  if (rc && t.destroy) t.destroy(t.user);
  return rc;
}

int fluid_comm_attach_rccl(fluid_ctx *c, void *nccl_comm) {
  int rc = comm_precheck(c);
  if (rc) return rc;
  if (!nccl_comm) return FLUID_E_INVALID;
  Transport t;
  char err[200];
  if (!rccl_transport_from_comm(nccl_comm, c->prm.slab_rank, c->prm.slab_count, &t, err, sizeof err))
    return fail(c, FLUID_E_COMM, "%s", err);
  rc = comm_attach_common(c, t);
  if (rc && t.destroy) t.destroy(t.user);
  return rc;
}

int fluid_comm_attach(fluid_ctx *c, const fluid_transport *ft) {
  int rc = comm_precheck(c);
  if (rc) return rc;
  if (!ft || !ft->group_start || !ft->send || !ft->recv || !ft->group_end) return FLUID_E_INVALID;
  Transport t;
  t.user = ft->user;
  t.group_start = ft->group_start;
  t.send = ft->send;
  t.recv = ft->recv;
  t.group_end = ft->group_end;
  t.capturable = ft->capturable != 0;
  t.name = "caller";
  return comm_attach_common(c, t);
}

int fluid_comm_detach(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  if (!c->has_comm) return FLUID_OK;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->comm_stream));
  drop_graphs(c);
  if (c->tr.destroy) c->tr.destroy(c->tr.user);
  c->tr = Transport();
  c->has_comm = false;
  c->halo_state = HALO_NONE;
  return FLUID_OK;
}

// Halo rows of the current state in place when the stream gets here (diagnostics across the
// seams, a caller's own kernels): posts the exchange unless one is already posted, and waits.
int fluid_comm_refresh(fluid_ctx *c) {
  if (!c) return FLUID_E_INVALID;
  if (!c->slab || !c->has_comm) return FLUID_E_UNSUPPORTED;
  HIPCHK(c, hipSetDevice(c->prm.device));
  int rc;
  if (c->halo_state == HALO_NONE) {
    if ((rc = post_exchange(c, FLUID_PART_VELOCITY | FLUID_PART_DENSITY))) return rc;
    c->halo_state = HALO_POSTED;
  }
  if (c->halo_state == HALO_POSTED) {
    if ((rc = wait_exchange(c))) return rc;
    c->halo_state = HALO_JOINED;
  }
  return FLUID_OK;
}

int fluid_comm_selftest_rccl(int device, size_t bytes, int use_graph, char *msg, size_t msg_bytes) {
  char local[256];
  if (!msg || !msg_bytes) { msg = local; msg_bytes = sizeof local; }
  if (!bytes) return FLUID_E_INVALID;
  return rccl_selftest(device, bytes, use_graph != 0, msg, msg_bytes) ? FLUID_OK : FLUID_E_COMM;
}

int fluid_comm_info(const fluid_ctx *c, fluid_comm_stats *out) {
  if (!c || !out) return FLUID_E_INVALID;
  memset(out, 0, sizeof *out);
  if (is_inner(c)) {  // no transport: the slabs' halos are device copies; what it reports is the graph path
    out->graph = c->comm_graph && !(c->prm.flags & FLUID_FLAG_NO_GRAPH);
    out->graphs_built = (int)c->graphs.size();
    snprintf(out->transport, sizeof out->transport, "inner");
    snprintf(out->note, sizeof out->note, "%s", c->comm_note);
    return FLUID_OK;
  }
  out->attached = c->has_comm;
  out->capturable = c->has_comm && c->tr.capturable;
  out->graph = c->has_comm && c->comm_graph && c->tr.capturable && !(c->prm.flags & FLUID_FLAG_NO_GRAPH);
  out->graphs_built = (int)c->comm_graphs.size();
  out->exchanges = c->comm_exchanges;
  out->bytes_sent = c->comm_bytes;
  snprintf(out->transport, sizeof out->transport, "%s", c->has_comm ? c->tr.name : "");
  snprintf(out->note, sizeof out->note, "%s", c->comm_note);
  out->ranks = c->has_comm && c->tr.ranks ? c->tr.ranks(c->tr.user) : 0;
  return FLUID_OK;
}

void *fluid_stream(fluid_ctx *c) { return c ? (void *)c->stream : nullptr; }

int fluid_set_stream(fluid_ctx *c, void *hip_stream) {
  if (!c) return FLUID_E_INVALID;
  HIPCHK(c, hipSetDevice(c->prm.device));
  if (is_inner(c)) {
    const int rc = inner_sync(c);
    if (rc) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drop_graphs(c);
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return FLUID_OK;
}

// ---- inner slabs: one GPU's domain as G slabs side by side ---------------------------------------
// No counterpart in the reference.  fluid_params.inner_slabs = G >= 2 makes fluid_init build G slab
// contexts of the same GPU (the ordinary ones: own stream, own fields with halos, the step's
// dependency cone recomputed in the halo) under a parent that owns no field.  The slabs of a step
// are independent of each other, so their launches overlap on the GPU and fill each other's ramp-up
// and drain (include/fluid.h).  Halo rows are PUSHED: each slab copies its own edge
// rows into its neighbours' halo rows, on its own stream, so the source needs no cross-stream
// ordering at all.  Per step and slab:
//
//   wait for my neighbours' pushes (ev_halo)   advection, diffusion -> ev_ready ("my halo rows of
//   the buffers this step ends in are no longer read")   divergence + pressure sweeps   last kernel
//   on the EDGE rows   wait for my neighbours' ev_ready, push my edge rows -> ev_halo   last kernel
//   on the INTERIOR rows
//
// Why the destination is free after the diffusion: the last kernel writes the buffers the step ends
// in; the velocity one of them was the input of the launch before it in the velocity chain (the last
// diffusion launch, or the advection with D = 0), which read its halo rows — nothing later does; the
// density one was last read a step ago.  Enqueue order on the host: every slab up to its edge rows,
// then every slab's pushes and interior (all the events waited for are recorded by then).  A slab may
// run ahead of its neighbour by the length of the interior launch and lag behind it by most of a step.
static int inner_fail(fluid_ctx *P, const fluid_ctx *s, int rc) {
  if (rc) snprintf(P->err, sizeof P->err, "inner slab %d: %.200s", s->prm.slab_rank, s->err);
  return rc;
}

static int inner_init(fluid_ctx **out, const fluid_params &prm) {
  const bool multi = prm.n_devices >= 2;  // the slabs on device_ids[] instead of side by side on prm.device
  const int G = multi ? prm.n_devices : prm.inner_slabs;
  if (prm.slab_count > 1 || G > 16) return FLUID_E_UNSUPPORTED;  // no slabs of slabs
  if (prm.flags & FLUID_FLAG_UNFUSED) return FLUID_E_UNSUPPORTED;
  fluid_ctx *P = new (std::nothrow) fluid_ctx();
  if (!P) return FLUID_E_NOMEM;
  P->prm = prm;
  const int ny = prm.ny ? prm.ny : prm.n;
  P->g = Geom{prm.n, 0, ny, ny};
  P->rows = ny;
  const int rc = [&]() -> int {
    HIPCHK(P, hipSetDevice(prm.device));
    HIPCHK(P, hipStreamCreateWithFlags(&P->own_stream, hipStreamNonBlocking));
    P->stream = P->own_stream;
    P->graph_steps = env_int("FLUID_GRAPH_STEPS", 16);
    if (P->graph_steps < 1) P->graph_steps = 1;
    P->comm_graph = env_int("FLUID_INNER_GRAPH", 1) != 0;  // whole steps of all slabs as one hipGraph (inner_steps)
    {
      // hipStreamEndCapture of this fork (G sibling streams of real launches) segfaults in the HIP
      // runtime 7.0.51831 that PyTorch-ROCm 2.10 bundles and loads first in a Python process, and
      // works in ROCm 7.2's own (7.2.26015) — same library, same calls (tools/inner_dbg.py with and
      // without torch, gpurun_out/r3_inner_dbg*.log).  Older runtimes get direct launches.
      int rv = 0;
      if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
      if (P->comm_graph && rv < 70200000 && !env_int("FLUID_INNER_GRAPH_FORCE", 0)) {
        P->comm_graph = false;
        snprintf(P->comm_note, sizeof P->comm_note, "inner slabs: direct launches (HIP runtime %d < 7.2 crashes in hipStreamEndCapture)", rv);
      }
    }
    HIPCHK(P, hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
    bool one_device = true;
    for (int r = 0; r < G; ++r) {
      fluid_params sp = prm;
      sp.inner_slabs = 0;
      sp.n_devices = 0;
      if (multi) sp.device = prm.device_ids[r];
      one_device = one_device && sp.device == prm.device;
      sp.slab_rank = r;
      sp.slab_count = G;
      sp.pressure_groups = 0;
      fluid_ctx *s = nullptr;
      // slabs that share a GPU size their launches to their share of it; alone on its GPU a slab is a rank
      int share = G;
      if (multi) {
        share = 0;
        for (int q = 0; q < G; ++q) share += prm.device_ids[q] == prm.device_ids[r];
      }
      const int src = init_ctx(&s, &sp, share > 1 ? share : -1);
      if (src) return src;  // e.g. FLUID_E_UNSUPPORTED: a slab too short for its halo
      P->subs.push_back(s);
      HIPCHK(P, hipSetDevice(sp.device));  // the events a slab's stream records live on its device
      hipEvent_t e = nullptr;
      HIPCHK(P, hipEventCreateWithFlags(&e, hipEventDisableTiming));
      P->ev_done.push_back(e);
      HIPCHK(P, hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));  // "my halo rows may be written"
      HIPCHK(P, hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));   // "my pushes have landed"
      s->interior_split = env_int("FLUID_INTERIOR_SPLIT", 1);  // nothing slow has to get in between: one launch
      s->interior_split = s->interior_split < 1 ? 1 : s->interior_split > 8 ? 8 : s->interior_split;
    }
    P->peer_copies = !one_device || env_int("FLUID_FORCE_PEER_COPY", 0) != 0;
    if (!one_device) {
      // no hipGraph across devices (not tried: nothing here has more than one GPU); peer access lets
      // the copies go GPU to GPU over xGMI, without it the runtime stages them — either way correct
      P->comm_graph = false;
      snprintf(P->comm_note, sizeof P->comm_note, "slabs on several devices: direct launches");
      for (int r = 0; r + 1 < G; ++r) {
        const int a = P->subs[r]->prm.device, b = P->subs[r + 1]->prm.device;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }
        if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can) { (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0); }
        (void)hipGetLastError();  // "already enabled" is fine
      }
    }
    HIPCHK(P, hipSetDevice(prm.device));
    return FLUID_OK;
  }();
  if (rc != FLUID_OK) {
    fluid_destroy(P);
    return rc;
  }
  *out = P;
  return FLUID_OK;
}

// Whatever is enqueued on a slab's stream is enqueued with that slab's device current (the slabs of a
// several-GPU context live on different devices; a no-op when they share one).
#define ON_DEVICE_OF(P, s) HIPCHK((P), hipSetDevice((s)->prm.device))

// The slabs' streams pick up after this context's stream ...
static int inner_fork(fluid_ctx *P) {
  HIPCHK(P, hipSetDevice(P->prm.device));
  HIPCHK(P, hipEventRecord(P->ev_fork, P->stream));
  for (fluid_ctx *s : P->subs) {
    ON_DEVICE_OF(P, s);
    HIPCHK(P, hipStreamWaitEvent(s->stream, P->ev_fork, 0));
  }
  return FLUID_OK;
}
// ... and it picks up after them (pushes included: they are on the slabs' streams).
static int inner_join(fluid_ctx *P) {
  for (size_t i = 0; i < P->subs.size(); ++i) {
    ON_DEVICE_OF(P, P->subs[i]);
    HIPCHK(P, hipEventRecord(P->ev_done[i], P->subs[i]->stream));
  }
  HIPCHK(P, hipSetDevice(P->prm.device));
  for (size_t i = 0; i < P->subs.size(); ++i) HIPCHK(P, hipStreamWaitEvent(P->stream, P->ev_done[i], 0));
  return FLUID_OK;
}

// Slab i's edge rows of the current velocity and density into its neighbours' halo rows, on its own
// stream, once the neighbour no longer reads them (its ev_ready, recorded for the state in question).
static int inner_push(fluid_ctx *P, size_t i) {
  fluid_ctx *s = P->subs[i];
  ON_DEVICE_OF(P, s);
  for (int dir = -1; dir <= 1; dir += 2) {
    if (dir < 0 ? !s->has_lo : !s->has_hi) continue;
    fluid_ctx *q = P->subs[i + dir];
    HIPCHK(P, hipStreamWaitEvent(s->stream, q->ev_ready, 0));
    for (int part : {FLUID_PART_VELOCITY, FLUID_PART_DENSITY}) {
      HaloField fs, fq;
      if (!halo_field(s, part, &fs) || !halo_field(q, part, &fq) || fq.rows <= 0) continue;
      const size_t row = (size_t)s->g.n * fs.esize, bytes = fq.rows * row;
      const size_t rs = dir < 0 ? s->own_r0 : s->own_r0 + s->rows - fq.rows;  // my owned rows next to that neighbour
      const size_t rd = dir < 0 ? q->own_r0 + q->rows : q->own_r0 - fq.rows;  // its halo rows on my side
      // (the neighbour on another GPU: a device-to-device copy, peer to peer over xGMI where the two
      // can reach each other; on the same GPU the copy kernel, which gets onto a busy device sooner)
      if (P->peer_copies) HIPCHK(P, hipMemcpyAsync(fq.base + rd * row, fs.base + rs * row, bytes, hipMemcpyDeviceToDevice, s->stream));
      else launch_copy_block(s->stream, fq.base + rd * row, fs.base + rs * row, bytes);
    }
  }
  HIPCHK(P, hipEventRecord(s->ev_halo, s->stream));
  return FLUID_OK;
}

// Halo rows for the current state, unless the pushes are already enqueued.
static int inner_refresh(fluid_ctx *P) {
  if (P->inner_posted) return FLUID_OK;
  for (fluid_ctx *s : P->subs) {
    ON_DEVICE_OF(P, s);
    HIPCHK(P, hipEventRecord(s->ev_ready, s->stream));
  }
  for (size_t i = 0; i < P->subs.size(); ++i) {
    const int rc = inner_push(P, i);
    if (rc) return rc;
  }
  P->inner_posted = true;
  return FLUID_OK;
}

// A slab's stream waits for its neighbours' pushes into its halo rows.
static int inner_wait_halos(fluid_ctx *P, size_t i) {
  fluid_ctx *s = P->subs[i];
  ON_DEVICE_OF(P, s);
  if (s->has_lo) HIPCHK(P, hipStreamWaitEvent(s->stream, P->subs[i - 1]->ev_halo, 0));
  if (s->has_hi) HIPCHK(P, hipStreamWaitEvent(s->stream, P->subs[i + 1]->ev_halo, 0));
  return FLUID_OK;
}

// nsteps steps of every slab, enqueue only (between inner_fork and inner_join)
// halos_joined: the pushes for the current state are not only enqueued but already joined into this
// context's stream (every inner_* call ends with inner_join), so the fork orders the first step
// after them and its wait on the neighbours' ev_halo is skipped — under stream capture that wait
// would name events recorded outside the capture, which the runtime refuses ("dependency created on
// uncaptured work in another stream").
static int inner_enqueue(fluid_ctx *P, int nsteps, bool halos_joined = false) {
  const size_t G = P->subs.size();
  int rc;
  for (int k = 0; k < nsteps; ++k) {
    if ((rc = inner_refresh(P))) return rc;
    for (size_t i = 0; i < G; ++i) {
      fluid_ctx *s = P->subs[i];
      ON_DEVICE_OF(P, s);
      if (!(halos_joined && k == 0) && (rc = inner_wait_halos(P, i))) return rc;
      // enq_velocity_head in two pieces, with the event between them
      const int m_diff = s->prm.pressure_iters + 1;
      const bool vrt = (s->prm.flags & FLUID_FLAG_VORTICITY) != 0;
      const int m_vrt = vrt ? 2 : 0;
      enq_advect_velocity(s, m_diff + m_vrt + s->prm.diffusion_iters);
      enq_diffuse(s, m_diff + m_vrt);
      if (vrt) enq_vorticity(s, m_diff);  // in place on the buffer the diffusion ended in
      HIPCHK(P, hipEventRecord(s->ev_ready, s->stream));
      enq_pressure_group(s, 0);
      enq_gradient_density_rows(s, true);  // the rows the neighbours wait for; flips to the new buffers
    }
    for (size_t i = 0; i < G; ++i) {
      fluid_ctx *s = P->subs[i];
      if ((rc = inner_push(P, i))) return rc;  // (leaves slab i's device current)
      enq_gradient_density_rows(s, false);
      s->pgroup = 0;
      s->tail_edges_done = false;
      if ((rc = check_launch(s))) return inner_fail(P, s, rc);
    }
  }
  return FLUID_OK;
}

// Whole steps of every slab as ONE hipGraph per batch (power-of-two batches as in run_steps): the
// origin is this context's stream, the fork pulls the G sibling streams into the capture, the join
// brings them back.  A graph begins with the halo pushes of its start state enqueued AND joined into
// the origin (so nothing inside it waits for an event recorded outside it, see inner_enqueue) and
// ends the same way.  Round 2 recorded a host segfault in hipStreamEndCapture for a first attempt at
// this; tools/ubench/capture_fork.hip replays the stream/event shape with empty kernels: forks,
// sibling-to-sibling waits and events re-recorded across 16 steps all capture and replay, and a wait
// on an event recorded outside the capture is REFUSED by hipStreamWaitEvent (that attempt had such
// waits: the previous call's pushes).  launch == false: only build the graphs.
static int inner_direct(fluid_ctx *P, int nsteps) {
  int rc = inner_fork(P);
  if (rc) return rc;
  if ((rc = inner_enqueue(P, nsteps))) return rc;
  return inner_join(P);
}

static int inner_steps(fluid_ctx *P, int nsteps, bool launch) {
  // Under the caller's stream capture (fluid_set_stream + hipStreamBeginCapture) the fork below would
  // pull the G sibling streams into that capture, on top of events recorded outside it.  Refused, not
  // attempted.
  HIPCHK(P, hipSetDevice(P->prm.device));
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(P->stream, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone)
    return fail(P, FLUID_E_UNSUPPORTED, "fluid_step on an inner_slabs context cannot run under stream capture");
  const bool graph = P->comm_graph && !(P->prm.flags & FLUID_FLAG_NO_GRAPH);
  if (!graph) return launch ? inner_direct(P, nsteps) : FLUID_OK;
  const size_t G = P->subs.size();
  std::vector<int> st0(G), first(G);
  for (size_t i = 0; i < G; ++i) first[i] = state_of(P->subs[i]);
  for (int s = 0; s < nsteps;) {
    int rc;
    if (!P->inner_posted) {  // eagerly: pushes for the current state, joined into this stream
      if ((rc = inner_fork(P)) || (rc = inner_refresh(P)) || (rc = inner_join(P))) return rc;
    }
    int batch = 1;
    while (batch * 2 <= P->graph_steps && batch * 2 <= nsteps - s) batch *= 2;
    const int key = state_of(P->subs[0]) | (batch << 8);  // the slabs' ping-pong indices move together
    auto it = P->graphs.find(key);
    if (it == P->graphs.end()) {
      GraphEntry ge;
      for (size_t i = 0; i < G; ++i) st0[i] = state_of(P->subs[i]);
      const bool trace = env_int("FLUID_TRACE", 0) != 0;
#define TR(msg) do { if (trace) { fprintf(stderr, "[inner graph] %s\n", msg); fflush(stderr); } } while (0)
      if (trace) { int rv = 0; (void)hipRuntimeGetVersion(&rv); fprintf(stderr, "[inner graph] hip runtime version %d\n", rv); }
      TR("begin capture");
      HIPCHK(P, hipStreamBeginCapture(P->stream, hipStreamCaptureModeRelaxed));
      rc = inner_fork(P);
      TR("forked");
      if (!rc) rc = inner_enqueue(P, batch, true);
      TR("enqueued");
      if (!rc) rc = inner_join(P);
      TR("joined");
      const hipError_t e = hipStreamEndCapture(P->stream, &ge.graph);
      TR("capture ended");
      ge.next_state = state_of(P->subs[0]);
      for (size_t i = 0; i < G; ++i) { set_state(P->subs[i], st0[i]); P->subs[i]->enq_rc = 0; }
      hipError_t ei = hipSuccess;
      if (!rc && e == hipSuccess) ei = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
      TR("instantiated");
#undef TR
      if (rc || e != hipSuccess || ei != hipSuccess) {  // nothing of it has run: direct launches from now on
        snprintf(P->comm_note, sizeof P->comm_note, "inner slabs: graph capture given up (%s)",
                 rc ? "enqueue" : e != hipSuccess ? hipGetErrorString(e) : hipGetErrorString(ei));
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        (void)hipGetLastError();
        P->comm_graph = false;
        if (launch) return inner_direct(P, nsteps - s);
        for (size_t i = 0; i < G; ++i) set_state(P->subs[i], first[i]);  // only building: nothing has run
        return FLUID_OK;
      }
      it = P->graphs.emplace(key, ge).first;
    }
    if (launch) HIPCHK(P, hipGraphLaunch(it->second.exec, P->stream));
    // the ping-pong indices `batch` steps flip, for every slab alike (undone below when only building)
    for (size_t i = 0; i < G; ++i) set_state(P->subs[i], state_of(P->subs[i]) ^ st_delta(key & 7, it->second.next_state));
    s += batch;
  }
  if (!launch)
    for (size_t i = 0; i < G; ++i) set_state(P->subs[i], first[i]);
  return FLUID_OK;
}

static int inner_reset(fluid_ctx *P) {
  int rc = inner_fork(P);
  for (size_t i = 0; i < P->subs.size() && !rc; ++i) rc = inner_fail(P, P->subs[i], fluid_reset(P->subs[i]));
  P->inner_posted = false;
  return rc ? rc : inner_join(P);
}

static int inner_add_source(fluid_ctx *P, float sx, float sy, float fx, float fy, const float rgba[4]) {
  int rc = inner_fork(P);
  for (size_t i = 0; i < P->subs.size() && !rc; ++i)
    rc = inner_fail(P, P->subs[i], fluid_add_source(P->subs[i], sx, sy, fx, fy, rgba));
  P->inner_posted = false;  // the halos hold the state before the splat
  return rc ? rc : inner_join(P);
}

static size_t inner_elem(int field) {
  switch (field) {
    case FLUID_VELOCITY: return sizeof(float2);
    case FLUID_DENSITY: return sizeof(float4);
    case FLUID_PRESSURE: case FLUID_DIVERGENCE: case FLUID_VORTICITY: return sizeof(float);
  }
  return 0;
}

// The slabs' owned rows in y order: the whole grid, as a whole-domain context hands it out.
static int inner_read(fluid_ctx *P, int field, void *dst, size_t dst_bytes) {
  const size_t elem = inner_elem(field);
  if (!elem) return FLUID_E_INVALID;
  if (dst_bytes < (size_t)P->g.n * P->g.ny * elem) return FLUID_E_SIZE;
  int rc = inner_fork(P);
  for (size_t i = 0; i < P->subs.size() && !rc; ++i) {
    fluid_ctx *s = P->subs[i];
    rc = inner_fail(P, s, fluid_read(s, field, (char *)dst + (size_t)s->row0 * s->g.n * elem, cells_owned(s) * elem));
  }
  return rc;
}

static int inner_write(fluid_ctx *P, int field, const void *src, size_t src_bytes) {
  const size_t elem = inner_elem(field);
  if (!elem || field == FLUID_VORTICITY) return FLUID_E_INVALID;
  if (src_bytes != (size_t)P->g.n * P->g.ny * elem) return FLUID_E_SIZE;
  int rc = inner_fork(P);
  for (size_t i = 0; i < P->subs.size() && !rc; ++i) {
    fluid_ctx *s = P->subs[i];
    rc = inner_fail(P, s, fluid_write(s, field, (const char *)src + (size_t)s->row0 * s->g.n * elem, cells_owned(s) * elem));
  }
  P->inner_posted = false;
  return rc;
}

static int inner_sync(fluid_ctx *P) {
  HIPCHK(P, hipSetDevice(P->prm.device));
  HIPCHK(P, hipStreamSynchronize(P->stream));
  for (fluid_ctx *s : P->subs) {
    const int rc = fluid_sync(s);  // both of its streams, then its halo flag
    if (rc) return inner_fail(P, s, rc);
  }
  return FLUID_OK;
}

// max over the slabs / sum over the slabs (the two stencil quantities read one halo row on each side)
static int inner_diagnostics(fluid_ctx *P, fluid_diag *out) {
  int rc = inner_fork(P);
  if (!rc) rc = inner_refresh(P);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  auto mx = [](float a, float b) { return a != a || b != b ? a + b : (a > b ? a : b); };  // NaN if either is
  for (size_t i = 0; i < P->subs.size(); ++i) {
    fluid_ctx *s = P->subs[i];
    if ((rc = inner_wait_halos(P, i))) return rc;
    fluid_diag d;
    if ((rc = fluid_diagnostics(s, &d))) return inner_fail(P, s, rc);
    out->max_abs_u = mx(out->max_abs_u, d.max_abs_u);
    out->max_abs_v = mx(out->max_abs_v, d.max_abs_v);
    out->max_abs_div = mx(out->max_abs_div, d.max_abs_div);
    out->max_abs_p = mx(out->max_abs_p, d.max_abs_p);
    out->max_jacobi_update = mx(out->max_jacobi_update, d.max_jacobi_update);
    out->max_abs_div_velocity = mx(out->max_abs_div_velocity, d.max_abs_div_velocity);
    for (int k = 0; k < 4; ++k) out->dye[k] += d.dye[k];
  }
  return inner_join(P);
}

static int inner_time_steps(fluid_ctx *P, int nsteps, float *ms_out) {
  HIPCHK(P, hipSetDevice(P->prm.device));
  hipEvent_t e0, e1;
  HIPCHK(P, hipEventCreate(&e0));
  HIPCHK(P, hipEventCreate(&e1));
  const int rc = [&]() -> int {
    int r = inner_steps(P, nsteps, false);  // graphs built outside the timed region
    if (r) return r;
    HIPCHK(P, hipEventRecord(e0, P->stream));
    r = inner_steps(P, nsteps, true);
    if (r) return r;
    HIPCHK(P, hipEventRecord(e1, P->stream));
    HIPCHK(P, hipEventSynchronize(e1));
    HIPCHK(P, hipEventElapsedTime(ms_out, e0, e1));
    return FLUID_OK;
  }();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

}  // extern "C"
