// fluid_comm.hip — RCCL binding of the halo transport (include/fluid.h, fluid_comm_*).
//
// No counterpart in the reference (single GPU, no communication backend: SURVEY.md 5).  The slab
// step in fluid_api.hip moves halo rows through a `Transport` (four calls: group start, send,
// recv, group end); this file fills one in with RCCL's ncclSend / ncclRecv over xGMI, one
// communicator rank per slab (rank == slab_rank).
//
// librccl is bound at run time (dlopen on first use), not at link time: the library must load
// and serve single-GPU callers on hosts without RCCL, and a process that already holds a copy
// (PyTorch-ROCm bundles its own librccl.so.1) must not end up with two.  dlopen by SONAME gives
// the copy that is already mapped when there is one.
#include "fluid_internal.h"

#include <dlfcn.h>

// The handful of RCCL declarations this file needs, stated here (they are RCCL's / NCCL's stable
// public ABI: rccl.h, nccl.h) so that building the library needs no RCCL header either — the
// library itself is only ever bound to librccl at run time.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;  // NCCL_UNIQUE_ID_BYTES
typedef enum { ncclSuccess = 0 } ncclResult_t;         // every other value is an error
typedef enum { ncclChar = 0 } ncclDataType_t;          // ncclInt8 = ncclChar = 0
}

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct RcclApi {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
std::mutex g_api_mutex;

template <class F>
bool bind(void *dl, const char *name, F *&fn) {
  fn = reinterpret_cast<F *>(dlsym(dl, name));
  return fn != nullptr;
}

// nullptr on success, else what went wrong (static text or dlerror()).
const char *load_rccl() {
  std::lock_guard<std::mutex> lock(g_api_mutex);
  if (g_api.dl) return nullptr;
  void *dl = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    dl = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (dl) break;
  }
  if (!dl) return "librccl.so.1 not found (dlopen)";
  RcclApi a;
  a.dl = dl;
  const bool ok = bind(dl, "ncclGetUniqueId", a.GetUniqueId) && bind(dl, "ncclCommInitRank", a.CommInitRank) &&
                  bind(dl, "ncclCommDestroy", a.CommDestroy) && bind(dl, "ncclCommCount", a.CommCount) &&
                  bind(dl, "ncclCommUserRank", a.CommUserRank) && bind(dl, "ncclGroupStart", a.GroupStart) &&
                  bind(dl, "ncclGroupEnd", a.GroupEnd) && bind(dl, "ncclSend", a.Send) &&
                  bind(dl, "ncclRecv", a.Recv) && bind(dl, "ncclGetErrorString", a.GetErrorString);
  if (!ok) {
    dlclose(dl);
    return "librccl is missing an entry point this library needs";
  }
  g_api = a;
  return nullptr;
}

struct RcclUser {
  ncclComm_t comm = nullptr;
  bool owned = false;  // created by rccl_transport_init (destroyed with the transport)
  char err[160] = {0};
};

int note(RcclUser *u, ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return 0;
  snprintf(u->err, sizeof u->err, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
  return (int)r;
}

int t_group_start(void *user) { return note((RcclUser *)user, g_api.GroupStart(), "ncclGroupStart"); }
int t_send(void *user, const void *src, size_t bytes, int peer, void *stream) {
  RcclUser *u = (RcclUser *)user;
  return note(u, g_api.Send(src, bytes, ncclChar, peer, u->comm, (hipStream_t)stream), "ncclSend");
}
int t_recv(void *user, void *dst, size_t bytes, int peer, void *stream) {
  RcclUser *u = (RcclUser *)user;
  return note(u, g_api.Recv(dst, bytes, ncclChar, peer, u->comm, (hipStream_t)stream), "ncclRecv");
}
int t_group_end(void *user, void *) { return note((RcclUser *)user, g_api.GroupEnd(), "ncclGroupEnd"); }
const char *t_error(void *user) { return ((RcclUser *)user)->err; }
int t_ranks(void *user) {  // ncclCommCount: what RCCL itself says, not what the caller asked for
  int n = 0;
  RcclUser *u = (RcclUser *)user;
  return u->comm && g_api.CommCount(u->comm, &n) == ncclSuccess ? n : 0;
}
void t_destroy(void *user) {
  RcclUser *u = (RcclUser *)user;
  if (u->owned && u->comm && g_api.CommDestroy) (void)g_api.CommDestroy(u->comm);
  delete u;
}

void fill(Transport *out, RcclUser *u) {
  out->user = u;
  out->group_start = t_group_start;
  out->send = t_send;
  out->recv = t_recv;
  out->group_end = t_group_end;
  out->error = t_error;
  out->ranks = t_ranks;
  out->destroy = t_destroy;
  out->capturable = true;  // RCCL send/recv may be issued under stream capture (graph nodes)
  out->name = "rccl";
}

}  // namespace

bool rccl_unique_id(void *out, size_t bytes, char *err, size_t errlen) {
  if (const char *e = load_rccl()) { snprintf(err, errlen, "%s", e); return false; }
  if (bytes < sizeof(ncclUniqueId)) { snprintf(err, errlen, "id buffer of %zu bytes, need %zu", bytes, sizeof(ncclUniqueId)); return false; }
  ncclUniqueId id;
  const ncclResult_t r = g_api.GetUniqueId(&id);
  if (r != ncclSuccess) { snprintf(err, errlen, "ncclGetUniqueId: %s", g_api.GetErrorString(r)); return false; }
  memcpy(out, &id, sizeof id);
  return true;
}

// The calling thread's current device must be the context's.  Collective: every rank of the
// communicator calls it (ncclCommInitRank blocks until all have).
bool rccl_transport_init(const void *id_bytes, size_t bytes, int rank, int nranks, Transport *out, char *err,
                         size_t errlen) {
  if (const char *e = load_rccl()) { snprintf(err, errlen, "%s", e); return false; }
  if (bytes < sizeof(ncclUniqueId)) { snprintf(err, errlen, "id of %zu bytes, need %zu", bytes, sizeof(ncclUniqueId)); return false; }
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  RcclUser *u = new RcclUser();
  const ncclResult_t r = g_api.CommInitRank(&u->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    snprintf(err, errlen, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, g_api.GetErrorString(r));
    delete u;
    return false;
  }
  u->owned = true;
  fill(out, u);
  return true;
}

// An existing communicator (the caller keeps ownership); its rank / size must be the slab's.
bool rccl_transport_from_comm(void *comm, int rank, int nranks, Transport *out, char *err, size_t errlen) {
  if (const char *e = load_rccl()) { snprintf(err, errlen, "%s", e); return false; }
  int cr = -1, cn = -1;
  if (g_api.CommUserRank((ncclComm_t)comm, &cr) != ncclSuccess || g_api.CommCount((ncclComm_t)comm, &cn) != ncclSuccess) {
    snprintf(err, errlen, "not a usable ncclComm_t");
    return false;
  }
  if (cr != rank || cn != nranks) {
    snprintf(err, errlen, "communicator is rank %d of %d, the context is slab %d of %d", cr, cn, rank, nranks);
    return false;
  }
  RcclUser *u = new RcclUser();
  u->comm = (ncclComm_t)comm;
  fill(out, u);
  return true;
}

// One-GPU check of this file's RCCL path (the GPU tests run on a single device, where a slab
// context has no neighbour): a one-rank communicator sends `bytes` bytes to itself through the
// Transport calls — eagerly, then (use_graph) captured once as a hipGraph and replayed twice with
// new contents — and the received bytes are compared on the host.  true on success; `msg` says
// what was done or what failed.
bool rccl_selftest(int device, size_t bytes, bool use_graph, char *msg, size_t msglen) {
  auto bad = [&](const char *what, const char *detail) {
    snprintf(msg, msglen, "%s: %s", what, detail);
    return false;
  };
  if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice", "failed");
  char err[200] = {0};
  unsigned char id[sizeof(ncclUniqueId)];
  if (!rccl_unique_id(id, sizeof id, err, sizeof err)) return bad("unique id", err);
  Transport t;
  if (!rccl_transport_init(id, sizeof id, 0, 1, &t, err, sizeof err)) return bad("communicator", err);
  hipStream_t s = nullptr;
  unsigned char *a = nullptr, *b = nullptr;
  std::vector<unsigned char> host(bytes);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool ok = true;
  const char *where = "";
  auto H = [&](hipError_t e, const char *w) { if (ok && e != hipSuccess) { ok = false; where = w; snprintf(err, sizeof err, "%s", hipGetErrorString(e)); } };
  auto exchange = [&]() {
    int rc = t.group_start(t.user);
    if (!rc) rc = t.send(t.user, a, bytes, 0, s);
    if (!rc) rc = t.recv(t.user, b, bytes, 0, s);
    const int rc2 = t.group_end(t.user, s);
    if (ok && (rc || rc2)) { ok = false; where = "send/recv"; snprintf(err, sizeof err, "%s", t.error(t.user)); }
  };
  H(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "stream");
  H(hipMalloc(&a, bytes), "hipMalloc");
  H(hipMalloc(&b, bytes), "hipMalloc");
  int rounds_graph = 0;
  for (int round = 0; ok && round < (use_graph ? 3 : 1); ++round) {
    const int pattern = 0x11 * (round + 1);
    H(hipMemsetAsync(a, pattern, bytes, s), "memset");
    H(hipMemsetAsync(b, 0, bytes, s), "memset");
    if (!ok) break;
    if (round == 0) {
      exchange();  // eager: also sets up the connection outside any capture
    } else {
      if (!exec) {
        H(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed), "begin capture");
        if (ok) exchange();
        hipError_t e = hipStreamEndCapture(s, &graph);
        H(e, "end capture");
        if (ok) H(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "instantiate");
      }
      if (ok) H(hipGraphLaunch(exec, s), "graph launch");
      if (ok) ++rounds_graph;
    }
    H(hipMemcpyAsync(host.data(), b, bytes, hipMemcpyDeviceToHost, s), "copy back");
    H(hipStreamSynchronize(s), "synchronize");
    if (ok)
      for (size_t i = 0; i < bytes; ++i)
        if (host[i] != (unsigned char)pattern) { ok = false; where = "compare"; snprintf(err, sizeof err, "byte %zu of round %d is %#x, not %#x", i, round, host[i], pattern); break; }
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (s) (void)hipStreamDestroy(s);
  if (t.destroy) t.destroy(t.user);
  (void)hipGetLastError();
  if (!ok) return bad(where, err);
  snprintf(msg, msglen, "rccl self send/recv of %zu bytes: 1 eager exchange, %d graph replays", bytes, rounds_graph);
  return true;
}
