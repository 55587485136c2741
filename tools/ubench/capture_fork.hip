// capture_fork.hip — the stream/event shape of inner_enqueue (csrc/fluid_api.hip) with empty
// kernels, to find which part of it hipStreamEndCapture cannot digest.
// Usage: capture_fork VARIANT [G] [STEPS]
//   A  origin forks G streams, siblings wait on each other's events (all recorded inside the
//      capture, each event recorded once), everything joins the origin        (1 step)
//   B  the same for STEPS steps: the sibling events are re-recorded inside one capture
//   C  as A, but the first sibling wait is on an event recorded BEFORE the capture began
//      (inner_posted == true: the halo pushes of the previous, uncaptured call)
//   E  as C, but the refused wait is ignored and the capture carried on to hipStreamEndCapture
//      (what a caller that only collects error codes does)
//   D  as B, with the cross-stream waits replaced by waits on the origin (control: fork/join only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
static char g_variant = 'A';
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess && g_variant == 'E') { printf("%s: %s (line %d), carrying on\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); (void)hipGetLastError(); } else if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); return 2; } } while (0)
__global__ void k_nop(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

int main(int argc, char **argv) {
  const char v = argc > 1 ? argv[1][0] : 'A';
  g_variant = v;
  const int G = argc > 2 ? atoi(argv[2]) : 2, steps = argc > 3 ? atoi(argv[3]) : (v == 'B' || v == 'D' ? 3 : 1);
  hipStream_t P;
  CK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
  std::vector<hipStream_t> s(G);
  std::vector<hipEvent_t> ready(G), halo(G), done(G);
  hipEvent_t fork;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int i = 0; i < G; ++i) {
    CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ready[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&halo[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
  }
  auto push = [&](int i) -> int {  // inner_push: wait for the neighbours' ready, copy, record my halo
    for (int d = -1; d <= 1; d += 2) {
      const int q = i + d;
      if (q < 0 || q >= G) continue;
      if (v != 'D') CK(hipStreamWaitEvent(s[i], ready[q], 0));
      k_nop<<<1, 64, 0, s[i]>>>(nullptr);
    }
    CK(hipEventRecord(halo[i], s[i]));
    return 0;
  };
  if (v == 'C' || v == 'E') {  // the previous call's pushes, outside any capture
    for (int i = 0; i < G; ++i) CK(hipEventRecord(ready[i], s[i]));
    for (int i = 0; i < G; ++i) if (push(i)) return 2;
    CK(hipDeviceSynchronize());
  }
  printf("variant %c, G=%d, steps=%d: begin capture\n", v, G, steps); fflush(stdout);
  CK(hipStreamBeginCapture(P, hipStreamCaptureModeRelaxed));
  CK(hipEventRecord(fork, P));
  for (int i = 0; i < G; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
  if (v != 'C' && v != 'E') {  // inner_refresh
    for (int i = 0; i < G; ++i) CK(hipEventRecord(ready[i], s[i]));
    for (int i = 0; i < G; ++i) if (push(i)) return 2;
  }
  for (int k = 0; k < steps; ++k) {
    for (int i = 0; i < G; ++i) {
      if (v != 'D') {
        if (v == 'E') {
          hipError_t e1 = i > 0 ? hipStreamWaitEvent(s[i], halo[i - 1], 0) : hipSuccess;
          hipError_t e2 = i < G - 1 ? hipStreamWaitEvent(s[i], halo[i + 1], 0) : hipSuccess;
          if (e1 != hipSuccess || e2 != hipSuccess) { printf("wait refused (%s), carrying on\n", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); fflush(stdout); }
          (void)hipGetLastError();
        } else {
        if (i > 0) CK(hipStreamWaitEvent(s[i], halo[i - 1], 0));
        if (i < G - 1) CK(hipStreamWaitEvent(s[i], halo[i + 1], 0));
        }
      }
      k_nop<<<1, 64, 0, s[i]>>>(nullptr);
      CK(hipEventRecord(ready[i], s[i]));
      k_nop<<<1, 64, 0, s[i]>>>(nullptr);
    }
    for (int i = 0; i < G; ++i) {
      if (push(i)) return 2;
      k_nop<<<1, 64, 0, s[i]>>>(nullptr);
    }
  }
  for (int i = 0; i < G; ++i) {
    CK(hipEventRecord(done[i], s[i]));
    CK(hipStreamWaitEvent(P, done[i], 0));
  }
  printf("end capture ...\n"); fflush(stdout);
  hipGraph_t g = nullptr;
  CK(hipStreamEndCapture(P, &g));
  size_t nodes = 0;
  CK(hipGraphGetNodes(g, nullptr, &nodes));
  printf("captured: %zu nodes\n", nodes); fflush(stdout);
  hipGraphExec_t ex = nullptr;
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, P));
  CK(hipStreamSynchronize(P));
  printf("variant %c: instantiated and replayed 3 times: OK\n", v);
  return 0;
}
