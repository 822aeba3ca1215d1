// Is a hipMemsetAsync captured in front of a kernel ordered before that kernel on every replay,
// also when two graphs are replayed concurrently on two streams?  (Round 3 saw a fault with
// memset nodes in front of captured conv launches, round 5 a ticket word that was not zero at
// kernel start in tests/test_parity_gpu.py::test_two_live_captures...; this program has none of
// the library's slot / workspace logic in it.)
//
//   graph g = [ zero(buf_g, NW words) ; kernel: every workgroup atomicAdd(buf_g[w], 1) and
//               counts the values it drew that a launch starting from zero cannot draw ]
// zero = hipMemsetAsync (a memset node) or a one-workgroup kernel (a kernel node).
// Replays: graph 0 alone; graphs 0 and 1 alternating on ONE stream; graphs 0 and 1 on TWO
// streams without a synchronisation in between.  Prints anomalies per mode.
//
//   hipcc --offload-arch=gfx950 -O2 -o graph_memset_order tools/ubench/graph_memset_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int NW = 24;        // ticket words (96 bytes, as the failing test's memset)
constexpr int PER = 2;        // arrivals per word and launch

__global__ void draw(unsigned *buf, unsigned *bad, int spin) {
  // blockIdx.x = word * PER + arrival
  const int w = blockIdx.x / PER;
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(buf + w, 1u);
    if (t >= (unsigned)PER) atomicAdd(bad, 1u);
  }
  // some work, so that launches overlap in time
  volatile float x = 1.0f;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
}

__global__ void zero_words(unsigned *buf, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = 0u;
}

struct G { hipGraph_t g; hipGraphExec_t e; unsigned *buf; unsigned *base; };

// interior: the target is an INTERIOR pointer of a larger allocation (as a framework's pool block
// or the library's slot pool is), at the offset the failing test had (0xa00)
static bool g_interior = false;

static G capture(hipStream_t st, unsigned *bad, bool memset_node, int spin) {
  G r;
  CK(hipMalloc((void **)&r.base, 1 << 21));
  CK(hipMemset(r.base, 0x5a, 1 << 21));
  r.buf = g_interior ? r.base + 0xa00 / 4 : r.base;
  CK(hipMemset(r.buf, 0, 4096));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  if (memset_node) CK(hipMemsetAsync(r.buf, 0, NW * sizeof(unsigned), st));
  else hipLaunchKernelGGL(zero_words, dim3(1), dim3(64), 0, st, r.buf, NW);
  hipLaunchKernelGGL(draw, dim3(NW * PER), dim3(64), 0, st, r.buf, bad, spin);
  CK(hipStreamEndCapture(st, &r.g));
  CK(hipGraphInstantiate(&r.e, r.g, nullptr, nullptr, 0));
  return r;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
  const int spin = argc > 2 ? std::atoi(argv[2]) : 2000;
  hipStream_t s0, s1, cap;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  unsigned *bad;
  CK(hipMalloc((void **)&bad, 4));
  for (int kind = 0; kind < 4; ++kind) {
    const bool memset_node = (kind & 1) == 0;
    g_interior = kind >= 2;
    G a = capture(cap, bad, memset_node, spin), b = capture(cap, bad, memset_node, spin);
    for (int mode = 0; mode < 4; ++mode) {
      CK(hipMemset(bad, 0, 4));
      CK(hipMemset(a.buf, 0, 4096));
      CK(hipMemset(b.buf, 0, 4096));
      CK(hipDeviceSynchronize());
      for (int i = 0; i < iters; ++i) {
        if (mode == 0) { CK(hipGraphLaunch(a.e, s0)); }
        else if (mode == 1) { CK(hipGraphLaunch(a.e, s0)); CK(hipGraphLaunch(b.e, s0)); }
        else if (mode == 2) { CK(hipGraphLaunch(a.e, s0)); CK(hipGraphLaunch(b.e, s1)); }
        else {               // two streams, host waits for both every iteration (the test's pattern)
          CK(hipGraphLaunch(a.e, s0)); CK(hipGraphLaunch(b.e, s1)); CK(hipDeviceSynchronize());
        }
      }
      CK(hipDeviceSynchronize());
      unsigned nbad = 0;
      CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
      static const char *names[4] = {"one graph, one stream", "two graphs alternating on one stream",
                                     "two graphs on two streams, no host sync", "two graphs on two streams, host sync per pair"};
      std::printf("%-12s %-9s %-50s %d replays each: %u draws of a non-zero-start value\n",
                  memset_node ? "memset node" : "kernel node", g_interior ? "interior" : "base", names[mode], iters, nbad);
    }
    CK(hipGraphExecDestroy(a.e)); CK(hipGraphDestroy(a.g)); CK(hipFree(a.base));
    CK(hipGraphExecDestroy(b.e)); CK(hipGraphDestroy(b.g)); CK(hipFree(b.base));
  }
  return 0;
}
