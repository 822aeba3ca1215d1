// Micro-benchmark: ds_read_b128 rate of the A-fragment access patterns of the conv
// kernels (lane -> byte offset), 8 waves per CU, against a linear pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(int *out, const int *offs, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 4; i += 512) ((int *)lds)[i] = i;
  __syncthreads();
  int off = offs[lane];
  v4i acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 18; ++j) {      // 18 reads at tap-like constant offsets
      const int c = ((j / 2) / 3 * 10 + (j / 2) % 3) * 32 + (j & 1) * 3200;
      v4i v = *(v4i *)(lds + off + c);
      acc += v;
    }
    asm volatile("" : "+v"(off));
  }
  out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
}

static int ty_of(int n) { return ((n >> 2) & 1) | ((n >> 4) << 1); }
static int tx_of(int n) { return (n & 3) | (((n >> 3) & 1) << 2); }

int g_S, g_P, g_SW, g_HO;
static int gen(int l) {
  int n = l & 31, h = l >> 5, ty = ty_of(n), tx = tx_of(n);
  int hh = h;
  if (g_SW == 1) hh ^= ty & 1;
  if (g_SW == 2) hh ^= (ty >> 1) & 1;
  if (g_SW == 3) hh ^= tx & 1;
  if (g_SW == 4) hh ^= (tx >> 2) & 1;
  if (g_S == 32) return (ty * g_P + tx) * 32 + hh * 16;
  return (ty * g_P + tx) * 16 + hh * g_HO;          // one plane per lane half
}

int main() {
  int *out, *doff;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&doff, 64 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char *name) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = gen(l);
    (void)hipMemcpy(doff, h, sizeof(h), hipMemcpyHostToDevice);
    float ms = 0;
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, doff, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = 8.0 * iters * 18 * 1024;
    printf("%s S=%d P=%d swz=%d HO=%d: %.1f B/clk\n", name, g_S, g_P, g_SW, g_HO, bytes / (ms * 1e-3 * 2.4e9));
  };
  for (g_S = 32, g_HO = 0, g_P = 10; g_P <= 16; ++g_P)
    for (g_SW = 0; g_SW <= 4; ++g_SW) run("pixel32");
  for (g_S = 16, g_P = 10; g_P <= 16; ++g_P)
    for (g_SW = 0; g_SW <= 2; ++g_SW)
      for (int a = 0; a < 4; ++a) { g_HO = 4096 + a * 64; run("planes16"); }
  return 0;
}
