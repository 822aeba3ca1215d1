// Micro-benchmark 2: the conv3x3 MFMA step in isolation -- 36 register-resident B
// fragments, 9 taps x (8 ds_read_b128 one tap ahead + 8 MFMAs), optional barrier
// and LDS staging writes per step.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int READS, int BARRIER, int WRITES>
__global__ void __launch_bounds__(256, 1) k(const v4i *w, int *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 12800];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 12800 / 4; i += 256) ((int *)lds)[i] = i & 1;
  __syncthreads();
  v4i bf[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bf[t][kk] = w[(t * 4 + kk) * 64 + lane];
  const int n = lane & 31, h = lane >> 5;
  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1), tx = (n & 3) | (((n >> 3) & 1) << 2);
  int aoff[9][4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int hy = ty + tap / 3, hx = tx + tap % 3;
    const int g = ((hy & 3) << 1) | ((hx >> 1) & 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) aoff[tap][kk] = (hy * 10 + hx) * 128 + (((kk * 2) ^ (h ^ g)) << 4);
  }
  v16i accs = {0};
  for (int it = 0; it < iters; ++it) {
    const unsigned char *base = lds + (it & 1) * 12800;
    v16i acc0 = {0}, acc1 = {0};
    v4i A[2][8];
    if (READS) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        A[0][kk] = *(const v4i *)(base + aoff[0][kk]);
        A[0][4 + kk] = *(const v4i *)(base + aoff[0][kk] + 5120);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) A[0][j] = v4i{lane, j, it, 1};
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (READS) {
            A[(tap + 1) & 1][kk] = *(const v4i *)(base + aoff[tap + 1][kk]);
            A[(tap + 1) & 1][4 + kk] = *(const v4i *)(base + aoff[tap + 1][kk] + 5120);
          } else {
            A[(tap + 1) & 1][kk] = A[tap & 1][kk];
            A[(tap + 1) & 1][4 + kk] = A[tap & 1][4 + kk];
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][kk], bf[tap][kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][4 + kk], bf[tap][kk], acc1, 0, 0, 0);
      }
    }
    for (int i = 0; i < 16; ++i) accs[i] += acc0[i] ^ acc1[i];
    if (WRITES) {
      v4i v = {it, tid, 1, 0};
      *(v4i *)(lds + ((it + 1) & 1) * 12800 + ((tid * 16) % 12800)) = v;
      *(v4i *)(lds + ((it + 1) & 1) * 12800 + ((tid * 16 + 4096) % 12800)) = v;
    }
    if (BARRIER) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
  }
  int s = 0;
  for (int i = 0; i < 16; ++i) s += accs[i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int R, int B, int W>
void run(const char *name, const v4i *w, int *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, B, W>), dim3(256), dim3(256), 0, 0, w, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-40s %.3f ms, %.1f ns per MFMA per SIMD\n", name, ms, ms * 1e6 / (iters * 72.0));
}

int main() {
  int *out; v4i *w;
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&w, 36 * 64 * 16);
  (void)hipMemset(w, 1, 36 * 64 * 16);
  run<0, 0, 0>("no reads", w, out);
  run<1, 0, 0>("reads", w, out);
  run<1, 1, 0>("reads + barrier", w, out);
  run<1, 1, 1>("reads + barrier + staging writes", w, out);
  run<0, 1, 1>("no reads, barrier + writes", w, out);
  return 0;
}
