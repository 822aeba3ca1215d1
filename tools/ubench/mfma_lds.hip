// Micro-benchmark: v_mfma_i32_32x32x32_i8 fed by one ds_read_b128 per MFMA,
// one wave per SIMD (512-register kernels), prefetch depth D reads ahead.
// Variants: no LDS reads / reads conflict-free / reads as conv kernel swizzle.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(int *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 4; i += blockDim.x) ((int *)lds)[i] = i & 1;
  __syncthreads();
  v16i acc0 = {0}, acc1 = {0};
  v4i b = {lane, 1, 2, 3};
  // MODE 1: linear 16 B per lane (conflict-free).  MODE 2: 128 B pixel stride + XOR swizzle
  int off[8];
  for (int j = 0; j < 8; ++j) {
    if (MODE == 2) {
      const int n = lane & 31, h = lane >> 5;
      const int ty = ((n >> 2) & 1) | ((n >> 4) << 1), tx = (n & 3) | (((n >> 3) & 1) << 2);
      const int hy = ty + (j >> 2) * 4, hx = tx;
      const int g = ((hy & 3) << 1) | ((hx >> 1) & 1);
      off[j] = (hy * 10 + hx) * 128 + ((((j & 3) * 2) ^ (h ^ g)) << 4);
    } else {
      off[j] = (lane * 16 + j * 1024) & 32767;
    }
  }
  v4i A[2][8];
  if (MODE)
    for (int j = 0; j < 8; ++j) A[0][j] = *(v4i *)(lds + off[j]);
  else
    for (int j = 0; j < 8; ++j) A[0][j] = v4i{lane, j, 1, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (MODE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) A[p ^ 1][j] = *(v4i *)(lds + ((off[j] + it * 16) & 32767 & ~15));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) A[p ^ 1][j] = A[p][j];
      }
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p][j], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p][j + 1], b, acc1, 0, 0, 0);
      }
    }
  }
  int s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int WAVES>
void run(const char *name, int *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double mfma_per_simd = (double)iters * 16 * (WAVES / 4.0);
  printf("%-28s waves/CU %d: %.3f ms, %.1f ns per MFMA per SIMD (32 cycles @2.4GHz = 13.3 ns)\n",
         name, WAVES, ms, ms * 1e6 / mfma_per_simd);
}

int main() {
  int *out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<0, 4>("no LDS reads", out);
  run<1, 4>("linear reads", out);
  run<2, 4>("conv-swizzle reads", out);
  run<0, 8>("no LDS reads", out);
  run<1, 8>("linear reads", out);
  run<2, 8>("conv-swizzle reads", out);
  return 0;
}
