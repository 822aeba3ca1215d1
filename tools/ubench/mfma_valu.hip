// Micro-benchmark 3: does VALU work issued between MFMAs of the SAME wave overlap
// with the matrix pipe?  One wave per SIMD; per MFMA, NV independent v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int MF, int WPS>
__global__ void __launch_bounds__(256 * WPS > 1024 ? 1024 : 256) k(int *out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63;
  v16i acc0 = {0}, acc1 = {0};
  v4i av = {lane, 1, 2, 3}, bv = {3, lane, 1, 0};
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  v16f f0 = {0}, f1 = {0};
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF == 3) {        // one dependent chain
        f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
      } else if (MF == 2) {
        if (m & 1) f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 4, 2, 0, 127, 0, 127);
        else f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
      } else if (MF) {
        if (m & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc0, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v)
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[v % 8]) : "v"(x[v % 8]), "v"(a), "v"(b));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  int r = (int)s;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + (int)f0[i] + (int)f1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NV, int MF, int WPS>
void run(int *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MF, WPS>), dim3(256 * WPS), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("MFMA %d, VALU/MFMA-slot %d, waves/SIMD %d: %.3f ms -> %.1f ns per slot per wave\n", MF, NV, WPS, ms,
         ms * 1e6 / (iters * 8.0));
}

int main() {
  int *out; (void)hipMalloc(&out, 256 * 1024 * 4 * 4);
  run<0, 1, 1>(out);
  run<4, 0, 1>(out);
  run<4, 1, 1>(out);
  run<6, 0, 1>(out);
  run<6, 1, 1>(out);
  run<8, 0, 1>(out);
  run<8, 1, 1>(out);
  run<8, 1, 2>(out);
  run<8, 0, 2>(out);
  printf("-- f8f6f4 (fp4 x fp6) MFMA, K = 64 --\n");
  run<0, 2, 1>(out);
  run<4, 2, 1>(out);
  run<8, 2, 1>(out);
  run<12, 2, 1>(out);
  run<8, 2, 2>(out);
  run<12, 2, 2>(out);
  run<12, 0, 2>(out);
  printf("-- f8f6f4, one dependent chain --\n");
  run<0, 3, 1>(out);
  run<8, 3, 1>(out);
  run<8, 3, 2>(out);
  run<10, 3, 2>(out);
  return 0;
}
