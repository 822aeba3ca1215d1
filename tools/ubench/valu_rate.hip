// Micro-benchmark: VALU issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 at
// 1 / 2 / 4 waves per SIMD (independent chains).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
    if (PK) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        v2f v = {x[i], x[i + 1]}, av = {a, a}, bv = {b, b};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(av), "v"(bv));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(a), "v"(b));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD
      const int blocks = 256 * wps;                  // 256 CUs x wps blocks of 4 waves
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * (pk ? 8 : 16) * wps;   // wave-instr per SIMD
      const double elems_per_simd = (double)iters * 16 * wps * 64;
      printf("%s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD, %.1f Gelem-fma/s per SIMD\n",
             pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, ms * 1e6 / instr_per_simd,
             elems_per_simd / (ms * 1e6));
    }
  return 0;
}
