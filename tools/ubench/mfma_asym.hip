// Micro-benchmark: two waves per SIMD with asymmetric roles.  Wave A: per slot one
// f8f6f4 MFMA (dependent chain) + NA VALU instructions of its own; wave B: NB VALU
// instructions per slot, no MFMA.  Does B's VALU work overlap A's MFMAs when A spaces
// its MFMAs with its own VALU work?  Wall time per slot vs the pieces alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NA, int NB, int MF, int BON>
__global__ void __launch_bounds__(512, 1) k(float *out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  v16f f0 = {0};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (MF) f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
#pragma unroll
        for (int v = 0; v < NA; ++v)
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[v % 8]) : "v"(x[v % 8]), "v"(a), "v"(b));
      }
    }
  } else if (BON) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int v = 0; v < NB; ++v)
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[v % 8]) : "v"(x[v % 8]), "v"(a), "v"(b));
    }
  }
  float sacc = 0;
  for (int i = 0; i < 8; ++i) sacc += x[i];
  for (int i = 0; i < 16; ++i) sacc += f0[i];
  out[blockIdx.x * 512 + threadIdx.x] = sacc;
}

template <int NA, int NB>
void run(float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  auto go = [&](auto kern) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e6 / (iters * 8.0);
  };
  const float tA = go(k<NA, NB, 1, 0>), tAv = go(k<NA, NB, 0, 0>), tB = go(k<0, NB, 0, 1>),
              tAB = go(k<NA, NB, 1, 1>), tABv = go(k<NA, NB, 0, 1>);
  printf("A: MFMA + %2d VALU, B: %2d VALU | A alone %.1f (its VALU alone %.1f) | B alone %.1f | both %.1f ns/slot"
         " (both without MFMA %.1f)\n", NA, NB, tA, tAv, tB, tAB, tABv);
}

// three waves per SIMD: waves 0-3 pure MFMA chain, waves 4-11 VALU only (NB per slot each)
template <int NB, int MF, int BON, int KIND>
__global__ void __launch_bounds__(768, 1) k3(float *out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  v16f f0 = {0};
  float x[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[4];
  unsigned long long m = 0;
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  for (int i = 0; i < 4; ++i) p[i] = v2f{(float)lane, (float)i};
  if (wave < 4) {
    if (MF)
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 8; ++s)
          f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
  } else if (BON) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int v = 0; v < NB; ++v) {
          if (KIND == 0 || (v % 4) == 3)
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[v % 8]) : "v"(x[v % 8]), "v"(a), "v"(b));
          else if ((v % 4) == 0)
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[v % 4]) : "v"(p[v % 4]), "v"(p[(v + 1) % 4]));
          else if ((v % 4) == 1)
            asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x[v % 8]), "v"(a));
          else
            asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x[v % 8]) : "v"(x[v % 8]), "s"(m));
        }
    }
  }
  float sacc = (float)m;
  for (int i = 0; i < 8; ++i) sacc += x[i];
  for (int i = 0; i < 4; ++i) sacc += p[i].x + p[i].y;
  for (int i = 0; i < 16; ++i) sacc += f0[i];
  out[blockIdx.x * 768 + threadIdx.x] = sacc;
}

template <int NB, int KIND>
void run3(float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  auto go = [&](auto kern) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(768), 0, 0, out, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms * 1e6 / (iters * 8.0);
  };
  const float tM = go(k3<NB, 1, 0, KIND>), tV = go(k3<NB, 0, 1, KIND>), tB = go(k3<NB, 1, 1, KIND>);
  printf("3 waves/SIMD: 1 MFMA wave + 2 VALU waves x %d %s per slot | MFMA alone %.1f | VALU alone %.1f | both %.1f ns/slot\n",
         NB, KIND ? "mixed (pk/cmp/cnd/fma)" : "v_fma", tM, tV, tB);
}

int main() {
  float *out; (void)hipMalloc(&out, 256 * 768 * 4);
  run3<4, 0>(out);
  run3<5, 0>(out);
  run3<4, 1>(out);
  run3<5, 1>(out);
  run3<6, 1>(out);
  run3<8, 1>(out);
  run<0, 8>(out);
  run<4, 8>(out);
  run<8, 8>(out);
  run<4, 12>(out);
  run<8, 16>(out);
  run<2, 16>(out);
  return 0;
}
