// Micro-benchmark: which VALU instruction kinds overlap an f8f6f4 (or int8) MFMA issued
// by the same wave?  Per slot: one MFMA (dependent chain) + NV filler instructions of
// one kind; two waves per SIMD.  Reported: ns per slot per wave with and without MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32   1: v_pk_add_f32   2: v_cmp_le_f32 -> sgpr pair   3: v_cndmask (sgpr mask)
//      4: v_cvt_u32_f32   5: v_mov_b32   6: v_cmp + v_cndmask pairs   7: v_writelane
template <int KIND, int NV, int MF>
__global__ void __launch_bounds__(512, 1) k(float *out, int iters, float a, float b) {
  const int lane = threadIdx.x & 63;
  v16f f0 = {0};
  v16i i0 = {0};
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  v4i a4 = {lane, 1, 2, 3}, b4 = {3, lane, 1, 0};
  float x[8];
  v2f p[8];
  unsigned long long m = 0;
  unsigned w = 0;
  for (int i = 0; i < 8; ++i) { x[i] = lane + i; p[i] = v2f{(float)lane, (float)i}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MF == 1) f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
      if (MF == 2) i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, i0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int r = v % 8;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[r]) : "v"(x[r]), "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 1) % 8]));
        if (KIND == 2) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x[r]), "v"(a));
        if (KIND == 3) asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x[r]) : "v"(x[r]), "s"(m));
        if (KIND == 4) asm volatile("v_cvt_u32_f32_e32 %0, %1" : "=v"(x[r]) : "v"(x[r]));
        if (KIND == 5) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x[r]) : "v"(x[(r + 1) % 8]));
        if (KIND == 6) {
          if (v & 1) asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x[r]) : "v"(x[r]), "s"(m));
          else asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x[r]), "v"(a));
        }
        if (KIND == 7) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(w) : "s"((unsigned)m));
      }
    }
  }
  float sacc = w;
  for (int i = 0; i < 8; ++i) sacc += x[i] + p[i].x + p[i].y;
  for (int i = 0; i < 16; ++i) sacc += f0[i] + i0[i];
  out[blockIdx.x * 512 + threadIdx.x] = sacc + (float)m;
}

template <int KIND, int NV>
void run(float *out, const char *name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  float t[3];
  auto go = [&](auto kern, int idx) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.5f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    t[idx] = ms * 1e6 / (iters * 8.0);
  };
  go(k<KIND, NV, 0>, 0);
  go(k<KIND, NV, 1>, 1);
  go(k<KIND, NV, 2>, 2);
  printf("%-22s x%d per slot: alone %.1f ns | + f8f6f4 %.1f ns | + int8 %.1f ns   (MFMA alone ~14-15 ns; 2 waves/SIMD)\n",
         name, NV, t[0], t[1], t[2]);
}

int main() {
  float *out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<0, 8>(out, "v_fma_f32");
  run<1, 4>(out, "v_pk_add_f32");
  run<1, 8>(out, "v_pk_add_f32");
  run<2, 8>(out, "v_cmp -> sgpr");
  run<3, 8>(out, "v_cndmask sgpr mask");
  run<6, 8>(out, "v_cmp + v_cndmask");
  run<4, 8>(out, "v_cvt_u32_f32");
  run<5, 8>(out, "v_mov_b32");
  run<7, 4>(out, "v_writelane");
  return 0;
}
