// Micro-benchmark: which instruction classes of the conv epilogue overlap with an int8 /
// f8f6f4 MFMA issued by the SAME wave (one wave per SIMD)?  Per slot: one MFMA (two
// independent chains alternate) + NV instructions of one class.  Prints ns per slot for
// the MFMA alone, the class alone and both; overlap = both < alone + alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// CLS: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cmp + v_cndmask (through an SGPR pair), 3 v_cvt,
//      4 ds_read_b32, 5 ds_read_b128, 6 v_writelane, 7 v_pk_mul + v_pk_add,
//      8 v_cmp + v_cndmask through VCC, 9 v_cvt_i32_f32
template <int CLS, int NV, int MF, int WPS>   // MF: 0 none, 1 int8 32x32x32, 2 f8f6f4 fp4 x fp6
__global__ void __launch_bounds__(256, WPS) k(int *out, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 0.5f;
  __syncthreads();
  v16i acc0 = {0}, acc1 = {0};
  v16f f0 = {0}, f1 = {0};
  v4i av = {lane, 1, 2, 3}, bv = {3, lane, 1, 0};
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  float x[8];
  v2f p[8];
  v4i q[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int xi[8];
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds + lane * 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = lane + i; p[i] = v2f{(float)lane, (float)i}; xi[i] = lane * i; }
  unsigned long long m = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MF == 1) {
        if (s & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc0, 0, 0, 0);
      } else if (MF == 2) {
        if (s & 1) f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 4, 2, 0, 127, 0, 127);
        else f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 127, 0, 127);
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int r = v % 8;
        if (CLS == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[r]) : "v"(x[r]), "v"(a), "v"(b));
        if (CLS == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 1) % 8]), "v"(p[(r + 2) % 8]));
        if (CLS == 2) {
          if (v & 1) asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x[r]) : "v"(x[r]), "s"(m));
          else asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(x[r]));
        }
        if (CLS == 8) {       // the same through VCC (VOP2 / VOPC encodings)
          if (v & 1) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x[r]) : "v"(x[r]), "v"(b) : );
          else asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(a), "v"(x[r]) : "vcc");
        }
        if (CLS == 9) asm volatile("v_cvt_i32_f32_e32 %0, %1" : "=v"(xi[r]) : "v"(x[r]));
        if (CLS == 3) asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(x[r]) : "v"(xi[r]));
        if (CLS == 4) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x[r]) : "v"(addr), "n"(256 * 1));
        if (CLS == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[v % 4]) : "v"(addr), "n"(1024));
        if (CLS == 6) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(xi[r]) : "s"((int)m));
        if (CLS == 7) {
          if (v & 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 1) % 8]));
          else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[r]) : "v"(p[r]), "v"(p[(r + 1) % 8]));
        }
      }
      if (CLS == 4 || CLS == 5) asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  float sacc = 0;
  for (int i = 0; i < 8; ++i) sacc += x[i] + p[i].x + p[i].y + xi[i];
  int r = (int)sacc + (int)m;
  for (int i = 0; i < 4; ++i) r += q[i].x + q[i].w;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + (int)f0[i] + (int)f1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CLS, int NV, int MF, int WPS>
float run(int *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CLS, NV, MF, WPS>), dim3(256 * WPS), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e6f / (iters * 8.0f);
}

template <int CLS, int NV, int WPS = 1>
void row(int *out, const char *name) {
  const float v = run<CLS, NV, 0, WPS>(out), i8 = run<CLS, 0, 1, WPS>(out), f6 = run<CLS, 0, 2, WPS>(out);
  const float bi = run<CLS, NV, 1, WPS>(out), bf = run<CLS, NV, 2, WPS>(out);
  printf("[%d wave/SIMD] %-24s x%-2d alone %5.1f ns | int8 MFMA %5.1f, both %5.1f (sum %5.1f) | f8f6f4 %5.1f, both %5.1f (sum %5.1f)\n",
         WPS, name, NV, v, i8, bi, v + i8, f6, bf, v + f6);
}

int main() {
  int *out; (void)hipMalloc(&out, 2 * 256 * 256 * 4);
  row<0, 8>(out, "v_fma_f32");
  row<1, 4>(out, "v_pk_fma_f32");
  row<7, 4>(out, "v_pk_mul/add_f32");
  row<2, 8>(out, "v_cmp(sgpr)+v_cndmask");
  row<8, 8>(out, "v_cmp(vcc)+v_cndmask e32");
  row<3, 8>(out, "v_cvt_f32_i32");
  row<9, 8>(out, "v_cvt_i32_f32");
  row<4, 8>(out, "ds_read_b32 + wait");
  row<5, 2>(out, "ds_read_b128 + wait");
  row<5, 4>(out, "ds_read_b128 + wait");
  row<6, 8>(out, "v_writelane_b32");
  // two waves per SIMD: ns per slot of EACH wave (two slots complete in that time)
  row<0, 8, 2>(out, "v_fma_f32");
  row<1, 4, 2>(out, "v_pk_fma_f32");
  row<2, 8, 2>(out, "v_cmp(sgpr)+v_cndmask");
  row<8, 8, 2>(out, "v_cmp(vcc)+v_cndmask e32");
  row<3, 8, 2>(out, "v_cvt_f32_i32");
  row<9, 8, 2>(out, "v_cvt_i32_f32");
  row<5, 2, 2>(out, "ds_read_b128 + wait");
  row<6, 8, 2>(out, "v_writelane_b32");
  return 0;
}
