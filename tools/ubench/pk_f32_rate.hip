// Issue rate of packed float32 vector instructions on gfx950: v_pk_fma_f32 / v_pk_add_f32
// against v_fma_f32 / v_add_f32, independent chains, 1 to 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  v2f x[8];
  for (int i = 0; i < 8; ++i) x[i] = v2f{(float)threadIdx.x + i, (float)i};
  const v2f av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(av), "v"(bv));
        if (MODE == 1) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i].x) : "v"(x[i].x), "v"(a), "v"(b));
                         asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i].y) : "v"(x[i].y), "v"(a), "v"(b)); }
        if (MODE == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(av));
        if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[i]), "v"(av));
        // the event layer's threshold / reset pair: compare into an SGPR pair, select on it
        if (MODE == 4) { unsigned long long m;
                         asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(x[i].x));
                         asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(x[i].x) : "v"(x[i].x), "s"(m)); }
        if (MODE == 5) { unsigned long long m;       // compares only (two per pair)
                         asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(x[i].x));
                         asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(x[i].y)); }
        if (MODE == 6) { asm volatile("v_cndmask_b32_e64 %0, %1, 0, vcc" : "=v"(x[i].x) : "v"(x[i].x));   // selects only
                         asm volatile("v_cndmask_b32_e64 %0, %1, 0, vcc" : "=v"(x[i].y) : "v"(x[i].y)); }
        if (MODE == 7) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(x[i].x) : "v"(x[i].x), "v"(a));
                         asm volatile("v_min_f32 %0, %1, %2" : "=v"(x[i].y) : "v"(x[i].y), "v"(b)); }
        if (MODE == 8) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(x[i].x) : "v"(x[i].x), "v"(a));
                         asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i].y) : "v"(x[i].y), "v"(b)); }
        if (MODE == 9) { asm volatile("v_mul_f32 %0, %1, %2 clamp" : "=v"(x[i].x) : "v"(x[i].x), "v"(a));
                         asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[i].y) : "v"(x[i].y), "v"(b)); }
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *out, int wgs_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  // value-updates per wave: iters * 32 pairs * 2; per SIMD: wgs_per_cu waves
  const double pairs = (double)iters * 32 * wgs_per_cu;             // pair-updates per SIMD
  printf("%-22s waves/SIMD %d: %.2f ns per pair (two instructions, or one packed) per SIMD = %.2f cycles at 2.4 GHz\n",
         name, wgs_per_cu, ms * 1e6 / pairs, ms * 1e6 / pairs * 2.4);
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 2; w <= 4; w *= 2) {
    run<0>("v_pk_fma_f32", out, w); run<1>("2 x v_fma_f32", out, w);
    run<2>("v_pk_add_f32", out, w); run<3>("v_pk_mul_f32", out, w);
    run<4>("v_cmp + v_cndmask", out, w); run<5>("2 x v_cmp_le_f32", out, w);
    run<6>("2 x v_cndmask", out, w); run<7>("v_max + v_min", out, w);
    run<8>("v_and + v_add_u32", out, w); run<9>("v_mul clamp + v_sub", out, w);
  }
  return 0;
}
