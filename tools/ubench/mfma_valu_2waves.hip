// Micro-benchmark: two waves per SIMD, one issuing a dependent chain of f8f6f4 MFMAs
// (fp4 x fp6), the other a stream of VALU work (packed f32 / cmp / cndmask like the
// neuron epilogue).  Cycles per instruction for each wave alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int VAR = 0>   // 1: MFMA waves only, 2: VALU waves only, 3: both
__global__ void __launch_bounds__(512, 1) k(const v8i *a, const v8i *b, float *d,
                                            unsigned long long *cyc, int iters) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  unsigned long long t0 = 0, t1 = 0;
  if (mf) {
    if (MODE & 1) {
      v8i av = a[l], bv = b[l];
      v16f c = {0}, c2 = {0};
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 18; ++j) {
          if (VAR == 2 && (j & 1))
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c2, 4, 2, 0, 129, 0, 127);
          else
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 2, 0, 129, 0, 127);
          if (VAR == 1) __builtin_amdgcn_s_sleep(1);
          if (VAR == 3) asm volatile("s_nop 15\n\ts_nop 15");
          if (VAR == 4) __builtin_amdgcn_s_setprio(0);
        }
      d[blockIdx.x * 512 + threadIdx.x] = c[0] + c2[1];
      t1 = __builtin_amdgcn_s_memtime();
    }
  } else if (MODE & 2) {
    if (VAR == 5) __builtin_amdgcn_s_setprio(3);
    v2f u[8], x[8];
    for (int i = 0; i < 8; ++i) { u[i] = v2f{0.1f * l, 0.2f}; x[i] = v2f{1.0f + i, 0.5f * l}; }
    float acc = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {          // 3 pk + 2 cmp + 2 cndmask per pair
        v2f dd = x[i] - u[i];
        dd = dd * 0.5f;
        v2f uu = u[i] + dd;
        u[i].x = uu.x >= 1.0f ? 0.0f : uu.x;
        u[i].y = uu.y >= 1.0f ? 0.0f : uu.y;
      }
    }
    for (int i = 0; i < 8; ++i) acc += u[i].x + u[i].y;
    d[blockIdx.x * 512 + threadIdx.x] = acc;
    t1 = __builtin_amdgcn_s_memtime();
  }
  if (blockIdx.x == 0 && l == 0) cyc[wave] = t1 - t0;
}

int main() {
  v8i *da, *db; float *dd; unsigned long long *dc, hc[8];
  (void)hipMalloc(&da, 64 * 32); (void)hipMalloc(&db, 64 * 32);
  (void)hipMalloc(&dd, 256 * 512 * 4); (void)hipMalloc(&dc, 64);
  (void)hipMemset(da, 0x22, 64 * 32); (void)hipMemset(db, 0x08, 64 * 32);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char *name) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, da, db, dd, dc, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("[%.3f ms] ", ms);
    (void)hipMemcpy(hc, dc, 64, hipMemcpyDeviceToHost);
    printf("%-12s MFMA wave: %.1f cycles / MFMA   VALU wave: %.1f cycles / pair (7 instr)\n", name,
           (double)hc[0] / (iters * 18.0), (double)hc[4] / (iters * 8.0));
  };
  run(k<1>, "MFMA alone");
  run(k<2>, "VALU alone");
  run(k<3>, "together");
  run(k<3, 1>, "tog s_sleep");
  run(k<3, 2>, "tog 2 chains");
  run(k<3, 3>, "tog s_nop");
  run(k<3, 5>, "tog valu prio3");
  return 0;
}
