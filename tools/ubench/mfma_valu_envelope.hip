// Micro-benchmark: the envelope of the fused conv kernel's inner loop.  Every wave runs
// slots of { one f8f6f4 MFMA of a dependent chain ; NV independent-ish VALU instructions ;
// optionally one ds_read_b128 whose result feeds the MFMA PF slots later }, at one or two
// waves per SIMD (one or two 4-wave workgroups per CU).  Prints cycles (s_memtime) and ns per
// slot-pair for NV = 0..14: what the matrix pipe, the vector issue port and their
// interference allow for a given number of VALU instructions per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NV, int WPS, bool LDSR, int CHAINS>
__global__ void __launch_bounds__(256, WPS) k(float *out, unsigned long long *cyc, int iters, float a, float b) {
  __shared__ __attribute__((aligned(16))) int lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0x22222222;
  __syncthreads();
  v16f f0 = {0}, f1 = {0};
  v8i a8 = {0x22222222, 0x22, 0, 0x2200, 0, 0, 0, 0}, b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = lane + i;
  v4i q[4];
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)lds + lane * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = v4i{0x22222222, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      v8i av = a8;
      if (LDSR) { av[0] = q[s % 4].x; av[1] = q[s % 4].y; av[2] = q[s % 4].z; av[3] = q[s % 4].w; }
      if (CHAINS == 2 && (s & 1)) f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, f1, 4, 2, 0, 127, 0, 127);
      else f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, f0, 4, 2, 0, 127, 0, 127);
      __builtin_amdgcn_sched_barrier(0);
      if (LDSR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(s + 3) % 4]) : "v"(addr), "n"(1024));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int r = v % 8;    // two interleaved dependent chains of 4-ish, like the epilogue's pairs
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[r]) : "v"(x[r]), "v"(a), "v"(b));
      }
      if (LDSR) asm volatile("s_waitcnt lgkmcnt(2)");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sacc = 0;
  for (int i = 0; i < 8; ++i) sacc += x[i];
  for (int i = 0; i < 16; ++i) sacc += f0[i] + f1[i];
  for (int i = 0; i < 4; ++i) sacc += q[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int WPS, bool LDSR, int CHAINS>
void run(float *out, unsigned long long *dc) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, WPS, LDSR, CHAINS>), dim3(256 * WPS), dim3(256), 0, 0, out, dc, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long hc = 0;
  (void)hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  const double slots = iters * 18.0;
  // per slot of ONE wave; at WPS waves per SIMD the SIMD completes WPS slots in that time
  printf("NV %2d  waves/SIMD %d  lds %d chains %d : %6.1f ns/slot  %6.1f memtime-ticks/slot  -> MFMA pipe busy %4.1f %% (32-cycle MFMA at 2.4 GHz)\n",
         NV, WPS, (int)LDSR, CHAINS, ms * 1e6 / slots, (double)hc / slots, 100.0 * WPS * (32.0 / 2.4) / (ms * 1e6 / slots));
}

template <int WPS, bool LDSR, int CHAINS>
void sweep(float *out, unsigned long long *dc) {
  run<0, WPS, LDSR, CHAINS>(out, dc);
  run<2, WPS, LDSR, CHAINS>(out, dc);
  run<4, WPS, LDSR, CHAINS>(out, dc);
  run<6, WPS, LDSR, CHAINS>(out, dc);
  run<8, WPS, LDSR, CHAINS>(out, dc);
  run<10, WPS, LDSR, CHAINS>(out, dc);
  run<12, WPS, LDSR, CHAINS>(out, dc);
  run<14, WPS, LDSR, CHAINS>(out, dc);
}

int main() {
  float *out; unsigned long long *dc;
  (void)hipMalloc(&out, 2 * 256 * 256 * 4); (void)hipMalloc(&dc, 64);
  sweep<1, false, 1>(out, dc);
  sweep<2, false, 1>(out, dc);
  sweep<2, true, 1>(out, dc);
  sweep<1, true, 1>(out, dc);
  sweep<2, false, 2>(out, dc);
  return 0;
}
