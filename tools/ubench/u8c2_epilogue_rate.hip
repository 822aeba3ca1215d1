// The event layer's per-tile vector work, alone: 8 x { v_pk_add_f32, v_pk_fma_f32, 2 v_cmp_le_f32
// into SGPR pairs, 2 v_cndmask on them, s_or_b64, s_or_b32, v_writelane } as the kernel emits them
// (conv3x3_u8c2.hip, table path with the fused update), 4 waves per SIMD, no LDS, no MFMA.
// Prints SIMD cycles per tile: what the instruction mix costs when nothing else is in the way.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int VARIANT>
__global__ void __launch_bounds__(256, 4) k(float *out, int iters, float kk, float th) {
  v2f u[8], x[8];
  for (int i = 0; i < 8; ++i) { u[i] = v2f{0.1f * threadIdx.x, 0.2f * i}; x[i] = v2f{0.3f + i, 0.7f}; }
  const v2f kv = {kk, kk};
  unsigned word = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v2f t;
      unsigned long long m0, m1;
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x[i]), "v"(u[i]));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(kv), "v"(u[i]));
      if (VARIANT == 0) {
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t.x));
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t.y));
        asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u[i].x) : "v"(t.x), "s"(m0));
        asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u[i].y) : "v"(t.y), "s"(m1));
        unsigned long long m = m0 | m1;
        const unsigned w = (unsigned)m | (unsigned)(m >> 32);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
      } else if (VARIANT == 2) {    // threshold into EXEC (and an SGPR pair), reset as a masked move
        asm volatile("v_cmpx_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t.x) : "exec");
        asm volatile("v_mov_b32 %0, 0\n\ts_mov_b64 exec, -1" : "+v"(t.x) : : "exec");
        asm volatile("v_cmpx_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t.y) : "exec");
        asm volatile("v_mov_b32 %0, 0\n\ts_mov_b64 exec, -1" : "+v"(t.y) : : "exec");
        u[i] = t;
        unsigned long long m = m0 | m1;
        const unsigned w = (unsigned)m | (unsigned)(m >> 32);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
      } else if (VARIANT == 3) {    // plain compare, mask moved to EXEC by the scalar unit
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t.x));
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t.y));
        asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, 0\n\ts_mov_b64 exec, %2\n\tv_mov_b32 %3, 0\n\ts_mov_b64 exec, -1"
                     : "+v"(t.x) : "s"(m0), "s"(m1), "v"(t.y) : "exec");
        u[i] = t;
        unsigned long long m = m0 | m1;
        const unsigned w = (unsigned)m | (unsigned)(m >> 32);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
      } else {                      // the arithmetic only
        u[i] = t;
      }
    }
  }
  float s = (float)word;
  for (int i = 0; i < 8; ++i) s += u[i].x + u[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char *name, float *out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(256 * 4), dim3(256), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double tiles = (double)iters * 4;      // tile-steps per SIMD (4 waves)
  printf("%-28s %.1f SIMD cycles per tile of 16 registers (2.4 GHz)\n", name, ms * 1e6 / tiles * 2.4);
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  run<0>("update + threshold + reset", out);
  run<1>("update only", out);
  run<2>("v_cmpx + masked v_mov", out);
  run<3>("v_cmp + s_mov exec + v_mov", out);
  return 0;
}
