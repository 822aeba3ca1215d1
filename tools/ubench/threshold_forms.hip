// Encodings of the threshold + hard reset of the fused conv epilogues on gfx950: does the
// VOP2/VOPC (e32, implicit VCC) form of v_cmp / v_cndmask issue faster than the VOP3 (e64,
// SGPR pair) form the kernels use?  Per 32-pixel x 32-channel tile-step a wave runs 8 pairs of
// { update, 2 compares, 2 selects, the spike word }.  4 waves per SIMD, no LDS, no MFMA.
// Prints SIMD cycles per tile (2.4 GHz).  hipcc --offload-arch=gfx950 -O3 threshold_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

// UPD: 0 = packed update (conv0: v_pk_add + v_pk_fma), 1 = scalar update (conv1/2: 2 x sub, 2 x fma),
//      2 = no update (threshold / reset alone)
// THR: 0 = e64 compare into SGPR pairs + e64 select (what the kernels emit)
//      1 = e32 compare into VCC + e32 select on VCC + s_mov of VCC (inverted sense: th > t keeps)
//      2 = e64 compare into VCC + e64 select on VCC + s_mov
//      3 = e32 compare into VCC + e64 select on VCC + s_mov
//      4 = none
template <int UPD, int THR>
__global__ void __launch_bounds__(256) k(float *out, int iters, float kk, float th) {
  v2f u[8], x[8];
  for (int i = 0; i < 8; ++i) { u[i] = v2f{0.1f * threadIdx.x, 0.2f * i}; x[i] = v2f{0.3f + i, 0.7f}; }
  const v2f kv = {kk, kk};
  unsigned word = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v2f t = u[i];
      if (UPD == 0) {
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x[i]), "v"(u[i]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(kv), "v"(u[i]));
      } else if (UPD == 3) {         // conv1/2 today: dequantise (3), BatchNorm multiply, sub, fma
        float y0, y1, q0, q1, d0, d1;
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(q0) : "v"(kk), "v"(x[i].x));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(q1) : "v"(kk), "v"(x[i].y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(x[i].x), "v"(th), "v"(q0));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(x[i].y), "v"(th), "v"(q1));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(kk), "v"(q0));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(kk), "v"(q1));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(th), "v"(y0));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(th), "v"(y1));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(y0), "v"(u[i].x));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(y1), "v"(u[i].y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(d0), "v"(kk), "v"(u[i].x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(d1), "v"(kk), "v"(u[i].y));
      } else if (UPD == 4) {         // conv1/2 with the table dequantisation: BatchNorm multiply, sub, fma
        float y0, y1, d0, d1;
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(th), "v"(x[i].x));
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(th), "v"(x[i].y));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(y0), "v"(u[i].x));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(y1), "v"(u[i].y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(d0), "v"(kk), "v"(u[i].x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(d1), "v"(kk), "v"(u[i].y));
      } else if (UPD == 1) {
        float d0, d1;
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(x[i].x), "v"(u[i].x));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(x[i].y), "v"(u[i].y));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "v"(d0), "v"(kk), "v"(u[i].x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "v"(d1), "v"(kk), "v"(u[i].y));
      }
      unsigned long long m0 = 0, m1 = 0;
      if (THR == 0) {
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t.x));
        asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t.y));
        asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u[i].x) : "v"(t.x), "s"(m0));
        asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u[i].y) : "v"(t.y), "s"(m1));
      } else if (THR == 1) {
        asm volatile("v_cmp_gt_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, 0, %3, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].x), "=s"(m0) : "v"(th), "v"(t.x) : "vcc");
        asm volatile("v_cmp_gt_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, 0, %3, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].y), "=s"(m1) : "v"(th), "v"(t.y) : "vcc");
      } else if (THR == 2) {
        asm volatile("v_cmp_le_f32_e64 vcc, %2, %3\n\tv_cndmask_b32_e64 %0, %3, 0, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].x), "=s"(m0) : "v"(th), "v"(t.x) : "vcc");
        asm volatile("v_cmp_le_f32_e64 vcc, %2, %3\n\tv_cndmask_b32_e64 %0, %3, 0, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].y), "=s"(m1) : "v"(th), "v"(t.y) : "vcc");
      } else if (THR == 3) {
        asm volatile("v_cmp_le_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e64 %0, %3, 0, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].x), "=s"(m0) : "v"(th), "v"(t.x) : "vcc");
        asm volatile("v_cmp_le_f32_e32 vcc, %2, %3\n\tv_cndmask_b32_e64 %0, %3, 0, vcc\n\ts_mov_b64 %1, vcc"
                     : "=v"(u[i].y), "=s"(m1) : "v"(th), "v"(t.y) : "vcc");
      } else if (THR == 5) {     // v_cmpx into EXEC (and the mask), zero the spiking lanes, EXEC back
        asm volatile("v_cmpx_le_f32_e64 %1, %2, %0\n\tv_mov_b32_e32 %0, 0\n\ts_mov_b64 exec, -1"
                     : "+v"(t.x), "=s"(m0) : "v"(th));
        asm volatile("v_cmpx_le_f32_e64 %1, %2, %0\n\tv_mov_b32_e32 %0, 0\n\ts_mov_b64 exec, -1"
                     : "+v"(t.y), "=s"(m1) : "v"(th));
        u[i] = t;
      } else {
        u[i] = t;
      }
      if (THR != 4) {
        // THR 1 holds the KEEP masks (th > t): spike = not keep; the pooled word is the OR of
        // the four spike bits of a channel
        const unsigned long long m = THR == 1 ? ~(m0 & m1) : (m0 | m1);
        const unsigned w = (unsigned)m | (unsigned)(m >> 32);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
      }
    }
  }
  float s = (float)word;
  for (int i = 0; i < 8; ++i) s += u[i].x + u[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The table-mode epilogue (BatchNorm multiply, sub, fma, compare, select, spike word) with G
// pairs advanced together stage by stage: 2 G independent instructions between dependent ones.
template <int G>
__global__ void __launch_bounds__(256) kg(float *out, int iters, float kk, float th) {
  float u[16], x[16];
  for (int i = 0; i < 16; ++i) { u[i] = 0.1f * threadIdx.x + 0.2f * i; x[i] = 0.3f + i; }
  unsigned word = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int b = 0; b < 16; b += 2 * G) {
      float y[2 * G], d[2 * G], t[2 * G];
      unsigned long long m[2 * G];
#pragma unroll
      for (int i = 0; i < 2 * G; ++i) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y[i]) : "v"(th), "v"(x[b + i]));
#pragma unroll
      for (int i = 0; i < 2 * G; ++i) asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d[i]) : "v"(y[i]), "v"(u[b + i]));
#pragma unroll
      for (int i = 0; i < 2 * G; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t[i]) : "v"(d[i]), "v"(kk), "v"(u[b + i]));
#pragma unroll
      for (int i = 0; i < 2 * G; ++i) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(th), "v"(t[i]));
#pragma unroll
      for (int i = 0; i < 2 * G; ++i) asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u[b + i]) : "v"(t[i]), "s"(m[i]));
#pragma unroll
      for (int i = 0; i < 2 * G; i += 2) {
        const unsigned long long mm = m[i] | m[i + 1];
        const unsigned w = (unsigned)mm | (unsigned)(mm >> 32);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
      }
    }
  }
  float s = (float)word;
  for (int i = 0; i < 16; ++i) s += u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int G>
void run_g(float *out, int waves_per_simd) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kg<G>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double tiles = (double)iters * waves_per_simd;
  printf("table-mode epilogue, %d pairs advanced together          %d waves/SIMD: %.1f SIMD cycles per tile of 16 registers\n",
         G, waves_per_simd, ms * 1e6 / tiles * 2.4);
}

template <int UPD, int THR>
void run(const char *name, float *out, int waves_per_simd = 4) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<UPD, THR>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double tiles = (double)iters * waves_per_simd;      // tile-steps per SIMD
  printf("%-52s %d waves/SIMD: %.1f SIMD cycles per tile of 16 registers (2.4 GHz)\n", name,
         waves_per_simd, ms * 1e6 / tiles * 2.4);
}

int main() {
  float *out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  run<2, 4>("nothing (loop only)", out);
  run<0, 4>("packed update only", out);
  run<1, 4>("scalar update only", out);
  run<2, 0>("threshold+reset e64/SGPR (kernels today)", out);
  run<2, 1>("threshold+reset e32/VCC + s_mov", out);
  run<2, 2>("threshold+reset e64/VCC + s_mov", out);
  run<2, 3>("threshold e32/VCC, reset e64/VCC + s_mov", out);
  run<2, 5>("threshold+reset v_cmpx + masked v_mov", out);
  run<0, 5>("packed update + v_cmpx + masked v_mov", out);
  run<0, 0>("packed update + e64/SGPR (conv0 today)", out);
  run<0, 1>("packed update + e32/VCC", out);
  run<0, 3>("packed update + e32 cmp, e64 select", out);
  run<1, 0>("scalar update + e64/SGPR (conv1/2 today)", out);
  run<1, 1>("scalar update + e32/VCC", out);
  // the bits kernel's whole epilogue (dequantise 3, BatchNorm multiply, sub, fma, threshold,
  // reset, spike word) alone, at its two waves per SIMD: what its vector instructions cost
  run<3, 0>("conv1/2 epilogue: dq + bn + update + thr", out, 2);
  run<3, 0>("conv1/2 epilogue: dq + bn + update + thr", out, 4);
  run<3, 4>("conv1/2 epilogue without threshold/reset", out, 2);
  run<4, 0>("conv1/2 epilogue, table dequantisation: bn + update + thr", out, 2);
  run<4, 0>("conv1/2 epilogue, table dequantisation: bn + update + thr", out, 4);
  run_g<1>(out, 2); run_g<2>(out, 2); run_g<4>(out, 2); run_g<8>(out, 2);
  run_g<1>(out, 4); run_g<2>(out, 4);
  return 0;
}
