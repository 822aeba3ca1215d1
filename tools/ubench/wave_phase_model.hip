// Would the bits kernel gain from MORE waves per SIMD at the price of its intra-wave software
// pipeline?  Model of one wave's tile-step -- 18 dependent f8f6f4 MFMAs with a ds_read_b128
// each, the neuron epilogue's vector instructions (8 pairs x { 3+3 dequantise, 2 BatchNorm
// multiply, 2 sub, 2 fma, 2 compare, 2 select, s_or, v_writelane }) and NX further plain vector
// instructions standing in for staging and addressing -- arranged two ways:
//   interleaved: every MFMA slot carries its share of the vector work (the kernel today: the
//                epilogue of timestep s beside the MFMAs of s + 1, two accumulator sets)
//   phased     : the 18 MFMAs back to back, then the whole epilogue on their result (one
//                accumulator set; the overlap has to come from the SIMD's other waves)
// at 2, 3 and 4 waves per SIMD.  Prints SIMD cycles per tile-step (2.4 GHz): the kernel takes
// 1186, the matrix pipe alone needs 576.   hipcc --offload-arch=gfx950 -O3 wave_phase_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void pair(float &u0, float &u1, float a0, float a1, float k, float th,
                                     unsigned &word) {
  float q0, q1, y0, y1, d0, d1, t0, t1;
  unsigned long long m0, m1;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(q0) : "v"(k), "v"(a0));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(q1) : "v"(k), "v"(a1));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(a0), "v"(th), "v"(q0));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(a1), "v"(th), "v"(q1));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(k), "v"(q0));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(k), "v"(q1));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(th), "v"(y0));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(th), "v"(y1));
  asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(y0), "v"(u0));
  asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(y1), "v"(u1));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(d0), "v"(k), "v"(u0));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(d1), "v"(k), "v"(u1));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t0));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t1));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u0) : "v"(t0), "s"(m0));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u1) : "v"(t1), "s"(m1));
  const unsigned long long m = m0 | m1;
  const unsigned w = (unsigned)m | (unsigned)(m >> 32);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
}

// The pair with the dequantisation read from an LDS table instead (round 3's question: the
// accumulator's bit pattern as the table address): the six dequantisation instructions become
// two ds_read_b32 issued a pair ahead, plus AV address instructions per value (1: v_and of a
// magic-constant accumulator; 0: the accumulator is the address).
template <int AV>
__device__ __forceinline__ void pair_tab_issue(float &y0, float &y1, float a0, float a1, uint32_t tab) {
  uint32_t i0 = __float_as_uint(a0), i1 = __float_as_uint(a1);
  if (AV) {
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(i0) : "s"(0x7FCu), "v"(i0));
    asm volatile("v_and_b32 %0, %1, %2" : "=v"(i1) : "s"(0x7FCu), "v"(i1));
  }
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(y0) : "v"(i0), "n"(32768));
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(y1) : "v"(i1), "n"(32768));
  (void)tab;
}
__device__ __forceinline__ void pair_tab_use(float &u0, float &u1, float y0, float y1, float k, float th,
                                             unsigned &word) {
  float d0, d1, t0, t1;
  unsigned long long m0, m1;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(th), "v"(y0));
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(th), "v"(y1));
  asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d0) : "v"(y0), "v"(u0));
  asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(d1) : "v"(y1), "v"(u1));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(d0), "v"(k), "v"(u0));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(d1), "v"(k), "v"(u1));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t0));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t1));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u0) : "v"(t0), "s"(m0));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u1) : "v"(t1), "s"(m1));
  const unsigned long long m = m0 | m1;
  const unsigned w = (unsigned)m | (unsigned)(m >> 32);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
}

// PHASED 0: interleaved, 1: phased.  NX: extra plain vector instructions per tile-step.
// TAB 0: arithmetic dequantisation; 1: table, one address instruction per value; 2: table, none
template <int PHASED, int NX, int TAB = 0>
__global__ void __launch_bounds__(256) k(float *out, int iters, float kk, float th) {
  __shared__ __attribute__((aligned(16))) int lds[8192 + 512];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192 + 512; i += 256) lds[i] = i < 8192 ? 0x22222222 : 0x3f800000 + i;
  __syncthreads();
  v16f fa = {0}, fb = {0};
  v8i b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  float u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) u[i] = 0.01f * lane + i;
  int xs[4] = {lane, lane + 1, lane + 2, lane + 3};
  float ty[2][2] = {{0, 0}, {0, 0}};
  float adr[16];            // TAB == 2: addresses that look like the accumulator of a sparse layer
#pragma unroll
  for (int i = 0; i < 16; ++i) adr[i] = __uint_as_float((uint32_t)(4 * ((lane * 7 + i * 13) % 97 + 200)));
  unsigned word = 0;
  v4i q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = v4i{0x22222222, 0, 0, 0};
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)lds + lane * 16;
  auto mfma = [&](v16f &f, int s) __attribute__((always_inline)) {
    v8i av = {q[s % 4].x, q[s % 4].y, q[s % 4].z, q[s % 4].w, 0, 0, 0, 0};
    if (s == 0) f = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, v16f{0}, 4, 2, 0, 127, 0, 127);
    else f = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, f, 4, 2, 0, 127, 0, 127);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(s + 3) % 4]) : "v"(addr), "n"(1024));
  };
  auto extra = [&](int n) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < n; ++i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(xs[i % 4]) : "v"(xs[i % 4]), "v"(lane));
  };
  for (int it = 0; it < iters; it += 2) {
    if (PHASED) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int s = 0; s < 18; ++s) { mfma(fa, s); asm volatile("s_waitcnt lgkmcnt(2)"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < 8; ++j) pair(u[2 * j], u[2 * j + 1], fa[2 * j], fa[2 * j + 1], kk, th, word);
        extra(NX);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // step A: MFMAs into fb beside the epilogue on fa; step B: the roles swapped
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        v16f &fn = half ? fa : fb;
        const v16f &fc = half ? fb : fa;
#pragma unroll
        for (int s = 0; s < 18; ++s) {
          mfma(fn, s);
          // 8 pairs over 18 slots: pair j in slot 2 j + 1 (a pair is 17 instructions: the
          // kernel spreads them more finely, 8-9 per slot; the totals are the same)
          if (TAB == 0) {
            if ((s & 1) && s / 2 < 8) pair(u[s - 1], u[s], fc[s - 1], fc[s], kk, th, word);
          } else if ((s & 1) && s / 2 < 8) {
            // the reads of pair j were issued in slot 2 j - 1 (pair 0: slot 0); LDS results return
            // in order: all but this slot's A read have arrived
            if (s == 1) pair_tab_issue<TAB == 1>(ty[0][0], ty[0][1], TAB == 1 ? fc[0] : adr[0], TAB == 1 ? fc[1] : adr[1], 0);
            const int j = s / 2;
            if (j + 1 < 8)
              pair_tab_issue<TAB == 1>(ty[(j + 1) & 1][0], ty[(j + 1) & 1][1], TAB == 1 ? fc[2 * j + 2] : adr[2 * j + 2],
                                       TAB == 1 ? fc[2 * j + 3] : adr[2 * j + 3], 0);
            asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(2));   // (j + 1 < 8: the two just issued stay out)
            __builtin_amdgcn_sched_barrier(0);
            pair_tab_use(u[s - 1], u[s], ty[j & 1][0], ty[j & 1][1], kk, th, word);
          }
          extra((NX * (s + 1)) / 18 - (NX * s) / 18);
          asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(TAB ? 4 : 2));
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(fn));       // (TAB == 2 does not read the accumulators: keep the chain)
      }
    }
  }
  float sacc = (float)word + xs[0] + xs[1] + xs[2] + xs[3];
  for (int i = 0; i < 16; ++i) sacc += u[i] + fa[i] + fb[i];
  for (int i = 0; i < 4; ++i) sacc += q[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

// The same tile-step on v_mfma_scale_f32_16x16x128_f8f6f4: a wave owns 16 channels x 64 pixels
// (four 16 x 16 tiles, nine dependent MFMAs of 16 cycles each per tile, chains interleaved), half
// the B fragments (54 VGPRs: a third wave per SIMD fits) and twice the A reads (36 per tile-step).
template <int NX>
__global__ void __launch_bounds__(256) k16(float *out, int iters, float kk, float th) {
  __shared__ __attribute__((aligned(16))) int lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0x22222222;
  __syncthreads();
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f fa[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, fb[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v8i b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
  float u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) u[i] = 0.01f * lane + i;
  int xs[4] = {lane, lane + 1, lane + 2, lane + 3};
  unsigned word = 0;
  v4i q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = v4i{0x22222222, 0, 0, 0};
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)lds + lane * 16;
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      v4f *fn = half ? fa : fb;
      const v4f *fc = half ? fb : fa;
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        const int m = s & 3;
        v8i av = {q[s % 4].x, q[s % 4].y, q[s % 4].z, q[s % 4].w, 0, 0, 0, 0};
        if (s < 4) fn[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, b8, v4f{0, 0, 0, 0}, 4, 2, 0, 127, 0, 127);
        else fn[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, b8, fn[m], 4, 2, 0, 127, 0, 127);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(s + 3) % 4]) : "v"(addr), "n"(1024));
        // 8 pairs over 36 slots: pair j in slot 4 j + 3
        if ((s & 3) == 3 && s / 4 < 8) {
          const int j = s / 4;
          float a0 = fc[j >> 1][2 * (j & 1)], a1 = fc[j >> 1][2 * (j & 1) + 1];
          pair(u[2 * j], u[2 * j + 1], a0, a1, kk, th, word);
        }
#pragma unroll
        for (int i = 0; i < (NX * (s + 1)) / 36 - (NX * s) / 36; ++i)
          asm volatile("v_add_u32 %0, %1, %2" : "=v"(xs[i % 4]) : "v"(xs[i % 4]), "v"(lane));
        asm volatile("s_waitcnt lgkmcnt(2)");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) asm volatile("" : "+v"(fn[m]));
    }
  }
  float sacc = (float)word + xs[0] + xs[1] + xs[2] + xs[3];
  for (int i = 0; i < 16; ++i) sacc += u[i];
  for (int m = 0; m < 4; ++m) for (int i = 0; i < 4; ++i) sacc += fa[m][i] + fb[m][i];
  for (int i = 0; i < 4; ++i) sacc += q[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

template <int NX>
void run16(float *out, int wps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NX>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double steps = (double)iters * wps;
  const double cyc = ms * 1e6 / steps * 2.4;
  printf("16x16x128    arith   extra %2d  waves/SIMD %d: %7.1f SIMD cycles per tile-step  (matrix pipe %4.1f %% busy)\n",
         NX, wps, cyc, 100.0 * 576.0 / cyc);
}

// Wave specialisation: a workgroup of 8 waves, waves 0-3 run only the matrix side of a tile-step
// (18 dependent MFMAs, a ds_read_b128 each, then the 16 accumulator registers written to LDS),
// waves 4-7 only the vector side (accumulators read back from LDS, the neuron epilogue, the
// staging stand-in), one s_barrier per tile-step between them (double-buffered hand-over).
// Two workgroups per CU: per SIMD two matrix waves and two vector waves.
template <int NX>
__global__ void __launch_bounds__(512) kspec(float *out, int iters, float kk, float th) {
  __shared__ __attribute__((aligned(16))) int lds[8192 + 2 * 4 * 1024];     // A image | 2 x 4 waves x 4 KiB of accumulators
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 0x22222222;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)lds;
  const uint32_t addr = base + lane * 16;
  const uint32_t xaddr = base + 32768 + (wave & 3) * 4096 + lane * 16;      // hand-over slot of the pair
  float sacc = 0;
  if (wave < 4) {
    v16f f = {0};
    v8i b8 = {0x08208208, lane & 7, 0, 0, 0, 0, 0, 0};
    v4i q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = v4i{0x22222222, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        v8i av = {q[s % 4].x, q[s % 4].y, q[s % 4].z, q[s % 4].w, 0, 0, 0, 0};
        if (s == 0) f = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, v16f{0}, 4, 2, 0, 127, 0, 127);
        else f = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b8, f, 4, 2, 0, 127, 0, 127);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[(s + 3) % 4]) : "v"(addr), "n"(1024));
        asm volatile("s_waitcnt lgkmcnt(2)");
        __builtin_amdgcn_sched_barrier(0);
      }
      const uint32_t xa = xaddr + (it & 1) * 16384;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(xa), "v"(v4i{__float_as_int(f[4 * g]), __float_as_int(f[4 * g + 1]), __float_as_int(f[4 * g + 2]), __float_as_int(f[4 * g + 3])}), "n"(0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
    }
    for (int i = 0; i < 16; ++i) sacc += f[i];
    for (int i = 0; i < 4; ++i) sacc += q[i].x;
  } else {
    float u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = 0.01f * lane + i;
    int xs[4] = {lane, lane + 1, lane + 2, lane + 3};
    unsigned word = 0;
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_barrier" : : : "memory");                  // the accumulators of step it are in slot it & 1
      const uint32_t xa = xaddr + (it & 1) * 16384;
      v4i a4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(a4[g]) : "v"(xa + 0 * g));
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int g = j >> 1, e = (j & 1) * 2;
        pair(u[2 * j], u[2 * j + 1], __int_as_float(a4[g][e]), __int_as_float(a4[g][e + 1]), kk, th, word);
#pragma unroll
        for (int i = 0; i < (NX * (j + 1)) / 8 - (NX * j) / 8; ++i)
          asm volatile("v_add_u32 %0, %1, %2" : "=v"(xs[i % 4]) : "v"(xs[i % 4]), "v"(lane));
      }
    }
    sacc = (float)word + xs[0] + xs[1] + xs[2] + xs[3];
    for (int i = 0; i < 16; ++i) sacc += u[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

template <int NX>
void run_spec(float *out, int wgs_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kspec<NX>), dim3(256 * wgs_per_cu), dim3(512), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  // a workgroup finishes 4 tile-steps per iteration on 4 SIMDs: one per SIMD
  const double steps = (double)iters * wgs_per_cu;
  const double cyc = ms * 1e6 / steps * 2.4;
  printf("specialised  arith   extra %2d  %d workgroup(s) of 4 + 4 waves per CU: %7.1f SIMD cycles per tile-step  (matrix pipe %4.1f %% busy)\n",
         NX, wgs_per_cu, cyc, 100.0 * 576.0 / cyc);
}

template <int PHASED, int NX, int TAB = 0>
void run(float *out, int wps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PHASED, NX, TAB>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.5f, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double steps = (double)iters * wps;      // tile-steps per SIMD
  const double cyc = ms * 1e6 / steps * 2.4;
  printf("%-12s %s extra %2d  waves/SIMD %d: %7.1f SIMD cycles per tile-step  (matrix pipe %4.1f %% busy)\n",
         PHASED ? "phased" : "interleaved", TAB == 0 ? "arith  " : TAB == 1 ? "table+1" : "table+0", NX, wps, cyc, 100.0 * 576.0 / cyc);
}

int main() {
  float *out; (void)hipMalloc(&out, 256 * 8 * 512 * 4);
  for (int w = 1; w <= 4; ++w) {
    run<0, 48>(out, w);
    run<1, 48>(out, w);
  }
  for (int w = 2; w <= 4; ++w) {
    run<0, 0>(out, w);
    run<1, 0>(out, w);
  }
  for (int w = 2; w <= 4; ++w) run16<48>(out, w);
  for (int w = 1; w <= 3; ++w) run_spec<48>(out, w);
  for (int w = 2; w <= 3; ++w) {
    run<0, 48, 0>(out, w);
    run<0, 48, 1>(out, w);
    run<0, 48, 2>(out, w);
  }
  return 0;
}
