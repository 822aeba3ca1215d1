// The event layer's tile-step, built up piece by piece at 4 waves per SIMD, to see which piece
// costs what beside the vector instructions of the neuron update (u8c2_epilogue_rate.hip: 245
// SIMD cycles per tile on their own):
//   0  update + threshold + reset + word                       (the vector instructions alone)
//   1  + the 16 table reads (ds_read_b32 at per-lane addresses spread over `SPREAD` entries,
//        all issued first, consumed pair by pair behind counted lgkmcnt waits)
//   2  + the MFMA that produces those addresses (A fragment from LDS by two ds_read_b64), so
//        read -> MFMA -> table read -> update is one dependent chain per tile, as in the kernel
//   3  = 2 with the table reads of tile t + 1 issued BEFORE the update of tile t (two tiles in
//        flight per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void update_pair(v2f &u, v2f x, v2f kv, float th, unsigned &word, int j) {
  v2f t;
  unsigned long long m0, m1;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "v"(u));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(kv), "v"(u));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(th), "v"(t.x));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(th), "v"(t.y));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u.x) : "v"(t.x), "s"(m0));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(u.y) : "v"(t.y), "s"(m1));
  const unsigned long long m = m0 | m1;
  const unsigned w = (unsigned)m | (unsigned)(m >> 32);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(w), "n"(0));
}

// two pairs at once, instruction by instruction: two independent dependency chains per wave
__device__ __forceinline__ void update_two(v2f &ua, v2f &ub, v2f xa, v2f xb, v2f kv, float th,
                                           unsigned &word) {
  v2f ta, tb;
  unsigned long long a0, a1, b0, b1;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(ta) : "v"(xa), "v"(ua));
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tb) : "v"(xb), "v"(ub));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(ta) : "v"(ta), "v"(kv), "v"(ua));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(tb) : "v"(tb), "v"(kv), "v"(ub));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(a0) : "v"(th), "v"(ta.x));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(a1) : "v"(th), "v"(ta.y));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(b0) : "v"(th), "v"(tb.x));
  asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(b1) : "v"(th), "v"(tb.y));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(ua.x) : "v"(ta.x), "s"(a0));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(ua.y) : "v"(ta.y), "s"(a1));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(ub.x) : "v"(tb.x), "s"(b0));
  asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(ub.y) : "v"(tb.y), "s"(b1));
  const unsigned long long ma = a0 | a1, mb = b0 | b1;
  const unsigned wa = (unsigned)ma | (unsigned)(ma >> 32), wb = (unsigned)mb | (unsigned)(mb >> 32);
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(wa), "n"(0));
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(wb), "n"(1));
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k(float *out, int iters, float kk, float th, int spread) {
  // one array: the table at LDS offset 0 (an accumulator IS a table address, as in the kernel,
  // where constant k rows add the table base inside the MFMA), the A image behind it
  __shared__ __attribute__((aligned(16))) float smem[4096 + 2048];
  float *table = smem;
  int *img = (int *)(smem + 4096);
  for (int i = threadIdx.x; i < 4096; i += 256) table[i] = 0.25f + (i & 7) * 0.125f;
  // A bytes 0..3 (or all 0 for spread 1): acc = 4 * sum of 32 bytes = a multiple of 4 below 400
  for (int i = threadIdx.x; i < 2048; i += 256)
    img[i] = spread > 1 ? (int)(((i * 2654435761u) >> 13) & 0x03030303u) : 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v2f u[2][8];
  for (int tl = 0; tl < 2; ++tl) for (int i = 0; i < 8; ++i) u[tl][i] = v2f{0.01f * lane, 0.02f * i};
  const v2f kv = {kk, kk};
  unsigned word = 0;
  // per-lane table addresses (bytes): a pseudo-random entry among `spread`, like accumulators
  int addr[16];
  for (int i = 0; i < 16; ++i) addr[i] = (int)(((lane * 37 + i * 11) % spread) * 4 + (lane & 31) * 0);
  const v4i bw = {0x04040404, 0x04040404, 0x04040404, 0x04040404};
  const unsigned tb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)table;   // 0
  const unsigned ib = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int *)img + lane * 8;
  v2f xn[8];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      v2f x[8];
      if (VARIANT == 0 || VARIANT == 5) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v2f{0.3f + j, 0.7f};
      } else {
        int a16[16];
        if (VARIANT >= 2) {
          v2i lo, hi;
          asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(ib + (unsigned)(tl * 512)));
          asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(hi) : "v"(ib + (unsigned)(tl * 512)));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{lo.x, lo.y, hi.x, hi.y}, bw, acc, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 16; ++i) a16[i] = acc[i];
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) a16[i] = addr[i];
        }
        float xr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(xr[i]) : "v"(tb + (unsigned)a16[i]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(14 - 2 * j) : "memory");
          x[j] = v2f{xr[2 * j], xr[2 * j + 1]};
          asm volatile("" : "+v"(x[j]));
        }
      }
      if (VARIANT == 5) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) update_two(u[tl][j], u[tl][j + 1], x[j], x[j + 1], kv, th, word);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) update_pair(u[tl][j], x[j], kv, th, word, j);
      }
    }
  }
  float s = (float)word;
  for (int tl = 0; tl < 2; ++tl) for (int i = 0; i < 8; ++i) s += u[tl][i].x + u[tl][i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char *name, float *out, int spread, int wps = 4) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 10000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.5f, 1.0f, spread);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double tiles = (double)iters * 2 * wps;    // tile-steps per SIMD (2 tiles per step, wps waves)
  printf("%-46s spread %4d, %d waves/SIMD: %.1f SIMD cycles per tile (2.4 GHz)\n", name, spread, wps,
         ms * 1e6 / tiles * 2.4);
}
int main() {
  float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0>("vector instructions alone", out, 1);
  run<1>("+ 16 table reads, one entry (broadcast)", out, 1);
  run<1>("+ 16 table reads, 253 entries", out, 253);
  run<2>("+ A reads + MFMA -> table addresses", out, 1);
  run<2>("+ A reads + MFMA -> table addresses", out, 253);
  for (int w = 2; w <= 8; ++w) run<2>("whole tile-step (conflict-free table)", out, 1, w);
  for (int w = 2; w <= 8; w += 2) run<0>("vector instructions alone", out, 1, w);
  for (int w = 2; w <= 8; w += 2) run<5>("vector instructions, two pairs interleaved", out, 1, w);
  return 0;
}
