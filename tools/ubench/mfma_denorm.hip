// Does v_mfma_scale_f32_32x32x64_f8f6f4 accumulate exactly in the float32 DENORMAL range?
// With the block scales chosen so that one code unit is 2^-147 (= 4 denormal steps), the bit
// pattern of the accumulator is 4 * (integer sum): a byte offset into a table of 4-byte entries,
// with no conversion instruction.  A = all 1.0 (fp4), B = fp6 codes -7..7 with known column sums,
// chains of 18 MFMAs, C preloaded with 4 * OFF steps.  Prints mismatches against the integer
// sums, and the time of the chain against the same chain at scale 1.
//   hipcc --offload-arch=gfx950 -O3 mfma_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int CH = 18;

template <int SA, int SB>
__global__ void __launch_bounds__(64) k(const int *b6, float cinit, float *out, int iters) {
  const int lane = threadIdx.x;
  v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
  v8i b[CH];
  for (int c = 0; c < CH; ++c) {
    b[c] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < 6; ++r) b[c][r] = b6[(c * 64 + lane) * 6 + r];
  }
  v16f acc;
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 16; ++i) acc[i] = cinit;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b[c], acc, 4, 2, 0, SA, 0, SB);
    if (it + 1 < iters) asm volatile("" : "+v"(acc));
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 64 + lane) * 16 + i] = acc[i];
}

static int enc6(int v) {
  static const int mag[8] = {0, 8, 16, 20, 24, 26, 28, 30};
  return (v < 0 ? 32 : 0) | mag[abs(v)];
}

int main() {
  const int OFF = 2048;
  std::vector<int> b6(CH * 64 * 6, 0);
  std::vector<long> colsum(32, 0);
  srand(7);
  for (int c = 0; c < CH; ++c)
    for (int lane = 0; lane < 64; ++lane) {
      unsigned long long lo = 0, mid = 0, hi = 0;   // 192-bit stream
      unsigned char bytes[24]; memset(bytes, 0, 24);
      for (int j = 0; j < 32; ++j) {
        int v = (rand() % 10 == 0) ? (rand() % 15 - 7) : 0;          // 90 % pruned
        if (lane % 32 == 5) v = 7;                                      // one column at the bound
        if (lane % 32 == 6) v = -7;
        colsum[lane & 31] += v;
        const int e = enc6(v), bit = 6 * j;
        for (int q = 0; q < 6; ++q) if ((e >> q) & 1) bytes[(bit + q) >> 3] |= 1 << ((bit + q) & 7);
      }
      memcpy(&b6[(c * 64 + lane) * 6], bytes, 24);
      (void)lo; (void)mid; (void)hi;
    }
  int *db; float *dout;
  (void)hipMalloc(&db, b6.size() * 4); (void)hipMalloc(&dout, 1024 * 64 * 16 * 4);
  (void)hipMemcpy(db, b6.data(), b6.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> out(64 * 16);
  // exactness: scale 2^-147 per unit, C = 4 * OFF denormal steps
  float cinit; { unsigned bits = 4u * OFF; memcpy(&cinit, &bits, 4); }
  hipLaunchKernelGGL((k<0, 107>), dim3(1), dim3(64), 0, 0, db, cinit, dout, 1);
  (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 16; ++i) {
      unsigned bits; memcpy(&bits, &out[lane * 16 + i], 4);
      const long want = 4 * (colsum[lane & 31] + OFF);
      if ((long)bits != want) { if (bad < 6) printf("lane %d reg %d: bits %u want %ld\n", lane, i, bits, want); ++bad; }
    }
  printf("denormal accumulate (scale 2^-147, C = %d steps): %d mismatches of 1024; column sums %ld .. %ld\n",
         4 * OFF, bad, colsum[6], colsum[5]);
  // the same at scale 1 with the magic constant 1.5 * 2^21 (ulp 0.25: bits = 0x4A400000 + 4 n)
  hipLaunchKernelGGL((k<127, 127>), dim3(1), dim3(64), 0, 0, db, 3145728.0f, dout, 1);
  (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 16; ++i) {
      unsigned bits; memcpy(&bits, &out[lane * 16 + i], 4);
      const long want = 0x4A400000L + 4 * colsum[lane & 31];
      if ((long)bits != want) { if (bad < 6) printf("lane %d reg %d: bits %x want %lx\n", lane, i, bits, want); ++bad; }
    }
  printf("magic-constant accumulate (scale 1, C = 1.5 * 2^21): %d mismatches of 1024\n", bad);
  // speed: denormal range against normal range
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL((k<0, 107>), dim3(1024), dim3(64), 0, 0, db, cinit, dout, 2000);
      else hipLaunchKernelGGL((k<127, 127>), dim3(1024), dim3(64), 0, 0, db, 3145728.0f, dout, 2000);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s: %.3f ms for 1024 waves x 2000 chains of %d MFMAs\n", mode == 0 ? "denormal range" : "normal range  ", ms, CH);
  }
  return 0;
}
