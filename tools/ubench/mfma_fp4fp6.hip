// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with A = fp4 (e2m1) spikes and
// B = fp6 (e2m3) integer weight codes: operand layout, exactness, rate.
// Hypothesis: lane (r = l & 31, h = l >> 5) holds k = 32 h + j, j = 0..31, packed
// little-endian (fp4: nibble j of 4 dwords; fp6: bits [6j, 6j+6) of 6 dwords).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void one(const v8i *a, const v8i *b, float *d) {
  const int l = threadIdx.x;
  v16f c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 4 /*A fp4*/, 2 /*B fp6*/,
                                                      0, 127, 0, 127);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}

template <int AF, int BF>
__global__ void __launch_bounds__(256) rate(const v8i *a, const v8i *b, float *d, int iters) {
  const int l = threadIdx.x & 63;
  v8i av = a[l], bv = b[l];
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c0, AF, BF, 0, 127, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c1, AF, BF, 0, 127, 0, 127);
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c2, AF, BF, 0, 127, 0, 127);
    c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c3, AF, BF, 0, 127, 0, 127);
  }
  d[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// one dependent chain (what a single wave in its MFMA phase issues)
template <int AF, int BF, int NACC>
__global__ void __launch_bounds__(256) chain(const v8i *a, const v8i *b, float *d, int iters) {
  const int l = threadIdx.x & 63;
  v8i av = a[l], bv = b[l];
  v16f c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = v16f{0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c[j % NACC] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c[j % NACC], AF, BF, 0, 127, 0, 127);
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i][i];
  d[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void scaled(const v8i *a, const v8i *b, float *d) {
  const int l = threadIdx.x;
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 1000.0f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 4, 2, 0, 129, 0, 127);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}

static int fp6_of(int v) {       // e2m3 code of integer v in [-7, 7]
  static const int mag[8] = {0x00, 0x08, 0x10, 0x14, 0x18, 0x1A, 0x1C, 0x1E};
  return (v < 0 ? 0x20 : 0) | mag[abs(v)];
}

int main() {
  std::vector<int> A(32 * 64), B(64 * 32);
  srand(7);
  for (auto &x : A) x = (rand() % 100) < 30;            // spikes
  for (auto &x : B) x = (rand() % 15) - 7;              // codes
  std::vector<uint32_t> ha(64 * 8, 0), hb(64 * 8, 0);
  for (int l = 0; l < 64; ++l) {
    const int r = l & 31, h = l >> 5;
    for (int j = 0; j < 32; ++j) {
      const int k = 32 * h + j;
      const uint32_t a4 = A[r * 64 + k] ? 0x2 : 0x0;       // 1.0 in e2m1
      ha[l * 8 + j / 8] |= a4 << (4 * (j % 8));
      const uint64_t b6 = (uint64_t)fp6_of(B[k * 32 + r]);
      const int bit = 6 * j;
      hb[l * 8 + bit / 32] |= (uint32_t)(b6 << (bit % 32));
      if (bit % 32 > 26) hb[l * 8 + bit / 32 + 1] |= (uint32_t)(b6 >> (32 - bit % 32));
    }
  }
  v8i *da, *db; float *dd;
  (void)hipMalloc(&da, 64 * 32); (void)hipMalloc(&db, 64 * 32); (void)hipMalloc(&dd, 256 * 256 * 4);
  (void)hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<float> hd(64 * 16);
  (void)hipMemcpy(hd.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 16; ++i) {
      const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += A[row * 64 + k] * B[k * 32 + col];
      if (hd[l * 16 + i] != (float)ref) {
        if (bad < 8) printf("mismatch lane %d reg %d: got %g want %d\n", l, i, hd[l * 16 + i], ref);
        ++bad;
      }
    }
  printf("layout check (A fp4 / B fp6, k = 32h + j): %d mismatches of 1024\n", bad);

  {
    std::vector<float> h1(64 * 16);
    hipLaunchKernelGGL(scaled, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(h1.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad4 = 0;
    for (int i = 0; i < 64 * 16; ++i) bad4 += h1[i] != 1000.0f + 4.0f * hd[i];
    printf("scale_a = 2^2 with C = 1000: %d mismatches (want 1000 + 4 * acc)\n", bad4);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  auto time = [&](auto kern, const char *name) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, da, db, dd, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-22s %.3f ms: %.1f ns per MFMA per SIMD, %.2f PFLOP/s chip\n", name, ms,
           ms * 1e6 / (iters * 4.0), 256.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e15);
  };
  time(chain<4, 2, 1>, "fp4xfp6 1 chain");
  time(chain<4, 2, 2>, "fp4xfp6 2 chains");
  time(chain<4, 4, 1>, "fp4xfp4 1 chain");
  time(rate<4, 2>, "A fp4 x B fp6");
  time(rate<4, 4>, "A fp4 x B fp4");
  time(rate<2, 2>, "A fp6 x B fp6");
  time(rate<0, 0>, "A fp8 x B fp8");
  time(rate<4, 0>, "A fp4 x B fp8");
  return 0;
}
