// Probe: is a chain of v_mfma_f32_32x32x2_f32 (and 16x16x4) bit-identical to a float32
// fmaf chain over k ascending?  Random operands of mixed magnitude; compares with the
// host fmaf chain in k order (and, for orientation, with k pairs swapped).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int K = 256;

__global__ void k32(const float *A, const float *B, float *D) {   // A [32][K], B [K][32]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v16f c = {0};
  for (int k = 0; k < K; k += 2)
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

__global__ void k16(const float *A, const float *B, float *D) {   // A [16][K], B [K][16]
  const int l = threadIdx.x, r = l & 15, h = l >> 4;
  v4f c = {0};
  for (int k = 0; k < K; k += 4)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + h], B[(k + h) * 16 + r], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}

int main() {
  std::vector<float> A(32 * K), B(K * 32);
  srand(11);
  auto rnd = [] { return (float)((rand() % 2001 - 1000) / 1000.0) * powf(2.0f, (float)(rand() % 9 - 4)); };
  for (auto &x : A) x = rnd();
  for (auto &x : B) x = rnd();
  float *dA, *dB, *dD;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, 64 * 16 * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> D(64 * 16);
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
  int bad_seq = 0, bad_swap = 0, bad_nofma = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 16; ++i) {
      const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
      float s = 0, s2 = 0, s3 = 0;
      for (int k = 0; k < K; ++k) s = fmaf(A[row * K + k], B[k * 32 + col], s);
      for (int k = 0; k < K; k += 2) {
        s2 = fmaf(A[row * K + k + 1], B[(k + 1) * 32 + col], s2);
        s2 = fmaf(A[row * K + k], B[k * 32 + col], s2);
      }
      for (int k = 0; k < K; ++k) { volatile float p = A[row * K + k] * B[k * 32 + col]; s3 = s3 + p; }
      bad_seq += D[l * 16 + i] != s;
      bad_swap += D[l * 16 + i] != s2;
      bad_nofma += D[l * 16 + i] != s3;
    }
  printf("32x32x2 f32 chain, K = %d: mismatches vs fmaf k-ascending %d / 1024, vs pair-swapped %d, vs mul+add %d\n",
         K, bad_seq, bad_swap, bad_nofma);
  // 16x16x4: A rows 0..15, B cols 0..15 of the same data (B stride 32 -> repack)
  std::vector<float> B16(K * 16);
  for (int k = 0; k < K; ++k) for (int c = 0; c < 16; ++c) B16[k * 16 + c] = B[k * 32 + c];
  (void)hipMemcpy(dB, B16.data(), B16.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(D.data(), dD, 64 * 4 * 4, hipMemcpyDeviceToHost);
  int bad16 = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int col = l & 15, row = (l >> 4) * 4 + i;
      float s = 0;
      for (int k = 0; k < K; ++k) s = fmaf(A[row * K + k], B16[k * 16 + col], s);
      bad16 += D[l * 4 + i] != s;
    }
  printf("16x16x4 f32 chain: mismatches vs fmaf k-ascending %d / 256\n", bad16);
  return 0;
}
