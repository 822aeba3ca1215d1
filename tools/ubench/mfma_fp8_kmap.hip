// Probe: which k does (lane half, position) of an fp4 A operand and of an fp8 B operand of
// v_mfma_scale_f32_32x32x64_f8f6f4 address when the formats are mixed?  One non-zero in A (row 0)
// and one in B (column 0); D[0][0] != 0 marks the pairs that meet.  Then the block scale: which
// lane's / byte's scale multiplies which K block.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_fp8_kmap.hip -o /tmp/kmap && /tmp/kmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(float *out, int bfmt) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int pa = 0; pa < 64; ++pa)
    for (int pb = 0; pb < 64; ++pb) {
      v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
      const int ha = pa >> 5, ja = pa & 31, hb = pb >> 5, jb = pb & 31;
      if (r == 0 && h == ha) a[ja >> 3] = 0x2 << (4 * (ja & 7));          // fp4 1.0
      if (r == 0 && h == hb) b[jb >> 2] = 0x40 << (8 * (jb & 3));         // fp8 e4m3 2.0 (OCP) / 1.0 (fnuz)
      v16f c = {0};
      if (bfmt == 0)
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 0, 0, 127, 0, 127);
      else
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 1, 0, 127, 0, 127);
      if (l == 0) out[pa * 64 + pb] = c[0];
    }
}

__global__ void scale_probe(float *out) {
  // A: 1.0 at every k of row 0; B: 2.0 (fp8 0x40) at every k of column 0 -> per block 32 * 2 = 64.
  // scale_b varies by lane half / byte; D[0][0] tells which scale hit which block.
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r == 0) { for (int i = 0; i < 4; ++i) a[i] = 0x22222222; for (int i = 0; i < 8; ++i) b[i] = 0x40404040; }
  const int cases[6][2] = {{127, 127}, {127, 131}, {131, 127}, {127 | (131 << 8), 127 | (131 << 8)}, {131, 131}, {127 | (131 << 8), 131 | (127 << 8)}};
  for (int cs = 0; cs < 6; ++cs) {
    const int sb = h ? cases[cs][1] : cases[cs][0];
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 0, 0, 127, 0, sb);
    if (l == 0) out[cs] = c[0];
    v16f c2 = {0};
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 0, 0, sb, 0, 127);
    if (l == 0) out[8 + cs] = c2[0];
  }
}

int main() {
  float *d; hipMalloc(&d, 64 * 64 * 4);
  std::vector<float> m(64 * 64);
  for (int bf = 0; bf < 2; ++bf) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, bf);
    hipMemcpy(m.data(), d, 64 * 64 * 4, hipMemcpyDeviceToHost);
    printf("B format %s: value of the met pair %g; A position (half, j) -> B position it meets:\n", bf ? "bf8" : "fp8", 0.0);
    for (int pa = 0; pa < 64; ++pa) {
      int hit = -1, n = 0; float v = 0;
      for (int pb = 0; pb < 64; ++pb) if (m[pa * 64 + pb] != 0) { hit = pb; ++n; v = m[pa * 64 + pb]; }
      if (pa % 8 == 0 || n != 1) printf("  A(%d,%2d) -> B(%d,%2d)  n=%d v=%g\n", pa >> 5, pa & 31, hit >> 5, hit & 31, n, v);
    }
  }
  hipLaunchKernelGGL(scale_probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(m.data(), d, 16 * 4, hipMemcpyDeviceToHost);
  const char *names[6] = {"127/127", "h0 127, h1 131", "h0 131, h1 127", "bytes 127|131<<8 both", "131/131", "h0 127|131<<8, h1 131|127<<8"};
  for (int cs = 0; cs < 6; ++cs) printf("scale_b %-30s D = %g      as scale_a: D = %g\n", names[cs], m[cs], m[8 + cs]);
  return 0;
}
