// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with A = fp4 spikes and B = fp8 (e4m3) DIGITS of wide
// integer codes, code = 16 hi + lo: lo digits in K block 0 (lanes of half 0), hi digits in K block 1
// (lanes of half 1) with a block scale of 2^4 on B's second block -- does ONE instruction return
// sum(code * spike) exactly?  Also: which fp8 encoding (OCP e4m3 vs fnuz), which byte / lane the
// scale comes from.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_fp8_digits.hip -o /tmp/fp8d && /tmp/fp8d
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void one(const v8i *a, const v8i *b, const int *sa, const int *sb, float *d) {
  const int l = threadIdx.x;
  v16f c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 4 /*A fp4*/, 0 /*B fp8*/,
                                                      0, sa[l], 0, sb[l]);
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = c[i];
}

static uint32_t enc8_ocp(int v) {       // OCP e4m3 (bias 7)
  static const uint32_t tab[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
  return tab[abs(v)] | (v < 0 ? 0x80u : 0u);
}
static uint32_t enc8_fnuz(int v) {      // e4m3fnuz (bias 8)
  static const uint32_t tab[9] = {0x00, 0x40, 0x48, 0x4C, 0x50, 0x52, 0x54, 0x56, 0x58};
  return tab[abs(v)] | (v < 0 ? 0x80u : 0u);
}

int main() {
  srand(11);
  std::vector<int> S(32 * 9), Cd(9 * 32);              // spikes [row][tap], codes [tap][col]
  for (auto &x : S) x = (rand() % 100) < 40;
  for (auto &x : Cd) x = (rand() % 255) - 127;
  for (int enc = 0; enc < 2; ++enc)
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: scale of block 1 in B's lanes of half 1 (byte 0); 1: in every lane's byte 1; 2: A side
      std::vector<uint32_t> ha(64 * 8, 0), hb(64 * 8, 0);
      std::vector<int> sa(64, 127), sb(64, 127);
      for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        for (int tap = 0; tap < 9; ++tap) {
          ha[l * 8 + tap / 8] |= (S[r * 9 + tap] ? 0x2u : 0u) << (4 * (tap % 8));
          const int code = Cd[tap * 32 + r];
          const int lo = ((code + 8) & 15) - 8, hi = (code - lo) / 16;
          const uint32_t e = enc ? enc8_fnuz(h ? hi : lo) : enc8_ocp(h ? hi : lo);
          hb[l * 8 + tap / 4] |= e << (8 * (tap % 4));
        }
        if (mode == 0) sb[l] = h ? 131 : 127;
        if (mode == 1) sb[l] = 127 | (131 << 8);
        if (mode == 2) sa[l] = h ? 131 : 127;
      }
      v8i *da, *db; int *dsa, *dsb; float *dd;
      hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 16 * 4);
      hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
      hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
      hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
      hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
      std::vector<float> d(64 * 16);
      hipMemcpy(d.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
      int bad = 0, shown = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
          const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
          int ref = 0;
          for (int tap = 0; tap < 9; ++tap) ref += S[row * 9 + tap] * Cd[tap * 32 + col];
          if (d[l * 16 + i] != (float)ref) {
            ++bad;
            if (shown++ < 3) printf("   row %d col %d: got %g want %d\n", row, col, d[l * 16 + i], ref);
          }
        }
      printf("enc %s  scale mode %d: %d of 1024 wrong\n", enc ? "fnuz" : "ocp ", mode, bad);
    }
  return 0;
}
