// Float32 connection (no neuron) on the f32 MFMA: the `fseq` contract of the layer API
// -- y[m][n] = fmaf chain over k ascending of x[m][k] * w[k][n], k = (kh, kw, cin) for
// convolutions -- for the real-valued activations of the reference's TCJA blocks
// (gate * spikes into conv_t and dense1, examples/tcja/models.py:149-187, 200-208).
// Replaces lax.conv_general_dilated (flax_qconv.py:158-168) / lax.dot_general
// (flax_qdense.py:87-89) for float32 inputs and fake-quantised float32 kernels.
//
// v_mfma_f32_32x32x2_f32 accumulates its two k products into C in ascending k order with
// one rounding each -- a chain of these MFMAs IS the fmaf chain (tools/ubench/
// mfma_f32_order.hip: 0 mismatches of 1024 against the host fmaf chain, ~70 % against any
// other order), so the matrix cores run the contract bit for bit.
//
// GEMM M x N x K with M = images x OH x OW output pixels, N = Cout, K = KH KW Cin.  A
// workgroup (4 waves) owns a 128 x 128 tile, wave w the 64 x 64 quadrant (w >> 1, w & 1)
// = 2 x 2 MFMA tiles.  K is walked in chunks of 16: the A chunk is gathered from the NHWC
// input (runs of 4 consecutive channels never straddle a tap because Cin % 4 == 0; zero for
// padding) and kept k-major in LDS, the B chunk is 16 rows of the [K][N] kernel;
// register-staged one chunk ahead.  Shapes: stride-1, undilated, ungrouped convolutions
// whose output has the input's size (3x3 pad 1, the 1-D k = 4 SAME convolutions of the
// TCJA gate, ...) and dense layers (1x1 on a 1x1 image); everything else stays on the
// direct-form kernel.
#include "kernels.h"

namespace snnqp {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int FG_KC = 16;
// TM: 32 x 32 MFMA tiles per wave and direction.  2: a workgroup owns 128 x 128 outputs (the
// throughput shape: every operand read from LDS feeds two MFMAs); 1: 64 x 64 -- for problems of a
// few hundred rows (config C1: 320 x 512 outputs are 12 workgroups of the large tile on 256 CUs,
// and K is one dependent chain per output whatever the tile): four times the workgroups, a
// quarter of the chain work per wave.
template <int TM> struct FgTile {
  static constexpr int BM = 64 * TM, BN = 64 * TM;
  static constexpr int LD = BM + 32;     // LDS row stride (floats) of one k: the two lane halves
                                         // (k, k + 1) of an MFMA operand read disjoint banks
  static constexpr int EPT = 4 * TM;     // elements of a k chunk a thread stages per operand
};

struct FseqGemmArgs {
  const void *x;        // [NB][H][W][Cin] float32 / uint8, or [NB][H][W][Cin/32] spike words
  const float *w;       // [K][N]
  float *y;             // [M][N]
  int64_t M;
  int32_t N, K, H, W, Cin, KH, KW, pad_h, pad_w;
};

// Input element types: float32, or integer-typed activations (uint8 counts, bit-packed
// spikes) widened to float32 on the fly -- an unquantised (float32) kernel fed by spikes
// runs the same fmaf chain, with x in {0, 1, 2, ...} exactly.
// MODE 0: float32, Cin % 16 == 0 (a k chunk lies in one tap: one address computation)
//      1: float32, Cin % 4 == 0 (runs of 4 channels)
//      2: any type and Cin, element by element (the 2-channel event input; uint8; bits)
//      3: uint8 / spike words, one tap per chunk -- a 1 x 1 kernel (dense layers: k = channel, the
//         pixel is the row) or Cin % 16 == 0: four or eight bytes / bits per load, at most one tap
//         decode per chunk -- unquantised blocks on integer-typed activations (config C1, the
//         float32 baseline nets) spent their time in MODE 2's two divisions per element
template <int IN> __device__ __forceinline__ float fg_load(const void *x, int64_t pix, int c, int Cin) {
  if (IN == SNNQP_F32) return ((const float *)x)[pix * Cin + c];
  if (IN == SNNQP_U8) return (float)((const uint8_t *)x)[pix * Cin + c];
  const uint32_t w = ((const uint32_t *)x)[pix * ((Cin + 31) >> 5) + (c >> 5)];
  return (float)((w >> (c & 31)) & 1u);
}

template <int MODE, int IN, int TM = 2>
__global__ void __launch_bounds__(256)
fseq_gemm_kernel(FseqGemmArgs a) {
  constexpr int FG_BM = FgTile<TM>::BM, FG_BN = FgTile<TM>::BN, FG_LD = FgTile<TM>::LD, EPT = FgTile<TM>::EPT;
  __shared__ __attribute__((aligned(16))) float lds[2][2][FG_KC * FG_LD];   // [buf][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * FG_BM;
  const int n0 = blockIdx.y * FG_BN;
  const int wm = (wave >> 1) * 32 * TM, wn = (wave & 1) * 32 * TM;

  // staging tasks.  A: row ar of the tile, EPT consecutive k (TM float4); B: k row bk,
  // EPT consecutive columns
  constexpr int TPR = FG_KC / EPT;           // threads per A row
  const int ar = tid / TPR, ak = (tid % TPR) * EPT;
  const int64_t am = m0 + ar;
  const bool arow = am < a.M;
  int oy = 0, ox = 0;
  int64_t img = 0;
  if (arow) {
    const int64_t hw = (int64_t)a.H * a.W;
    img = am / hw;
    const int p = (int)(am - img * hw);
    oy = p / a.W;
    ox = p - oy * a.W;
  }
  const int bk = tid >> 4, bc = (tid & 15) * EPT;
  v4f ra[TM], rb[TM];
  auto load_chunk = [&](int kc) {          // global -> registers
    const int k0 = kc * FG_KC;
    if (MODE == 0) {
      const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      const int iy = oy + kh - a.pad_h, ix = ox + kw - a.pad_w;
      const bool ok = arow && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float *src = (const float *)a.x + ((img * a.H + iy) * a.W + ix) * a.Cin + c0 + ak;
#pragma unroll
      for (int j = 0; j < TM; ++j) ra[j] = ok ? *(const v4f *)(src + 4 * j) : v4f{0.f, 0.f, 0.f, 0.f};
    } else if (MODE == 3) {
      // one tap per chunk: a 1 x 1 kernel (k = channel, the pixel is the row; any Cin), or
      // Cin % 16 == 0 (a chunk of 16 k lies in one tap, as MODE 0)
      int64_t pix = am;
      int kk = k0 + ak;                          // channel of the thread's first element, a multiple of EPT
      bool ok = arow && kk < a.K;
      if (!(a.KH == 1 && a.KW == 1)) {
        const int tap = k0 / a.Cin;
        kk = k0 - tap * a.Cin + ak;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int iy = oy + kh - a.pad_h, ix = ox + kw - a.pad_w;
        ok = arow && k0 < a.K && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pix = (img * a.H + iy) * a.W + ix;
      }
      uint32_t q[TM];                            // EPT values, a byte each
#pragma unroll
      for (int j = 0; j < TM; ++j) q[j] = 0;
      if (ok) {
        if (IN == SNNQP_U8) {
#pragma unroll
          for (int j = 0; j < TM; ++j) q[j] = *(const uint32_t *)((const uint8_t *)a.x + pix * a.Cin + kk + 4 * j);
        } else {
          const uint32_t w = ((const uint32_t *)a.x)[pix * ((a.Cin + 31) >> 5) + (kk >> 5)] >> (kk & 31);
#pragma unroll
          for (int j = 0; j < TM; ++j) q[j] = (((w >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u;
        }
      }
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[j][e] = (float)((q[j] >> (8 * e)) & 0xFFu);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = k0 + ak + 4 * j + e;
          const int tap = kk / a.Cin, c = kk - tap * a.Cin;
          const int kh = tap / a.KW, kw = tap - kh * a.KW;
          const int iy = oy + kh - a.pad_h, ix = ox + kw - a.pad_w;
          const bool ok = arow && kk < a.K && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          ra[j][e] = ok ? fg_load<IN>(a.x, (img * a.H + iy) * a.W + ix, c, a.Cin) : 0.0f;
        }
    } else
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int kk = k0 + ak + 4 * j;          // 4 consecutive k = 4 channels of one tap
      const int tap = kk / a.Cin, c0 = kk - tap * a.Cin;
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      const int iy = oy + kh - a.pad_h, ix = ox + kw - a.pad_w;
      const bool ok = arow && kk < a.K && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      ra[j] = ok ? *(const v4f *)((const float *)a.x + ((img * a.H + iy) * a.W + ix) * a.Cin + c0)
                 : v4f{0.f, 0.f, 0.f, 0.f};
    }
    const int kb = k0 + bk;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int col = n0 + bc + 4 * j;
      v4f v = {0.f, 0.f, 0.f, 0.f};
      if (kb < a.K) {
        const float *wr = a.w + (int64_t)kb * a.N + col;
        if (col + 3 < a.N && (a.N & 3) == 0) {
          v = *(const v4f *)wr;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < a.N) v[e] = wr[e];
        }
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {        // registers -> LDS (A transposed to k-major)
    float *la = lds[buf][0], *lb = lds[buf][1];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) la[(ak + 4 * j + e) * FG_LD + ar] = ra[j][e];
      *(v4f *)(lb + bk * FG_LD + bc + 4 * j) = rb[j];
    }
  };

  v16f acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nchunks = (a.K + FG_KC - 1) / FG_KC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) load_chunk(kc + 1);
    const float *la = lds[buf][0], *lb = lds[buf][1];
#pragma unroll
    for (int s = 0; s < FG_KC / 2; ++s) {          // k pairs, ascending
      float av[TM], bv[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        av[i] = la[(2 * s + h) * FG_LD + wm + 32 * i + r];
        bv[i] = lb[(2 * s + h) * FG_LD + wn + 32 * i + r];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // C/D layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int col = n0 + wn + 32 * j + r;
      if (col >= a.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.M) a.y[row * a.N + col] = acc[i][j][e];
      }
    }
}

// nullptr when the kernel serves the request, else the reason
const char *fseq_gemm_unsupported(int in_type, const snnqp_conv_geom_t *g,
                                  const snnqp_weight_t *w) {
  if (w->wtype != SNNQP_W_F32) return "kernel is not float32";
  if (in_type != SNNQP_F32 && in_type != SNNQP_U8 && in_type != SNNQP_BITS) return "input type";
  if (g->groups != 1) return "grouped";
  if (g->stride_h != 1 || g->stride_w != 1) return "strided";
  if (g->in_dil_h != 1 || g->in_dil_w != 1 || g->k_dil_h != 1 || g->k_dil_w != 1) return "dilated";
  if (g->pad_h_lo + g->pad_h_hi != g->KH - 1 || g->pad_w_lo + g->pad_w_hi != g->KW - 1)
    return "output size differs from input size";
  return nullptr;
}

int run_fseq_gemm(const void *x, int in_type, int64_t NB, const snnqp_conv_geom_t *g,
                  const snnqp_weight_t *w, float *y, hipStream_t st) {
  FseqGemmArgs a;
  a.x = x; a.w = (const float *)w->w; a.y = y;
  a.M = NB * g->H * g->W;
  a.N = g->Cout; a.K = g->KH * g->KW * g->Cin;
  a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.KH = g->KH; a.KW = g->KW;
  a.pad_h = g->pad_h_lo; a.pad_w = g->pad_w_lo;
  if (a.M == 0 || a.N == 0) return SNNQP_OK;
  // the small tile when the large one would leave most of the chip idle
  const int64_t big = ceil_div64(a.M, 128) * ((a.N + 127) / 128);
  // ... or when at most 64 columns exist (the TCJA gate's convolution along the channels has
  // N = T = 20 outputs: three quarters of a 128-column tile would multiply zeros)
  const int tm = (big < 192 || a.N <= 64) ? 1 : 2;
  const int64_t gx = ceil_div64(a.M, 64 * tm);
  SNNQP_REQUIRE(gx < (1ll << 31), SNNQP_EINVAL, "fseq gemm: grid too large");
  const dim3 grid((unsigned)gx, (unsigned)((a.N + 64 * tm - 1) / (64 * tm)));
  const bool one = a.KH == 1 && a.KW == 1 && a.pad_h == 0 && a.pad_w == 0;    // k = channel, pixel = row
#define SNNQP_FG_LAUNCH(MODE, IN)                                                                  \
  do {                                                                                             \
    if (tm == 1) hipLaunchKernelGGL((fseq_gemm_kernel<MODE, IN, 1>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((fseq_gemm_kernel<MODE, IN, 2>), grid, dim3(256), 0, st, a);           \
  } while (0)
  const bool tapwise = a.Cin % FG_KC == 0;     // a k chunk lies in one tap
  if (in_type == SNNQP_U8 && ((one && a.Cin % 8 == 0) || tapwise) && ((uintptr_t)x & 7) == 0) SNNQP_FG_LAUNCH(3, SNNQP_U8);
  else if (in_type == SNNQP_BITS && (one || tapwise)) SNNQP_FG_LAUNCH(3, SNNQP_BITS);
  else if (in_type == SNNQP_U8) SNNQP_FG_LAUNCH(2, SNNQP_U8);
  else if (in_type == SNNQP_BITS) SNNQP_FG_LAUNCH(2, SNNQP_BITS);
  else if (a.Cin % FG_KC == 0) SNNQP_FG_LAUNCH(0, SNNQP_F32);
  else if (a.Cin % 4 == 0) SNNQP_FG_LAUNCH(1, SNNQP_F32);
  else SNNQP_FG_LAUNCH(2, SNNQP_F32);
#undef SNNQP_FG_LAUNCH
  SNNQP_CHECK_LAUNCH("fseq_gemm_kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
