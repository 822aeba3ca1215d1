// QuantDense on the channel-major flattening of (gate x spike raster): the first dense block of
// the reference's full model behind its second TCJA gate (examples/tcja/models.py:97 ->
// :189-190 -> :200-216; x[k] = gate[c] * s[c, h, w], k = (c H + h) W + w, into QuantDense,
// flax_qdense.py:87-89), in the 'gint' contract:
//     I[c, o] = sum over the H W positions of code[(c, h, w), o] * s[h, w, c]      (exact integer)
//     acc[o]  = fmaf(gate[c], I[c, o], acc[o])   for c = 0 .. C - 1                 (start +0)
//     current = fl(fl(acc / L) * m)
// As conv_gated.hip, transposed: the matrix instruction's rows are OUTPUTS and its columns IMAGES
// (a row of the dense layer's input), because here every image has gates of its own -- in the C/D
// layout a lane is a column, so gate[image][c] is a per-lane operand of the fmaf.  One
// v_mfma_scale_f32_32x32x64_f8f6f4 per (32 outputs, channel, 32 images) with K = the H W positions
// of that channel (fp6 codes as A, fp4 spikes as B).  The float32 connection this replaces is a
// K = C H W GEMM on the f32 MFMA (0.45 ms for 2048 -> 512 at 20 480 rows); this one does C fmaf per
// output instead of C H W.
//
// A workgroup owns 32 images and up to 512 outputs (4 waves x 4 output tiles): the spike bits of
// its images are transposed once into LDS as B-operand dwords (Bt[c][image] = the fp4 nibbles of
// the positions of channel c) beside the gates (Gt[c][image]); the fp6 codes come pre-packed per
// (channel, 32 outputs).
#include <type_traits>

#include "kernels.h"

namespace snnqp {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

constexpr int DG_CMAX = 128;          // input channels
constexpr int DG_PMAX = 16;           // positions per channel (H W)

struct DenseGatedArgs {
  const uint32_t *s;      // [NB][HW][CW] spike words
  const float *gate;      // [NB][C]
  const uint32_t *ap;     // packed codes [C][OT][64 lanes][4 dwords] (pack_codes_dense_gated_kernel)
  float *y;               // [NB][N]
  int64_t NB;
  int32_t HW, C, CW, N, OT;
  float L, m;
};

__device__ __forceinline__ uint32_t dg_enc6(int v) {      // integer -7..7 -> e2m3 (runtime.hip enc6)
  const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
  const uint32_t code = mag == 0 ? 0u : mag == 1 ? 8u : mag == 2 ? 16u : mag == 3 ? 20u : mag == 4 ? 24u
                        : mag == 5 ? 26u : mag == 6 ? 28u : 30u;
  return code | (v < 0 ? 32u : 0u);
}

}  // namespace

// codes int8 [C * HW][N] (k = c HW + p) -> ap[c][ot][lane][4]: lane (o, h = 0) holds the fp6 values
// of positions 0 .. HW - 1 at bits [6 p, 6 p + 6) of dwords 0..2; lanes of half 1 and outputs
// beyond N are zero.
__global__ void __launch_bounds__(256)
pack_codes_dense_gated_kernel(const int8_t *w, int32_t C, int32_t HW, int32_t N, int32_t OT, uint32_t *ap) {
  const int64_t total = (int64_t)C * OT * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), n = lane & 31, h = lane >> 5;
    const int ot = (int)((i >> 6) % OT), c = (int)((i >> 6) / OT);
    const int o = ot * 32 + n;
    uint32_t d[4] = {0, 0, 0, 0};
    if (o < N && h == 0) {
      for (int p = 0; p < HW; ++p) {
        const uint32_t e = dg_enc6(w[((int64_t)c * HW + p) * N + o]);
        const int bit = 6 * p;
        d[bit >> 5] |= e << (bit & 31);
        if ((bit & 31) > 26) d[(bit >> 5) + 1] |= e >> (32 - (bit & 31));
      }
    }
    for (int j = 0; j < 4; ++j) ap[i * 4 + j] = d[j];
  }
}

// Codes beyond e2m3 (|code| <= 127) as two six-bit digits in the e3m2 format, code = 16 hi + lo
// (conv_gated.hip: one digit per K block, the 2^4 block scale on the second): lane (o, h = 0) holds
// the lo digits of the sixteen positions at bits [6 p, 6 p + 6) of dwords 0..2, lane (o, h = 1) the
// hi digits; the spike operand holds the positions in both lane halves.  Same record as the narrow
// form: ap[c][ot][lane][4 dwords].
__device__ __forceinline__ uint32_t dg_enc6w(int v) {     // integer -8..8 -> fp6 e3m2
  const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
  const uint32_t tab[9] = {0x00u, 0x0Cu, 0x10u, 0x12u, 0x14u, 0x15u, 0x16u, 0x17u, 0x18u};
  return tab[mag] | (v < 0 ? 0x20u : 0u);
}
__global__ void __launch_bounds__(256)
pack_codes_dense_gated_wide_kernel(const int8_t *w, int32_t C, int32_t HW, int32_t N, int32_t OT, uint32_t *ap) {
  const int64_t total = (int64_t)C * OT * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), n = lane & 31, h = lane >> 5;
    const int ot = (int)((i >> 6) % OT), c = (int)((i >> 6) / OT);
    const int o = ot * 32 + n;
    uint32_t d[4] = {0, 0, 0, 0};
    if (o < N) {
      for (int p = 0; p < HW; ++p) {
        const int code = w[((int64_t)c * HW + p) * N + o];
        const int lo = ((code + 8) & 15) - 8, hi = (code - lo) / 16;
        const uint32_t e = dg_enc6w(h ? hi : lo);
        const int bit = 6 * p;
        d[bit >> 5] |= e << (bit & 31);
        if ((bit & 31) > 26) d[(bit >> 5) + 1] |= e >> (32 - (bit & 31));
      }
    }
    for (int j = 0; j < 4; ++j) ap[i * 4 + j] = d[j];
  }
}

template <bool WIDE>
__global__ void __launch_bounds__(256, 2)
dense_gated_kernel(DenseGatedArgs a) {
  __shared__ u2 Bt[DG_CMAX][32];
  __shared__ float Gt[DG_CMAX][32];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t img0 = (int64_t)blockIdx.x * 32;

  // ---- the tile's spike bits as B-operand dwords, its gates ---------------------------------
  {
    const int img = tid & 31, wg = (tid >> 5) & 3, half = tid >> 7;       // 32 x 4 x 2 threads
    const int64_t gi = img0 + img;
    uint32_t w[DG_PMAX];
#pragma unroll
    for (int p = 0; p < DG_PMAX; ++p)
      w[p] = (gi < a.NB && p < a.HW && wg < a.CW) ? a.s[(gi * a.HW + p) * a.CW + wg] : 0u;
    if (wg < a.CW) {
#pragma unroll 4
      for (int bb = 0; bb < 16; ++bb) {
        const int b = half * 16 + bb;
        uint32_t d0 = 0, d1 = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          d0 |= ((w[p] >> b) & 1u) << (4 * p + 1);              // fp4 1.0 = 0b0010
          d1 |= ((w[8 + p] >> b) & 1u) << (4 * p + 1);
        }
        Bt[wg * 32 + b][img] = u2{d0, d1};
      }
    }
    for (int idx = tid; idx < a.C * 32; idx += 256) {
      const int im = idx / a.C, c = idx - im * a.C;
      Gt[c][im] = (img0 + im < a.NB) ? a.gate[(img0 + im) * a.C + c] : 0.0f;
    }
  }
  __syncthreads();

  // ---- the chain over the channels: four output tiles per wave ------------------------------
  const int ot0 = (blockIdx.y * 4 + wave) * 4;
  if (ot0 >= a.OT) return;
  v16f acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const v16f zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  typedef v4i avec;
  const v4i *apl = (const v4i *)a.ap + (int64_t)ot0 * 64 + lane;
  const int sa = WIDE ? (h ? 131 : 127) : 127;             // E8M0 scale of A's K block h: 2^0 / 2^4
  constexpr int GC = 4;
  avec acur[GC][4], anxt[GC][4];
  auto load_group = [&](avec (&dst)[GC][4], int c0) {
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const int c = min(c0 + j, a.C - 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[j][t] = apl[((int64_t)c * a.OT + min(t, a.OT - 1 - ot0)) * 64];
    }
  };
  load_group(acur, 0);
  for (int c0 = 0; c0 < a.C; c0 += GC) {
    load_group(anxt, c0 + GC < a.C ? c0 + GC : c0);
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const int c = c0 + j;
      const u2 bb = Bt[c][n];
      const float g = Gt[c][n];
      // fp6: the positions are k = 0..15 (lane half 0); wide: k = 0..15 and k = 32..47 -- both halves
      const v8i B = WIDE ? v8i{(int)bb.x, (int)bb.y, 0, 0, 0, 0, 0, 0}
                         : v8i{(int)(h ? 0u : bb.x), (int)(h ? 0u : bb.y), 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v16f I;
        const v8i A = {acur[j][t].x, acur[j][t].y, acur[j][t].z, 0, 0, 0, 0, 0};
        if constexpr (WIDE) {
          I = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, zero16, 3 /* A: fp6 e3m2 */, 4 /* B: fp4 */,
                                                              0, sa, 0, 127);
        } else {
          I = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, zero16, 2 /* A: fp6 */, 4 /* B: fp4 */,
                                                              0, 127, 0, 127);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = __builtin_fmaf(g, I[i], acc[t][i]);
      }
    }
#pragma unroll
    for (int j = 0; j < GC; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acur[j][t] = anxt[j][t];
  }

  // ---- dequantise and store: lane = image, register i = output (i & 3) + 8 (i >> 2) + 4 h ----
  const int64_t gi = img0 + n;
  if (gi < a.NB) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (ot0 + t < a.OT) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int o = (ot0 + t) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (o < a.N) {
            const float q = acc[t][i] / a.L;
            a.y[gi * a.N + o] = q * a.m;
          }
        }
      }
    }
  }
}

const char *dense_gated_unsupported(int32_t HW, int32_t C, int32_t N, const snnqp_weight_t *w) {
  if (w->wtype != SNNQP_W_I8) return "weights are not int8 codes";
  if (!(w->code_max > 0 && w->code_max <= 127)) return "unknown code range (code_max)";
  if (HW < 1 || HW > DG_PMAX) return "more than 16 positions per channel";
  if (C < 32 || C > DG_CMAX || C % 32) return "channels not 32, 64, 96 or 128";
  if (N < 1) return "no outputs";
  return nullptr;
}

}  // namespace snnqp

extern "C" int64_t snnqp_dense_gated_packed_bytes_ex(int32_t C, int32_t N, int32_t code_max) {
  if (C <= 0 || N <= 0 || code_max <= 0 || code_max > 127) return 0;
  return (int64_t)C * ((N + 31) / 32) * 64 * 4 * 4;        // (both layouts: four dwords per lane)
}
extern "C" int64_t snnqp_dense_gated_packed_bytes(int32_t C, int32_t N) {
  return snnqp_dense_gated_packed_bytes_ex(C, N, 7);
}

extern "C" int snnqp_pack_codes_dense_gated_ex(const int8_t *w, int32_t C, int32_t HW, int32_t N, int32_t code_max,
                                               void *packed, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w && packed && C > 0 && HW > 0 && HW <= DG_PMAX && N > 0 && code_max > 0 && code_max <= 127,
                SNNQP_EINVAL, "pack_codes_dense_gated: bad argument");
  const int OT = (N + 31) / 32;
  const int64_t total = (int64_t)C * OT * 64;
  const int64_t blocks = (total + 255) / 256;
  if (code_max <= 7)
    hipLaunchKernelGGL(pack_codes_dense_gated_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, C, HW, N, OT, (uint32_t *)packed);
  else
    hipLaunchKernelGGL(pack_codes_dense_gated_wide_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256),
                       0, (hipStream_t)stream, w, C, HW, N, OT, (uint32_t *)packed);
  SNNQP_CHECK_LAUNCH("pack_codes_dense_gated_kernel");
  return SNNQP_OK;
}
extern "C" int snnqp_pack_codes_dense_gated(const int8_t *w, int32_t C, int32_t HW, int32_t N, void *packed,
                                            snnqp_stream_t stream) {
  return snnqp_pack_codes_dense_gated_ex(w, C, HW, N, 7, packed, stream);
}

extern "C" int snnqp_dense_gated_forward(const uint32_t *s, const float *gate, int64_t NB, int32_t HW,
                                         int32_t C, int32_t N, const snnqp_weight_t *w,
                                         const void *packed, float *y, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w && packed && NB >= 0 && ((s && gate && y) || NB == 0), SNNQP_EINVAL,
                "dense_gated_forward: bad argument");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  const char *why = dense_gated_unsupported(HW, C, N, w);
  SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "dense_gated_forward: %s", why);
  if (NB == 0) return SNNQP_OK;
  DenseGatedArgs a;
  a.s = s; a.gate = gate; a.ap = (const uint32_t *)packed; a.y = y;
  a.NB = NB; a.HW = HW; a.C = C; a.CW = C / 32; a.N = N; a.OT = (N + 31) / 32;
  a.L = w->L; a.m = w->m;
  const int64_t gx = (NB + 31) / 32;
  SNNQP_REQUIRE(gx < ((int64_t)1 << 31), SNNQP_EUNSUPPORTED, "dense_gated_forward: more than 2^31 workgroups");
  if (w->code_max > 7)          // two fp8 digits per code (packed by snnqp_pack_codes_dense_gated_ex)
    hipLaunchKernelGGL(dense_gated_kernel<true>, dim3((unsigned)gx, (unsigned)((a.OT + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(dense_gated_kernel<false>, dim3((unsigned)gx, (unsigned)((a.OT + 15) / 16)), dim3(256), 0,
                       (hipStream_t)stream, a);
  SNNQP_CHECK_LAUNCH("dense_gated_kernel");
  return SNNQP_OK;
}
