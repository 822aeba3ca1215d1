// QuantConv 3x3 on (gate x spike raster): the conv block behind a TCJA gate in the reference's
// full model (examples/tcja/models.py:95-97 -> :149-187; x = s * sigmoid(...)[:, :, None, None, :]
// into QuantConv, flax_qconv.py:158-168), in the 'gint' contract:
//     I[p, c, o] = sum over the nine taps of code[tap, c, o] * s[tap's pixel, c]     (exact integer)
//     acc[p, o]  = fmaf(gate[c], I[p, c, o], acc[p, o])   for c = 0 .. C - 1          (start +0)
//     current    = fl(fl(acc / L) * m)
// The gate of a channel multiplies all nine taps of that channel, so it is factored out of them:
// the taps are summed as integers on the matrix pipe -- one v_mfma_scale_f32_32x32x64_f8f6f4 per
// (32 pixels, channel, 32 outputs) with K = the nine taps (fp4 spikes x fp6 codes, exact in the
// float32 accumulator) -- and the C gates are applied by one fmaf chain per output on the vector
// unit.  The float32 connection this replaces (fseq_gemm.hip: K = 9 C fmaf on the f32 MFMA) took
// 3.1 ms for the 8 x 8 x 128 -> 128 layer of CextNet at B = 1024, T = 20; this one is bound by its
// 64 fmaf per MFMA.
//
// A wave owns a 4 x 8-pixel patch of one image and all (up to 128) outputs of a blockIdx.y:
//  1. the nine tap words of its pixels are transposed once into an LDS table of A-operand dwords,
//     At[c][pixel] = the eight fp4 nibbles of taps 0..7 of channel c (16 KiB per wave); tap 8 stays
//     in registers as the spike words of that tap;
//  2. loop over c: A = {At[c][pixel], tap 8}, B = the pre-packed fp6 codes of (c, 32 outputs),
//     MFMA from C = 0, then acc = fmaf(gate[c], result, acc) -- gate[c] is a scalar register;
//  3. dequantise, store float32 currents [NB][H][W][Cout] (BatchNorm and the neuron follow as
//     the stand-alone scan, as behind the float32 connection).
#include <type_traits>

#include "kernels.h"

namespace snnqp {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int CG_WAVES = 4;
constexpr int CG_CMAX = 128;          // input channels (the LDS table is CMAX x 32 dwords per wave)

struct ConvGatedArgs {
  const uint32_t *s;      // [NB][H][W][CW] spike words
  const float *gate;      // [NB][C]
  const uint32_t *bp;     // packed codes [C][OT][64 lanes][2 dwords] (pack_codes_gated_kernel)
  float *y;               // [NB][H][W][Cout]
  int64_t NB, npatch;
  int32_t H, W, C, CW, Cout, OT, tiles_y, tiles_x;
  int32_t ot_base;        // first output tile of this launch
  float L, m;
};

// integer -7..7 -> e2m3 (runtime.hip enc6)
__device__ __forceinline__ uint32_t enc6(int v) {
  const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
  const uint32_t code = mag == 0 ? 0u : mag == 1 ? 8u : mag == 2 ? 16u : mag == 3 ? 20u : mag == 4 ? 24u
                        : mag == 5 ? 26u : mag == 6 ? 28u : 30u;
  return code | (v < 0 ? 32u : 0u);
}

}  // namespace

// codes int8 HWIO [9][C][Cout] -> bp[c][ot][lane][2]: lane (n, h = 0) holds the fp6 values of taps
// 0..7 at bits [6 j, 6 j + 6) and, lane (n, h = 1), tap 8 at bits [0, 6) -- k = 32 h + j of the
// 64-deep matrix instruction; columns beyond Cout are zero.
__global__ void __launch_bounds__(256)
pack_codes_gated_kernel(const int8_t *w, int32_t C, int32_t Cout, int32_t OT, uint32_t *bp) {
  const int64_t total = (int64_t)C * OT * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), n = lane & 31, h = lane >> 5;
    const int ot = (int)((i >> 6) % OT), c = (int)((i >> 6) / OT);
    const int o = ot * 32 + n;
    unsigned long long bits = 0;
    if (o < Cout) {
      if (h == 0) {
        for (int tap = 0; tap < 8; ++tap)
          bits |= (unsigned long long)enc6(w[((int64_t)tap * C + c) * Cout + o]) << (6 * tap);
      } else {
        bits = enc6(w[((int64_t)8 * C + c) * Cout + o]);
      }
    }
    bp[i * 2] = (uint32_t)bits;
    bp[i * 2 + 1] = (uint32_t)(bits >> 32);
  }
}

// integer -8..8 -> fp6 e3m2 (bias 3, three significant bits: every integer up to 8 is exact;
// 1 = 0x0C, 2 = 0x10, 3 = 0x12, 4 = 0x14, 5 = 0x15, 6 = 0x16, 7 = 0x17, 8 = 0x18)
__device__ __forceinline__ uint32_t enc6w(int v) {
  const uint32_t mag = (uint32_t)(v < 0 ? -v : v);
  const uint32_t tab[9] = {0x00u, 0x0Cu, 0x10u, 0x12u, 0x14u, 0x15u, 0x16u, 0x17u, 0x18u};
  return tab[mag] | (v < 0 ? 0x20u : 0u);
}
// Codes beyond e2m3 (|code| <= 127: up to 8 bits) as TWO six-bit digits, code = 16 hi + lo with lo in
// [-8, 7], hi in [-8, 8] -- exact in the OTHER fp6 format, e3m2 -- one digit per 32-deep K block of
// the 64-deep instruction: the block scale of the second block is 2^4 (v_mfma_scale: an E8M0 scale
// per lane half = per K block), so ONE instruction returns sum(lo s) + 16 sum(hi s): the exact
// integer, at the fp6 rate and with not one vector instruction more than the narrow form.  (fp8
// digits work too -- tools/ubench/mfma_fp8_kmap.hip, mfma_fp8_digits.hip: an 8-bit operand beside a
// 4-bit one counts k differently, the digits then sit in bytes 0..8 / 16..24 of the lanes of half 0
// -- and were the first version: twice the matrix-pipe time and eight operand dwords, 2.09 ms for
// CextNet's 8-bit layer against 1.38 with e3m2.)  bp[c][ot][lane][2]: lane (n, h = 0) holds the lo digits
// of the nine taps at bits [6 j, 6 j + 6), lane (n, h = 1) the hi digits; the spike operand holds
// the nine taps in both lane halves.
__global__ void __launch_bounds__(256)
pack_codes_gated_wide_kernel(const int8_t *w, int32_t C, int32_t Cout, int32_t OT, uint32_t *bp) {
  const int64_t total = (int64_t)C * OT * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), n = lane & 31, h = lane >> 5;
    const int ot = (int)((i >> 6) % OT), c = (int)((i >> 6) / OT);
    const int o = ot * 32 + n;
    unsigned long long bits = 0;
    if (o < Cout) {
      for (int tap = 0; tap < 9; ++tap) {
        const int code = w[((int64_t)tap * C + c) * Cout + o];
        const int lo = ((code + 8) & 15) - 8, hi = (code - lo) / 16;
        bits |= (unsigned long long)enc6w(h ? hi : lo) << (6 * tap);
      }
    }
    bp[i * 2] = (uint32_t)bits;
    bp[i * 2 + 1] = (uint32_t)(bits >> 32);
  }
}

// NT: output tiles (of 32) this launch's workgroups hold -- a template parameter, so that the four
// matrix instructions of a channel and their 64 fmaf are one basic block the scheduler can
// overlap (with a run-time count every instruction sat in a block of its own, waited for alone:
// 1.48 ms for CextNet's layer, against ... with this)
template <int NT, bool WIDE = false>
__global__ void __launch_bounds__(CG_WAVES * 64, 2)
conv_gated_kernel(ConvGatedArgs a) {
  __shared__ uint32_t At[CG_WAVES][CG_CMAX][32];
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t patch = (int64_t)blockIdx.x * CG_WAVES + wave;
  if (patch >= a.npatch) return;                      // (no barrier below: waves are independent)
  const int ppi = a.tiles_y * a.tiles_x;
  const int64_t img = patch / ppi;
  const int pi = (int)(patch - img * ppi);
  const int y0 = (pi / a.tiles_x) * 4, x0 = (pi % a.tiles_x) * 8;
  const int py = n >> 3, px = n & 7;                  // this lane's pixel of the 4 x 8 patch
  const uint32_t *simg = a.s + img * a.H * a.W * a.CW;

  // ---- 1. the tap words of pixel (py, px): lanes of half h transpose the channels
  //         [h C / 2, (h + 1) C / 2); every lane keeps the words of tap 8 -------------------------
  auto tap_word = [&](int tap, int wg) -> uint32_t {
    const int gy = y0 + py + tap / 3 - 1, gx = x0 + px + tap % 3 - 1;
    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && wg < a.CW;
    return in ? simg[((int64_t)gy * a.W + gx) * a.CW + wg] : 0u;
  };
  uint32_t w8[CG_CMAX / 32];
#pragma unroll
  for (int wg = 0; wg < CG_CMAX / 32; ++wg) w8[wg] = tap_word(8, wg);
  const int wgs_half = (a.CW + 1) / 2;                // word groups per lane half
  for (int q = 0; q < wgs_half; ++q) {
    const int wg = h * wgs_half + q;
    uint32_t tw[8];
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) tw[tap] = tap_word(tap, wg);
    if (wg < a.CW) {
#pragma unroll 4
      for (int b = 0; b < 32; ++b) {
        uint32_t d0 = 0;
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) d0 |= ((tw[tap] >> b) & 1u) << (4 * tap + 1);   // fp4 1.0 = 0b0010
        At[wave][wg * 32 + b][n] = d0;
      }
    }
  }
  // (the table is read by the lanes of this wave only: LDS operations of a wave complete in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- 2. the chain over the channels -------------------------------------------------------
  const int ot0 = a.ot_base + blockIdx.y * 4;
  constexpr int nt = NT;
  v16f acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const v16f zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float *grow = a.gate + img * a.C;             // wave-uniform: scalar loads
  // four channels per group; the codes of the next group are requested before this group's
  // matrix instructions (an L2 round trip per group would otherwise sit in front of every one)
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  typedef u2 bvec;
  constexpr int GC = 4;
  const uint32_t *bpl = a.bp + ((int64_t)ot0 * 64 + lane) * 2;
  bvec bcur[GC][4], bnxt[GC][4];
  auto load_group = [&](bvec (&dst)[GC][4], int c0) {
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const int c = min(c0 + j, a.C - 1);
      const uint32_t *bc = bpl + (int64_t)c * a.OT * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[j][t] = *(const u2 *)(bc + (t < nt ? t : 0) * 128);
    }
  };
  // WIDE: the E8M0 scale of B's K block h comes from the lanes of half h: 2^0 (lo digits), 2^4 (hi)
  const int sb = WIDE ? (h ? 131 : 127) : 127;
  load_group(bcur, 0);
  // the gates of a group are requested a group ahead as well (scalar loads: their latency would
  // otherwise sit in front of the group's first fmaf)
  float g[GC], gn[GC];
#pragma unroll
  for (int j = 0; j < GC; ++j) g[j] = grow[j];
  for (int c0 = 0; c0 < a.C; c0 += GC) {
    const int cn = c0 + GC < a.C ? c0 + GC : c0;
    load_group(bnxt, cn);
    // (the requests must leave HERE: without the clobber the compiler sinks them over the back edge
    // to the top of the group that consumes them -- every group then opened with an exposed L2
    // round trip: 1.17 -> 1.02 ms for CextNet's layer, round 6)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < GC; ++j) gn[j] = grow[cn + j];
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const int c = c0 + j;
      const uint32_t d0 = At[wave][c][n];
      // lanes of half 1 hold k = 32 ..: tap 8 in nibble 0 (fp4 1.0 = 0b0010)
      const uint32_t t8 = ((w8[(c >> 5) & (CG_CMAX / 32 - 1)] >> (c & 31)) & 1u) << 1;
      // fp6: tap 8 is k = 32 (lane half 1); WIDE: both halves hold the nine taps (k = 0..8 of
      // their block), against the lo and the hi digits
      const v8i A = WIDE ? v8i{(int)d0, (int)t8, 0, 0, 0, 0, 0, 0} : v8i{(int)(h ? t8 : d0), 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < nt) {
          v8i B;
          v16f I;
          // K elements 9.. of A are zero and no fp6 encoding is an infinity or a NaN: what the other
          // four dwords of the B tuple hold does not matter, so they are left undefined (sixteen
          // tuples per group otherwise cost 64 v_mov: 358 -> 310 vector instructions per 16 MFMAs)
          const v2i bxy = {(int)bcur[j][t].x, (int)bcur[j][t].y};
          if constexpr (WIDE) {
            B = __builtin_shufflevector(bxy, bxy, 0, 1, -1, -1, -1, -1, -1, -1);
            I = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, zero16, 4 /* fp4 */, 3 /* fp6 e3m2 */,
                                                                0, 127, 0, sb);
          } else {
            B = __builtin_shufflevector(bxy, bxy, 0, 1, -1, -1, -1, -1, -1, -1);
            I = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, zero16, 4 /* fp4 */, 2 /* fp6 */,
                                                                0, 127, 0, 127);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] = __builtin_fmaf(g[j], I[i], acc[t][i]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      g[j] = gn[j];
#pragma unroll
      for (int t = 0; t < 4; ++t) bcur[j][t] = bnxt[j][t];
    }
  }

  // ---- 3. dequantise and store: lane = output, register i = pixel (i & 3) + 8 (i >> 2) + 4 h ----
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int o = (ot0 + t) * 32 + n;
    if (t < nt && o < a.Cout) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int gy = y0 + (row >> 3), gx = x0 + (row & 7);
        if (gy < a.H && gx < a.W) {
          const float q = acc[t][i] / a.L;
          a.y[((img * a.H + gy) * a.W + gx) * a.Cout + o] = q * a.m;
        }
      }
    }
  }
}

const char *conv_gated_unsupported(const snnqp_conv_geom_t *g, const snnqp_weight_t *w) {
  if (w->wtype != SNNQP_W_I8) return "weights are not int8 codes";
  if (!(w->code_max > 0 && w->code_max <= 127)) return "unknown code range (code_max)";
  if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1) return "not 3x3 / stride 1";
  if (g->pad_h_lo != 1 || g->pad_h_hi != 1 || g->pad_w_lo != 1 || g->pad_w_hi != 1) return "padding is not 1";
  if (g->in_dil_h != 1 || g->in_dil_w != 1 || g->k_dil_h != 1 || g->k_dil_w != 1 || g->groups != 1)
    return "dilated or grouped convolution";
  if (g->Cin < 32 || g->Cin > CG_CMAX || g->Cin % 32) return "input channels not 32, 64, 96 or 128";
  if (g->H <= 0 || g->W <= 0 || g->Cout <= 0) return "empty geometry";
  return nullptr;
}

}  // namespace snnqp

extern "C" int64_t snnqp_conv_gated_packed_bytes_ex(int32_t Cin, int32_t Cout, int32_t code_max) {
  if (Cin <= 0 || Cout <= 0 || code_max <= 0 || code_max > 127) return 0;
  return (int64_t)Cin * ((Cout + 31) / 32) * 64 * 2 * 4;       // (both layouts: two dwords per lane)
}
extern "C" int64_t snnqp_conv_gated_packed_bytes(int32_t Cin, int32_t Cout) {
  return snnqp_conv_gated_packed_bytes_ex(Cin, Cout, 7);
}

extern "C" int snnqp_pack_codes_gated_ex(const int8_t *w, int32_t Cin, int32_t Cout, int32_t code_max,
                                         void *packed, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w && packed && Cin > 0 && Cout > 0 && code_max > 0 && code_max <= 127, SNNQP_EINVAL,
                "pack_codes_gated: bad argument");
  const int OT = (Cout + 31) / 32;
  const int64_t total = (int64_t)Cin * OT * 64;
  const int64_t blocks = (total + 255) / 256;
  if (code_max <= 7)
    hipLaunchKernelGGL(pack_codes_gated_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, Cin, Cout, OT, (uint32_t *)packed);
  else
    hipLaunchKernelGGL(pack_codes_gated_wide_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       (hipStream_t)stream, w, Cin, Cout, OT, (uint32_t *)packed);
  SNNQP_CHECK_LAUNCH("pack_codes_gated_kernel");
  return SNNQP_OK;
}
extern "C" int snnqp_pack_codes_gated(const int8_t *w, int32_t Cin, int32_t Cout, void *packed,
                                      snnqp_stream_t stream) {
  return snnqp_pack_codes_gated_ex(w, Cin, Cout, 7, packed, stream);
}

extern "C" int snnqp_conv_gated_forward(const uint32_t *s, const float *gate, int64_t NB,
                                        const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                                        const void *packed, float *y, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(g && w && packed && NB >= 0 && ((s && gate && y) || NB == 0), SNNQP_EINVAL,
                "conv_gated_forward: bad argument");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  const char *why = conv_gated_unsupported(g, w);
  SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "conv_gated_forward: %s", why);
  if (NB == 0) return SNNQP_OK;
  ConvGatedArgs a;
  a.s = s; a.gate = gate; a.bp = (const uint32_t *)packed; a.y = y;
  a.NB = NB; a.H = g->H; a.W = g->W; a.C = g->Cin; a.CW = (g->Cin + 31) / 32; a.Cout = g->Cout;
  a.OT = (g->Cout + 31) / 32;
  a.tiles_y = (g->H + 3) / 4; a.tiles_x = (g->W + 7) / 8;
  a.npatch = NB * a.tiles_y * a.tiles_x;
  a.L = w->L; a.m = w->m;
  a.ot_base = 0;
  const int64_t gx = (a.npatch + CG_WAVES - 1) / CG_WAVES;
  SNNQP_REQUIRE(gx < ((int64_t)1 << 31), SNNQP_EUNSUPPORTED, "conv_gated_forward: more than 2^31 workgroups");
  // full groups of four output tiles, then the remainder (its blockIdx.y = 0 is tile group OT / 4)
  const int full = a.OT / 4, rest = a.OT % 4;
  const bool wide = w->code_max > 7;          // two fp8 digits per code (packed by snnqp_pack_codes_gated_ex)
#define SNNQP_CG_LAUNCH(NTV, ARGS, GY)                                                                  \
  do {                                                                                                  \
    if (wide) hipLaunchKernelGGL((conv_gated_kernel<NTV, true>), dim3((unsigned)gx, (unsigned)(GY)),     \
                                 dim3(CG_WAVES * 64), 0, (hipStream_t)stream, ARGS);                    \
    else hipLaunchKernelGGL((conv_gated_kernel<NTV, false>), dim3((unsigned)gx, (unsigned)(GY)),         \
                            dim3(CG_WAVES * 64), 0, (hipStream_t)stream, ARGS);                         \
  } while (0)
  if (full > 0) SNNQP_CG_LAUNCH(4, a, full);
  if (rest > 0) {
    ConvGatedArgs b = a;
    b.ot_base = full * 4;
    if (rest == 1) SNNQP_CG_LAUNCH(1, b, 1);
    else if (rest == 2) SNNQP_CG_LAUNCH(2, b, 1);
    else SNNQP_CG_LAUNCH(3, b, 1);
  }
#undef SNNQP_CG_LAUNCH
  SNNQP_CHECK_LAUNCH("conv_gated_kernel");
  return SNNQP_OK;
}
