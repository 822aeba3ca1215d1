// Direct-form SpikingBlock: one thread owns one output neuron for all T steps
// (membrane potential in a register), computing its convolution / dense sum
// from the input each step.  Serves every geometry and element type of
// QuantConv (flax_qconv.py:93-171: 1-D, 2-D and -- with a depth axis, snnqp_conv3d_* -- 3-D) and
// QuantDense (flax_qdense.py:58-89, the 1x1 convolution on a 1x1 image); the MFMA kernels take
// over the shapes the BASELINE configs use.  Two arithmetic paths:
//   INT   int8 codes x integer input (u8 counts or spike bits): exact int32
//         accumulator, current = fl(fl(acc / L) * m)
//   FSEQ  float32 weights x any input: fmaf chain over (kd, kh, kw, cin) ascending
#include "kernels.h"

namespace snnqp {

struct GenericArgs {
  const void *x;
  int64_t xs_t, xs_b;        // element (word for BITS) strides of t and b
  int32_t T, B;
  snnqp_conv_geom_t g;
  int32_t OH, OW, Hd, Wd, CinG, CoutG, CWin;
  const void *w;
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  void *s_out;               // spikes (neuron) or float32 currents (no neuron)
  int32_t s_type;
  int32_t *acc_out;          // optional int32 accumulators (no-neuron mode)
  int64_t total;             // threads: B * OD * WH * WW * Cout
  int64_t total_out;         // outputs: B * OD * OH * OW * Cout (OH, OW: pooled when the pool is fused)
  int32_t WH, WW;            // windows walked: OH x OW, or ceil(FH / 2) x ceil(FW / 2) when a fused pool
                             // over an odd image must still carry the potentials of the edge neurons
  int32_t FH, FW;            // full-resolution output size (u0 / u_out)
  const int32_t *pred;       // nullable device word: skip the launch unless *pred != 0
  // depth axis of a 3-D convolution (images [D][H][W][Cin], kernels [KD][KH][KW][Cin/g][Cout]);
  // D = KD = OD = 1 and unit stride / dilation, zero padding for everything else
  int32_t D, KD, OD, Dd, stride_d, pad_d_lo, in_dil_d, k_dil_d;
};

template <int IN>
__device__ __forceinline__ int load_int(const void *x, int64_t pix_off, int32_t c) {
  if (IN == SNNQP_U8) return (int)((const uint8_t *)x)[pix_off + c];
  // BITS: pix_off is in words
  return (int)((((const uint32_t *)x)[pix_off + (c >> 5)] >> (c & 31)) & 1u);
}

template <int IN>
__device__ __forceinline__ float load_float(const void *x, int64_t pix_off,
                                            int32_t c) {
  if (IN == SNNQP_F32) return ((const float *)x)[pix_off + c];
  return (float)load_int<IN == SNNQP_F32 ? SNNQP_U8 : IN>(x, pix_off, c);
}

// POOL = 2: a thread owns the 2x2 window of neurons behind one POOLED output (their four membrane
// potentials) and writes the OR of their spikes -- the max-pool of examples/tcja/models.py:145-147
// fused, as the MFMA kernels have it; a.total, a.OH / a.OW then count pooled outputs, a.FH / a.FW
// the full-resolution ones (u0 / u_out).  An odd FH / FW leaves a last row / column of neurons the
// pool drops (reduce_window without padding); when their potentials are wanted (u_out) the walk
// covers ceil(FH / 2) x ceil(FW / 2) windows, the edge ones partly outside, and only windows
// inside OH x OW write a spike.  pred: the whole launch is skipped unless *pred != 0 when
// the stream reaches it (the float32 re-evaluation behind a speculative integer launch).
template <int IN, bool INTPATH, int POOL>
__global__ void __launch_bounds__(256)
generic_block_kernel(GenericArgs a) {
  if (a.pred && *(const volatile int32_t *)a.pred == 0) return;
  constexpr int P = POOL * POOL;
  const snnqp_conv_geom_t &g = a.g;
  const int64_t pix_elems = (IN == SNNQP_BITS) ? a.CWin : g.Cin;
  const bool has_bn = a.bn.mean != nullptr;
  const bool word_aligned = (g.Cout & 31) == 0;
  const int32_t CWout = (g.Cout + 31) / 32;
  const int64_t span = (a.total + 255) / 256 * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < span; idx += (int64_t)gridDim.x * 256) {
    const bool live = idx < a.total;
    int64_t r = live ? idx : 0;
    const int32_t co = (int32_t)(r % g.Cout); r /= g.Cout;
    const int32_t px = (int32_t)(r % a.WW); r /= a.WW;
    const int32_t py = (int32_t)(r % a.WH); r /= a.WH;
    const int32_t pz = (int32_t)(r % a.OD); r /= a.OD;
    const int32_t b = (int32_t)r;
    const int32_t grp = co / a.CoutG;
    const int32_t cin0 = grp * a.CinG;

    float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f;
    if (has_bn) { bmean = a.bn.mean[co]; bmul = a.bn.mul[co]; bbias = a.bn.bias[co]; }
    if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[co];
    float u[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      u[p] = 0.0f;
      if (live && a.u0 && a.nrn.kind != SNNQP_NEURON_NONE && py * POOL + p / POOL < a.FH && px * POOL + p % POOL < a.FW)
        u[p] = a.u0[((((int64_t)b * a.OD + pz) * a.FH + (py * POOL + p / POOL)) * a.FW + (px * POOL + p % POOL)) * g.Cout + co];
    }

    for (int32_t t = 0; t < a.T; ++t) {
      const int64_t img_off = (int64_t)t * a.xs_t + (int64_t)b * a.xs_b;
      // (every window inside OH x OW has all its neurons inside FH x FW)
      const bool out_ok = live && py < a.OH && px < a.OW;
      const int64_t o_pix = (((int64_t)b * a.OD + pz) * a.OH + py) * a.OW + px;
      const int64_t o = (int64_t)t * a.total_out + o_pix * g.Cout + co;
      bool s = false;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int32_t oy = py * POOL + p / POOL, ox = px * POOL + p % POOL;
        int iacc = 0;
        float facc = 0.0f;
        const bool inb = live && oy < a.FH && ox < a.FW;
        if (inb) {
          for (int32_t kd = 0; kd < a.KD; ++kd) {
            const int32_t zd = pz * a.stride_d - a.pad_d_lo + kd * a.k_dil_d;
            if (zd < 0 || zd >= a.Dd || (zd % a.in_dil_d) != 0) continue;
            const int32_t iz = zd / a.in_dil_d;
            for (int32_t kh = 0; kh < g.KH; ++kh) {
              const int32_t yd = oy * g.stride_h - g.pad_h_lo + kh * g.k_dil_h;
              if (yd < 0 || yd >= a.Hd || (yd % g.in_dil_h) != 0) continue;
              const int32_t iy = yd / g.in_dil_h;
              for (int32_t kw = 0; kw < g.KW; ++kw) {
                const int32_t xd = ox * g.stride_w - g.pad_w_lo + kw * g.k_dil_w;
                if (xd < 0 || xd >= a.Wd || (xd % g.in_dil_w) != 0) continue;
                const int32_t ix = xd / g.in_dil_w;
                const int64_t pix_off = img_off + (((int64_t)iz * g.H + iy) * g.W + ix) * pix_elems;
                const int64_t wbase = ((int64_t)((kd * g.KH + kh) * g.KW + kw) * a.CinG) * g.Cout + co;
                if (INTPATH) {
                  const int8_t *w = (const int8_t *)a.w;
                  for (int32_t ci = 0; ci < a.CinG; ++ci)
                    iacc += load_int<IN>(a.x, pix_off, cin0 + ci) *
                            (int)w[wbase + (int64_t)ci * g.Cout];
                } else {
                  const float *w = (const float *)a.w;
                  for (int32_t ci = 0; ci < a.CinG; ++ci)
                    facc = __builtin_fmaf(load_float<IN>(a.x, pix_off, cin0 + ci),
                                          w[wbase + (int64_t)ci * g.Cout], facc);
                }
              }
            }
          }
        }
        float cur = INTPATH ? dequant_acc(iacc, a.dq) : facc;
        if (has_bn) cur = bn_apply(cur, bmean, bmul, bbias);
        if (a.nrn.kind == SNNQP_NEURON_NONE) {       // (POOL == 1: run_generic)
          if (live) {
            ((float *)a.s_out)[o] = cur;
            if (INTPATH && a.acc_out) a.acc_out[o] = iacc;
          }
          continue;
        }
        if (inb) s |= neuron_step(u[p], cur, a.nrn, dec);
      }
      if (a.nrn.kind == SNNQP_NEURON_NONE) continue;
      if (a.s_type == SNNQP_F32) {
        if (out_ok) ((float *)a.s_out)[o] = s ? 1.0f : 0.0f;
      } else if (word_aligned) {
        // Cout % 32 == 0: the packed layout is the linear bit index o; 32 consecutive lanes are the
        // 32 channels of one word of one pixel (idx, hence co, is a multiple of 32 at lane 0 / 32)
        const unsigned long long m = __ballot(s);
        const int lane = threadIdx.x & 63;
        if (out_ok && (lane & 31) == 0)
          ((uint32_t *)a.s_out)[o >> 5] = (uint32_t)(lane ? (m >> 32) : m);
      } else if (out_ok && s) {
        const int64_t rows = a.total_out / g.Cout;
        atomicOr(&((uint32_t *)a.s_out)[((int64_t)t * rows + o_pix) * CWout + (co >> 5)],
                 1u << (co & 31));
      }
    }
    if (live && a.u_out && a.nrn.kind != SNNQP_NEURON_NONE) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (py * POOL + p / POOL < a.FH && px * POOL + p % POOL < a.FW)
          a.u_out[((((int64_t)b * a.OD + pz) * a.FH + (py * POOL + p / POOL)) * a.FW + (px * POOL + p % POOL)) * g.Cout + co] = u[p];
    }
  }
}

// words := 0 unless *pred == 0 (the packed raster of a predicated launch whose Cout is not a
// multiple of 32 is assembled with atomicOr)
__global__ void __launch_bounds__(256) zero_words_if_kernel(const int32_t *pred, uint32_t *p, int64_t n) {
  if (pred && *(const volatile int32_t *)pred == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}

template <int IN, bool INTPATH>
static int launch_generic(const GenericArgs &a, int pool, hipStream_t st) {
  int64_t blocks = ceil_div64(a.total, 256);
  // a grid-stride walk: a predicated launch that is not taken costs one small grid, and no
  // launch needs more than 2^31 workgroups
  if (blocks > 16384) blocks = 16384;
  if (pool == 2)
    hipLaunchKernelGGL((generic_block_kernel<IN, INTPATH, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((generic_block_kernel<IN, INTPATH, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  SNNQP_CHECK_LAUNCH("generic_block_kernel");
  return SNNQP_OK;
}

int check_geom(const snnqp_conv_geom_t *g, int32_t *OH, int32_t *OW) {
  SNNQP_REQUIRE(g, SNNQP_EINVAL, "null geometry");
  SNNQP_REQUIRE(g->H >= 0 && g->W >= 0 && g->Cin > 0 && g->Cout > 0, SNNQP_EINVAL,
                "bad geometry H=%d W=%d Cin=%d Cout=%d", g->H, g->W, g->Cin, g->Cout);
  SNNQP_REQUIRE(g->groups > 0 && g->Cin % g->groups == 0 && g->Cout % g->groups == 0,
                SNNQP_EINVAL,  // flax_qconv.py:117 assert
                "in_features %d / features %d not divisible by feature_group_count %d",
                g->Cin, g->Cout, g->groups);
  int rc = snnqp_conv_out_shape(g, OH, OW);
  return rc;
}

// Shared by snnqp_conv_forward / snnqp_conv_lif_forward / snnqp_dense_lif_forward.
int run_generic(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T,
                int32_t B, const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                float *u_out, void *s_out, int s_type, int32_t *acc_out,
                hipStream_t st, int pool, const int32_t *pred, const GenericDepth *dz) {
  int32_t OH, OW;
  int rc = check_geom(g, &OH, &OW);
  if (rc) return rc;
  SNNQP_REQUIRE(!dz || pool == 1, SNNQP_EINVAL, "generic block: the fused pool is 2-D");
  SNNQP_REQUIRE(pool == 1 || (pool == 2 && nrn && nrn->kind != SNNQP_NEURON_NONE), SNNQP_EINVAL,
                "generic block: pool must be 1, or 2 with a neuron");
  SNNQP_REQUIRE(w && w->w && ((x && s_out) || T == 0 || B == 0), SNNQP_EINVAL, "generic block: null pointer");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "generic block: negative T/B");
  GenericArgs a;
  a.x = x; a.xs_t = xs_t; a.xs_b = xs_b; a.T = T; a.B = B; a.g = *g;
  a.FH = OH; a.FW = OW;
  a.OH = OH / pool; a.OW = OW / pool;
  a.pred = pred;
  a.D = dz ? dz->D : 1; a.KD = dz ? dz->KD : 1; a.OD = dz ? dz->OD : 1;
  a.stride_d = dz ? dz->stride : 1; a.pad_d_lo = dz ? dz->pad_lo : 0;
  a.in_dil_d = dz ? dz->in_dil : 1; a.k_dil_d = dz ? dz->k_dil : 1;
  a.Dd = a.D > 0 ? (a.D - 1) * a.in_dil_d + 1 : 0;
  a.Hd = g->H > 0 ? (g->H - 1) * g->in_dil_h + 1 : 0;
  a.Wd = g->W > 0 ? (g->W - 1) * g->in_dil_w + 1 : 0;
  a.CinG = g->Cin / g->groups; a.CoutG = g->Cout / g->groups;
  a.CWin = (g->Cin + 31) / 32;
  a.w = w->w;
  a.dq = make_dequant(w->wtype == SNNQP_W_I8 ? w->L : 1.0f,
                      w->wtype == SNNQP_W_I8 ? w->m : 1.0f);
  a.bn = make_bn(bn); a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out; a.s_type = s_type;
  a.acc_out = acc_out;
  a.WH = a.OH; a.WW = a.OW;
  if (pool == 2 && u_out) { a.WH = (OH + 1) / 2; a.WW = (OW + 1) / 2; }
  a.total_out = (int64_t)B * a.OD * a.OH * a.OW * g->Cout;
  a.total = (int64_t)B * a.OD * a.WH * a.WW * g->Cout;
  if (a.total == 0 || T == 0) return SNNQP_OK;
  if (a.nrn.kind == SNNQP_NEURON_LIF)
    SNNQP_REQUIRE(a.nrn.decay, SNNQP_EINVAL, "LIF neuron needs a decay vector");
  SNNQP_CHECK_BN(bn);
  if (a.nrn.kind != SNNQP_NEURON_NONE) {
    SNNQP_REQUIRE(s_type == SNNQP_F32 || s_type == SNNQP_BITS, SNNQP_EINVAL,
                  "spike output type must be F32 or BITS");
    if (s_type == SNNQP_BITS && (g->Cout & 31) != 0) {
      const int64_t words = (int64_t)T * B * a.OD * a.OH * a.OW * ((g->Cout + 31) / 32);
      const int64_t zb = (words + 255) / 256;
      hipLaunchKernelGGL(zero_words_if_kernel, dim3((unsigned)(zb < 4096 ? zb : 4096)), dim3(256), 0, st,
                         pred, (uint32_t *)s_out, words);
      SNNQP_CHECK_LAUNCH("zero_words_if_kernel");
    }
  }
  const bool intpath = (w->wtype == SNNQP_W_I8);
  if (intpath) {
    SNNQP_REQUIRE(in_type == SNNQP_U8 || in_type == SNNQP_BITS, SNNQP_EUNSUPPORTED,
                  "int8 codes need integer-typed input (U8/BITS); pass the "
                  "fake-quantised float kernel for float32 input");
    SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
    // int32 accumulator cannot overflow: |acc| <= 255 * 127 * K
    const int64_t kk = (int64_t)a.KD * g->KH * g->KW * a.CinG;
    SNNQP_REQUIRE(kk * 255 * 127 < (1ll << 31), SNNQP_EUNSUPPORTED,
                  "contraction length %lld overflows int32", (long long)kk);
    return in_type == SNNQP_U8 ? launch_generic<SNNQP_U8, true>(a, pool, st)
                               : launch_generic<SNNQP_BITS, true>(a, pool, st);
  }
  SNNQP_REQUIRE(w->wtype == SNNQP_W_F32, SNNQP_EINVAL, "unknown weight type");
  switch (in_type) {
    case SNNQP_F32: return launch_generic<SNNQP_F32, false>(a, pool, st);
    case SNNQP_U8: return launch_generic<SNNQP_U8, false>(a, pool, st);
    case SNNQP_BITS: return launch_generic<SNNQP_BITS, false>(a, pool, st);
  }
  set_error("unknown input type %d", in_type);
  return SNNQP_EINVAL;
}

}  // namespace snnqp

extern "C" int snnqp_conv_forward(const void *x, int in_type, int64_t NB,
                                  const snnqp_conv_geom_t *g,
                                  const snnqp_weight_t *w, float *y,
                                  int32_t *acc, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(g, SNNQP_EINVAL, "conv_forward: null geometry");
  SNNQP_REQUIRE(NB >= 0 && NB < (1ll << 31), SNNQP_EINVAL, "conv_forward: bad NB");
  const int64_t pix = (in_type == SNNQP_BITS) ? (g->Cin + 31) / 32 : g->Cin;
  const int64_t img = (int64_t)g->H * g->W * pix;
  // float32 inputs and kernels of the common shapes: the same fmaf chain on the f32 MFMA
  if (x && w && w->w && y && !acc && !fseq_gemm_unsupported(in_type, g, w)) {
    int32_t OH, OW;
    const int rc = check_geom(g, &OH, &OW);
    if (rc) return rc;
    return run_fseq_gemm(x, in_type, NB, g, w, y, (hipStream_t)stream);
  }
  // the NB images are the "batch"; T = 1
  return run_generic(x, in_type, 0, img, 1, (int32_t)NB, g, w, nullptr, nullptr,
                     nullptr, nullptr, y, SNNQP_F32, acc, (hipStream_t)stream);
}

// The predicated form of snnqp_conv_forward (snnqp.h): the float32 re-evaluation of a connection
// whose integer launch met a value that is not an integer in [0, 255].  Direct form (the same fmaf
// chain as the f32-MFMA kernel, which is not predicated): the rare path.
extern "C" int snnqp_conv_forward_if(const int32_t *pred, const void *x, int in_type, int64_t NB,
                                     const snnqp_conv_geom_t *g, const snnqp_weight_t *w, float *y,
                                     snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(pred && g, SNNQP_EINVAL, "conv_forward_if: null argument");
  SNNQP_REQUIRE(NB >= 0 && NB < (1ll << 31), SNNQP_EINVAL, "conv_forward_if: bad NB");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "conv_forward_if")) return rc;
  const int64_t pix = (in_type == SNNQP_BITS) ? (g->Cin + 31) / 32 : g->Cin;
  const int64_t img = (int64_t)g->H * g->W * pix;
  return run_generic(x, in_type, 0, img, 1, (int32_t)NB, g, w, nullptr, nullptr, nullptr, nullptr, y,
                     SNNQP_F32, nullptr, (hipStream_t)stream, 1, pred);
}

// ---- 3-D convolutions (flax_qconv.py:93-171 with three spatial axes) --------------------------
namespace snnqp {
static int split_geom3(const snnqp_conv3d_geom_t *g3, snnqp_conv_geom_t *g, GenericDepth *dz) {
  SNNQP_REQUIRE(g3, SNNQP_EINVAL, "conv3d: null geometry");
  SNNQP_REQUIRE(g3->D >= 0 && g3->KD > 0 && g3->stride[0] > 0 && g3->in_dil[0] > 0 && g3->k_dil[0] > 0 &&
                    g3->pad_lo[0] >= 0 && g3->pad_hi[0] >= 0,
                SNNQP_EINVAL, "conv3d: bad depth geometry");
  g->H = g3->H; g->W = g3->W; g->Cin = g3->Cin; g->Cout = g3->Cout;
  g->KH = g3->KH; g->KW = g3->KW;
  g->stride_h = g3->stride[1]; g->stride_w = g3->stride[2];
  g->pad_h_lo = g3->pad_lo[1]; g->pad_h_hi = g3->pad_hi[1];
  g->pad_w_lo = g3->pad_lo[2]; g->pad_w_hi = g3->pad_hi[2];
  g->in_dil_h = g3->in_dil[1]; g->in_dil_w = g3->in_dil[2];
  g->k_dil_h = g3->k_dil[1]; g->k_dil_w = g3->k_dil[2];
  g->groups = g3->groups;
  dz->D = g3->D; dz->KD = g3->KD; dz->stride = g3->stride[0]; dz->pad_lo = g3->pad_lo[0];
  dz->in_dil = g3->in_dil[0]; dz->k_dil = g3->k_dil[0];
  const int64_t dd = g3->D > 0 ? (int64_t)(g3->D - 1) * g3->in_dil[0] + 1 : 0;
  const int64_t kd = (int64_t)(g3->KD - 1) * g3->k_dil[0] + 1;
  const int64_t td = dd + g3->pad_lo[0] + g3->pad_hi[0];
  dz->OD = td < kd ? 0 : (int32_t)((td - kd) / g3->stride[0] + 1);
  return SNNQP_OK;
}
}  // namespace snnqp

extern "C" int snnqp_conv3d_out_shape(const snnqp_conv3d_geom_t *g3, int32_t *OD, int32_t *OH, int32_t *OW) {
  using namespace snnqp;
  SNNQP_REQUIRE(OD && OH && OW, SNNQP_EINVAL, "conv3d_out_shape: null argument");
  snnqp_conv_geom_t g;
  GenericDepth dz;
  if (int rc = split_geom3(g3, &g, &dz)) return rc;
  *OD = dz.OD;
  return snnqp_conv_out_shape(&g, OH, OW);
}

extern "C" int snnqp_conv3d_lif_forward(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                                        int64_t x_stride_b, int32_t T, int32_t B,
                                        const snnqp_conv3d_geom_t *g3, const snnqp_weight_t *w,
                                        const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                                        float *u_out, void *s_out, int s_type, snnqp_stream_t stream) {
  using namespace snnqp;
  snnqp_conv_geom_t g;
  GenericDepth dz;
  if (int rc = split_geom3(g3, &g, &dz)) return rc;
  if (int rc = refuse_after_device_report((hipStream_t)stream, "conv3d_lif_forward")) return rc;
  SNNQP_REQUIRE(!nrn || (nrn->kind >= SNNQP_NEURON_NONE && nrn->kind <= SNNQP_NEURON_LIF), SNNQP_EINVAL,
                "conv3d_lif_forward: unknown neuron kind %d", nrn ? nrn->kind : -1);
  return run_generic(x, in_type, x_stride_t, x_stride_b, T, B, &g, w, bn, nrn, u0, u_out, s_out, s_type,
                     nullptr, (hipStream_t)stream, 1, pred, &dz);
}
