// Entry points of the fused SpikingBlock (spiking_learning.py:446-462) and the
// dispatch between the direct-form kernel and the int8 MFMA kernels.
#include "common.h"
#include "kernels.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace snnqp;

// A block that IMPL_AUTO hands to the direct-form kernel runs 20-25 x slower than on the MFMA
// kernels (one thread per output neuron): counted, with the reason, so that a caller (and
// bench.py's JSON line) can see the cliff instead of guessing at it.
namespace {
std::atomic<int64_t> g_fallback_conv{0}, g_fallback_dense{0};
std::mutex g_fallback_mu;
char g_fallback_reason[256] = "";

void note_fallback(bool dense, const char *why) {
  (dense ? g_fallback_dense : g_fallback_conv).fetch_add(1, std::memory_order_relaxed);
  static const bool log = std::getenv("SNNQP_LOG_FALLBACKS") != nullptr;
  std::lock_guard<std::mutex> lock(g_fallback_mu);
  const bool fresh = std::strncmp(g_fallback_reason + (dense ? 7 : 6), why ? why : "", 200) != 0;
  std::snprintf(g_fallback_reason, sizeof(g_fallback_reason), "%s%s", dense ? "dense: " : "conv: ",
                why ? why : "");
  if (log && fresh)
    std::fprintf(stderr, "libsnnqp: %s block on the direct-form kernel: %s\n",
                 dense ? "dense" : "conv", why ? why : "");
}
}  // namespace

extern "C" {

int snnqp_fallback_counts(int64_t *conv_blocks, int64_t *dense_blocks, char *reason,
                          int32_t reason_len, int reset) {
  if (conv_blocks) *conv_blocks = g_fallback_conv.load(std::memory_order_relaxed);
  if (dense_blocks) *dense_blocks = g_fallback_dense.load(std::memory_order_relaxed);
  std::lock_guard<std::mutex> lock(g_fallback_mu);
  if (reason && reason_len > 0) {
    std::strncpy(reason, g_fallback_reason, (size_t)reason_len - 1);
    reason[reason_len - 1] = 0;
  }
  if (reset) {
    g_fallback_conv = 0;
    g_fallback_dense = 0;
    g_fallback_reason[0] = 0;
  }
  return SNNQP_OK;
}

int snnqp_conv_dequant_form(const snnqp_weight_t *w, const snnqp_neuron_t *nrn) {
  SNNQP_REQUIRE(w && nrn, SNNQP_EINVAL, "conv_dequant_form: null descriptor");
  SNNQP_REQUIRE(w->wtype == SNNQP_W_I8, SNNQP_EINVAL, "conv_dequant_form: int8 codes only");
  return conv3x3_bits_dequant_form(w, nrn);
}

int snnqp_conv_lif_forward(const void *x, int in_type, int64_t x_stride_t,
                           int64_t x_stride_b, int32_t T, int32_t B,
                           const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                           const int8_t *wt, const snnqp_bn_t *bn,
                           const snnqp_neuron_t *nrn,
                           const float *u0, float *u_out, void *s_out,
                           int s_type, int pool, int impl, int x_max, int32_t *x_seen,
                           int32_t *x_flags, snnqp_stream_t stream) {
  SNNQP_REQUIRE(g && w && nrn, SNNQP_EINVAL, "conv_lif_forward: null descriptor");
  if (in_type == SNNQP_F32 && w->wtype == SNNQP_W_I8)
    SNNQP_REQUIRE(x_flags != nullptr, SNNQP_EINVAL,
                  "conv_lif_forward: float32 input into integer codes needs x_flags (snnqp.h)");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "conv_lif_forward")) return rc;
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF &&
                    nrn->kind <= SNNQP_NEURON_LIF,
                SNNQP_EINVAL, "conv_lif_forward: unknown neuron kind %d", nrn->kind);
  SNNQP_REQUIRE(pool == 1 || pool == 2, SNNQP_EINVAL,
                "conv_lif_forward: pool must be 1 or 2");
  SNNQP_REQUIRE(impl >= SNNQP_IMPL_AUTO && impl <= SNNQP_IMPL_MFMA, SNNQP_EINVAL,
                "conv_lif_forward: unknown impl %d", impl);
  const char *why = conv3x3_mfma_unsupported(in_type, g, w, wt, nrn, s_type);
  if (impl == SNNQP_IMPL_MFMA)
    SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "conv_lif_forward: MFMA kernel: %s", why);
  if (!why && impl != SNNQP_IMPL_GENERIC)
    return run_conv3x3_mfma(x, in_type, x_stride_t, x_stride_b, T, B, g, w, wt, bn,
                            nrn, u0, u_out, (uint32_t *)s_out, pool, x_max, x_seen, x_flags,
                            (hipStream_t)stream);
  SNNQP_REQUIRE(!(in_type == SNNQP_F32 && w->wtype == SNNQP_W_I8), SNNQP_EUNSUPPORTED,
                "conv_lif_forward: float32 input into integer codes is staged by the event-layer MFMA "
                "kernel only (%s); narrow it first (snnqp_narrow_f32 / snnqp_pack_bits_checked)",
                why ? why : "impl = GENERIC");
  SNNQP_REQUIRE(pool == 1, SNNQP_EUNSUPPORTED,
                "conv_lif_forward: the direct-form kernel does not fuse the "
                "max-pool; call snnqp_maxpool2x2 after it");
  if (impl == SNNQP_IMPL_AUTO) note_fallback(false, why);
  return run_generic(x, in_type, x_stride_t, x_stride_b, T, B, g, w, bn, nrn, u0,
                     u_out, s_out, s_type, nullptr, (hipStream_t)stream);
}

int snnqp_conv_lif_forward_pred(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                                int64_t x_stride_b, int32_t T, int32_t B,
                                const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                                const int8_t *wt, const snnqp_bn_t *bn,
                                const snnqp_neuron_t *nrn,
                                const float *u0, float *u_out, void *s_out,
                                int s_type, int pool, int x_max, int32_t *x_seen,
                                int32_t *x_flags, snnqp_stream_t stream) {
  SNNQP_REQUIRE(pred && g && w && nrn, SNNQP_EINVAL, "conv_lif_forward_pred: null argument");
  if (in_type == SNNQP_F32 && w->wtype == SNNQP_W_I8)
    SNNQP_REQUIRE(x_flags != nullptr, SNNQP_EINVAL,
                  "conv_lif_forward_pred: float32 input into integer codes needs x_flags (snnqp.h)");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "conv_lif_forward_pred")) return rc;
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF && nrn->kind <= SNNQP_NEURON_LIF, SNNQP_EINVAL,
                "conv_lif_forward_pred: unknown neuron kind %d", nrn->kind);
  SNNQP_REQUIRE(pool == 1 || pool == 2, SNNQP_EINVAL, "conv_lif_forward_pred: pool must be 1 or 2");
  SNNQP_REQUIRE(in_type == SNNQP_U8 || in_type == SNNQP_F32 || in_type == SNNQP_EV4, SNNQP_EUNSUPPORTED,
                "conv_lif_forward_pred: byte, nibble or float32 frames (the event layer's own formats)");
  const char *why = conv3x3_mfma_unsupported(in_type, g, w, wt, nrn, s_type);
  SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "conv_lif_forward_pred: MFMA kernel: %s", why);
  return run_conv3x3_mfma(x, in_type, x_stride_t, x_stride_b, T, B, g, w, wt, bn, nrn, u0, u_out,
                          (uint32_t *)s_out, pool, x_max, x_seen, x_flags, (hipStream_t)stream, pred);
}

int snnqp_conv_lif_forward_if(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                              int64_t x_stride_b, int32_t T, int32_t B,
                              const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                              const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                              float *u_out, void *s_out, int s_type, int pool,
                              snnqp_stream_t stream) {
  SNNQP_REQUIRE(pred && g && w && nrn, SNNQP_EINVAL, "conv_lif_forward_if: null argument");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "conv_lif_forward_if")) return rc;
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF && nrn->kind <= SNNQP_NEURON_LIF,
                SNNQP_EINVAL, "conv_lif_forward_if: unknown neuron kind %d", nrn->kind);
  SNNQP_REQUIRE(pool == 1 || pool == 2, SNNQP_EINVAL, "conv_lif_forward_if: pool must be 1 or 2");
  return run_generic(x, in_type, x_stride_t, x_stride_b, T, B, g, w, bn, nrn, u0, u_out, s_out, s_type,
                     nullptr, (hipStream_t)stream, pool, pred);
}

int snnqp_dense_lif_forward_if(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                               int64_t x_stride_b, int32_t T, int32_t B, int32_t K, int32_t N,
                               const snnqp_weight_t *w, const snnqp_bn_t *bn,
                               const snnqp_neuron_t *nrn, const float *u0, float *u_out,
                               void *s_out, int s_type, snnqp_stream_t stream) {
  SNNQP_REQUIRE(pred && w && nrn && K > 0 && N > 0, SNNQP_EINVAL, "dense_lif_forward_if: bad argument");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "dense_lif_forward_if")) return rc;
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF && nrn->kind <= SNNQP_NEURON_LIF,
                SNNQP_EINVAL, "dense_lif_forward_if: unknown neuron kind %d", nrn->kind);
  snnqp_conv_geom_t g;
  g.H = 1; g.W = 1; g.Cin = K; g.Cout = N; g.KH = 1; g.KW = 1;
  g.stride_h = g.stride_w = 1;
  g.pad_h_lo = g.pad_h_hi = g.pad_w_lo = g.pad_w_hi = 0;
  g.in_dil_h = g.in_dil_w = g.k_dil_h = g.k_dil_w = 1;
  g.groups = 1;
  return run_generic(x, in_type, x_stride_t, x_stride_b, T, B, &g, w, bn, nrn, u0, u_out, s_out, s_type,
                     nullptr, (hipStream_t)stream, 1, pred);
}

int snnqp_current_min(const snnqp_weight_t *w, const snnqp_bn_t *bn, int32_t bound,
                      int32_t Cout, uint32_t *out_bits, snnqp_stream_t stream) {
  return run_current_min(w, bn, bound, Cout, out_bits, (hipStream_t)stream);
}

int64_t snnqp_dense_workspace_bytes(int in_type, int32_t T, int32_t B, int32_t K, int32_t N,
                                    const snnqp_weight_t *w) {
  if (!w || in_type != SNNQP_BITS || w->wtype != SNNQP_W_I8 || !w->wt_fp6 || w->code_max <= 0 ||
      w->code_max > 7 || K <= 0 || N <= 0)
    return 0;
  return dense_fp6_workspace_bytes(T, B, K, N);
}

int snnqp_dense_lif_forward(const void *x, int in_type, int64_t x_stride_t,
                            int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                            int32_t N, const snnqp_weight_t *w,
                            const int8_t *wt, const snnqp_bn_t *bn,
                            const snnqp_neuron_t *nrn, const float *u0,
                            float *u_out, void *s_out, int s_type, int impl,
                            snnqp_stream_t stream) {
  return snnqp_dense_lif_forward_ws(x, in_type, x_stride_t, x_stride_b, T, B, K, N, w, wt, bn, nrn, u0,
                                    u_out, s_out, s_type, impl, nullptr, nullptr, 0, stream);
}

int snnqp_dense_lif_forward_ws(const void *x, int in_type, int64_t x_stride_t,
                               int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                               int32_t N, const snnqp_weight_t *w,
                               const int8_t *wt, const snnqp_bn_t *bn,
                               const snnqp_neuron_t *nrn, const float *u0,
                               float *u_out, void *s_out, int s_type, int impl,
                               int32_t *x_flags, void *ws, int64_t ws_bytes, snnqp_stream_t stream) {
  SNNQP_REQUIRE(w && nrn, SNNQP_EINVAL, "dense_lif_forward: null descriptor");
  const bool f32_int = in_type == SNNQP_F32 && w->wtype == SNNQP_W_I8;
  if (f32_int)
    SNNQP_REQUIRE(x_flags != nullptr, SNNQP_EINVAL,
                  "dense_lif_forward: float32 rows into integer codes need x_flags (snnqp.h)");
  if (int rc = refuse_after_device_report((hipStream_t)stream, "dense_lif_forward")) return rc;
  SNNQP_REQUIRE(K > 0 && N > 0, SNNQP_EINVAL, "dense_lif_forward: bad K/N");
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF &&
                    nrn->kind <= SNNQP_NEURON_LIF,
                SNNQP_EINVAL, "dense_lif_forward: unknown neuron kind %d", nrn->kind);
  SNNQP_REQUIRE(impl >= SNNQP_IMPL_AUTO && impl <= SNNQP_IMPL_MFMA, SNNQP_EINVAL,
                "dense_lif_forward: unknown impl %d", impl);
  // codes that fit fp6, packed as fp6 tiles, over bit-packed rows: the f8f6f4 kernel
  if (impl != SNNQP_IMPL_GENERIC && in_type == SNNQP_BITS && s_type == SNNQP_BITS &&
      w->wtype == SNNQP_W_I8 && w->wt_fp6 && w->code_max > 0 && w->code_max <= 7 && T <= 160 &&
      (int64_t)7 * K < ((int64_t)1 << 24) &&      // every partial sum an integer float32 holds exactly
      !(nrn->kind == SNNQP_NEURON_LIF && !nrn->decay) && x_stride_t >= 0 && x_stride_b >= 0 &&
      (int64_t)(T > 0 ? T - 1 : 0) * x_stride_t + 160 * x_stride_b + (K + 31) / 32 < ((int64_t)1 << 31))
    return run_dense_fp6(x, x_stride_t, x_stride_b, T, B, K, N, w, bn, nrn, u0, u_out,
                         (uint32_t *)s_out, 0, ws, ws_bytes, (hipStream_t)stream);
  // more than 128 features: a workgroup per 256 / 512-column block, every row read once
  if (impl != SNNQP_IMPL_GENERIC &&
      !dense_wide_unsupported(in_type, T, K, N, x_stride_t, x_stride_b, x, w, wt, nrn, s_type))
    return run_dense_wide(x, in_type, x_stride_t, x_stride_b, T, B, K, N, w, wt, bn, nrn, u0, u_out,
                          (uint32_t *)s_out, x_flags, (hipStream_t)stream);
  SNNQP_REQUIRE(!f32_int, SNNQP_EUNSUPPORTED,
                "dense_lif_forward: float32 rows into integer codes are staged by the wide MFMA kernel only "
                "(more than 128 features, T <= 64, K %% 16 == 0, 16-byte aligned rows); narrow them first "
                "(snnqp_narrow_f32)");
  const char *why = dense_mfma_unsupported(in_type, K, N, w, wt, nrn, s_type);
  if (!why && T > (in_type == SNNQP_U8 ? 64 : 96))
    why = "more than 96 (uint8 input: 64) timesteps (one sample must fit a row tile)";
  if (!why && in_type == SNNQP_U8 &&
      ((((uintptr_t)x) & 15) != 0 || x_stride_t % 16 != 0 || x_stride_b % 16 != 0))
    why = "uint8 rows not 16-byte aligned";
  // the fused kernel addresses the rows of one workgroup (at most 96 samples) with 32-bit
  // word offsets from the workgroup's first sample
  if (!why && ((int64_t)(T > 0 ? T - 1 : 0) * x_stride_t + 96 * x_stride_b + (K + 31) / 32 >= ((int64_t)1 << 31) ||
               x_stride_t < 0 || x_stride_b < 0))
    why = "input strides beyond 32-bit word offsets within a workgroup";
  if (impl == SNNQP_IMPL_MFMA)
    SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "dense_lif_forward: MFMA kernel: %s", why);
  if (!why && impl != SNNQP_IMPL_GENERIC)
    return run_dense_mfma(x, in_type, x_stride_t, x_stride_b, T, B, K, N, w, wt, bn, nrn,
                          u0, u_out, (uint32_t *)s_out, (hipStream_t)stream);
  if (impl == SNNQP_IMPL_AUTO) note_fallback(true, why);
  snnqp_conv_geom_t g;
  g.H = 1; g.W = 1; g.Cin = K; g.Cout = N; g.KH = 1; g.KW = 1;
  g.stride_h = g.stride_w = 1;
  g.pad_h_lo = g.pad_h_hi = g.pad_w_lo = g.pad_w_hi = 0;
  g.in_dil_h = g.in_dil_w = g.k_dil_h = g.k_dil_w = 1;
  g.groups = 1;
  return run_generic(x, in_type, x_stride_t, x_stride_b, T, B, &g, w, bn, nrn, u0,
                     u_out, s_out, s_type, nullptr, (hipStream_t)stream);
}

}  // extern "C"
