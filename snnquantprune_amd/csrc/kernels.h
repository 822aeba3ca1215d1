// Internal declarations shared by the .hip translation units of libsnnqp.
#pragma once
#include "common.h"

namespace snnqp {

// depth axis of a 3-D convolution for the direct-form kernel (OD resolved by the caller)
struct GenericDepth {
  int32_t D, KD, OD, stride, pad_lo, in_dil, k_dil;
};
int run_generic(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T,
                int32_t B, const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                float *u_out, void *s_out, int s_type, int32_t *acc_out,
                hipStream_t st, int pool = 1, const int32_t *pred = nullptr,
                const GenericDepth *dz = nullptr);

// nullptr when the MFMA kernel can serve the request, else the reason.
const char *conv3x3_mfma_unsupported(int in_type, const snnqp_conv_geom_t *g,
                                     const snnqp_weight_t *w, const int8_t *wt,
                                     const snnqp_neuron_t *nrn, int s_type);
int run_current_min(const snnqp_weight_t *w, const snnqp_bn_t *bn, int32_t bound, int32_t Cout,
                    uint32_t *out_bits, hipStream_t st);
int conv3x3_bits_dequant_form(const snnqp_weight_t *w, const snnqp_neuron_t *nrn);
int run_conv3x3_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b,
                     int32_t T, int32_t B, const snnqp_conv_geom_t *g,
                     const snnqp_weight_t *w, const int8_t *wt,
                     const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                     const float *u0, float *u_out, uint32_t *s_out, int pool,
                     int x_max, int32_t *x_seen, int32_t *x_flags, hipStream_t st,
                     const int32_t *pred = nullptr);

// per-device state (runtime.hip): the status word kernels report broken invariants into, the
// probe of the matrix pipe's denormal arithmetic behind DQ_TABLE
uint32_t *device_status_word(int dev);
// `nwords` 32-bit words at `p` := 0 on `st`, by a kernel of the library (a kernel node when the
// stream is being captured).  Not hipMemsetAsync: on replay the runtime (ROCm 7.2.0) executed a
// small memset node of one graph with stale data -- the node was present and correctly addressed
// (hipGraphMemsetNodeGetParams), yet its target received the argument block of the fill kernel of
// ANOTHER graph's memset node ({dst, pattern chunk, dst}, the chunk advancing per replay) instead
// of zeros: profiles/r06_capture_memset_nodes.txt (tools/capture_ws_debug.py --dump); round 3 met
// the same with the conv kernels' queue slots.  A kernel node carries its pointer in its own
// argument block and always has.
int zero_words_async(uint32_t *p, int64_t nwords, hipStream_t st);
uint32_t device_status_read(int dev);
int refuse_after_device_report(hipStream_t st, const char *who);   // SNNQP_EHIP while the status word is set
const char *device_status_text(uint32_t code);
bool dq_table_trusted(int dev, hipStream_t st);
void check_code_bound_once(int dev, const int8_t *w, int64_t K, int32_t N, int32_t bound, hipStream_t st);
int64_t dq_table_fallbacks(bool reset);
int stream_device(hipStream_t st);

// float32 x float32 connection on the f32 MFMA (fseq_gemm.hip)
const char *fseq_gemm_unsupported(int in_type, const snnqp_conv_geom_t *g,
                                  const snnqp_weight_t *w);
int run_fseq_gemm(const void *x, int in_type, int64_t NB, const snnqp_conv_geom_t *g,
                  const snnqp_weight_t *w, float *y, hipStream_t st);

const char *dense_mfma_unsupported(int in_type, int32_t K, int32_t N,
                                   const snnqp_weight_t *w, const int8_t *wt,
                                   const snnqp_neuron_t *nrn, int s_type);
int run_dense_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B,
                   int32_t K, int32_t N, const snnqp_weight_t *w, const int8_t *wt,
                   const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                   float *u_out, uint32_t *s_out, hipStream_t st);

// all the columns of a 256 / 512-column block per workgroup, neuron from the accumulator
// registers (dense_wide.hip); nullptr when it can serve the request
const char *dense_wide_unsupported(int in_type, int32_t T, int32_t K, int32_t N, int64_t xs_t,
                                   int64_t xs_b, const void *x, const snnqp_weight_t *w,
                                   const int8_t *wt, const snnqp_neuron_t *nrn, int s_type);
int run_dense_wide(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B,
                   int32_t K, int32_t N, const snnqp_weight_t *w, const int8_t *wt,
                   const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                   float *u_out, uint32_t *s_out, int32_t *x_flags, hipStream_t st);

// 3x3 convolution on gate x raster (conv_gated.hip); nullptr when it can serve the request
const char *conv_gated_unsupported(const snnqp_conv_geom_t *g, const snnqp_weight_t *w);
// dense connection on the channel-major flattening of gate x raster (dense_gated.hip)
const char *dense_gated_unsupported(int32_t HW, int32_t C, int32_t N, const snnqp_weight_t *w);

// codes of magnitude <= 7 on the f8f6f4 MFMA (dense_fp6.hip); row_tiles 0 = choose
// (ws / ws_bytes: optional workspace for a K split over workgroups, dense_fp6_workspace_bytes)
int run_dense_fp6(const void *x, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B, int32_t K,
                  int32_t N, const snnqp_weight_t *w, const snnqp_bn_t *bn,
                  const snnqp_neuron_t *nrn, const float *u0, float *u_out, uint32_t *s_out,
                  int row_tiles, void *ws, int64_t ws_bytes, hipStream_t st);
int64_t dense_fp6_workspace_bytes(int32_t T, int32_t B, int32_t K, int32_t N);

}  // namespace snnqp
