// Stand-alone neuron scan over T (spiking_learning.py:357-438 applied by the
// scan at :446-462) and eval-mode BatchNorm (models.py:101-107).  Used when a
// SpikingBlock's connection is not one of the fused kinds.  HBM-bound: each
// current is read once, each spike written once, u stays in a register.
#include "kernels.h"

namespace snnqp {

__global__ void __launch_bounds__(256)
lif_scan_kernel(const float *__restrict__ x, int32_t T, int64_t n, int32_t C,
                BnP bn, NeuronP nrn, const float *__restrict__ u0,
                float *__restrict__ u_out, void *__restrict__ s_out,
                int32_t s_type) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < n;
  const int32_t c = (int32_t)((live ? idx : 0) % C);
  float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f;
  const bool has_bn = bn.mean != nullptr;
  if (has_bn) { bmean = bn.mean[c]; bmul = bn.mul[c]; bbias = bn.bias[c]; }
  if (nrn.kind == SNNQP_NEURON_LIF) dec = nrn.decay[c];
  float u = (live && u0) ? u0[idx] : 0.0f;
  const bool word_aligned = (C & 31) == 0;
  const int32_t CW = (C + 31) / 32;
  for (int32_t t = 0; t < T; ++t) {
    const int64_t o = (int64_t)t * n + idx;
    bool s = false;
    if (live) {
      float cur = x[o];
      if (has_bn) cur = bn_apply(cur, bmean, bmul, bbias);
      s = neuron_step(u, cur, nrn, dec);
    }
    if (s_type == SNNQP_F32) {
      if (live) ((float *)s_out)[o] = s ? 1.0f : 0.0f;
    } else if (word_aligned) {
      const unsigned long long m = __ballot(s);
      const int lane = threadIdx.x & 63;
      if (live && (lane & 31) == 0)
        ((uint32_t *)s_out)[o >> 5] = (uint32_t)(lane ? (m >> 32) : m);
    } else if (live && s) {
      const int64_t row = idx / C, rows = n / C;
      atomicOr(&((uint32_t *)s_out)[((int64_t)t * rows + row) * CW + (c >> 5)],
               1u << (c & 31));
    }
  }
  if (live && u_out) u_out[idx] = u;
}

__global__ void __launch_bounds__(256)
batchnorm_kernel(const float *__restrict__ x, int64_t n, int32_t C, BnP bn,
                 float *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = (int32_t)(i % C);
    y[i] = bn_apply(x[i], bn.mean[c], bn.mul[c], bn.bias[c]);
  }
}

}  // namespace snnqp

using namespace snnqp;

extern "C" {

int snnqp_lif_forward(const float *x, int32_t T, int64_t R, int32_t C,
                      const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                      const float *u0, float *u_out, void *s_out, int s_type,
                      snnqp_stream_t stream) {
  SNNQP_REQUIRE(T >= 0 && R >= 0 && C > 0, SNNQP_EINVAL, "lif_forward: bad shape");
  SNNQP_REQUIRE(nrn && ((x && s_out) || T == 0 || R == 0), SNNQP_EINVAL, "lif_forward: null argument");
  SNNQP_REQUIRE(nrn->kind >= SNNQP_NEURON_MULTI_STEP_LIF &&
                    nrn->kind <= SNNQP_NEURON_LIF,
                SNNQP_EINVAL, "lif_forward: unknown neuron kind %d", nrn->kind);
  SNNQP_REQUIRE(nrn->kind != SNNQP_NEURON_LIF || nrn->decay, SNNQP_EINVAL,
                "lif_forward: LIF neuron needs a decay vector");
  SNNQP_REQUIRE(s_type == SNNQP_F32 || s_type == SNNQP_BITS, SNNQP_EINVAL,
                "lif_forward: spike output type must be F32 or BITS");
  SNNQP_CHECK_BN(bn);
  const int64_t n = R * C;
  if (n == 0 || T == 0) return SNNQP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (s_type == SNNQP_BITS && (C & 31) != 0)
    // (a kernel of the library, not a memset node: kernels.h zero_words_async)
    if (int rc = zero_words_async((uint32_t *)s_out, (int64_t)T * R * ((C + 31) / 32), st)) return rc;
  const int64_t blocks = ceil_div64(n, 256);
  SNNQP_REQUIRE(blocks < (1ll << 31), SNNQP_EINVAL, "lif_forward: grid too large");
  hipLaunchKernelGGL(lif_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x,
                     T, n, C, make_bn(bn), make_neuron(nrn), u0, u_out, s_out,
                     s_type);
  SNNQP_CHECK_LAUNCH("lif_scan_kernel");
  return SNNQP_OK;
}

int snnqp_batchnorm_forward(const float *x, int64_t rows, int32_t C,
                            const snnqp_bn_t *bn, float *y,
                            snnqp_stream_t stream) {
  SNNQP_REQUIRE(bn && ((x && y) || rows == 0), SNNQP_EINVAL, "batchnorm_forward: null argument");
  SNNQP_CHECK_BN(bn);
  SNNQP_REQUIRE(rows >= 0 && C > 0, SNNQP_EINVAL, "batchnorm_forward: bad shape");
  const int64_t n = rows * C;
  if (n == 0) return SNNQP_OK;
  int64_t blocks = ceil_div64(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(batchnorm_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, x, n, C, make_bn(bn), y);
  SNNQP_CHECK_LAUNCH("batchnorm_kernel");
  return SNNQP_OK;
}

}  // extern "C"
