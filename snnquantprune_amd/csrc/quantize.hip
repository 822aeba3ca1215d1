// Weight transforms: quantiser forwards (quant.py:322-625) + prune mask
// (quant.py:472-491), emitted as float32 fake-quantised weights and/or int8
// codes.  Element-wise, HBM-bound, run once per weight version (the "pack"
// step that replaces the reference's per-timestep fake-quant).
#include "common.h"

namespace snnqp {

struct QuantP {
  int kind;
  float p0, p1, L, scale;
  float lo;     // lower clip bound in units of the range: -1 (sign = True) or 0 (sign = False)
};

__device__ __forceinline__ float clipf(float x, float lo, float hi) {
  return fminf(fmaxf(x, lo), hi);
}

// Returns the integer code (integer-valued float) and the dequantised value.
__device__ __forceinline__ void quant_one(float w, const QuantP &p, float &q,
                                          float &fq) {
  switch (p.kind) {
    case SNNQP_Q_DUQ: {                       // quant.py:443,466-467
      float x = clipf(w / p.p0, -1.0f, 1.0f); // hard_tanh(x / a)
      q = rintf(x * p.L);
      fq = (q / p.L) * p.p1;
      break;
    }
    case SNNQP_Q_UNIFORM_STATIC: {            // quant.py:350-358
      float x = clipf(w / p.p0, p.lo, 1.0f) * p.p0;
      q = rintf(x / p.scale);
      fq = q * p.scale;
      break;
    }
    case SNNQP_Q_PARAMETRIC_D: {              // quant.py:420-425
      float v = clipf(w / p.p0, p.lo * p.L, p.L);   // q_neg = -q_pos or 0, quant.py:378-384
      q = rintf(v);
      fq = q * p.p0;
      break;
    }
    default: {                                // parametric_d_xmax, quant.py:617-625
      float x = clipf(w / p.p1, p.lo, 1.0f) * p.p1;
      q = rintf(x / p.p0);
      fq = p.p0 * q;
      break;
    }
  }
}

__global__ void __launch_bounds__(256)
quantize_kernel(const float *__restrict__ w, const float *__restrict__ mask,
                int64_t n, QuantP p, float *__restrict__ fq_out,
                int8_t *__restrict__ codes_out, int32_t *__restrict__ flags) {
  int32_t f = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float q, fq;
    quant_one(w[i], p, q, fq);
    if (mask) {
      const float mk = mask[i];
      if (mk != 0.0f && mk != 1.0f) f |= SNNQP_FLAG_MASK_NOT_BINARY;
      fq = fq * mk;                          // quant.py:491
      q = (mk == 0.0f) ? 0.0f : q;
    }
    if (fq_out) fq_out[i] = fq;
    if (codes_out) {
      if (!(fabsf(q) <= 127.0f)) f |= SNNQP_FLAG_CODE_OVERFLOW;
      codes_out[i] = (int8_t)clipf(q, -127.0f, 127.0f);
    }
  }
  if (flags && f) atomicOr(flags, f);
}

// MFMA B-operand tiles: for the 32-column block nb and the 32-deep k-step ks,
// lane l = (n & 31) + 32 * h holds bytes k = 32 ks + 16 h + j (j < 16) of column
// n = 32 nb + (l & 31): one contiguous 1 KiB read per wave and k-step.
__global__ void __launch_bounds__(256)
pack_codes_mfma_kernel(const int8_t *__restrict__ w, int64_t K, int32_t N,
                       int32_t Npad, int8_t *__restrict__ wt) {
  const int64_t total = (int64_t)Npad * K;
  const int64_t KS = K / 32;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 15);
    const int lane = (int)((i >> 4) & 63);
    const int64_t tile = i >> 10;
    const int64_t ks = tile % KS, nb = tile / KS;
    const int64_t n = nb * 32 + (lane & 31);
    const int64_t k = ks * 32 + 16 * (lane >> 5) + j;
    wt[i] = n < N ? w[k * N + n] : (int8_t)0;
  }
}

}  // namespace snnqp

extern "C" int snnqp_pack_codes_mfma(const int8_t *w, int64_t K, int32_t N,
                                     int32_t Npad, int8_t *wt,
                                     snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w && wt && K > 0 && N > 0, SNNQP_EINVAL, "pack_codes_mfma: bad argument");
  SNNQP_REQUIRE((K & 31) == 0 && Npad >= N && (Npad & 31) == 0, SNNQP_EINVAL,
                "pack_codes_mfma: K and Npad must be multiples of 32, Npad >= N");
  const int64_t blocks = ceil_div64((int64_t)Npad * K, 256);
  hipLaunchKernelGGL(pack_codes_mfma_kernel, dim3((int)(blocks < 4096 ? blocks : 4096)),
                     dim3(256), 0, (hipStream_t)stream, w, K, N, Npad, wt);
  SNNQP_CHECK_LAUNCH("pack_codes_mfma_kernel");
  return SNNQP_OK;
}

extern "C" int snnqp_quantize_ex(int kind, const float *w, const float *mask,
                                 int64_t n, int bits, int sign, float p0, float p1,
                                 float *fq_out, int8_t *codes_out, int32_t *flags,
                                 snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "quantize: negative size");
  SNNQP_REQUIRE(kind >= SNNQP_Q_DUQ && kind <= SNNQP_Q_PARAMETRIC_D_XMAX,
                SNNQP_EINVAL, "quantize: unknown quantiser %d", kind);
  // quant.py:332-336 "Bit widths below 2 bits are not supported"
  SNNQP_REQUIRE(bits > 1 && bits <= 24, SNNQP_EINVAL,
                "quantize: bits must be in [2, 24], got %d", bits);
  // unsigned levels 2^bits - 1 (quant.py:338-341, :378-384, :458-461, :532-535): float32 holds them
  // up to 2^24; int8 codes only when they fit (the kernel flags SNNQP_FLAG_CODE_OVERFLOW otherwise)
  SNNQP_REQUIRE(sign || bits <= 23, SNNQP_EINVAL,
                "quantize: unsigned bits must be in [2, 23], got %d", bits);
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(w, SNNQP_EINVAL, "quantize: null weights");
  QuantP p;
  p.kind = kind;
  p.p0 = p0;
  p.p1 = p1;
  p.L = sign ? (float)((1 << (bits - 1)) - 1) : (float)((1 << bits) - 1);
  // DuQ clips with hard_tanh whatever the sign (quant.py:466): only its level count changes
  p.lo = (sign || kind == SNNQP_Q_DUQ) ? -1.0f : 0.0f;
  p.scale = p0 / p.L;  // uniform_static: xmax / num_levels, quant.py:357
  const int64_t blocks = ceil_div64(n, 256);
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  hipLaunchKernelGGL(quantize_kernel, dim3(grid), dim3(256), 0,
                     (hipStream_t)stream, w, mask, n, p, fq_out, codes_out,
                     flags);
  SNNQP_CHECK_LAUNCH("quantize_kernel");
  return SNNQP_OK;
}

extern "C" int snnqp_quantize(int kind, const float *w, const float *mask,
                              int64_t n, int bits, float p0, float p1,
                              float *fq_out, int8_t *codes_out, int32_t *flags,
                              snnqp_stream_t stream) {
  return snnqp_quantize_ex(kind, w, mask, n, bits, 1, p0, p1, fq_out, codes_out, flags, stream);
}
