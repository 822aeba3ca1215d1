// Activation-format helpers and the small element-wise ops around the blocks:
// float32 <-> u8 / bit-packed spikes, 2x2 max-pool (models.py:145-147), rate
// vote (models.py:253-255).  All HBM-bound single-pass kernels.
#include "kernels.h"

namespace snnqp {

__global__ void __launch_bounds__(256)
inspect_f32_kernel(const float *__restrict__ x, int64_t n,
                   int32_t *__restrict__ flags) {
  int32_t f = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (!(v >= 0.0f && v <= 255.0f) || v != rintf(v)) f |= SNNQP_FLAG_NOT_INTEGER;
    if (v > 1.0f) f |= SNNQP_FLAG_GT_ONE;
    if (v > 127.0f) f |= SNNQP_FLAG_GT_127;
  }
  if (f) atomicOr(flags, f);
}

// Maximum of a u8 tensor: 16 bytes per lane and step; the bytes of a word are reduced
// as two packed u16 pairs (v_pk_max_u16), so the pass stays HBM-bound.
__global__ void __launch_bounds__(256)
inspect_u8_kernel(const uint8_t *__restrict__ x, int64_t n, int32_t *__restrict__ flags) {
  typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  v2u16 m0 = {0, 0}, m1 = {0, 0};
  // aligned body
  const uintptr_t addr = (uintptr_t)x;
  const int64_t head = (int64_t)(((addr + 15) & ~(uintptr_t)15) - addr) < n
                           ? (int64_t)(((addr + 15) & ~(uintptr_t)15) - addr) : n;
  const int64_t nvec = (n - head) / 16;
  const v4u *xv = (const v4u *)(x + head);
  auto fold = [&](const v4u &w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = w[j] & 0x00FF00FFu, hi = (w[j] >> 8) & 0x00FF00FFu;
      m0 = __builtin_elementwise_max(m0, __builtin_bit_cast(v2u16, lo));
      m1 = __builtin_elementwise_max(m1, __builtin_bit_cast(v2u16, hi));
    }
  };
  int64_t i = tid;
  for (; i + 3 * nthreads < nvec; i += 4 * nthreads) {      // four loads in flight
    const v4u w0 = xv[i], w1 = xv[i + nthreads], w2 = xv[i + 2 * nthreads],
              w3 = xv[i + 3 * nthreads];
    fold(w0); fold(w1); fold(w2); fold(w3);
  }
  for (; i < nvec; i += nthreads) fold(xv[i]);
  int32_t m = max(max((int)m0.x, (int)m0.y), max((int)m1.x, (int)m1.y));
  // unaligned head and tail bytes
  for (int64_t k = tid; k < head; k += nthreads) m = max(m, (int32_t)x[k]);
  for (int64_t k = head + nvec * 16 + tid; k < n; k += nthreads) m = max(m, (int32_t)x[k]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  // one atomic per workgroup (they serialise on the single word)
  __shared__ int32_t wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    // word = (max << 8) | flags(max) grows with max, so one atomicMax keeps both
    const int32_t f = (m > 1 ? SNNQP_FLAG_GT_ONE : 0) | (m > 127 ? SNNQP_FLAG_GT_127 : 0);
    if (m) atomicMax(flags, (m << 8) | f);
  }
}

// One pass over a float32 activation tensor: the uint8 copy (meaningful when no
// element is flagged NOT_INTEGER) and the word (max << 8) | flags of inspect_f32 /
// inspect_u8 at once -- a float32 frame batch is read once instead of three times.
__global__ void __launch_bounds__(256)
narrow_f32_kernel(const float *__restrict__ x, uint8_t *__restrict__ y, int64_t n,
                  int32_t *__restrict__ flags) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const bool vec = (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 3) == 0);
  const int64_t nvec = vec ? n / 4 : 0;
  int32_t m = 0, bad = 0;
  auto one = [&](float v) -> uint32_t {
    if (!(v >= 0.0f && v <= 255.0f) || v != rintf(v)) bad = 1;
    const int iv = (int)v;
    m = max(m, iv < 0 ? 0 : (iv > 255 ? 255 : iv));
    return (uint32_t)(uint8_t)v;
  };
  for (int64_t i = tid; i < nvec; i += nthreads) {
    const v4f w = ((const v4f *)x)[i];
    ((uint32_t *)y)[i] = one(w.x) | (one(w.y) << 8) | (one(w.z) << 16) | (one(w.w) << 24);
  }
  for (int64_t i = nvec * 4 + tid; i < n; i += nthreads) y[i] = (uint8_t)one(x[i]);
  for (int o = 32; o > 0; o >>= 1) {
    m = max(m, __shfl_xor(m, o));
    bad |= __shfl_xor(bad, o);
  }
  __shared__ int32_t wm[4], wb[4];
  if ((threadIdx.x & 63) == 0) { wm[threadIdx.x >> 6] = m; wb[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    bad = wb[0] | wb[1] | wb[2] | wb[3];
    const int32_t f = (m > 1 ? SNNQP_FLAG_GT_ONE : 0) | (m > 127 ? SNNQP_FLAG_GT_127 : 0);
    if (m) atomicMax(flags, (m << 8) | f);      // grows with max: one word keeps both
    if (bad) atomicOr(flags + 1, SNNQP_FLAG_NOT_INTEGER);
  }
}

__global__ void __launch_bounds__(256)
f32_to_u8_kernel(const float *__restrict__ x, uint8_t *__restrict__ y,
                 int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (uint8_t)x[i];
}

// One wave packs 64 consecutive channels of one row with a ballot.
template <typename T>
__global__ void __launch_bounds__(256)
pack_bits_kernel(const T *__restrict__ x, int64_t rows, int32_t C, int32_t CW,
                 uint32_t *__restrict__ bits, int32_t *__restrict__ flags) {
  bool bad = false;                  // a value that is neither 0 nor 1 (flags != null)
  const int lane = threadIdx.x & 63;
  const int64_t cpr = (C + 63) / 64;  // chunks per row
  const int64_t nchunks = rows * cpr;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t ch = wave0; ch < nchunks; ch += nwaves) {
    const int64_t row = ch / cpr;
    const int32_t c0 = (int32_t)(ch % cpr) * 64;
    const int32_t c = c0 + lane;
    bool v = false;
    if (c < C) {
      const T e = x[row * C + c];
      v = e != (T)0;
      bad |= v && !(e == (T)1);
    }
    const unsigned long long m = __ballot(v);
    const int32_t w0 = c0 >> 5;
    if (lane == 0) bits[row * CW + w0] = (uint32_t)m;
    if (lane == 1 && w0 + 1 < CW) bits[row * CW + w0 + 1] = (uint32_t)(m >> 32);
  }
  if (flags && __ballot(bad) != 0ull && lane == 0) atomicOr(flags, SNNQP_FLAG_GT_ONE);
}

__global__ void __launch_bounds__(256)
unpack_bits_kernel(const uint32_t *__restrict__ bits, int64_t rows, int32_t C,
                   int32_t CW, float *__restrict__ y) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C;
    const int32_t c = (int32_t)(i - row * C);
    y[i] = (float)((bits[row * CW + (c >> 5)] >> (c & 31)) & 1u);
  }
}

__global__ void __launch_bounds__(256)
maxpool_f32_kernel(const float *__restrict__ x, int64_t NB, int32_t H, int32_t W,
                   int32_t C, float *__restrict__ y) {
  const int32_t OH = H / 2, OW = W / 2;
  const int64_t n = NB * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int32_t c = (int32_t)(r % C); r /= C;
    const int32_t ox = (int32_t)(r % OW); r /= OW;
    const int32_t oy = (int32_t)(r % OH); r /= OH;
    const float *p = x + ((r * H + 2 * oy) * W + 2 * ox) * C + c;
    const float a = fmaxf(p[0], p[C]);
    const float b = fmaxf(p[(int64_t)W * C], p[(int64_t)W * C + C]);
    y[i] = fmaxf(a, b);
  }
}

// max over {0,1} is OR: one thread per output word.
__global__ void __launch_bounds__(256)
maxpool_bits_kernel(const uint32_t *__restrict__ x, int64_t NB, int32_t H,
                    int32_t W, int32_t CW, uint32_t *__restrict__ y) {
  const int32_t OH = H / 2, OW = W / 2;
  const int64_t n = NB * OH * OW * CW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i;
    const int32_t c = (int32_t)(r % CW); r /= CW;
    const int32_t ox = (int32_t)(r % OW); r /= OW;
    const int32_t oy = (int32_t)(r % OH); r /= OH;
    const uint32_t *p = x + ((r * H + 2 * oy) * W + 2 * ox) * CW + c;
    y[i] = p[0] | p[CW] | p[(int64_t)W * CW] | p[(int64_t)W * CW + CW];
  }
}

template <bool BITS>
__global__ void __launch_bounds__(256)
vote_kernel(const void *__restrict__ s, int32_t T, int32_t B, int32_t N,
            int32_t group, float *__restrict__ logits, const int32_t *pred) {
  if (pred && *(const volatile int32_t *)pred == 0) return;
  const int32_t NC = N / group;
  const int32_t CW = (N + 31) / 32;
  const int64_t n = (int64_t)B * NC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = (int32_t)(i / NC);
    const int32_t k = (int32_t)(i % NC);
    float acc2 = 0.0f;
    for (int32_t j = 0; j < group; ++j) {
      const int32_t nn = k * group + j;
      float acc = 0.0f;
      for (int32_t t = 0; t < T; ++t) {
        float v;
        if (BITS) {
          const uint32_t w =
              ((const uint32_t *)s)[((int64_t)t * B + b) * CW + (nn >> 5)];
          v = (float)((w >> (nn & 31)) & 1u);
        } else {
          v = ((const float *)s)[((int64_t)t * B + b) * N + nn];
        }
        acc = acc + v;                       // jnp.mean(x, 0): sum ...
      }
      acc2 = acc2 + acc / (float)T;          // ... / T, then sum over group
    }
    logits[i] = acc2 / (float)group;
  }
}

// TCJA pieces (examples/tcja/models.py:41-99) ---------------------------------
// mean over the HW pixels of each (image, channel): one thread per (image, c),
// sequential float32 sum over pixels (exact integer count for spikes) / HW.
template <bool BITS>
__global__ void __launch_bounds__(256)
spatial_mean_kernel(const void *__restrict__ x, int64_t NB, int32_t HW, int32_t C,
                    float *__restrict__ y) {
  const int32_t CW = (C + 31) / 32;
  const int64_t n = NB * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = i / C;
    const int32_t c = (int32_t)(i - img * C);
    float acc = 0.0f;
    if (BITS) {
      // the sequential float32 sum of 0 / 1 values is their exact count (HW < 2^24): counted as an
      // integer, eight independent loads in flight (the dependent float chain took 0.29 ms for a
      // 16 x 16 x 128 raster at B = 1024, T = 20)
      const uint32_t *xw = (const uint32_t *)x + img * HW * CW + (c >> 5);
      uint32_t cnt = 0;
      int32_t p = 0;
      for (; p + 8 <= HW; p += 8) {
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = xw[(int64_t)(p + j) * CW];
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += (w[j] >> (c & 31)) & 1u;
      }
      for (; p < HW; ++p) cnt += (xw[(int64_t)p * CW] >> (c & 31)) & 1u;
      acc = (float)cnt;
    } else {
      for (int32_t p = 0; p < HW; ++p) acc = acc + ((const float *)x)[(img * HW + p) * C + c];
    }
    y[i] = acc / (float)HW;
  }
}

// gate = sigmoid(a * b): float32 product, logistic evaluated in float64 and
// rounded once (the array is tiny: [T, B, C]).
__global__ void __launch_bounds__(256)
sigmoid_gate_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n,
                    float *__restrict__ g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float z = a[i] * b[i];
    g[i] = (float)(1.0 / (1.0 + exp(-(double)z)));
  }
}

// y[img][p][c] = x[img][p][c] * g[img][c]   (x spikes or float32)
template <bool BITS>
__global__ void __launch_bounds__(256)
apply_gate_kernel(const void *__restrict__ x, const float *__restrict__ g, int64_t NB,
                  int32_t HW, int32_t C, float *__restrict__ y) {
  const int32_t CW = (C + 31) / 32;
  const int64_t n = NB * HW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / C;
    const int32_t c = (int32_t)(i - pix * C);
    const int64_t img = pix / HW;
    float v;
    if (BITS)
      v = (float)((((const uint32_t *)x)[pix * CW + (c >> 5)] >> (c & 31)) & 1u);
    else
      v = ((const float *)x)[i];
    y[i] = v * g[img * C + c];
  }
}

// Event front end (examples/input_pipeline.py:142-219, split_by = "number"):
// N time-ordered events -> T frames of N // T events each (the last takes the
// remainder), each a per-pixel, per-polarity count.  counts: int32 [T][H][W][2].
__global__ void __launch_bounds__(256)
events_to_frames_kernel(const int32_t *__restrict__ ex, const int32_t *__restrict__ ey,
                        const int32_t *__restrict__ ep, int64_t N, int32_t T, int32_t H,
                        int32_t W, float scale, int32_t *__restrict__ counts) {
  const int64_t di = N / T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = di > 0 ? i / di : T - 1;
    if (f > T - 1) f = T - 1;
    const int32_t x = (int32_t)floorf((float)ex[i] / scale);
    const int32_t y = (int32_t)floorf((float)ey[i] / scale);
    if (x < 0 || x >= W || y < 0 || y >= H) continue;
    const int32_t c = ep[i] == 0 ? 0 : 1;            // mask[0] = (p == 0), :180-183
    atomicAdd(&counts[((f * H + y) * W + x) * 2 + c], 1);
  }
}

__global__ void __launch_bounds__(256)
i32_to_u8_sat_kernel(const int32_t *__restrict__ x, uint8_t *__restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (uint8_t)min(max(x[i], 0), 255);
}

// Activation density probe (examples/tcja/models.py:128-142): fraction of non-zero
// entries of each of the NB leading slices of n elements.  One wave per slice chunk.
template <int TYPE>
__global__ void __launch_bounds__(256)
density_kernel(const void *__restrict__ x, int64_t NB, int64_t n, int32_t C,
               int32_t *__restrict__ nnz) {
  // BITS: the slice is n / C rows of ceil(C / 32) words (padding bits are zero)
  const int64_t units = TYPE == SNNQP_BITS ? (n / C) * ((C + 31) / 32) : n;
  const int64_t total = NB * units;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = i / units;
    int32_t c;
    if (TYPE == SNNQP_BITS) c = __popc(((const uint32_t *)x)[i]);
    else if (TYPE == SNNQP_U8) c = ((const uint8_t *)x)[i] != 0;
    else c = ((const float *)x)[i] != 0.0f;
    if (c) atomicAdd(&nnz[img], c);
  }
}

static inline int grid_for(int64_t n) {
  const int64_t b = ceil_div64(n, 256);
  return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

}  // namespace snnqp

using namespace snnqp;

extern "C" {

int snnqp_inspect_f32(const float *x, int64_t n, int32_t *flags,
                      snnqp_stream_t stream) {
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "inspect_f32: negative size");
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && flags, SNNQP_EINVAL, "inspect_f32: null argument");
  hipLaunchKernelGGL(inspect_f32_kernel, dim3(grid_for(n)), dim3(256), 0,
                     (hipStream_t)stream, x, n, flags);
  SNNQP_CHECK_LAUNCH("inspect_f32_kernel");
  return SNNQP_OK;
}

int snnqp_inspect_u8(const uint8_t *x, int64_t n, int32_t *flags,
                     snnqp_stream_t stream) {
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "inspect_u8: negative size");
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && flags, SNNQP_EINVAL, "inspect_u8: null argument");
  hipLaunchKernelGGL(inspect_u8_kernel, dim3(grid_for(n / 64 + 1) < 2048 ? grid_for(n / 64 + 1) : 2048),
                     dim3(256), 0,
                     (hipStream_t)stream, x, n, flags);
  SNNQP_CHECK_LAUNCH("inspect_u8_kernel");
  return SNNQP_OK;
}

int snnqp_narrow_f32(const float *x, uint8_t *y, int64_t n, int32_t *flags,
                     snnqp_stream_t stream) {
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "narrow_f32: negative size");
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y && flags, SNNQP_EINVAL, "narrow_f32: null argument");
  const int64_t work = n / 4 + 1;
  const int grid = grid_for(work) < 4096 ? grid_for(work) : 4096;
  hipLaunchKernelGGL(narrow_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n,
                     flags);
  SNNQP_CHECK_LAUNCH("narrow_f32_kernel");
  return SNNQP_OK;
}

int snnqp_f32_to_u8(const float *x, uint8_t *y, int64_t n,
                    snnqp_stream_t stream) {
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "f32_to_u8: negative size");
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y, SNNQP_EINVAL, "f32_to_u8: null argument");
  hipLaunchKernelGGL(f32_to_u8_kernel, dim3(grid_for(n)), dim3(256), 0,
                     (hipStream_t)stream, x, y, n);
  SNNQP_CHECK_LAUNCH("f32_to_u8_kernel");
  return SNNQP_OK;
}

int snnqp_pack_bits(const void *x, int in_type, int64_t rows, int32_t C,
                    uint32_t *bits, snnqp_stream_t stream) {
  return snnqp_pack_bits_checked(x, in_type, rows, C, bits, nullptr, stream);
}

int snnqp_pack_bits_checked(const void *x, int in_type, int64_t rows, int32_t C,
                            uint32_t *bits, int32_t *flags, snnqp_stream_t stream) {
  SNNQP_REQUIRE(rows >= 0 && C > 0, SNNQP_EINVAL, "pack_bits: bad shape");
  SNNQP_REQUIRE(in_type == SNNQP_F32 || in_type == SNNQP_U8, SNNQP_EINVAL,
                "pack_bits: input must be F32 or U8");
  if (rows == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && bits, SNNQP_EINVAL, "pack_bits: null argument");
  const int32_t CW = (C + 31) / 32;
  const int64_t nchunks = rows * ((C + 63) / 64);
  const int grid = grid_for(nchunks * 64);
  if (in_type == SNNQP_F32)
    hipLaunchKernelGGL(pack_bits_kernel<float>, dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, (const float *)x, rows, C, CW, bits, flags);
  else
    hipLaunchKernelGGL(pack_bits_kernel<uint8_t>, dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, (const uint8_t *)x, rows, C, CW, bits, flags);
  SNNQP_CHECK_LAUNCH("pack_bits_kernel");
  return SNNQP_OK;
}

int snnqp_unpack_bits(const uint32_t *bits, int64_t rows, int32_t C, float *y,
                      snnqp_stream_t stream) {
  SNNQP_REQUIRE(rows >= 0 && C > 0, SNNQP_EINVAL, "unpack_bits: bad shape");
  if (rows == 0) return SNNQP_OK;
  SNNQP_REQUIRE(bits && y, SNNQP_EINVAL, "unpack_bits: null argument");
  hipLaunchKernelGGL(unpack_bits_kernel, dim3(grid_for(rows * C)), dim3(256), 0,
                     (hipStream_t)stream, bits, rows, C, (C + 31) / 32, y);
  SNNQP_CHECK_LAUNCH("unpack_bits_kernel");
  return SNNQP_OK;
}

int snnqp_maxpool2x2(const void *x, int type, int64_t NB, int32_t H, int32_t W,
                     int32_t C, void *y, snnqp_stream_t stream) {
  SNNQP_REQUIRE(NB >= 0 && H >= 0 && W >= 0 && C > 0 && ((x && y) || NB * (H / 2) * (W / 2) == 0), SNNQP_EINVAL,
                "maxpool2x2: bad argument");
  SNNQP_REQUIRE(type == SNNQP_F32 || type == SNNQP_BITS, SNNQP_EINVAL,
                "maxpool2x2: type must be F32 or BITS");
  const int64_t opix = NB * (H / 2) * (W / 2);
  if (opix == 0) return SNNQP_OK;
  if (type == SNNQP_F32) {
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3(grid_for(opix * C)), dim3(256),
                       0, (hipStream_t)stream, (const float *)x, NB, H, W, C,
                       (float *)y);
  } else {
    const int32_t CW = (C + 31) / 32;
    hipLaunchKernelGGL(maxpool_bits_kernel, dim3(grid_for(opix * CW)),
                       dim3(256), 0, (hipStream_t)stream, (const uint32_t *)x,
                       NB, H, W, CW, (uint32_t *)y);
  }
  SNNQP_CHECK_LAUNCH("maxpool kernel");
  return SNNQP_OK;
}

int snnqp_spatial_mean(const void *x, int type, int64_t NB, int32_t HW, int32_t C,
                       float *y, snnqp_stream_t stream) {
  SNNQP_REQUIRE(NB >= 0 && HW > 0 && C > 0, SNNQP_EINVAL, "spatial_mean: bad shape");
  SNNQP_REQUIRE(type == SNNQP_F32 || type == SNNQP_BITS, SNNQP_EINVAL,
                "spatial_mean: type must be F32 or BITS");
  if (NB == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y, SNNQP_EINVAL, "spatial_mean: null argument");
  if (type == SNNQP_BITS)
    hipLaunchKernelGGL(spatial_mean_kernel<true>, dim3(grid_for(NB * C)), dim3(256), 0,
                       (hipStream_t)stream, x, NB, HW, C, y);
  else
    hipLaunchKernelGGL(spatial_mean_kernel<false>, dim3(grid_for(NB * C)), dim3(256), 0,
                       (hipStream_t)stream, x, NB, HW, C, y);
  SNNQP_CHECK_LAUNCH("spatial_mean_kernel");
  return SNNQP_OK;
}

int snnqp_sigmoid_gate(const float *a, const float *b, int64_t n, float *g,
                       snnqp_stream_t stream) {
  SNNQP_REQUIRE(n >= 0, SNNQP_EINVAL, "sigmoid_gate: negative size");
  if (n == 0) return SNNQP_OK;
  SNNQP_REQUIRE(a && b && g, SNNQP_EINVAL, "sigmoid_gate: null argument");
  hipLaunchKernelGGL(sigmoid_gate_kernel, dim3(grid_for(n)), dim3(256), 0,
                     (hipStream_t)stream, a, b, n, g);
  SNNQP_CHECK_LAUNCH("sigmoid_gate_kernel");
  return SNNQP_OK;
}

int snnqp_apply_gate(const void *x, int type, const float *g, int64_t NB, int32_t HW,
                     int32_t C, float *y, snnqp_stream_t stream) {
  SNNQP_REQUIRE(NB >= 0 && HW > 0 && C > 0, SNNQP_EINVAL, "apply_gate: bad shape");
  SNNQP_REQUIRE(type == SNNQP_F32 || type == SNNQP_BITS, SNNQP_EINVAL,
                "apply_gate: type must be F32 or BITS");
  if (NB == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && g && y, SNNQP_EINVAL, "apply_gate: null argument");
  const int64_t n = NB * HW * C;
  if (type == SNNQP_BITS)
    hipLaunchKernelGGL(apply_gate_kernel<true>, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, x, g, NB, HW, C, y);
  else
    hipLaunchKernelGGL(apply_gate_kernel<false>, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, x, g, NB, HW, C, y);
  SNNQP_CHECK_LAUNCH("apply_gate_kernel");
  return SNNQP_OK;
}

int snnqp_events_to_frames(const int32_t *ex, const int32_t *ey, const int32_t *ep,
                           int64_t N, int32_t T, int32_t H, int32_t W, float scale,
                           int32_t *counts, uint8_t *frames_u8, snnqp_stream_t stream) {
  SNNQP_REQUIRE(N >= 0 && T > 0 && H > 0 && W > 0 && scale > 0.0f, SNNQP_EINVAL,
                "events_to_frames: bad shape");
  SNNQP_REQUIRE(counts, SNNQP_EINVAL, "events_to_frames: null counts buffer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)T * H * W * 2;
  if (int rc = zero_words_async((uint32_t *)counts, n, st)) return rc;
  if (N > 0) {
    SNNQP_REQUIRE(ex && ey && ep, SNNQP_EINVAL, "events_to_frames: null events");
    hipLaunchKernelGGL(events_to_frames_kernel, dim3(grid_for(N)), dim3(256), 0, st, ex, ey,
                       ep, N, T, H, W, scale, counts);
    SNNQP_CHECK_LAUNCH("events_to_frames_kernel");
  }
  if (frames_u8) {
    hipLaunchKernelGGL(i32_to_u8_sat_kernel, dim3(grid_for(n)), dim3(256), 0, st, counts,
                       frames_u8, n);
    SNNQP_CHECK_LAUNCH("i32_to_u8_sat_kernel");
  }
  return SNNQP_OK;
}

int snnqp_density(const void *x, int type, int64_t NB, int64_t n, int32_t C, int32_t *nnz,
                  snnqp_stream_t stream) {
  SNNQP_REQUIRE(NB >= 0 && n > 0 && C > 0 && n % C == 0, SNNQP_EINVAL, "density: bad shape");
  SNNQP_REQUIRE(type == SNNQP_F32 || type == SNNQP_BITS || type == SNNQP_U8, SNNQP_EINVAL,
                "density: type must be F32, U8 or BITS");
  if (NB == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && nnz, SNNQP_EINVAL, "density: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = zero_words_async((uint32_t *)nnz, NB, st)) return rc;
  const int64_t units = type == SNNQP_BITS ? (n / C) * ((C + 31) / 32) : n;
  if (type == SNNQP_BITS)
    hipLaunchKernelGGL(density_kernel<SNNQP_BITS>, dim3(grid_for(NB * units)), dim3(256), 0, st, x,
                       NB, n, C, nnz);
  else if (type == SNNQP_U8)
    hipLaunchKernelGGL(density_kernel<SNNQP_U8>, dim3(grid_for(NB * units)), dim3(256), 0, st, x,
                       NB, n, C, nnz);
  else
    hipLaunchKernelGGL(density_kernel<SNNQP_F32>, dim3(grid_for(NB * units)), dim3(256), 0, st, x,
                       NB, n, C, nnz);
  SNNQP_CHECK_LAUNCH("density_kernel");
  return SNNQP_OK;
}

int snnqp_vote(const void *s, int type, int32_t T, int32_t B, int32_t N,
               int32_t group, float *logits, snnqp_stream_t stream) {
  return snnqp_vote_if(nullptr, s, type, T, B, N, group, logits, stream);
}

int snnqp_vote_if(const int32_t *pred, const void *s, int type, int32_t T, int32_t B, int32_t N,
                  int32_t group, float *logits, snnqp_stream_t stream) {
  SNNQP_REQUIRE(((s && logits) || B == 0) && T > 0 && B >= 0 && N > 0 && group > 0,
                SNNQP_EINVAL, "vote: bad argument");
  SNNQP_REQUIRE(N % group == 0, SNNQP_EINVAL,
                "vote: N=%d not divisible by group=%d", N, group);
  SNNQP_REQUIRE(type == SNNQP_F32 || type == SNNQP_BITS, SNNQP_EINVAL,
                "vote: type must be F32 or BITS");
  if (B == 0) return SNNQP_OK;
  const int64_t n = (int64_t)B * (N / group);
  if (type == SNNQP_BITS)
    hipLaunchKernelGGL(vote_kernel<true>, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, s, T, B, N, group, logits, pred);
  else
    hipLaunchKernelGGL(vote_kernel<false>, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, s, T, B, N, group, logits, pred);
  SNNQP_CHECK_LAUNCH("vote_kernel");
  return SNNQP_OK;
}

}  // extern "C"
