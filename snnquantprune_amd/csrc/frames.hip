// Wire formats of the model input (include/snnqp.h SNNQP_EV1 / SNNQP_EV4): the 2-channel
// event frames the reference's pipeline produces (examples/input_pipeline.py:195-218) and
// prefetches to the device (:17-27), bit-packed (binary frames) or nibble-packed (counts
// <= 15) so that the host -> device feed moves 1/8 or 1/2 of the uint8 bytes.  One pass
// each, HBM-bound; conv3x3_u8c2.hip stages EV1 frames directly, every other consumer
// unpacks first.
#include "kernels.h"

namespace snnqp {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// 32 bytes (two 16-byte words) -> 32 bits, bit j = (byte j != 0); `gt` collects bytes > 1
__device__ __forceinline__ uint32_t bits_of_16(const v4u &w, uint32_t &gt) {
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t x = w[j];
    gt |= x & 0xFEFEFEFEu;
    // non-zero test per byte without carries between bytes: (x | (x + 0x7F..)) bit 7
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    // gather bits 7, 15, 23, 31 into a nibble
    out |= (((nz >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * j);
  }
  return out;
}

__global__ void __launch_bounds__(256)
pack_ev1_kernel(const uint8_t *__restrict__ x, int64_t frames, int64_t fbytes, int64_t fwords,
                uint32_t *__restrict__ y, int32_t *__restrict__ flags) {
  const int64_t n = frames * fwords;
  uint32_t gt = 0;
  const bool vec = ((uintptr_t)x & 15) == 0 && (fbytes & 31) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / fwords, wi = i - f * fwords;
    const uint8_t *src = x + f * fbytes + wi * 32;
    uint32_t word;
    if (vec) {
      const v4u a = ((const v4u *)src)[0], b = ((const v4u *)src)[1];
      word = bits_of_16(a, gt) | (bits_of_16(b, gt) << 16);
    } else {
      word = 0;
      const int64_t left = fbytes - wi * 32;
      for (int j = 0; j < 32 && j < left; ++j) {
        const uint32_t v = src[j];
        gt |= v & 0xFEu;
        word |= (v != 0 ? 1u : 0u) << j;
      }
    }
    y[i] = word;
  }
  if (flags && gt) atomicOr(flags, SNNQP_FLAG_GT_ONE);
}

// float32 frames (the reference's own input dtype, flax_qconv.py:101) -> EV1: 32 values = 128 bytes
// per word.  A value that is neither 0.0 (-0.0 counts) nor 1.0 -- a count, a fraction, a NaN --
// raises SNNQP_FLAG_GT_ONE (and SNNQP_FLAG_NOT_INTEGER when it is not an integer in [0, 255]): the
// packed frames then are not the tensor.
__global__ void __launch_bounds__(256)
pack_ev1_f32_kernel(const float *__restrict__ x, int64_t frames, int64_t fvals, int64_t fwords,
                    uint32_t *__restrict__ y, int32_t *__restrict__ flags) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int64_t n = frames * fwords;
  uint32_t notbin = 0, notint = 0;
  const bool vec = ((uintptr_t)x & 15) == 0 && (fvals & 31) == 0;
  auto one = [&](float v) -> uint32_t {
    const bool is1 = v == 1.0f, is0 = v == 0.0f;
    notbin |= (is1 || is0) ? 0u : 1u;
    notint |= (v >= 0.0f && v <= 255.0f && v == __builtin_rintf(v)) ? 0u : 1u;
    return is1 ? 1u : 0u;
  };
  if (vec) {
    // frames of whole words lie back to back: a lane takes 16 bytes (four values, a nibble of the
    // word), a wave 1 KiB in one coalesced request; the eight lanes of a word OR their nibbles
    // together (units = 8 x words and the grid stride is a multiple of 64: a word's lanes are
    // always in the loop together)
    const int64_t units = n * 8;
    const int lane8 = threadIdx.x & 7;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units;
         u += (int64_t)gridDim.x * blockDim.x) {
      const v4f v = ((const v4f *)x)[u];
      uint32_t w = (one(v.x) | (one(v.y) << 1) | (one(v.z) << 2) | (one(v.w) << 3)) << (4 * lane8);
      w |= (uint32_t)__shfl_xor((int)w, 1);
      w |= (uint32_t)__shfl_xor((int)w, 2);
      w |= (uint32_t)__shfl_xor((int)w, 4);
      if (lane8 == 0) y[u >> 3] = w;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t f = i / fwords, wi = i - f * fwords;
      const float *src = x + f * fvals + wi * 32;
      uint32_t word = 0;
      const int64_t left = fvals - wi * 32;
      for (int j = 0; j < 32 && j < left; ++j) word |= one(src[j]) << j;
      y[i] = word;
    }
  }
  if (flags && (notbin | notint))
    atomicOr(flags, (notbin ? SNNQP_FLAG_GT_ONE : 0) | (notint ? SNNQP_FLAG_NOT_INTEGER : 0));
}

__global__ void __launch_bounds__(256)
pack_ev4_kernel(const uint8_t *__restrict__ x, int64_t npix, uint8_t *__restrict__ y,
                int32_t *__restrict__ flags) {
  uint32_t gt = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t p0 = x[2 * i], p1 = x[2 * i + 1];
    gt |= (p0 | p1) & 0xF0u;
    y[i] = (uint8_t)((p0 > 15 ? 15u : p0) | ((p1 > 15 ? 15u : p1) << 4));
  }
  if (flags && gt) atomicOr(flags, SNNQP_FLAG_GT_15);
}

__global__ void __launch_bounds__(256)
unpack_ev1_kernel(const uint32_t *__restrict__ x, int64_t frames, int64_t fbytes, int64_t fwords,
                  uint8_t *__restrict__ y) {
  const int64_t n = frames * fwords;
  const bool vec = ((uintptr_t)y & 15) == 0 && (fbytes & 31) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / fwords, wi = i - f * fwords;
    const uint32_t w = x[i];
    uint8_t *dst = y + f * fbytes + wi * 32;
    if (vec) {
      v4u o[2];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j >> 2][j & 3] = (((w >> (4 * j)) & 0xFu) * 0x00204081u) & 0x01010101u;
      ((v4u *)dst)[0] = o[0];
      ((v4u *)dst)[1] = o[1];
    } else {
      const int64_t left = fbytes - wi * 32;
      for (int j = 0; j < 32 && j < left; ++j) dst[j] = (uint8_t)((w >> j) & 1u);
    }
  }
}

__global__ void __launch_bounds__(256)
unpack_ev4_kernel(const uint8_t *__restrict__ x, int64_t npix, uint8_t *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = x[i];
    y[2 * i] = (uint8_t)(v & 0xFu);
    y[2 * i + 1] = (uint8_t)(v >> 4);
  }
}

static unsigned grid_of(int64_t n) {
  const int64_t b = ceil_div64(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace snnqp

using namespace snnqp;

extern "C" {

int snnqp_pack_frames(const uint8_t *x, int64_t frames, int32_t H, int32_t W, int fmt,
                      void *y, int32_t *flags, snnqp_stream_t stream) {
  SNNQP_REQUIRE(frames >= 0 && H > 0 && W > 0, SNNQP_EINVAL, "pack_frames: bad shape");
  SNNQP_REQUIRE(fmt == SNNQP_EV1 || fmt == SNNQP_EV4, SNNQP_EINVAL,
                "pack_frames: format must be SNNQP_EV1 or SNNQP_EV4");
  if (frames == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y, SNNQP_EINVAL, "pack_frames: null argument");
  const int64_t fbytes = (int64_t)H * W * 2;
  if (fmt == SNNQP_EV1) {
    const int64_t fwords = (fbytes + 31) / 32;
    hipLaunchKernelGGL(pack_ev1_kernel, dim3(grid_of(frames * fwords)), dim3(256), 0,
                       (hipStream_t)stream, x, frames, fbytes, fwords, (uint32_t *)y, flags);
    SNNQP_CHECK_LAUNCH("pack_ev1_kernel");
  } else {
    hipLaunchKernelGGL(pack_ev4_kernel, dim3(grid_of(frames * H * W)), dim3(256), 0,
                       (hipStream_t)stream, x, frames * (int64_t)H * W, (uint8_t *)y, flags);
    SNNQP_CHECK_LAUNCH("pack_ev4_kernel");
  }
  return SNNQP_OK;
}

int snnqp_pack_frames_checked(const void *x, int in_type, int64_t frames, int32_t H, int32_t W,
                              uint32_t *y, int32_t *flags, snnqp_stream_t stream) {
  SNNQP_REQUIRE(frames >= 0 && H > 0 && W > 0, SNNQP_EINVAL, "pack_frames_checked: bad shape");
  SNNQP_REQUIRE(in_type == SNNQP_U8 || in_type == SNNQP_F32, SNNQP_EINVAL,
                "pack_frames_checked: frames must be uint8 or float32");
  SNNQP_REQUIRE(flags, SNNQP_EINVAL, "pack_frames_checked: null flag word");
  // the flag word is zeroed on the stream in front of every pass (a kernel node under capture)
  if (int rc = zero_words_async((uint32_t *)flags, 1, (hipStream_t)stream)) return rc;
  if (frames == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y, SNNQP_EINVAL, "pack_frames_checked: null argument");
  const int64_t fvals = (int64_t)H * W * 2, fwords = (fvals + 31) / 32;
  if (in_type == SNNQP_U8)
    hipLaunchKernelGGL(pack_ev1_kernel, dim3(grid_of(frames * fwords)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)x, frames, fvals, fwords, y, flags);
  else
    hipLaunchKernelGGL(pack_ev1_f32_kernel, dim3(grid_of(frames * fwords * 8)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)x, frames, fvals, fwords, y, flags);
  SNNQP_CHECK_LAUNCH("pack_ev1 (checked)");
  return SNNQP_OK;
}

int snnqp_unpack_frames(const void *x, int fmt, int64_t frames, int32_t H, int32_t W,
                        uint8_t *y, snnqp_stream_t stream) {
  SNNQP_REQUIRE(frames >= 0 && H > 0 && W > 0, SNNQP_EINVAL, "unpack_frames: bad shape");
  SNNQP_REQUIRE(fmt == SNNQP_EV1 || fmt == SNNQP_EV4, SNNQP_EINVAL,
                "unpack_frames: format must be SNNQP_EV1 or SNNQP_EV4");
  if (frames == 0) return SNNQP_OK;
  SNNQP_REQUIRE(x && y, SNNQP_EINVAL, "unpack_frames: null argument");
  const int64_t fbytes = (int64_t)H * W * 2;
  if (fmt == SNNQP_EV1) {
    const int64_t fwords = (fbytes + 31) / 32;
    hipLaunchKernelGGL(unpack_ev1_kernel, dim3(grid_of(frames * fwords)), dim3(256), 0,
                       (hipStream_t)stream, (const uint32_t *)x, frames, fbytes, fwords, y);
    SNNQP_CHECK_LAUNCH("unpack_ev1_kernel");
  } else {
    hipLaunchKernelGGL(unpack_ev4_kernel, dim3(grid_of(frames * H * W)), dim3(256), 0,
                       (hipStream_t)stream, (const uint8_t *)x, frames * (int64_t)H * W, y);
    SNNQP_CHECK_LAUNCH("unpack_ev4_kernel");
  }
  return SNNQP_OK;
}

}  // extern "C"
