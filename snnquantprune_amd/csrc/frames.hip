// Wire formats of the model input (include/snnqp.h SNNQP_EV1 / SNNQP_EV4): the 2-channel
// event frames the reference's pipeline produces (examples/input_pipeline.py:195-218) and
// prefetches to the device (:17-27), bit-packed (binary frames) or nibble-packed (counts
// <= 15) so that the host -> device feed moves 1/8 or 1/2 of the uint8 bytes.  One pass
// each, HBM-bound; conv3x3_u8c2.hip stages EV1 frames directly, every other consumer
// unpacks first.
#include "common.h"

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
