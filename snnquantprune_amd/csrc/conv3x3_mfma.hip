// Fused SpikingBlock for the 3x3 / stride 1 / pad 1 QuantConv layers of the
// DVS128 topology (examples/tcja/models.py:111-147): implicit-GEMM int8 MFMA
// (v_mfma_i32_32x32x32_i8) + dequantisation + eval BatchNorm + neuron update +
// optional 2x2 max-pool, with the T loop inside the kernel.
//
// Mapping (one 256-thread workgroup = 4 waves, persistent over patches):
//  * a patch is 8x8 output pixels of one sample = two 32-row MFMA tiles (4x8
//    pixels each); wave w owns output channels [32w, 32w+32) of a 128-channel
//    block (blockIdx.y);
//  * the wave's weights -- all 9 taps x Cin for its 32 channels -- live in
//    VGPRs for the whole launch (the B operand; 144 registers at Cin = 128);
//  * per timestep the 10x10 halo of input spikes is expanded from bits to
//    {0,1} bytes into LDS once (XOR-swizzled 16-byte chunks) and every tap's A
//    fragment is one ds_read_b128 at a shifted pixel;
//  * C/D layout: lane = output channel, register = pixel, so the per-channel
//    dequant/BatchNorm constants are per-lane registers, the membrane potential
//    of the patch stays in 32 VGPRs for all T, and the v_cmp that thresholds a
//    register *is* the packed spike word of two pixels (64-bit lane mask);
//    pooling is an OR of those scalar masks.
#include "kernels.h"

namespace snnqp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int HALO = 10;
constexpr int PIXB = 128;  // LDS bytes per halo pixel (bit-input variant)

struct ConvMfmaArgs {
  const void *x;
  int64_t xs_t, xs_b;
  int32_t T, B, H, W, Cin, Cout;
  const int8_t *w;   // HWIO int8 codes
  const int8_t *wt;  // the same codes, MFMA-tiled (snnqp_pack_codes_mfma)
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  uint32_t *s_out;
  int32_t pool;
  int32_t tiles_y, tiles_x;
  int64_t npatch;
};

// 16-byte chunk c16 of halo pixel (hy, hx).  Two pixels share a 256-byte bank
// row; the XOR makes the 16 lanes of every ds_read_b128 lane group (4 pixel
// rows x 4 consecutive pixel columns, one chunk index) hit 16 distinct slots.
__device__ __forceinline__ int halo_addr(int hy, int hx, int c16) {
  const int g = ((hy & 3) << 1) | ((hx >> 1) & 1);
  return (hy * HALO + hx) * PIXB + ((c16 ^ g) << 4);
}

__device__ __forceinline__ v4i expand16(uint32_t b) {
  v4i o;
  o.x = (int)((((b >> 0) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.y = (int)((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.z = (int)((((b >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.w = (int)((((b >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

// Dequant + BN + neuron for one 32x32 tile.  m[i] = lane mask of register i:
// low 32 bits = pixel (ty = 2*(i>>3), tx = i&7), high = (ty + 1, tx).
template <bool FAST>
__device__ __forceinline__ void tile_epilogue(const v16i &acc, float (&u)[16],
                                              const Dequant &dq, float bmean,
                                              float bmul, float bbias,
                                              const NeuronP &nrn, float dec,
                                              unsigned long long (&m)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float cur = dequant_acc(acc[i], dq);
    cur = bn_apply(cur, bmean, bmul, bbias);
    bool s;
    if (FAST) {  // multi_step_LIF, tau a power of two (spiking_learning.py:410-414)
      const float d = cur - (u[i] - nrn.vr);
      u[i] = u[i] + d * nrn.inv_k;
      s = (u[i] - nrn.vth) >= 0.0f;
      u[i] = s ? nrn.vr : u[i];
    } else {
      s = neuron_step(u[i], cur, nrn, dec);
    }
    m[i] = __ballot(s);
  }
}

// Writes the spike words of one tile.  cw = word index of this wave's channels.
__device__ __forceinline__ void tile_store(const unsigned long long (&m)[16],
                                           const ConvMfmaArgs &a, int t, int b,
                                           int y0, int x0, int tl, int cw,
                                           int lane) {
  const int CW = a.Cout >> 5;
  if (a.pool == 2) {
    uint32_t myw = 0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const unsigned long long o = m[i] | m[i + 1];
      const uint32_t pw = (uint32_t)o | (uint32_t)(o >> 32);
      if (lane == (i >> 1)) myw = pw;
    }
    if (lane < 8) {
      const int OH = a.H >> 1, OW = a.W >> 1;
      const int oy = (y0 >> 1) + tl * 2 + (lane >> 2);
      const int ox = (x0 >> 1) + (lane & 3);
      a.s_out[((((int64_t)t * a.B + b) * OH + oy) * OW + ox) * CW + cw] = myw;
    }
  } else {
    uint32_t myw = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rlo = (i & 3) + 8 * (i >> 2);
      if (lane == rlo) myw = (uint32_t)m[i];
      if (lane == rlo + 4) myw = (uint32_t)(m[i] >> 32);
    }
    if (lane < 32) {
      const int ty = ((lane >> 2) & 1) | ((lane >> 4) << 1);
      const int tx = (lane & 3) | (((lane >> 3) & 1) << 2);
      const int oy = y0 + tl * 4 + ty, ox = x0 + tx;
      a.s_out[((((int64_t)t * a.B + b) * a.H + oy) * a.W + ox) * CW + cw] = myw;
    }
  }
}

template <bool LOAD>
__device__ __forceinline__ void u_io(float (&u)[2][16], const ConvMfmaArgs &a,
                                     int b, int y0, int x0, int cout, int h) {
  float *uo = a.u_out;
  const float *ui = a.u0;
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int y = y0 + tl * 4 + (h | ((i >> 3) << 1));
      const int x = x0 + (i & 7);
      const int64_t o = (((int64_t)b * a.H + y) * a.W + x) * a.Cout + cout;
      if (LOAD) u[tl][i] = ui[o];
      else uo[o] = u[tl][i];
    }
}

// ---------------------------------------------------------------------------
// Bit-packed input, Cin = CIN (multiple of 32, <= 128).
// ---------------------------------------------------------------------------
template <int CIN, bool FAST>
__global__ void __launch_bounds__(256, 1)
conv3x3_bits_kernel(ConvMfmaArgs a) {
  constexpr int KK = CIN / 32;
  constexpr int NTASK = HALO * HALO * KK;        // (pixel, word) staging tasks
  constexpr int TPT = (NTASK + 255) / 256;       // tasks per thread
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * HALO * HALO * PIXB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int cout_base = blockIdx.y * 128 + wave * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cw = cout_base >> 5;

  // B operand: lane (n, h) holds W[tap][cin = 32 kk + 16 h + j][cout], j < 16:
  // k-step tap * KK + kk of this wave's 32-column block in the MFMA-tiled codes.
  v4i bf[9][KK];
  {
    const v4i *wtile = (const v4i *)a.wt + ((int64_t)(cout_base >> 5) * (9 * KK)) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        bf[tap][kk] = wave_on ? wtile[(tap * KK + kk) * 64] : v4i{0, 0, 0, 0};
  }

  float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f;
  if (a.bn.mean) { bmean = a.bn.mean[cout]; bmul = a.bn.mul[cout]; bbias = a.bn.bias[cout]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[cout];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  const uint32_t *xb = (const uint32_t *)a.x;

  for (int64_t p = blockIdx.x; p < a.npatch; p += gridDim.x) {
    int64_t q = p;
    const int px = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int py = (int)(q % a.tiles_y); q /= a.tiles_y;
    const int b = (int)q;
    const int y0 = py * 8, x0 = px * 8;

    float u[2][16];
    if (a.u0 && wave_on) {
      u_io<true>(u, a, b, y0, x0, cout, h);
    } else {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int i = 0; i < 16; ++i) u[tl][i] = 0.0f;
    }

    uint32_t stg[TPT];
    auto stage_load = [&](int t) {
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        const int task = tid + k * 256;
        uint32_t wv = 0;
        if (task < NTASK) {
          const int pix = task / KK, wi = task % KK;
          const int gy = y0 + pix / HALO - 1, gx = x0 + pix % HALO - 1;
          if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            wv = xb[(int64_t)t * a.xs_t + (int64_t)b * a.xs_b +
                    ((int64_t)gy * a.W + gx) * KK + wi];
        }
        stg[k] = wv;
      }
    };
    auto stage_store = [&](int buf) {
      uint8_t *base = lds + buf * (HALO * HALO * PIXB);
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        const int task = tid + k * 256;
        if (task < NTASK) {
          const int pix = task / KK, wi = task % KK;
          const int hy = pix / HALO, hx = pix % HALO;
          *(v4i *)(base + halo_addr(hy, hx, wi * 2)) = expand16(stg[k] & 0xFFFFu);
          *(v4i *)(base + halo_addr(hy, hx, wi * 2 + 1)) = expand16(stg[k] >> 16);
        }
      }
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
      if (t + 1 < a.T) stage_load(t + 1);
      if (wave_on) {
        const uint8_t *base = lds + (t & 1) * (HALO * HALO * PIXB);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int kk = 0; kk < KK; ++kk) {
                const v4i av = *(const v4i *)(base + halo_addr(tl * 4 + ty + dy,
                                                               tx + dx, kk * 2 + h));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bf[dy * 3 + dx][kk],
                                                            acc, 0, 0, 0);
              }
          unsigned long long m[16];
          tile_epilogue<FAST>(acc, u[tl], a.dq, bmean, bmul, bbias, a.nrn, dec, m);
          tile_store(m, a, t, b, y0, x0, tl, cw, lane);
        }
      }
      if (t + 1 < a.T) stage_store((t + 1) & 1);
      __syncthreads();
    }
    if (a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
  }
}

// ---------------------------------------------------------------------------
// u8 event-count input with Cin = 2 (the DVS polarity pair, conv0): K = 18
// padded to one 32-deep MFMA step.  k = 2 * tap + cin.
// ---------------------------------------------------------------------------
constexpr int HROW2 = 24;  // LDS bytes per halo row (10 pixels x 2 B, padded)

template <bool FAST>
__global__ void __launch_bounds__(256, 2)
conv3x3_u8c2_kernel(ConvMfmaArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * HALO * HROW2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int cout_base = blockIdx.y * 128 + wave * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cw = cout_base >> 5;

  v4i bf;
  {
    int v[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t pk = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * h + 4 * d + j;
        uint32_t bv = 0;
        if (k < 18) bv = (uint8_t)a.w[(int64_t)k * a.Cout + cout];  // HWIO, Cin = 2
        pk |= bv << (8 * j);
      }
      v[d] = (int)pk;
    }
    bf = v4i{v[0], v[1], v[2], v[3]};
  }

  float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f;
  if (a.bn.mean) { bmean = a.bn.mean[cout]; bmul = a.bn.mul[cout]; bbias = a.bn.bias[cout]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[cout];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  const uint8_t *xb = (const uint8_t *)a.x;

  for (int64_t p = blockIdx.x; p < a.npatch; p += gridDim.x) {
    int64_t q = p;
    const int px = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int py = (int)(q % a.tiles_y); q /= a.tiles_y;
    const int b = (int)q;
    const int y0 = py * 8, x0 = px * 8;

    float u[2][16];
    if (a.u0 && wave_on) {
      u_io<true>(u, a, b, y0, x0, cout, h);
    } else {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int i = 0; i < 16; ++i) u[tl][i] = 0.0f;
    }

    uint16_t stg = 0;
    auto stage_load = [&](int t) {
      stg = 0;
      if (tid < HALO * HALO) {
        const int gy = y0 + tid / HALO - 1, gx = x0 + tid % HALO - 1;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
          stg = *(const uint16_t *)(xb + (int64_t)t * a.xs_t + (int64_t)b * a.xs_b +
                                    ((int64_t)gy * a.W + gx) * 2);
      }
    };
    auto stage_store = [&](int buf) {
      if (tid < HALO * HALO)
        *(uint16_t *)(lds + buf * (HALO * HROW2) + (tid / HALO) * HROW2 +
                      (tid % HALO) * 2) = stg;
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
      if (t + 1 < a.T) stage_load(t + 1);
      if (wave_on) {
        const uint8_t *base = lds + (t & 1) * (HALO * HROW2);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          // A fragment: dword d = taps (8h + 2d, 8h + 2d + 1), each one u16.
          int av[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            uint32_t lo = 0, hi = 0;
            const int t0 = 8 * h + 2 * d, t1 = t0 + 1;
            if (t0 < 9)
              lo = *(const uint16_t *)(base + (tl * 4 + ty + t0 / 3) * HROW2 +
                                       (tx + t0 % 3) * 2);
            if (t1 < 9)
              hi = *(const uint16_t *)(base + (tl * 4 + ty + t1 / 3) * HROW2 +
                                       (tx + t1 % 3) * 2);
            av[d] = (int)(lo | (hi << 16));
          }
          v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{av[0], av[1], av[2], av[3]},
                                                      bf, acc, 0, 0, 0);
          unsigned long long m[16];
          tile_epilogue<FAST>(acc, u[tl], a.dq, bmean, bmul, bbias, a.nrn, dec, m);
          tile_store(m, a, t, b, y0, x0, tl, cw, lane);
        }
      }
      if (t + 1 < a.T) stage_store((t + 1) & 1);
      __syncthreads();
    }
    if (a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

const char *conv3x3_mfma_unsupported(int in_type, const snnqp_conv_geom_t *g,
                                     const snnqp_weight_t *w, const int8_t *wt,
                                     const snnqp_neuron_t *nrn, int s_type) {
  if (w->wtype != SNNQP_W_I8) return "weights are not int8 codes";
  if (in_type == SNNQP_BITS && !wt) return "MFMA-tiled codes `wt` not given";
  if (g->KH != 3 || g->KW != 3) return "kernel is not 3x3";
  if (g->stride_h != 1 || g->stride_w != 1) return "stride is not 1";
  if (g->pad_h_lo != 1 || g->pad_h_hi != 1 || g->pad_w_lo != 1 || g->pad_w_hi != 1)
    return "padding is not ((1,1),(1,1))";
  if (g->in_dil_h != 1 || g->in_dil_w != 1 || g->k_dil_h != 1 || g->k_dil_w != 1)
    return "dilated convolution";
  if (g->groups != 1) return "grouped convolution";
  if (g->H <= 0 || g->W <= 0 || (g->H & 7) || (g->W & 7))
    return "H and W must be positive multiples of 8";
  if (g->Cout & 31) return "Cout must be a multiple of 32";
  if (s_type != SNNQP_BITS) return "spike output must be bit-packed";
  if (in_type == SNNQP_BITS) {
    if (g->Cin != 128) return "bit input needs Cin == 128";
  } else if (in_type == SNNQP_U8) {
    if (g->Cin != 2) return "u8 input needs Cin == 2";
  } else {
    return "input must be BITS or U8";
  }
  if (nrn->kind == SNNQP_NEURON_LIF && !nrn->decay) return "LIF without decay";
  return nullptr;
}

template <typename K>
static int persistent_grid(K kernel, int64_t npatch) {
  int dev = 0, cus = 256, occ = 2;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess ||
      occ < 1)
    occ = 2;
  if (occ > 4) occ = 4;
  const int64_t gmax = (int64_t)cus * occ;
  return (int)(npatch < gmax ? npatch : gmax);
}

int run_conv3x3_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b,
                     int32_t T, int32_t B, const snnqp_conv_geom_t *g,
                     const snnqp_weight_t *w, const int8_t *wt,
                     const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                     const float *u0, float *u_out, uint32_t *s_out, int pool,
                     hipStream_t st) {
  SNNQP_REQUIRE(x && w->w && s_out, SNNQP_EINVAL, "conv3x3 mfma: null pointer");
  SNNQP_REQUIRE(in_type != SNNQP_BITS || wt, SNNQP_EINVAL,
                "conv3x3 mfma: bit input needs the MFMA-tiled codes `wt`");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "conv3x3 mfma: negative T/B");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  if (bn) SNNQP_REQUIRE(bn->mean && bn->mul && bn->bias, SNNQP_EINVAL,
                        "batch-norm descriptor with null arrays");
  if (T == 0 || B == 0) return SNNQP_OK;
  ConvMfmaArgs a;
  a.x = x; a.xs_t = xs_t; a.xs_b = xs_b; a.T = T; a.B = B;
  a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.Cout = g->Cout;
  a.w = (const int8_t *)w->w;
  a.wt = wt;
  a.dq = make_dequant(w->L, w->m);
  a.bn = make_bn(bn);
  a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out; a.pool = pool;
  a.tiles_y = g->H / 8; a.tiles_x = g->W / 8;
  a.npatch = (int64_t)B * a.tiles_y * a.tiles_x;
  const bool fast = a.nrn.kind == SNNQP_NEURON_MULTI_STEP_LIF && a.nrn.inv_k != 0.0f;
  const unsigned gy = (unsigned)((g->Cout + 127) / 128);
  if (in_type == SNNQP_BITS) {
    if (fast) {
      const int gx = persistent_grid(conv3x3_bits_kernel<128, true>, a.npatch);
      hipLaunchKernelGGL((conv3x3_bits_kernel<128, true>), dim3(gx, gy), dim3(256), 0, st, a);
    } else {
      const int gx = persistent_grid(conv3x3_bits_kernel<128, false>, a.npatch);
      hipLaunchKernelGGL((conv3x3_bits_kernel<128, false>), dim3(gx, gy), dim3(256), 0, st, a);
    }
  } else {
    if (fast) {
      const int gx = persistent_grid(conv3x3_u8c2_kernel<true>, a.npatch);
      hipLaunchKernelGGL((conv3x3_u8c2_kernel<true>), dim3(gx, gy), dim3(256), 0, st, a);
    } else {
      const int gx = persistent_grid(conv3x3_u8c2_kernel<false>, a.npatch);
      hipLaunchKernelGGL((conv3x3_u8c2_kernel<false>), dim3(gx, gy), dim3(256), 0, st, a);
    }
  }
  SNNQP_CHECK_LAUNCH("conv3x3 mfma kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
