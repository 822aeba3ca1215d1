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
//    registers for the whole launch (the B operand; 144 registers at Cin = 128),
//    loaded once from the MFMA-tiled codes (snnqp_pack_codes_mfma);
//  * per timestep the 10x10 halo of input spikes is expanded from bits to
//    {0,1} bytes into LDS once (XOR-swizzled 16-byte chunks) and every tap's A
//    fragment is one ds_read_b128 at a shifted pixel, fetched one tap ahead of
//    the MFMAs that consume it;
//  * C/D layout: lane = output channel, register = pixel, so the per-channel
//    dequant/BatchNorm constants are per-lane registers, the membrane potential
//    of the patch stays in 32 VGPRs for all T, and the v_cmp that thresholds a
//    register *is* the packed spike word of two pixels (64-bit lane mask);
//    pooling is an OR of those scalar masks;
//  * the loop is software-pipelined over t: the MFMAs of step t+1 and the
//    dequant/BN/neuron epilogue of step t are independent instruction streams
//    in one basic block, so the matrix pipe and the VALU overlap inside a wave.
#include "kernels.h"

namespace snnqp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int HALO = 10;
constexpr int PIXB = 128;                       // LDS bytes per halo pixel
constexpr int HALO_BYTES = HALO * HALO * PIXB;  // one expanded halo image

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

struct LaneConsts {
  float bmean, bmul, bbias, dec;
};

// Dequant + BN + neuron for one 32x32 tile, straight-line.  The lane mask of
// register i holds pixel (ty = 2*(i>>3), tx = i&7) in its low half and
// (ty + 1, tx) in its high half.  Returns the word this lane stores:
//   POOL : lanes 0..7  = pooled pixel (pty = lane >> 2, ptx = lane & 3)
//   !POOL: lanes 0..31 = pixel row `lane` of the tile
template <bool FAST>
__device__ __forceinline__ unsigned long long neuron_elem(int acc, float &u,
                                                          const Dequant &dq,
                                                          const LaneConsts &lc,
                                                          const NeuronP &nrn) {
  float cur = dequant_acc_nb(acc, dq);
  cur = bn_apply(cur, lc.bmean, lc.bmul, lc.bbias);
  bool s;
  if (FAST) {
    // multi_step_LIF with tau a power of two and v_reset == 0
    // (spiking_learning.py:410-414): u - 0 == u exactly, and with float32
    // subnormals kept (hipcc default) (u - v_th) >= 0  <=>  u >= v_th.
    const float d = cur - u;
    u = u + d * nrn.inv_k;
    s = u >= nrn.vth;
    u = s ? 0.0f : u;
  } else {
    s = neuron_step(u, cur, nrn, lc.dec);
  }
  return __ballot(s);
}

template <bool FAST, bool POOL>
__device__ __forceinline__ uint32_t tile_epilogue(const v16i &acc, float (&u)[16],
                                                  const Dequant &dq,
                                                  const LaneConsts &lc,
                                                  const NeuronP &nrn, int lane) {
  uint32_t myw = 0;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {     // masks are consumed pair by pair
    const unsigned long long m0 = neuron_elem<FAST>(acc[i], u[i], dq, lc, nrn);
    const unsigned long long m1 = neuron_elem<FAST>(acc[i + 1], u[i + 1], dq, lc, nrn);
    if (POOL) {
      const unsigned long long o = m0 | m1;
      const uint32_t pw = (uint32_t)o | (uint32_t)(o >> 32);
      myw = (lane == (i >> 1)) ? pw : myw;
    } else {
      const int r0 = (i & 3) + 8 * (i >> 2);          // row of element i, low half
      myw = (lane == r0) ? (uint32_t)m0 : myw;
      myw = (lane == r0 + 4) ? (uint32_t)(m0 >> 32) : myw;
      myw = (lane == r0 + 1) ? (uint32_t)m1 : myw;
      myw = (lane == r0 + 5) ? (uint32_t)(m1 >> 32) : myw;
    }
  }
  return myw;
}

// Per-lane output word offset (in words) of tile `tl` inside one (t, b) image.
template <bool POOL>
__device__ __forceinline__ int out_word_offset(const ConvMfmaArgs &a, int y0,
                                                   int x0, int tl, int cw, int lane) {
  const int CW = a.Cout >> 5;
  if (POOL) {
    const int OW = a.W >> 1;
    const int oy = (y0 >> 1) + tl * 2 + ((lane >> 2) & 1);
    const int ox = (x0 >> 1) + (lane & 3);
    return (oy * OW + ox) * CW + cw;
  }
  const int ty = ((lane >> 2) & 1) | (((lane >> 4) & 1) << 1);
  const int tx = (lane & 3) | (((lane >> 3) & 1) << 2);
  return ((y0 + tl * 4 + ty) * a.W + (x0 + tx)) * CW + cw;
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

__device__ __forceinline__ void zero_u(float (&u)[2][16]) {
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int i = 0; i < 16; ++i) u[tl][i] = 0.0f;
}

#define ZERO16 v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}

// ---------------------------------------------------------------------------
// Bit-packed input, Cin = 128.
// ---------------------------------------------------------------------------
template <bool FAST, bool POOL>
__global__ void __launch_bounds__(256, 1)
conv3x3_bits_kernel(ConvMfmaArgs a) {
  constexpr int CIN = 128;
  constexpr int KK = CIN / 32;
  constexpr int NTASK = HALO * HALO * KK;        // (pixel, word) staging tasks
  constexpr int TPT = (NTASK + 255) / 256;       // tasks per thread
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * HALO_BYTES];

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

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cout]; lc.bmul = a.bn.mul[cout]; lc.bbias = a.bn.bias[cout]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cout];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // LDS byte offset of this lane's A fragment for (tap, kk), tile 0; tile 1 is
  // 4 halo rows further.  (kk*2 + h) ^ g == (kk*2) ^ (h ^ g) since h is bit 0.
  int aoff[9][KK];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int hy = ty + tap / 3, hx = tx + tap % 3;
    const int g = ((hy & 3) << 1) | ((hx >> 1) & 1);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      aoff[tap][kk] = (hy * HALO + hx) * PIXB + (((kk * 2) ^ (h ^ g)) << 4);
  }
  constexpr int TILE1 = 4 * HALO * PIXB;
  const uint32_t *xb = (const uint32_t *)a.x;
  const int64_t img_words = (int64_t)(POOL ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * (a.Cout >> 5);

  for (int64_t p = blockIdx.x; p < a.npatch; p += gridDim.x) {
    int64_t q = p;
    const int px = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int py = (int)(q % a.tiles_y); q /= a.tiles_y;
    const int b = (int)q;
    const int y0 = py * 8, x0 = px * 8;

    float u[2][16];
    if (a.u0 && wave_on) u_io<true>(u, a, b, y0, x0, cout, h);
    else zero_u(u);

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
      uint8_t *base = lds + buf * HALO_BYTES;
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
    // all 72 MFMAs of one step; A fragments are fetched one tap ahead
    auto mfma_step = [&](const uint8_t *base, v16i &acc0, v16i &acc1) {
      v4i A[2][2 * KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        A[0][kk] = *(const v4i *)(base + aoff[0][kk]);
        A[0][KK + kk] = *(const v4i *)(base + aoff[0][kk] + TILE1);
      }
      acc0 = ZERO16;
      acc1 = ZERO16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            A[(tap + 1) & 1][kk] = *(const v4i *)(base + aoff[tap + 1][kk]);
            A[(tap + 1) & 1][KK + kk] = *(const v4i *)(base + aoff[tap + 1][kk] + TILE1);
          }
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][kk], bf[tap][kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][KK + kk], bf[tap][kk], acc1, 0, 0, 0);
        }
      }
    };
    const int ow0 = out_word_offset<POOL>(a, y0, x0, 0, cw, lane);
    const int ow1 = out_word_offset<POOL>(a, y0, x0, 1, cw, lane);
    const bool store_lane = wave_on && (POOL ? lane < 8 : lane < 32);
    auto epilogue = [&](const v16i &acc0, const v16i &acc1, int t) {
      const uint32_t w0 = tile_epilogue<FAST, POOL>(acc0, u[0], a.dq, lc, a.nrn, lane);
      const uint32_t w1 = tile_epilogue<FAST, POOL>(acc1, u[1], a.dq, lc, a.nrn, lane);
      if (store_lane) {
        uint32_t *o = a.s_out + ((int64_t)t * a.B + b) * img_words;
        o[ow0] = w0;
        o[ow1] = w1;
      }
    };

    // pipeline prologue: halo(0) staged, MFMA(0) done, halo(1) staged
    v16i accA0, accA1, accB0, accB1;
    stage_load(0);
    stage_store(0);
    if (a.T > 1) stage_load(1);
    __syncthreads();
    mfma_step(lds, accA0, accA1);
    if (a.T > 1) stage_store(1);
    __syncthreads();

    // steady state, unrolled by two so the accumulator roles alternate:
    //   MFMA(t+1) -> next  ||  epilogue(t) <- cur ; then stage halo(t+2)
    int t = 0;
    for (; t + 2 < a.T; t += 2) {
      stage_load(t + 2);
      mfma_step(lds + HALO_BYTES, accB0, accB1);       // step t+1 (odd buffer)
      epilogue(accA0, accA1, t);
      stage_store(0);                                  // halo(t+2) -> even buffer
      __syncthreads();
      if (t + 3 < a.T) stage_load(t + 3);
      mfma_step(lds, accA0, accA1);                    // step t+2 (even buffer)
      epilogue(accB0, accB1, t + 1);
      if (t + 3 < a.T) stage_store(1);                 // halo(t+3) -> odd buffer
      __syncthreads();
    }
    // here MFMA(t) is in accA and, if t+1 < T, halo(t+1) is staged in the odd buffer
    if (t + 1 < a.T) {
      mfma_step(lds + HALO_BYTES, accB0, accB1);
      epilogue(accA0, accA1, t);
      epilogue(accB0, accB1, t + 1);
    } else {
      epilogue(accA0, accA1, t);
    }
    __syncthreads();   // LDS is re-staged by the next patch
    if (a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
  }
}

// ---------------------------------------------------------------------------
// u8 event-count input with Cin = 2 (the DVS polarity pair, conv0): K = 18
// padded to one 32-deep MFMA step, k = 2 * tap + cin.  The kernel is bound by
// the per-neuron epilogue (one MFMA per 1024 neuron updates), so the halo of
// ALL timesteps of a patch is staged in LDS at once (T x 240 B) and the t loop
// runs without global loads or barriers.
// ---------------------------------------------------------------------------
constexpr int HROW2 = 24;                 // LDS bytes per halo row (10 px x 2 B, padded)
constexpr int HIMG2 = HALO * HROW2;       // one timestep
constexpr int TCHUNK = 32;                // timesteps staged per pass

template <bool FAST, bool POOL>
__global__ void __launch_bounds__(256, 3)
conv3x3_u8c2_kernel(ConvMfmaArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[TCHUNK * HIMG2];
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

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cout]; lc.bmul = a.bn.mul[cout]; lc.bbias = a.bn.bias[cout]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cout];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // LDS offsets of the 8 taps this lane gathers (tap = 8h + j); lanes with
  // h = 1 only own tap 8, their other reads are masked to zero (no branches)
  int toff[8];
  uint32_t amask[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int tap = 8 * h + j;
    toff[j] = tap < 9 ? (ty + tap / 3) * HROW2 + (tx + tap % 3) * 2 : 0;
  }
#pragma unroll
  for (int d = 0; d < 4; ++d)
    amask[d] = h == 0 ? 0xFFFFFFFFu : (d == 0 ? 0x0000FFFFu : 0u);
  const uint8_t *xb = (const uint8_t *)a.x;
  const int64_t img_words = (int64_t)(POOL ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * (a.Cout >> 5);

  for (int64_t p = blockIdx.x; p < a.npatch; p += gridDim.x) {
    int64_t q = p;
    const int px = (int)(q % a.tiles_x); q /= a.tiles_x;
    const int py = (int)(q % a.tiles_y); q /= a.tiles_y;
    const int b = (int)q;
    const int y0 = py * 8, x0 = px * 8;

    float u[2][16];
    if (a.u0 && wave_on) u_io<true>(u, a, b, y0, x0, cout, h);
    else zero_u(u);
    const int ow0 = out_word_offset<POOL>(a, y0, x0, 0, cw, lane);
    const int ow1 = out_word_offset<POOL>(a, y0, x0, 1, cw, lane);
    const bool store_lane = wave_on && (POOL ? lane < 8 : lane < 32);

    for (int tc = 0; tc < a.T; tc += TCHUNK) {
      const int nt = min(TCHUNK, a.T - tc);
      __syncthreads();                       // previous readers of the LDS image are done
      {
        constexpr int NT2 = (TCHUNK * HALO * HALO + 255) / 256;
        uint16_t v[NT2];
#pragma unroll
        for (int k = 0; k < NT2; ++k) {      // all loads first, then all LDS writes
          const int task = tid + k * 256;
          const int tt = task / (HALO * HALO), pix = task % (HALO * HALO);
          const int gy = y0 + pix / HALO - 1, gx = x0 + pix % HALO - 1;
          v[k] = 0;
          if (tt < nt && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v[k] = *(const uint16_t *)(xb + (int64_t)(tc + tt) * a.xs_t +
                                       (int64_t)b * a.xs_b + ((int64_t)gy * a.W + gx) * 2);
        }
#pragma unroll
        for (int k = 0; k < NT2; ++k) {
          const int task = tid + k * 256;
          const int tt = task / (HALO * HALO), pix = task % (HALO * HALO);
          if (tt < nt)
            *(uint16_t *)(lds + tt * HIMG2 + (pix / HALO) * HROW2 + (pix % HALO) * 2) = v[k];
        }
      }
      __syncthreads();
      if (wave_on) {
        for (int tt = 0; tt < nt; ++tt) {
          const uint8_t *base = lds + tt * HIMG2;
          uint32_t words[2];
#pragma unroll
          for (int tl = 0; tl < 2; ++tl) {
            // A fragment: dword d = taps (8h + 2d, 8h + 2d + 1), each one u16
            int av[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const uint32_t lo = *(const uint16_t *)(base + toff[2 * d] + tl * 4 * HROW2);
              const uint32_t hi = *(const uint16_t *)(base + toff[2 * d + 1] + tl * 4 * HROW2);
              av[d] = (int)((lo | (hi << 16)) & amask[d]);
            }
            v16i acc = ZERO16;
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{av[0], av[1], av[2], av[3]},
                                                        bf, acc, 0, 0, 0);
            words[tl] = tile_epilogue<FAST, POOL>(acc, u[tl], a.dq, lc, a.nrn, lane);
          }
          if (store_lane) {
            uint32_t *o = a.s_out + ((int64_t)(tc + tt) * a.B + b) * img_words;
            o[ow0] = words[0];
            o[ow1] = words[1];
          }
        }
      }
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
static void launch_persistent(K kernel, const ConvMfmaArgs &a, unsigned gy, hipStream_t st) {
  int dev = 0, cus = 256, occ = 2;
  if (hipGetDevice(&dev) == hipSuccess)
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess ||
      occ < 1)
    occ = 1;
  if (occ > 8) occ = 8;
  const int64_t gmax = (int64_t)cus * occ;
  const unsigned gx = (unsigned)(a.npatch < gmax ? a.npatch : gmax);
  hipLaunchKernelGGL(kernel, dim3(gx, gy), dim3(256), 0, st, a);
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
  const bool fast = a.nrn.kind == SNNQP_NEURON_MULTI_STEP_LIF && a.nrn.inv_k != 0.0f &&
                    a.nrn.vr == 0.0f;
  const bool pl = pool == 2;
  const unsigned gy = (unsigned)((g->Cout + 127) / 128);
  if (in_type == SNNQP_BITS) {
    if (fast && pl) launch_persistent(conv3x3_bits_kernel<true, true>, a, gy, st);
    else if (fast) launch_persistent(conv3x3_bits_kernel<true, false>, a, gy, st);
    else if (pl) launch_persistent(conv3x3_bits_kernel<false, true>, a, gy, st);
    else launch_persistent(conv3x3_bits_kernel<false, false>, a, gy, st);
  } else {
    if (fast && pl) launch_persistent(conv3x3_u8c2_kernel<true, true>, a, gy, st);
    else if (fast) launch_persistent(conv3x3_u8c2_kernel<true, false>, a, gy, st);
    else if (pl) launch_persistent(conv3x3_u8c2_kernel<false, true>, a, gy, st);
    else launch_persistent(conv3x3_u8c2_kernel<false, false>, a, gy, st);
  }
  SNNQP_CHECK_LAUNCH("conv3x3 mfma kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
