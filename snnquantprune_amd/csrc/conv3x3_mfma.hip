// Fused SpikingBlock for the 3x3 / stride 1 / pad 1 QuantConv layers of the
// DVS128 topology (examples/tcja/models.py:111-147): implicit-GEMM int8 MFMA
// (v_mfma_i32_32x32x32_i8) + dequantisation + eval BatchNorm + neuron update +
// optional 2x2 max-pool, with the T loop inside the kernel.  Two kernels:
//   conv3x3_bits_kernel  bit-packed input, Cin <= 128, any int8 codes and neuron kind
//                        (codes of magnitude <= 7 go to conv3x3_fp6.hip instead)
//   conv3x3_u8c2_kernel  uint8 event counts, Cin = 2 (the first layer)
//
// Mapping of the bits kernel (one 256-thread workgroup = 4 waves, persistent over patches):
//  * a patch is 8x8 output pixels of one sample = two 32-row MFMA tiles (4x8
//    pixels each); wave w owns output channels [32w, 32w+32) of a 128-channel
//    block (blockIdx.y);
//  * the wave's weights -- all 9 taps x Cin for its 32 channels -- live in
//    registers for the whole launch (the B operand; 144 registers at Cin = 128),
//    loaded once from the MFMA-tiled codes (snnqp_pack_codes_mfma);
//  * per timestep the 10x10 halo of input spikes is expanded from bits to
//    {0,1} bytes into LDS once (one plane per k-step, conv_tile.h) and every tap's A
//    fragment is one ds_read_b128 at an immediate offset, fetched one tap ahead of
//    the MFMAs that consume it;
//  * C/D layout: lane = output channel, register = pixel, so the per-channel
//    dequant/BatchNorm constants are per-lane registers, the membrane potential
//    of the patch stays in 32 VGPRs for all T, and the v_cmp that thresholds a
//    register *is* the packed spike word of two pixels (64-bit lane mask);
//    pooling is an OR of those scalar masks;
//  * the loop is software-pipelined over t: the MFMAs of step t+1 and the
//    dequant/BN/neuron epilogue of step t alternate in one basic block, so the
//    matrix pipe and the VALU overlap inside the single wave each SIMD holds.
#include <type_traits>

#include "conv_tile.h"

namespace snnqp {

#ifdef SNNQP_CLOCK_PROBE
// Diagnostic build only (python csrc/build.py with SNNQP_PROBE=1): shader-clock
// and 100 MHz real-time stamps of workgroup 0 around the persistent loop, to
// read the clock the chip sustains inside this kernel.  Never in the product.
__device__ unsigned long long snnqp_clock_probe[4];
extern "C" int snnqp_debug_read_probe(unsigned long long *out4) {
  return (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(snnqp_clock_probe), 32);
}
#define PROBE_BEGIN()                                                        \
  unsigned long long pc0 = 0, pr0 = 0;                                       \
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {              \
    pc0 = __builtin_amdgcn_s_memtime();                                      \
    pr0 = __builtin_amdgcn_s_memrealtime();                                  \
  }
#define PROBE_END()                                                          \
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {              \
    snnqp_clock_probe[0] = __builtin_amdgcn_s_memtime() - pc0;               \
    snnqp_clock_probe[1] = __builtin_amdgcn_s_memrealtime() - pr0;           \
  }
#define PHASE_DECL() unsigned long long ph_t = 0, ph_acc[3] = {0, 0, 0};
#define PHASE_START() ph_t = __builtin_amdgcn_s_memtime();
#define PHASE_MARK(i)                                         \
  {                                                           \
    const unsigned long long n__ = __builtin_amdgcn_s_memtime(); \
    ph_acc[i] += n__ - ph_t;                                  \
    ph_t = n__;                                               \
  }
#define PHASE_DUMP()                                                    \
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {         \
    snnqp_clock_probe[1] = ph_acc[0];                                   \
    snnqp_clock_probe[2] = ph_acc[1];                                   \
    snnqp_clock_probe[3] = ph_acc[2];                                   \
  }
#else
#define PROBE_BEGIN()
#define PROBE_END()
#define PHASE_DECL()
#define PHASE_START()
#define PHASE_MARK(i)
#define PHASE_DUMP()
#endif

// ---------------------------------------------------------------------------
// Bit-packed input, Cin = 128.
// ---------------------------------------------------------------------------
template <int NF, bool POOL, int LUTM, int CIN>
__device__ __forceinline__ void conv3x3_bits_body(const ConvMfmaArgs &a) {
  static_assert(LUTM != LUT_CHANNEL, "per-channel tables of K = 1152 do not fit LDS");
  static_assert(CIN == 64 || CIN == 128, "two or four 32-channel planes");
  constexpr int KK = CIN / 32;
  constexpr int NSLOT = 18 * KK;                 // MFMAs of one step (9 taps x KK x 2 tiles)
  constexpr int PPS = (64 + NSLOT - 1) / NSLOT;  // epilogue pieces per MFMA slot
  constexpr int NTASK = HALO * HALO * KK;        // (pixel, word) staging tasks
  constexpr int TPT = (NTASK + 255) / 256;       // tasks per thread
  constexpr int LUT_BYTES = LutBytes<LUTM>::value;
  constexpr int LUT_OFF = 2 * HALO_BYTES;
  constexpr int FL = OutStage<POOL>::FL;
  __shared__ __attribute__((aligned(16))) uint8_t
      lds[2 * HALO_BYTES + LUT_BYTES + OutStage<POOL>::BYTES];
  uint32_t *obuf = (uint32_t *)(lds + LUT_OFF + LUT_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int cout_base = blockIdx.y * 128 + wave * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cpar = cout < a.Cout ? cout : a.Cout - 1;      // parameter loads
  const uint32_t cmask = chan_mask(cout_base, a.Cout);
  if (LUTM == LUT_SHARED) build_lut((float *)(lds + LUT_OFF), a.lut_bound, a.dq, tid);
  // start value of every accumulator chain: the address of the entry of acc = 0
  const v16i cb = splat16(LUTM == LUT_SHARED ? (int)lds_addr(lds) + LUT_OFF + 4 * a.lut_bound : 0);

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

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f, a.nrn.vr};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cpar]; lc.bmul = a.bn.mul[cpar]; lc.bbias = a.bn.bias[cpar]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cpar];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // A fragment of (tap, kk), tile tl: pixel (4 tl + ty + dy, tx + dx) of plane kk; the
  // lane halves are swapped on odd halo rows, so dy = 1 uses the other base
  const int pixb = (ty * HPITCH + tx) * 32;
  const int abase_even = pixb + ((h ^ (ty & 1)) << 4);
  const int abase_odd = pixb + ((h ^ (ty & 1) ^ 1) << 4);
  auto aoff = [&](int tap, int kk, int tl) -> int {
    return ((tap / 3) & 1 ? abase_odd : abase_even) + kk * HPLANE +
           ((tap / 3 + 4 * tl) * HPITCH + tap % 3) * 32;
  };
  const uint32_t *xb = (const uint32_t *)a.x;
  // a pixel has ceil(Cin / 32) spike words in memory; planes beyond them stay zero
  const int wpm = (a.Cin + 31) >> 5;
  // LDS word index of this lane's spike word (tile 0 / 1) inside one obuf slot
  const int ob0 = out_pix<POOL>(0, lane) * 4 + wave;
  const int ob1 = out_pix<POOL>(1, lane) * 4 + wave;
  const bool store_lane = POOL ? lane < 8 : lane < 32;

  PROBE_BEGIN()
  PatchWalk pw(a);
  for (int64_t r = pw.first; r < pw.count; r += pw.stride) {
    int b, y0, x0;
    pw.decode(a, r, b, y0, x0);

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
          if (wi < wpm && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            wv = xb[(int64_t)t * a.xs_t + (int64_t)b * a.xs_b +
                    ((int64_t)gy * a.W + gx) * wpm + wi];
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
          *(v4i *)(base + halo_addr(hy, hx, wi, 0)) = expand16<LUTM != LUT_NONE>(stg[k] & 0xFFFFu);
          *(v4i *)(base + halo_addr(hy, hx, wi, 1)) = expand16<LUTM != LUT_NONE>(stg[k] >> 16);
        }
      }
    };
    // all 72 MFMAs of one step; A fragments are fetched one tap ahead
    auto mfma_step = [&](const uint8_t *base, v16i &acc0, v16i &acc1) {
      v4i A[2][2 * KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        A[0][kk] = *(const v4i *)(base + aoff(0, kk, 0));
        A[0][KK + kk] = *(const v4i *)(base + aoff(0, kk, 1));
      }
      acc0 = cb;
      acc1 = cb;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            A[(tap + 1) & 1][kk] = *(const v4i *)(base + aoff(tap + 1, kk, 0));
            A[(tap + 1) & 1][KK + kk] = *(const v4i *)(base + aoff(tap + 1, kk, 1));
          }
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][kk], bf[tap][kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][KK + kk], bf[tap][kk], acc1, 0, 0, 0);
        }
      }
    };
    // One pipelined step in a hand-placed order: 72 issue slots (36 at Cin <= 64), each =
    // one MFMA of step t+1, one A-fragment read for the next tap, and one (two) quarter(s)
    // of a neuron pair of step t's epilogue (16 pairs x 4 pieces = 64 pieces), fenced
    // with sched_barrier so the order survives.  An in-order wave overlaps the
    // matrix pipe and the VALU only when their instructions alternate; left to
    // itself the scheduler emits bursts of MFMAs and bursts of VALU (measured:
    // time = sum of the two instead of their maximum).
    //   piece 1 (pair j+1): dequantise (LDS table reads or packed arithmetic)
    //   piece 2 (pair j)  : BatchNorm (3 packed ops)
    //   piece 3 (pair j)  : membrane update (3 packed ops) + threshold compares
    //   piece 4 (pair j)  : reset + spike word select
    auto fused_step = [&](const uint8_t *base, v16i &accN0, v16i &accN1,
                          const v16i &accC0, const v16i &accC1, int t) {
      v4i A[2][2 * KK];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        A[0][2 * kk] = *(const v4i *)(base + aoff(0, kk, 0));
        A[0][2 * kk + 1] = *(const v4i *)(base + aoff(0, kk, 1));
      }
      accN0 = cb;
      accN1 = cb;
      v2f y[2], x, uu;
      unsigned long long m0 = 0, m1 = 0;
      uint32_t w0 = 0, w1 = 0;
      auto piece1 = [&](int j) {            // j = pair index 0..15
        const int a0 = (j < 8) ? accC0[(j & 7) * 2] : accC1[(j & 7) * 2];
        const int a1 = (j < 8) ? accC0[(j & 7) * 2 + 1] : accC1[(j & 7) * 2 + 1];
        y[j & 1] = dequant_pair<LUTM>(a0, a1, a.dq);
      };
      piece1(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int slot = 0; slot < NSLOT; ++slot) {
        const int tap = slot / (2 * KK), m = slot % (2 * KK), kk = m >> 1, tl = m & 1;
        // A fragment for the same position of the next tap
        if (tap + 1 < 9)
          A[(tap + 1) & 1][m] = *(const v4i *)(base + aoff(tap + 1, kk, tl));
#if defined(SNNQP_BITS_ABL) && (SNNQP_BITS_ABL & 1)   // diagnostic build: 2 of 72 MFMAs
        if (slot < 2) {
#else
        {
#endif
          if (tl == 0)
            accN0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][m], bf[tap][kk], accN0, 0, 0, 0);
          else
            accN1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[tap & 1][m], bf[tap][kk], accN1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < PPS; ++q) {
          const int pc = slot * PPS + q;
#if defined(SNNQP_BITS_ABL) && (SNNQP_BITS_ABL & 2)   // diagnostic build: no epilogue
          if (true) continue;
#endif
          if (pc >= 64) continue;
          const int j = pc >> 2, piece = pc & 3;
          float *up = (j < 8) ? &u[0][(j & 7) * 2] : &u[1][(j & 7) * 2];
          if (piece == 0) {
            if (j + 1 < 16) piece1(j + 1);
          } else if (piece == 1) {
            x = y[j & 1] - lc.bmean;
            x = x * lc.bmul;
            x = x + lc.bbias;
          } else if (piece == 2) {
            uu = neuron_update<NF>(x, v2f{up[0], up[1]}, lc, a.nrn);
            m0 = __ballot(uu.x >= a.nrn.vth);
            m1 = __ballot(uu.y >= a.nrn.vth);
          } else {
            up[0] = neuron_reset<NF>(uu.x, m0, lc);
            up[1] = neuron_reset<NF>(uu.y, m1, lc);
            uint32_t &w = (j < 8) ? w0 : w1;
            const int i = (j & 7) * 2;
            if (POOL) {
              const unsigned long long o = m0 | m1;
              const uint32_t pw = (uint32_t)o | (uint32_t)(o >> 32);
              w = writelane_u32(pw, i >> 1, w);
            } else {
              const int r0 = (i & 3) + 8 * (i >> 2);
              w = writelane_u32((uint32_t)m0, r0, w);
              w = writelane_u32((uint32_t)(m0 >> 32), r0 + 4, w);
              w = writelane_u32((uint32_t)m1, r0 + 1, w);
              w = writelane_u32((uint32_t)(m1 >> 32), r0 + 5, w);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (store_lane) {
        uint32_t *o = obuf + (t % FL) * (OutStage<POOL>::NPIX * 4);
        o[ob0] = w0 & cmask;
        o[ob1] = w1 & cmask;
      }
    };
    auto epilogue = [&](const v16i &acc0, const v16i &acc1, int t) {
      const uint32_t w0 = tile_epilogue<NF, POOL, LUTM>(acc0, u[0], a.dq, lc, a.nrn, lane);
      const uint32_t w1 = tile_epilogue<NF, POOL, LUTM>(acc1, u[1], a.dq, lc, a.nrn, lane);
      if (store_lane) {
        uint32_t *o = obuf + (t % FL) * (OutStage<POOL>::NPIX * 4);
        o[ob0] = w0 & cmask;
        o[ob1] = w1 & cmask;
      }
    };
    // call right after the barrier that follows epilogue(t)
    auto flush_after = [&](int t) {
      if ((t + 1) % FL == 0 || t + 1 == a.T) {
        flush_out<POOL>(obuf, a, t - t % FL, t % FL + 1, b, y0, x0, tid);
        lds_barrier();
      }
    };

    // pipeline prologue: halo(0) staged, MFMA(0) done, halo(1) staged
    v16i accA0, accA1, accB0, accB1;
    stage_load(0);
    stage_store(0);
    if (a.T > 1) stage_load(1);
    lds_barrier();
    mfma_step(lds, accA0, accA1);
    if (a.T > 1) stage_store(1);
    lds_barrier();

    // steady state, unrolled by two so the accumulator roles alternate:
    //   MFMA(t+1) -> next  ||  epilogue(t) <- cur ; then stage halo(t+2)
    int t = 0;
    for (; t + 2 < a.T; t += 2) {
      stage_load(t + 2);
      fused_step(lds + HALO_BYTES, accB0, accB1, accA0, accA1, t);   // MFMA(t+1) || epilogue(t)
      stage_store(0);                                  // halo(t+2) -> even buffer
      lds_barrier();
      flush_after(t);
      if (t + 3 < a.T) stage_load(t + 3);
      fused_step(lds, accA0, accA1, accB0, accB1, t + 1);            // MFMA(t+2) || epilogue(t+1)
      if (t + 3 < a.T) stage_store(1);                 // halo(t+3) -> odd buffer
      lds_barrier();
      flush_after(t + 1);
    }
    // here MFMA(t) is in accA and, if t+1 < T, halo(t+1) is staged in the odd buffer
    if (t + 1 < a.T) {
      fused_step(lds + HALO_BYTES, accB0, accB1, accA0, accA1, t);
      lds_barrier();
      flush_after(t);
      epilogue(accB0, accB1, t + 1);
      lds_barrier();
      flush_after(t + 1);
    } else {
      epilogue(accA0, accA1, t);
      lds_barrier();   // also: LDS is re-staged by the next patch
      flush_after(t);
    }
    if (a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
  }
  PROBE_END()
}

template <int NF, bool POOL, int LUTM>
__global__ void __launch_bounds__(256, 1)
conv3x3_bits_kernel(ConvMfmaArgs a) {            // 64 < Cin <= 128
  conv3x3_bits_body<NF, POOL, LUTM, 128>(a);
}

template <int NF, bool POOL, int LUTM>
__global__ void __launch_bounds__(256, 1)
conv3x3_bits64_kernel(ConvMfmaArgs a) {          // Cin <= 64: half the k-steps
  conv3x3_bits_body<NF, POOL, LUTM, 64>(a);
}

// ---------------------------------------------------------------------------
// u8 event-count input with Cin = 2 (the DVS polarity pair, conv0): K = 18 of
// one 32-deep MFMA step.  One MFMA feeds 1024 neuron updates, so the kernel is
// bound by the epilogue and everything else is kept off the VALU:
//  * the halo of ALL timesteps of a patch (a chunk of <= 32) is staged in LDS at
//    once, so the t loop has no global loads and no barriers;
//  * each timestep image holds the 10 x 10 x 2-byte halo twice, copy c with pixel
//    hx at byte 2 hx + 2 c of its 24-byte row: the 8 bytes that start at any pixel
//    are then a 4-byte-aligned ds_read2_b32 in the copy of the pixel's parity, and
//    a lane's A fragment is two such reads and no arithmetic:
//      half 0: k 0..7 = row dy 0, k 8..15 = row dy 1   (byte b = 2 dx + cin, b < 6)
//      half 1: k 16..23 = row dy 2, k 24..31 = constants {127,127,127,127,1,0,0,0}
//    bytes 6, 7 of a row read belong to the next pixel; their B rows are zero;
//  * the constant k rows carry the table address: B rows 24..28 of a channel sum
//    to the byte address of its entry of acc = 0 (C = 0, no accumulator preload).
// ---------------------------------------------------------------------------
constexpr int HROW2 = 24;                 // LDS bytes per halo row
constexpr int HCOPY2 = HALO * HROW2;      // one copy of one timestep
constexpr int HCONST2 = 2 * HCOPY2;       // the 8 constant bytes
constexpr int HIMG2 = 496;                // one timestep image
constexpr int TCHUNK = 32;                // most timesteps staged per pass

typedef int v2i_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(3))) const v2i_a4 lds_cv2i_t;

// LDS bytes of the u8c2 kernel: images | table | spike words
__host__ __device__ inline int u8c2_table_bytes(int lutm, int bound) {
  const int b = lutm == LUT_CHANNEL ? 128 * lut_channel_rows(bound) * 4
                : lutm == LUT_SHARED ? (2 * bound + 2) * 4 : 0;
  return (b + 15) & ~15;
}

template <int NF, bool POOL, int LUTM>
__global__ void __launch_bounds__(256, SNNQP_U8C2_WPS)
conv3x3_u8c2_kernel(ConvMfmaArgs a) {
  constexpr int FL = OutStage<POOL>::FL;
  static_assert(TCHUNK % FL == 0, "flush period must divide the staging chunk");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tc = a.tchunk;                       // multiple of 8, <= TCHUNK
  const int lut_off = tc * HIMG2;
  uint32_t *obuf = (uint32_t *)(lds + lut_off + u8c2_table_bytes(LUTM, a.lut_bound));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int cout_base = blockIdx.y * 128 + wave * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cpar = cout < a.Cout ? cout : a.Cout - 1;      // parameter loads
  const uint32_t cmask = chan_mask(cout_base, a.Cout);
  // smallest non-zero |input current| of this workgroup's channels (per-channel tables
  // only): decides whether the membrane update may be one fused multiply-add
  uint32_t *wgmin = obuf + OutStage<POOL>::BYTES / 4;
  if (LUTM == LUT_CHANNEL) {
    if (tid == 0) *wgmin = 0x7F800000u;
    lds_barrier();
  }
  // tables and constants become visible with the first staging barrier
  if (LUTM == LUT_SHARED) build_lut((float *)(lds + lut_off), a.lut_bound, a.dq, tid);
  if (LUTM == LUT_CHANNEL) {
    const uint32_t mb = build_lut_channel((float *)(lds + lut_off), a.lut_bound, a.dq, a.bn,
                                          blockIdx.y * 128, a.Cout, tid);
    atomicMin(wgmin, mb);
  }
  if (tid < tc) {
    *(uint32_t *)(lds + tid * HIMG2 + HCONST2) = 0x7F7F7F7Fu;
    *(uint32_t *)(lds + tid * HIMG2 + HCONST2 + 4) = 0x00000001u;
  }

  // without a table the input is taken as x - 128 (a signed int8 for every count up to
  // 255; padding pixels are x = 0 like any other) and 128 * sum_k w[k] is added back to
  // the accumulator in the epilogue (an integer below 2^24: exact in float32)
  constexpr bool OFFS = LUTM == LUT_NONE;
  float acc_off = 0.0f;
  v4i bf;
  {
    // byte address of this lane's table entry of acc = 0, as 127 * q + r over the
    // constant k rows: rows 24..27 take q in parts of at most 127, row 28 takes r
    int bias = 0;
    if (LUTM == LUT_SHARED) bias = (int)lds_addr(lds) + lut_off + 4 * a.lut_bound;
    if (LUTM == LUT_CHANNEL)     // the wave's block, row of acc = 0, this lane's channel
      bias = (int)lds_addr(lds) + lut_off +
             4 * ((wave * lut_channel_rows(a.lut_bound) + a.lut_bound) * 32 + n);
    int q = bias / 127;
    const int r = bias - 127 * q;
    int wsum = 0;
    int v[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t pk = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * h + 4 * d + j;
        uint32_t bv = 0;
        if (k < 24) {
          const int dy = k >> 3, b = k & 7;
          if (b < 6 && cout < a.Cout) {   // HWIO with Cin = 2: row (3 dy + dx) * 2 + cin = 6 dy + b
            const int code = a.w[(int64_t)(6 * dy + b) * a.Cout + cout];
            wsum += code;
            bv = (uint8_t)(LUTM == LUT_CHANNEL ? code * 8 : code);   // see build_lut_channel
          }
        } else if (k < 28) {
          const int part = q < 127 ? q : 127;
          q -= part;
          bv = (uint32_t)part;
        } else if (k == 28) {
          bv = (uint32_t)r;
        }
        pk |= bv << (8 * j);
      }
      v[d] = (int)pk;
    }
    bf = v4i{v[0], v[1], v[2], v[3]};
    // the two lane halves hold k 0..15 and 16..23 of the same channel
    if (OFFS) acc_off = 128.0f * (float)(wsum + __shfl_xor(wsum, 32));
  }

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f, a.nrn.vr};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cpar]; lc.bmul = a.bn.mul[cpar]; lc.bbias = a.bn.bias[cpar]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cpar];

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // the two 8-byte reads of this lane's fragment (tile 1 is 4 halo rows further)
  const int cpy = tx & 1;
  const int px0 = (int)lds_addr(lds) + cpy * HCOPY2 + 2 * tx + 2 * cpy;
  int offA[2], offB[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    offA[tl] = px0 + (ty + 4 * tl + (h ? 2 : 0)) * HROW2;
    offB[tl] = h ? (int)lds_addr(lds) + HCONST2 : px0 + (ty + 4 * tl + 1) * HROW2;
  }
  const uint8_t *xb = (const uint8_t *)a.x;
  const int ob0 = out_pix<POOL>(0, lane) * 4 + wave;
  const int ob1 = out_pix<POOL>(1, lane) * 4 + wave;
  const bool store_lane = POOL ? lane < 8 : lane < 32;

  PHASE_DECL()
  PROBE_BEGIN()
  PatchWalk pw(a);
  for (int64_t r = pw.first; r < pw.count; r += pw.stride) {
    int b, y0, x0;
    pw.decode(a, r, b, y0, x0);

    float u[2][16];
    if (a.u0 && wave_on) u_io<true>(u, a, b, y0, x0, cout, h);
    else zero_u(u);

    for (int t0 = 0; t0 < a.T; t0 += tc) {
      const int nt = min(tc, a.T - t0);
      PHASE_START()
      lds_barrier();                       // previous readers of the LDS images are done
      for (int tb = 0; tb < nt * (HALO * HALO); tb += 8 * 256) {
        uint16_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {        // all loads first, then all LDS writes
          const int task = tb + tid + k * 256;
          const int tt = task / (HALO * HALO), pix = task % (HALO * HALO);
          const int gy = y0 + pix / HALO - 1, gx = x0 + pix % HALO - 1;
          v[k] = 0;
          if (tt < nt && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v[k] = *(const uint16_t *)(xb + (int64_t)(t0 + tt) * a.xs_t +
                                       (int64_t)b * a.xs_b + ((int64_t)gy * a.W + gx) * 2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int task = tb + tid + k * 256;
          const int tt = task / (HALO * HALO), pix = task % (HALO * HALO);
          if (tt < nt) {    // table modes: both bytes scale without a carry (counts <= 31 / 7)
            const uint16_t val = LUTM == LUT_CHANNEL  ? (uint16_t)(v[k] << 4)
                                 : LUTM == LUT_SHARED ? (uint16_t)(v[k] << 2)
                                                      : (uint16_t)(v[k] ^ 0x8080u);   // x - 128
            uint8_t *p = lds + tt * HIMG2 + (pix / HALO) * HROW2 + (pix % HALO) * 2;
            *(uint16_t *)p = val;
            *(uint16_t *)(p + HCOPY2 + 2) = val;
          }
        }
      }
      lds_barrier();
      PHASE_MARK(0)
      // the FL-step blocks of the chunk, with the membrane update as a fused
      // multiply-add where that is proven bit-identical for this launch
      auto run_chunk = [&](auto fma_tag) {
      constexpr bool FMA = decltype(fma_tag)::value;
      for (int tf = 0; tf < nt; tf += FL) {          // FL steps, then flush
        const int nf = min(FL, nt - tf);
#pragma unroll SNNQP_U8C2_UNROLL
        for (int tt = tf; tt < tf + nf; ++tt) {
          const uint32_t img = (uint32_t)(tt * HIMG2);
          uint32_t words[2];
#pragma unroll
          for (int tl = 0; tl < 2; ++tl) {
            const v2i_a4 lo = *(lds_cv2i_t *)(uintptr_t)(img + (uint32_t)offA[tl]);
            const v2i_a4 hi = *(lds_cv2i_t *)(uintptr_t)(img + (uint32_t)offB[tl]);
            v16i acc = splat16(0);
#if defined(SNNQP_ABL) && (SNNQP_ABL & 2)   // diagnostic build: no MFMA
            acc[0] = bf.w + ((lo.x ^ lo.y ^ hi.x) & 4);
#else
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{lo.x, lo.y, hi.x, hi.y}, bf, acc,
                                                        0, 0, 0);
#endif
            words[tl] = tile_epilogue<NF, POOL, LUTM, FMA, OFFS>(acc, u[tl], a.dq, lc, a.nrn,
                                                                   lane, acc_off);
          }
          if (store_lane) {
            uint32_t *o = obuf + ((t0 + tt) % FL) * (OutStage<POOL>::NPIX * 4);
            o[ob0] = words[0] & cmask;
            o[ob1] = words[1] & cmask;
          }
        }
        PHASE_MARK(1)
        lds_barrier();
        flush_out<POOL>(obuf, a, t0 + tf, nf, b, y0, x0, tid);
        lds_barrier();
        PHASE_MARK(2)
      }
      };
      bool fma_ok = false;
      if (NF == NF_MUL0 && LUTM == LUT_CHANNEL)
        fma_ok = lif_fma_is_exact(*wgmin, a.nrn.k_log2, a.T, a.u0 != nullptr);
      if (NF == NF_MUL0 && LUTM == LUT_CHANNEL && fma_ok) run_chunk(std::true_type{});
      else run_chunk(std::false_type{});
    }
    if (a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
  }
  PROBE_END()
  PHASE_DUMP()
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
  if (g->H <= 0 || g->W <= 0) return "empty image";    // any size: edge patches are clipped
  if (g->Cout <= 0) return "no output channels";    // any count: the last word is masked
  if (s_type != SNNQP_BITS) return "spike output must be bit-packed";
  if (in_type == SNNQP_BITS) {
    // any width up to 128: `wt` is tiled from the kernel zero-padded along Cin to 64
    // (Cin <= 64) or 128; the spike words beyond ceil(Cin / 32) are not read
    if (g->Cin < 1 || g->Cin > 128) return "bit input needs Cin <= 128";
  } else if (in_type == SNNQP_U8) {     // any count 0..255 (taken as x - 128 without a table)
    if (g->Cin != 2) return "u8 input needs Cin == 2";
  } else {
    return "input must be BITS or U8";
  }
  if (nrn->kind == SNNQP_NEURON_LIF && !nrn->decay) return "LIF without decay";
  return nullptr;
}

int run_conv3x3_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b,
                     int32_t T, int32_t B, const snnqp_conv_geom_t *g,
                     const snnqp_weight_t *w, const int8_t *wt,
                     const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                     const float *u0, float *u_out, uint32_t *s_out, int pool,
                     int x_max, hipStream_t st) {
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
  a.tiles_y = (g->H + 7) / 8; a.tiles_x = (g->W + 7) / 8;
  a.npatch = (int64_t)B * a.tiles_y * a.tiles_x;
  const int nf = neuron_form(a.nrn);          // which straight-line epilogue (conv_tile.h)
  const bool pl = pool == 2;
  const unsigned gy = (unsigned)((g->Cout + 127) / 128);
  // |acc| <= abs_sum_max * x_max; small enough -> dequantise through an LDS table
  // (the A operand then carries 4 * x, which must stay an int8)
  const int64_t xm = in_type == SNNQP_BITS ? 1 : x_max;
  const int64_t bound = (int64_t)w->abs_sum_max * xm;
  const bool lut = w->abs_sum_max > 0 && xm > 0 && xm <= LUT_XMAX && bound <= LUT_CAP;
  a.lut_bound = lut ? (int32_t)bound : 0;
  a.tchunk = T >= TCHUNK ? TCHUNK : (T + 7) & ~7;
  const size_t lds_fixed = (size_t)a.tchunk * HIMG2 + 16 +
                           (pl ? OutStage<true>::BYTES : OutStage<false>::BYTES);
  // per-channel tables (BatchNorm folded in) while the workgroup stays within 64 KiB of LDS
  // (there the accumulator counts table rows of 128 B: A = 16 x input, B = 8 x code)
  const bool lutc = lut && in_type == SNNQP_U8 && bound <= LUT2_CAP && xm <= 7 &&
                    w->code_max > 0 && w->code_max <= 15 &&
                    lds_fixed + u8c2_table_bytes(LUT_CHANNEL, (int)bound) <= 65536;
#define SNNQP_CONV_LAUNCH_NF(KERN, NFV, LM, LDS)                                   \
  do {                                                                             \
    if (pl) launch_persistent(KERN<NFV, true, LM>, a, gy, st, LDS);                 \
    else launch_persistent(KERN<NFV, false, LM>, a, gy, st, LDS);                   \
  } while (0)
#define SNNQP_CONV_LAUNCH(KERN, LM, LDS)                                           \
  do {                                                                             \
    if (nf == NF_MUL0) SNNQP_CONV_LAUNCH_NF(KERN, NF_MUL0, LM, LDS);                \
    else if (nf == NF_MUL) SNNQP_CONV_LAUNCH_NF(KERN, NF_MUL, LM, LDS);             \
    else if (nf == NF_DIV) SNNQP_CONV_LAUNCH_NF(KERN, NF_DIV, LM, LDS);             \
    else SNNQP_CONV_LAUNCH_NF(KERN, NF_DECAY, LM, LDS);                             \
  } while (0)
  if (in_type == SNNQP_BITS && w->code_max > 0 && w->code_max <= 7) {
    launch_conv3x3_fp6(a, nf, pl, lut, gy, st);  // codes exact in fp6: f8f6f4 MFMA
  } else if (in_type == SNNQP_BITS) {
    if (g->Cin <= 64) {
      if (lut) SNNQP_CONV_LAUNCH(conv3x3_bits64_kernel, LUT_SHARED, 0);
      else SNNQP_CONV_LAUNCH(conv3x3_bits64_kernel, LUT_NONE, 0);
    } else {
      if (lut) SNNQP_CONV_LAUNCH(conv3x3_bits_kernel, LUT_SHARED, 0);
      else SNNQP_CONV_LAUNCH(conv3x3_bits_kernel, LUT_NONE, 0);
    }
  } else {
    const int lm = lutc ? LUT_CHANNEL : lut ? LUT_SHARED : LUT_NONE;
    const size_t ldsb = lds_fixed + u8c2_table_bytes(lm, a.lut_bound);
    if (lutc) SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_CHANNEL, ldsb);
    else if (lut) SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_SHARED, ldsb);
    else SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_NONE, ldsb);
  }
#undef SNNQP_CONV_LAUNCH
#undef SNNQP_CONV_LAUNCH_NF
  SNNQP_CHECK_LAUNCH("conv3x3 mfma kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
