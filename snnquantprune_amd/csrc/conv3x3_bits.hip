// Fused QuantConv(3x3, stride 1, pad 1) + BatchNorm + neuron (+ 2x2 max-pool) over
// all T timesteps for bit-packed spikes, Cin <= 128, any int8 weight codes.  Replaces
// SpikingBlock.__call__, spiking_learning.py:446-462, with QuantConv
// flax_qconv.py:147-188 and the pool of examples/tcja/models.py:145-147.
//
// The contraction runs on one of two matrix instructions (template parameter FMT):
//  * codes of magnitude <= 7 (DuQ up to 4 bits): the block-scaled f8f6f4 MFMA
//    (32x32x64, K = 64 per instruction), A = spikes as fp4 (e2m1: 0 or 1.0), B = codes
//    as fp6 (e2m3: every integer up to 7 is exact), f32 accumulation of integers below
//    2^24 -- the same integer an int8 MFMA accumulates, at half the A bytes per MAC
//    through LDS and twice the matrix rate;
//  * wider codes (8-bit DuQ, the reference's shipped configs): v_mfma_i32_32x32x32_i8,
//    A = spikes as bytes, B = the int8 codes, 36 k-steps instead of 18.
//
// A workgroup is 4 waves on one 4x8-pixel tile (patch), two workgroups per CU = two waves
// per SIMD: wave w owns output channels [32 w, +32) of the tile, its B fragments for
// the whole launch (108 registers fp6, 144 int8), the tile's membrane potentials and two
// accumulator sets.  (SNNQP_BITS_TILES = 2: 8 waves on an 8x8 patch, wave w on tile w >> 2.)  The loop is software-pipelined over t inside each wave: the
// MFMAs of step t + 1 alternate with the instructions of the neuron epilogue of step
// t; one workgroup barrier per step.  (Two waves per SIMD measured 10-20 % faster than
// the one-wave, two-tiles-per-wave form of the int8 kernel this file replaced.)
//
// The halo of step t + 1 (10 x 10 pixels x Cin spike bits) is expanded to the A format
// (fp4 by a byte -> 8-nibble LDS table, bytes by arithmetic) and written to the other
// LDS buffer during step t; its global load was issued a step earlier.  LDS image: one
// plane per k-step of a tap (64 fp4 / 32 byte channels = 32 B per pixel), rows of 12
// pixels, the two 16-byte lane halves swapped on odd rows: every tap/half offset is an
// instruction immediate and the reads are bank-conflict free.
#include <type_traits>

#include "conv_tile.h"

namespace snnqp {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const v4i lds_cv4i_t;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;

// k-steps of 64: (tap, 64-channel group); Cin = 64 or 128 (template parameter CIN)
constexpr int F6_PITCH = HPITCH;             // pixels per LDS halo row (10 used)
// 4x8-pixel tiles of a workgroup's patch.  1: a workgroup is 4 waves on a 4x8 patch and two
// of them share a CU -- while one sits at its per-step barrier (the waves of a workgroup
// drift apart; removing the barrier measured -7 %) the other's waves have the SIMDs: conv1
// 6.68 -> 6.15 ms although the halo staged per output grows from 100/64 to 60/32 pixels.
// 2: the 8-wave workgroup on an 8x8 patch, one per CU.
#ifndef SNNQP_BITS_TILES
#define SNNQP_BITS_TILES 1
#endif
constexpr int F6_TILES = SNNQP_BITS_TILES;
constexpr int F6_ROWS = 4 * F6_TILES + 2;    // halo rows of a patch
constexpr int F6_NT = 256 * F6_TILES;        // threads of a workgroup: 4 waves per tile
constexpr int F6_PLANE = F6_ROWS * HPITCH * 32;   // one k-step plane of a halo image
constexpr int F6_TAB = 1024;                 // byte -> 8 fp4 nibbles
#ifndef SNNQP_F6_PREFETCH
#define SNNQP_F6_PREFETCH 4
#endif
#ifndef SNNQP_F6_YDIST
#define SNNQP_F6_YDIST 2
#endif

// 4 int8 codes (|c| <= 7) -> 4 e2m3 codes, one per byte
__device__ __forceinline__ uint32_t fp6_codes4(uint32_t x) {
  const uint32_t m1 = (x >> 7) & 0x01010101u;       // 1 where negative
  const uint32_t mag = (x ^ (m1 * 0xFFu)) + m1;     // |c| per byte (no carries)
  // magnitude 0..7 -> 0x00 0x08 0x10 0x14 0x18 0x1A 0x1C 0x1E (v_perm byte select)
  const uint32_t code = __builtin_amdgcn_perm(0x1E1C1A18u, 0x14100800u, mag);
  return code | (m1 << 5);
}

// four 6-bit codes in the bytes of c -> 24 contiguous bits
__device__ __forceinline__ uint32_t squeeze6(uint32_t c) {
  return (c & 0x3Fu) | ((c >> 2) & 0xFC0u) | ((c >> 4) & 0x3F000u) | ((c >> 6) & 0xFC0000u);
}

// 32 int8 codes in k order (lo = k 0..15, hi = k 16..31) -> 32 fp6 values, value j
// at bits [6j, 6j + 6) of 6 dwords (the B fragment of one lane for one k-step)
__device__ __forceinline__ void fp6_pack32(const v4i &lo, const v4i &hi, int (&d)[6]) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t[i] = squeeze6(fp6_codes4((uint32_t)lo[i]));
    t[4 + i] = squeeze6(fp6_codes4((uint32_t)hi[i]));
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    d[3 * g + 0] = (int)(t[4 * g] | (t[4 * g + 1] << 24));
    d[3 * g + 1] = (int)((t[4 * g + 1] >> 8) | (t[4 * g + 2] << 16));
    d[3 * g + 2] = (int)((t[4 * g + 2] >> 16) | (t[4 * g + 3] << 8));
  }
}

#ifdef SNNQP_F6_TRACE
// Diagnostic build only: clock stamps of waves 0 and 4 of workgroup 0 inside one
// timestep (tools/f6_trace.py).  Never in the product.
__device__ unsigned long long snnqp_f6_trace[2][8];
extern "C" int snnqp_debug_read_f6_trace(unsigned long long *out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(snnqp_f6_trace), 128);
}
#define F6_MARK(i)                                                                  \
  if (trace_on && t == 10) snnqp_f6_trace[role][i] = __builtin_amdgcn_s_memtime();
#else
#define F6_MARK(i)
#endif

// FMT: the matrix instruction the contraction runs on
//   FMT_FP6: v_mfma_scale_f32_32x32x64_f8f6f4, fp4 spikes x fp6 codes (|code| <= 7), K = 64
//   FMT_I8 : v_mfma_i32_32x32x32_i8, byte spikes x int8 codes (any 8-bit code), K = 32
// Same workgroup shape, LDS image geometry (a plane = the 32 bytes a pixel contributes
// to one k-step), pipeline and epilogue; the formats differ in the B fragments, the
// bit -> operand expansion of the halo and the accumulator type.
enum { FMT_FP6 = 0, FMT_I8 = 1 };

template <int FMT, int CIN, int NF, bool POOL, bool LUT, bool FMA = false>
__global__ void __launch_bounds__(F6_NT, 2 / F6_TILES)
conv3x3_bits_kernel(ConvMfmaArgs a) {
  static_assert(CIN == 64 || CIN == 128, "one or two 64-channel planes");
  constexpr bool I8 = FMT == FMT_I8;
  typedef typename std::conditional<I8, v16i, v16f>::type acc_t;
  constexpr int KCH = I8 ? 32 : 64;              // input channels of one k-step
  constexpr int BR = I8 ? 4 : 6;                 // registers of one B fragment
  constexpr int NP = CIN / KCH;                  // planes of the halo image
  constexpr int WPP = CIN / 32;                  // spike words per pixel
  constexpr int F6_KS = 9 * NP;
  constexpr int F6_HALO = NP * F6_PLANE;         // one fp4 halo image
  constexpr int PPS = (32 + F6_KS - 1) / F6_KS;  // epilogue pieces per MFMA slot
  constexpr int FL = POOL ? 16 : 4;              // timesteps per flush block
  constexpr int SLOTS = 2 * FL;                  // ring of staged spike words
  constexpr int NPIX = OutStage<POOL>::NPIX * F6_TILES / 2;
  constexpr int TAB_OFF = 2 * F6_HALO;
  constexpr int LUT_OFF = TAB_OFF + F6_TAB;
  constexpr int LUT_BYTES = LUT ? (2 * LUT_CAP + 2) * 4 : 0;
  constexpr int LUT_ZERO = LUT_OFF + 4 * LUT_CAP;
  static_assert(LUT_ZERO < 65536, "the table's centre must be a ds_read immediate offset");
  constexpr int OB_OFF = LUT_OFF + LUT_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t lds[OB_OFF + SLOTS * NPIX * 16 + 16];
  uint32_t *nxt = (uint32_t *)(lds + OB_OFF + SLOTS * NPIX * 16);   // claimed patch (PatchWalk)
  uint32_t *obuf = (uint32_t *)(lds + OB_OFF);
  const uint32_t lds0 = lds_addr(lds) & 0x3FFFFu;   // < 2^18: offsets fold into immediates

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int role = F6_TILES == 2 ? wave >> 2 : 0;   // the tile (4x8 pixels) of this wave
  const int cg = wave & 3;
  const int cout_base = blockIdx.y * 128 + cg * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cpar = cout < a.Cout ? cout : a.Cout - 1;      // parameter loads
  const uint32_t cmask = chan_mask(cout_base, a.Cout);

  // tables: byte -> 8 nibbles (bit i set -> 1.0 = 0x2 in nibble i); dequant table
  if (!I8 && tid < 256) {                        // (int8: bits -> bytes by arithmetic)
    uint32_t v = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) v |= ((tid >> bit) & 1) ? (0x2u << (4 * bit)) : 0u;
    ((uint32_t *)(lds + TAB_OFF))[tid] = v;
  }
  // the entry of acc = 0 sits at the fixed byte LUT_ZERO whatever the launch's bound is,
  // so the table read takes the (signed) accumulator as its address register and
  // LUT_ZERO as the instruction's immediate offset
  if (LUT) build_lut((float *)(lds + LUT_OFF) + (LUT_CAP - a.lut_bound), a.lut_bound, a.dq, tid, F6_NT);

  // B operand: k-step ks = NP tap + kk covers channels 64 kk .. +63 of the tap; lane
  // (n, h) holds k = 32 h + j, i.e. both 16-byte halves of int8 tile WPP tap + 2 kk + h
  // (int8: k-step ks = WPP tap + kk is int8 tile ks as it is, lane (n, h) holds k = 16 h + j)
  int bf[F6_KS][BR];
  {
    const v4i *wtile = (const v4i *)a.wt + (int64_t)(cout_base >> 5) * (9 * WPP) * 64;
#pragma unroll
    for (int ks = 0; ks < F6_KS; ++ks) {
      if constexpr (I8) {
        const v4i t = wave_on ? wtile[ks * 64 + lane] : v4i{0, 0, 0, 0};
        bf[ks][0] = t.x; bf[ks][1] = t.y; bf[ks][2] = t.z; bf[ks][3] = t.w;
      } else {
        const int ks8 = (ks / NP) * WPP + (ks % NP) * 2 + h;
        v4i lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (wave_on) {
          lo = wtile[ks8 * 64 + n];
          hi = wtile[ks8 * 64 + 32 + n];
        }
        fp6_pack32(lo, hi, bf[ks]);
      }
    }
  }

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f, a.nrn.vr};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cpar]; lc.bmul = a.bn.mul[cpar]; lc.bbias = a.bn.bias[cpar]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cpar];

  // table mode: spikes count 4 (block scale 2^2 on A / spike bytes of 4), so the accumulator
  // is the byte offset of its table entry from the entry of acc = 0.  The chain starts from
  // the inline constant 0: no accumulator preload in either mode.
  constexpr int SCALE_A = LUT ? 129 : 127;       // E8M0: 2^(s - 127)

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // A fragment of tap (dy, dx), half kk: pixel (4 role + ty + dy, tx + dx), 32 B per
  // pixel, the two 16-byte lane halves swapped on odd halo rows.  With the 12-pixel
  // row pitch this makes every ds_read_b128 of the wave bank-conflict free
  // (tools/ubench/lds_conv_patterns.hip: 222 B/clk/CU against 63 for the plain layout).
  const uint32_t pixb = lds0 + (uint32_t)(((role * 4 + ty) * F6_PITCH + tx) * 32);
  const uint32_t abase_even = pixb + (uint32_t)((h ^ (ty & 1)) * 16);       // dy = 0, 2
  const uint32_t abase_odd = pixb + (uint32_t)((h ^ (ty & 1) ^ 1) * 16);    // dy = 1

  // staging task of this thread: word wi of halo pixel pix
  const int s_pix = tid / WPP, s_wi = tid % WPP;
  const bool s_task = tid < F6_ROWS * HALO * WPP;
  // a pixel has ceil(Cin / 32) spike words in memory; the planes beyond them (Cin below
  // the template's 64 / 128) are zero spikes against zero codes
  const int wpm = (a.Cin + 31) >> 5;
  const int s_hy = s_pix / HALO, s_hx = s_pix % HALO;
  // fp4: word wi = half wi & 1 of plane wi >> 1; bytes: word wi = both halves of plane wi
  uint8_t *s_dst = lds + (I8 ? s_wi : s_wi >> 1) * F6_PLANE + (s_hy * F6_PITCH + s_hx) * 32 +
                   ((((I8 ? 0 : s_wi) & 1) ^ (s_hy & 1)) * 16);
  const uint32_t tab0 = lds0 + TAB_OFF;
  const uint32_t *xb = (const uint32_t *)a.x;

  const int ob = out_pix<POOL>(role, lane) * 4 + cg;
  const bool store_lane = POOL ? lane < 8 : lane < 32;

  lds_barrier();                                 // tables are visible
#ifdef SNNQP_F6_TRACE
  const bool trace_on = blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && cg == 0;
#endif

  PatchWalk pw(a);
  int r = (int)pw.first;                 // patch indices fit 31 bits (launch check)
  while (r < (int)pw.count) {
    int r_next = r + (int)pw.stride;
    if (pw.queue && tid == 0) r_next = (int)pw.claim(); // next patch, a patch ahead
    int b, y0, x0;
    pw.decode(a, r, b, y0, x0);

    float u[16];
    if (a.u0 && wave_on) {
      u_io_tile<true>(u, a, b, y0, x0, cout, h, role);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) u[i] = 0.0f;
    }

    const int gy = y0 + s_hy - 1, gx = x0 + s_hx - 1;
    const bool s_valid = s_task && s_wi < wpm && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int64_t s_goff = (int64_t)b * a.xs_b + ((int64_t)gy * a.W + gx) * wpm + s_wi;
    // the spike word of halo(t) is requested two steps before it is expanded (a step is about a
    // microsecond, a loaded global round trip can be longer): two registers, by parity of t
    uint32_t stgq[2] = {0, 0}, stg_cur = 0;
    auto stage_load = [&](int t, int par) {
      stgq[par] = s_valid ? xb[(int64_t)t * a.xs_t + s_goff] : 0u;
    };
    v4i s_exp = {0, 0, 0, 0};
    auto stage_expand = [&](int par) {           // table reads; consumed by stage_write
      if (I8) {
        stg_cur = stgq[par];
      } else if (s_task) {
        const uint32_t sw = stgq[par];
        s_exp.x = (int)*(lds_cu32_t *)(uintptr_t)(tab0 + ((sw << 2) & 0x3FCu));
        s_exp.y = (int)*(lds_cu32_t *)(uintptr_t)(tab0 + ((sw >> 6) & 0x3FCu));
        s_exp.z = (int)*(lds_cu32_t *)(uintptr_t)(tab0 + ((sw >> 14) & 0x3FCu));
        s_exp.w = (int)*(lds_cu32_t *)(uintptr_t)(tab0 + ((sw >> 22) & 0x3FCu));
      }
    };
    auto stage_write = [&](int buf) {
      if (I8) {
        if (s_task) {
          uint8_t *d = s_dst + buf * F6_HALO;
          *(v4i *)d = expand16<LUT>(stg_cur & 0xFFFFu);                 // channels 0..15
          *(v4i *)(d + (s_hy & 1 ? -16 : 16)) = expand16<LUT>(stg_cur >> 16);
        }
      } else if (s_task) {
        *(v4i *)(s_dst + buf * F6_HALO) = s_exp;
      }
    };
    auto a_read = [&](int buf, int ks) -> v4i {
      const int tap = ks / NP;
      const uint32_t off = (uint32_t)(buf * F6_HALO + (ks % NP) * F6_PLANE +
                                      ((tap / 3) * F6_PITCH + tap % 3) * 32);
      return *(lds_cv4i_t *)(uintptr_t)(((tap / 3) & 1 ? abase_odd : abase_even) + off);
    };
    constexpr int PF = SNNQP_F6_PREFETCH;        // A fragments in flight ahead of the MFMA
    auto mfma_acc = [&](int ks, const v4i &av, const acc_t &c) -> acc_t {
      if constexpr (I8) {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(
            av, v4i{bf[ks][0], bf[ks][1], bf[ks][2], bf[ks][3]}, c, 0, 0, 0);
      } else {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
            v8i{av.x, av.y, av.z, av.w, 0, 0, 0, 0},
            v8i{bf[ks][0], bf[ks][1], bf[ks][2], bf[ks][3], bf[ks][4], bf[ks][5], 0, 0}, c,
            4 /* A: fp4 */, 2 /* B: fp6 */, 0, SCALE_A, 0, 127);
      }
    };
    auto mfma_one = [&](int ks, const v4i &av, acc_t &acc) { acc = mfma_acc(ks, av, acc); };
    // the first MFMA of a chain takes C = 0 (an inline constant of the instruction)
    auto mfma_first = [&](const v4i &av, acc_t &acc) {
      const acc_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc = mfma_acc(0, av, zero16);
    };
    // table entry of the accumulator value a4 = 4 acc: address register = (int)a4, which
    // is negative for negative sums; the LDS address adder wraps, so base + LUT_ZERO is
    // the entry (the compiler folds LUT_ZERO into the offset field: ds_read_b32 ... offset:)
    auto lut_read = [&](auto a4) -> float {       // float accumulators: one v_cvt_i32_f32
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 64)   // diagnostic build: the cvt without the table read
      return __int_as_float((int)a4);
#endif
      return *(lds_cfloat_t *)((lds_cu8_t *)lds + LUT_ZERO + (int)a4);
    };
    auto dequant2 = [&](auto a0, auto a1) -> v2f {
      if (LUT) return v2f{lut_read(a0), lut_read(a1)};
      const v2f af = {(float)a0, (float)a1};     // exact integers: the division of common.h
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 512)   // diagnostic build: one-multiply division
      return (af * a.dq.rL) * a.dq.m;
#endif
      const v2f q = fma2(af, v2f{a.dq.rL, a.dq.rL}, af * a.dq.rLlo);
      return q * a.dq.m;
    };
    // MFMA(0) of a patch: nothing to overlap with
    auto mfma_only = [&](int buf, acc_t &acc) {
      v4i A[PF + 1];
#pragma unroll
      for (int i = 0; i < PF; ++i) A[i] = a_read(buf, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < F6_KS; ++ks) {
        if (ks + PF < F6_KS) A[(ks + PF) % (PF + 1)] = a_read(buf, ks + PF);
        if (ks == 0) mfma_first(A[0], acc);
        else mfma_one(ks, A[ks % (PF + 1)], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // One pipelined step in a hand-placed order: 9 NP slots, each = one A read PF
    // k-steps ahead, one MFMA of step t + 1 and two quarter-pairs of the epilogue of
    // step t (8 pairs x 4 pieces = 32 pieces), fenced so the order survives.  An MFMA
    // that waits at the issue port (matrix pipe busy, or its accumulator not ready)
    // blocks VALU issue for every wave of the SIMD -- another wave's VALU work does
    // NOT overlap a back-to-back MFMA chain (tools/ubench/mfma_valu_2waves.hip: time =
    // sum) -- so each wave spaces its MFMAs with its own epilogue instructions.
    //   piece 0 (pair j + YD): dequantise (table reads or packed arithmetic), far enough
    //                          ahead that the LDS round trip is over when piece 1 wants it
    //   piece 1 (pair j)    : BatchNorm (3 packed ops)
    //   piece 2 (pair j)    : membrane update (3 packed ops) + threshold compares
    //   piece 3 (pair j)    : reset + spike word select
    auto fused_step = [&](int buf, acc_t &accN, const acc_t &accC, int t) {
      v4i A[PF + 1];
#pragma unroll
      for (int i = 0; i < PF; ++i) A[i] = a_read(buf, i);
      constexpr int YD = SNNQP_F6_YDIST;         // pairs the table reads run ahead
      v2f y[YD + 1], x = {0.f, 0.f}, uu = {0.f, 0.f};
      unsigned long long m0 = 0, m1 = 0;
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < YD; ++j) y[j] = dequant2(accC[2 * j], accC[2 * j + 1]);
      __builtin_amdgcn_sched_barrier(0);
      auto piece = [&](int p) {
        const int j = p >> 2, q = p & 3;
        if (q == 0) {
          if (j + YD < 8) y[(j + YD) % (YD + 1)] = dequant2(accC[2 * (j + YD)], accC[2 * (j + YD) + 1]);
        } else if (q == 1) {
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 128)   // diagnostic build: BatchNorm = the multiply only
          x = y[j % (YD + 1)] * lc.bmul;
#else
          x = y[j % (YD + 1)] - lc.bmean;
          x = x * lc.bmul;
          x = x + lc.bbias;
#endif
        } else if (q == 2) {       // membrane update of the neuron form (conv_tile.h)
          uu = neuron_update<NF, FMA, false>(x, v2f{u[2 * j], u[2 * j + 1]}, lc, a.nrn);
          m0 = __ballot(uu.x >= a.nrn.vth);
          m1 = __ballot(uu.y >= a.nrn.vth);
        } else {
          u[2 * j] = neuron_reset<NF>(uu.x, m0, lc);
          u[2 * j + 1] = neuron_reset<NF>(uu.y, m1, lc);
          const int i = 2 * j;
          if (POOL) {
            const unsigned long long o = m0 | m1;
            w = writelane_u32((uint32_t)o | (uint32_t)(o >> 32), i >> 1, w);
          } else {
            const int r0 = (i & 3) + 8 * (i >> 2);
            w = writelane_u32((uint32_t)m0, r0, w);
            w = writelane_u32((uint32_t)(m0 >> 32), r0 + 4, w);
            w = writelane_u32((uint32_t)m1, r0 + 1, w);
            w = writelane_u32((uint32_t)(m1 >> 32), r0 + 5, w);
          }
        }
      };
#pragma unroll
      for (int ks = 0; ks < F6_KS; ++ks) {
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 32)   // diagnostic build: half of the A reads
        if (ks + PF < F6_KS) {
          if ((ks & 1) == 0) A[(ks + PF) % (PF + 1)] = a_read(buf, ks + PF);
          else A[(ks + PF) % (PF + 1)] = A[(ks + PF - 1) % (PF + 1)];
        }
#else
        if (ks + PF < F6_KS) A[(ks + PF) % (PF + 1)] = a_read(buf, ks + PF);
#endif
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 1)   // diagnostic build: 2 of 18 MFMAs
        if (ks < 2)
#endif
        if (ks == 0) mfma_first(A[0], accN);
        else mfma_one(ks, A[ks % (PF + 1)], accN);
#if !(defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 2))   // diagnostic build: no epilogue
#pragma unroll
        for (int q = 0; q < PPS; ++q)
          if (ks * PPS + q < 32) piece(ks * PPS + q);
#endif
        // pin the slot: everything it produced is an operand of an (empty) volatile
        // asm, so neither the MFMA nor the pieces can drift to another slot (the
        // scheduling fence alone does not stop earlier passes from clustering the
        // MFMAs at the end of the step)
        // (the table reads stay free: tying them would wait out an LDS round trip)
        asm volatile("" : "+v"(accN), "+v"(x), "+v"(uu), "+v"(w), "+s"(m0), "+s"(m1));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (store_lane) obuf[(t % SLOTS) * (NPIX * 4) + ob] = w & cmask;
    };
    auto epilogue = [&](const acc_t &acc, int t) {
      v2f y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = dequant2(acc[2 * j], acc[2 * j + 1]);
      const uint32_t w = tile_neurons<NF, POOL, false, FMA>(y, u, lc, a.nrn);
      if (store_lane) obuf[(t % SLOTS) * (NPIX * 4) + ob] = w & cmask;
    };
    // staging of halo(t2) into its buffer around a step: the table reads go first, the
    // word of halo(t2 + 1) is requested as soon as its register is free (one step ahead is
    // enough: two steps ahead measured the same), the LDS
    // write comes after the step's MFMAs
    auto stage_begin = [&](int t2, int par) {    // par = t2 & 1, a constant at every call
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 4)   // diagnostic build: no halo staging
      return;
#endif
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 1024)   // diagnostic build: halo staged for t = 0, 1 only
      if (t2 >= 2) return;
#endif
      if (t2 < a.T) stage_expand(par);
      if (t2 + 2 < a.T) stage_load(t2 + 2, par);
    };
    auto stage_end = [&](int t2) {
#if defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 1024)
      if (t2 >= 2) return;
#endif
      if (t2 < a.T) stage_write(t2 & 1);
    };
    // one pipeline step: MFMA(t + 1) || epilogue(t); halo(t + 2) staged meanwhile.
    // One barrier per step: inside it every wave reads halo(t + 1) and writes
    // halo(t + 2) into the other buffer.
    auto step = [&](int t, int par, acc_t &accN, const acc_t &accC) {
#if !(defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 8))   // diagnostic build: no flush
      if (t >= FL && t % FL == 0)                // steps < t are behind a barrier
        flush_ring<POOL, SLOTS, F6_NT, NPIX>(obuf, a, t - FL, FL, b, y0, x0, tid);
#endif
      F6_MARK(0)
      stage_begin(t + 2, par);
      F6_MARK(1)
      fused_step((t + 1) & 1, accN, accC, t);
      F6_MARK(2)
      stage_end(t + 2);
      F6_MARK(3)
#if !(defined(SNNQP_F6_ABL) && (SNNQP_F6_ABL & 256))   // diagnostic build: no per-step barrier (races)
      lds_barrier();
#endif
      F6_MARK(4)
    };

    // pipeline prologue: halo(0) staged; MFMA(0) while halo(1) is staged
    acc_t accA, accB;
    stage_load(0, 0);
    if (a.T > 1) stage_load(1, 1);
    stage_begin(0, 0);
    stage_end(0);
    lds_barrier();
    stage_begin(1, 1);
    mfma_only(0, accA);
    stage_end(1);
    lds_barrier();

    // steady state, unrolled by two so the accumulator roles alternate
    int t = 0;
    for (; t + 2 < a.T; t += 2) {
      step(t, 0, accB, accA);
      step(t + 1, 1, accA, accB);
    }
    if (t + 1 < a.T) {
      step(t, 0, accB, accA);
      epilogue(accB, t + 1);
    } else {
      epilogue(accA, t);
    }
    lds_barrier();
    {
      const int done = a.T >= 2 ? ((a.T - 2) / FL) * FL : 0;
      flush_ring<POOL, SLOTS, F6_NT, NPIX>(obuf, a, done, a.T - done, b, y0, x0, tid);
    }
    if (a.u_out && wave_on) u_io_tile<false>(u, a, b, y0, x0, cout, h, role);
    if (pw.queue) {                      // the claimed patch, to the whole workgroup
      if (tid == 0) nxt[0] = (uint32_t)r_next;
      lds_barrier();
      r_next = __builtin_amdgcn_readfirstlane((int)nxt[0]);
    }
    r = r_next;
  }
  if (pw.queue && tid == 0) pw.finish();
}

template <int FMT, int CIN, int NF>
static void launch_fp6_nf(const ConvMfmaArgs &a, bool pool, bool lut, unsigned gy,
                          hipStream_t st) {
  if (pool && lut) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, true, true>, a, gy, st, 0, F6_NT);
  else if (pool) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, true, false>, a, gy, st, 0, F6_NT);
  else if (lut) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, false, true>, a, gy, st, 0, F6_NT);
  else launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, false, false>, a, gy, st, 0, F6_NT);
}

template <int FMT, int CIN>
static void launch_fp6_cin(const ConvMfmaArgs &a, int nf, bool pool, bool lut, unsigned gy,
                           hipStream_t st) {
  if (nf == NF_MUL0) launch_fp6_nf<FMT, CIN, NF_MUL0>(a, pool, lut, gy, st);
  else if (nf == NF_MUL) launch_fp6_nf<FMT, CIN, NF_MUL>(a, pool, lut, gy, st);
  else if (nf == NF_DIV) launch_fp6_nf<FMT, CIN, NF_DIV>(a, pool, lut, gy, st);
  else launch_fp6_nf<FMT, CIN, NF_DECAY>(a, pool, lut, gy, st);
}

// i8: codes wider than fp6 holds (|code| > 7) -> the int8 instruction
// NF_MUL0 with the fused membrane update
template <int FMT, int CIN>
static void launch_fp6_fma(const ConvMfmaArgs &a, bool pool, bool lut, unsigned gy, hipStream_t st) {
  if (pool && lut) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF_MUL0, true, true, true>, a, gy, st, 0, F6_NT);
  else if (pool) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF_MUL0, true, false, true>, a, gy, st, 0, F6_NT);
  else if (lut) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF_MUL0, false, true, true>, a, gy, st, 0, F6_NT);
  else launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF_MUL0, false, false, true>, a, gy, st, 0, F6_NT);
}

void launch_conv3x3_bits(const ConvMfmaArgs &a0, bool i8, int nf, bool pool, bool lut, bool fma,
                         unsigned gy, hipStream_t st) {
  ConvMfmaArgs a = a0;                       // this kernel's patch: F6_TILES tiles of 4x8 pixels
  a.patch_h = 4 * F6_TILES;
  a.tiles_y = (a.H + a.patch_h - 1) / a.patch_h;
  a.npatch = (int64_t)a.B * a.tiles_y * a.tiles_x;
  if (fma && nf == NF_MUL0) {
    if (i8) {
      if (a.Cin <= 64) launch_fp6_fma<FMT_I8, 64>(a, pool, lut, gy, st);
      else launch_fp6_fma<FMT_I8, 128>(a, pool, lut, gy, st);
    } else {
      if (a.Cin <= 64) launch_fp6_fma<FMT_FP6, 64>(a, pool, lut, gy, st);
      else launch_fp6_fma<FMT_FP6, 128>(a, pool, lut, gy, st);
    }
    return;
  }
  if (i8) {
    if (a.Cin <= 64) launch_fp6_cin<FMT_I8, 64>(a, nf, pool, lut, gy, st);
    else launch_fp6_cin<FMT_I8, 128>(a, nf, pool, lut, gy, st);
  } else {
    if (a.Cin <= 64) launch_fp6_cin<FMT_FP6, 64>(a, nf, pool, lut, gy, st);
    else launch_fp6_cin<FMT_FP6, 128>(a, nf, pool, lut, gy, st);
  }
}

}  // namespace snnqp
