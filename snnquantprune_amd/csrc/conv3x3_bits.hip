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
// accumulator sets.  The loop is software-pipelined over t inside each wave: step s runs
// the MFMAs of timestep s + 1 interleaved, slot by slot, with the neuron epilogue of
// timestep s (one slot = one MFMA, one A-fragment read a few slots ahead, an even share of
// the epilogue's VALU instructions).
//
// Halo images (10 x 6 pixels x Cin spike bits, expanded to the A format: fp4 through a
// byte -> 8-nibble LDS table, bytes by arithmetic) live in a ring of THREE LDS buffers:
// during step s the waves read halo(s + 1), write halo(s + 2) early in the step and meet at
// ONE barrier a few slots later.  Nothing is read across that barrier that was written after
// the previous one, so the A fragments of step s + 1 are requested during the last slots of
// step s and the matrix pipe does not drain at a step boundary (with two buffers and the
// barrier at the end of the step every step began with an LDS round trip).
// LDS image: one plane per k-step of a tap (64 fp4 / 32 byte channels = 32 B per pixel),
// rows of 12 pixels, the two 16-byte lane halves swapped on odd rows: every tap/half offset
// is an instruction immediate and the reads are bank-conflict free.
#include <type_traits>

#include "conv_tile.h"

namespace snnqp {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const v4i lds_cv4i_t;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;

constexpr int F6_PITCH = HPITCH;             // pixels per LDS halo row (10 used)
constexpr int F6_ROWS = 6;                   // halo rows of a 4x8 patch
constexpr int F6_NT = 256;                   // threads of a workgroup: 4 waves
constexpr int F6_PLANE = F6_ROWS * HPITCH * 32;   // one k-step plane of a halo image
constexpr int F6_NBUF = 3;                   // halo images in the ring
// byte -> 8 fp4 nibbles, 32 interleaved copies: entry e of copy c at dword 32 e + c, so lane
// l of a 32-lane group reads bank l whatever its byte is (ds_read_b32 banks: (a / 4) % 32)
constexpr int F6_TAB = 256 * 32 * 4;
// (DQT_MAXA = 2047, conv_tile.h: the |acc| bound DQ_TABLE's table is sized for)
constexpr int DQT_BYTES = 16384;             // 4095 entries

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

// FMT: the matrix instruction the contraction runs on
//   FMT_FP6: v_mfma_scale_f32_32x32x64_f8f6f4, fp4 spikes x fp6 codes (|code| <= 7), K = 64
//   FMT_I8 : v_mfma_i32_32x32x32_i8, byte spikes x int8 codes (any 8-bit code), K = 32
// DQ (conv_tile.h): how the accumulator becomes the current fl(fl(acc / L) * m)
//   DQ_ARITH: three float32 instructions (the exact two-instruction division of common.h)
//   DQ_ONE  : L == 1 (2-bit DuQ codes, the step quantisers): one multiply
//   DQ_TABLE: (FMT_FP6, |acc| <= DQT_MAXA) no vector instruction at all.  The block scales of
//             the matrix instruction are chosen so that one code unit is 2^-147 = four steps of
//             the float32 DENORMAL range, and the chain starts from the constant whose bit
//             pattern is (LDS address of the table's middle entry): the accumulator's bit pattern
//             IS the byte address of fl(fl(acc / L) * m) in a table the workgroup built with the
//             DQ_ARITH instructions -- one ds_read_b32 per value, addressed by the accumulator
//             register itself.  The matrix pipe adds denormals exactly
//             (tools/ubench/mfma_denorm.hip: bit patterns equal to the integer sums over the
//             whole range, at the speed of the normal range); partial sums never leave
//             [-A, A], so the pattern never goes negative.  The chain constant sits in sixteen
//             registers of its own (the C operand of every chain's first MFMA); the A-fragment
//             ring is 3 deep in this form to make room for them (re-reading the constants into
//             the consumed accumulator registers, four ds_read_b128 per timestep, with the ring
//             of 6: 1 % slower).
//   (round 1's table form -- v_cvt + shift + ds_read_b32 on an int32 accumulator -- saved one
//   instruction per value; this one saves three of the epilogue's 8.6)
// BNF: every BatchNorm mean and bias of the launch is zero (snnqp_bn_t.flags): x = y * mul
// BNU: ... and every channel has the same multiplier (SNNQP_BN_MUL_UNIFORM; DQ_TABLE only): the
//      table entry is fl(current * mul) -- the same float32 product, once per entry instead of once
//      per neuron update -- and the epilogue has no BatchNorm instruction left
enum { FMT_FP6 = 0, FMT_I8 = 1 };

constexpr int F6_WR_SLOT = 2;    // slot of a step after which the staged halo is written
constexpr int F6_BAR_SLOT = 4;   // slot of a step after which the step's barrier sits
constexpr int F6_PF = 4;         // A fragments in flight (ring of 6); 2 when a step has 9 slots

template <int FMT, int CIN, int NF, bool POOL, int DQ, bool FMA = false, bool BNF = false, bool BNU = false>
__global__ void __launch_bounds__(F6_NT, 2)
conv3x3_bits_kernel(ConvMfmaArgs a) {
  static_assert(CIN == 64 || CIN == 128, "one or two 64-channel planes");
  static_assert(!BNU || (BNF && DQ == DQ_TABLE), "the uniform multiplier folds into the shared table");
  static_assert(DQ == DQ_ARITH || DQ == DQ_ONE || DQ == DQ_TABLE, "dequantisation mode");
  constexpr bool I8 = FMT == FMT_I8;
  constexpr bool TABLE = DQ == DQ_TABLE;
  static_assert(!(TABLE && I8), "the table is addressed by a float32 accumulator");
  typedef typename std::conditional<I8, v16i, v16f>::type acc_t;
  constexpr int KCH = I8 ? 32 : 64;              // input channels of one k-step
  constexpr int BR = I8 ? 4 : 6;                 // registers of one B fragment
  constexpr int NP = CIN / KCH;                  // planes of the halo image
  constexpr int WPP = CIN / 32;                  // spike words per pixel
  constexpr int KS = 9 * NP;                     // MFMAs (slots) of one timestep
  constexpr int HALO_B = NP * F6_PLANE;          // one halo image
  constexpr int FL = POOL ? 16 : 4;              // timesteps per flush block
  constexpr int SLOTS = 2 * FL;                  // ring of staged spike words
  constexpr int NPIX = OutStage<POOL>::NPIX / 2;
  constexpr int TAB_OFF = F6_NBUF * HALO_B;
  constexpr int OB_OFF = TAB_OFF + (I8 ? 0 : F6_TAB);
  constexpr int DQT_OFF = OB_OFF + SLOTS * NPIX * 16 + 16;
  __shared__ __attribute__((aligned(128))) uint8_t lds[DQT_OFF + (TABLE ? DQT_BYTES : 0)];
  uint32_t *nxt = (uint32_t *)(lds + OB_OFF + SLOTS * NPIX * 16);   // claimed patch (PatchWalk)
  uint32_t *obuf = (uint32_t *)(lds + OB_OFF);
  const uint32_t lds0 = lds_addr(lds) & 0x3FFFFu;   // < 2^18: offsets fold into immediates

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int cg = wave & 3;
  const int cout_base = blockIdx.y * 128 + cg * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cpar = cout < a.Cout ? cout : a.Cout - 1;      // parameter loads
  const uint32_t cmask = chan_mask(cout_base, a.Cout);

  // table: byte -> 8 nibbles (bit i set -> 1.0 = 0x2 in nibble i), 32 copies
  if (!I8) {                                     // (int8: bits -> bytes by arithmetic)
    for (int i = tid; i < 256 * 32; i += F6_NT) {
      const int e = i >> 5;
      uint32_t v = 0;
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) v |= ((e >> bit) & 1) ? (0x2u << (4 * bit)) : 0u;
      ((uint32_t *)(lds + TAB_OFF))[i] = v;
    }
  }
  // B operand: k-step ks = NP tap + kk covers channels 64 kk .. +63 of the tap; lane
  // (n, h) holds k = 32 h + j, i.e. both 16-byte halves of int8 tile WPP tap + 2 kk + h
  // (int8: k-step ks = WPP tap + kk is int8 tile ks as it is, lane (n, h) holds k = 16 h + j)
  int bf[KS][BR];
  {
    const v4i *wtile = (const v4i *)a.wt + (int64_t)(cout_base >> 5) * (9 * WPP) * 64;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
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

  // DQ_TABLE: entry i of the table is the current of acc = i - A (A = a.lut_bound), computed
  // with the instructions DQ_ARITH would run on the accumulator; the chain constant is the
  // address of entry A as a bit pattern
  const uint32_t dqt_mid = lds0 + (uint32_t)DQT_OFF + 4u * (uint32_t)(TABLE ? a.lut_bound : 0);
  const float chain0 = TABLE ? __uint_as_float(dqt_mid) : 0.0f;
  if (TABLE) {
    const float mul0 = (BNU && a.bn.mul) ? a.bn.mul[0] : 1.0f;      // BNU: the same for every channel
    for (int i = tid; i <= 2 * a.lut_bound; i += F6_NT) {
      const float af = (float)(i - a.lut_bound);
      const float q = __builtin_fmaf(af, a.dq.rL, af * a.dq.rLlo);   // exact af / L (common.h)
      const float cur = q * a.dq.m;
      ((float *)(lds + DQT_OFF))[i] = BNU ? cur * mul0 : cur;
    }
  }

  LaneConsts lc = {0.f, 1.f, 0.f, 0.f, a.nrn.vr};
  if (a.bn.mean) { lc.bmean = a.bn.mean[cpar]; lc.bmul = a.bn.mul[cpar]; lc.bbias = a.bn.bias[cpar]; }
  if (a.nrn.kind == SNNQP_NEURON_LIF) lc.dec = a.nrn.decay[cpar];

  // E8M0 block scales 2^(s - 127).  DQ_TABLE: 2^-63 * 2^-84 = 2^-147 per (spike x code unit);
  // both are inline constants.  Otherwise 1: the chain starts from the inline constant 0.
  constexpr int SCALE_A = TABLE ? 64 : 127;
  constexpr int SCALE_B = TABLE ? 43 : 127;

  const int ty = ((n >> 2) & 1) | ((n >> 4) << 1);
  const int tx = (n & 3) | (((n >> 3) & 1) << 2);
  // A fragment of tap (dy, dx), half kk: pixel (ty + dy, tx + dx), 32 B per pixel, the two
  // 16-byte lane halves swapped on odd halo rows.  With the 12-pixel row pitch this makes
  // every ds_read_b128 of the wave bank-conflict free (tools/ubench/lds_conv_patterns.hip).
  const uint32_t pixb = lds0 + (uint32_t)((ty * F6_PITCH + tx) * 32);
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
  const uint32_t s_dst = lds0 + (uint32_t)((I8 ? s_wi : s_wi >> 1) * F6_PLANE +
                                           (s_hy * F6_PITCH + s_hx) * 32 +
                                           ((((I8 ? 0 : s_wi) & 1) ^ (s_hy & 1)) * 16));
  const uint32_t tabl = lds0 + TAB_OFF + (uint32_t)(lane & 31) * 4;   // this lane's table copy
  const uint32_t *xb = (const uint32_t *)a.x;

  const int ob = out_pix<POOL>(0, lane) * 4 + cg;
  const bool store_lane = POOL ? lane < 8 : lane < 32;

  lds_barrier();                                 // tables are visible

  // DQ_TABLE: where every chain starts -- sixteen registers for the whole launch
  acc_t cblk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (TABLE) {
#pragma unroll
    for (int i = 0; i < 16; ++i) cblk[i] = chain0;
    asm volatile("" : "+v"(cblk));               // not a constant per use
  }
  PatchWalk pw(a);
  // patch indices (+ one grid stride) stay below 2^31 (run_conv3x3_mfma refuses launches of
  // 2^30 patches or more) and are workgroup-uniform: scalar registers
  int r = __builtin_amdgcn_readfirstlane((int)pw.first);
  uint32_t processed = 0;                       // patches of this workgroup (scalar)
  while (r < (int)pw.count) {
    ++processed;
    int claimed = 0;
    if (pw.queue && tid == 0) claimed = (int)pw.claim();   // next patch, a patch ahead
    int b, y0, x0;
    pw.decode(a, r, b, y0, x0);

    float u[16];
    if (a.u0 && wave_on) {
      u_io_tile<true>(u, a, b, y0, x0, cout, h, 0);
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
        // byte k of the word -> entry (byte << 7) of this lane's copy: conflict-free whatever
        // the bytes of the 32 lanes are
        s_exp.x = (int)*(lds_cu32_t *)(uintptr_t)(((sw & 0xFFu) << 7) + tabl);
        s_exp.y = (int)*(lds_cu32_t *)(uintptr_t)((((sw >> 8) & 0xFFu) << 7) + tabl);
        s_exp.z = (int)*(lds_cu32_t *)(uintptr_t)((((sw >> 16) & 0xFFu) << 7) + tabl);
        s_exp.w = (int)*(lds_cu32_t *)(uintptr_t)(((sw >> 24) << 7) + tabl);
      }
    };
    auto stage_write = [&](uint32_t bufoff) {
      typedef __attribute__((address_space(3))) v4i lds_v4i_t;
      if (I8) {
        if (s_task) {
          const uint32_t d = s_dst + bufoff;
          *(lds_v4i_t *)(uintptr_t)d = expand16<false>(stg_cur & 0xFFFFu);        // channels 0..15
          *(lds_v4i_t *)(uintptr_t)(s_hy & 1 ? d - 16 : d + 16) = expand16<false>(stg_cur >> 16);
        }
      } else if (s_task) {
        *(lds_v4i_t *)(uintptr_t)(s_dst + bufoff) = s_exp;
      }
    };
    // A fragment of k-step ks from the halo image whose lane bases are (aeven, aodd)
    auto a_read = [&](uint32_t aeven, uint32_t aodd, int ks) -> v4i {
      const int tap = ks / NP;
      const uint32_t off = (uint32_t)((ks % NP) * F6_PLANE + ((tap / 3) * F6_PITCH + tap % 3) * 32);
      return *(lds_cv4i_t *)(uintptr_t)(((tap / 3) & 1 ? aodd : aeven) + off);
    };
    auto mfma_acc = [&](int ks, const v4i &av, const acc_t &c) -> acc_t {
      if constexpr (I8) {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(
            av, v4i{bf[ks][0], bf[ks][1], bf[ks][2], bf[ks][3]}, c, 0, 0, 0);
      } else {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
            v8i{av.x, av.y, av.z, av.w, 0, 0, 0, 0},
            v8i{bf[ks][0], bf[ks][1], bf[ks][2], bf[ks][3], bf[ks][4], bf[ks][5], 0, 0}, c,
            4 /* A: fp4 */, 2 /* B: fp6 */, 0, SCALE_A, 0, SCALE_B);
      }
    };
    const acc_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto dequant1 = [&](auto a0) -> float {
      if constexpr (TABLE) return *(lds_cfloat_t *)(uintptr_t)__float_as_uint((float)a0);
      const float af = (float)a0;                // exact integer
      if (DQ == DQ_ONE) return af * a.dq.m;      // L == 1
      const float q = __builtin_fmaf(af, a.dq.rL, af * a.dq.rLlo);   // exact af / L (common.h)
      return q * a.dq.m;
    };

    // ---- the neuron epilogue of one timestep as a list of small stages -----------------
    // pair j = accumulator registers 2j, 2j + 1 (two pixels of this lane's channel):
    //   stage 0  dequantise
    //   stage 1  BatchNorm sub, mul        stage 2  BatchNorm add, membrane update
    //            (BNF: the multiply only)
    //   stage 3  threshold                 stage 4  reset + spike word
    // Executed strictly in order; the slots of a pipelined step take an even share each.
    constexpr int NST = 5;
    constexpr int YD = TABLE ? 1 : 0;            // pairs the dequantisation runs ahead (the
                                                 // table reads: an LDS round trip)
    // GP pairs advance through the stages together (stage q of pair 2g, stage q of pair 2g + 1,
    // stage q + 1 of pair 2g, ...).  Alone, the table-mode epilogue takes 320 SIMD cycles per
    // tile-step pair after pair and 287 two pairs at a time (two waves per SIMD: a stage right
    // behind the one it depends on waits for its result; tools/ubench/threshold_forms.hip);
    // among the MFMAs and the other wave's instructions the kernel measures the same either
    // way (conv1 5.30 / 5.30 ms, 8-bit codes 8.63 / 8.64, random BatchNorm 5.58 / 5.57), so
    // pairs go one at a time: four temporaries fewer
    constexpr int GP = 1;
    constexpr int EYN = YD + GP;                 // currents in flight (ring over pairs)
    auto st_j = [](int st) { return (st / (NST * GP)) * GP + (st % (NST * GP)) % GP; };
    auto st_q = [](int st) { return (st % (NST * GP)) / GP; };
    // (the temporaries live inside a step: masks that crossed the loop's back edge would
    // leave the scalar registers)
    // The spike word: the compare masks are wave-uniform 64-bit lane masks (lane = channel).
    // Combining them is scalar work (s_or) that feeds a v_writelane; issued right behind the
    // v_cmp that produced the masks, each of those hops stalls the wave (VALU -> SALU -> VALU
    // round trips: 0.5 of 4.3 ms measured), so the masks of pair j are combined during pair
    // j + 1 and written into the word during pair j + 2.
    struct ETmp {
      float ey[EYN][2], ex[GP][2], euu[GP][2];
      unsigned long long m0[8], m1[8];
      uint32_t pw[8];                            // pooled word of a pair (scalar)
      uint32_t ew;
    };
    auto word_combine = [&](ETmp &e, int j) {    // scalar: masks of pair j -> its pooled word
      if (POOL) {
        const unsigned long long o = e.m0[j] | e.m1[j];
        e.pw[j] = (uint32_t)o | (uint32_t)(o >> 32);
      }
    };
    auto word_insert = [&](ETmp &e, int j) {     // v_writelane: pair j into the stored word
      const int i = 2 * j;
      if (POOL) {
        e.ew = writelane_u32(e.pw[j], i >> 1, e.ew);
      } else {
        const int r0 = (i & 3) + 8 * (i >> 2);
        e.ew = writelane_u32((uint32_t)e.m0[j], r0, e.ew);
        e.ew = writelane_u32((uint32_t)(e.m0[j] >> 32), r0 + 4, e.ew);
        e.ew = writelane_u32((uint32_t)e.m1[j], r0 + 1, e.ew);
        e.ew = writelane_u32((uint32_t)(e.m1[j] >> 32), r0 + 5, e.ew);
      }
    };
    auto estage = [&](acc_t &accC, int st, ETmp &e) {
      const int j = st_j(st), q = st_q(st), jg = j % GP;
      if (q == 0) {
        if (j + YD < 8) {
          e.ey[(j + YD) % EYN][0] = dequant1(accC[2 * (j + YD)]);
          e.ey[(j + YD) % EYN][1] = dequant1(accC[2 * (j + YD) + 1]);
        }
      } else if (q == 1) {
        if (BNU) {                   // the table entry is fl(y * mul) already
          e.ex[jg][0] = e.ey[j % EYN][0];
          e.ex[jg][1] = e.ey[j % EYN][1];
        } else if (BNF) {            // mean == 0: fl(y - 0) = y
          e.ex[jg][0] = e.ey[j % EYN][0] * lc.bmul;
          e.ex[jg][1] = e.ey[j % EYN][1] * lc.bmul;
        } else {
          e.ex[jg][0] = (e.ey[j % EYN][0] - lc.bmean) * lc.bmul;
          e.ex[jg][1] = (e.ey[j % EYN][1] - lc.bmean) * lc.bmul;
        }
      } else if (q == 2) {                // (bias == 0: fl(x + 0) = x)
        const v2f xx = BNF ? v2f{e.ex[jg][0], e.ex[jg][1]}
                           : v2f{e.ex[jg][0] + lc.bbias, e.ex[jg][1] + lc.bbias};
        const v2f uu = neuron_update<NF, FMA, false>(xx, v2f{u[2 * j], u[2 * j + 1]}, lc, a.nrn);
        e.euu[jg][0] = uu.x; e.euu[jg][1] = uu.y;
      } else if (q == 3) {
        e.m0[j] = __ballot(e.euu[jg][0] >= a.nrn.vth);
        e.m1[j] = __ballot(e.euu[jg][1] >= a.nrn.vth);
      } else {
        u[2 * j] = neuron_reset<NF>(e.euu[jg][0], e.m0[j], lc);
        u[2 * j + 1] = neuron_reset<NF>(e.euu[jg][1], e.m1[j], lc);
        if (j >= 1) word_combine(e, j - 1);
        if (j >= 2) word_insert(e, j - 2);
        if (j == 7) {                     // drain
          word_insert(e, 6);
          word_combine(e, 7);
          word_insert(e, 7);
        }
      }
    };
    auto estage_head = [&](acc_t &accC, ETmp &e) {   // the dequantisations that run ahead
      e.ew = 0;
#pragma unroll
      for (int g = 0; g < GP; ++g) e.ex[g][0] = e.ex[g][1] = e.euu[g][0] = e.euu[g][1] = 0.f;
#pragma unroll
      for (int j = 0; j < EYN; ++j) e.ey[j][0] = e.ey[j][1] = 0.f;
#pragma unroll
      for (int j = 0; j < YD; ++j) {
        e.ey[j][0] = dequant1(accC[2 * j]);
        e.ey[j][1] = dequant1(accC[2 * j + 1]);
      }
    };
    constexpr int NSTAGES = 8 * NST;             // 40 stages per timestep
    auto store_word = [&](int t, uint32_t ew) {
      if (store_lane) obuf[(t % SLOTS) * (NPIX * 4) + ob] = ew & cmask;
    };

    // ---- A-fragment ring: PF fragments in flight, across step boundaries ----------------
    constexpr int RING = KS % 6 == 0 && !TABLE ? 6 : 3;    // divides KS (9, 18, 36): static indices
    constexpr int PF = RING == 6 ? F6_PF : 2;
    static_assert(KS % RING == 0 && PF < RING, "ring indices must repeat every step");
    v4i A[RING];

    // One pipelined step s: MFMA(s + 1) from the image (rd_e, rd_o) while the epilogue of
    // timestep s runs on accC; halo(s + 2) is written early, the barrier follows two slots
    // later; the last PF slots request the first fragments of the NEXT step from (rn_e, rn_o).
    // The table form and the int8 instruction want both late (write after 5/9 of the slots,
    // barrier after 5/6, right before the next step's first fragments are requested): conv1
    // 5.19 ms against 5.28 with the early pair (5.28 again with the write one slot before the
    // barrier), 8-bit codes 8.70 against 8.89; the fp6 kernel with arithmetic dequantisation
    // keeps the early pair (2-bit layer of C5: 1.50 against 1.53 ms late).
    constexpr bool LATE = TABLE || I8;
    constexpr int BAR_LATE = KS * 5 / 6 < KS - PF ? KS * 5 / 6 : KS - PF - 1;
    constexpr int WR_SLOT = LATE ? KS * 5 / 9 : F6_WR_SLOT < KS - PF - 2 ? F6_WR_SLOT : 1;
    constexpr int BAR_SLOT = LATE ? BAR_LATE : F6_BAR_SLOT < KS - PF ? F6_BAR_SLOT : WR_SLOT + 1;
    static_assert(WR_SLOT < BAR_SLOT && BAR_SLOT < KS - PF,
                  "the halo is written before the barrier, the next step's fragments read after it");
    // LDS instructions the epilogue stages of slots WR_SLOT + 1 .. BAR_SLOT issue (DQ_TABLE)
    constexpr int TAB_OPS = [] {
      int n = 0;
      if (TABLE)
        for (int st = (WR_SLOT + 1) * (8 * NST) / KS; st < (BAR_SLOT + 1) * (8 * NST) / KS; ++st)
          if ((st % (NST * GP)) / GP == 0) {       // stage 0 of pair j (st_q, st_j)
            const int j = (st / (NST * GP)) * GP + (st % (NST * GP)) % GP;
            if (j + YD < 8) n += 2;
          }
      return n;
    }();
    auto fused_step = [&](acc_t &accN, acc_t &accC, int s, uint32_t rd_e, uint32_t rd_o,
                          uint32_t rn_e, uint32_t rn_o, uint32_t wr_off, int par, bool more) {
      if (s + 2 < a.T) stage_expand(par);
      if (s + 4 < a.T) stage_load(s + 4, par);
      ETmp e;
      estage_head(accC, e);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks == 0) accN = mfma_acc(0, A[0], TABLE ? cblk : zero16);   // C = inline 0 (table:
        else accN = mfma_acc(ks, A[ks % RING], accN);                   // the chain constants)
        __builtin_amdgcn_sched_barrier(0);
        // the slot's A read: PF slots ahead, wrapping into the next step's image
        if (ks + PF < KS) A[(ks + PF) % RING] = a_read(rd_e, rd_o, ks + PF);
        else if (more) A[(ks + PF) % RING] = a_read(rn_e, rn_o, ks + PF - KS);
        // an even share of the epilogue stages
        {
          const int lo = ks * NSTAGES / KS, hi = (ks + 1) * NSTAGES / KS;
#pragma unroll
          for (int st = lo; st < hi; ++st) estage(accC, st, e);
        }
        if (ks == WR_SLOT && s + 2 < a.T) stage_write(wr_off);
        if (ks == BAR_SLOT) {
          // The barrier publishes the halo image written in slot WR_SLOT (and the spike word
          // stored at the end of the previous step).  LDS operations of a wave complete in
          // order, and at least BAR_SLOT - WR_SLOT of them (the A reads of the slots in
          // between) were issued after that write: waiting until so many are outstanding
          // retires the write without draining the A fragments in flight across the barrier
          // (lgkmcnt(0), which a release fence emits, cost an LDS round trip per step).  No
          // scalar load is in flight here (they return out of order): the loop has none.
          // (DQ_TABLE: the table reads of the slots in between count too)
          // (the counter holds 15: a larger count means "everything but the 15 youngest", which
          // still covers the write)
          constexpr int BAR_WAIT = BAR_SLOT - WR_SLOT + TAB_OPS < 15 ? BAR_SLOT - WR_SLOT + TAB_OPS : 15;
          asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" : : "n"(BAR_WAIT) : "memory");
          if (s >= FL && s % FL == 0)              // timesteps < s are behind this barrier
            flush_ring<POOL, SLOTS, F6_NT, NPIX>(obuf, a, s - FL, FL, b, y0, x0, tid);
        }
        // pin the slot: what it produced is an operand of an (empty) volatile asm, so neither
        // the MFMA nor the stages drift into another slot
        // (the table's values are not pinned: that would wait for a read in the slot that issued it)
        asm volatile("" : "+v"(accN), "+v"(e.ew));
#pragma unroll
        for (int g = 0; g < GP; ++g)
          asm volatile("" : "+v"(e.ex[g][0]), "+v"(e.ex[g][1]), "+v"(e.euu[g][0]), "+v"(e.euu[g][1]));
        if constexpr (!TABLE) {
#pragma unroll
          for (int g = 0; g < EYN; ++g) asm volatile("" : "+v"(e.ey[g][0]), "+v"(e.ey[g][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      store_word(s, e.ew);
    };
    auto epilogue = [&](acc_t &acc, int t) {
      ETmp e;
      estage_head(acc, e);
#pragma unroll
      for (int st = 0; st < NSTAGES; ++st) estage(acc, st, e);
      store_word(t, e.ew);
    };

    // lane bases of the halo images: halo(t) lives in image t % 3 of the ring
    auto img_e = [&](int k) { return abase_even + (uint32_t)(k * HALO_B); };
    auto img_o = [&](int k) { return abase_odd + (uint32_t)(k * HALO_B); };

    // pipeline prologue: halo(0), halo(1) staged; MFMA(0) alone
    acc_t accA, accB;
    stage_load(0, 0);
    if (a.T > 1) stage_load(1, 1);
    stage_expand(0);
    if (a.T > 2) stage_load(2, 0);
    stage_write(0u);
    if (a.T > 1) {
      stage_expand(1);
      if (a.T > 3) stage_load(3, 1);
      stage_write((uint32_t)HALO_B);
    }
    lds_barrier();
    // (every wave has read the previous patch's claim: behind the barrier above)
    if (pw.queue && tid == 0) nxt[0] = (uint32_t)claimed;
    {
      const uint32_t e0 = img_e(0), o0 = img_o(0), e1 = img_e(1), o1 = img_o(1);
#pragma unroll
      for (int i = 0; i < PF; ++i) A[i] = a_read(e0, o0, i);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks == 0) accA = mfma_acc(0, A[0], TABLE ? cblk : zero16);
        else accA = mfma_acc(ks, A[ks % RING], accA);
        if (ks + PF < KS) A[(ks + PF) % RING] = a_read(e0, o0, ks + PF);
        else if (a.T > 1) A[(ks + PF) % RING] = a_read(e1, o1, ks + PF - KS);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // steady state, unrolled by two so the accumulator roles alternate; the image indices
    // (s + 1) % 3, (s + 2) % 3 rotate in scalar registers
    int s = 0;
    int k1 = 1, k2 = 2;                  // images of halo(s + 1), halo(s + 2)
    for (; s + 2 < a.T; s += 2) {
      fused_step(accB, accA, s, img_e(k1), img_o(k1), img_e(k2), img_o(k2), (uint32_t)(k2 * HALO_B),
                 0, true);
      const int k3 = k1 == 0 ? 2 : k1 - 1;        // image of halo(s + 3)
      fused_step(accA, accB, s + 1, img_e(k2), img_o(k2), img_e(k3), img_o(k3),
                 (uint32_t)(k3 * HALO_B), 1, s + 3 < a.T);
      k1 = k3;
      k2 = k3 == 2 ? 0 : k3 + 1;
    }
    if (s + 1 < a.T) {
      fused_step(accB, accA, s, img_e(k1), img_o(k1), img_e(k2), img_o(k2), (uint32_t)(k2 * HALO_B),
                 0, false);
      epilogue(accB, s + 1);
    } else {
      epilogue(accA, s);
    }
    lds_barrier();
    {
      const int done = a.T >= 2 ? ((a.T - 2) / FL) * FL : 0;
      flush_ring<POOL, SLOTS, F6_NT, NPIX>(obuf, a, done, a.T - done, b, y0, x0, tid);
    }
    if (a.u_out && wave_on) u_io_tile<false>(u, a, b, y0, x0, cout, h, 0);
    // the claimed patch: published behind this patch's barriers, and not overwritten before
    // every wave has passed the next patch's first barrier
    int r_next = r + (int)pw.stride;
    if (pw.queue) r_next = __builtin_amdgcn_readfirstlane((int)nxt[0]);
    r = r_next;
  }
  if (pw.queue && tid == 0) pw.finish(processed, a.npatch, a.status);
}

template <int FMT, int CIN, int NF, int DQ, bool FMA, bool BNF, bool BNU = false>
static void launch_bits_pool(const ConvMfmaArgs &a, bool pool, unsigned gy, hipStream_t st) {
  if (pool) launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, true, DQ, FMA, BNF, BNU>, a, gy, st, 0, F6_NT);
  else launch_persistent(conv3x3_bits_kernel<FMT, CIN, NF, false, DQ, FMA, BNF, BNU>, a, gy, st, 0, F6_NT);
}

// The general instance of a neuron form (three-instruction dequantisation, full BatchNorm,
// two-rounding membrane update) serves every launch; multi_step_LIF / PLIF with v_reset = 0
// (NF_MUL0, what every shipped config uses) additionally has the shortcuts the launch may
// prove: fused membrane update (fma), multiply-only BatchNorm (bnf), and on the fp6
// instruction the one-multiply dequantisation (L == 1).
template <int FMT, int CIN>
static void launch_bits_nf(const ConvMfmaArgs &a, int nf, bool pool, int dq, bool fma, bool bnf,
                           unsigned gy, hipStream_t st) {
  if (nf == NF_MUL) return launch_bits_pool<FMT, CIN, NF_MUL, DQ_ARITH, false, false>(a, pool, gy, st);
  if (nf == NF_DIV) return launch_bits_pool<FMT, CIN, NF_DIV, DQ_ARITH, false, false>(a, pool, gy, st);
  if (nf == NF_DECAY) return launch_bits_pool<FMT, CIN, NF_DECAY, DQ_ARITH, false, false>(a, pool, gy, st);
  if constexpr (FMT == FMT_FP6) {
    if (dq == DQ_TABLE) {
      // (the multiplier every channel shares folded into the table: the headline's form)
      if (fma && bnf && (a.bn.flags & SNNQP_BN_MUL_UNIFORM))
        return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_TABLE, true, true, true>(a, pool, gy, st);
      if (fma && bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_TABLE, true, true>(a, pool, gy, st);
      if (fma) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_TABLE, true, false>(a, pool, gy, st);
      if (bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_TABLE, false, true>(a, pool, gy, st);
      return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_TABLE, false, false>(a, pool, gy, st);
    }
    if (dq == DQ_ONE) {
      if (fma && bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ONE, true, true>(a, pool, gy, st);
      if (fma) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ONE, true, false>(a, pool, gy, st);
      if (bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ONE, false, true>(a, pool, gy, st);
      return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ONE, false, false>(a, pool, gy, st);
    }
  }
  if (fma && bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ARITH, true, true>(a, pool, gy, st);
  if (fma) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ARITH, true, false>(a, pool, gy, st);
  if (bnf) return launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ARITH, false, true>(a, pool, gy, st);
  launch_bits_pool<FMT, CIN, NF_MUL0, DQ_ARITH, false, false>(a, pool, gy, st);
}

// i8: codes wider than fp6 holds (|code| > 7) -> the int8 instruction
// dq: DQ_ARITH / DQ_ONE (conv_tile.h; ONE is a shortcut, ARITH with L = 1 computes the same);
// fma: NF_MUL0 with the fused membrane update; bnf: BatchNorm means and biases all zero
void launch_conv3x3_bits(const ConvMfmaArgs &a0, bool i8, int nf, bool pool, int dq, bool fma,
                         bool bnf, unsigned gy, hipStream_t st) {
  ConvMfmaArgs a = a0;                       // this kernel's patch: one tile of 4x8 pixels
  a.patch_h = 4;
  a.tiles_y = (a.H + 3) / 4;
  a.npatch = (int64_t)a.B * a.tiles_y * a.tiles_x;
  if (i8) {
    if (a.Cin <= 64) launch_bits_nf<FMT_I8, 64>(a, nf, pool, dq, fma, bnf, gy, st);
    else launch_bits_nf<FMT_I8, 128>(a, nf, pool, dq, fma, bnf, gy, st);
  } else {
    if (a.Cin <= 64) launch_bits_nf<FMT_FP6, 64>(a, nf, pool, dq, fma, bnf, gy, st);
    else launch_bits_nf<FMT_FP6, 128>(a, nf, pool, dq, fma, bnf, gy, st);
  }
}

}  // namespace snnqp
