// Fused SpikingBlock for the first 3x3 / stride 1 / pad 1 QuantConv layer of the DVS128
// topology (examples/tcja/models.py:111-147, Cin = 2 event-count frames): implicit-GEMM
// int8 MFMA (v_mfma_i32_32x32x32_i8) + dequantisation + eval BatchNorm + neuron update +
// optional 2x2 max-pool, with the T loop inside the kernel (conv3x3_u8c2_kernel), and
// the host side of both fused conv kernels: what they support
// (conv3x3_mfma_unsupported) and the dispatch (run_conv3x3_mfma) -- bit-packed inputs
// go to conv3x3_bits.hip.
//
// C/D layout of the MFMA: lane = output channel, register = pixel, so the per-channel
// dequant / BatchNorm constants are per-lane registers, the membrane potentials of a
// tile stay in 16 VGPRs for all T, and the v_cmp that thresholds a register *is* the
// packed spike word of two pixels (64-bit lane mask); pooling is an OR of those masks.
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>

#include <vector>

#include "conv_tile.h"

namespace snnqp {

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
//
// The table modes (LUTM, sized by the host for the input values it EXPECTS: the hint
// x_limit) are valid only while every staged input value is <= x_limit.  Nothing is
// trusted: while a chunk's halo sits in registers on its way to LDS the workgroup takes
// its maximum, and a chunk that holds a larger value (a hot pixel, event counts where
// binary frames were expected, a stale hint) runs the general path -- input as x - 128,
// arithmetic dequantisation, any count up to 255 -- for that chunk only.  The largest
// value a launch saw goes to x_seen[0] and the count of staged chunks by their largest value to
// x_seen[1..5] (<= 1, <= 2, <= 7, <= 31, above), from which the caller refines its next hint without
// ever waiting for it (no inspection pass, no host synchronisation before the launch).
// ---------------------------------------------------------------------------
constexpr int HROW2 = 24;                 // LDS bytes per halo row
constexpr int HCOPY2 = HALO * HROW2;      // one copy of one timestep
constexpr int HCONST2 = 2 * HCOPY2;       // the 8 constant bytes
constexpr int HIMG2 = 496;                // one timestep image
constexpr int TCHUNK = 32;                // most timesteps staged per pass
constexpr int U8C2_WPS = 4;               // waves per SIMD the kernel is compiled for (128 VGPRs)
constexpr int STG_N = TCHUNK / 2;           // timesteps a staging thread loads per chunk (every other one)

typedef int v2i_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(3))) const v2i_a4 lds_cv2i_t;

// LDS bytes of the u8c2 kernel: images | table | spike words
__host__ __device__ inline int u8c2_table_bytes(int lutm, int bound, int rows) {
  const int b = lutm == LUT_CHANNEL ? rows * 128 : lutm == LUT_SHARED ? (2 * bound + 2) * 4 : 0;
  return (b + 15) & ~15;
}

// IN: SNNQP_U8 (one byte per count) or SNNQP_EV1 (bit-packed binary frames, snnqp.h): the
// same kernel behind another staging loop.  EV1 frames cannot hold a value above 1, so the
// table modes need no check of the chunk (and the chunk no reduction, no atomic and one
// barrier less); a thread stages one halo ROW of one timestep -- 20 bits out of two words,
// expanded through a 256-entry byte -> 8-byte LDS table -- instead of one pixel.
// Waves per SIMD the variants are compiled for (profiles/r05_conv0_waves.txt).  The bit-packed
// variants: five.  With the potentials' life starting behind the staging code (template parameter
// ONE: nothing carried in or out, one chunk) they need about 90 registers and spill nothing: 4.95 ms
// on the headline layer against 5.32-5.35 for four waves (122 registers) on the same box; the
// general variant (potentials carried, several chunks) spills 31 dwords around its staging code at
// five waves and is still the faster one (5.10-5.18).
#ifndef SNNQP_U8C2_EV1_WPS
#define SNNQP_U8C2_EV1_WPS 5
#endif
// The byte formats' ONE variants: 102-103 registers and no scratch at four waves (128 with 40 bytes of
// scratch before the potentials moved behind the staging code): uint8 5.80 -> 5.56 ms, counts 5.85
// -> 5.67, nibbles 6.18 -> 6.00, float32 6.06 -> 5.85.  At five waves they spill 6 dwords per patch:
// 5.24 (uint8) / 5.51 (float32), counts unchanged (their tables leave LDS for four workgroups) --
// not taken while it costs scratch traffic.
#ifndef SNNQP_U8C2_ONE_WPS
#define SNNQP_U8C2_ONE_WPS 4
#endif
// SIX waves for the one variant that fits 80 registers without a spill -- bit-packed frames, ONE,
// per-channel tables, the tau = 2^j neuron with reset to 0, fused pool: the headline's -- with its
// table entries read in two halves (tile_epilogue_halves), tile 1's fragment and output word at
// immediate offsets from tile 0's and the wave index a scalar: 4.82-4.83 ms against 4.96-4.98 at
// five waves on the same box.  Its siblings would spill 2-6 dwords per patch at six: five.
#ifndef SNNQP_U8C2_EV1_ONE_WPS
#define SNNQP_U8C2_EV1_ONE_WPS 6
#endif
template <int NF, bool POOL, int LUTM, int IN, bool ONE>
constexpr int u8c2_waves() {
  if (IN == SNNQP_EV1 && ONE && LUTM == LUT_CHANNEL && NF == NF_MUL0 && POOL) return SNNQP_U8C2_EV1_ONE_WPS;
  if (IN == SNNQP_EV1) return SNNQP_U8C2_EV1_WPS;
  return ONE ? SNNQP_U8C2_ONE_WPS : U8C2_WPS;
}
template <int NF, bool POOL, int LUTM, int IN = SNNQP_U8, bool ONE = false>
__global__ void __launch_bounds__(256, (u8c2_waves<NF, POOL, LUTM, IN, ONE>()))
conv3x3_u8c2_kernel(ConvMfmaArgs a) {
  constexpr int FL = OutStage<POOL>::FL;
  constexpr bool EV1 = IN == SNNQP_EV1;
  constexpr bool EV4 = IN == SNNQP_EV4;     // one byte per pixel: polarity 0 low nibble, 1 high
  // float32 frames as the reference hands them over (flax_qconv.py:101): eight bytes per pixel,
  // converted to the two count bytes and checked while they wait in registers (snnqp.h, x_flags)
  constexpr bool F32IN = IN == SNNQP_F32;
  static_assert(TCHUNK % FL == 0, "flush period must divide the staging chunk");
  // a predicated launch (snnqp_conv_lif_forward_pred: the frames again in their own format behind
  // a speculative bit-packed launch) returns at once unless the word is set -- before any
  // workgroup has touched a queue word, which therefore stay zero for the slot's next user
  if constexpr (!EV1) {
    if (a.pred && *(const volatile int32_t *)a.pred == 0) return;
  }
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tc = a.tchunk;                       // <= TCHUNK
  const int lut_off = tc * HIMG2;
  uint32_t *obuf = (uint32_t *)(lds + lut_off + u8c2_table_bytes(LUTM, a.lut_bound, a.lut_rows));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: what hangs on it stays out of the VGPRs)
  const int n = lane & 31, h = lane >> 5;
  const int cout_base = blockIdx.y * 128 + wave * 32;
  const bool wave_on = cout_base < a.Cout;
  const int cout = wave_on ? cout_base + n : n;
  const int cpar = cout < a.Cout ? cout : a.Cout - 1;      // parameter loads
  const uint32_t cmask = chan_mask(cout_base, a.Cout);
  // workgroup words behind the spike-word ring:
  //   [0] smallest non-zero |input current| of this workgroup's channels (table kernels):
  //       decides whether the membrane update may be one fused multiply-add
  //   [1] largest input value of the chunk being staged   [2] the claimed next patch
  uint32_t *wgw = obuf + OutStage<POOL>::BYTES / 4;
  if (tid == 0) { wgw[0] = 0x7F800000u; wgw[1] = 0u; }
  // EV1: byte of 8 input bits -> 8 operand bytes, in the scale of the kernel's path: 16 x
  // (per-channel tables), 4 x (shared table), or x - 128 (no table: general path)
  typedef int v2i_a8 __attribute__((ext_vector_type(2)));
  v2i_a8 *xtab = (v2i_a8 *)(wgw + 4);
  if (EV1) {
    const uint32_t one = LUTM == LUT_CHANNEL ? 16u : LUTM == LUT_SHARED ? 4u : 0x81u;
    const uint32_t zero = LUTM == LUT_NONE ? 0x80u : 0u;
    uint32_t e[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j >> 2] |= (((uint32_t)tid >> j) & 1u ? one : zero) << (8 * (j & 3));
    xtab[tid] = v2i_a8{(int)e[0], (int)e[1]};       // 256 threads, 256 entries
  }
  if (LUTM != LUT_NONE) lds_barrier();
  // tables and constants become visible with the first staging barrier
  if (LUTM == LUT_SHARED) {
    build_lut((float *)(lds + lut_off), a.lut_bound, a.dq, tid);
    if (NF == NF_MUL0) {             // BatchNorm of every entry, per channel: may the update fuse?
      lds_barrier();
      atomicMin(wgw, lut_bn_min_bits((const float *)(lds + lut_off), a.lut_bound, a.bn,
                                     blockIdx.y * 128, a.Cout, tid, 256));
    }
  }
  int ch_row0 = 0, ch_col = 0;      // LUT_CHANNEL: the table row and column of this lane's entry of acc = 0
  if (LUTM == LUT_CHANNEL) {
    // what this lane's channel can accumulate: the sums of its positive and |negative| codes
    // (the lane halves hold k 0..15 and 16..23 of the same channel), times the largest input
    int pos = 0, neg = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = 16 * h + j, dy = k >> 3, b = k & 7;
      if (k < 24 && b < 6 && wave_on && cout < a.Cout) {      // (a wave beyond Cout: one row per lane)
        const int code = a.w[(int64_t)(6 * dy + b) * a.Cout + cout];
        pos += code > 0 ? code : 0;
        neg += code < 0 ? -code : 0;
      }
    }
    pos += __shfl_xor(pos, 32);
    neg += __shfl_xor(neg, 32);
    uint32_t *scr = obuf;            // (the spike-word ring is not in use before the first patch)
    // the bank column of this channel's table and its place in the column's stack (conv_tile.h)
    int slot = 4 * n + wave;
    if (a.ch_slots) slot = a.ch_slots[blockIdx.y * 128 + wave * 32 + n] & 127;
    if (h == 0) {
      scr[slot] = (uint32_t)((pos + neg) * a.x_limit + 1);
      scr[128 + wave * 32 + n] = (uint32_t)(neg * a.x_limit);
      scr[256 + wave * 32 + n] = (uint32_t)slot;
    }
    lds_barrier();
    ch_col = slot >> 2;
    const int start = lut_column_start(scr, slot);
    int total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) total += (int)scr[(slot & ~3) + k];
    ch_row0 = start + neg * a.x_limit;
    // a column taller than the launch reserved: snnqp_weight_t.ch_stack_max is below the codes it
    // came with.  Its entries are not written (reads beyond the allocation return zeros): wrong
    // currents, reported -- the next call fails until the status is reset (snnqp.h).
    if (total > a.lut_rows && a.status) *(volatile uint32_t *)a.status = SNNQP_STATUS_BOUND;
    const uint32_t mb = build_lut_channel((float *)(lds + lut_off), scr, a.lut_rows, a.dq, a.bn,
                                          blockIdx.y * 128, a.Cout, tid);
    atomicMin(wgw, mb);
    lds_barrier();                   // the scratch becomes the spike-word ring again
  }
  if (tid < tc) {
    *(uint32_t *)(lds + tid * HIMG2 + HCONST2) = 0x7F7F7F7Fu;
    *(uint32_t *)(lds + tid * HIMG2 + HCONST2 + 4) = 0x00000001u;
  }

  // B operands.  Table path (`bf`): the codes (x 8 for per-channel tables) and, over the
  // constant k rows, the byte address of this lane's table entry of acc = 0 as 127 q + r
  // (rows 24..27 take q in parts of at most 127, row 28 takes r).  General path (`bfg`): the
  // plain codes; the input is taken as x - 128 (a signed int8 for every count up to 255;
  // padding pixels are x = 0 like any other) and 128 * sum_k w[k] is added back to the
  // accumulator in the epilogue (an integer below 2^24: exact in float32).
  float acc_off = 0.0f;
  v4i bf, bfg;
  {
    int bias = 0;
    if (LUTM == LUT_SHARED) bias = (int)lds_addr(lds) + lut_off + 4 * a.lut_bound;
    if (LUTM == LUT_CHANNEL)     // row of acc = 0 of this lane's channel, its column
      bias = (int)lds_addr(lds) + lut_off + 4 * (ch_row0 * 32 + ch_col);
    int q = bias / 127;
    const int r = bias - 127 * q;
    int wsum = 0;
    int v[4], vg[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t pk = 0, pkg = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * h + 4 * d + j;
        uint32_t bv = 0, bg = 0;
        if (k < 24) {
          const int dy = k >> 3, b = k & 7;
          if (b < 6 && cout < a.Cout) {   // HWIO with Cin = 2: row (3 dy + dx) * 2 + cin = 6 dy + b
            const int code = a.w[(int64_t)(6 * dy + b) * a.Cout + cout];
            wsum += code;
            bg = (uint8_t)code;
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
        pkg |= bg << (8 * j);
      }
      v[d] = (int)pk;
      vg[d] = (int)pkg;
    }
    bf = v4i{v[0], v[1], v[2], v[3]};
    bfg = v4i{vg[0], vg[1], vg[2], vg[3]};
    // the two lane halves hold k 0..15 and 16..23 of the same channel
    acc_off = 128.0f * (float)(wsum + __shfl_xor(wsum, 32));
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
  uint32_t seen = 0;                       // largest input value this thread's waves met
  // staged chunks by their largest value: <= 1, <= 2, <= 7, <= 31, above (workgroup-uniform
  // counters; they reach x_seen[1..5] once, at the end of the launch)
  uint32_t hist[6] = {0, 0, 0, 0, 0, 0};
  // staging tasks: waves 0-1 take the even timesteps of a chunk, waves 2-3 the odd ones; the
  // first 100 threads of each pair take one halo pixel each.  The timestep is uniform over a
  // wave, so a load is a scalar frame address plus this thread's 32-bit pixel offset.
  const int s_pix = tid & 127;
  const bool s_task = s_pix < HALO * HALO;
  const int s_half = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int s_dst = (s_pix / HALO) * HROW2 + (s_pix % HALO) * 2;

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

    // ONE: no potentials carried in or out and all timesteps in one chunk -- the potentials then
    // start their life behind the staging code instead of through it (32 registers the staging
    // does not have to work around)
    float u[2][16];
    if constexpr (!ONE) {
      if (a.u0 && wave_on) u_io<true>(u, a, b, y0, x0, cout, h);
      else zero_u(u);
    }
    uint32_t pseen = 0;                  // largest input value of this patch so far
    const int s_gy = y0 + s_pix / HALO - 1, s_gx = x0 + s_pix % HALO - 1;
    const bool s_valid = s_task && s_gy >= 0 && s_gy < a.H && s_gx >= 0 && s_gx < a.W;
    // byte offset of the pixel within a frame (frames are below 2 GiB: launch check); threads
    // without a pixel read the frame's first one and drop it
    const uint32_t s_off = s_valid ? (uint32_t)(s_gy * a.W + s_gx) * (EV4 ? 1u : F32IN ? 8u : 2u) : 0u;
    const uint8_t *s_frames = xb + (int64_t)b * a.xs_b * (F32IN ? 4 : 1);
    // EV1: the 20 bits of a halo row start `lead` bits before pixel x0 of the row (none when
    // x0 = 0: the pixel left of the image does not exist); pixels outside the image are
    // cleared by a mask of two bits per halo column (workgroup-uniform: scalars)
    const uint32_t e_lead = x0 > 0 ? 2u : 0u;
    const int e_hi = min(HALO - 1, a.W - x0);                  // last halo column inside the image
    const uint32_t e_colmask = ((1u << (2 * (e_hi + 1))) - 1u) & ~((1u << (2 - (int)e_lead)) - 1u);
    const uint32_t e_fwords = (uint32_t)(((int64_t)a.H * a.W * 2 + 31) >> 5);
    const uint32_t *e_frames = (const uint32_t *)a.x + (int64_t)b * a.xs_b;

    for (int t0 = 0; t0 < a.T; t0 += tc) {
      const int nt = min(tc, a.T - t0);
      bool general = LUTM == LUT_NONE;
      if constexpr (EV1) {
        // ---- stage the chunk from bit-packed frames: one (timestep, halo row) per thread ----
        lds_barrier();                     // previous readers of the LDS images are done
        for (int task = tid; task < nt * HALO; task += 256) {
          const int tt = task / HALO, hy = task - tt * HALO;
          const int gy = y0 + hy - 1;
          const bool rv = gy >= 0 && gy < a.H;
          const uint32_t start = (uint32_t)((rv ? gy : 0) * a.W + x0) * 2u - e_lead;
          const uint32_t wi = start >> 5;
          const uint32_t *f = e_frames + (int64_t)(t0 + tt) * a.xs_t;
          // (the word behind the frame's last one is never needed: its bits would be pixels
          // beyond the image, which the mask clears)
          const uint32_t lo = f[wi], hi = f[min(wi + 1u, e_fwords - 1u)];
          uint32_t v = __builtin_amdgcn_alignbit(hi, lo, start & 31u);
          v = (v << (2u - e_lead)) & (rv ? e_colmask : 0u);
          const v2i_a8 d0 = xtab[v & 0xFFu], d1 = xtab[(v >> 8) & 0xFFu], d2 = xtab[(v >> 16) & 0xFu];
          uint8_t *row = lds + tt * HIMG2 + hy * HROW2;
          *(v2i_a8 *)row = d0;
          *(v2i_a8 *)(row + 8) = d1;
          *(v2i_a8 *)(row + 16) = d2;
          // copy 1: the same bytes two to the right (pixel hx at byte 2 hx + 2)
          const uint32_t q0 = (uint32_t)d0.x, q1 = (uint32_t)d0.y, q2 = (uint32_t)d1.x,
                         q3 = (uint32_t)d1.y, q4 = (uint32_t)d2.x, q5 = (uint32_t)d2.y;
          uint8_t *row1 = row + HCOPY2;
          *(v2i_a8 *)row1 = v2i_a8{(int)(q0 << 16), (int)__builtin_amdgcn_alignbit(q1, q0, 16)};
          *(v2i_a8 *)(row1 + 8) = v2i_a8{(int)__builtin_amdgcn_alignbit(q2, q1, 16),
                                         (int)__builtin_amdgcn_alignbit(q3, q2, 16)};
          *(v2i_a8 *)(row1 + 16) = v2i_a8{(int)__builtin_amdgcn_alignbit(q4, q3, 16),
                                          (int)__builtin_amdgcn_alignbit(q5, q4, 16)};
        }
        lds_barrier();
      } else {
      // ---- stage the chunk: load, take the maximum, choose the path, write ----
      // Thread = one halo pixel (both polarities: 2 bytes) of every other timestep: its only
      // per-patch state is one source pointer, so nothing else is live while the byte pairs
      // wait in registers for the workgroup to agree on the path.
      uint32_t v[STG_N];
      uint32_t mx = 0;
      if constexpr (F32IN) {
        // two rounds of STG_N / 2 eight-byte loads: the raw pairs need two registers each
        uint32_t fbad = 0;
#pragma unroll
        for (int r0 = 0; r0 < STG_N; r0 += STG_N / 2) {
          typedef float f2v __attribute__((ext_vector_type(2)));
          f2v raw[STG_N / 2];
#pragma unroll
          for (int i = r0; i < r0 + STG_N / 2; ++i) {
            const int tt = 2 * i + s_half;
            raw[i - r0] = f2v{0.0f, 0.0f};
            if (tt < nt) raw[i - r0] = *(const f2v *)(s_frames + (int64_t)(t0 + tt) * a.xs_t * 4 + s_off);
          }
#pragma unroll
          for (int i = r0; i < r0 + STG_N / 2; ++i) {
            // fl(cvt(x)) - x is +0.0 exactly for the integers v_cvt_u32_f32 holds (-0.0 counts
            // as 0), NaN for NaN, non-zero otherwise; the range is checked on the integers
            const f2v f = raw[i - r0];
            const uint32_t ua = __float2uint_rz(f.x), ub = __float2uint_rz(f.y);
            fbad |= __float_as_uint(__uint2float_rn(ua) - f.x) | __float_as_uint(__uint2float_rn(ub) - f.y);
            fbad |= (ua | ub) >> 8;
            v[i] = (ua & 0xFFu) | ((ub & 0xFFu) << 8);
          }
        }
        fbad = s_valid ? fbad : 0u;
        if (__ballot(fbad != 0u) != 0ull && lane == 0 && a.x_flags) atomicOr(a.x_flags, SNNQP_FLAG_NOT_INTEGER);
      } else {
#pragma unroll
      for (int i = 0; i < STG_N; ++i) {    // all loads first; the skip is a scalar branch
        const int tt = 2 * i + s_half;
        v[i] = 0;
        if (tt < nt) {
          if constexpr (EV4) {           // nibble-packed counts: the two bytes the uint8 frame would hold
            const uint32_t pb = *(s_frames + (int64_t)(t0 + tt) * a.xs_t + s_off);
            v[i] = (pb | (pb << 4)) & 0x0F0Fu;          // 0xHL -> 0x0H0L
          } else {
            v[i] = *(const uint16_t *)(s_frames + (int64_t)(t0 + tt) * a.xs_t + s_off);
          }
        }
      }
      }
#pragma unroll
      for (int i = 0; i < STG_N; ++i) {
        v[i] = s_valid ? v[i] : 0u;
        mx = max(mx, max(v[i] & 0xFFu, v[i] >> 8));
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
      // The halo is written in the scale of the path the hint EXPECTS (round 5; it used to wait
      // for the workgroup's verdict: a barrier more per chunk, and every write behind the
      // reduction) while the chunk's maximum travels through an LDS atomic; a chunk that exceeds
      // the hint -- rare by construction of the hint -- is loaded and written again in the
      // general path's scale.
      // (every wave read the last chunk's maximum before the barriers of its run_chunk)
      if (tid == 0) wgw[1] = 0u;
      lds_barrier();                       // previous readers of the LDS images are done
      if (s_task) {
#pragma unroll
        for (int i = 0; i < STG_N; ++i) {
          const int tt = 2 * i + s_half;
          if (tt < nt) {    // table modes: both bytes scale without a carry (counts <= 31 / 7)
            const uint16_t val = (uint16_t)(LUTM == LUT_NONE      ? v[i] ^ 0x8080u    // x - 128
                                            : LUTM == LUT_CHANNEL ? v[i] << 4
                                                                  : v[i] << 2);
            uint8_t *p = lds + tt * HIMG2 + s_dst;
            *(uint16_t *)p = val;
            *(uint16_t *)(p + HCOPY2 + 2) = val;
          }
        }
      }
      if (lane == 0 && mx != 0) atomicMax(&wgw[1], mx);
      lds_barrier();
      const uint32_t cmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wgw[1]);   // one word
      seen = max(seen, cmax);
      pseen = max(pseen, cmax);
      hist[0] += cmax <= 1u;
      hist[1] += cmax == 2u;
      hist[2] += cmax > 2u && cmax <= 7u;
      hist[3] += cmax > 7u && cmax <= 31u;
      hist[4] += cmax > 31u;
      hist[5] += cmax == 3u;
      general = LUTM == LUT_NONE || cmax > (uint32_t)a.x_limit;
      if (LUTM != LUT_NONE && general) {
        // the rare chunk: its values again (an L2 hit; nothing was kept in registers for it)
        if (s_task) {
#pragma unroll 1
          for (int tt = s_half; tt < nt; tt += 2) {
            uint32_t w2 = 0;
            if (s_valid) {
              if constexpr (F32IN) {
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v f = *(const f2v *)(s_frames + (int64_t)(t0 + tt) * a.xs_t * 4 + s_off);
                w2 = (__float2uint_rz(f.x) & 0xFFu) | ((__float2uint_rz(f.y) & 0xFFu) << 8);
              } else if constexpr (EV4) {
                const uint32_t pb = *(s_frames + (int64_t)(t0 + tt) * a.xs_t + s_off);
                w2 = (pb | (pb << 4)) & 0x0F0Fu;
              } else {
                w2 = *(const uint16_t *)(s_frames + (int64_t)(t0 + tt) * a.xs_t + s_off);
              }
            }
            const uint16_t val = (uint16_t)(w2 ^ 0x8080u);
            uint8_t *p = lds + tt * HIMG2 + s_dst;
            *(uint16_t *)p = val;
            *(uint16_t *)(p + HCOPY2 + 2) = val;
          }
        }
        lds_barrier();
      }
      }
      // the FL-step blocks of the chunk: table path (with the membrane update as a fused
      // multiply-add where that is proven bit-identical for this launch) or general path
      auto run_chunk = [&](auto mode_tag, auto fma_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool FMA = decltype(fma_tag)::value;
      constexpr bool OFFS = MODE == LUT_NONE;
      const v4i bw = OFFS ? bfg : bf;
      for (int tf = 0; tf < nt; tf += FL) {          // FL steps, then flush
        const int nf = min(FL, nt - tf);
#pragma unroll 1
        for (int tt = tf; tt < tf + nf; ++tt) {
          const uint32_t img = (uint32_t)(tt * HIMG2);
          uint32_t words[2];
#pragma unroll
          for (int tl = 0; tl < 2; ++tl) {
            // (tile 1 is four halo rows below tile 0: an immediate offset of the read, no register)
            const v2i_a4 lo = *(lds_cv2i_t *)(uintptr_t)(img + (uint32_t)offA[0] + (uint32_t)(tl * 4 * HROW2));
            const v2i_a4 hi = *(lds_cv2i_t *)(uintptr_t)(img + (uint32_t)offB[tl]);
            v16i acc = splat16(0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{lo.x, lo.y, hi.x, hi.y}, bw, acc,
                                                        0, 0, 0);
            // (the six-wave build reads its table entries in two halves: eight in flight, not sixteen)
            if constexpr (u8c2_waves<NF, POOL, LUTM, IN, ONE>() >= 6)
              words[tl] = tile_epilogue_halves<NF, POOL, MODE, FMA, OFFS>(acc, u[tl], a.dq, lc, a.nrn,
                                                                            lane, acc_off);
            else
              words[tl] = tile_epilogue<NF, POOL, MODE, FMA, OFFS>(acc, u[tl], a.dq, lc, a.nrn,
                                                                     lane, acc_off);
          }
          if (store_lane) {
            uint32_t *o = obuf + ((t0 + tt) % FL) * (OutStage<POOL>::NPIX * 4);
            o[ob0] = words[0] & cmask;
            o[ob0 + (POOL ? 32 : 128)] = words[1] & cmask;       // out_pix(1, lane) = out_pix(0, lane) + 8 / + 32
          }
        }
        lds_barrier();
        flush_out<POOL>(obuf, a, t0 + tf, nf, b, y0, x0, tid);
        lds_barrier();
      }
      };
      if constexpr (ONE) zero_u(u);
      if (general) {
        run_chunk(std::integral_constant<int, LUT_NONE>{}, std::false_type{});
      } else if constexpr (LUTM != LUT_NONE) {
        // a patch whose earlier chunk took the general path holds potentials the table
        // path's power-of-two argument does not cover: fused only while the whole patch
        // so far stayed within the tables
        const bool fma_ok = NF == NF_MUL0 && pseen <= (uint32_t)a.x_limit &&
                            lif_fma_is_exact(wgw[0], a.nrn.k_log2, a.T, a.u0 != nullptr);
        if (fma_ok) run_chunk(std::integral_constant<int, LUTM>{}, std::true_type{});
        else run_chunk(std::integral_constant<int, LUTM>{}, std::false_type{});
      }
      if constexpr (ONE) break;          // (a.T <= tc: the launcher's condition for this variant)
    }
    if (!ONE && a.u_out && wave_on) u_io<false>(u, a, b, y0, x0, cout, h);
    int r_next = r + (int)pw.stride;
    if (pw.queue) {                      // the claimed patch, to the whole workgroup
      if (tid == 0) wgw[2] = (uint32_t)claimed;
      lds_barrier();
      r_next = __builtin_amdgcn_readfirstlane((int)wgw[2]);
    }
    r = r_next;
  }
  if (a.x_seen && tid == 0) {
    if (seen != 0) atomicMax(a.x_seen, seen);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (hist[k] != 0) atomicAdd((uint32_t *)a.x_seen + 1 + k, hist[k]);
  }
  if (pw.queue && tid == 0) pw.finish(processed, a.npatch, a.status);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// snnqp_current_min: one thread per (channel, table slice)
__global__ void __launch_bounds__(256)
current_min_kernel(Dequant dq, BnP bn, int bound, int Cout, uint32_t *out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  uint32_t mb = 0x7F800000u;
  if (c < Cout) {
    float mean = 0.f, mul = 1.f, bias = 0.f;
    if (bn.mean) { mean = bn.mean[c]; mul = bn.mul[c]; bias = bn.bias[c]; }
    for (int i = (int)(threadIdx.x >> 6) + 4 * (int)blockIdx.y - bound; i <= bound; i += 4 * (int)gridDim.y) {
      float x = dequant_acc_nb(i, dq) - mean;       // the epilogue's operation order
      x = x * mul;
      x = x + bias;
      const uint32_t b = __float_as_uint(x) & 0x7FFFFFFFu;
      if (b != 0 && b < mb) mb = b;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)mb, off);
    mb = o < mb ? o : mb;
  }
  if ((threadIdx.x & 63) == 0 && mb != 0x7F800000u) atomicMin(out, mb);
}

namespace {
struct SchedPool {
  uint32_t *words = nullptr;
  hipEvent_t busy[SCHED_SLOTS] = {};
  bool has_event[SCHED_SLOTS] = {};
  bool retired[SCHED_SLOTS] = {};      // no event could be made: never handed out again
  unsigned next = 0;
  unsigned next_capture = 0;           // slots SCHED_SLOTS + i: one per captured launch
  std::vector<int> capture_log;        // capture slots in the order they were handed out
  std::vector<int> capture_free;       // ... and the ones handed back (snnqp_workqueue_capture_release)
};
std::atomic<int64_t> g_static_captured{0}, g_static_busy{0};
std::mutex g_sched_mu;
SchedPool g_sched[64];

// The pool's words and events belong to the device of the launch stream, which need not be
// the calling thread's current device: allocate and create them with that device current.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
    if (cur == dev) return;
    if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
    prev = cur;
  }
  ~DeviceGuard() {
    if (prev >= 0 && hipSetDevice(prev) != hipSuccess) (void)hipGetLastError();
  }
};
}  // namespace

uint32_t *sched_acquire(hipStream_t st, int *dev_out, int *slot_out) {
  int dev = 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  hipDevice_t sdev;
  if (hipStreamGetDevice(st, &sdev) == hipSuccess) dev = (int)sdev;
  else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_sched_mu);
  SchedPool &p = g_sched[dev];
  if (capturing) {
    // a slot of its own, zero since the pool was allocated and zero again after every launch
    // that walked it (conv_tile.h)
    if (!p.words) { g_static_captured.fetch_add(1, std::memory_order_relaxed); return nullptr; }
    int slot;
    if (!p.capture_free.empty()) {
      slot = p.capture_free.back();
      p.capture_free.pop_back();
    } else if (p.next_capture < (unsigned)SCHED_CAPTURE_SLOTS) {
      slot = SCHED_SLOTS + (int)p.next_capture++;
    } else {
      g_static_captured.fetch_add(1, std::memory_order_relaxed);     // every capture slot is taken
      return nullptr;
    }
    p.capture_log.push_back(slot);
    *dev_out = dev;
    *slot_out = slot;
    return p.words + (size_t)slot * SCHED_WORDS;
  }
  if (!p.words) {
    DeviceGuard on(dev);
    if (!on.ok) return nullptr;
    uint32_t *w = nullptr;
    if (hipMalloc((void **)&w, (size_t)(SCHED_SLOTS + SCHED_CAPTURE_SLOTS) * SCHED_WORDS *
                                   sizeof(uint32_t)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    // (synchronous, once per device: the capture slots rely on it)
    if (hipMemset(w, 0, (size_t)(SCHED_SLOTS + SCHED_CAPTURE_SLOTS) * SCHED_WORDS * sizeof(uint32_t)) !=
        hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(w);
      return nullptr;
    }
    p.words = w;
  }
  const int slot = (int)(p.next++ % SCHED_SLOTS);
  if (p.retired[slot]) return nullptr;
  if (!p.has_event[slot]) {
    // the event that will guard the slot exists BEFORE the slot is handed out: a slot whose
    // launches could not be tracked would look free while one is still walking it
    DeviceGuard on(dev);
    if (!on.ok || hipEventCreateWithFlags(&p.busy[slot], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      p.retired[slot] = true;
      return nullptr;
    }
    p.has_event[slot] = true;
  } else if (hipEventQuery(p.busy[slot]) != hipSuccess) {
    (void)hipGetLastError();            // hipErrorNotReady: the slot's last launch is in flight
    g_static_busy.fetch_add(1, std::memory_order_relaxed);
    return nullptr;
  }
  uint32_t *words = p.words + (size_t)slot * SCHED_WORDS;
  if (hipMemsetAsync(words, 0, SCHED_WORDS * sizeof(uint32_t), st) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *dev_out = dev;
  *slot_out = slot;
  return words;
}

void sched_release(int dev, int slot, hipStream_t st) {
  if (slot >= SCHED_SLOTS) return;      // a captured launch's slot: never reused, nothing to track
  std::lock_guard<std::mutex> lock(g_sched_mu);
  SchedPool &p = g_sched[dev];
  // (the event was created in sched_acquire).  A failed record leaves the event at its
  // previous state -- "complete" -- while the launch runs: the slot can no longer be
  // proven free, so it leaves the rotation.
  if (hipEventRecord(p.busy[slot], st) != hipSuccess) {
    (void)hipGetLastError();
    p.retired[slot] = true;
  }
}

int stream_device(hipStream_t st) {
  int dev = 0;
  hipDevice_t sdev;
  if (hipStreamGetDevice(st, &sdev) == hipSuccess) return (int)sdev;
  (void)hipGetLastError();
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return dev;
}

void persistent_limits(const void *kernel, int threads, size_t dyn_lds, int dev, int *cus_out,
                       int *occ_out) {
  typedef std::tuple<const void *, int, size_t, int> Key;
  static std::mutex mu;
  static std::map<Key, std::pair<int, int>> cache;
  const Key key(kernel, threads, dyn_lds, dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) {
    int cus = 256, occ = 1;
    DeviceGuard on(dev);               // the occupancy query answers for the current device
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      cus = 256;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, dyn_lds) != hipSuccess) {
      (void)hipGetLastError();
      occ = 1;
    }
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;
    it = cache.emplace(key, std::make_pair(cus, occ)).first;
  }
  *cus_out = it->second.first;
  *occ_out = it->second.second;
}

int run_current_min(const snnqp_weight_t *w, const snnqp_bn_t *bn, int32_t bound, int32_t Cout,
                    uint32_t *out_bits, hipStream_t st) {
  SNNQP_REQUIRE(w && out_bits && Cout > 0 && bound >= 0, SNNQP_EINVAL, "current_min: bad argument");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  SNNQP_CHECK_BN(bn);
  const int slices = bound >= 256 ? 16 : 1;
  hipLaunchKernelGGL(current_min_kernel, dim3((Cout + 63) / 64, slices), dim3(256), 0, st,
                     make_dequant(w->L, w->m), make_bn(bn), bound, Cout, out_bits);
  SNNQP_CHECK_LAUNCH("current_min_kernel");
  return SNNQP_OK;
}

const char *conv3x3_mfma_unsupported(int in_type, const snnqp_conv_geom_t *g,
                                     const snnqp_weight_t *w, const int8_t *wt,
                                     const snnqp_neuron_t *nrn, int s_type) {
  if (w->wtype != SNNQP_W_I8) return "weights are not int8 codes";
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
    if ((int64_t)g->H * g->W * 2 >= (int64_t)1 << 31) return "u8 frame of 2 GiB or more";
  } else if (in_type == SNNQP_EV1) {    // bit-packed binary event frames, staged directly
    if (g->Cin != 2) return "EV1 frames have Cin == 2";
    if ((int64_t)g->H * g->W * 2 >= (int64_t)1 << 31) return "EV1 frame of 2^31 bits or more";
  } else if (in_type == SNNQP_EV4) {    // nibble-packed count frames (<= 15), staged directly
    if (g->Cin != 2) return "EV4 frames have Cin == 2";
    if ((int64_t)g->H * g->W >= (int64_t)1 << 31) return "EV4 frame of 2 GiB or more";
  } else if (in_type == SNNQP_F32) {    // integer-valued float32 frames, staged in place and checked
    if (g->Cin != 2) return "float32 input into integer codes needs Cin == 2";
    if ((int64_t)g->H * g->W * 8 >= (int64_t)1 << 31) return "float32 frame of 2 GiB or more";
  } else {
    return "input must be BITS, U8, EV1, EV4 or (Cin == 2) F32";
  }
  if (nrn->kind == SNNQP_NEURON_LIF && !nrn->decay) return "LIF without decay";
  if (in_type == SNNQP_BITS && !wt) return "MFMA-tiled codes `wt` not given";
  return nullptr;
}

// DQ_ONE / DQ_TABLE / DQ_ARITH (conv_tile.h) for a bit-input block on these weights
int conv3x3_bits_dequant_form(const snnqp_weight_t *w, const snnqp_neuron_t *nrn) {
  if (w->L == 1.0f) return DQ_ONE;
  const bool fp6 = w->code_max > 0 && w->code_max <= 7;
  const bool tab = fp6 && neuron_form(make_neuron(nrn)) == NF_MUL0 && w->abs_sum_max > 0 &&
                   w->abs_sum_max <= DQT_MAXA;
  return tab ? DQ_TABLE : DQ_ARITH;
}

int run_conv3x3_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b,
                     int32_t T, int32_t B, const snnqp_conv_geom_t *g,
                     const snnqp_weight_t *w, const int8_t *wt,
                     const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                     const float *u0, float *u_out, uint32_t *s_out, int pool,
                     int x_max, int32_t *x_seen, int32_t *x_flags, hipStream_t st, const int32_t *pred) {
  SNNQP_REQUIRE(w->w && ((x && s_out) || T == 0 || B == 0), SNNQP_EINVAL, "conv3x3 mfma: null pointer");
  SNNQP_REQUIRE(!pred || in_type == SNNQP_U8 || in_type == SNNQP_F32 || in_type == SNNQP_EV4, SNNQP_EUNSUPPORTED,
                "conv3x3 mfma: only the event layer on byte / nibble / float32 frames takes a predicate");   // (an empty batch has no buffers)
  SNNQP_REQUIRE(in_type != SNNQP_BITS || wt, SNNQP_EINVAL,
                "conv3x3 mfma: bit input needs the MFMA-tiled codes `wt`");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "conv3x3 mfma: negative T/B");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  SNNQP_CHECK_BN(bn);
  if (T == 0 || B == 0) return SNNQP_OK;
  // the kernels keep a patch index (+ one grid stride) in a 32-bit scalar register; the
  // smallest patch is the bits kernel's 4 x 8 pixels
  SNNQP_REQUIRE((int64_t)B * ((g->H + 3) / 4) * ((g->W + 7) / 8) < ((int64_t)1 << 30),
                SNNQP_EUNSUPPORTED, "conv3x3 mfma: more than 2^30 patches in one launch");
  ConvMfmaArgs a;
  a.x = x; a.xs_t = xs_t; a.xs_b = xs_b; a.T = T; a.B = B;
  a.H = g->H; a.W = g->W; a.Cin = g->Cin; a.Cout = g->Cout;
  a.w = (const int8_t *)w->w;
  a.wt = wt;
  a.dq = make_dequant(w->L, w->m);
  a.bn = make_bn(bn);
  a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out; a.pool = pool;
  a.x_seen = x_seen;
  a.x_flags = nullptr;
  a.pred = pred;
  if (in_type == SNNQP_F32) {
    SNNQP_REQUIRE(x_flags != nullptr && (((uintptr_t)x) & 7) == 0 && xs_t % 2 == 0 && xs_b % 2 == 0,
                  SNNQP_EINVAL, "conv3x3 mfma: float32 frames need x_flags and 8-byte aligned pixels");
    if (int rc = zero_words_async((uint32_t *)x_flags, 1, st)) return rc;
    a.x_flags = x_flags;
  }
  a.patch_h = 8;
  a.tiles_y = (g->H + 7) / 8; a.tiles_x = (g->W + 7) / 8;
  a.npatch = (int64_t)B * a.tiles_y * a.tiles_x;
  const int nf = neuron_form(a.nrn);          // which straight-line epilogue (conv_tile.h)
  const bool pl = pool == 2;
  const unsigned gy = (unsigned)((g->Cout + 127) / 128);
  // |acc| <= abs_sum_max * x_max; small enough -> dequantise through an LDS table
  // (the A operand then carries 4 * x, which must stay an int8).  For U8 input x_max is the
  // value the caller EXPECTS not to be exceeded (0 / unknown: binary events); the kernel
  // checks every chunk it stages and runs the general path where the hint does not hold
  const bool ev1 = in_type == SNNQP_EV1;
  const int64_t xm = (in_type == SNNQP_BITS || ev1) ? 1 : (x_max > 0 ? x_max : 1);
  a.x_limit = (int32_t)(xm > 255 ? 255 : xm);
  const int64_t bound = (int64_t)w->abs_sum_max * xm;
  const bool lut = w->abs_sum_max > 0 && xm > 0 && xm <= LUT_XMAX && bound <= LUT_CAP;
  a.lut_bound = lut ? (int32_t)bound : 0;
  a.tchunk = T >= TCHUNK ? TCHUNK : T;
  // images | table | spike-word ring | 4 workgroup words | EV1: byte -> 8-byte table
  const size_t lds_fixed = (size_t)a.tchunk * HIMG2 + 16 + (ev1 ? 2048 : 0) +
                           (pl ? OutStage<true>::BYTES : OutStage<false>::BYTES);
  // per-channel tables (BatchNorm folded in; the accumulator counts table rows of 128 B: A = 16 x
  // input, B = 8 x code), every channel over its own accumulator range, stacked per LDS bank
  // (conv_tile.h build_lut_channel): ch_stack_max x the largest input + 4 rows -- 8 x abs_sum_max
  // bounds it when the caller does not know
  const int64_t stack = w->ch_stack_max > 0 ? (int64_t)w->ch_stack_max : 8 * (int64_t)w->abs_sum_max;
  const int64_t crows = stack * xm + 4;
  const bool lutc = lut && in_type != SNNQP_BITS && crows <= LUT2_ROWS && xm <= 7 &&
                    w->code_max > 0 && w->code_max <= 15;
  a.lut_rows = lutc ? (int32_t)crows : 0;
  a.ch_slots = lutc && w->ch_stack_max > 0 ? w->ch_slots : nullptr;
#define SNNQP_CONV_LAUNCH_IN(KERN, NFV, PL, LM, LDS)                               \
  do {                                                                             \
    if (ev1 && one) launch_persistent(KERN<NFV, PL, LM, SNNQP_EV1, true>, a, gy, st, LDS); \
    else if (ev1) launch_persistent(KERN<NFV, PL, LM, SNNQP_EV1>, a, gy, st, LDS);       \
    else if (in_type == SNNQP_EV4 && one) launch_persistent(KERN<NFV, PL, LM, SNNQP_EV4, true>, a, gy, st, LDS); \
    else if (in_type == SNNQP_EV4) launch_persistent(KERN<NFV, PL, LM, SNNQP_EV4>, a, gy, st, LDS); \
    else if (in_type == SNNQP_F32 && one) launch_persistent(KERN<NFV, PL, LM, SNNQP_F32, true>, a, gy, st, LDS); \
    else if (in_type == SNNQP_F32) launch_persistent(KERN<NFV, PL, LM, SNNQP_F32>, a, gy, st, LDS); \
    else if (one) launch_persistent(KERN<NFV, PL, LM, SNNQP_U8, true>, a, gy, st, LDS);            \
    else launch_persistent(KERN<NFV, PL, LM, SNNQP_U8>, a, gy, st, LDS);            \
  } while (0)
#define SNNQP_CONV_LAUNCH_NF(KERN, NFV, LM, LDS)                                   \
  do {                                                                             \
    if (pl) SNNQP_CONV_LAUNCH_IN(KERN, NFV, true, LM, LDS);                         \
    else SNNQP_CONV_LAUNCH_IN(KERN, NFV, false, LM, LDS);                           \
  } while (0)
#define SNNQP_CONV_LAUNCH(KERN, LM, LDS)                                           \
  do {                                                                             \
    if (nf == NF_MUL0) SNNQP_CONV_LAUNCH_NF(KERN, NF_MUL0, LM, LDS);                \
    else if (nf == NF_MUL) SNNQP_CONV_LAUNCH_NF(KERN, NF_MUL, LM, LDS);             \
    else if (nf == NF_DIV) SNNQP_CONV_LAUNCH_NF(KERN, NF_DIV, LM, LDS);             \
    else SNNQP_CONV_LAUNCH_NF(KERN, NF_DECAY, LM, LDS);                             \
  } while (0)
  if (in_type == SNNQP_BITS) {
    // conv3x3_bits.hip: codes exact in fp6 -> f8f6f4 MFMA, wider codes -> int8 MFMA
    // (min_current_bits covers |acc| <= abs_sum_max, table or not)
    const bool fma = nf == NF_MUL0 && w->min_current_bits != 0 && w->abs_sum_max > 0 &&
                     lif_fma_is_exact(w->min_current_bits, a.nrn.k_log2, T, u0 != nullptr);
    // dequantisation of the bits kernel: one multiply when L == 1 (2-bit DuQ, the step
    // quantisers), else the current is read from an LDS table at the address the accumulator
    // spells (fp6 instruction, |acc| <= abs_sum_max <= DQT_MAXA; conv3x3_bits.hip), else the
    // three-instruction form; BatchNorm is the multiply alone when the caller knows every mean
    // and bias is zero
    const bool i8 = !(w->code_max > 0 && w->code_max <= 7);
    // (the table form rests on the matrix pipe adding float32 denormals exactly: probed once
    // per device, runtime.hip; a device that does not gets the arithmetic form, same results)
    int dq = conv3x3_bits_dequant_form(w, nrn);
    if (dq == DQ_TABLE && !dq_table_trusted(stream_device(st), st)) dq = DQ_ARITH;
    const bool tab = dq == DQ_TABLE;
    const bool bnf = (a.bn.flags & (SNNQP_BN_MEAN_ZERO | SNNQP_BN_BIAS_ZERO)) ==
                     (SNNQP_BN_MEAN_ZERO | SNNQP_BN_BIAS_ZERO);
    a.lut_bound = tab ? (int32_t)w->abs_sum_max : 0;
    if (tab) check_code_bound_once(stream_device(st), (const int8_t *)w->w, (int64_t)9 * g->Cin, g->Cout,
                                   w->abs_sum_max, st);
    launch_conv3x3_bits(a, i8, nf, pl, dq, fma, bnf, gy, st);
  } else {
    const int lm = lutc ? LUT_CHANNEL : lut ? LUT_SHARED : LUT_NONE;
    if (lut) check_code_bound_once(stream_device(st), (const int8_t *)w->w, (int64_t)9 * g->Cin, g->Cout,
                                   w->abs_sum_max, st);
    // LDS decides how many workgroups share a CU (every variant needs < 128 VGPRs: up to four
    // waves per SIMD): stage fewer timesteps per pass rather than lose a workgroup to LDS --
    // four per CU measured 7.1 ms on the headline shape, three 7.8, two 11.2.  Any chunk length
    // works (the staging loops test every timestep against it), so the chunk is the largest
    // that fits: at T = 20 all of it in ONE pass (5.35 ms against 5.57 for 16 + 4: a staging
    // phase, a flush and three barriers fewer per patch), at T = 50 20 + 20 + 10.
    const size_t lds_rest = lds_fixed - (size_t)a.tchunk * HIMG2 + u8c2_table_bytes(lm, a.lut_bound, a.lut_rows);
    for (int wgs = 4; wgs >= 2; --wgs) {
      const size_t per_wg = (size_t)(160 * 1024) / wgs - 512;
      int tc = a.tchunk;
      while (tc > 16 && (size_t)tc * HIMG2 + lds_rest > per_wg) tc -= 1;
      if ((size_t)tc * HIMG2 + lds_rest <= per_wg) {
        a.tchunk = tc;
        break;
      }
    }
    const size_t ldsb = (size_t)a.tchunk * HIMG2 + lds_rest;
    // nothing carried in or out and every timestep in one chunk: the variant whose potentials start
    // their life behind the staging code (template parameter ONE)
    const bool one = !a.u0 && !a.u_out && a.T <= a.tchunk;
    if (lutc) SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_CHANNEL, ldsb);
    else if (lut) SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_SHARED, ldsb);
    else SNNQP_CONV_LAUNCH(conv3x3_u8c2_kernel, LUT_NONE, ldsb);
  }
#undef SNNQP_CONV_LAUNCH
#undef SNNQP_CONV_LAUNCH_NF
#undef SNNQP_CONV_LAUNCH_IN
  SNNQP_CHECK_LAUNCH("conv3x3 mfma kernel");
  return SNNQP_OK;
}

}  // namespace snnqp

// ---- work-queue bookkeeping visible to the binding -----------------------------------------
extern "C" int snnqp_workqueue_capture_mark(int device, int64_t *mark) {
  using namespace snnqp;
  SNNQP_REQUIRE(device >= 0 && device < 64 && mark, SNNQP_EINVAL, "workqueue_capture_mark: bad argument");
  std::lock_guard<std::mutex> lock(g_sched_mu);
  *mark = (int64_t)g_sched[device].capture_log.size();
  return SNNQP_OK;
}

extern "C" int snnqp_workqueue_capture_release(int device, int64_t mark_begin, int64_t mark_end) {
  using namespace snnqp;
  SNNQP_REQUIRE(device >= 0 && device < 64, SNNQP_EINVAL, "workqueue_capture_release: bad device");
  std::lock_guard<std::mutex> lock(g_sched_mu);
  SchedPool &p = g_sched[device];
  SNNQP_REQUIRE(0 <= mark_begin && mark_begin <= mark_end && mark_end <= (int64_t)p.capture_log.size(),
                SNNQP_EINVAL, "workqueue_capture_release: marks out of range");
  for (int64_t i = mark_begin; i < mark_end; ++i) {
    const int slot = p.capture_log[(size_t)i];
    if (slot < 0) continue;                        // handed back already
    p.capture_free.push_back(slot);
    p.capture_log[(size_t)i] = -1;
  }
  return SNNQP_OK;
}

extern "C" int snnqp_workqueue_stats(int64_t *captured_static, int64_t *busy_static,
                                     int64_t *dequant_fallbacks, int reset) {
  using namespace snnqp;
  if (captured_static) *captured_static = g_static_captured.load(std::memory_order_relaxed);
  if (busy_static) *busy_static = g_static_busy.load(std::memory_order_relaxed);
  if (dequant_fallbacks) *dequant_fallbacks = dq_table_fallbacks(reset != 0);
  if (reset) { g_static_captured = 0; g_static_busy = 0; }
  return SNNQP_OK;
}

extern "C" int snnqp_debug_workqueue_poke(int device, int64_t mark, int word, uint32_t value) {
  using namespace snnqp;
  SNNQP_REQUIRE(device >= 0 && device < 64 && word >= 0 && word < SCHED_WORDS, SNNQP_EINVAL,
                "debug_workqueue_poke: bad argument");
  std::lock_guard<std::mutex> lock(g_sched_mu);
  SchedPool &p = g_sched[device];
  SNNQP_REQUIRE(p.words && mark >= 0 && mark < (int64_t)p.capture_log.size() && p.capture_log[(size_t)mark] >= 0,
                SNNQP_EINVAL, "debug_workqueue_poke: no live capture slot at mark %lld", (long long)mark);
  SNNQP_HIP(hipMemcpy(p.words + (size_t)p.capture_log[(size_t)mark] * SCHED_WORDS + word, &value, 4,
                      hipMemcpyHostToDevice));
  return SNNQP_OK;
}
