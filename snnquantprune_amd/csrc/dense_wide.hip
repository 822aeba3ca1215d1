// The dense head of the reference model (examples/tcja/models.py:200-255) on the int8 MFMA:
//   SpikingBlock(QuantDense(K -> N1), neuron)                       (one block, any N1), or
//   SpikingBlock(QuantDense(K -> N1)) -> SpikingBlock(QuantDense(N1 -> N2)) -> vote
//                                                                   (the whole head, ONE launch)
// for int8 codes (flax_qdense.py:74-89 after the pack step) over uint8 rows read in place
// (x - 128 against the codes; the 128 * col_sum that gives the sum over x back is what the
// accumulators start from), bit-packed rows, or FLOAT32 rows as the reference hands them over
// (flax_qdense.py:67 casts every input to float32): staged in place -- sixteen bytes of four
// values per lane and load, converted to the same x - 128 bytes on their way into LDS and checked
// while they wait in registers (an integer in [0, 255]?  else SNNQP_FLAG_NOT_INTEGER is OR-ed into
// the launch's flag word and the caller's predicated float32 launch redoes the block).
//
// What differs from dense_mfma.hip (128 columns per workgroup, two K groups, the int32 tile
// transposed through LDS for the neuron):
//  * a workgroup owns its rows for ALL the columns of a 256 / 512-column block: 8 waves x CT
//    column tiles, no K split.  Every uint8 row is read from HBM once (N1 = 512 used to read it
//    four times), a wave re-uses each A fragment for CT MFMAs and each B fragment for RT, and
//    the code stream a CU pulls through its vector L1 -- the bound of the older kernel, 64 B/clk
//    against 2 MFMAs -- meets RT x CT = 8 MFMAs per 2 KiB;
//  * the rows of a sample are laid out so that ONE LANE HALF holds all of them: row tile r,
//    register i of the 32x32 C/D layout is row 32 r + 8 (i >> 2) + 4 h + (i & 3) for the lanes
//    of half h = lane >> 5, so half h owns the rows with bit 2 == h, 16 per row tile.  The
//    staging puts timestep t of the half's sample number j at the half's row k = j T + t: the
//    neuron then walks its OWN accumulator registers in time order -- no transpose through LDS,
//    no barrier between the contraction and the neuron.  Lane = output feature, as in the
//    conv kernels: the ballot of a compare is the packed spike word of two samples;
//  * the hidden raster stays in LDS as bits (2 KiB .. 8 KiB); the second block expands them to
//    {0, 1} bytes as it reads its A fragments, runs the same neuron walk and leaves spike
//    counts / T for the vote (models.py:253-255), which the workgroup finishes;
//  * batches that fill at most half the chip run TWO workgroups per tile of samples, 256 hidden
//    columns each, which hand their halves of the hidden raster over through a caller workspace
//    (write-through stores, a ticket; the last arriver runs the second block and the vote).
//
// K loop as in dense_fp6.hip: chunks of 128 k, three LDS images (chunk c computes from image
// c mod 3 while chunk c + 2 is staged, the barrier waits with a counted lgkmcnt so that the
// next chunk's first fragments stay in flight), staged rows requested four chunks ahead, B
// fragments in a ring of RB k-steps, every wave interleaving its MFMAs with its share of the
// loads and of the staging slot by slot.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace snnqp {

// v_writelane_b32 through the LLVM intrinsic (conv_tile.h: the compiler then knows the wait
// states a VALU-written SGPR needs before it)
extern "C" __device__ uint32_t snnqp_writelane_i32(uint32_t, uint32_t, uint32_t)
    __asm("llvm.amdgcn.writelane.i32");

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const uint16_t lds_cu16_t;

constexpr int W_BK = 128;         // bytes of a row per chunk (int8: 128 k)
constexpr int W_KSC = W_BK / 32;  // MFMA k-steps per chunk
constexpr int W_NBUF = 3;
constexpr int W_THREADS = 512;
constexpr int W_PF = 3;           // A fragments in flight
constexpr int W_S1P = 17;         // words per row of the hidden raster in LDS (16 + 1: no bank conflict)
constexpr int W_VOTE_SB = 16;     // the fused head keeps at most this many samples per workgroup

struct DenseWideArgs {
  const void *x;                  // uint8 rows (bytes) or bit-packed rows (words)
  int64_t xs_t, xs_b;             // byte (u8) / word (bits) strides
  int32_t T, B, K, N, KS, KW;     // KS = ceil(K / 32) k-steps, KW = ceil(K / 32) words of a bit row
  int32_t SPH;                    // samples per lane half (a workgroup holds 2 SPH samples)
  const int8_t *wt;               // MFMA tiles [Npad/32][KS][64][16]
  const int32_t *col_sum;
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  uint32_t *s_out;                // nullable in the fused head
  // second block + vote (fused head)
  int32_t N2, KS2, group;
  const int8_t *wt2;
  Dequant dq2;
  NeuronP nrn2;
  uint32_t *s2_out;               // nullable
  float *logits;
  // fused head with the hidden columns split over two workgroups per tile (blockIdx.y): each
  // leaves its half of the hidden raster in the workspace, the last arriver runs the rest
  int32_t csplit;
  uint32_t *hs_tickets;           // [tiles], zeroed on the stream in front of every launch
  uint32_t *hs_raster;            // [tiles][2][ROWS][8 words]
  uint32_t *status;               // the device's status word (runtime.hip), or null
  int32_t *x_flags;               // float32 rows: OR-ed with SNNQP_FLAG_NOT_INTEGER (zeroed by the launcher)
};

typedef float v4f __attribute__((ext_vector_type(4)));

// four float32 values -> the four bytes x - 128 of the int8 operand; `bad` collects what is not
// an integer in [0, 255]: fl(cvt(x)) - x is +0.0 exactly for the integers v_cvt_u32_f32 can hold
// (-0.0 counts as 0), a NaN for a NaN, non-zero otherwise; the range is checked on the integers
__device__ __forceinline__ uint32_t f32x4_to_i8x4(const v4f &f, uint32_t &bad) {
  const uint32_t u0 = __float2uint_rz(f.x), u1 = __float2uint_rz(f.y), u2 = __float2uint_rz(f.z),
                 u3 = __float2uint_rz(f.w);
  const float d0 = __uint2float_rn(u0) - f.x, d1 = __uint2float_rn(u1) - f.y,
              d2 = __uint2float_rn(u2) - f.z, d3 = __uint2float_rn(u3) - f.w;
  bad |= __float_as_uint(d0) | __float_as_uint(d1);
  bad |= __float_as_uint(d2) | __float_as_uint(d3);
  bad |= (u0 | u1 | u2 | u3) >> 8;
  return (u0 | (u1 << 8) | (u2 << 16) | (u3 << 24)) ^ 0x80808080u;
}

__device__ __forceinline__ void wide_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 16-byte piece c (of 8) of row `row` (dense_fp6.hip: the ds_read_b128 of an A fragment and
// the ds_write_b128 of the staging are conflict-free)
__device__ __forceinline__ int wa_addr(int row, int c) {
  return row * W_BK + ((c ^ ((row >> 1) & 7)) << 4);
}

// row of the MFMA tile set <-> (lane half h, the half's row k)
__device__ __forceinline__ int rho_of(int k, int h) {
  return (k >> 4) * 32 + ((k >> 2) & 3) * 8 + h * 4 + (k & 3);
}

__device__ __forceinline__ v4i expand16w(uint32_t b) {
  v4i o;
  o.x = (int)((((b >> 0) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.y = (int)((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.z = (int)((((b >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.w = (int)((((b >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

// u += (x - (u - v_reset)) * m ; s = (u - v_th) >= 0 ; hard reset  (spiking_learning.py:410-414,
// :381-385) when FASTN, else the neuron of common.h
template <bool FASTN>
__device__ __forceinline__ bool wide_neuron(float &u, float x, const NeuronP &p, float dec) {
  if constexpr (FASTN) {
    const float d = x - (u - p.vr);
    u = u + d * p.inv_k;
    const bool s = (u - p.vth) >= 0.0f;
    u = s ? p.vr : u;
    return s;
  } else {
    return neuron_step(u, x, p, dec);
  }
}

// The neuron of one wave's column tiles over the rows its lanes own: acc[r][ct] in the MFMA
// C/D layout, half h walks its rows k = 0 .. 16 RT - 1 in order = (sample j = k / T, t = k % T).
// words[]: bit-packed spikes, word (k, ct, half) in lane / register (k CT + ct) 2 + half.
// per_update(ct, spike, t, sample, live): called after every update (the fused head counts with it).
template <int RT, int CT, bool FASTN, bool HASBN, typename F>
__device__ __forceinline__ void neuron_walk(const v16i (&acc)[RT][CT], const int (&off)[CT],
                                            const bool (&col_live)[CT], const int (&col)[CT],
                                            const Dequant &dq, const BnP &bn, const NeuronP &nrn,
                                            int T, int SPH, int nsamp, int b0, int N,
                                            const float *u0, float *u_out, int h,
                                            uint32_t (&words)[(16 * RT * CT * 2 + 63) / 64],
                                            F &&per_update) {
  float bmean[CT], bmul[CT], bbias[CT], dec[CT], u[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    bmean[ct] = 0.f; bmul[ct] = 1.f; bbias[ct] = 0.f; dec[ct] = 0.f; u[ct] = 0.f;
    if (col_live[ct]) {
      if (HASBN) { bmean[ct] = bn.mean[col[ct]]; bmul[ct] = bn.mul[col[ct]]; bbias[ct] = bn.bias[col[ct]]; }
      if (nrn.kind == SNNQP_NEURON_LIF) dec[ct] = nrn.decay[col[ct]];
    }
  }
  int t = 0, j = 0;                               // wave-uniform
#pragma unroll
  for (int k = 0; k < 16 * RT; ++k) {
    const int s = 2 * j + h;                      // sample of this lane half
    const bool slive = j < SPH && s < nsamp;
    if (t == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        u[ct] = (u0 && slive && col_live[ct]) ? u0[(int64_t)(b0 + s) * N + col[ct]] : 0.0f;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float cur = dequant_acc(acc[k >> 4][ct][k & 15] + off[ct], dq);
      if (HASBN) cur = bn_apply(cur, bmean[ct], bmul[ct], bbias[ct]);
      const bool sp = wide_neuron<FASTN>(u[ct], cur, nrn, dec[ct]) && slive && col_live[ct];
      const unsigned long long m = __ballot(sp);
      const int idx = (k * CT + ct) * 2;
      words[idx >> 6] = snnqp_writelane_i32((uint32_t)m, (uint32_t)(idx & 63), words[idx >> 6]);
      words[(idx + 1) >> 6] = snnqp_writelane_i32((uint32_t)(m >> 32), (uint32_t)((idx + 1) & 63),
                                                  words[(idx + 1) >> 6]);
      per_update(ct, sp, t, s, slive);
    }
    if (t == T - 1 && u_out) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        if (slive && col_live[ct]) u_out[(int64_t)(b0 + s) * N + col[ct]] = u[ct];
    }
    ++t;
    if (t == T) { t = 0; ++j; }
  }
}

// the reset of the fast walk: u with the lanes of `mask` zeroed (conv_tile.h reset_where)
__device__ __forceinline__ float zero_where(float u, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(u), "s"(mask));
  return r;
}

// The same walk for the neuron every shipped config uses -- u += (x - u) m with v_reset = 0
// (multi_step_LIF with tau a power of two, parametric_leaky_IF), no carried-in state, none
// returned -- as straight-line code.  A wave's instruction stream is what the walk costs (two
// waves per SIMD, one dependent chain per column tile: about 4.5 cycles per instruction, vector
// or scalar), so everything that is not the update itself is taken out of the per-row code:
//  * the reset at a sample's first step is the hard reset of the step before it, forced at
//    t = T - 1 by an AND with a per-row word (0 at a sample's last row, else all ones: lane k of
//    `vnotlast`, one v_readlane per row) -- v_reset = 0 = the initial potential;
//  * sample and column liveness are applied to the packed words afterwards, 64 words at a time
//    (mask_words), not to every ballot;
//  * (u - v_th) >= 0 <=> u >= v_th with float32 subnormals kept (hipcc's default; conv_tile.h).
// Eleven vector instructions per update: cvt, 3 dequantise, sub, mul, add, cmp, select, and, and
// two v_writelane per 64 updates' ballot.  on_end(ct, cnt, k): after a sample's last row, with
// the lane's spike count of that sample when COUNT.
template <int RT, int CT, bool HASBN, bool COUNT, typename F>
__device__ __forceinline__ void neuron_walk_fast(const v16i (&acc)[RT][CT],
                                                 const unsigned long long (&colmask)[CT],
                                                 const int (&col)[CT], const Dequant &dq,
                                                 const BnP &bn, float inv_k, float vth, int T,
                                                 int nrows,
                                                 uint32_t (&words)[(16 * RT * CT * 2 + 63) / 64],
                                                 F &&on_end) {
  float bmean[CT], bmul[CT], bbias[CT], u[CT];
  int cnt[CT];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    bmean[ct] = 0.f; bmul[ct] = 1.f; bbias[ct] = 0.f; u[ct] = 0.f; cnt[ct] = 0;
    if (HASBN && ((colmask[ct] >> (lane & 31)) & 1ull)) {
      bmean[ct] = bn.mean[col[ct]]; bmul[ct] = bn.mul[col[ct]]; bbias[ct] = bn.bias[col[ct]];
    }
  }
  const int vnotlast = (lane % T == T - 1) ? 0 : -1;      // lane k: row k of a lane half
#pragma unroll
  for (int k = 0; k < 16 * RT; ++k) {
    if (k < nrows) {                  // (rows behind the last sample of a half: T = 20, RT 2: 12 of 32)
    const int nl = __builtin_amdgcn_readlane(vnotlast, k);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float cur = dequant_acc(acc[k >> 4][ct][k & 15], dq);
      if (HASBN) cur = bn_apply(cur, bmean[ct], bmul[ct], bbias[ct]);
      const float d = cur - u[ct];
      const float un = u[ct] + d * inv_k;
      const bool sp = un >= vth;
      const unsigned long long m = __ballot(sp);
      u[ct] = __builtin_bit_cast(float, __builtin_bit_cast(int, zero_where(un, m)) & nl);
      if (COUNT) cnt[ct] += sp ? 1 : 0;
      const int idx = (k * CT + ct) * 2;
      words[idx >> 6] = snnqp_writelane_i32((uint32_t)m, (uint32_t)(idx & 63), words[idx >> 6]);
      words[(idx + 1) >> 6] = snnqp_writelane_i32((uint32_t)(m >> 32), (uint32_t)((idx + 1) & 63),
                                                  words[(idx + 1) >> 6]);
    }
    if (COUNT && nl == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { on_end(ct, cnt[ct], k); cnt[ct] = 0; }
    }
    }
  }
}

// words of neuron_walk_fast -> the words of live samples and live columns (the rest zero)
template <int RT, int CT>
__device__ __forceinline__ void mask_words(uint32_t (&words)[(16 * RT * CT * 2 + 63) / 64],
                                           const unsigned long long (&colmask)[CT], int T, int SPH,
                                           int nsamp) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int v = 0; v < (16 * RT * CT * 2 + 63) / 64; ++v) {
    const int idx = v * 64 + lane;
    const int hh = idx & 1, ct = (idx >> 1) % CT, k = idx / (2 * CT);
    // k / T without an integer division: k < 64, T <= 64, so (k + 0.5) / T is at least 1 / 128 away
    // from an integer and the float quotient truncates to the right one
    const int j = (int)(((float)k + 0.5f) * (1.0f / (float)T));
    uint32_t cw = (uint32_t)colmask[0];
#pragma unroll
    for (int c = 1; c < CT; ++c) cw = ct == c ? (uint32_t)colmask[c] : cw;
    words[v] &= (j < SPH && 2 * j + hh < nsamp) ? cw : 0u;
  }
}

}  // namespace

template <int RT, int CT, int IN, bool FUSE>
__global__ void __launch_bounds__(W_THREADS)
dense_wide_kernel(DenseWideArgs a) {
  constexpr bool F32IN = IN == SNNQP_F32;
  constexpr bool U8 = IN == SNNQP_U8 || F32IN;     // the operand bytes are x - 128
  constexpr int ROWS = RT * 32;
  constexpr int ABYTES = ROWS * W_BK;
  // staging tasks of a chunk: one 16-byte piece of a uint8 row, or one 32-bit word of a bit row
  constexpr int WPR = U8 ? W_BK / 16 : W_BK / 32;
  constexpr int NTASK = ROWS * WPR;
  constexpr int TPT = (NTASK + W_THREADS - 1) / W_THREADS;
  typedef typename std::conditional<U8, v4i, uint32_t>::type stg_t;
  // float32 rows: NU 16-byte loads per thread and chunk; load j of wave w covers the chunk's 512
  // bytes of the two rows rho = 16 j + 2 w + (lane >> 5), four values per lane (1 KiB, coalesced)
  constexpr int NU = F32IN ? 2 * RT : 0;
  // ... of which W_FRING are in flight per thread (the unrolled body of three chunks holds 3 NU
  // units: a multiple of the ring, so that every ring index is an immediate).  W_FRING x 16 B x 512
  // threads is what a CU has on its way from HBM: 48 KiB at 6 -- the stream then runs at
  // (bytes in flight) / (HBM latency under load, ~2.2 us) = 22 GB/s per CU, 5.6 TB/s over the chip
#ifndef SNNQP_W_FRING
  constexpr int W_FRING = RT == 3 ? 9 : 6;
#else
  constexpr int W_FRING = SNNQP_W_FRING;
#endif
  static_assert(!F32IN || (3 * NU) % W_FRING == 0, "ring must divide the unrolled body");
  constexpr int NFRAG = W_KSC * RT;               // A fragments of a chunk
  constexpr int NSLOT = NFRAG * CT;               // MFMAs of a chunk
  // B ring, k-steps (the unrolled body is 12 long).  float32 rows: the loop runs at the pace of
  // HBM, a look-ahead of three k-steps covers the L2 round trip, and the sixteen registers pay for
  // the row loads in flight (RT 4 x CT 2 spilled 450 bytes with a ring of 6)
#ifndef SNNQP_W_RBF
  constexpr int RB = RT * CT >= 6 ? (F32IN ? 3 : 6) : 12;
#else
  constexpr int RB = RT * CT >= 6 ? (F32IN ? SNNQP_W_RBF : 6) : 12;
#endif
  constexpr int NW = (16 * RT * CT * 2 + 63) / 64;
  // (the fused head overlays the dead A images: raster | count / T per (sample, feature) | a flag)
  constexpr int LDSB = W_NBUF * ABYTES > ROWS * W_S1P * 4 + W_VOTE_SB * 512 + 16 ? W_NBUF * ABYTES
                                                                                : ROWS * W_S1P * 4 + W_VOTE_SB * 512 + 16;
  __shared__ __attribute__((aligned(128))) uint8_t lds[LDSB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int SB = 2 * a.SPH;
  const int b0 = blockIdx.x * SB;
  const int nsamp = min(SB, a.B - b0);
  const int wl = wave * CT;                           // first column tile of the wave within the block
  const int nb0 = blockIdx.y * 8 * CT + wl;           // ... within the layer
  const int NB = (a.N + 31) >> 5;
  const int nchunks = (a.KS + W_KSC - 1) / W_KSC;
  const int nsteps = (nchunks + W_NBUF - 1) / W_NBUF * W_NBUF;
  // every workgroup streams the same code tiles; the walk over K starts at another chunk on
  // every XCD (exact integer sums may be taken in any order; dense_fp6.hip)
  const int rot = (int)(((blockIdx.x & 7u) * (unsigned)nchunks) >> 3);
  auto phys = [&](int lc) -> int {
    const int pc = lc + rot >= nchunks ? lc + rot - nchunks : lc + rot;
    return lc < nchunks ? pc : nchunks;               // steps beyond the last chunk: a dead chunk
  };

  // lane = output feature.  uint8 rows enter as x - 128: the 128 * col_sum that gives the sum
  // over x back is what every accumulator of the lane starts from
  int col[CT];
  bool col_live[CT];
  int acc0[CT];
  v16i acc[RT][CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    col[ct] = (nb0 + ct) * 32 + n;
    col_live[ct] = col[ct] < a.N;
    acc0[ct] = (U8 && col_live[ct]) ? 128 * a.col_sum[col[ct]] : 0;
  }
  // (float32 rows: the accumulators are set behind the prologue, whose sixteen row loads in
  // flight live in the registers they will occupy)
  auto set_acc = [&]() {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int o = acc0[ct];
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r][ct] = v16i{o, o, o, o, o, o, o, o, o, o, o, o, o, o, o, o};
    }
  };
  if constexpr (!F32IN) set_acc();

  // ---- staging tasks: the same (row, piece) for every chunk -------------------------------
  uint32_t roff[F32IN ? NU : TPT];
  uint32_t rmask[F32IN ? 1 : TPT];
  int wr_off[F32IN ? 1 : TPT];
  uint32_t bad = 0;
  const uint8_t *xb = (const uint8_t *)a.x + (U8 ? (int64_t)b0 * a.xs_b * (F32IN ? 4 : 1) : 0);
  const uint32_t *xw = (const uint32_t *)a.x + (U8 ? 0 : (int64_t)b0 * a.xs_b);
  if constexpr (F32IN) {
    rmask[0] = 0u;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int rho = 16 * j + 2 * wave + h;
      const int wv = rho & 31;
      const int hh = (wv >> 2) & 1, kk = (rho >> 5) * 16 + (wv >> 3) * 4 + (wv & 3);
      const int jj = kk / a.T, tt = kk - jj * a.T, ss = 2 * jj + hh;
      const bool live = jj < a.SPH && ss < nsamp;
      rmask[0] |= live ? 1u << j : 0u;
      // byte offset of the row (< 2^31: launch check)
      roff[j] = live ? (uint32_t)(((int64_t)tt * a.xs_t + (int64_t)ss * a.xs_b) * 4) : 0u;
    }
    // rho >> 1 = 8 j + wave: the swizzle of wa_addr does not depend on j, a row pair is 2 KiB on
    wr_off[0] = wa_addr(2 * wave + h, n >> 2) + (n & 3) * 4;
  }
#pragma unroll
  for (int q = 0; q < (F32IN ? 0 : TPT); ++q) {
    const int task = tid + q * W_THREADS;
    const int rho = (task / WPR) % ROWS, wi = task % WPR;
    const int wv = rho & 31;
    const int hh = (wv >> 2) & 1, kk = (rho >> 5) * 16 + (wv >> 3) * 4 + (wv & 3);
    const int jj = kk / a.T, tt = kk - jj * a.T, ss = 2 * jj + hh;
    const bool live = task < NTASK && jj < a.SPH && ss < nsamp;
    rmask[q] = live ? 0xFFFFFFFFu : 0u;
    roff[q] = live ? (uint32_t)((int64_t)tt * a.xs_t + (int64_t)ss * a.xs_b) : 0u;   // < 2^31: launch check
    wr_off[q] = wa_addr(rho, U8 ? wi : wi * 2);
  }
  auto chunk_word = [&](int chunk, int q) { return chunk * WPR + (tid + q * W_THREADS) % WPR; };
  auto stage_load1 = [&](int chunk, int q) -> stg_t {
    if constexpr (U8) {
      const int piece = chunk_word(chunk, q);
      return *(const v4i *)(xb + roff[q] + (uint32_t)(piece * 16 < a.K ? piece * 16 : 0));
    } else {
      return xw[roff[q] + (uint32_t)min(chunk_word(chunk, q), a.KW - 1)];
    }
  };
  // one task's share of a chunk into LDS image `buf` (dead rows and k beyond K: zero bytes)
  auto stage_store1 = [&](const stg_t &v, int chunk, int q, int buf) {
    if (tid + q * W_THREADS >= NTASK) return;
    uint8_t *base = lds + buf * ABYTES;
    if constexpr (U8) {
      const int m = (chunk_word(chunk, q) * 16 < a.K ? -1 : 0) & (int)rmask[q];
      *(v4i *)(base + wr_off[q]) = v4i{(v.x ^ (int)0x80808080) & m, (v.y ^ (int)0x80808080) & m,
                                       (v.z ^ (int)0x80808080) & m, (v.w ^ (int)0x80808080) & m};
    } else {
      const uint32_t w = v & rmask[q] & (uint32_t)((chunk_word(chunk, q) - a.KW) >> 31);
      *(v4i *)(base + wr_off[q]) = expand16w(w & 0xFFFFu);
      *(v4i *)(base + (wr_off[q] ^ 16)) = expand16w(w >> 16);     // piece 2 wi + 1: the swizzle flips bit 4 only
    }
  };

  // float32 rows: unit (chunk, j).  A chunk beyond K (or the dead chunk behind the last one)
  // loads the row's first bytes and stores zeros
  auto f32_load = [&](int chunk, int j) -> v4f {
    const uint32_t cb = (uint32_t)(chunk * W_BK + n * 4) * 4u;
    return *(const v4f *)(xb + roff[j] + (chunk * W_BK + n * 4 < a.K ? cb : 0u));
  };
  auto f32_store = [&](const v4f &v, int chunk, int j, int buf) {
    uint32_t flag = 0;
    const uint32_t w = f32x4_to_i8x4(v, flag);
    const bool on = ((rmask[0] >> j) & 1u) && chunk * W_BK + n * 4 < a.K;
    bad |= on ? flag : 0u;
    *(uint32_t *)(lds + buf * ABYTES + wr_off[0] + j * 2048) = on ? w : 0u;
  };
  v4f fring[F32IN ? W_FRING : 1];

  // ---- B fragments: ring of RB k-steps ------------------------------------------------------
  const v4i *wtile[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    wtile[ct] = (const v4i *)a.wt + ((int64_t)(nb0 + ct < NB ? nb0 + ct : 0) * a.KS) * 64 + lane;
  v4i bring[RB][CT];
  // k-step `ks` (may run past W_KSC: into the following chunks) of loop step `lc`
  auto load_b1 = [&](v4i &dst, int lc, int ks, int ct) {
    const int lcl = lc + ks / W_KSC;
    const int kg = min(phys(lcl) * W_KSC + ks % W_KSC, a.KS - 1);   // beyond K: zero A bytes, any codes do
    dst = wtile[ct][(int64_t)kg * 64];
  };

  int rd_off[W_KSC];
#pragma unroll
  for (int ks = 0; ks < W_KSC; ++ks) rd_off[ks] = wa_addr(n, ks * 2 + h);
  // fragment f of a chunk: k-step f / RT, row tile f % RT
  auto frag = [&](int buf, int f) -> v4i {
    return *(const v4i *)(lds + buf * ABYTES + (f % RT) * 32 * W_BK + rd_off[f / RT]);
  };

  stg_t stgr[F32IN ? 1 : W_NBUF][F32IN ? 1 : TPT];
  v4i av[W_PF + 1];
  constexpr int XSLOT = (NFRAG - W_PF) * CT > 0 ? (NFRAG - W_PF) * CT : 1;   // slots that may write LDS
  // One chunk = NSLOT slots; slot s = (k-step, row tile, column tile): the MFMA, then -- at the
  // first column tile of a fragment -- the read of the fragment W_PF ahead (the last W_PF of a
  // chunk come from the NEXT chunk's image), and an even share of: the B loads of the k-step
  // RB - 1 ahead, the row loads of the chunk four ahead, the staging of the chunk two ahead.
  auto fused_chunk = [&](int i, int lc) {
    const int rbuf = i, nbuf = (i + 1) % W_NBUF, wbuf = (i + 2) % W_NBUF;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int f = s / CT, ct = s % CT, ks = f / RT, r = f % RT;
      acc[r][ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[f % (W_PF + 1)], bring[(i * W_KSC + ks) % RB][ct],
                                                         acc[r][ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ct == 0) {
        if (f + W_PF < NFRAG) av[(f + W_PF) % (W_PF + 1)] = frag(rbuf, f + W_PF);
        else av[(f + W_PF) % (W_PF + 1)] = frag(nbuf, f + W_PF - NFRAG);
      }
      // B fragments of k-step ks + RB - 1: after this k-step's first MFMAs, one per slot
      if (r == 0) {
        load_b1(bring[(i * W_KSC + ks + RB - 1) % RB][ct], lc, ks + RB - 1, ct);
      }
      if constexpr (F32IN) {
        // float32 rows: unit g of chunk lc + 2 leaves its register for image (i + 2) % 3, and the
        // register takes the unit W_FRING further on (the unrolled body of three chunks holds
        // 3 NU units, a multiple of the ring: every index below is an immediate)
#pragma unroll
        for (int gq = 0; gq < NU; ++gq)
          if (s == (gq * NSLOT) / NU) {
            const int slot = (i * NU + 2 * NU + gq) % W_FRING;
            f32_store(fring[slot], phys(lc + 2), gq, wbuf);
            const int ahead = gq + W_FRING;
            fring[slot] = f32_load(phys(lc + 2 + ahead / NU), ahead % NU);
          }
      } else {
      // rows of chunk lc + 4 (ring slot (i + 1) % 3: its chunk lc + 1 was staged a step ago)
#pragma unroll
      for (int q = 0; q < TPT; ++q)
        if (s == (q * NSLOT) / TPT + (NSLOT > 1 ? 1 : 0))
          stgr[(i + 1) % W_NBUF][q] = stage_load1(phys(lc + 4), q);
      // staging of chunk lc + 2 into image (i + 2) % 3
#pragma unroll
      for (int q = 0; q < TPT; ++q)
        if (s == (q * XSLOT) / TPT) stage_store1(stgr[(i + 2) % W_NBUF][q], phys(lc + 2), q, wbuf);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // LDS operations of a wave complete in order and the W_PF youngest are the next chunk's
    // fragment reads: this retires every write of the chunk without draining those
    asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" : : "n"(W_PF) : "memory");
  };

  // ---- prologue ------------------------------------------------------------------------------
  if constexpr (F32IN) {
    // chunks 0 and 1 into their images -- all 2 NU loads in flight at once (one round trip to HBM
    // instead of three: 16 k -> 8 k cycles per tile) --, then the ring as step 0 expects it: the
    // first units of chunk 2 in flight
    {
      v4f pro[2 * NU];
#pragma unroll
      for (int u = 0; u < 2 * NU; ++u) pro[u] = f32_load(phys(u / NU), u % NU);
#pragma unroll
      for (int ks = 0; ks < RB - 1; ++ks)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) load_b1(bring[ks][ct], 0, ks, ct);
#pragma unroll
      for (int u = 2 * NU; u < 2 * NU + W_FRING; ++u) fring[u % W_FRING] = f32_load(phys(u / NU), u % NU);
#pragma unroll
      for (int u = 0; u < 2 * NU; ++u) f32_store(pro[u], phys(u / NU), u % NU, u / NU);
    }
    set_acc();
  } else {
#pragma unroll
  for (int d = 0; d < W_NBUF; ++d)
#pragma unroll
    for (int q = 0; q < TPT; ++q) stgr[d][q] = stage_load1(phys(d), q);
#pragma unroll
  for (int ks = 0; ks < RB - 1; ++ks)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) load_b1(bring[ks][ct], 0, ks, ct);
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int q = 0; q < TPT; ++q) stage_store1(stgr[d][q], phys(d), q, d);
#pragma unroll
  for (int q = 0; q < TPT; ++q) stgr[0][q] = stage_load1(phys(3), q);
  }
  wide_barrier();
#pragma unroll
  for (int f = 0; f < W_PF; ++f) av[f] = frag(0, f);
  for (int c = 0; c < nsteps; c += W_NBUF) {
#pragma unroll
    for (int i = 0; i < W_NBUF; ++i)
      if (c + i < nchunks) fused_chunk(i, c + i);   // (the chunks beyond K are the last ones)
  }
  wide_barrier();                                   // every wave is done with the A images
  if constexpr (F32IN) {
    if (__ballot(bad != 0u) != 0ull && lane == 0 && a.x_flags) atomicOr(a.x_flags, SNNQP_FLAG_NOT_INTEGER);
  }

  // (fused head: the second block's B fragments are requested now, they land during the walk)
  v4i b2[FUSE ? 16 : 1];
  if constexpr (FUSE) {
    const int NB2 = (a.N2 + 31) >> 5;
    const v4i *wt2 = (const v4i *)a.wt2 + ((int64_t)(wave < NB2 ? wave : 0) * a.KS2) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) b2[ks] = wt2[(int64_t)min(ks, a.KS2 - 1) * 64];
  }
  // ---- neuron of the first block, from the accumulator registers ----------------------------
  uint32_t *s1 = (uint32_t *)lds;                   // hidden raster [ROWS][W_S1P] words
  const int wofs = (FUSE && a.csplit) ? (int)blockIdx.y * 8 : 0;   // this block's words within a row
  float *vbuf = (float *)(lds + ROWS * W_S1P * 4);  // fused head: spike count / T  [SB][128]
  const bool fast1 = a.nrn.kind != SNNQP_NEURON_LIF && a.nrn.inv_k != 0.0f;
  {
    int off[CT];
    unsigned long long colmask[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { off[ct] = 0; colmask[ct] = __ballot(col_live[ct]); }
    uint32_t words[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) words[v] = 0u;
    auto nothing = [](int, bool, int, int, bool) {};
    auto nothing3 = [](int, int, int) {};
    const bool straight = fast1 && a.nrn.vr == 0.0f && !a.u0 && !a.u_out;
    if (nb0 < NB) {
      if (straight) {
        if (a.bn.mean) neuron_walk_fast<RT, CT, true, false>(acc, colmask, col, a.dq, a.bn, a.nrn.inv_k, a.nrn.vth, a.T, a.SPH * a.T, words, nothing3);
        else neuron_walk_fast<RT, CT, false, false>(acc, colmask, col, a.dq, a.bn, a.nrn.inv_k, a.nrn.vth, a.T, a.SPH * a.T, words, nothing3);
        mask_words<RT, CT>(words, colmask, a.T, a.SPH, nsamp);
      } else if (a.bn.mean) {
        if (fast1) neuron_walk<RT, CT, true, true>(acc, off, col_live, col, a.dq, a.bn, a.nrn, a.T, a.SPH, nsamp, b0, a.N, a.u0, a.u_out, h, words, nothing);
        else neuron_walk<RT, CT, false, true>(acc, off, col_live, col, a.dq, a.bn, a.nrn, a.T, a.SPH, nsamp, b0, a.N, a.u0, a.u_out, h, words, nothing);
      } else {
        if (fast1) neuron_walk<RT, CT, true, false>(acc, off, col_live, col, a.dq, a.bn, a.nrn, a.T, a.SPH, nsamp, b0, a.N, a.u0, a.u_out, h, words, nothing);
        else neuron_walk<RT, CT, false, false>(acc, off, col_live, col, a.dq, a.bn, a.nrn, a.T, a.SPH, nsamp, b0, a.N, a.u0, a.u_out, h, words, nothing);
      }
    }
    // word (k, ct, half) -> the raster row of (k, half)
#pragma unroll
    for (int v = 0; v < NW; ++v) {
      const int idx = v * 64 + lane;
      if (idx < 16 * RT * CT * 2) {
        const int hh = idx & 1, ct = (idx >> 1) % CT, k = idx / (2 * CT);
        s1[rho_of(k, hh) * W_S1P + wofs + wl + ct] = words[v];
      }
    }
  }
  wide_barrier();
  if (a.s_out) {
    const int CW = (a.N + 31) >> 5;
    for (int q = tid; q < ROWS * 16; q += W_THREADS) {
      const int rho = q >> 4, w = q & 15;
      const int wv = rho & 31;
      const int hh = (wv >> 2) & 1, kk = (rho >> 5) * 16 + (wv >> 3) * 4 + (wv & 3);
      const int jj = kk / a.T, tt = kk - jj * a.T, ss = 2 * jj + hh;
      const int gw = blockIdx.y * 8 * CT + w;
      if (w < 8 * CT && jj < a.SPH && ss < nsamp && gw < CW)
        a.s_out[((int64_t)tt * a.B + (b0 + ss)) * CW + gw] = s1[rho * W_S1P + wofs + w];
    }
  }
  // ---- column split: hand the half raster over, the last arriver of the tile goes on ----------
  // (cdna_hip_programming.md, hand-off recipe R1: the payload by agent-scope relaxed stores --
  // write-through --, every storing wave drains, one relaxed agent-scope ticket; the last
  // arriver reads the other half with agent-scope loads: no release / acquire fence, which on
  // this part would write back / invalidate an L2)
  if constexpr (FUSE) {
    if (a.csplit) {
      const int tile = blockIdx.x, half = blockIdx.y;
      uint32_t *mine = a.hs_raster + ((int64_t)(tile * 2 + half) * ROWS) * 8;
      for (int q = tid; q < ROWS * 8; q += W_THREADS)
        __hip_atomic_store(mine + q, s1[(q >> 3) * W_S1P + wofs + (q & 7)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      uint32_t *flag = (uint32_t *)(lds + ROWS * W_S1P * 4 + W_VOTE_SB * 512);
      if (tid == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(a.hs_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket >= 2u && a.status) *(volatile uint32_t *)a.status = SNNQP_STATUS_TICKET;
        const bool last = ticket == 1u;
        *flag = last ? 1u : 0u;
      }
      __syncthreads();
      if (*flag == 0u) return;
      const uint32_t *other = a.hs_raster + ((int64_t)(tile * 2 + (1 - half)) * ROWS) * 8;
      for (int q = tid; q < ROWS * 8; q += W_THREADS)
        s1[(q >> 3) * W_S1P + (8 - wofs) + (q & 7)] =
            __hip_atomic_load(other + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wide_barrier();
    }
  }
  // ---- second block + vote --------------------------------------------------------------------
  if constexpr (FUSE) {
    const int NB2 = (a.N2 + 31) >> 5;               // <= 4 column tiles, one wave each
    v16i acc2[RT][1];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc2[r][0] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (wave < NB2) {
      const uint32_t s1h = (uint32_t)(uintptr_t)(lds_cu16_t *)lds + (uint32_t)(n * W_S1P * 4 + h * 2);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks < a.KS2) {
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            // the lane's 16 k of row 32 r + n: halfword 2 ks + h of the raster row
            const uint32_t bits = *(lds_cu16_t *)(uintptr_t)(s1h + (uint32_t)(r * 32 * W_S1P * 4 + ks * 4));
            acc2[r][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(expand16w(bits), b2[ks], acc2[r][0], 0, 0, 0);
          }
        }
      }
    }
    wide_barrier();
    if (wave < NB2) {
      int off[1] = {0}, col2[1] = {wave * 32 + n};
      bool col2_live[1] = {col2[0] < a.N2};
      uint32_t words[(16 * RT * 2 + 63) / 64];
#pragma unroll
      for (int v = 0; v < (16 * RT * 2 + 63) / 64; ++v) words[v] = 0u;
      const int CW2 = (a.N2 + 31) >> 5;
      const float Tf = (float)a.T;
      const bool fast2 = a.nrn2.kind != SNNQP_NEURON_LIF && a.nrn2.inv_k != 0.0f;
      BnP nobn; nobn.mean = nullptr; nobn.mul = nullptr; nobn.bias = nullptr; nobn.flags = 0;
      if (fast2 && a.nrn2.vr == 0.0f) {
        // jnp.mean over T of 0/1 values: the exact count, one division (models.py:253)
        auto sample_done = [&](int, int cnt, int k) {
          const int j = k / a.T, s = 2 * j + h;
          if (j < a.SPH && s < nsamp && col2_live[0]) vbuf[s * 128 + col2[0]] = (float)cnt / Tf;
        };
        const unsigned long long colmask2[1] = {__ballot(col2_live[0])};
        neuron_walk_fast<RT, 1, false, true>(acc2, colmask2, col2, a.dq2, nobn, a.nrn2.inv_k, a.nrn2.vth, a.T, a.SPH * a.T, words, sample_done);
        mask_words<RT, 1>(words, colmask2, a.T, a.SPH, nsamp);
      } else {
        int cnt = 0;
        auto count = [&](int, bool sp, int t, int s, bool slive) {
          cnt += sp ? 1 : 0;
          if (t == a.T - 1) {
            if (slive && col2_live[0]) vbuf[s * 128 + col2[0]] = (float)cnt / Tf;
            cnt = 0;
          }
        };
        if (fast2) neuron_walk<RT, 1, true, false>(acc2, off, col2_live, col2, a.dq2, nobn, a.nrn2, a.T, a.SPH, nsamp, b0, a.N2, nullptr, nullptr, h, words, count);
        else neuron_walk<RT, 1, false, false>(acc2, off, col2_live, col2, a.dq2, nobn, a.nrn2, a.T, a.SPH, nsamp, b0, a.N2, nullptr, nullptr, h, words, count);
      }
      if (a.s2_out) {
#pragma unroll
        for (int v = 0; v < (16 * RT * 2 + 63) / 64; ++v) {
          const int idx = v * 64 + lane;
          if (idx < 16 * RT * 2) {
            const int hh = idx & 1, k = idx >> 1;
            const int jj = k / a.T, tt = k - jj * a.T, ss = 2 * jj + hh;
            if (jj < a.SPH && ss < nsamp)
              a.s2_out[((int64_t)tt * a.B + (b0 + ss)) * CW2 + wave] = words[v];
          }
        }
      }
    }
    wide_barrier();
    // mean over each class's `group` neurons, in order (ops.vote / snnqp_vote)
    const int NC = a.N2 / a.group;
    for (int q = tid; q < nsamp * NC; q += W_THREADS) {
      const int s = q / NC, c = q - s * NC;
      float sum = 0.0f;
      for (int g = 0; g < a.group; ++g) sum = sum + vbuf[s * 128 + c * a.group + g];
      a.logits[(int64_t)(b0 + s) * NC + c] = sum / (float)a.group;
    }
  }
}

// ---- host side -----------------------------------------------------------------------------

// Rows of a workgroup: RT row tiles hold 2 * (16 RT / T) samples.  What a choice costs: (rounds of
// workgroups over the 256 CUs) x max(RT CT, 4) -- below four MFMAs per k-step and wave the code
// stream through the CU's L1 sets the pace, not the matrix pipe.
static int pick_wide_rt(int T, int B, int CT, unsigned gy, int sph_cap, int *sph_out) {
  int best = 0;
  int64_t best_cost = 0;
  for (int rt = 1; rt <= 4; ++rt) {
    int sph = 16 * rt / T;
    if (sph < 1) continue;
    if (sph > sph_cap) sph = sph_cap;
    const int64_t wgs = (int64_t)((B + 2 * sph - 1) / (2 * sph)) * gy;
    const int64_t per = rt * CT < 4 ? 4 : rt * CT;
    const int64_t cost = ((wgs + 255) / 256) * per;
    // (ties to the larger tile: B = 4096, T = 20: four rounds of RT 3 take 0.156 ms, three of RT 4 0.145)
    if (best == 0 || cost <= best_cost) { best = rt; best_cost = cost; *sph_out = sph; }
  }
  return best;
}

const char *dense_wide_unsupported(int in_type, int32_t T, int32_t K, int32_t N, int64_t xs_t,
                                   int64_t xs_b, const void *x, const snnqp_weight_t *w,
                                   const int8_t *wt, const snnqp_neuron_t *nrn, int s_type) {
  // (float32 rows are staged as the uint8 ones are: the same requirements on K and col_sum)
  const char *why = dense_mfma_unsupported(in_type == SNNQP_F32 ? SNNQP_U8 : in_type, K, N, w, wt, nrn, s_type);
  if (why) return why;
  if (T > 64) return "more than 64 timesteps (a sample must fit the rows of one lane half)";
  if (N <= 128) return "at most 128 features: the 128-column kernel";
  if (xs_t < 0 || xs_b < 0) return "negative strides";
  if (in_type == SNNQP_U8 && ((((uintptr_t)x) & 15) != 0 || xs_t % 16 != 0 || xs_b % 16 != 0))
    return "uint8 rows not 16-byte aligned";
  if (in_type == SNNQP_F32 && ((((uintptr_t)x) & 15) != 0 || xs_t % 4 != 0 || xs_b % 4 != 0))
    return "float32 rows not 16-byte aligned";
  // 32-bit offsets from the workgroup's first sample (at most 128 samples)
  if (((int64_t)(T - 1) * xs_t + 128 * xs_b + (in_type == SNNQP_BITS ? (K + 31) / 32 : K)) *
          (in_type == SNNQP_F32 ? 4 : 1) >= ((int64_t)1 << 31))
    return "input strides beyond 32-bit offsets within a workgroup";
  return nullptr;
}

template <int RT, int CT, int IN, bool FUSE>
static void launch_wide(const DenseWideArgs &a, unsigned gx, unsigned gy, hipStream_t st) {
  hipLaunchKernelGGL((dense_wide_kernel<RT, CT, IN, FUSE>), dim3(gx, gy), dim3(W_THREADS), 0, st, a);
}

template <int CT, int IN, bool FUSE>
static void launch_wide_rt(int rt, const DenseWideArgs &a, unsigned gx, unsigned gy, hipStream_t st) {
  switch (rt) {
    case 4: launch_wide<4, CT, IN, FUSE>(a, gx, gy, st); break;
    case 3: launch_wide<3, CT, IN, FUSE>(a, gx, gy, st); break;
    case 2: launch_wide<2, CT, IN, FUSE>(a, gx, gy, st); break;
    default: launch_wide<1, CT, IN, FUSE>(a, gx, gy, st); break;
  }
}

static int wide_row_tiles_override() {
  static const int v = [] {
    const char *e = std::getenv("SNNQP_DENSE_WIDE_RT");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}

// The fused head with its hidden columns split over two workgroups per tile: when the batch gives
// the chip at most half a grid (config C2: B = 256 is 128 tiles of two samples), two workgroups of
// 256 columns each pull half of the first block's codes through their CU's L1 -- the bound of the
// unsplit launch -- and hand their half of the hidden raster over through the workspace.
constexpr int64_t W_TICKET_BYTES = 4096;         // a fixed head of the workspace: a ticket per tile
struct WidePlan { int rt, sph, ct, csplit; unsigned gx, gy; };

static WidePlan plan_wide(const DenseWideArgs &a, bool fuse, bool may_split) {
  WidePlan p = {};
  p.ct = a.N > 256 ? 2 : 1;
  p.gy = fuse ? 1u : (unsigned)((a.N + 256 * p.ct - 1) / (256 * p.ct));
  const int cap = fuse ? W_VOTE_SB / 2 : 64;
  p.rt = pick_wide_rt(a.T, a.B, p.ct, p.gy, cap, &p.sph);
  const int forced = wide_row_tiles_override();
  if (forced >= 1 && forced <= 4 && 16 * forced >= a.T) {
    p.rt = forced;
    p.sph = 16 * p.rt / a.T;
    if (p.sph > cap) p.sph = cap;
  }
  if (p.rt > 0) {
    p.gx = (unsigned)((a.B + 2 * p.sph - 1) / (2 * p.sph));
    static const bool no_split = std::getenv("SNNQP_DENSE_HEAD_NOSPLIT") != nullptr;
    if (fuse && may_split && !no_split && a.N > 256 && p.gx * 2u <= 256u && p.gx <= W_TICKET_BYTES / 4) {
      p.csplit = 1;
      p.ct = 1;
      p.gy = 2;
    }
  }
  return p;
}

static int64_t wide_workspace_bytes(const WidePlan &p) {
  return p.csplit ? W_TICKET_BYTES + (int64_t)p.gx * 2 * (p.rt * 32) * 8 * 4 : 0;
}

static int fill_and_launch(DenseWideArgs &a, int in_type, bool fuse, void *ws, int64_t ws_bytes,
                           hipStream_t st) {
  WidePlan p = plan_wide(a, fuse, ws != nullptr);
  if (p.csplit && (wide_workspace_bytes(p) > ws_bytes || ((uintptr_t)ws & 255) != 0)) p = plan_wide(a, fuse, false);
  SNNQP_REQUIRE(p.rt > 0, SNNQP_EUNSUPPORTED, "dense wide: T too large");
  a.SPH = p.sph;
  a.csplit = p.csplit;
  if (p.csplit) {
    a.hs_tickets = (uint32_t *)ws;
    a.hs_raster = (uint32_t *)((uint8_t *)ws + W_TICKET_BYTES);
    a.status = device_status_word(stream_device(st));
    // zeroed on the stream in front of every launch (dense_fp6.hip: a kernel node under capture)
    if (int rc = zero_words_async((uint32_t *)ws, (int64_t)p.gx, st)) return rc;
  }
  const int rt = p.rt;
  const unsigned gx = p.gx, gy = p.gy;
  if (in_type == SNNQP_F32) {
    SNNQP_REQUIRE(a.x_flags != nullptr, SNNQP_EINVAL, "dense wide: float32 rows need x_flags");
    if (int rc = zero_words_async((uint32_t *)a.x_flags, 1, st)) return rc;
  }
#define SNNQP_WIDE_IN(CTV, FUSEV)                                                          \
  do {                                                                                     \
    if (in_type == SNNQP_U8) launch_wide_rt<CTV, SNNQP_U8, FUSEV>(rt, a, gx, gy, st);        \
    else if (in_type == SNNQP_F32) launch_wide_rt<CTV, SNNQP_F32, FUSEV>(rt, a, gx, gy, st); \
    else launch_wide_rt<CTV, SNNQP_BITS, FUSEV>(rt, a, gx, gy, st);                          \
  } while (0)
  if (fuse) {
    if (p.ct == 2) SNNQP_WIDE_IN(2, true); else SNNQP_WIDE_IN(1, true);
  } else {
    if (p.ct == 2) SNNQP_WIDE_IN(2, false); else SNNQP_WIDE_IN(1, false);
  }
#undef SNNQP_WIDE_IN
  SNNQP_CHECK_LAUNCH("dense_wide_kernel");
  return SNNQP_OK;
}

int run_dense_wide(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B,
                   int32_t K, int32_t N, const snnqp_weight_t *w, const int8_t *wt,
                   const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                   float *u_out, uint32_t *s_out, int32_t *x_flags, hipStream_t st) {
  SNNQP_REQUIRE((x && s_out) || T == 0 || B == 0, SNNQP_EINVAL, "dense wide: null pointer");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "dense wide: negative T/B");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  SNNQP_CHECK_BN(bn);
  if (T == 0 || B == 0) return SNNQP_OK;
  DenseWideArgs a = {};
  a.x = x; a.xs_t = xs_t; a.xs_b = xs_b;
  a.T = T; a.B = B; a.K = K; a.N = N; a.KS = (K + 31) / 32; a.KW = (K + 31) / 32;
  a.wt = wt; a.col_sum = w->col_sum;
  a.dq = make_dequant(w->L, w->m);
  a.bn = make_bn(bn); a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out;
  a.x_flags = x_flags;
  return fill_and_launch(a, in_type, false, nullptr, 0, st);
}

}  // namespace snnqp

extern "C" int snnqp_dense_head_forward(const void *x, int in_type, int64_t x_stride_t,
                                        int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                                        int32_t N1, const snnqp_weight_t *w1, const int8_t *wt1,
                                        const snnqp_neuron_t *nrn1, int32_t N2,
                                        const snnqp_weight_t *w2, const int8_t *wt2,
                                        const snnqp_neuron_t *nrn2, int32_t group,
                                        uint32_t *s1_out, uint32_t *s2_out, float *logits,
                                        int32_t *x_flags, void *ws, int64_t ws_bytes,
                                        snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w1 && w2 && nrn1 && nrn2 && ((x && logits) || T == 0 || B == 0), SNNQP_EINVAL,
                "dense_head_forward: null argument");
  if (const uint32_t code = device_status_read(stream_device((hipStream_t)stream))) {
    set_error("dense_head_forward: device status 0x%x: %s (snnqp_device_status(..., reset = 1) clears it)",
              (unsigned)code, device_status_text(code));
    return SNNQP_EHIP;
  }
  SNNQP_REQUIRE(T >= 0 && B >= 0 && K > 0 && N1 > 0 && N2 > 0 && group > 0, SNNQP_EINVAL,
                "dense_head_forward: bad sizes");
  SNNQP_REQUIRE(N2 % group == 0, SNNQP_EINVAL, "dense_head_forward: N2=%d not divisible by group=%d", N2, group);
  for (const snnqp_neuron_t *nr : {nrn1, nrn2})
    SNNQP_REQUIRE(nr->kind >= SNNQP_NEURON_MULTI_STEP_LIF && nr->kind <= SNNQP_NEURON_LIF, SNNQP_EINVAL,
                  "dense_head_forward: unknown neuron kind %d", nr->kind);
  SNNQP_REQUIRE(w1->L >= 1.0f && w2->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  const char *why = nullptr;
  if (N1 > 512) why = "more than 512 hidden features";
  else if (N2 > 128) why = "more than 128 output features";
  else if (T > 64) why = "more than 64 timesteps";
  else if (T < 1) why = "no timestep";
  if (!why) {
    // (the 128-feature floor of the stand-alone wide kernel does not apply to the fused head)
    why = dense_mfma_unsupported(in_type == SNNQP_F32 ? SNNQP_U8 : in_type, K, N1, w1, wt1, nrn1, SNNQP_BITS);
    if (!why) why = dense_mfma_unsupported(SNNQP_BITS, N1, N2, w2, wt2, nrn2, SNNQP_BITS);
    if (!why && (x_stride_t < 0 || x_stride_b < 0)) why = "negative strides";
    if (!why && in_type == SNNQP_U8 &&
        ((((uintptr_t)x) & 15) != 0 || x_stride_t % 16 != 0 || x_stride_b % 16 != 0))
      why = "uint8 rows not 16-byte aligned";
    if (!why && in_type == SNNQP_F32 &&
        ((((uintptr_t)x) & 15) != 0 || x_stride_t % 4 != 0 || x_stride_b % 4 != 0))
      why = "float32 rows not 16-byte aligned";
    if (!why && in_type == SNNQP_F32 && !x_flags) why = "float32 rows need x_flags";
    if (!why && ((int64_t)(T - 1) * x_stride_t + 128 * x_stride_b + K) * (in_type == SNNQP_F32 ? 4 : 1) >=
                    ((int64_t)1 << 31))
      why = "input strides beyond 32-bit offsets within a workgroup";
  }
  SNNQP_REQUIRE(!why, SNNQP_EUNSUPPORTED, "dense_head_forward: %s", why);
  if (B == 0) return SNNQP_OK;
  DenseWideArgs a = {};
  a.x = x; a.xs_t = x_stride_t; a.xs_b = x_stride_b;
  a.T = T; a.B = B; a.K = K; a.N = N1; a.KS = (K + 31) / 32; a.KW = (K + 31) / 32;
  a.wt = wt1; a.col_sum = w1->col_sum;
  a.dq = make_dequant(w1->L, w1->m);
  a.bn = make_bn(nullptr); a.bn.mean = nullptr;
  a.nrn = make_neuron(nrn1);
  a.s_out = s1_out;
  a.N2 = N2; a.KS2 = (N1 + 31) / 32; a.group = group;
  a.wt2 = wt2; a.dq2 = make_dequant(w2->L, w2->m); a.nrn2 = make_neuron(nrn2);
  a.s2_out = s2_out; a.logits = logits;
  a.x_flags = x_flags;
  return fill_and_launch(a, in_type, true, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int64_t snnqp_dense_head_workspace_bytes(int32_t T, int32_t B, int32_t N1) {
  using namespace snnqp;
  if (T < 1 || T > 64 || B < 1 || N1 < 1 || N1 > 512) return 0;
  DenseWideArgs a = {};
  a.T = T; a.B = B; a.N = N1;
  return wide_workspace_bytes(plan_wide(a, true, true));
}
