// Device helpers shared by the fused conv3x3 MFMA kernels (conv3x3_u8c2.hip,
// conv3x3_bits.hip): patch schedule, LDS tables, the neuron epilogue on the MFMA
// C/D layout, spike-word staging.  gfx950 only.
#pragma once
#include "kernels.h"

namespace snnqp {


typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// Two neurons (two pixels of one channel) travel through the epilogue side by side.
// As SCALAR float32 instructions: on gfx950 the packed forms (v_pk_fma/mul/add_f32) take
// two passes anyway and, unlike scalar VALU ops, do not overlap with an MFMA in flight
// (tools/ubench/mfma_overlap_classes.hip: 4 v_pk_fma + 1 MFMA = 25.5 ns against 9.7 +
// 13.9 apart; 8 v_fma_f32 + 1 f8f6f4 MFMA = 21.7 ns against 18.4 + 15.3).
struct v2f {
  float x, y;
};
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f operator+(v2f a, float b) { return v2f{a.x + b, a.y + b}; }
__device__ __forceinline__ v2f operator-(v2f a, float b) { return v2f{a.x - b, a.y - b}; }
__device__ __forceinline__ v2f operator*(v2f a, float b) { return v2f{a.x * b, a.y * b}; }
__device__ __forceinline__ v2f operator-(v2f a) { return v2f{-a.x, -a.y}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) {
  return v2f{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}


// Dequantisation by LDS table (fast neuron path only).  The accumulator is made
// to BE the table address: the A operand carries 4x the input (spike bytes {0,4},
// event counts << 2) and the MFMA chain starts from C = byte address of the entry
// of acc = 0, so ds_read_b32 takes the MFMA result as is (no index arithmetic).
//   LUT_SHARED : one table, entry = fl(fl(acc / L) * m), |acc| <= LUT_CAP
//   LUT_CHANNEL: one table per output channel with BatchNorm applied to the entry
//                as well (conv0: each channel over its own accumulator range), so the
//                epilogue starts at the membrane update
enum { LUT_NONE = 0, LUT_SHARED = 1, LUT_CHANNEL = 2 };
// 4095: a 32 KiB table.  conv0 on event counts with 8-bit codes (|acc| <= sum|w| * x_max, a few
// thousand) then still dequantises by table and keeps three workgroups per CU; in the bits
// kernel the entry of acc = 0 stays an instruction immediate (< 64 KiB from the LDS base).
constexpr int LUT_CAP = 4095;
// LUT_CHANNEL: at most 32 KiB of tables -- three workgroups per CU still fit; taller tables cost
// an occupancy step that the shared table does not (headline layer, count frames, hint 3 / 229 rows:
// 6.68 ms; hint 4 / 304 rows at two workgroups per CU: 8.58; shared table: 7.03)
constexpr int LUT2_ROWS = 256;
constexpr int LUT_XMAX = 31;                    // 4 * x must stay an int8

constexpr int HALO = 10;
// LDS image of one timestep's halo for the int8 kernels: four planes (the 32-channel
// k-steps kk of a tap), rows of HPITCH pixels, 32 B per pixel = the two 16-byte lane
// halves, swapped on odd halo rows.  Every tap / k-step / tile offset of an A fragment
// is then an instruction immediate on one of two lane bases, and the wave's
// ds_read_b128 are bank-conflict free (tools/ubench/lds_conv_patterns.hip: 222
// B/clk/CU; the earlier pixel-major image with an XOR swizzle measured 63).
constexpr int HPITCH = 12;
constexpr int HPLANE = HALO * HPITCH * 32;      // one k-step plane

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
  int32_t lut_bound;  // > 0: |acc| <= lut_bound while inputs <= x_limit, dequant by LDS table
  int32_t lut_rows;   // u8c2 kernel, LUT_CHANNEL: rows (of 32 entries) the launch reserved for the tables
  const int32_t *ch_slots;  // u8c2 kernel, LUT_CHANNEL: table slot of every channel (snnqp_weight_t), or null
  int32_t x_limit;    // u8c2 kernel: largest input value the table mode is sized for
  int32_t *x_seen;    // u8c2 kernel: (nullable) atomically max-ed with the largest input seen
  int32_t *x_flags;   // u8c2 kernel, float32 frames: OR-ed with SNNQP_FLAG_NOT_INTEGER (snnqp.h)
  int32_t xcd_split;  // patch schedule keeps a sample on one XCD (grid % 8 == 0, B >= 8)
  int32_t tchunk;     // u8c2 kernel: timesteps staged per pass (<= 32)
  uint32_t *sched;    // work queues of this launch (launch_persistent), or null: static walk
  uint32_t *status;   // the device's status word (runtime.hip), or null
  int32_t patch_h;    // rows of a patch: 8 (two 4x8 tiles), or 4 (conv3x3_bits.hip, one tile)
  const int32_t *pred;  // u8c2 kernel on byte / float32 frames: (nullable) the launch runs only if *pred != 0
};

// Work queues of one launch: per blockIdx.y, one patch counter per XCD queue and one count
// of finished workgroups (the last one zeroes the words for the slot's next user).
constexpr int SCHED_Y = 8;                  // blockIdx.y values a slot serves (Cout <= 1024)
constexpr int SCHED_WORDS = SCHED_Y * 16;   // words of one slot
constexpr int SCHED_SLOTS = 64;             // launches in flight per device
constexpr int SCHED_CAPTURE_SLOTS = 960;    // launches captured into graphs per device (never reused)

// 16 spike bits -> 16 bytes {0, 1} (or {0, 4} when the accumulator indexes a table)
template <bool X4>
__device__ __forceinline__ v4i expand16(uint32_t b) {
  constexpr uint32_t MUL = X4 ? 0x00810204u : 0x00204081u;
  constexpr uint32_t AND = X4 ? 0x04040404u : 0x01010101u;
  v4i o;
  o.x = (int)((((b >> 0) & 0xFu) * MUL) & AND);
  o.y = (int)((((b >> 4) & 0xFu) * MUL) & AND);
  o.z = (int)((((b >> 8) & 0xFu) * MUL) & AND);
  o.w = (int)((((b >> 12) & 0xFu) * MUL) & AND);
  return o;
}

// Patch schedule of a persistent workgroup.  With xcd_split the workgroups that
// share an XCD (blockIdx.x % 8 under the observed round-robin placement -- a speed
// assumption only) walk the samples b = xcd (mod 8), neighbouring patches at the
// same time, so the halo lines neighbouring patches share are served by that
// XCD's L2 instead of being fetched once per XCD.
struct PatchWalk {
  int64_t first, count, stride;
  int ppb, xcd;
  bool split;
  uint32_t *queue;    // this workgroup's patch counter, or null
  uint32_t *done;
  __device__ __forceinline__ explicit PatchWalk(const ConvMfmaArgs &a) {
    ppb = a.tiles_y * a.tiles_x;
    split = a.xcd_split != 0;
    if (split) {
      xcd = blockIdx.x & 7;
      first = blockIdx.x >> 3;
      stride = gridDim.x >> 3;
      count = (int64_t)((a.B - xcd + 7) >> 3) * ppb;
    } else {
      xcd = 0;
      first = blockIdx.x;
      stride = gridDim.x;
      count = a.npatch;
    }
    queue = a.sched ? a.sched + blockIdx.y * 16 + xcd : nullptr;
    done = a.sched ? a.sched + blockIdx.y * 16 + 8 : nullptr;
  }
  // Dynamic schedule: the first patch of a workgroup is its static one; every further patch
  // is claimed from the queue (workgroups that the CU's oldest-first issue favours finish
  // their patches sooner and would otherwise idle while the youngest still has a quarter
  // of its static share left: 4.75 against 7.44 ms measured inside conv0).  One thread
  // claims, a patch ahead so that the atomic's round trip hides behind the patch.
  // (never beyond `count`, whatever the counter holds: a word that was not zero at the start
  // of the launch must not cost an out-of-range patch index -- it costs patches, and finish()
  // reports that)
  __device__ __forceinline__ int64_t claim() const {
    const int64_t r = stride + (int64_t)atomicAdd(queue, 1u);
    return r < count ? r : count;
  }
  // After the last claim of the workgroup (one thread): add the patches this workgroup processed
  // to the launch's tally and count the workgroup; the last one of the launch checks the tally
  // against the number of patches the launch had -- a queue word that was not zero when the
  // launch began (an aborted launch, a graph replayed concurrently with itself) makes
  // workgroups skip patches, which must not pass silently: the device's status word gets
  // SNNQP_STATUS_QUEUE_CORRUPT and the next call into the library fails (runtime.hip) -- and
  // zeroes the slot for its next user.
  __device__ __forceinline__ void finish(uint32_t processed, int64_t npatch, uint32_t *status) const {
    atomicAdd(done + 1, processed);
    __threadfence();
    const uint32_t total = gridDim.x;
    if (atomicAdd(done, 1u) + 1u == total) {
      __threadfence();
      const uint32_t tally = atomicAdd(done + 1, 0u);
      if (tally != (uint32_t)npatch && status) *(volatile uint32_t *)status = SNNQP_STATUS_QUEUE_CORRUPT;
      for (int i = 0; i < 8; ++i) done[i - 8] = 0u;
      done[1] = 0u;
      __threadfence();
      *done = 0u;
    }
  }
  __device__ __forceinline__ void decode(const ConvMfmaArgs &a, int64_t r, int &b, int &y0,
                                         int &x0) const {
    const int within = (int)(r % ppb);
    const int bi = (int)(r / ppb);
    b = split ? xcd + 8 * bi : bi;
    y0 = (within / a.tiles_x) * a.patch_h;
    x0 = (within % a.tiles_x) * 8;
  }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences
// global memory, i.e. waits (vmcnt(0)) for the spike stores of the previous step
// and the prefetched halo loads -- a full memory round trip per timestep.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// old with lane `lane` replaced by the wave-uniform `val` (v_writelane_b32).
// This hipcc has no __builtin_amdgcn_writelane; binding the LLVM intrinsic by its
// name keeps the instruction visible to the compiler, which then inserts the wait
// states a VALU-written SGPR needs before it (inline asm would hide that hazard).
extern "C" __device__ uint32_t snnqp_writelane_i32(uint32_t, uint32_t, uint32_t)
    __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t writelane_u32(uint32_t val, int lane, uint32_t old) {
  return snnqp_writelane_i32(val, (uint32_t)lane, old);
}

// u with the lanes of `mask` zeroed: the hard reset reuses the ballot of the
// threshold compare (written as a C++ select the compiler emits a second, negated
// v_cmp per element).  VALU readers of a VALU-written SGPR need no wait states.
__device__ __forceinline__ float reset_where(float u, unsigned long long mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(u), "s"(mask));
  return r;
}

struct LaneConsts {
  float bmean, bmul, bbias, dec;
  float vr;      // v_reset (a VGPR operand of the reset select when it is not 0)
};

// Output-channel block of a wave when Cout is not a multiple of 32: `cout` is the real
// channel (u_io skips lanes beyond Cout), `cpar` a clamped index for the per-channel
// parameter loads, `cmask` the bits of the block's spike word that exist.
__device__ __forceinline__ uint32_t chan_mask(int cout_base, int Cout) {
  const int valid = Cout - cout_base;
  return valid >= 32 ? 0xFFFFFFFFu : valid <= 0 ? 0u : (1u << valid) - 1u;
}

// The neuron update forms of spiking_learning.py, one straight-line epilogue each
// (template parameter NF of the conv kernels):
//   NF_MUL0   u += (x - u) * m, reset to 0        multi_step_LIF with tau = 2^j (m = 1/tau,
//                                                 exact) and parametric_leaky_IF (m =
//                                                 sigmoid(tau)), v_reset = 0     :381,410
//   NF_MUL    u += (x - (u - v_reset)) * m        the same neurons, v_reset != 0
//   NF_DIV    u += (x - (u - v_reset)) / tau      multi_step_LIF, any other tau    :410
//   NF_DECAY  u = u * decay[c] + x                LIF (per-feature sigmoid(tau))   :432
// all followed by s = (u >= v_th) and the hard reset u = s ? v_reset : u (:412-414).
enum { NF_MUL0 = 1, NF_MUL = 2, NF_DIV = 3, NF_DECAY = 4 };

__host__ inline int neuron_form(const NeuronP &n) {
  if (n.kind == SNNQP_NEURON_LIF) return NF_DECAY;
  if (n.inv_k == 0.0f) return NF_DIV;               // multi_step_LIF, tau not a power of two
  return n.vr == 0.0f ? NF_MUL0 : NF_MUL;
}

// LDS accesses by absolute 32-bit LDS address (address space 3): the table reads
// take the MFMA result itself as the address, with no per-read base add.
typedef __attribute__((address_space(3))) const float lds_cfloat_t;
typedef __attribute__((address_space(3))) const uint8_t lds_cu8_t;
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(lds_cu8_t *)p;
}
__device__ __forceinline__ float lds_read_f32(uint32_t addr) {
  return *(lds_cfloat_t *)(uintptr_t)addr;
}

// lut[i] = fl(fl((i - bound) / L) * m): the dequantised current of accumulator
// value i - bound, built once per workgroup (same exact division, common.h).
__device__ __forceinline__ void build_lut(float *lut, int bound, const Dequant &dq,
                                          int tid, int nthreads = 256) {
  for (int i = tid; i <= 2 * bound; i += nthreads) lut[i] = dequant_acc_nb(i - bound, dq);
}

// Per-channel tables of the workgroup's 128 output channels: one block per wave,
// [acc + bound][32 channels] (rows of 128 B), entry = BatchNorm_c(dequant(acc)), the
// same op sequence the epilogue would run (bn_apply on dequant_acc_nb), so folding
// changes no bit.  With this order lane (channel c) reads LDS bank c whatever its
// accumulator is: the table reads of a wave never conflict (the channel-major order
// [channel][acc] lost half of its LDS cycles to conflicts of the lanes with acc != 0).
// The accumulator must then count 128 B per unit: A carries 16 x input, B 8 x code.
// LUT_CHANNEL tables (conv3x3_u8c2.hip): every channel of the workgroup's 128 keeps the entries of
// acc = -neg_c x_limit .. +pos_c x_limit (pos_c / neg_c = the sums of its positive / |negative|
// codes: what its accumulator can reach with inputs 0 .. x_limit) in ONE of the 32 bank columns of
// a table of 128-byte rows, stacked with the three other channels of that column: slot = 4 x column
// + position (snnqp_weight_t.ch_slots; default column = c mod 32, position = wave).  The 32 channels
// of a wave sit in 32 different columns, so its lanes never collide whatever their accumulators
// are, and the table is as tall as the tallest column -- not four times the widest channel's
// symmetric range.
//   scr[0..127]    rows by slot           scr[128..255]  neg_c x_limit by channel
//   scr[256..383]  slot by channel
// Returns the bits of the smallest non-zero |entry| this thread wrote (+inf if none).
__device__ __forceinline__ int lut_column_start(const uint32_t *scr, int slot) {
  int st = 0;
  for (int k = 0; k < (slot & 3); ++k) st += (int)scr[(slot & ~3) + k];
  return st;
}
__device__ __forceinline__ uint32_t build_lut_channel(float *lut, const uint32_t *scr, int max_rows,
                                                      const Dequant &dq, const BnP &bn, int cout0,
                                                      int Cout, int tid) {
  uint32_t minbits = 0x7F800000u;
  const int c = tid & 127, hf = tid >> 7;
  const int slot = (int)scr[256 + c], col = slot >> 2;
  const int st = lut_column_start(scr, slot);
  const int rows = (int)scr[slot], negx = (int)scr[128 + c];
  if (st + rows > max_rows) return minbits;                    // (reported by the caller)
  const int co = cout0 + c < Cout ? cout0 + c : Cout - 1;
  float bm = 0.f, bmul = 1.f, bb = 0.f;
  if (bn.mean) { bm = bn.mean[co]; bmul = bn.mul[co]; bb = bn.bias[co]; }
  for (int i = hf; i < rows; i += 2) {
    const float y = bn_apply(dequant_acc_nb(i - negx, dq), bm, bmul, bb);
    lut[(st + i) * 32 + col] = y;
    const uint32_t mag = __float_as_uint(y) & 0x7FFFFFFFu;
    if (mag != 0u && mag < minbits) minbits = mag;
  }
  return minbits;
}

// fma(d, 2^-j, u) == fl(u + fl(d * 2^-j)) for all T steps of a launch that starts from
// u = 0 when every input current x is 0 or has |x| >= 2^(j (T + 1) - 126):
// all x are then multiples of q = 2^(e_min - 23); by induction u_t and d_t = fl(x - u_t)
// are multiples of q 2^(-j t) (rounding a multiple of a power of two to 24 bits keeps
// it one), so d_t 2^-j is a multiple of q 2^(-j (t + 1)) >= 2^-149 with the significand
// of d_t: representable, the product rounds nothing away.  `min_x_bits` = bits of the
// smallest non-zero |x| the launch can see.
__host__ __device__ __forceinline__ bool lif_fma_is_exact(uint32_t min_x_bits, int j, int T, bool has_u0) {
  if (has_u0 || j < 0) return false;
  const long long e = (long long)j * (T + 1) - 126;     // needed exponent of min |x|
  if (e > 100) return false;
  const uint32_t need = e <= -126 ? 0x00800000u : (uint32_t)(e + 127) << 23;
  return min_x_bits >= need;
}

// Smallest non-zero |BatchNorm_c(entry)| (as float32 bits) over the shared table and the 128
// channels of a workgroup: the `min_x_bits` of lif_fma_is_exact for the kernels that apply
// BatchNorm in the epilogue.  Same operation order as the epilogue.  nthreads % 128 == 0,
// so a thread keeps one channel.
__device__ __forceinline__ uint32_t lut_bn_min_bits(const float *lut, int bound, const BnP &bn,
                                                    int cout0, int Cout, int tid, int nthreads) {
  int c = cout0 + (tid & 127);
  if (c >= Cout) c = Cout - 1;
  float mean = 0.f, mul = 1.f, bias = 0.f;
  if (bn.mean) { mean = bn.mean[c]; mul = bn.mul[c]; bias = bn.bias[c]; }
  uint32_t mb = 0x7F800000u;
  for (int i = tid >> 7; i <= 2 * bound; i += nthreads >> 7) {
    float x = lut[i] - mean;
    x = x * mul;
    x = x + bias;
    const uint32_t b = __float_as_uint(x) & 0x7FFFFFFFu;
    if (b != 0 && b < mb) mb = b;
  }
  return mb;
}

// Dequantised currents of two accumulator registers (two pixels, same channel).
// Table modes: the register is the LDS address of its entry.  Otherwise packed
// float32 ops (v_pk_*_f32 keep every rounding of the scalar sequence).
// `off` (an integer < 2^24 as float, exact): constant added to the accumulator first --
// the u8 kernel accumulates (x - 128) * w so that counts up to 255 are int8 operands,
// and adds 128 * sum(w) back here.
template <int LUTM, bool OFFS = false>
__device__ __forceinline__ v2f dequant_pair(int a0, int a1, const Dequant &dq, float off = 0.0f) {
  if (LUTM != LUT_NONE) return v2f{lds_read_f32((uint32_t)a0), lds_read_f32((uint32_t)a1)};
  v2f a = {(float)a0, (float)a1};
  if (OFFS) a = a + off;
  const v2f q = fma2(a, v2f{dq.rL, dq.rL}, a * dq.rLlo);     // exact a / L (common.h)
  return q * dq.m;
}

// u -> u' for two pixels of one channel (before threshold and reset).
// FMA (NF_MUL0 only): u + d * m as one fused multiply-add.  Identical to the two-step
// form whenever d * m is exact, i.e. never a subnormal with bits shifted out; the
// caller proves that for the launch (lif_fma_is_exact) before taking this variant.
template <int NF, bool FMA = false, bool PACKED = true>
__device__ __forceinline__ v2f neuron_update(v2f x, v2f u, const LaneConsts &lc,
                                             const NeuronP &nrn) {
  if (NF == NF_DECAY) {
    const v2f ud = u * lc.dec;
    return ud + x;
  }
  // u - 0 == u exactly, so NF_MUL0 skips the subtraction
  if (FMA && PACKED) {
    // conv0's table kernel: no MFMA worth overlapping (one per 1024 updates), and one
    // v_pk_add + one v_pk_fma per pair issue in fewer slots than four scalar ops
    // (measured 7.8 against 8.1 ms), so this variant keeps the packed forms
    typedef float v2fp __attribute__((ext_vector_type(2)));
    const v2fp up = {u.x, u.y}, xp = {x.x, x.y};
    const v2fp dp = NF == NF_MUL0 ? xp - up : xp - (up - lc.vr);
    const v2fp r = __builtin_elementwise_fma(dp, v2fp{nrn.inv_k, nrn.inv_k}, up);
    return v2f{r.x, r.y};
  }
  const v2f d = NF == NF_MUL0 ? x - u : x - (u - lc.vr);
  if (NF == NF_DIV) return u + v2f{d.x / nrn.k, d.y / nrn.k};
  if (FMA) return fma2(d, v2f{nrn.inv_k, nrn.inv_k}, u);
  const v2f dk = d * nrn.inv_k;
  return u + dk;
}

// hard reset of the lanes in `mask`
template <int NF>
__device__ __forceinline__ float neuron_reset(float u, unsigned long long mask,
                                              const LaneConsts &lc) {
  if (NF == NF_MUL0) return reset_where(u, mask);
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(lc.vr), "s"(mask));
  return r;
}

// BatchNorm (unless the table already applied it) + neuron for two dequantised
// currents (two pixels, same channel).  With float32 subnormals kept (hipcc default)
// (u - v_th) >= 0  <=>  u >= v_th.
template <int NF, bool BNDONE, bool FMA = false>
__device__ __forceinline__ void neuron_pair(v2f y, float &u0, float &u1,
                                            const LaneConsts &lc, const NeuronP &nrn,
                                            unsigned long long &m0,
                                            unsigned long long &m1) {
  v2f x = y;
  if (!BNDONE) {
    x = x - lc.bmean;
    x = x * lc.bmul;
    x = x + lc.bbias;
  }
  const v2f uu = neuron_update<NF, FMA>(x, v2f{u0, u1}, lc, nrn);
  m0 = __ballot(uu.x >= nrn.vth);
  m1 = __ballot(uu.y >= nrn.vth);
  u0 = neuron_reset<NF>(uu.x, m0, lc);
  u1 = neuron_reset<NF>(uu.y, m1, lc);
}

// BatchNorm + neuron + spike words of one 32x32 tile from its 8 dequantised pairs.
// The lane mask of register i holds pixel (ty = 2*(i>>3), tx = i&7) in its low half
// and (ty + 1, tx) in its high half.  Returns the word this lane stores:
//   POOL : lanes 0..7  = pooled pixel (pty = lane >> 2, ptx = lane & 3)
//   !POOL: lanes 0..31 = pixel row `lane` of the tile
template <int NF, bool POOL, bool BNDONE, bool FMA = false>
__device__ __forceinline__ uint32_t tile_neurons(const v2f (&y)[8], float (&u)[16],
                                                 const LaneConsts &lc, const NeuronP &nrn) {
  uint32_t myw = 0;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {     // masks are consumed pair by pair
    unsigned long long m0, m1;
    neuron_pair<NF, BNDONE, FMA>(y[i >> 1], u[i], u[i + 1], lc, nrn, m0, m1);
    // the masks are wave-uniform: v_writelane drops each word into the lane that
    // stores it (no per-lane compare masks to keep in SGPRs)
    if (POOL) {
      const unsigned long long o = m0 | m1;
      const uint32_t pw = (uint32_t)o | (uint32_t)(o >> 32);
      myw = writelane_u32(pw, i >> 1, myw);
    } else {
      const int r0 = (i & 3) + 8 * (i >> 2);          // row of element i, low half
      myw = writelane_u32((uint32_t)m0, r0, myw);
      myw = writelane_u32((uint32_t)(m0 >> 32), r0 + 4, myw);
      myw = writelane_u32((uint32_t)m1, r0 + 1, myw);
      myw = writelane_u32((uint32_t)(m1 >> 32), r0 + 5, myw);
    }
  }
  return myw;
}

// Whole-tile epilogue (used where no MFMA stream runs beside it): all table
// reads are issued first, then the pairs are processed.
template <int NF, bool POOL, int LUTM, bool FMA = false, bool OFFS = false>
__device__ __forceinline__ uint32_t tile_epilogue(const v16i &acc, float (&u)[16],
                                                  const Dequant &dq,
                                                  const LaneConsts &lc,
                                                  const NeuronP &nrn, int lane,
                                                  float off = 0.0f) {
  v2f y[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = dequant_pair<LUTM, OFFS>(acc[2 * j], acc[2 * j + 1], dq, off);
  return tile_neurons<NF, POOL, LUTM == LUT_CHANNEL, FMA>(y, u, lc, nrn);
}

// The same in two halves of four pairs: eight table reads in flight instead of sixteen -- fewer
// registers alive, for the build that buys a sixth wave per SIMD with them (conv3x3_u8c2.hip)
template <int NF, bool POOL, int LUTM, bool FMA = false, bool OFFS = false>
__device__ __forceinline__ uint32_t tile_epilogue_halves(const v16i &acc, float (&u)[16],
                                                         const Dequant &dq, const LaneConsts &lc,
                                                         const NeuronP &nrn, int lane, float off = 0.0f) {
  uint32_t myw = 0;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    v2f y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = dequant_pair<LUTM, OFFS>(acc[8 * g + 2 * j], acc[8 * g + 2 * j + 1], dq, off);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 8 * g + 2 * j;
      unsigned long long m0, m1;
      neuron_pair<NF, LUTM == LUT_CHANNEL, FMA>(y[j], u[i], u[i + 1], lc, nrn, m0, m1);
      if (POOL) {
        const unsigned long long o = m0 | m1;
        const uint32_t pw = (uint32_t)o | (uint32_t)(o >> 32);
        myw = writelane_u32(pw, i >> 1, myw);
      } else {
        const int r0 = (i & 3) + 8 * (i >> 2);
        myw = writelane_u32((uint32_t)m0, r0, myw);
        myw = writelane_u32((uint32_t)(m0 >> 32), r0 + 4, myw);
        myw = writelane_u32((uint32_t)m1, r0 + 1, myw);
        myw = writelane_u32((uint32_t)(m1 >> 32), r0 + 5, myw);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return myw;
}

// Spike words are staged in LDS, obuf[slot = t % FL][pixel][4 words of the 128-
// channel block], and flushed with 16-byte stores: no global store sits in the
// per-timestep loop (vmcnt stays a pure load counter there).
template <bool POOL>
struct OutStage {
  static constexpr int NPIX = POOL ? 16 : 64;   // (pooled) pixels of one patch
  static constexpr int FL = POOL ? 32 : 8;      // timesteps between flushes
  static constexpr int BYTES = FL * NPIX * 16;
};

// patch-pixel index this lane's word of tile `tl` belongs to
template <bool POOL>
__device__ __forceinline__ int out_pix(int tl, int lane) {
  if (POOL) return (tl * 2 + ((lane >> 2) & 1)) * 4 + (lane & 3);
  const int ty = ((lane >> 2) & 1) | (((lane >> 4) & 1) << 1);
  const int tx = (lane & 3) | (((lane >> 3) & 1) << 2);
  return (tl * 4 + ty) * 8 + tx;
}

// Writes timesteps [t0, t0 + n) of the patch at (y0, x0) (full-resolution
// coordinates) from obuf (a ring of SLOTS timesteps) to global memory.  All NT
// threads of the workgroup take part.
template <bool POOL, int SLOTS, int NT, int NPIX = OutStage<POOL>::NPIX>
__device__ __forceinline__ void flush_ring(const uint32_t *obuf, const ConvMfmaArgs &a,
                                           int t0, int n, int b, int y0, int x0, int tid) {
  constexpr int PW = POOL ? 4 : 8;              // patch width in output pixels
  const int CW = (a.Cout + 31) >> 5;
  const int cwb = blockIdx.y * 4;
  const int nw = min(4, CW - cwb);
  const int OH = POOL ? a.H >> 1 : a.H, OW = POOL ? a.W >> 1 : a.W;
  const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
  for (int i = tid; i < n * NPIX; i += NT) {
    const int slot = i / NPIX, pix = i % NPIX;
    const int t = t0 + slot;
    const uint32_t *src = obuf + ((t % SLOTS) * NPIX + pix) * 4;
    const int oy = oy0 + pix / PW, ox = ox0 + pix % PW;
    if (oy >= OH || ox >= OW) continue;          // edge patch of an image not 8-aligned
    uint32_t *dst = a.s_out + ((((int64_t)t * a.B + b) * OH + oy) * OW + ox) * CW + cwb;
    if (nw == 4 && (CW & 3) == 0) {
      *(v4i *)dst = *(const v4i *)src;
    } else {
      for (int w = 0; w < nw; ++w) dst[w] = src[w];
    }
  }
}

template <bool POOL>
__device__ __forceinline__ void flush_out(const uint32_t *obuf, const ConvMfmaArgs &a,
                                          int t0, int n, int b, int y0, int x0, int tid) {
  flush_ring<POOL, OutStage<POOL>::FL, 256>(obuf, a, t0, n, b, y0, x0, tid);
}

// membrane potentials of one tile (tl) of the patch <-> global memory
template <bool LOAD>
__device__ __forceinline__ void u_io_tile(float (&u)[16], const ConvMfmaArgs &a, int b,
                                          int y0, int x0, int cout, int h, int tl) {
  float *uo = a.u_out;
  const float *ui = a.u0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int y = y0 + tl * 4 + (h | ((i >> 3) << 1));
    const int x = x0 + (i & 7);
    const int64_t o = (((int64_t)b * a.H + y) * a.W + x) * a.Cout + cout;
    const bool in = y < a.H && x < a.W && cout < a.Cout;
    if (LOAD) u[i] = in ? ui[o] : 0.0f;
    else if (in) uo[o] = u[i];
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
      const bool in = y < a.H && x < a.W && cout < a.Cout;
      if (LOAD) u[tl][i] = in ? ui[o] : 0.0f;
      else if (in) uo[o] = u[tl][i];
    }
}

__device__ __forceinline__ void zero_u(float (&u)[2][16]) {
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int i = 0; i < 16; ++i) u[tl][i] = 0.0f;
}

__device__ __forceinline__ v16i splat16(int v) {
  return v16i{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v};
}

// Work-queue words for a launch: a per-device pool of SCHED_SLOTS slots handed out round-robin,
// allocated once per device and kept for the life of the process.  A slot is self-validating:
//  * it is zeroed on the launch's stream right before the launch (so whatever an earlier,
//    possibly aborted, launch left in it cannot be walked), and
//  * an event recorded after the launch marks it busy: when the round-robin pointer comes
//    back to a slot whose last launch has not completed (more than SCHED_SLOTS launches in
//    flight across streams / threads), the launch takes the static walk instead (sched =
//    nullptr), which is always correct, only less well balanced (C3 under a graph without
//    queues: 13.6 ms against 12.3).
// A launch that is being CAPTURED into a graph (events cannot be queried there) gets a slot of
// its own out of SCHED_CAPTURE_SLOTS further ones, handed back when the caller says the graph is
// gone (snnqp_workqueue_capture_release; linen.CapturedApply does).  Nothing is added to
// the graph for it: the pool is zeroed when it is allocated, the last workgroup of every launch
// leaves the slot's words zero again (PatchWalk::finish), and a graph does not run concurrently
// with itself -- every replay finds zeroed words.  (A memset node in front of the kernel node,
// the obvious way, aborted on the replay of the second graph captured in a process in round 3,
// and in round 5 a memset node in front of the dense hand-over never reached its target in the
// second graph a framework captured -- profiles/r05_capture_memset.txt; round 6 walked the nodes
// (profiles/r06_capture_memset_nodes.txt): the node is there and correctly addressed, the runtime's
// replay of it writes another memset node's fill-kernel arguments instead of zeros.  The library
// records no memset node: kernels zero.)  A word that is NOT zero at a replay makes workgroups skip
// patches: finish() tallies the patches and reports a mismatch through the device's status word
// (runtime.hip).  The pool must
// exist by then -- nothing may be allocated during a capture: one eager launch on the device,
// the usual warm-up.  With all 960 capture slots taken, or without the pool, a captured launch
// takes the static walk (counted: snnqp_workqueue_stats).
// sched_acquire returns the slot's words (or nullptr) and its index; sched_release records
// the event.  The device comes from the stream, not from the calling thread's current device.
uint32_t *sched_acquire(hipStream_t st, int *dev, int *slot);
void sched_release(int dev, int slot, hipStream_t st);
uint32_t *device_status_word(int dev);            // runtime.hip

// CUs of device `dev` and the workgroups of `kernel` one CU holds (threads, dynamic LDS):
// asked from the runtime once per (kernel, threads, LDS bytes, device) and remembered -- the
// two queries cost tens of microseconds, as much as a small layer's kernel.
void persistent_limits(const void *kernel, int threads, size_t dyn_lds, int dev, int *cus, int *occ);
int stream_device(hipStream_t st);

template <typename K>
static inline void launch_persistent(K kernel, ConvMfmaArgs a, unsigned gy, hipStream_t st,
                                     size_t dyn_lds = 0, int threads = 256) {
  int dev = 0, cus = 256, occ = 2, slot = -1;
  a.sched = gy <= (unsigned)SCHED_Y ? sched_acquire(st, &dev, &slot) : nullptr;   // npatch < 2^30: run_conv3x3_mfma
  if (!a.sched) dev = stream_device(st);
  a.status = a.sched ? device_status_word(dev) : nullptr;
  persistent_limits((const void *)kernel, threads, dyn_lds, dev, &cus, &occ);
  const int64_t gmax = (int64_t)cus * occ;
  unsigned gx = (unsigned)(a.npatch < gmax ? a.npatch : gmax);
  a.xcd_split = 0;
  if (gx >= 64 && a.B >= 8) {     // whole samples per XCD
    gx &= ~7u;
    a.xcd_split = 1;
  }
  hipLaunchKernelGGL(kernel, dim3(gx, gy), dim3(threads), dyn_lds, st, a);
  if (a.sched) sched_release(dev, slot, st);
}

// conv3x3_bits.hip: bit-packed input, Cin <= 128; i8 = codes wider than fp6 holds
// dq: how the accumulator becomes the current -- DQ_ARITH (three float32 instructions),
// DQ_ONE (L == 1: one multiply), DQ_TABLE (fp6 instruction, NF_MUL0, 0 < a.lut_bound =
// abs_sum_max <= DQT_MAXA: an LDS table addressed by the accumulator's bit pattern)
// fma: the membrane update as one fused multiply-add (NF_MUL0, proven exact for this launch
// by the caller: snnqp_weight_t.min_current_bits + lif_fma_is_exact)
// bnf: every BatchNorm mean and bias is zero (snnqp_bn_t.flags): x = y * mul
enum { DQ_ARITH = 1, DQ_ONE = 2, DQ_TABLE = 3 };
constexpr int DQT_MAXA = 2047;
void launch_conv3x3_bits(const ConvMfmaArgs &a, bool i8, int nf, bool pool, int dq, bool fma,
                         bool bnf, unsigned gy, hipStream_t st);

}  // namespace snnqp
