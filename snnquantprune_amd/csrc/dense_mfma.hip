// Fused SpikingBlock(QuantDense, neuron) (spiking_learning.py:446-462 with
// flax_qdense.py:87 as the connection) on int8 MFMA: bit-packed spikes x int8
// codes -> int32 -> dequantise -> [BatchNorm] -> neuron over T -> packed spikes.
// A second instantiation (IN = SNNQP_U8) reads uint8 rows as they are -- the [B, T, K]
// event-count input of a dense-only model (config C2) -- with no packing pass and no
// inspection of the values: a byte x enters the MFMA as the int8 x - 128 (x ^ 0x80, exact
// for every count 0..255) and 128 * sum_k w[k][n] (snnqp_weight_t.col_sum) is added back to
// the int32 accumulator.
//
// The connection is stateless across t, so the contraction is one GEMM over
// rows m = (sample, t); only the neuron is sequential:
//  * a workgroup owns SB samples x all T steps (<= RT*32 rows) and a block of 128
//    output features; it is 8 waves = two groups of 4 (32 features per wave) that
//    split K: group g walks the chunks c = g (mod 2) with its own LDS buffers, and the
//    int32 partial tiles are added once at the end;
//  * K is walked in chunks of 256: the rows' spike bits are expanded to {0,1}
//    bytes in LDS (row stride 256 B, 16-byte chunks XOR-swizzled by row so the
//    ds_read_b128 of an A fragment is conflict-free); the B fragments stream
//    from the MFMA-tiled codes (snnqp_pack_codes_mfma: one contiguous 1 KiB
//    read per wave and k-step) straight into registers and are reused by the
//    RT row tiles; spike words and B fragments are requested three chunks ahead of
//    their use (register rings, counted s_waitcnt), and inside a chunk every wave
//    interleaves its MFMAs with its share of the loads and of the next chunk's
//    expansion, slot by slot (fused_chunk);
//  * after the K loop the int32 tile goes through LDS once so that each thread
//    gets the T currents of one (sample, feature) pair in order, runs the
//    neuron with u in a register and ballots the spikes into 32-bit words.
#include <type_traits>

#include "kernels.h"

namespace snnqp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK = 256;          // k per chunk (bytes per LDS row)
constexpr int KSC = BK / 32;     // MFMA k-steps per chunk

struct DenseMfmaArgs {
  const uint32_t *x;             // bit-packed rows (words) or uint8 rows (bytes)
  const int32_t *col_sum;        // U8 input: sum over k of the codes of each feature
  int64_t xs_t, xs_b;            // word (bits) / byte (u8) strides
  int32_t T, B, K, N, KS, SB;    // KS = ceil(K / 32), SB samples per workgroup
  const int8_t *wt;              // MFMA-tiled codes [Npad/32][KS][64][16]
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  uint32_t *s_out;
};

__device__ __forceinline__ v4i expand16b(uint32_t b) {
  v4i o;
  o.x = (int)((((b >> 0) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.y = (int)((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.z = (int)((((b >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.w = (int)((((b >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

// Workgroup barrier ordering LDS traffic only (see conv_tile.h): the K loop
// keeps the next chunk's global loads in flight across it.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int a_addr(int row, int c16) {
  return row * BK + ((c16 ^ (row & 15)) << 4);
}

constexpr int KGROUPS = 2;       // wave groups splitting K

template <int RT, int IN>
__global__ void __launch_bounds__(256 * KGROUPS)
dense_mfma_kernel(DenseMfmaArgs a) {
  constexpr bool U8 = IN == SNNQP_U8;
  constexpr int ROWS = RT * 32;
  // staging tasks of a row and chunk: one 32-bit spike word -> 32 bytes, or one 16-byte piece
  // of uint8 counts -> the same 16 bytes ^ 0x80
  constexpr int WPR = U8 ? BK / 16 : BK / 32;
  typedef typename std::conditional<U8, v4i, uint32_t>::type stg_t;
  constexpr int NTASK = ROWS * WPR;
  constexpr int TPT = (NTASK + 255) / 256;
  // two A buffers per group; the epilogue tile (ROWS x 128 int32) overlays them
  constexpr int ABYTES = ROWS * BK;
  constexpr int EBYTES = ROWS * 128 * 4;
  constexpr int LDSB = (2 * KGROUPS * ABYTES > EBYTES) ? 2 * KGROUPS * ABYTES : EBYTES;
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];

  // (group and wave indices in scalar registers: so are the addresses of the B fragments)
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // K group of this wave
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t *abuf = lds + grp * 2 * ABYTES;
  const int n = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * a.SB;
  const int nsamp = min(a.SB, a.B - b0);
  const int rows = nsamp * a.T;                   // live rows of this workgroup
  const int nb = blockIdx.y * 4 + wave;           // 32-column block of this wave
  const bool wave_on = nb * 32 < a.N;
  const int nchunks = (a.KS + KSC - 1) / KSC;      // chunk j of this group = 2 j + grp

  v16i acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    acc[r] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // Global loads run D - 1 chunks ahead of their use, in rings of D register sets (spike
  // words and B fragments): a chunk of MFMAs is about 0.3 us of a wave, a loaded global round
  // trip several times that -- with one chunk of lookahead the K loop ran at the pace of the
  // memory latency, not of the matrix pipe or of LDS.  (uint8 rows stage four times the
  // registers per chunk: one chunk less of lookahead.)
  constexpr int D = U8 ? 3 : 4;
  constexpr int U = D % 2 ? 2 * D : D;             // unroll: ring slot i % D, LDS buffer i & 1
  stg_t stgr[D][TPT];
  // staging tasks: word wi of row `row`, the same (row, wi) for every chunk.  The loads of the
  // K loop are unconditional (clamped addresses, the value dropped afterwards): with loads
  // under lane masks the compiler falls back to s_waitcnt vmcnt(0), which drains the ring.
  uint32_t roff[TPT];                              // word / byte offset within this workgroup's samples
  uint32_t rmask[TPT];
  const uint32_t *xw = a.x + (U8 ? 0 : (int64_t)b0 * a.xs_b);
  const uint8_t *xbytes = (const uint8_t *)a.x + (U8 ? (int64_t)b0 * a.xs_b : 0);
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int task = tid + k * 256;
    const int row = task / WPR;
    const bool live = task < NTASK && row < rows;
    rmask[k] = live ? 0xFFFFFFFFu : 0u;
    roff[k] = 0;
    if (live) {
      const int bl = row / a.T, t = row - bl * a.T;
      roff[k] = (uint32_t)((int64_t)t * a.xs_t + (int64_t)bl * a.xs_b);   // < 2^31: launch check
    }
  }
  auto chunk_word = [&](int chunk, int k) { return (chunk * KGROUPS + grp) * WPR + (tid + k * 256) % WPR; };
  // uint8: 16-byte piece `chunk_word` of the row; K is a multiple of 16 (launch check).  Pieces
  // beyond K re-read the row's first one and are stored as zero bytes (the k-steps of a chunk
  // beyond K reuse the last real k-step's codes); the rows beyond `rows` need no masking,
  // nothing reads their accumulators
  auto stage_load1 = [&](int chunk, int k) -> stg_t {
    if constexpr (U8) {
      const int piece = chunk_word(chunk, k);
      return *(const v4i *)(xbytes + roff[k] + (uint32_t)(piece * 16 < a.K ? piece * 16 : 0));
    } else {
      return xw[roff[k] + (uint32_t)min(chunk_word(chunk, k), a.KS - 1)];
    }
  };
  auto stage_load = [&](stg_t (&stg)[TPT], int chunk) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) stg[k] = stage_load1(chunk, k);
  };
  // the words of dead rows and of k beyond K are zeroed here, by masks (a select next to the
  // load would put its wait there)
  auto stage_store = [&](const stg_t (&stg)[TPT], int chunk, int buf) {
    uint8_t *base = abuf + buf * ABYTES;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int task = tid + k * 256;
      if (task < NTASK) {
        const int row = task / WPR, wi = task % WPR;
        if constexpr (U8) {
          const v4i v = stg[k];
          const int m = chunk_word(chunk, k) * 16 < a.K ? -1 : 0;
          *(v4i *)(base + a_addr(row, wi)) =
              v4i{(v.x ^ (int)0x80808080) & m, (v.y ^ (int)0x80808080) & m,
                  (v.z ^ (int)0x80808080) & m, (v.w ^ (int)0x80808080) & m};
        } else {
          const uint32_t wv = stg[k] & rmask[k] & (uint32_t)((chunk_word(chunk, k) - a.KS) >> 31);
          *(v4i *)(base + a_addr(row, wi * 2)) = expand16b(wv & 0xFFFFu);
          *(v4i *)(base + a_addr(row, wi * 2 + 1)) = expand16b(wv >> 16);
        }
      }
    }
  };

  // B fragments of a whole chunk (KSC k-steps) are prefetched into registers next to the A
  // words of that chunk
  const v4i *wtile = (const v4i *)a.wt + ((int64_t)(wave_on ? nb : 0) * a.KS) * 64;   // wave-uniform
  v4i bfr[D][KSC];
  auto load_b = [&](v4i (&bf)[KSC], int chunk) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      // (k-steps beyond K meet zero A bytes: any codes do)
      const int kg = min((chunk * KGROUPS + grp) * KSC + ks, a.KS - 1);
      bf[ks] = (wtile + (int64_t)kg * 64)[lane];
    }
  };
  // One chunk of the K loop, as NSLOT = KSC * RT slots: slot s issues MFMA s of chunk c, the
  // A-fragment read of slot s + PF, and an even share of everything else the iteration has to
  // do -- the global loads of chunk c + D - 1 and the expansion of the spike words of chunk
  // c + 1 into the other LDS buffer (one dword of 4 bytes per piece: bit field, multiply, mask).
  // Vector instructions only overlap the matrix pipe when the SAME wave issues them between
  // its MFMAs: with the chunk as "all MFMAs, then all staging" the two never co-executed
  // (SQ_VALU_MFMA_COEXEC_CYCLES 2 % of the MFMA-busy cycles, each of MFMA / VALU / LDS busy a
  // third of the time), whichever way the two waves of a SIMD were phased.
  constexpr int NSLOT = KSC * RT;
  constexpr int PF = 3;                               // A fragments in flight
  constexpr int LSTEP = NSLOT / TPT;                  // slots between two staging loads
  static_assert(NSLOT % TPT == 0 && LSTEP >= 2, "one staging load every LSTEP slots");
  constexpr int PPT = U8 ? 4 : 8;                     // dwords a staging task expands to
  constexpr int NPIECE = TPT * PPT;                   // dwords the thread expands per chunk
  // loop-invariant LDS offsets: fragment reads (per k-step; row tile and buffer are immediates)
  // and the 16-byte stores of each staging task
  int rd_off[KSC];
#pragma unroll
  for (int ks = 0; ks < KSC; ++ks) rd_off[ks] = a_addr(n, ks * 2 + h);
  int wr_off[TPT][2];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int task = tid + k * 256;
    const int row = task < NTASK ? task / WPR : 0, wi = task % WPR;
    wr_off[k][0] = U8 ? a_addr(row, wi) : a_addr(row, wi * 2);
    wr_off[k][1] = U8 ? 0 : a_addr(row, wi * 2 + 1);
  }
  auto frag = [&](int rbuf, int s) -> v4i {
    return *(const v4i *)(abuf + rbuf * ABYTES + (s % RT) * 32 * BK + rd_off[s / RT]);
  };
  auto fused_chunk = [&](int rbuf, const v4i (&bf)[KSC], stg_t (&ld_stg)[TPT], v4i (&ld_bf)[KSC],
                         int ld_chunk, const stg_t (&st_stg)[TPT], int st_chunk) {
    v4i av[PF + 1];
#pragma unroll
    for (int s = 0; s < PF; ++s) av[s] = frag(rbuf, s);
    stg_t wv[TPT];
    int inK[TPT];                      // uint8: all ones while the piece lies inside K
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      inK[k] = -1;
      if constexpr (U8) {
        wv[k] = st_stg[k];
        inK[k] = chunk_word(st_chunk, k) * 16 < a.K ? -1 : 0;
      } else {
        wv[k] = st_stg[k] & rmask[k] & (uint32_t)((chunk_word(st_chunk, k) - a.KS) >> 31);
      }
    }
    v4i ex = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      // (a wave beyond N multiplies block 0's codes into accumulators nobody reads)
      acc[s % RT] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s % (PF + 1)], bf[s / RT], acc[s % RT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + PF < NSLOT) av[(s + PF) % (PF + 1)] = frag(rbuf, s + PF);
      // global loads of the chunk D - 1 ahead: one B fragment per row-tile round, the spike
      // words in the first slots
      if (s % RT == 0) {
        const int ks = s / RT;
        const int kg = min((ld_chunk * KGROUPS + grp) * KSC + ks, a.KS - 1);
        ld_bf[ks] = (wtile + (int64_t)kg * 64)[lane];
      }
      if (s % LSTEP == 1 && s / LSTEP < TPT) {
        const int k = s / LSTEP;
        ld_stg[k] = stage_load1(ld_chunk, k);
      }
      // expansion pieces p with p * NSLOT / NPIECE == s
#pragma unroll
      for (int p = 0; p < NPIECE; ++p) {
        if (p * NSLOT / NPIECE == s) {
          const int k = p / PPT, d = p & 3;
          // the word passes through an empty volatile asm in every slot that expands a piece
          // of it: the piece cannot be computed before its slot (sched_barrier alone orders
          // instructions, not the values they were selected from)
          if constexpr (U8) {
            asm volatile("" : "+v"(wv[k]));
            ex[d] = (wv[k][d] ^ (int)0x80808080) & inK[k];
            if (d == 3 && tid + k * 256 < NTASK)
              *(v4i *)(abuf + (rbuf ^ 1) * ABYTES + wr_off[k][0]) = ex;
          } else {
            const int hf = (p / 4) & 1;
            asm volatile("" : "+v"(wv[k]));
            ex[d] = (int)((((wv[k] >> (16 * hf + 4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
            if (d == 3 && tid + k * 256 < NTASK)
              *(v4i *)(abuf + (rbuf ^ 1) * ABYTES + wr_off[k][hf]) = ex;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // chunks beyond K contribute zero A bytes (any codes do), so the loop runs over groups of U
  // chunks without a tail case
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    stage_load(stgr[d], d);
    load_b(bfr[d], d);
  }
  stage_store(stgr[0], 0, 0);
  lds_barrier();
  const int ngc = (nchunks + KGROUPS - 1) / KGROUPS;     // chunks per group
  for (int c = 0; c < ngc; c += U) {
#pragma unroll
    for (int i = 0; i < U; ++i) {                        // chunk c + i: ring slot i % D, LDS buffer i & 1
      fused_chunk(i & 1, bfr[i % D], stgr[(i + D - 1) % D], bfr[(i + D - 1) % D], c + i + D - 1,
                  stgr[(i + 1) % D], c + i + 1);
      lds_barrier();
    }
  }

  // int32 tile -> LDS [row][128]; C/D layout: col = lane & 31,
  // row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  // (group 1 stores its partial sums, group 0 adds its own on top: exact int32)
  int *et = (int *)lds;
#pragma unroll
  for (int g = KGROUPS - 1; g >= 0; --g) {
    if (wave_on && grp == g) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          int *e = et + row * 128 + wave * 32 + n;
          *e = (g == KGROUPS - 1) ? acc[r][i] : *e + acc[r][i];
        }
    }
    lds_barrier();
  }

  // one (sample, feature) pair per thread and pass; 64 consecutive features of
  // one sample per wave, so a ballot is two output words
  const int CW = (a.N + 31) >> 5;
  for (int p = threadIdx.x; p < a.SB * 128; p += 256 * KGROUPS) {
    const int bl = p >> 7, col = p & 127;
    const int feat = blockIdx.y * 128 + col;
    const bool live = bl < nsamp && feat < a.N;
    float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f, u = 0.0f;
    int off = 0;                       // uint8 input: 128 * sum_k w[k][feat] (the x - 128 operand)
    if (live) {
      if (U8) off = 128 * a.col_sum[feat];
      if (a.bn.mean) { bmean = a.bn.mean[feat]; bmul = a.bn.mul[feat]; bbias = a.bn.bias[feat]; }
      if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[feat];
      if (a.u0) u = a.u0[(int64_t)(b0 + bl) * a.N + feat];
    }
    for (int t = 0; t < a.T; ++t) {
      bool s = false;
      if (live) {
        float cur = dequant_acc(et[(bl * a.T + t) * 128 + col] + off, a.dq);
        if (a.bn.mean) cur = bn_apply(cur, bmean, bmul, bbias);
        s = neuron_step(u, cur, a.nrn, dec);
      }
      const unsigned long long m = __ballot(s);
      const int word = (blockIdx.y * 128 + (col & 64)) >> 5;     // wave-uniform
      if (bl < nsamp) {
        uint32_t *o = a.s_out + ((int64_t)t * a.B + (b0 + bl)) * CW;
        if (lane == 0 && word < CW) o[word] = (uint32_t)m;
        if (lane == 32 && word + 1 < CW) o[word + 1] = (uint32_t)(m >> 32);
      }
    }
    if (live && a.u_out) a.u_out[(int64_t)(b0 + bl) * a.N + feat] = u;
  }
}

const char *dense_mfma_unsupported(int in_type, int32_t K, int32_t N,
                                   const snnqp_weight_t *w, const int8_t *wt,
                                   const snnqp_neuron_t *nrn, int s_type) {
  if (w->wtype != SNNQP_W_I8) return "weights are not int8 codes";
  if (!wt) return "MFMA-tiled codes `wt` not given";
  if (in_type != SNNQP_BITS && in_type != SNNQP_U8) return "input must be bit-packed or uint8";
  if (s_type != SNNQP_BITS) return "spike output must be bit-packed";
  if (in_type == SNNQP_U8) {
    // uint8 rows are read in 16-byte pieces, as x - 128 against the int8 codes
    if (!w->col_sum) return "uint8 input needs snnqp_weight_t.col_sum";
    if (K % 16) return "uint8 input needs K % 16 == 0";
    if (K > 65536) return "uint8 input needs K <= 65536 (int32 accumulator)";
  }
  // bit-packed: any K -- the tiles `wt` are zero-padded to ceil(K / 32) k-steps and the packed
  // input rows carry zero bits beyond K
  if (nrn->kind == SNNQP_NEURON_LIF && !nrn->decay) return "LIF without decay";
  return nullptr;
}

template <int RT, int IN>
static void launch_dense(const DenseMfmaArgs &a, unsigned gx, unsigned gy, hipStream_t st) {
  hipLaunchKernelGGL((dense_mfma_kernel<RT, IN>), dim3(gx, gy), dim3(256 * KGROUPS), 0, st, a);
}

int run_dense_mfma(const void *x, int in_type, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B,
                   int32_t K, int32_t N, const snnqp_weight_t *w, const int8_t *wt,
                   const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                   float *u_out, uint32_t *s_out, hipStream_t st) {
  SNNQP_REQUIRE((x && s_out) || T == 0 || B == 0, SNNQP_EINVAL, "dense mfma: null pointer");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "dense mfma: negative T/B");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  SNNQP_CHECK_BN(bn);
  if (T == 0 || B == 0) return SNNQP_OK;
  SNNQP_REQUIRE(T <= 96, SNNQP_EUNSUPPORTED, "dense mfma: T > 96");
  const bool u8 = in_type == SNNQP_U8;
  if (u8)
    SNNQP_REQUIRE(((uintptr_t)x & 15) == 0 && xs_t % 16 == 0 && xs_b % 16 == 0, SNNQP_EUNSUPPORTED,
                  "dense mfma: uint8 rows must be 16-byte aligned");
  DenseMfmaArgs a;
  a.x = (const uint32_t *)x; a.xs_t = xs_t; a.xs_b = xs_b;
  a.col_sum = w->col_sum;
  a.T = T; a.B = B; a.K = K; a.N = N; a.KS = (K + 31) / 32;
  a.wt = wt;
  a.dq = make_dequant(w->L, w->m);
  a.bn = make_bn(bn); a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out;
  const unsigned gy = (unsigned)((N + 127) / 128);
  // largest row tile that still gives the chip enough workgroups, else the
  // smallest one that holds a whole sample (most workgroups); uint8 rows: at most 64 rows
  // (their staging ring takes the registers of the third row tile)
  static const int rts[3] = {3, 2, 1};
  int rt = 0;
  for (int i = u8 ? 1 : 0; i < 3; ++i) {
    const int sb = rts[i] * 32 / T;
    if (sb < 1) continue;
    rt = rts[i];
    if ((int64_t)((B + sb - 1) / sb) * gy >= 200) break;
  }
  SNNQP_REQUIRE(rt > 0, SNNQP_EUNSUPPORTED, "dense mfma: T too large");
  a.SB = rt * 32 / T;
  const unsigned gx = (unsigned)((B + a.SB - 1) / a.SB);
  if (u8) {
    if (rt == 2) launch_dense<2, SNNQP_U8>(a, gx, gy, st);
    else launch_dense<1, SNNQP_U8>(a, gx, gy, st);
  } else {
    switch (rt) {
      case 3: launch_dense<3, SNNQP_BITS>(a, gx, gy, st); break;
      case 2: launch_dense<2, SNNQP_BITS>(a, gx, gy, st); break;
      default: launch_dense<1, SNNQP_BITS>(a, gx, gy, st); break;
    }
  }
  SNNQP_CHECK_LAUNCH("dense_mfma_kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
