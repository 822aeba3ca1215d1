// Fused SpikingBlock(QuantDense, neuron) (spiking_learning.py:446-462 with flax_qdense.py:87
// as the connection) for codes of magnitude <= 7 (DuQ up to 4 bits, quant.py:443,467) on the
// block-scaled f8f6f4 MFMA: bit-packed spikes as fp4 (e2m1: 0 / 1.0) x codes as fp6 (e2m3:
// every integer up to 7 exact) -> float32 sums of integers below 2^24, i.e. the same integer
// the int8 kernel (dense_mfma.hip) accumulates -> dequantise -> [BatchNorm] -> neuron over T
// -> packed spikes.  Against the int8 formulation: K = 64 per MFMA at twice the rate (a
// quarter of the matrix time), the codes PACKED at 6 bits in HBM / L2 (snnqp_pack_codes_fp6:
// three quarters of the bytes every workgroup streams), and the spike bits expanded through
// a byte -> 8-nibble LDS table (one lookup per 8 spikes; bits -> bytes took three vector
// instructions per 4 spikes).
//
// Same decomposition as dense_mfma.hip: a workgroup owns SB samples x all T steps (<= RT * 32
// rows) and 128 output features; 8 waves = two groups of 4 (32 features per wave) that split
// K (group g walks the chunks c = g mod 2, own LDS buffers; the partial tiles are added once
// at the end -- exact: integers).  K is walked in chunks of 256: the rows' spike words become
// 128 bytes of fp4 per row in LDS (16-byte pieces XOR-swizzled by row: the ds_read_b128 of an
// A fragment is conflict-free), the B fragments (24 bytes per lane and k-step) stream from the
// fp6 tiles into registers and are reused by the RT row tiles.  A chunk is only 4 RT MFMAs, so
// the K loop must not drain at its barrier: THREE A buffers per group -- chunk c computes from
// buffer c mod 3 while the words of chunk c + 2 are expanded into buffer (c + 2) mod 3, so the
// barrier at the end of chunk c publishes a buffer that is first read a whole chunk later, and
// the first fragments of chunk c + 1 are requested BEFORE that barrier (its wait is counted:
// it retires the chunk's LDS writes, not the fragments in flight).  Spike words are requested
// four chunks ahead, B fragments two (register rings of three), and every wave interleaves its
// MFMAs with its share of the loads and of the expansion, slot by slot.
#include <cstdlib>

#include "kernels.h"

namespace snnqp {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const uint32_t lds_cu32_t;

namespace {

constexpr int F6_BK = 256;            // k per chunk
constexpr int F6_KSC = F6_BK / 64;    // MFMA k-steps per chunk
constexpr int F6_ROWB = F6_BK / 2;    // LDS bytes of a row per chunk (fp4)
constexpr int F6_TAB = 256 * 32 * 4;  // byte -> 8 nibbles, 32 interleaved copies (conv3x3_bits.hip)
constexpr int F6_KGROUPS = 2;
constexpr int F6_TILE = 1536;         // bytes of one B tile: 64 lanes x (16 + 8)

struct DenseFp6Args {
  const uint32_t *x;
  int64_t xs_t, xs_b;            // word strides
  int32_t T, B, K, N, KW, KS, SB;   // KW = ceil(K / 32) words per row, KS = ceil(K / 64) k-steps
  const uint8_t *wt6;            // fp6 tiles [Npad/32][KS][1536] (snnqp_pack_codes_fp6)
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  uint32_t *s_out;
  // K split over workgroups (blockIdx.z): partial tiles and tickets in a caller-owned workspace
  int32_t ksplit, gcz;           // workgroups per tile; group-chunks of one of them
  uint32_t *tickets;             // [tiles], zeroed on the stream in front of every launch (run_dense_fp6)
  float *slabs;                  // [tiles][ksplit][ROWS * 128]
  uint32_t *status;              // the device's status word (runtime.hip), or null
};

__device__ __forceinline__ void lds_barrier6() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 16-byte piece c (of 8) of row `row`: pieces swizzled by (row >> 1) & 7, so that the 16 lanes
// a ds_read_b128 serves per cycle ({0-3, 12-15, 20-27}, ...) hit 16 different bank quads
// (quad = 8 (row & 1) + (c ^ (row >> 1) & 7))
__device__ __forceinline__ int a6_addr(int row, int c) {
  return row * F6_ROWB + ((c ^ ((row >> 1) & 7)) << 4);
}

// 4 int8 codes (|c| <= 7) -> 4 e2m3 codes, one per byte (conv3x3_bits.hip)
__device__ __forceinline__ uint32_t fp6_codes4(uint32_t x) {
  const uint32_t m1 = (x >> 7) & 0x01010101u;
  const uint32_t mag = (x ^ (m1 * 0xFFu)) + m1;
  const uint32_t code = __builtin_amdgcn_perm(0x1E1C1A18u, 0x14100800u, mag);
  return code | (m1 << 5);
}
__device__ __forceinline__ uint32_t squeeze6(uint32_t c) {
  return (c & 0x3Fu) | ((c >> 2) & 0xFC0u) | ((c >> 4) & 0x3F000u) | ((c >> 6) & 0xFC0000u);
}

}  // namespace

// fp6 tiles of a dense kernel: for the 32-column block nb and the 64-deep k-step ks, lane
// l = (n & 31) + 32 h holds the codes of k = 64 ks + 32 h + j (j < 32) of column
// n = 32 nb + (l & 31) as e2m3 values, value j at bits [6 j, 6 j + 6) of six dwords; the tile
// stores dwords 0..3 of the 64 lanes (1 KiB), then dwords 4..5 (512 B): two coalesced reads
// per wave and k-step.  Rows beyond K and columns beyond N are zero codes.
__global__ void __launch_bounds__(256)
pack_codes_fp6_kernel(const int8_t *__restrict__ w, int64_t K, int32_t N, int32_t Npad,
                      uint32_t *__restrict__ wt6) {
  const int64_t KS = (K + 63) / 64;
  const int64_t total = (int64_t)(Npad / 32) * KS * 64;          // one thread per (tile, lane)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int64_t tile = i >> 6;
    const int64_t ks = tile % KS, nb = tile / KS;
    const int64_t n = nb * 32 + (lane & 31);
    const int64_t k0 = ks * 64 + 32 * (lane >> 5);
    uint32_t t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t k = k0 + 4 * q + j;
        const int c = (n < N && k < K) ? (int)w[k * N + n] : 0;
        x |= (uint32_t)(uint8_t)c << (8 * j);
      }
      t[q] = squeeze6(fp6_codes4(x));
    }
    uint32_t d[6];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      d[3 * g + 0] = t[4 * g] | (t[4 * g + 1] << 24);
      d[3 * g + 1] = (t[4 * g + 1] >> 8) | (t[4 * g + 2] << 16);
      d[3 * g + 2] = (t[4 * g + 2] >> 16) | (t[4 * g + 3] << 8);
    }
    uint32_t *tb = wt6 + tile * (F6_TILE / 4);
    tb[lane * 4 + 0] = d[0]; tb[lane * 4 + 1] = d[1]; tb[lane * 4 + 2] = d[2]; tb[lane * 4 + 3] = d[3];
    tb[256 + lane * 2 + 0] = d[4]; tb[256 + lane * 2 + 1] = d[5];
  }
}

template <int RT>
__global__ void __launch_bounds__(256 * F6_KGROUPS)
dense_fp6_kernel(DenseFp6Args a) {
  constexpr int ROWS = RT * 32;
  constexpr int WPR = F6_BK / 32;                 // spike words per row per chunk
  constexpr int NTASK = ROWS * WPR;
  constexpr int TPT = NTASK / 256;                // = RT
  static_assert(TPT == RT, "one spike word per thread and row tile");
  constexpr int ABYTES = ROWS * F6_ROWB;
  constexpr int EBYTES = ROWS * 128 * 4;
  constexpr int NBUF = 3;                         // A buffers of a group (see the header)
  constexpr int WORK = (NBUF * F6_KGROUPS * ABYTES > EBYTES) ? NBUF * F6_KGROUPS * ABYTES : EBYTES;
  __shared__ __attribute__((aligned(128))) uint8_t lds[F6_TAB + WORK];
  uint8_t *work = lds + F6_TAB;

  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // K group of this wave
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t *abuf = work + grp * NBUF * ABYTES;
  const int n = lane & 31, h = lane >> 5;
  const int b0 = blockIdx.x * a.SB;
  const int nsamp = min(a.SB, a.B - b0);
  const int rows = nsamp * a.T;                   // live rows of this workgroup
  const int nb = blockIdx.y * 4 + wave;           // 32-column block of this wave
  const bool wave_on = nb * 32 < a.N;
  const int nchunks = (a.KS + F6_KSC - 1) / F6_KSC;   // chunk j of this group = 2 j + grp
  const int ngc_all = (nchunks + F6_KGROUPS - 1) / F6_KGROUPS;  // chunks per group over all of K
  // K split over workgroups: this one walks the group-chunks [gc0, gc0 + ngc)
  const int gc0 = a.ksplit > 1 ? (int)blockIdx.z * a.gcz : 0;
  const int ngc = a.ksplit > 1 ? max(0, min(a.gcz, ngc_all - gc0)) : ngc_all;
  // Every workgroup streams the same code tiles.  Walking K in the same order, all of them ask
  // for the same lines at the same time; the sums are exact integers, so the walk may start
  // anywhere: step `lc` of the loop is chunk (lc + rot) mod ngc, with one `rot` per XCD
  // (blockIdx.x mod 8 under the observed round-robin placement -- a speed assumption only).
  // The workgroups of an XCD still share each tile in its L2 (measured HBM-side traffic 106 MB
  // against 104 MB unrotated; a rotation per WORKGROUP thrashes the 4 MiB L2 with 32 different
  // tiles at a time: 162 MB), the eight XCDs no longer hit the same memory channels together:
  // 0.072 ms against 0.078.
  const int rot = (int)(((blockIdx.x & 7u) * (unsigned)ngc) >> 3);
  auto phys = [&](int lc) -> int {     // loop step -> this group's chunk; steps beyond ngc: a dead chunk
    const int pc = lc + rot >= ngc ? lc + rot - ngc : lc + rot;
    return lc < ngc ? gc0 + pc : nchunks;
  };

  // table: byte -> 8 nibbles (bit i set -> 1.0 = 0x2 in nibble i), 32 copies: entry e of copy c
  // at dword 32 e + c, so lane l of a 32-lane group reads bank l whatever its byte is
  for (int i = threadIdx.x; i < 256 * 32; i += 256 * F6_KGROUPS) {
    const int e = i >> 5;
    uint32_t v = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) v |= ((e >> bit) & 1) ? (0x2u << (4 * bit)) : 0u;
    ((uint32_t *)lds)[i] = v;
  }
  typedef __attribute__((address_space(3))) const uint8_t lds_cu8_t;
  const uint32_t tabl = (uint32_t)(uintptr_t)(lds_cu8_t *)lds + (uint32_t)(lane & 31) * 4;

  v16f acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
    acc[r] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // rings of D = 3 register sets, slot = chunk mod 3 like the LDS buffer: the spike words of
  // chunk c + 4 are requested while those of c + 2 are expanded (c + 3 waits), the B fragments
  // of chunk c + 2 while those of c are multiplied (c + 1 waits)
  constexpr int D = 3;
  uint32_t stgr[D][TPT];
  uint32_t roff[TPT], rmask[TPT];
  const uint32_t *xw = a.x + (int64_t)b0 * a.xs_b;
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int task = tid + k * 256;
    const int row = task / WPR;
    const bool live = row < rows;
    rmask[k] = live ? 0xFFFFFFFFu : 0u;
    roff[k] = 0;
    if (live) {
      const int bl = row / a.T, t = row - bl * a.T;
      roff[k] = (uint32_t)((int64_t)t * a.xs_t + (int64_t)bl * a.xs_b);   // < 2^31: launch check
    }
  }
  // word `chunk_word` of the row; words beyond K re-read the row's last one and are masked
  auto chunk_word = [&](int chunk, int k) { return (chunk * F6_KGROUPS + grp) * WPR + (tid + k * 256) % WPR; };
  auto stage_load1 = [&](int chunk, int k) -> uint32_t {
    return xw[roff[k] + (uint32_t)min(chunk_word(chunk, k), a.KW - 1)];
  };
  auto masked = [&](uint32_t w, int chunk, int k) -> uint32_t {
    return w & rmask[k] & (uint32_t)((chunk_word(chunk, k) - a.KW) >> 31);
  };
  auto lookup = [&](uint32_t w, int d) -> int {
    return (int)*(lds_cu32_t *)(uintptr_t)((((w >> (8 * d)) & 0xFFu) << 7) + tabl);
  };
  int wr_off[TPT];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int task = tid + k * 256;
    wr_off[k] = a6_addr(task / WPR, task % WPR);
  }

  const uint8_t *wtile = a.wt6 + ((int64_t)(wave_on ? nb : 0) * a.KS) * F6_TILE;   // wave-uniform
  int bfr[D][F6_KSC][6];
  auto load_b1 = [&](int (&bf)[6], int chunk, int ks) {
    // (k-steps beyond K meet zero A nibbles: any codes do)
    const int kg = min((chunk * F6_KGROUPS + grp) * F6_KSC + ks, a.KS - 1);
    const uint8_t *tb = wtile + (int64_t)kg * F6_TILE;
    const v4i lo = ((const v4i *)tb)[lane];
    const v2i hi = ((const v2i *)(tb + 1024))[lane];
    bf[0] = lo.x; bf[1] = lo.y; bf[2] = lo.z; bf[3] = lo.w; bf[4] = hi.x; bf[5] = hi.y;
  };

  constexpr int NSLOT = F6_KSC * RT;                  // MFMAs of one chunk
  constexpr int PF = 3;                               // A fragments in flight
  int rd_off[F6_KSC];
#pragma unroll
  for (int ks = 0; ks < F6_KSC; ++ks) rd_off[ks] = a6_addr(n, ks * 2 + h);
  auto frag = [&](int rbuf, int s) -> v4i {
    // row tile s % RT: 32 rows further (the swizzle repeats every 16 rows)
    return *(const v4i *)(abuf + rbuf * ABYTES + (s % RT) * 32 * F6_ROWB + rd_off[s / RT]);
  };
  // One chunk as NSLOT slots: slot s issues MFMA s and the A-fragment read of slot s + PF --
  // which, in the last PF slots, is fragment s + PF - NSLOT of the NEXT chunk, from its own
  // buffer -- plus an even share of the chunk's other work: the global loads (spike words of
  // chunk c + 4, B fragments of chunk c + 2) and, in the slots before the last PF, the table
  // lookups / LDS writes that expand the spike words of chunk c + 2 into buffer (c + 2) mod 3.
  constexpr int NPIECE = TPT * 5;                     // per word: 4 lookups, 1 write
  constexpr int XSLOT = NSLOT - PF;                   // slots that carry expansion pieces
  static_assert(NSLOT % (PF + 1) == 0, "the fragment ring keeps its phase from chunk to chunk");
  v4i av[PF + 1];
  auto fused_chunk = [&](int rbuf, const int (&bf)[F6_KSC][6], uint32_t (&ld_stg)[TPT], int ld_schunk,
                         int (&ld_bf)[F6_KSC][6], int ld_bchunk, const uint32_t (&st_stg)[TPT],
                         int st_chunk) {
    const int nbuf = rbuf == NBUF - 1 ? 0 : rbuf + 1, wbuf = nbuf == NBUF - 1 ? 0 : nbuf + 1;
    uint32_t wv[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) wv[k] = masked(st_stg[k], st_chunk, k);
    v4i ex = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const v4i af = av[s % (PF + 1)];
      const int ks = s / RT;
      acc[s % RT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
          v8i{af.x, af.y, af.z, af.w, 0, 0, 0, 0},
          v8i{bf[ks][0], bf[ks][1], bf[ks][2], bf[ks][3], bf[ks][4], bf[ks][5], 0, 0}, acc[s % RT],
          4 /* A: fp4 */, 2 /* B: fp6 */, 0, 127, 0, 127);
      __builtin_amdgcn_sched_barrier(0);
      if (s + PF < NSLOT) av[(s + PF) % (PF + 1)] = frag(rbuf, s + PF);
      else av[(s + PF) % (PF + 1)] = frag(nbuf, s + PF - NSLOT);
      if (s % RT == 0) load_b1(ld_bf[s / RT], ld_bchunk, s / RT);
      if (s % F6_KSC == 1 && s / F6_KSC < TPT) ld_stg[s / F6_KSC] = stage_load1(ld_schunk, s / F6_KSC);
#pragma unroll
      for (int p = 0; p < NPIECE; ++p) {
        if (p * XSLOT / NPIECE == s) {
          const int k = p / 5, d = p % 5;
          asm volatile("" : "+v"(wv[k]));
          if (d < 4) ex[d] = lookup(wv[k], d);
          else *(v4i *)(abuf + wbuf * ABYTES + wr_off[k]) = ex;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // The barrier publishes buffer wbuf, which nobody reads before the chunk after next.  LDS
    // operations of a wave complete in order and the PF youngest are the fragment reads of the
    // next chunk: waiting until only those are outstanding retires every write of this chunk
    // without draining the fragments (no scalar load is in flight: the loop issues none).
    asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" : : "n"(PF) : "memory");
  };

  lds_barrier6();                                  // the table is visible
  // prologue: words of chunks 0..3, B fragments of chunks 0..1 requested; chunks 0 and 1 expanded
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) stgr[d][k] = stage_load1(phys(d), k);
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int ks = 0; ks < F6_KSC; ++ks) load_b1(bfr[d][ks], phys(d), ks);
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const uint32_t w = masked(stgr[d][k], phys(d), k);
      *(v4i *)(abuf + d * ABYTES + wr_off[k]) = v4i{lookup(w, 0), lookup(w, 1), lookup(w, 2), lookup(w, 3)};
    }
  }
  // (ring slots 0 and 1 are free again: chunk 3 goes to slot 0 now, chunk 4 to slot 1 in step 0)
#pragma unroll
  for (int k = 0; k < TPT; ++k) stgr[0][k] = stage_load1(phys(3), k);
  lds_barrier6();
#pragma unroll
  for (int s = 0; s < PF; ++s) av[s] = frag(0, s);
  for (int c = 0; c < ngc; c += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      // step c + i: compute from buffer i, B slot i; request the words of chunk c + i + 4 into
      // slot (i + 1) % 3 and the B fragments of chunk c + i + 2 into slot (i + 2) % 3; expand
      // the words of chunk c + i + 2 (slot (i + 2) % 3) into buffer (i + 2) % 3
      fused_chunk(i, bfr[i], stgr[(i + 1) % D], phys(c + i + 4), bfr[(i + 2) % D], phys(c + i + 2),
                  stgr[(i + 2) % D], phys(c + i + 2));
    }
  }
  lds_barrier6();                                  // every wave is done with the A buffers

  // partial tiles -> LDS [row][128] as float (exact integers; group 1 stores, group 0 adds);
  // C/D layout: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  float *et = (float *)work;
#pragma unroll
  for (int g = F6_KGROUPS - 1; g >= 0; --g) {
    if (wave_on && grp == g) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          float *e = et + row * 128 + wave * 32 + n;
          *e = (g == F6_KGROUPS - 1) ? acc[r][i] : *e + acc[r][i];
        }
    }
    lds_barrier6();
  }

  // K split over workgroups: every workgroup of a tile leaves its partial tile (exact integers
  // in float32) in its slab and draws a ticket; the one that draws the last adds the others'
  // slabs to its own tile and goes on to the neuron, the others are done.  The hand-off recipe of
  // cdna_hip_programming.md (R1): the slab by agent-scope relaxed stores (write-through), every
  // storing wave drains, one relaxed agent-scope ticket, the last arriver reads with agent-scope
  // loads -- no release / acquire fence: on this part a fence pair writes back / invalidates an
  // L2 and cost ~ 20 us per launch (round 4's first version: 0.075 ms against 0.066 unsplit).
  if (a.ksplit > 1) {
    typedef unsigned long long u64;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    u64 *mine = (u64 *)(a.slabs + ((int64_t)tile * a.ksplit + blockIdx.z) * (ROWS * 128));
    const u64 *et64 = (const u64 *)et;
    for (int i = threadIdx.x; i < ROWS * 64; i += 256 * F6_KGROUPS)
      __hip_atomic_store(mine + i, et64[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t *flag = (uint32_t *)lds;               // (the byte -> nibble table is no longer needed)
    if (threadIdx.x == 0) {
      const uint32_t ticket = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket >= (uint32_t)a.ksplit && a.status) *(volatile uint32_t *)a.status = SNNQP_STATUS_TICKET;
      const bool last = ticket + 1u == (uint32_t)a.ksplit;
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    for (int z = 0; z < a.ksplit; ++z) {
      if (z == (int)blockIdx.z) continue;
      const u64 *other = (const u64 *)(a.slabs + ((int64_t)tile * a.ksplit + z) * (ROWS * 128));
      for (int i = threadIdx.x; i < ROWS * 64; i += 256 * F6_KGROUPS) {
        const u64 o = __hip_atomic_load(other + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        et[2 * i] += __uint_as_float((uint32_t)o);
        et[2 * i + 1] += __uint_as_float((uint32_t)(o >> 32));
      }
    }
    __syncthreads();
  }

  const int CW = (a.N + 31) >> 5;
  for (int p = threadIdx.x; p < a.SB * 128; p += 256 * F6_KGROUPS) {
    const int bl = p >> 7, col = p & 127;
    const int feat = blockIdx.y * 128 + col;
    const bool live = bl < nsamp && feat < a.N;
    float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f, u = 0.0f;
    if (live) {
      if (a.bn.mean) { bmean = a.bn.mean[feat]; bmul = a.bn.mul[feat]; bbias = a.bn.bias[feat]; }
      if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[feat];
      if (a.u0) u = a.u0[(int64_t)(b0 + bl) * a.N + feat];
    }
    for (int t = 0; t < a.T; ++t) {
      bool s = false;
      if (live) {
        float cur = div_exact(et[(bl * a.T + t) * 128 + col], a.dq) * a.dq.m;
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

template <int RT>
static void launch_dense_fp6(const DenseFp6Args &a, unsigned gx, unsigned gy, hipStream_t st) {
  hipLaunchKernelGGL((dense_fp6_kernel<RT>), dim3(gx, gy, (unsigned)(a.ksplit > 1 ? a.ksplit : 1)),
                     dim3(256 * F6_KGROUPS), 0, st, a);
}

// Rows of a workgroup and workgroups per tile along K.  A workgroup streams the code matrix of
// its K range through its CU's vector L1, which is what bounds this kernel: more row tiles
// amortise that stream, but only while the grid still covers the chip; a split of K over `ks`
// workgroups per tile (partial tiles and tickets in a caller workspace, the last arriver of a
// tile runs the neuron) fills the chip when the batch alone does not.  The model behind the
// choice, in microseconds at K = 32768, fitted to the read-out (T = 20): a launch takes (rounds of
// workgroups over the 256 CUs) x (11 + (9.5 + 15.5 RT) / ks) -- B = 1024 unsplit: RT 3 / 4 / 5 =
// 0.067 / 0.084 / 0.098 ms -- plus, when split, the hand-over, 6 + 0.09 per KB of slab moved.
// Measured (write-through hand-over): B = 16 / 64 / 100 split (RT 1, ks 4) 0.025 / 0.027 / 0.031 ms
// against 0.042 / 0.043 / 0.044 unsplit; B = 1024 forced to (RT 5, ks 2) 0.071, (3, 2) 0.080,
// (5, 4) 0.096 against 0.066 unsplit -- with more rows per workgroup the per-k cost of a workgroup
// (fragment reads, expansion, MFMAs per row tile) grows as fast as the stream shrinks -- so the
// headline stays unsplit.  (SNNQP_DENSE_FP6_PLAN="rt,ks" forces a plan: a measurement knob.)
// workspace = F6_TICKET_BYTES of tickets (a fixed head, whatever the plan: a workspace that
// served another shape before still has zeros there), then the slabs
constexpr int64_t F6_TICKET_BYTES = 4096;
struct Fp6Plan { int rt, ks; };
static Fp6Plan pick_fp6_plan(int T, int B, int K, unsigned gy, bool may_split) {
  const int nchunks = ((K + 63) / 64 + F6_KSC - 1) / F6_KSC;
  Fp6Plan best = {0, 1};
  double best_cost = 0;
  for (int ks = 1; ks <= (may_split ? 4 : 1); ks *= 2) {
    if (ks > 1 && (K < 8192 || nchunks < 8 * ks)) break;
    for (int rt = 1; rt <= 5; ++rt) {
      const int sb = rt * 32 / T;
      if (sb < 1) continue;
      const int64_t wgs = (int64_t)((B + sb - 1) / sb) * gy * ks;
      if (ks > 1 && wgs / ks > F6_TICKET_BYTES / 4) continue;      // a ticket per tile
      const double kscale = (double)K / 32768.0;
      const double cost = (double)((wgs + 255) / 256) * (11.0 + kscale * (9.5 + 15.5 * rt) / ks) +
                          (ks > 1 ? 6.0 + 0.09 * (rt * 16 * ks) : 0.0);
      if (best.rt == 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && ks == best.ks)) {
        best = Fp6Plan{rt, ks};
        best_cost = cost;
      }
    }
  }
  return best;
}

static int64_t fp6_workspace_bytes(const Fp6Plan &p, int T, int B, unsigned gy) {
  if (p.ks <= 1) return 0;
  const int sb = p.rt * 32 / T;
  const int64_t tiles = (int64_t)((B + sb - 1) / sb) * gy;
  return F6_TICKET_BYTES + tiles * p.ks * (int64_t)(p.rt * 32 * 128) * 4;
}

int64_t dense_fp6_workspace_bytes(int32_t T, int32_t B, int32_t K, int32_t N) {
  if (T <= 0 || B <= 0 || T > 160) return 0;
  const unsigned gy = (unsigned)((N + 127) / 128);
  return fp6_workspace_bytes(pick_fp6_plan(T, B, K, gy, true), T, B, gy);
}

int run_dense_fp6(const void *x, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B, int32_t K,
                  int32_t N, const snnqp_weight_t *w, const snnqp_bn_t *bn,
                  const snnqp_neuron_t *nrn, const float *u0, float *u_out, uint32_t *s_out,
                  int row_tiles, void *ws, int64_t ws_bytes, hipStream_t st) {
  SNNQP_REQUIRE(w->wt_fp6 && ((x && s_out) || T == 0 || B == 0), SNNQP_EINVAL, "dense fp6: null pointer");
  SNNQP_REQUIRE(T >= 0 && B >= 0, SNNQP_EINVAL, "dense fp6: negative T/B");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  SNNQP_CHECK_BN(bn);
  if (T == 0 || B == 0) return SNNQP_OK;
  SNNQP_REQUIRE(T <= 160, SNNQP_EUNSUPPORTED, "dense fp6: T > 160");
  DenseFp6Args a = {};
  a.x = (const uint32_t *)x; a.xs_t = xs_t; a.xs_b = xs_b;
  a.T = T; a.B = B; a.K = K; a.N = N; a.KW = (K + 31) / 32; a.KS = (K + 63) / 64;
  a.wt6 = (const uint8_t *)w->wt_fp6;
  a.dq = make_dequant(w->L, w->m);
  a.bn = make_bn(bn); a.nrn = make_neuron(nrn);
  a.u0 = u0; a.u_out = u_out; a.s_out = s_out;
  const unsigned gy = (unsigned)((N + 127) / 128);
  // the workspace decides: the plan with a K split when the caller brought enough of it
  // (snnqp_dense_workspace_bytes, zero-filled once), else the best plan without
  Fp6Plan plan = pick_fp6_plan(T, B, K, gy, ws != nullptr);
  if (const char *e = ws ? std::getenv("SNNQP_DENSE_FP6_PLAN") : nullptr) {     // "rt,ks": measurement knob
    int r = 0, k = 0;
    if (std::sscanf(e, "%d,%d", &r, &k) == 2 && r >= 1 && r <= 5 && r * 32 >= T && (k == 1 || k == 2 || k == 4))
      plan = Fp6Plan{r, k};
  }
  if (plan.ks > 1 && (fp6_workspace_bytes(plan, T, B, gy) > ws_bytes || ((uintptr_t)ws & 255) != 0))
    plan = pick_fp6_plan(T, B, K, gy, false);
  if (row_tiles >= 1 && row_tiles <= 5 && row_tiles * 32 >= T) plan = Fp6Plan{row_tiles, 1};
  const int rt = plan.rt;
  SNNQP_REQUIRE(rt > 0, SNNQP_EUNSUPPORTED, "dense fp6: T too large");
  a.SB = rt * 32 / T;
  const unsigned gx = (unsigned)((B + a.SB - 1) / a.SB);
  a.ksplit = plan.ks;
  if (plan.ks > 1) {
    const int nchunks = (a.KS + F6_KSC - 1) / F6_KSC;
    const int ngc_all = (nchunks + F6_KGROUPS - 1) / F6_KGROUPS;
    a.gcz = (ngc_all + plan.ks - 1) / plan.ks;
    const int64_t tiles = (int64_t)gx * gy;
    a.tickets = (uint32_t *)ws;
    a.slabs = (float *)((uint8_t *)ws + F6_TICKET_BYTES);
    a.status = device_status_word(stream_device(st));
    // the tickets are zeroed on the stream in front of EVERY launch (a kernel node when the stream
    // is being captured: zero_words_async, kernels.h): whatever an earlier launch, an aborted replay or a stray store left in
    // them, this launch starts from zero (cdna_hip_programming.md, hand-off recipe: "zero the
    // counter per call; a reset by the last arriver alone fails the first, poisoned launch")
    if (int rc = zero_words_async((uint32_t *)ws, tiles, st)) return rc;
  }
  switch (rt) {
    case 5: launch_dense_fp6<5>(a, gx, gy, st); break;
    case 4: launch_dense_fp6<4>(a, gx, gy, st); break;
    case 3: launch_dense_fp6<3>(a, gx, gy, st); break;
    case 2: launch_dense_fp6<2>(a, gx, gy, st); break;
    default: launch_dense_fp6<1>(a, gx, gy, st); break;
  }
  SNNQP_CHECK_LAUNCH("dense_fp6_kernel");
  return SNNQP_OK;
}

}  // namespace snnqp

extern "C" int snnqp_pack_codes_fp6(const int8_t *w, int64_t K, int32_t N, int32_t Npad,
                                    void *wt6, snnqp_stream_t stream) {
  using namespace snnqp;
  SNNQP_REQUIRE(w && wt6 && K > 0 && N > 0, SNNQP_EINVAL, "pack_codes_fp6: bad argument");
  SNNQP_REQUIRE(Npad >= N && (Npad & 31) == 0, SNNQP_EINVAL,
                "pack_codes_fp6: Npad must be a multiple of 32, Npad >= N");
  const int64_t total = (int64_t)(Npad / 32) * ((K + 63) / 64) * 64;
  const int64_t blocks = ceil_div64(total, 256);
  hipLaunchKernelGGL(pack_codes_fp6_kernel, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, w, K, N, Npad, (uint32_t *)wt6);
  SNNQP_CHECK_LAUNCH("pack_codes_fp6_kernel");
  return SNNQP_OK;
}
