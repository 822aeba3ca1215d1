// SpikingBlock(QuantDense, neuron) for long contractions (the read-out of the conv nets:
// K = 32768, N = 110), as two kernels over a caller-owned workspace:
//
//  1. dense_splitk_gemm_kernel: the connection of every (sample, t) row, flax_qdense.py:87,
//     as ONE int8-MFMA GEMM  C[m][n] = sum_k spike[m][k] * code[k][n],  m = sample * T + t,
//     with K split over blockIdx.z.  The fused kernel of dense_mfma.hip has to keep the T rows
//     of a sample in one workgroup, which caps its row tile at what still gives the chip one
//     workgroup per CU (96 rows at B = 1024): every workgroup then streams the whole 4 MB of
//     codes through its CU (1 GB per launch out of L2) and reads each A fragment from LDS for
//     ONE MFMA -- LDS bandwidth, not the matrix pipe, sets the pace.  Here rows are just rows:
//     a workgroup takes 160 of them x 128 features x 1/S of K, a wave a 160 x 64 tile (every
//     A fragment feeds two MFMAs, every B fragment five), and the int32 partial tile goes to
//     `ws[z]` with plain coalesced stores (no atomics, no flags: the stream orders the two
//     kernels).
//  2. dense_splitk_lif_kernel: one thread per (sample, feature) adds the S partial sums of each
//     timestep (int32: exact in any order), dequantises, applies BatchNorm and the neuron of
//     spiking_learning.py:403-416 over T with u in a register, ballots the spikes into words.
//
// Results are those of dense_mfma.hip bit for bit: the same integers, the same float32
// operation sequence behind them.
#include "kernels.h"

namespace snnqp {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

constexpr int SK_BK = 256;                 // k of one staged chunk (bytes per LDS row)
constexpr int SK_KSC = SK_BK / 32;         // MFMA k-steps per chunk
constexpr int SK_GROUPS = 4;               // wave pairs; pair g takes k-steps 2 g, 2 g + 1 of a chunk
constexpr int SK_KPG = SK_KSC / SK_GROUPS; // k-steps of a chunk per pair
constexpr int SK_NT = 128 * SK_GROUPS;     // threads: 8 waves
constexpr int SK_R = 5;                    // 32-row tiles per workgroup (160 rows)

struct SplitKArgs {
  const uint32_t *x;
  int64_t xs_t, xs_b;            // word strides
  int32_t T, K, N, KS;           // KS = ceil(K / 32) k-steps
  int64_t M;                     // rows = B * T
  int32_t NB;                    // 32-feature blocks = ceil(N / 32)
  int32_t chunks, cps;           // chunks of K in all, per split
  const int8_t *wt;              // MFMA-tiled codes [NB][KS][64][16]
  int32_t *ws;                   // [S][gridDim.y][M][128]
};

__device__ __forceinline__ v4i expand16b(uint32_t b) {
  v4i o;
  o.x = (int)((((b >> 0) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.y = (int)((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.z = (int)((((b >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
  o.w = (int)((((b >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
  return o;
}

// LDS-only workgroup barrier (see conv_tile.h): global loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// byte offset of 16-byte chunk c16 of row `row`: chunks XOR-swizzled by row, so the
// ds_read_b128 of an A fragment (32 rows x one chunk index per half wave) is conflict-free
__device__ __forceinline__ int a_addr(int row, int c16) {
  return row * SK_BK + ((c16 ^ (row & 15)) << 4);
}

template <int R>
__global__ void __launch_bounds__(SK_NT)
dense_splitk_gemm_kernel(SplitKArgs a) {
  constexpr int ROWS = R * 32;
  constexpr int WPR = SK_BK / 32;                  // spike words per row per chunk
  constexpr int NTASK = ROWS * WPR;
  constexpr int TPT = (NTASK + SK_NT - 1) / SK_NT;
  constexpr int ABYTES = ROWS * SK_BK;             // one A buffer
  constexpr int EBYTES = ROWS * 128 * 4;           // the int32 tile, over the A buffers
  constexpr int LDSB = 2 * ABYTES > EBYTES ? 2 * ABYTES : EBYTES;
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDSB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 1, cw = wave & 1;        // k-step pair, 64-feature half
  const int n = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * ROWS;
  const int rows = (int)min((int64_t)ROWS, a.M - m0);
  const int nb0 = blockIdx.y * 4 + cw * 2;         // first of this wave's two 32-feature blocks
  const int c_lo = blockIdx.z * a.cps;
  const int c_hi = min(c_lo + a.cps, a.chunks);

  v16i acc[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      acc[r][cb] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // staging tasks: word wi of row `row`, the same (row, wi) for every chunk
  int64_t roff[TPT];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int task = tid + k * SK_NT;
    const int row = task / WPR;
    roff[k] = -1;
    if (task < NTASK && row < rows) {
      const int64_t m = m0 + row;
      const int64_t b = m / a.T;
      const int t = (int)(m - b * a.T);
      roff[k] = (int64_t)t * a.xs_t + b * a.xs_b + task % WPR;
    }
  }
  uint32_t stg[TPT];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int kw = chunk * WPR + (tid + k * SK_NT) % WPR;
      stg[k] = (roff[k] >= 0 && chunk < c_hi && kw < a.KS) ? a.x[roff[k] + (int64_t)chunk * WPR] : 0u;
    }
  };
  auto stage_store = [&](int buf) {
    uint8_t *base = lds + buf * ABYTES;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int task = tid + k * SK_NT;
      if (task < NTASK) {
        const int row = task / WPR, wi = task % WPR;
        *(v4i *)(base + a_addr(row, wi * 2)) = expand16b(stg[k] & 0xFFFFu);
        *(v4i *)(base + a_addr(row, wi * 2 + 1)) = expand16b(stg[k] >> 16);
      }
    }
  };
  // B fragments of this wave's k-steps of a chunk, one chunk ahead in registers
  const v4i *wt0 = (const v4i *)a.wt + lane;
  v4i bfA[SK_KPG][2], bfB[SK_KPG][2];
  auto load_b = [&](v4i (&bf)[SK_KPG][2], int chunk) {
#pragma unroll
    for (int j = 0; j < SK_KPG; ++j) {
      const int kg = chunk * SK_KSC + grp * SK_KPG + j;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const bool on = chunk < c_hi && kg < a.KS && nb0 + cb < a.NB;
        bf[j][cb] = on ? wt0[((int64_t)(nb0 + cb) * a.KS + kg) * 64] : v4i{0, 0, 0, 0};
      }
    }
  };
  auto compute = [&](const uint8_t *base, const v4i (&bf)[SK_KPG][2]) {
#pragma unroll
    for (int j = 0; j < SK_KPG; ++j) {
      const int ks = grp * SK_KPG + j;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const v4i av = *(const v4i *)(base + a_addr(r * 32 + n, ks * 2 + h));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[r][cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bf[j][cb], acc[r][cb], 0, 0, 0);
      }
    }
  };

  stage_load(c_lo);
  load_b(bfA, c_lo);
  stage_store(0);
  lds_barrier();
  // chunks at or beyond c_hi stage zero words and zero B fragments: the loop runs over pairs
  for (int c = c_lo; c < c_hi; c += 2) {
    stage_load(c + 1);
    load_b(bfB, c + 1);
    compute(lds, bfA);
    stage_store(1);
    lds_barrier();
    stage_load(c + 2);
    load_b(bfA, c + 2);
    compute(lds + ABYTES, bfB);
    stage_store(0);
    lds_barrier();
  }

  // the four pairs' partial tiles -> one int32 tile [row][128] in LDS (C/D layout: column =
  // lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)); pair 3 stores, the others add
  int *et = (int *)lds;
#pragma unroll 1
  for (int g = SK_GROUPS - 1; g >= 0; --g) {
    if (grp == g) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            int *e = et + row * 128 + cw * 64 + cb * 32 + n;
            *e = (g == SK_GROUPS - 1) ? acc[r][cb][i] : *e + acc[r][cb][i];
          }
    }
    lds_barrier();
  }
  // tile -> ws[z][y][m0 + row][0..127], 16 bytes per thread and pass
  int32_t *dst = a.ws + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * a.M + m0) * 128;
  for (int i = tid; i < rows * 32; i += SK_NT)
    ((v4i *)dst)[i] = ((const v4i *)et)[i];
}

struct SplitKLifArgs {
  const int32_t *ws;             // [S][GY][M][128]
  int32_t S, T, B, N;
  int64_t M;
  Dequant dq;
  BnP bn;
  NeuronP nrn;
  const float *u0;
  float *u_out;
  uint32_t *s_out;
};

constexpr int LIF_TB = 10;       // timesteps whose partial sums are loaded together

// grid (ceil(B / 2), GY): 256 threads = two samples x 128 features; a wave holds 64
// consecutive features of one sample, so a ballot is two output words
__global__ void __launch_bounds__(256)
dense_splitk_lif_kernel(SplitKLifArgs a) {
  const int lane = threadIdx.x & 63;
  const int bl = threadIdx.x >> 7, col = threadIdx.x & 127;
  const int b = blockIdx.x * 2 + bl;
  const int feat = blockIdx.y * 128 + col;
  const bool samp = b < a.B;
  const bool live = samp && feat < a.N;
  const int CW = (a.N + 31) >> 5;
  const int word = (blockIdx.y * 128 + (col & 64)) >> 5;       // wave-uniform
  float bmean = 0.f, bmul = 1.f, bbias = 0.f, dec = 0.f, u = 0.0f;
  if (live) {
    if (a.bn.mean) { bmean = a.bn.mean[feat]; bmul = a.bn.mul[feat]; bbias = a.bn.bias[feat]; }
    if (a.nrn.kind == SNNQP_NEURON_LIF) dec = a.nrn.decay[feat];
    if (a.u0) u = a.u0[(int64_t)b * a.N + feat];
  }
  const int64_t zstride = (int64_t)gridDim.y * a.M * 128;
  const int32_t *src = a.ws + ((int64_t)blockIdx.y * a.M + (int64_t)(samp ? b : 0) * a.T) * 128 + col;
  for (int t0 = 0; t0 < a.T; t0 += LIF_TB) {
    int sum[LIF_TB];
#pragma unroll
    for (int i = 0; i < LIF_TB; ++i) {
      sum[i] = 0;
      if (samp && t0 + i < a.T)
        for (int z = 0; z < a.S; ++z) sum[i] += src[z * zstride + (int64_t)(t0 + i) * 128];
    }
#pragma unroll
    for (int i = 0; i < LIF_TB; ++i) {
      const int t = t0 + i;
      if (t < a.T) {                                            // uniform
        bool s = false;
        if (live) {
          float cur = dequant_acc(sum[i], a.dq);
          if (a.bn.mean) cur = bn_apply(cur, bmean, bmul, bbias);
          s = neuron_step(u, cur, a.nrn, dec);
        }
        const unsigned long long m = __ballot(s);
        if (samp) {
          uint32_t *o = a.s_out + ((int64_t)t * a.B + b) * CW;
          if (lane == 0 && word < CW) o[word] = (uint32_t)m;
          if (lane == 32 && word + 1 < CW) o[word + 1] = (uint32_t)(m >> 32);
        }
      }
    }
  }
  if (live && a.u_out) a.u_out[(int64_t)b * a.N + feat] = u;
}

// how the contraction is split for this shape: S = 0 means "use the fused kernel"
struct SplitKPlan {
  int S, chunks, cps;
  unsigned gx, gy;
  int64_t ws_bytes;
};

SplitKPlan plan_splitk(int32_t T, int32_t B, int32_t K, int32_t N) {
  SplitKPlan p = {0, 0, 0, 0, 0, 0};
  const int64_t M = (int64_t)T * B;
  const int KS = (K + 31) / 32;
  p.chunks = (KS + SK_KSC - 1) / SK_KSC;
  p.gx = (unsigned)((M + SK_R * 32 - 1) / (SK_R * 32));
  p.gy = (unsigned)((N + 127) / 128);
  // worth two launches and a trip through the workspace only for a long contraction over
  // enough rows to fill row tiles of 160
  if (p.chunks < 16 || M < 4 * SK_R * 32 || p.gx > 65535u) return p;
  const int64_t tiles = (int64_t)p.gx * p.gy;
  int S = (int)((256 + tiles - 1) / tiles);          // one workgroup per CU at least
  if (S > 8) S = 8;
  if (S > p.chunks / 8) S = p.chunks / 8;            // at least 8 chunks per split
  if (S < 1) S = 1;
  p.cps = (p.chunks + S - 1) / S;
  p.cps += p.cps & 1;                                 // pairs of chunks
  p.S = (p.chunks + p.cps - 1) / p.cps;
  p.ws_bytes = (int64_t)p.S * p.gy * M * 128 * 4;
  return p;
}

}  // namespace

int64_t dense_splitk_workspace_bytes(int32_t T, int32_t B, int32_t K, int32_t N) {
  if (T <= 0 || B <= 0 || K <= 0 || N <= 0) return 0;
  return plan_splitk(T, B, K, N).ws_bytes;
}

int run_dense_splitk(const void *x, int64_t xs_t, int64_t xs_b, int32_t T, int32_t B,
                     int32_t K, int32_t N, const snnqp_weight_t *w, const int8_t *wt,
                     const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                     float *u_out, uint32_t *s_out, void *ws, int64_t ws_bytes,
                     hipStream_t st) {
  SNNQP_REQUIRE(x && s_out && ws, SNNQP_EINVAL, "dense split-K: null pointer");
  SNNQP_REQUIRE(w->L >= 1.0f, SNNQP_EINVAL, "dequant L must be >= 1");
  if (bn) SNNQP_REQUIRE(bn->mean && bn->mul && bn->bias, SNNQP_EINVAL,
                        "batch-norm descriptor with null arrays");
  const SplitKPlan p = plan_splitk(T, B, K, N);
  SNNQP_REQUIRE(p.S >= 1, SNNQP_EUNSUPPORTED, "dense split-K: shape is served by the fused kernel");
  SNNQP_REQUIRE(ws_bytes >= p.ws_bytes, SNNQP_EINVAL,
                "dense split-K: workspace of %lld bytes, %lld needed", (long long)ws_bytes,
                (long long)p.ws_bytes);
  SNNQP_REQUIRE(((uintptr_t)ws & 15) == 0, SNNQP_EINVAL, "dense split-K: workspace not 16-byte aligned");
  SplitKArgs g;
  g.x = (const uint32_t *)x; g.xs_t = xs_t; g.xs_b = xs_b;
  g.T = T; g.K = K; g.N = N; g.KS = (K + 31) / 32;
  g.M = (int64_t)T * B; g.NB = (N + 31) / 32;
  g.chunks = p.chunks; g.cps = p.cps;
  g.wt = wt; g.ws = (int32_t *)ws;
  hipLaunchKernelGGL((dense_splitk_gemm_kernel<SK_R>), dim3(p.gx, p.gy, (unsigned)p.S),
                     dim3(SK_NT), 0, st, g);
  SNNQP_CHECK_LAUNCH("dense_splitk_gemm_kernel");
  SplitKLifArgs e;
  e.ws = (const int32_t *)ws; e.S = p.S; e.T = T; e.B = B; e.N = N; e.M = g.M;
  e.dq = make_dequant(w->L, w->m);
  e.bn = make_bn(bn); e.nrn = make_neuron(nrn);
  e.u0 = u0; e.u_out = u_out; e.s_out = s_out;
  hipLaunchKernelGGL(dense_splitk_lif_kernel, dim3((unsigned)((B + 1) / 2), p.gy), dim3(256), 0, st, e);
  SNNQP_CHECK_LAUNCH("dense_splitk_lif_kernel");
  return SNNQP_OK;
}

}  // namespace snnqp
