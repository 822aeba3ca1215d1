/* snnqp.h -- C ABI of libsnnqp.so: the MI355X (gfx950) implementation of the
 * quantized / pruned SNN time-stepped forward pass of
 * Intelligent-Microsystems-Lab/SNNQuantPrune.
 *
 * The reference has no FFI: its hot path is Python on JAX (XLA emits the device
 * code).  Each entry point below therefore replaces a *Python call site* of the
 * reference; the `replaces:` lines give that call site (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add at each of them.
 *
 * Conventions
 *  - every pointer named x/w/y/u/s/... is a DEVICE pointer owned by the caller
 *    (torch); the library allocates nothing that outlives a call, with one exception:
 *    the fused conv kernels keep 512 KiB of work-queue counters per device (the device of
 *    `stream`): 64 round-robin slots for eager launches -- a slot is zeroed on `stream`
 *    right before its launch and guarded by an event; a launch that finds its slot still in
 *    flight (more than 64 conv launches outstanding on the device) walks its patches
 *    statically instead: same results, no queue -- and 960 slots that launches captured into
 *    a graph take one each until snnqp_workqueue_capture_release hands them back (the pool
 *    must exist before the capture: one eager launch on the device); 64 bytes of page-locked
 *    host memory per device hold the status word (snnqp_device_status);
 *  - descriptor structs (snnqp_*_t) are HOST structs read during the call;
 *  - all work is enqueued on `stream` (a hipStream_t); no call synchronises;
 *  - return value: 0 on success, <0 on error (SNNQP_E*); the message is
 *    available from snnqp_last_error() on the calling thread;
 *  - activations are time-major [T][B]...[C], channels innermost (NHWC), as
 *    inside the reference model (examples/tcja/models.py:109); `x_stride_t` /
 *    `x_stride_b` let a [B][T]... tensor be consumed in place;
 *  - SNNQP_BITS tensors pack channel c of a row into bit (c & 31) of 32-bit
 *    word (c >> 5); a row has ceil(C / 32) words.
 */
#ifndef SNNQP_H_
#define SNNQP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *snnqp_stream_t; /* hipStream_t */

enum {
  SNNQP_OK = 0,
  SNNQP_EINVAL = -1,       /* bad argument / shape */
  SNNQP_EUNSUPPORTED = -2, /* valid request the chosen kernel cannot serve */
  SNNQP_EHIP = -3          /* HIP runtime error */
};

/* element type of an activation tensor.
 * SNNQP_EV1 / SNNQP_EV4 are wire formats of the model INPUT, the 2-channel event frames
 * [H][W][2] of examples/input_pipeline.py:195-218 (what the host feeds, examples/
 * input_pipeline.py:17-27, 655 360 B per DVS128 sample as uint8):
 *   SNNQP_EV1  binary frames, one bit per element in index order: element
 *              i = (y * W + x) * 2 + polarity is bit (i & 31) of 32-bit word (i >> 5); a
 *              frame takes ceil(H * W * 2 / 32) words (frames are word-aligned) =
 *              numpy.packbits(frame.ravel(), bitorder="little") -- 81 920 B per sample;
 *   SNNQP_EV4  count frames with counts <= 15, one byte per pixel: polarity 0 in bits
 *              0..3, polarity 1 in bits 4..7; a frame takes H * W bytes -- 327 680 B.
 * Strides of such tensors are in words (EV1) / bytes (EV4). */
enum { SNNQP_F32 = 0, SNNQP_U8 = 1, SNNQP_BITS = 2, SNNQP_EV1 = 3, SNNQP_EV4 = 4 };
/* element type of a weight tensor */
enum { SNNQP_W_F32 = 0, SNNQP_W_I8 = 1 };
/* neuron models, spiking_learning.py:357-438 */
enum {
  SNNQP_NEURON_NONE = 0,
  SNNQP_NEURON_MULTI_STEP_LIF = 1,      /* spiking_learning.py:390-416 */
  SNNQP_NEURON_PARAMETRIC_LEAKY_IF = 2, /* spiking_learning.py:357-387 */
  SNNQP_NEURON_LIF = 3                  /* spiking_learning.py:419-438 */
};
/* quantisers, quant.py */
enum {
  SNNQP_Q_DUQ = 0,              /* quant.py:428-469  p0 = a, p1 = c           */
  SNNQP_Q_UNIFORM_STATIC = 1,   /* quant.py:322-358  p0 = xmax                */
  SNNQP_Q_PARAMETRIC_D = 2,     /* quant.py:361-425  p0 = step size           */
  SNNQP_Q_PARAMETRIC_D_XMAX = 3 /* quant.py:494-625  p0 = d, p1 = xmax (clipped) */
};
/* kernel selection for the fused blocks */
enum { SNNQP_IMPL_AUTO = 0, SNNQP_IMPL_GENERIC = 1, SNNQP_IMPL_MFMA = 2 };

/* flags written (OR-ed) by the checking kernels into a device int32 */
enum {
  SNNQP_FLAG_CODE_OVERFLOW = 1, /* |code| > 127: does not fit int8          */
  SNNQP_FLAG_MASK_NOT_BINARY = 2,
  SNNQP_FLAG_NOT_INTEGER = 4,   /* activation not an integer in [0, 255]    */
  SNNQP_FLAG_GT_ONE = 8,        /* activation > 1 (not a binary spike)      */
  SNNQP_FLAG_GT_127 = 16,       /* activation > 127 (not an int8 MFMA operand) */
  SNNQP_FLAG_GT_15 = 32         /* activation > 15 (does not fit an EV4 nibble)      */
};

/* Weights after the transforms of flax_qdense.py:74-85 / flax_qconv.py:147-156.
 * SNNQP_W_I8: integer codes (already multiplied by the 0/1 prune mask); the
 * contraction accumulates exactly in int32 and the current is
 * fl(fl(acc / L) * m)  (DuQ: L = n_lv - 1, m = c, quant.py:443,467; the other
 * quantisers: L = 1, m = step).  SNNQP_W_F32: fake-quantised float kernel; the
 * contraction is a float32 fmaf chain over k ascending ((kh, kw, cin) order).
 * Layout: [K][N] for dense, HWIO for convolutions, as the reference's params. */
typedef struct {
  int32_t wtype;
  const void *w;
  float L;
  float m;
  int32_t abs_sum_max; /* W_I8: a bound of |acc| per unit of input: with the non-negative
                        * inputs of the integer kernels (spikes, counts <= x_max),
                        * |acc| <= abs_sum_max * x_max.  Tightest: max over outputs of
                        * max(sum of positive codes, sum of |negative codes|); the
                        * max of sum_k |code| is valid too.  0 = unknown */
  int32_t code_max;    /* W_I8: max |code|; 0 = unknown.  Codes of magnitude <= 7 are
                          exact in fp6 (e2m3) and may take the f8f6f4 MFMA */
  const int32_t *col_sum; /* W_I8, dense layers, nullable: device int32 [N], sum over k of the
                          codes of each output feature.  Lets the MFMA dense kernel read
                          uint8 rows as x - 128 (any count 0..255) and add 128 * col_sum back */
  const void *wt_fp6;  /* W_I8, dense layers, nullable: the codes as fp6 MFMA tiles
                          (snnqp_pack_codes_fp6; needs code_max <= 7).  With it a dense block
                          over bit-packed rows runs on the f8f6f4 MFMA: 6 bits per code from
                          HBM / L2, K = 64 per instruction */
  uint32_t min_current_bits; /* float32 bits of the smallest non-zero |input current| the block
                          can see with the BatchNorm it is used with -- snnqp_current_min()
                          with bound = abs_sum_max (bit-packed inputs);
                          lets the conv kernels run u + (x - u) / tau as one fused
                          multiply-add when that is provably bit-identical.  0 = unknown */
  int32_t ch_stack_max; /* W_I8, the 2-channel event layer, 0 = unknown: with range(c) = the sum of
                          |code| of output channel c, the largest sum of the ranges of the four
                          channels that share an LDS bank column (ch_slots; without it the
                          channels 128 g + 32 w + n, w = 0..3).  Sizes the per-channel
                          dequantisation tables (a channel's table covers its own accumulator
                          range; the four channels of a column are stacked); unknown:
                          8 x abs_sum_max.  A value below the codes' is reported
                          (SNNQP_STATUS_BOUND) */
  const int32_t *ch_slots; /* W_I8, the event layer, nullable: device int32 [ceil(Cout / 128) * 128],
                          for every channel 4 * column + position: which of the 32 bank columns
                          of its 128-channel group holds its table and where in the column's
                          stack of four.  The 32 channels of a wave (32 w .. 32 w + 31) must name
                          32 different columns and no slot twice -- a balanced assignment keeps
                          the tallest column, hence the table, small.  NULL: column n = c mod 32,
                          position w */
} snnqp_weight_t;

/* Eval-mode BatchNorm folded on the host: y = fl(fl(fl(x - mean) * mul) + bias),
 * mul = rsqrt(var + eps) * scale (examples/tcja/models.py:101-107, applied at
 * spiking_learning.py:457-458).  All three device arrays have Cout entries.
 * `flags`: facts the caller knows about the arrays (it folded them on the host):
 * SNNQP_BN_MEAN_ZERO / SNNQP_BN_BIAS_ZERO = every mean / bias entry is +-0.0, so that
 * fl(x - 0) and fl(x + 0) equal x and a kernel may skip the instruction (a freshly
 * initialised BatchNorm has both).  Equal as numbers: for x = -0.0 the skipped form keeps
 * -0.0 where fl(-0.0 + 0.0) is +0.0.  The sign of a zero current never changes a spike or a
 * non-zero potential; it can show in the sign bit of a membrane potential that is exactly
 * zero (u_out), which compares equal.  SNNQP_BN_MUL_UNIFORM = every `mul` entry holds the same
 * bits (a freshly initialised BatchNorm: rsqrt(1 + eps) for every channel): a kernel that reads the
 * dequantised current from a table all channels share may fold the multiply into the entries --
 * fl(x * mul) computed once per entry instead of once per neuron update, the same float32
 * product.  0 = nothing known; other bits are refused. */
#define SNNQP_BN_MEAN_ZERO 1
#define SNNQP_BN_BIAS_ZERO 2
#define SNNQP_BN_MUL_UNIFORM 4
typedef struct {
  const float *mean;
  const float *mul;
  const float *bias;
  int32_t flags;
} snnqp_bn_t;

/* Neuron parameters.  `k`: tau for MULTI_STEP_LIF, sigmoid(tau_param) for
 * PARAMETRIC_LEAKY_IF; `decay`: device [Cout] sigmoid(tau) for LIF. */
typedef struct {
  int32_t kind;
  float k;
  float v_threshold;
  float v_reset;
  const float *decay;
} snnqp_neuron_t;

/* Convolution geometry (flax_qconv.py:76-94, :158-168); 1-D convolutions use
 * H = KH = 1.  A dense layer is the 1x1 convolution on a 1x1 image. */
typedef struct {
  int32_t H, W, Cin, Cout;
  int32_t KH, KW;
  int32_t stride_h, stride_w;
  int32_t pad_h_lo, pad_h_hi, pad_w_lo, pad_w_hi;
  int32_t in_dil_h, in_dil_w;   /* input_dilation  (lhs) */
  int32_t k_dil_h, k_dil_w;     /* kernel_dilation (rhs) */
  int32_t groups;               /* feature_group_count   */
} snnqp_conv_geom_t;

/* Geometry of a 3-D convolution (flax_qconv.py:93-171 with three spatial axes; index 0 = depth,
 * 1 = height, 2 = width): images [D][H][W][Cin], kernels [KD][KH][KW][Cin / groups][Cout]. */
typedef struct {
  int32_t D, H, W, Cin, Cout;
  int32_t KD, KH, KW;
  int32_t stride[3];
  int32_t pad_lo[3], pad_hi[3];
  int32_t in_dil[3];            /* input_dilation  (lhs) */
  int32_t k_dil[3];             /* kernel_dilation (rhs) */
  int32_t groups;               /* feature_group_count   */
} snnqp_conv3d_geom_t;

/* ABI version of this header: bumped whenever a struct or a signature below changes
 * (100: round 1; 200: snnqp_bn_t.flags, x_max / x_seen of snnqp_conv_lif_forward;
 * 300: SNNQP_EV1 / SNNQP_EV4 frame types, snnqp_pack_frames / snnqp_unpack_frames,
 * snnqp_fallback_counts; 301: snnqp_conv_dequant_form; 400: snnqp_dense_head_forward, snnqp_device_status,
 * snnqp_workqueue_*, snnqp_dense_lif_forward_ws; 500: float32 inputs into integer blocks --
 * x_flags of snnqp_conv_lif_forward / snnqp_dense_lif_forward_ws / snnqp_dense_head_forward, the
 * predicated snnqp_*_if entry points, snnqp_pack_bits_checked, snnqp_conv_gated_forward,
 * snnqp_dense_gated_forward, snnqp_quantize_ex, snnqp_conv_forward_if,
 * snnqp_conv3d_*; 501: snnqp_weight_t.ch_stack_max / ch_slots, the *_gated_*_ex pack calls; 502:
 * snnqp_pack_frames_checked, snnqp_conv_lif_forward_pred, SNNQP_BN_MUL_UNIFORM).  A binding compares snnqp_version()
 * with the SNNQP_VERSION it was written against and refuses a library of another version (_lib.py does). */
#define SNNQP_VERSION 502
int snnqp_version(void);
const char *snnqp_last_error(void);
/* Extra compiler flags the library was built with: "" for the product build
 * (snnquantprune_amd/csrc/build.py); diagnostic builds (tools/diag_build.py, written
 * under diag_build/, never in-tree) report their -D switches here.  Tests and bench.py
 * refuse a library whose string is not empty. */
const char *snnqp_build_flags(void);

/* Output spatial size of `g` (same rule as lax.conv_general_dilated). */
int snnqp_conv_out_shape(const snnqp_conv_geom_t *g, int32_t *OH, int32_t *OW);

/* ---- weight transforms --------------------------------------------------
 * replaces: cfg.weight(bits, g_scale)(kernel) + prune()(kernel_fwd) at
 *           flax_qdense.py:74-85 and flax_qconv.py:147-156
 *           (quantiser forwards quant.py:322-625, prune quant.py:472-491).
 * w, mask (nullable), fq_out (nullable, float32 fake-quantised*mask),
 * codes_out (nullable, int8 codes*mask), flags (nullable int32, OR-ed).
 * Rounding is round-half-to-even (jnp.round, quant.py:88-90). */
int snnqp_quantize(int kind, const float *w, const float *mask, int64_t n,
                   int bits, float p0, float p1, float *fq_out,
                   int8_t *codes_out, int32_t *flags, snnqp_stream_t stream);
/* The same with the reference's `sign` argument (Quantizer.__call__(x, sign), quant.py:331,
 * :374, :439, :512): sign = 0 quantises to the unsigned levels 0 .. 2^bits - 1 (num_levels /
 * q_pos = 2^bits - 1, lower clip bound 0; DuQ keeps hard_tanh and only changes n_lv,
 * quant.py:458-467).  snnqp_quantize is sign = 1, what QuantDense / QuantConv pass
 * (flax_qdense.py:76).  int8 codes exist for unsigned widths up to 7 bits only
 * (SNNQP_FLAG_CODE_OVERFLOW beyond). */
int snnqp_quantize_ex(int kind, const float *w, const float *mask, int64_t n,
                      int bits, int sign, float p0, float p1, float *fq_out,
                      int8_t *codes_out, int32_t *flags, snnqp_stream_t stream);

/* Layout step of the pack: int8 codes [K][N] (dense kernel / flattened HWIO
 * convolution kernel, as the reference stores them) -> MFMA B-operand tiles
 * wt[Npad/32][K/32][64][16]: for column block nb and k-step ks, lane
 * l = (n & 31) + 32 h holds bytes k = 32 ks + 16 h + j (j < 16) of column
 * n = 32 nb + (l & 31); columns >= N are zero.  One contiguous 1 KiB read per
 * wave and k-step.  K and Npad must be multiples of 32. */
int snnqp_pack_codes_mfma(const int8_t *w, int64_t K, int32_t N, int32_t Npad,
                          int8_t *wt, snnqp_stream_t stream);

/* The "packed int load" of a dense kernel whose codes have magnitude <= 7 (DuQ up to 4 bits):
 * int8 codes [K][N] -> fp6 (e2m3) B-operand tiles of v_mfma_scale_f32_32x32x64_f8f6f4,
 * wt6[Npad/32][ceil(K/64)][1536 B]: for column block nb and k-step ks, lane l = (n & 31) + 32 h
 * holds k = 64 ks + 32 h + j (j < 32) of column n = 32 nb + (l & 31), value j at bits
 * [6 j, 6 j + 6) of six dwords; a tile stores dwords 0..3 of its 64 lanes (1 KiB), then dwords
 * 4..5 (512 B).  Rows beyond K and columns beyond N are zero.  0.75 bytes per code. */
int snnqp_pack_codes_fp6(const int8_t *w, int64_t K, int32_t N, int32_t Npad, void *wt6,
                         snnqp_stream_t stream);

/* ---- activation format helpers ------------------------------------------
 * replaces: nothing in the reference (its activations are float32 arrays);
 * they convert between the reference's float32 layout and the packed formats
 * the kernels read/write.  rows x C logical elements. */
int snnqp_inspect_f32(const float *x, int64_t n, int32_t *flags,
                      snnqp_stream_t stream);
/* inspect_u8: *flags (zeroed by the caller) = (max(x) << 8) | SNNQP_FLAG_GT_* bits */
int snnqp_inspect_u8(const uint8_t *x, int64_t n, int32_t *flags,
                     snnqp_stream_t stream);
int snnqp_f32_to_u8(const float *x, uint8_t *y, int64_t n,
                    snnqp_stream_t stream);
/* narrow_f32: inspect_f32 + f32_to_u8 + inspect_u8 in one pass over x.  flags[0]
 * (zeroed by the caller) = (max << 8) | SNNQP_FLAG_GT_*, flags[1] = SNNQP_FLAG_NOT_INTEGER
 * or 0; y is the uint8 copy, meaningful when flags[1] == 0. */
int snnqp_narrow_f32(const float *x, uint8_t *y, int64_t n, int32_t *flags,
                     snnqp_stream_t stream);
/* Event frames between uint8 [frames][H][W][2] and the wire formats SNNQP_EV1 /
 * SNNQP_EV4 (device side of the host feed; the host packs with the same layout).
 * pack: a value the format cannot hold is saturated (EV1: > 1 -> 1, EV4: > 15 -> 15) and
 * flagged (SNNQP_FLAG_GT_ONE / SNNQP_FLAG_GT_15 OR-ed into the nullable device word
 * `flags`).  unpack: exact inverse on every tensor that packed without a flag. */
int snnqp_pack_frames(const uint8_t *x, int64_t frames, int32_t H, int32_t W, int fmt,
                      void *y, int32_t *flags, snnqp_stream_t stream);
int snnqp_unpack_frames(const void *x, int fmt, int64_t frames, int32_t H, int32_t W,
                        uint8_t *y, snnqp_stream_t stream);
/* uint8 (in_type SNNQP_U8) or float32 (SNNQP_F32: the reference's own input dtype,
 * flax_qconv.py:101) frames [frames][H][W][2] -> SNNQP_EV1 in ONE pass at the rate of HBM, with
 * the check the event layer would make while staging them: `flags` (required; zeroed by this call
 * on `stream`) receives SNNQP_FLAG_GT_ONE when a value is neither 0 (-0.0 counts) nor 1, and
 * SNNQP_FLAG_NOT_INTEGER when a float32 value is not an integer in [0, 255].  A flagged result is
 * not the tensor: it serves as the speculative half of
 *     snnqp_pack_frames_checked(x, ..., ev1, flags);
 *     snnqp_conv_lif_forward(ev1, SNNQP_EV1, ...);                 binary frames: 1/8 .. 1/32 of the bytes,
 *     snnqp_conv_lif_forward_pred(flags, x, in_type, ...);         the event layer's fastest variant
 * where the last call redoes the block on the frames as they are iff the word is set. */
int snnqp_pack_frames_checked(const void *x, int in_type, int64_t frames, int32_t H, int32_t W,
                              uint32_t *y, int32_t *flags, snnqp_stream_t stream);
int snnqp_pack_bits(const void *x, int in_type, int64_t rows, int32_t C,
                    uint32_t *bits, snnqp_stream_t stream);
/* the same, and SNNQP_FLAG_GT_ONE is OR-ed into the device word `flags` (nullable; zeroed by the
 * caller) when an element is neither 0 nor 1 -- the raster then is not the tensor */
int snnqp_pack_bits_checked(const void *x, int in_type, int64_t rows, int32_t C,
                            uint32_t *bits, int32_t *flags, snnqp_stream_t stream);
int snnqp_unpack_bits(const uint32_t *bits, int64_t rows, int32_t C, float *y,
                      snnqp_stream_t stream);

/* ---- connection only (no neuron) ------------------------------------------
 * replaces: lax.dot_general at flax_qdense.py:87-89 and
 *           lax.conv_general_dilated at flax_qconv.py:158-168.
 * x: NB images [NB][H][W][Cin] (type in_type); y: float32 [NB][OH][OW][Cout];
 * acc (nullable): int32 accumulators, same shape (W_I8 with integer input). */
int snnqp_conv_forward(const void *x, int in_type, int64_t NB,
                       const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                       float *y, int32_t *acc, snnqp_stream_t stream);
/* The same, executed only if *pred != 0 when the stream reaches it: the float32 connection behind
 * a speculative integer one (a float32 tensor narrowed by snnqp_narrow_f32, whose flag word is
 * `pred`) -- QuantDense / QuantConv called on their own with float32 inputs
 * (flax_qdense.py:67, flax_qconv.py:101).  Direct form, same fmaf chain. */
int snnqp_conv_forward_if(const int32_t *pred, const void *x, int in_type, int64_t NB,
                          const snnqp_conv_geom_t *g, const snnqp_weight_t *w, float *y,
                          snnqp_stream_t stream);

/* ---- 3-D QuantConv, alone or as a SpikingBlock --------------------------------------------
 * replaces: lax.conv_general_dilated at flax_qconv.py:158-168 for kernels with three spatial axes
 *           (and SpikingBlock.__call__, spiking_learning.py:446-462, around it).
 * x [T][B][D][H][W][Cin] (strides in elements, words for SNNQP_BITS; types F32 / U8 / BITS);
 * nrn null or kind SNNQP_NEURON_NONE: the connection alone (pass T = 1, B = the number of images),
 * s_out = float32 currents [B][OD][OH][OW][Cout] (s_type SNNQP_F32); else s_out = spikes
 * [T][B][OD][OH][OW][Cout] (F32 or BITS), u0 / u_out [B][OD][OH][OW][Cout].  Contracts as everywhere:
 * int8 codes x integer input = exact int32 sum, then fl(fl(acc / L) * m); float32 weights = fmaf
 * chain over (kd, kh, kw, cin) ascending.  Direct form (one thread per output neuron): no shipped
 * model has a 3-D layer.  pred (nullable): skip the launch unless *pred != 0, as the *_if calls. */
int snnqp_conv3d_out_shape(const snnqp_conv3d_geom_t *g, int32_t *OD, int32_t *OH, int32_t *OW);
int snnqp_conv3d_lif_forward(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                             int64_t x_stride_b, int32_t T, int32_t B,
                             const snnqp_conv3d_geom_t *g, const snnqp_weight_t *w,
                             const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                             float *u_out, void *s_out, int s_type, snnqp_stream_t stream);

/* ---- connection on gate x raster (no neuron) ------------------------------------
 * replaces: QuantConv (flax_qconv.py:93-171; 3x3, stride 1, pad 1) on the product of a spike
 *           raster and a per-(image, channel) gate -- the conv block behind a TCJA gate,
 *           x = s * sigmoid(...)[:, :, None, None, :], examples/tcja/models.py:95-97 -> :149-187.
 * The 'gint' contraction: the gate of a channel multiplies all
 * nine taps of that channel, so it is factored out of them --
 *     I[p][c][o] = sum over the taps of code * s          (exact integer, on the matrix pipe)
 *     acc[p][o]  = fmaf(gate[c], I[p][c][o], acc[p][o])   for c = 0 .. Cin - 1, from +0
 *     y[p][o]    = fl(fl(acc / L) * m)
 * s     [NB][H][W][ceil(Cin / 32)] spike words (SNNQP_BITS), gate [NB][Cin] float32,
 * y     float32 [NB][H][W][Cout].  w: SNNQP_W_I8 codes with code_max <= 7 (DuQ up to 4 bits; up to
 * 127 with the _ex pack below), HWIO; packed: the same codes as snnqp_pack_codes_gated lays them out
 * (snnqp_conv_gated_packed_bytes bytes).  Cin in {32, 64, 96, 128}; any H, W, Cout.
 * SNNQP_EUNSUPPORTED otherwise: the caller multiplies (snnqp_apply_gate) and takes the float32
 * connection (snnqp_conv_forward). */
int64_t snnqp_conv_gated_packed_bytes(int32_t Cin, int32_t Cout);
int snnqp_pack_codes_gated(const int8_t *w, int32_t Cin, int32_t Cout, void *packed,
                           snnqp_stream_t stream);
/* Codes beyond e2m3 (code_max = max |code| up to 127: DuQ up to 8 bits, the reference's shipped TCJA
 * configs): packed as two six-bit digits in the e3m2 format, code = 16 hi + lo, in the two K blocks
 * of the matrix instruction with a block scale of 2^4 on the second -- one instruction still
 * returns the exact integer sum of a channel's taps, at the same rate.  The _ex forms take code_max (<= 7: the fp6 layout above);
 * snnqp_conv_gated_forward picks the kernel by w->code_max, `packed` must come from the pack call
 * with the same code_max. */
int64_t snnqp_conv_gated_packed_bytes_ex(int32_t Cin, int32_t Cout, int32_t code_max);
int snnqp_pack_codes_gated_ex(const int8_t *w, int32_t Cin, int32_t Cout, int32_t code_max,
                              void *packed, snnqp_stream_t stream);
int snnqp_conv_gated_forward(const uint32_t *s, const float *gate, int64_t NB,
                             const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                             const void *packed, float *y, snnqp_stream_t stream);

/* The same contraction for a QuantDense (flax_qdense.py:87-89) on the channel-major flattening of
 * gate x raster -- the first dense block behind the second TCJA gate, x[(c HW + p)] = gate[c] *
 * s[p][c], examples/tcja/models.py:97 -> :189-190 -> :200-216:
 *     I[c][o] = sum over the HW positions of code[c HW + p][o] * s[p][c]     (exact integer)
 *     acc[o]  = fmaf(gate[c], I[c][o], acc[o])   for c = 0 .. C - 1, from +0
 *     y[o]    = fl(fl(acc / L) * m)
 * s [NB][HW][ceil(C / 32)] spike words, gate [NB][C] float32, y float32 [NB][N]; w: SNNQP_W_I8
 * codes [C HW][N] with code_max <= 7; packed: snnqp_pack_codes_dense_gated's layout
 * (snnqp_dense_gated_packed_bytes bytes).  HW <= 16, C in {32, 64, 96, 128}; SNNQP_EUNSUPPORTED
 * otherwise (the caller multiplies the gate out and takes snnqp_conv_forward). */
int64_t snnqp_dense_gated_packed_bytes(int32_t C, int32_t N);
int snnqp_pack_codes_dense_gated(const int8_t *w, int32_t C, int32_t HW, int32_t N, void *packed,
                                 snnqp_stream_t stream);
/* (codes up to 127 as two e3m2 digits, as snnqp_pack_codes_gated_ex: the _ex forms take code_max) */
int64_t snnqp_dense_gated_packed_bytes_ex(int32_t C, int32_t N, int32_t code_max);
int snnqp_pack_codes_dense_gated_ex(const int8_t *w, int32_t C, int32_t HW, int32_t N, int32_t code_max,
                                    void *packed, snnqp_stream_t stream);
int snnqp_dense_gated_forward(const uint32_t *s, const float *gate, int64_t NB, int32_t HW,
                              int32_t C, int32_t N, const snnqp_weight_t *w,
                              const void *packed, float *y, snnqp_stream_t stream);

/* ---- fused SpikingBlock ---------------------------------------------------
 * replaces: SpikingBlock.__call__ (nn.scan over T of connection -> norm ->
 *           neuron), spiking_learning.py:446-462, with QuantConv / QuantDense
 *           as connection_fn, nn.BatchNorm (eval) as norm_fn, and -- when
 *           pool == 2 -- the 2x2 max-pool that follows it in
 *           examples/tcja/models.py:145-147.
 * x     [T][B][H][W][Cin] with element strides x_stride_t / x_stride_b
 *       (words for SNNQP_BITS); rows [H][W][Cin] are contiguous.
 * bn    nullable.  u0 nullable (zeros, initialize_carry :464-472),
 *       float32 [B][OH][OW][Cout].  u_out nullable, same shape.
 * s_out spikes, type s_type (SNNQP_F32: 0.0/1.0, or SNNQP_BITS),
 *       [T][B][OH/pool][OW/pool][Cout].
 * impl  SNNQP_IMPL_GENERIC: direct form, any geometry / types.
 *       SNNQP_IMPL_MFMA: MFMA implicit GEMM (fp6 codes x fp4 spikes when
 *       code_max <= 7, else int8); needs W_I8, 3x3 / stride 1 / pad 1 / no
 *       dilation / groups 1 and (BITS input with Cin <= 128 and `wt` = the codes
 *       of the kernel zero-padded along Cin to Cpad = 64 (Cin <= 64) or 128,
 *       tiled by snnqp_pack_codes_mfma with K = 9 * Cpad (row = tap * Cpad + cin);
 *       a pixel keeps its ceil(Cin / 32) spike words, zero bits beyond Cin --
 *       or U8 input with Cin == 2, any count 0..255; or EV1 / EV4 input, Cin == 2: the packed
 *       frames are staged directly, 1/8 (binary) or 1/2 (counts <= 15) of the uint8 bytes;
 *       x_max / x_seen apply to EV4 as to U8), s_type BITS;
 *       any H, W, Cout and neuron kind.  SNNQP_IMPL_AUTO picks MFMA when it can.
 * x_max the largest input value the caller EXPECTS (1 for spikes and binary event frames;
 *       0 = unknown, taken as 1): with the weights' abs_sum_max it sizes the LDS tables the
 *       MFMA kernels dequantise through.  It is a hint, not a promise: the U8 kernel takes
 *       the maximum of every chunk of input it stages and runs the general path (any count up
 *       to 255, arithmetic dequantisation) for a chunk that exceeds it -- same results, slower.
 * x_seen nullable, EIGHT device words the caller zeroes: [0] is atomically max-ed with the
 *       largest U8 input value the launch met, [1..5] receive the number of staged chunks (a
 *       patch x up to 32 timesteps) whose largest value was <= 1, 2, <= 7, <= 31, above; [6] the
 *       number of those whose largest value was exactly 3 (part of [3]: the per-channel tables
 *       reach a hint of 3 on typical pruned layers, not 7); [7] is reserved.  From these the caller refines its next hint asynchronously -- a hint
 *       that covers MOST chunks is the fast one: a hot pixel then costs its own few chunks the
 *       general path instead of the whole batch the slower tables.  Nothing ever waits for it.
 * FLOAT32 INPUT INTO INTEGER CODES (the reference casts every input to float32, flax_qconv.py:101,
 *       flax_qdense.py:67; its event frames and spike rasters are integer-valued float32 tensors).
 *       With SNNQP_W_I8 weights, in_type SNNQP_F32 and Cin == 2 the event-layer MFMA kernel stages
 *       the float32 frames IN PLACE (strides in elements): every value is converted to its uint8
 *       count on the way into LDS and checked while it waits in a register.  x_flags: one device
 *       word, zeroed by this call on `stream` in front of the kernel; the kernel ORs
 *       SNNQP_FLAG_NOT_INTEGER into it when a staged value is not an integer in [0, 255] (-0.0
 *       counts as 0) -- the launch's results are then meaningless and the caller's
 *       snnqp_conv_lif_forward_if(x_flags, ...) with the float32 kernel, enqueued right behind,
 *       redoes the block.  Nothing is read back: the pair is enqueued unconditionally and can be
 *       captured into a graph.  x_flags is required for this combination, ignored otherwise.
 *       Other geometries refuse float32 input into integer codes (SNNQP_EUNSUPPORTED): narrow
 *       it with snnqp_narrow_f32 / snnqp_pack_bits_checked (their flag words serve as the
 *       predicate in the same way). */
int snnqp_conv_lif_forward(const void *x, int in_type, int64_t x_stride_t,
                           int64_t x_stride_b, int32_t T, int32_t B,
                           const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                           const int8_t *wt, const snnqp_bn_t *bn,
                           const snnqp_neuron_t *nrn, const float *u0,
                           float *u_out, void *s_out, int s_type, int pool,
                           int impl, int x_max, int32_t *x_seen, int32_t *x_flags,
                           snnqp_stream_t stream);

/* snnqp_conv_lif_forward on the event layer's MFMA kernel (3x3, stride 1, pad 1, Cin = 2; in_type
 * SNNQP_U8, SNNQP_EV4 or SNNQP_F32), enqueued unconditionally but EXECUTED only if *pred != 0 when the
 * stream reaches it: the frames in their own format behind a speculative launch on their
 * bit-packed copy (snnqp_pack_frames_checked).  Same arguments and results as
 * snnqp_conv_lif_forward with SNNQP_IMPL_MFMA (x_flags is zeroed on the stream whether or not the
 * kernel runs; x_seen is reported into only when it runs); SNNQP_EUNSUPPORTED for any other block.
 * When it does not run it costs one grid of workgroups that return at once.
 * replaces: the same call site as snnqp_conv_lif_forward. */
int snnqp_conv_lif_forward_pred(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                                int64_t x_stride_b, int32_t T, int32_t B,
                                const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                                const int8_t *wt, const snnqp_bn_t *bn,
                                const snnqp_neuron_t *nrn, const float *u0,
                                float *u_out, void *s_out, int s_type, int pool,
                                int x_max, int32_t *x_seen, int32_t *x_flags,
                                snnqp_stream_t stream);

/* The same block on the direct-form kernel (any geometry, any types; SNNQP_W_F32 weights: the
 * fmaf chain over (kh, kw, cin) ascending), enqueued unconditionally but EXECUTED only if
 * *pred != 0 when the stream reaches it (pred: a device word, e.g. the x_flags of the speculative
 * integer launch in front of it, or flags[1] of snnqp_narrow_f32).  It writes the same outputs
 * (pool == 2: the 2x2 max-pool fused, spikes [T][B][OH/2][OW/2][Cout]); when it does not run it
 * costs one small grid.  Together: a float32 tensor whose values are all integers in [0, 255]
 * gets the exact-integer contraction, any other tensor the reference's float32 one -- decided per
 * tensor on the device, with no read-back.
 * replaces: the same call site as snnqp_conv_lif_forward. */
int snnqp_conv_lif_forward_if(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                              int64_t x_stride_b, int32_t T, int32_t B,
                              const snnqp_conv_geom_t *g, const snnqp_weight_t *w,
                              const snnqp_bn_t *bn, const snnqp_neuron_t *nrn, const float *u0,
                              float *u_out, void *s_out, int s_type, int pool,
                              snnqp_stream_t stream);

/* Blocks that SNNQP_IMPL_AUTO handed to the direct-form kernel since the library was loaded
 * (or the last reset): it serves every geometry and type, 20-25 x slower than the MFMA kernels.
 * `reason` (nullable, reason_len bytes) receives why the last one fell back, e.g.
 * "conv: bit input needs Cin <= 128".  With SNNQP_LOG_FALLBACKS set in the environment every
 * new reason is also printed to stderr.  No reference counterpart (diagnostic). */
int snnqp_fallback_counts(int64_t *conv_blocks, int64_t *dense_blocks, char *reason,
                          int32_t reason_len, int reset);

/* ---- device status, work-queue bookkeeping (no reference counterpart: diagnostics) ----------
 * The conv kernels walk their patches through per-launch work queues and the split-K dense
 * kernel hands partial sums over through tickets; both rely on counters that are zero when a
 * launch begins.  A launch that ends with its bookkeeping not adding up (its results may be
 * wrong) stores a code in the device's status word -- page-locked host memory, read without any
 * synchronisation -- and every later snnqp_conv_lif_forward / snnqp_dense_lif_forward /
 * snnqp_dense_head_forward on that device returns SNNQP_EHIP until the word is reset. */
#define SNNQP_STATUS_QUEUE_CORRUPT 1u /* a conv launch's patches did not add up */
#define SNNQP_STATUS_TICKET 2u        /* a split-K dense launch drew a ticket out of range */
#define SNNQP_STATUS_BOUND 4u         /* snnqp_weight_t.abs_sum_max below a one-sided code sum of its
                                       * weights (checked once per weights on the device, after the
                                       * first launch that used them: the conv kernels size their
                                       * dequantisation tables by it) */
int snnqp_device_status(int device, uint32_t *codes, int reset);
/* Launches captured into a graph take a work-queue slot of their own (see the top of this file).
 * mark: the number of capture slots handed out on `device` so far.  release: the slots handed out
 * between two marks go back to the pool -- call it when the graph that was captured between the
 * marks has been destroyed and its last replay has completed. */
int snnqp_workqueue_capture_mark(int device, int64_t *mark);
int snnqp_workqueue_capture_release(int device, int64_t mark_begin, int64_t mark_end);
/* Since the library was loaded (or the last reset): conv launches that walked their patches
 * statically because no capture slot was left / because their eager slot was still in flight
 * (same results, worse balance), and bit-input conv launches that ran the arithmetic
 * dequantisation because the device failed (or, under capture, had not yet run) the probe of the
 * matrix pipe's float32-denormal arithmetic that SNNQP_DQ_TABLE rests on (same results). */
int snnqp_workqueue_stats(int64_t *captured_static, int64_t *busy_static,
                          int64_t *dequant_fallbacks, int reset);
/* test hook: writes `value` into word `word` of the capture slot that was handed out at `mark`
 * (synchronous) */
int snnqp_debug_workqueue_poke(int device, int64_t mark, int word, uint32_t value);

/* How the bit-input 3x3 MFMA kernels of snnqp_conv_lif_forward turn the integer accumulator
 * into the current fl(fl(acc / L) * m) for these weights and this neuron (host-side query, no
 * device work; < 0: error).  SNNQP_DQ_ARITH: three float32 instructions per value;
 * SNNQP_DQ_ONE: L == 1, one multiply; SNNQP_DQ_TABLE: codes within +-7, multi-step LIF / PLIF
 * with v_reset = 0 and abs_sum_max <= 2047 -- the accumulator's bit pattern is the address of
 * the current in an LDS table, no vector instruction (csrc/conv3x3_bits.hip).  All three give
 * the same bits.  No reference counterpart (diagnostic). */
#define SNNQP_DQ_ARITH 1
#define SNNQP_DQ_ONE 2
#define SNNQP_DQ_TABLE 3
int snnqp_conv_dequant_form(const snnqp_weight_t *w, const snnqp_neuron_t *nrn);

/* Smallest non-zero |BatchNorm_c(fl(fl(acc / L) * m))| over |acc| <= bound and the Cout
 * channels (bn nullable: identity), as float32 bits, atomically min-ed into *out_bits
 * (device word the caller initialises to 0x7F800000).  For snnqp_weight_t.min_current_bits:
 * once per (weights, BatchNorm) version. */
int snnqp_current_min(const snnqp_weight_t *w, const snnqp_bn_t *bn, int32_t bound,
                      int32_t Cout, uint32_t *out_bits, snnqp_stream_t stream);

/* Same for QuantDense: x [T][B][K], weights [K][N] (GENERIC) .
 * IMPL_MFMA additionally needs `wt`: the int8 codes tiled by
 * snnqp_pack_codes_mfma (Npad = N rounded up to 32; K rows zero-padded to a multiple
 * of 32 when K is not one), s_type BITS and either BITS input (zero bits beyond K),
 * T <= 96, or U8 input (any count 0..255, read in place: no packing pass, no inspection)
 * with w->col_sum, K % 16 == 0, K <= 65536, 16-byte aligned rows and T <= 64.  With
 * w->wt_fp6 and code_max <= 7, BITS rows run on the fp4 x fp6 MFMA instead (T <= 160, `wt` not
 * needed).  Longer runs and everything else: the direct-form kernel.  (More than 128 features
 * with T <= 64: a workgroup per 256 / 512-column block that reads every row once,
 * csrc/dense_wide.hip; else 128 columns per workgroup, csrc/dense_mfma.hip.) */
int snnqp_dense_lif_forward(const void *x, int in_type, int64_t x_stride_t,
                            int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                            int32_t N, const snnqp_weight_t *w,
                            const int8_t *wt, const snnqp_bn_t *bn,
                            const snnqp_neuron_t *nrn, const float *u0,
                            float *u_out, void *s_out, int s_type, int impl,
                            snnqp_stream_t stream);

/* The same with a workspace: the fp4 x fp6 kernel then may split K over several workgroups per
 * tile of rows (the read-out of config C3: 32768 -> 110 gives 256 workgroups of 80 rows too few
 * rows to amortise the 3 MB of codes each of them streams; two workgroups of 160 rows per tile
 * stream half each), handing partial sums over through `ws`.  ws: device memory, 256-byte
 * aligned, at least snnqp_dense_workspace_bytes(...) bytes, used by one launch at a time (its
 * content need not survive between launches: the call zeroes the tickets at its head on `stream`
 * in front of the kernel -- a kernel node when the stream is being captured -- so that nothing an
 * earlier launch, an aborted replay or a stray store left there can reach this one).
 * ws = NULL or too small: no split, as snnqp_dense_lif_forward.
 * snnqp_dense_workspace_bytes returns 0 when the split would not be used.
 * x_flags: float32 rows into integer codes (see snnqp_conv_lif_forward): the wide kernel (more
 * than 128 features, T <= 64, K % 16 == 0, rows 16-byte aligned, w->col_sum) stages them in place
 * and reports into the word; required then, ignored otherwise; other shapes refuse the
 * combination (SNNQP_EUNSUPPORTED: narrow the rows with snnqp_narrow_f32 first). */
int64_t snnqp_dense_workspace_bytes(int in_type, int32_t T, int32_t B, int32_t K, int32_t N,
                                    const snnqp_weight_t *w);
int snnqp_dense_lif_forward_ws(const void *x, int in_type, int64_t x_stride_t,
                               int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                               int32_t N, const snnqp_weight_t *w,
                               const int8_t *wt, const snnqp_bn_t *bn,
                               const snnqp_neuron_t *nrn, const float *u0,
                               float *u_out, void *s_out, int s_type, int impl,
                               int32_t *x_flags, void *ws, int64_t ws_bytes, snnqp_stream_t stream);
/* the dense block on the direct-form kernel, executed only if *pred != 0 (snnqp_conv_lif_forward_if) */
int snnqp_dense_lif_forward_if(const int32_t *pred, const void *x, int in_type, int64_t x_stride_t,
                               int64_t x_stride_b, int32_t T, int32_t B, int32_t K, int32_t N,
                               const snnqp_weight_t *w, const snnqp_bn_t *bn,
                               const snnqp_neuron_t *nrn, const float *u0, float *u_out,
                               void *s_out, int s_type, snnqp_stream_t stream);

/* ---- the dense head as one launch -----------------------------------------------
 * replaces: the two dense SpikingBlocks and the vote that end CextNet,
 *           examples/tcja/models.py:200-255 --
 *             SpikingBlock(QuantDense(N1), neuron) -> SpikingBlock(QuantDense(N2), neuron)
 *             -> mean over T -> mean over each class's `group` neurons
 *           (per block spiking_learning.py:446-462 with flax_qdense.py:74-89; no norm_fn, no
 *           bias: models.py:200-208, 231-236), i.e. config C2 of BASELINE.json as a whole.
 * x     [T][B][K] by strides, SNNQP_U8 (any count 0..255, read in place; needs w1->col_sum,
 *       K % 16 == 0, 16-byte aligned rows), SNNQP_BITS (zero bits beyond K), or SNNQP_F32 -- the
 *       rows as the reference holds them (flax_qdense.py:67), read in place under the same
 *       conditions as uint8 rows: 4 x the bytes, checked on the way into LDS; x_flags as for
 *       snnqp_conv_lif_forward (required for SNNQP_F32; when the word comes back set, the caller's
 *       predicated launches -- snnqp_dense_lif_forward_if with the float32 kernel of the first
 *       block into s1_out, snnqp_dense_lif_forward_if of the second block, snnqp_vote_if -- redo
 *       the head).
 * w1/wt1  int8 codes [K][N1] and their tiles (snnqp_pack_codes_mfma, Npad = N1 rounded up to 32,
 *       K rows zero-padded to a multiple of 32); w2/wt2 the same for [N1][N2] (its K is N1).
 * logits  float32 [B][N2 / group], as snnqp_vote gives them.
 * s1_out / s2_out  nullable: the two spike rasters, SNNQP_BITS [T][B][ceil(N / 32)], as
 *       snnqp_dense_lif_forward writes them (the hidden raster otherwise never leaves the CU).
 * Membrane potentials start from zero (initialize_carry, spiking_learning.py:464-472) and are
 * not returned: a caller that carries state uses snnqp_dense_lif_forward per block.
 * ws / ws_bytes  nullable workspace (device memory, 256-byte aligned, one launch at a time; its
 *       tickets are zeroed on `stream` by every call, as for snnqp_dense_lif_forward_ws;
 *       snnqp_dense_head_workspace_bytes says how much, 0 = it would not be used).  With it a batch that fills at most half the chip (config C2: B = 256) runs as two
 *       workgroups per tile of samples, each with half of the hidden columns -- half of the first
 *       block's codes through a CU's L1, the bound of the launch -- which hand their halves of the
 *       hidden raster over through `ws`; the last arriver runs the second block and the vote.
 * SNNQP_EUNSUPPORTED (nothing enqueued) unless N1 <= 512, N2 <= 128, 1 <= T <= 64 and both
 * weights are W_I8 with tiles: the caller then runs the blocks one by one. */
int64_t snnqp_dense_head_workspace_bytes(int32_t T, int32_t B, int32_t N1);
int snnqp_dense_head_forward(const void *x, int in_type, int64_t x_stride_t,
                             int64_t x_stride_b, int32_t T, int32_t B, int32_t K,
                             int32_t N1, const snnqp_weight_t *w1, const int8_t *wt1,
                             const snnqp_neuron_t *nrn1, int32_t N2,
                             const snnqp_weight_t *w2, const int8_t *wt2,
                             const snnqp_neuron_t *nrn2, int32_t group,
                             uint32_t *s1_out, uint32_t *s2_out, float *logits,
                             int32_t *x_flags, void *ws, int64_t ws_bytes, snnqp_stream_t stream);

/* ---- element-wise pieces ----------------------------------------------------
 * replaces: neural_dynamics(u, x) scanned over T, spiking_learning.py:460
 *           (multi_step_LIF :403-416, parametric_leaky_IF :370-387, LIF :426-438)
 *           with the optional norm_fn of :457-458 in front.
 * x float32 [T][R][C] currents; u0/u_out [R][C]; s_out [T][R][C] (F32/BITS). */
int snnqp_lif_forward(const float *x, int32_t T, int64_t R, int32_t C,
                      const snnqp_bn_t *bn, const snnqp_neuron_t *nrn,
                      const float *u0, float *u_out, void *s_out, int s_type,
                      snnqp_stream_t stream);

/* replaces: nn.BatchNorm(use_running_average=True), examples/tcja/models.py:101-107 */
int snnqp_batchnorm_forward(const float *x, int64_t rows, int32_t C,
                            const snnqp_bn_t *bn, float *y,
                            snnqp_stream_t stream);

/* replaces: lax.reduce_window(max, (1,1,2,2,1)), examples/tcja/models.py:145-147
 * x [NB][H][W][C] (F32 or BITS) -> y [NB][H/2][W/2][C]. */
int snnqp_maxpool2x2(const void *x, int type, int64_t NB, int32_t H, int32_t W,
                     int32_t C, void *y, snnqp_stream_t stream);

/* ---- TCJA attention pieces (examples/tcja/models.py:41-99) ---------------------
 * replaces: jnp.mean(x_seq, axis=[2, 3]) (:42): x [NB][HW][C] (F32 or BITS) ->
 *           y float32 [NB][C], sequential float32 sum over pixels / HW. */
int snnqp_spatial_mean(const void *x, int type, int64_t NB, int32_t HW, int32_t C,
                       float *y, snnqp_stream_t stream);
/* replaces: jax.nn.sigmoid(conv_c_out * conv_t_out) (:95): g = sigmoid(a * b),
 *           float32 product, logistic in float64 rounded once. */
int snnqp_sigmoid_gate(const float *a, const float *b, int64_t n, float *g,
                       snnqp_stream_t stream);
/* replaces: x_seq * out[:, :, None, None, :] (:97): y[img][p][c] = x * g[img][c]. */
int snnqp_apply_gate(const void *x, int type, const float *g, int64_t NB, int32_t HW,
                     int32_t C, float *y, snnqp_stream_t stream);

/* ---- event front end and probes ------------------------------------------------
 * replaces: preprocess_data_number, examples/input_pipeline.py:142-219
 *           (split_by = "number"): N time-ordered events (x, y, polarity) ->
 *           T frames of N // T events (the last takes the remainder), counts per
 *           pixel and polarity (channel 0 = polarity 0).  counts int32
 *           [T][H][W][2] (zeroed here); frames_u8 optional saturated copy.
 *           Coordinates are divided by `scale` (config.resolution_scale). */
int snnqp_events_to_frames(const int32_t *ex, const int32_t *ey, const int32_t *ep,
                           int64_t N, int32_t T, int32_t H, int32_t W, float scale,
                           int32_t *counts, uint8_t *frames_u8, snnqp_stream_t stream);
/* replaces: the activation-density probes sown at examples/tcja/models.py:128-142
 *           (consumed by examples/sparsity.py:143-170): non-zero count of each of
 *           NB slices of n elements (C innermost; F32, U8 or BITS). */
int snnqp_density(const void *x, int type, int64_t NB, int64_t n, int32_t C, int32_t *nnz,
                  snnqp_stream_t stream);

/* replaces: the rate "vote", examples/tcja/models.py:253-255
 * s [T][B][N] (F32 or BITS) -> logits float32 [B][N / group]:
 * mean over T (sequential float32 sum / T), then mean over `group` neurons. */
int snnqp_vote(const void *s, int type, int32_t T, int32_t B, int32_t N,
               int32_t group, float *logits, snnqp_stream_t stream);
/* executed only if *pred != 0 (snnqp_conv_lif_forward_if) */
int snnqp_vote_if(const int32_t *pred, const void *s, int type, int32_t T, int32_t B, int32_t N,
                  int32_t group, float *logits, snnqp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNNQP_H_ */
