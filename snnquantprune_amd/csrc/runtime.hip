// Per-device state of libsnnqp that is not a kernel:
//  * the STATUS WORD: four bytes of page-locked host memory mapped into the device's address
//    space.  A kernel that finds an invariant of its own bookkeeping broken (a work queue whose
//    patches do not add up at the end of a launch, a split-K ticket out of range) stores a code
//    there; the host reads the word, without any synchronisation, at the start of every fused
//    block call and refuses to go on (SNNQP_EHIP, the message says which invariant) until
//    snnqp_device_status(..., reset = 1) has been called: results that may be wrong are an
//    error, not a slower path;
//  * the DENORMAL PROBE: the table form of the bit-input conv kernels (DQ_TABLE,
//    conv3x3_bits.hip) needs v_mfma_scale_f32_32x32x64_f8f6f4 to add float32 DENORMALS exactly
//    -- a property of the matrix pipe that no document states.  Once per device, at the first
//    launch that wants the table form, one wave runs chains over the whole denormal range the
//    tables use and compares the accumulators' bit patterns with the integer sums; a device (or
//    firmware, or runtime) on which they differ gets the arithmetic form (DQ_ARITH: the same
//    results from three float32 instructions per value) and the fallback is counted
//    (snnqp_workqueue_stats).  Reference semantics at stake: quant.py:443,467 (the dequantised
//    current) feeding spiking_learning.py:410-414.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace snnqp {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int PROBE_CHAIN = 18;      // MFMAs per chain, as conv1 / conv2 issue them

struct DeviceState {
  uint32_t *status_host = nullptr;
  uint32_t *status_dev = nullptr;
  bool status_failed = false;        // no page-locked word could be made: kernels get nullptr
  int denorm = 0;                    // 0 not probed yet, 1 exact, -1 not exact / probe failed
};
std::mutex g_rt_mu;
DeviceState g_rt[64];
std::atomic<int64_t> g_dq_fallbacks{0};

struct RtDeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit RtDeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
    if (cur == dev) return;
    if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
    prev = cur;
  }
  ~RtDeviceGuard() {
    if (prev >= 0 && hipSetDevice(prev) != hipSuccess) (void)hipGetLastError();
  }
};

}  // namespace

__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t *p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}

int zero_words_async(uint32_t *p, int64_t nwords, hipStream_t st) {
  if (!p || nwords <= 0) return SNNQP_OK;
  const int64_t blocks = (nwords + 255) / 256;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, p, nwords);
  SNNQP_CHECK_LAUNCH("zero_words_kernel");
  return SNNQP_OK;
}

namespace {

// One wave: A = all 1.0 (fp4), B = fp6 codes of the lane's column, PROBE_CHAIN MFMAs of K = 64
// with the block scales of DQ_TABLE (2^-63 * 2^-84 = 2^-147 per spike x code unit = 4 denormal
// steps), the chain started from the bit pattern 4 * off.  out[lane][i] = accumulator bits.
__global__ void __launch_bounds__(64)
denorm_probe_kernel(const int *__restrict__ b6, uint32_t cinit_bits, uint32_t *__restrict__ out) {
  const int lane = threadIdx.x;
  const v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = __uint_as_float(cinit_bits);
  for (int c = 0; c < PROBE_CHAIN; ++c) {
    v8i b = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 6; ++r) b[r] = b6[(c * 64 + lane) * 6 + r];
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4 /* fp4 */, 2 /* fp6 */, 0, 64, 0, 43);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = __float_as_uint(acc[i]);
}

int enc6(int v) {                    // integer -7..7 -> e2m3
  static const int mag[8] = {0, 8, 16, 20, 24, 26, 28, 30};
  return (v < 0 ? 32 : 0) | mag[v < 0 ? -v : v];
}

// true iff every accumulator of every chain came out as 4 * (off + column sum) exactly
bool run_denorm_probe() {
  // three code patterns: all +7 (the positive end: 18 * 64 * 7 = 8064 units), all -7 from an
  // offset at the top, and pseudo-random sparse codes as a pruned layer has them
  const int OFF = 8192;
  std::vector<int> b6((size_t)PROBE_CHAIN * 64 * 6);
  int *db = nullptr;
  uint32_t *dout = nullptr;
  if (hipMalloc((void **)&db, b6.size() * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMalloc((void **)&dout, 64 * 16 * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(db); return false; }
  bool ok = true;
  uint32_t lcg = 12345u;
  for (int pattern = 0; pattern < 3 && ok; ++pattern) {
    long colsum[32] = {0};
    for (int c = 0; c < PROBE_CHAIN; ++c)
      for (int lane = 0; lane < 64; ++lane) {
        unsigned char bytes[24] = {0};
        for (int j = 0; j < 32; ++j) {
          int v;
          if (pattern == 0) v = 7;
          else if (pattern == 1) v = -7;
          else { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 24) % 10 == 0) ? (int)((lcg >> 8) % 15) - 7 : 0; }
          colsum[lane & 31] += v;
          const int e = enc6(v), bit = 6 * j;
          for (int q = 0; q < 6; ++q)
            if ((e >> q) & 1) bytes[(bit + q) >> 3] |= (unsigned char)(1 << ((bit + q) & 7));
        }
        memcpy(&b6[((size_t)c * 64 + lane) * 6], bytes, 24);
      }
    if (hipMemcpy(db, b6.data(), b6.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { ok = false; break; }
    hipLaunchKernelGGL(denorm_probe_kernel, dim3(1), dim3(64), 0, 0, db, 4u * OFF, dout);
    uint32_t out[64 * 16];
    if (hipGetLastError() != hipSuccess ||
        hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
    for (int lane = 0; lane < 64 && ok; ++lane)
      for (int i = 0; i < 16; ++i)
        if ((long)out[lane * 16 + i] != 4 * (colsum[lane & 31] + OFF)) { ok = false; break; }
  }
  (void)hipGetLastError();
  (void)hipFree(db);
  (void)hipFree(dout);
  return ok;
}

DeviceState *state_of(int dev) { return dev >= 0 && dev < 64 ? &g_rt[dev] : nullptr; }

// one thread per output column: the two one-sided code sums against the caller's bound
__global__ void __launch_bounds__(256)
check_code_bound_kernel(const int8_t *__restrict__ w, int64_t K, int32_t N, int32_t bound,
                        uint32_t *status) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int64_t pos = 0, neg = 0;
  for (int64_t k = 0; k < K; ++k) {
    const int c = w[k * N + n];
    pos += c > 0 ? c : 0;
    neg += c < 0 ? -c : 0;
  }
  if ((pos > bound || neg > bound) && status) *(volatile uint32_t *)status = SNNQP_STATUS_BOUND;
}

// (codes pointer, K, N, bound) tuples already checked on a device: a small ring
struct Checked { const void *w; int64_t K; int32_t N, bound; };
Checked g_checked[64][32];
unsigned g_checked_next[64];

}  // namespace

// Device pointer of the status word of `dev` (made on first use), or nullptr: the kernels then
// keep their checks to themselves.  Never called for the first time during a stream capture by
// the conv path (the work-queue pool, made by an eager launch, comes first).
uint32_t *device_status_word(int dev) {
  DeviceState *s = state_of(dev);
  if (!s) return nullptr;
  std::lock_guard<std::mutex> lock(g_rt_mu);
  if (s->status_dev || s->status_failed) return s->status_dev;
  RtDeviceGuard on(dev);
  void *h = nullptr, *d = nullptr;
  if (!on.ok || hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    s->status_failed = true;
    return nullptr;
  }
  memset(h, 0, 64);
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    s->status_failed = true;
    return nullptr;
  }
  s->status_host = (uint32_t *)h;
  s->status_dev = (uint32_t *)d;
  return s->status_dev;
}

// The sticky codes a kernel on `dev` has reported (0: none).  A plain host read.
uint32_t device_status_read(int dev) {
  DeviceState *s = state_of(dev);
  if (!s) return 0;
  std::lock_guard<std::mutex> lock(g_rt_mu);
  return s->status_host ? *(volatile uint32_t *)s->status_host : 0u;
}

const char *device_status_text(uint32_t code) {
  if (code & SNNQP_STATUS_QUEUE_CORRUPT)
    return "a conv launch finished with a patch count that does not match its work queue (a queue "
           "word was not zero when the launch began: an aborted launch or a graph replayed "
           "concurrently with itself): spike rasters since then may be wrong";
  if (code & SNNQP_STATUS_BOUND)
    return "snnqp_weight_t.abs_sum_max is smaller than a one-sided code sum of the weights it came "
           "with: the dequantisation tables of the conv kernels were read outside their range, "
           "spike rasters of that layer are wrong";
  if (code & SNNQP_STATUS_TICKET)
    return "a split-K dense launch drew a ticket outside its range (its counters were not zero "
           "when the launch began): spike rasters since then may be wrong";
  return "unknown device status";
}

// Whether DQ_TABLE may be used on the device of `st`.  The probe synchronises (two small
// copies), so it cannot run while `st` is being captured: an unprobed device then answers
// "no" for this launch only.
bool dq_table_trusted(int dev, hipStream_t st) {
  DeviceState *s = state_of(dev);
  if (!s) return false;
  {
    std::lock_guard<std::mutex> lock(g_rt_mu);
    if (s->denorm != 0) {
      if (s->denorm < 0) g_dq_fallbacks.fetch_add(1, std::memory_order_relaxed);
      return s->denorm > 0;
    }
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cap != hipStreamCaptureStatusNone) {
    g_dq_fallbacks.fetch_add(1, std::memory_order_relaxed);
    return false;
  }
  RtDeviceGuard on(dev);
  bool ok = on.ok && run_denorm_probe();
  static const bool force_fail = std::getenv("SNNQP_FORCE_DENORM_PROBE_FAIL") != nullptr;   // test hook
  if (force_fail) ok = false;
  std::lock_guard<std::mutex> lock(g_rt_mu);
  s->denorm = ok ? 1 : -1;
  if (!ok) g_dq_fallbacks.fetch_add(1, std::memory_order_relaxed);
  return ok;
}

// The LDS tables of the conv kernels are sized by the caller's snnqp_weight_t.abs_sum_max, and
// the accumulator addresses them as it is: a bound that is too small reads outside the table.
// The first launch that sees a (codes, shape, bound) tuple on a device also launches a check of
// the bound against the codes (one thread per output channel, a few microseconds); a violation
// goes to the status word, i.e. the NEXT call fails.  The tuple is remembered (a ring of 32 per
// device): codes rewritten in place under an unchanged bound are not checked again.
void check_code_bound_once(int dev, const int8_t *w, int64_t K, int32_t N, int32_t bound, hipStream_t st) {
  if (dev < 0 || dev >= 64 || !w || bound <= 0) return;
  {
    std::lock_guard<std::mutex> lock(g_rt_mu);
    for (const Checked &c : g_checked[dev])
      if (c.w == w && c.K == K && c.N == N && c.bound == bound) return;
  }
  uint32_t *status = device_status_word(dev);
  if (!status) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
  if (cap != hipStreamCaptureStatusNone) return;      // not into a graph: the next eager launch checks
  hipLaunchKernelGGL(check_code_bound_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, w, K, N,
                     bound, status);
  if (hipGetLastError() != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_rt_mu);
  g_checked[dev][g_checked_next[dev]++ % 32] = Checked{w, K, N, bound};
}

int64_t dq_table_fallbacks(bool reset) {
  return reset ? g_dq_fallbacks.exchange(0) : g_dq_fallbacks.load(std::memory_order_relaxed);
}

// a kernel on the device of `st` reported a broken invariant: the block entry points refuse to go on
int refuse_after_device_report(hipStream_t st, const char *who) {
  const uint32_t code = device_status_read(stream_device(st));
  if (code) {
    set_error("%s: device status 0x%x: %s (snnqp_device_status(..., reset = 1) clears it)", who,
              (unsigned)code, device_status_text(code));
    return SNNQP_EHIP;
  }
  return SNNQP_OK;
}

}  // namespace snnqp

extern "C" int snnqp_device_status(int device, uint32_t *codes, int reset) {
  using namespace snnqp;
  SNNQP_REQUIRE(device >= 0 && device < 64, SNNQP_EINVAL, "device_status: device %d out of range", device);
  const uint32_t c = device_status_read(device);
  if (codes) *codes = c;
  if (reset) {
    std::lock_guard<std::mutex> lock(g_rt_mu);
    if (g_rt[device].status_host) *(volatile uint32_t *)g_rt[device].status_host = 0u;
  }
  return SNNQP_OK;
}
