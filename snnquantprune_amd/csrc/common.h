// Shared host/device helpers for libsnnqp (gfx950 only).
// Build flags matter: -ffp-contract=off keeps every float op of the reference
// as its own rounding (fmaf appears only where it is written).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/snnqp.h"

namespace snnqp {

void set_error(const char *fmt, ...);

#define SNNQP_REQUIRE(cond, code, ...)  \
  do {                                  \
    if (!(cond)) {                      \
      ::snnqp::set_error(__VA_ARGS__);  \
      return (code);                    \
    }                                   \
  } while (0)

#define SNNQP_CHECK_LAUNCH(name)                                         \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      ::snnqp::set_error("%s: %s", (name), hipGetErrorString(e__));      \
      return SNNQP_EHIP;                                                 \
    }                                                                    \
  } while (0)

#define SNNQP_HIP(call)                                                  \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) {                                             \
      ::snnqp::set_error("%s: %s", #call, hipGetErrorString(e__));       \
      return SNNQP_EHIP;                                                 \
    }                                                                    \
  } while (0)

// ---- dequantisation:  x = fl(fl(acc / L) * m) -----------------------------
// The division is two instructions: t = a * r_lo; q = fma(a, r_hi, t) with the split
// reciprocal r_hi = fl(1/L), r_lo = fl(1/L - r_hi).  The fma rounds a/L * (1 + ~2^-48)
// once, and a quotient of an integer |a| < 2^24 by L is never that close to a rounding
// boundary of float32 (distance >= 2^-24 / L relative): it IS the IEEE quotient.
// oracle/oracle_c.c:oracle_check_div counts the exceptions exhaustively (none) for
// every L = 2^(b-1) - 1 (tests/test_oracle_cpu.py).
struct Dequant {
  float L, rL, rLlo, m;
};

inline Dequant make_dequant(float L, float m) {
  Dequant d;
  d.L = L;
  d.rL = 1.0f / L;
  d.rLlo = (float)(1.0 / (double)L - (double)d.rL);
  d.m = m;
  return d;
}

// exact a / L for an integer-valued a (L == 1: t = 0, q = a)
__device__ __forceinline__ float div_exact(float a, const Dequant &d) {
  return __builtin_fmaf(a, d.rL, a * d.rLlo);
}

__device__ __forceinline__ float dequant_acc_nb(int acc, const Dequant &d) {
  return div_exact((float)acc, d) * d.m;
}

__device__ __forceinline__ float dequant_acc(int acc, const Dequant &d) {
  return div_exact((float)acc, d) * d.m;
}

// ---- neuron update, spiking_learning.py:357-438 ----------------------------
struct NeuronP {
  int kind;
  float k;       // tau (MULTI_STEP_LIF) or sigmoid(tau_param) (PLIF)
  float inv_k;   // multiplier m of the update u += (x - (u - v_reset)) * m when the neuron
                 // has that form: 1/tau for MULTI_STEP_LIF with tau a power of two (the
                 // division is then exact), sigmoid(tau) for PLIF; else 0
  int k_log2;    // log2(1 / inv_k) when that is an integer (MULTI_STEP_LIF), else -1
  float vth, vr;
  const float *decay;
};

inline NeuronP make_neuron(const snnqp_neuron_t *n) {
  NeuronP p;
  p.kind = n ? n->kind : SNNQP_NEURON_NONE;
  p.k = n ? n->k : 1.0f;
  p.vth = n ? n->v_threshold : 1.0f;
  p.vr = n ? n->v_reset : 0.0f;
  p.decay = n ? n->decay : nullptr;
  p.inv_k = 0.0f;
  p.k_log2 = -1;
  if (p.kind == SNNQP_NEURON_PARAMETRIC_LEAKY_IF) p.inv_k = p.k;   // :381 multiplies
  if (p.kind == SNNQP_NEURON_MULTI_STEP_LIF) {
    int e;
    float mant = frexpf(p.k, &e);
    // x / 2^j == x * 2^-j exactly (same real value, one rounding)
    if (mant == 0.5f && e > -100 && e < 100) {
      p.inv_k = 1.0f / p.k;
      p.k_log2 = e - 1;
    }
  }
  return p;
}

// Returns the spike; updates u.  `dec` is the per-feature decay (LIF only).
__device__ __forceinline__ bool neuron_step(float &u, float x, const NeuronP &p,
                                            float dec) {
  if (p.kind == SNNQP_NEURON_MULTI_STEP_LIF) {
    float d = x - (u - p.vr);                       // :410
    u = u + (p.inv_k != 0.0f ? d * p.inv_k : d / p.k);
  } else if (p.kind == SNNQP_NEURON_PARAMETRIC_LEAKY_IF) {
    float d = x - (u - p.vr);                       // :381
    u = u + d * p.k;
  } else {
    u = u * dec + x;                                // :432
  }
  bool s = (u - p.vth) >= 0.0f;                     // :412 / :224 Heaviside
  u = s ? p.vr : u;                                 // :414
  return s;
}

struct BnP {
  const float *mean, *mul, *bias;
  int flags;            // SNNQP_BN_MEAN_ZERO | SNNQP_BN_BIAS_ZERO | SNNQP_BN_MUL_UNIFORM (caller-asserted)
};

// a BatchNorm descriptor is either absent or complete, and carries no flag bits this build
// does not know (an uninitialised `flags` of a caller built against an older header must
// not silently select the mean / bias shortcuts)
#define SNNQP_CHECK_BN(bn)                                                              \
  do {                                                                                  \
    if (bn) {                                                                           \
      SNNQP_REQUIRE((bn)->mean && (bn)->mul && (bn)->bias, SNNQP_EINVAL,                \
                    "batch-norm descriptor with null arrays");                          \
      SNNQP_REQUIRE(((bn)->flags & ~(SNNQP_BN_MEAN_ZERO | SNNQP_BN_BIAS_ZERO | SNNQP_BN_MUL_UNIFORM)) == 0, \
                    SNNQP_EINVAL, "batch-norm descriptor with unknown flag bits 0x%x "  \
                    "(built against another snnqp.h?)", (unsigned)(bn)->flags);         \
    }                                                                                   \
  } while (0)

inline BnP make_bn(const snnqp_bn_t *b) {
  BnP p;
  p.mean = b ? b->mean : nullptr;
  p.mul = b ? b->mul : nullptr;
  p.bias = b ? b->bias : nullptr;
  p.flags = b ? b->flags : (SNNQP_BN_MEAN_ZERO | SNNQP_BN_BIAS_ZERO | SNNQP_BN_MUL_UNIFORM);   // (no BatchNorm: mul = 1)
  return p;
}

__device__ __forceinline__ float bn_apply(float x, float mean, float mul,
                                          float bias) {
  x = x - mean;
  x = x * mul;
  return x + bias;
}

inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace snnqp
