"""CPU oracle for the quantized / pruned SNN time-stepped forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``snnquantprune_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the reported CPU baseline.

It restates, op for op and in float32, the arithmetic of the reference
(paths relative to the reference checkout):

  quant.py:88-90      round_* forward  = jnp.round (round half to even)
  quant.py:296-314    max_init / gaussian_init / percentile_init
  quant.py:322-358    uniform_static
  quant.py:361-425    parametric_d
  quant.py:428-469    DuQ
  quant.py:472-491    prune
  quant.py:494-625    parametric_d_xmax
  flax_qdense.py:58-106   QuantDense.__call__
  flax_qconv.py:93-188    QuantConv.__call__ (padding resolution :131-144)
  spiking_learning.py:139-241  spike functions (forward = Heaviside x >= 0)
  spiking_learning.py:357-438  parametric_leaky_IF / multi_step_LIF / LIF
  spiking_learning.py:441-472  SpikingBlock (scan over T) / initialize_carry
  examples/tcja/models.py:101-147,189-190,200-255  BN, 2x2 max-pool, flatten, vote
  examples/train_utils.py:210-225,370-390          mse_loss, compute_metrics, eval_step
  examples/train_inpt_spikingjelly.py:147-223      local / global magnitude masks

Pinning status.  The reference is Python on jax==0.2.27 /
flax==0.4.0 (README.md:15-32), neither of which is installed here, so it cannot
be imported or run and no golden outputs of the reference exist.  The oracle is
pinned by the invariants the reference's own tests assert:
  (1) no-quant QuantDense == x @ W            (flax_qdense_test.py:153-250)
  (2) no-quant QuantConv == NHWC/HWIO conv over the nine geometries of
      flax_qconv_test.py:148-285 (output sizes listed there)
  (3) integer data within the code range round-trips every quantiser
      (quant_test.py:141-185)
  (4) a signed b-bit quantiser emits 2**b - 1 distinct values (quant_test.py:187-250)
and by source-derived known answers (LIF constant-drive sequences, DuQ code
ranges, round-half-even ties).  For multi_step_LIF / SpikingBlock / DuQ / prune /
BatchNorm-in-scan / pooling / vote and every end-to-end number the reference's
tests hold no expected outputs: **parity unpinned** beyond those known answers.

Three arithmetic modes for the weight x input contraction:

  'int'    integer codes x integer inputs accumulated exactly, then
           y = fl(fl(acc / L) * m) with L = n_lv - 1, m = c (DuQ) or L = 1,
           m = step (the other quantisers).  This is the bit-exact contract
           for spike rasters and integer accumulators.
  'fseq'   float32 inputs x float32 (fake-quantised) weights as a k-ascending
           fmaf chain (k = flattened (kh, kw, cin) for convolutions).  Bit-exact
           contract for unquantised / real-valued layers.
  'float'  reference-literal: dense float32 fake-quantised weights, float32
           matmul / conv in BLAS order (what XLA-CPU executes, summation order
           unspecified).  Used to report the flip rate of 'int'/'fseq' against
           float summation order and as the timed CPU baseline.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np

F32 = np.float32

_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------
# C helper (exact fmaf chains).  Built by oracle/Makefile or on first use.
# ---------------------------------------------------------------------------

_clib = None


def build_c(force: bool = False) -> str:
  so = os.path.join(_HERE, "liboracle_c.so")
  src = os.path.join(_HERE, "oracle_c.c")
  if force or not os.path.exists(so) or (
      os.path.getmtime(so) < os.path.getmtime(src)):
    subprocess.check_call(
        ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared",
         "-fPIC", "-fopenmp", "-o", so, src, "-lm"])
  return so


def clib():
  global _clib
  if _clib is None:
    lib = ctypes.CDLL(build_c())
    i64, fp = ctypes.c_int64, ctypes.POINTER(ctypes.c_float)
    lib.oracle_fseq_matmul.argtypes = [fp, fp, fp, i64, i64, i64]
    lib.oracle_fseq_matmul.restype = None
    lib.oracle_check_div.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.oracle_check_div.restype = i64
    lib.oracle_fma_rows.argtypes = [fp, fp, fp, i64, i64]
    lib.oracle_fma_rows.restype = None
    _clib = lib
  return _clib


def _fp(a):
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def fseq_matmul(x: np.ndarray, w: np.ndarray) -> np.ndarray:
  """y[m, n] = fmaf chain over k ascending of x[m, k] * w[k, n], start +0."""
  x = np.ascontiguousarray(x, dtype=F32)
  w = np.ascontiguousarray(w, dtype=F32)
  m, k = x.shape
  k2, n = w.shape
  assert k == k2
  y = np.empty((m, n), dtype=F32)
  clib().oracle_fseq_matmul(_fp(x), _fp(w), _fp(y), m, k, n)
  return y


# ---------------------------------------------------------------------------
# Rounding / calibration / quantisers   (quant.py)
# ---------------------------------------------------------------------------


def round_half_even(x):
  """Forward of every round_* in quant.py:26-288 (jnp.round)."""
  return np.rint(np.asarray(x, dtype=F32)).astype(F32)


def max_init(x, bits, sign=True):
  """quant.py:296-298."""
  x = np.asarray(x, dtype=F32)
  if np.max(x) == 0:
    return F32(1 / 2 ** bits)
  return F32(np.max(np.abs(x)))


def gaussian_init(x, bits, sign=True):
  """quant.py:305-309: max(|mu - 3 sigma|, |mu + 3 sigma|)."""
  x = np.asarray(x, dtype=F32)
  if np.max(x) == 0:
    return F32(1 / 2 ** bits)
  mu = np.mean(x, dtype=F32)
  sigma = np.std(x, dtype=F32)
  return F32(np.maximum(np.abs(mu - F32(3) * sigma), np.abs(mu + F32(3) * sigma)))


def percentile_init(x, bits, sign, perc):
  """quant.py:312-314."""
  x = np.asarray(x, dtype=F32)
  if np.max(x) == 0:
    return F32(1 / 2 ** bits)
  return F32(np.percentile(np.abs(x), perc))


def hard_tanh(x):
  return np.clip(x, F32(-1), F32(1))


def duq_levels(bits: int, sign: bool = True) -> int:
  """n_lv of quant.py:458-461."""
  return 2 ** (bits - 1) if sign else 2 ** bits


def duq_codes(w, a, bits, sign=True):
  """Integer code q = round(hard_tanh(W / a) * (n_lv - 1)), quant.py:443,466-467.

  Returned as integer-valued float32, range [-(n_lv-1), n_lv-1]."""
  w = np.asarray(w, dtype=F32)
  L = F32(duq_levels(bits, sign) - 1)
  x = hard_tanh(w / F32(a))
  return round_half_even(x * L)


def duq_dequant(q, c, bits, sign=True):
  """w = fl(fl(q / (n_lv-1)) * c), quant.py:443,467."""
  L = F32(duq_levels(bits, sign) - 1)
  return (np.asarray(q, dtype=F32) / L) * F32(c)


def duq_forward(w, a, c, bits, sign=True):
  """DuQ.__call__, quant.py:428-469."""
  w = np.asarray(w, dtype=F32)
  if bits == -1:                     # quant.py:453-454
    return w
  if F32(a) == F32(-1):              # quant.py:469
    return w
  return duq_dequant(duq_codes(w, a, bits, sign), c, bits, sign).astype(F32)


def uniform_static_forward(x, xmax, bits, sign=True):
  """uniform_static.__call__, quant.py:331-358 (xmax = dynamic_range_no_train)."""
  assert bits > 1
  x = np.asarray(x, dtype=F32)
  xmax = F32(xmax)
  num_levels = F32(2 ** (bits - 1) - 1 if sign else 2 ** bits - 1)
  x = x / xmax
  x = np.clip(x, F32(-1) if sign else F32(0), F32(1)) * xmax
  scale = xmax / num_levels
  return (round_half_even(x / scale) * scale).astype(F32)


def uniform_static_init(x, bits, sign=True, init_fn=max_init):
  """quant.py:345-347."""
  v = init_fn(x, bits=bits, sign=sign)
  return F32(1.0) if v == 0 else F32(v)


def parametric_d_forward(x, step, bits, sign=True):
  """parametric_d.__call__, quant.py:374-425 (forward; gradscale is identity)."""
  x = np.asarray(x, dtype=F32)
  s = F32(step)
  q_pos = F32(2 ** (bits - 1) - 1 if sign else 2 ** bits - 1)
  q_neg = -q_pos if sign else F32(0)
  v = np.clip(x / s, q_neg, q_pos)
  return (round_half_even(v) * s).astype(F32)


def parametric_d_init(x, bits, sign=True, init_fn=max_init):
  """quant.py:395-399: step = init_fn(x) / sqrt(q_pos)."""
  q_pos = 2 ** (bits - 1) - 1 if sign else 2 ** bits - 1
  return F32(F32(init_fn(x, bits=bits, sign=sign)) / np.sqrt(F32(q_pos)))


def parametric_d_xmax_forward(x, d, xmax, sign=True, d_min=2 ** -12, d_max=1.0,
                              xmax_min=2 ** -8, xmax_max=127.0):
  """parametric_d_xmax.__call__ forward, quant.py:512-625."""
  x = np.asarray(x, dtype=F32)
  d = np.clip(F32(d), F32(d_min), F32(d_max))            # :579
  xmax = np.clip(F32(xmax), F32(xmax_min), F32(xmax_max))  # :582
  x = x / xmax
  x = np.clip(x, F32(-1) if sign else F32(0), F32(1)) * xmax
  return (d * round_half_even(x / d)).astype(F32)


def parametric_d_xmax_init(x, bits, sign=True, init_fn=None, act=False,
                           maxabs_w=None):
  """(d, xmax) of quant.py:553-573."""
  x = np.asarray(x, dtype=F32)
  num_levels = 2 ** (bits - 1) - 1 if sign else 2 ** bits - 1
  if init_fn is None:
    if act:
      return F32(2 ** -3), F32(2 ** -3 * (2. ** bits - 1))
    mw = F32(maxabs_w) if maxabs_w is not None else F32(np.max(np.abs(x)))
    lg = np.log2(mw / F32(2 ** (bits - 1) - 1))
    d = F32(2 ** (np.ceil(lg) if bits > 4 else np.floor(lg)))
    return d, F32(d * F32(2 ** (bits - 1) - 1))
  xmax = F32(init_fn(x, bits=bits, sign=sign))
  xmax = F32(1.0) if xmax == 0 else xmax
  return F32(xmax / F32(num_levels)), xmax


def prune_forward(w, mask):
  """prune.__call__, quant.py:489-491: W * mask."""
  return (np.asarray(w, dtype=F32) * np.asarray(mask, dtype=F32)).astype(F32)


def local_prune_mask(kernel, p):
  """update_prune_mask, examples/train_inpt_spikingjelly.py:147-157."""
  kernel = np.asarray(kernel)
  mask = np.ones(kernel.shape)
  k = int(np.prod(kernel.shape) * p)
  idx = np.argpartition(np.abs(kernel).reshape(-1), k)[:k]
  mask.reshape(-1)[idx] = 0
  return mask.astype(F32)


def global_prune_masks(kernels: Sequence[np.ndarray], p):
  """Global magnitude mask, examples/train_inpt_spikingjelly.py:174-223."""
  flat = np.concatenate([np.reshape(np.asarray(k), (-1)) for k in kernels])
  gm = np.ones(flat.shape)
  k = int(np.prod(flat.shape) * p)
  idx = np.argpartition(np.abs(flat), k)[:k]
  gm[idx] = 0
  out, off = [], 0
  for kern in kernels:
    n = int(np.prod(kern.shape))
    out.append(gm[off:off + n].reshape(kern.shape).astype(F32))
    off += n
  return out


# ---------------------------------------------------------------------------
# Layer weights: one description consumed by every contraction mode.
# ---------------------------------------------------------------------------


class QWeight:
  """A layer kernel after the weight transforms of flax_qdense.py:74-85.

  kernel   float32 raw kernel ([K, N] dense, HWIO conv)
  q        integer codes (integer-valued float32) after the mask, or None
  L, m     dequantisation y = fl(fl(acc / L) * m); None for unquantised
  w_fq     float32 fake-quantised + pruned kernel (the reference's kernel_fwd)
  """

  def __init__(self, kernel, quant: Optional[dict] = None, mask=None):
    kernel = np.asarray(kernel, dtype=F32)
    self.kernel = kernel
    self.q = None
    self.L = None
    self.m = None
    quant = quant or {}
    kind = quant.get("kind")
    bits = quant.get("bits", 8)
    if kind is None or bits == -1:
      w = kernel
    elif kind == "duq":
      a, c = F32(quant["a"]), F32(quant["c"])
      if a == F32(-1):
        w = kernel
      else:
        self.q = duq_codes(kernel, a, bits)
        self.L = F32(duq_levels(bits) - 1)
        self.m = c
        w = duq_dequant(self.q, c, bits)
    elif kind == "uniform_static":
      xmax = F32(quant["xmax"])
      L = F32(2 ** (bits - 1) - 1)
      scale = xmax / L
      xc = np.clip(kernel / xmax, F32(-1), F32(1)) * xmax
      self.q = round_half_even(xc / scale)
      self.L, self.m = F32(1), scale
      w = self.q * scale
    elif kind == "parametric_d":
      s = F32(quant["step"])
      qp = F32(2 ** (bits - 1) - 1)
      self.q = round_half_even(np.clip(kernel / s, -qp, qp))
      self.L, self.m = F32(1), s
      w = self.q * s
    elif kind == "parametric_d_xmax":
      d = np.clip(F32(quant["d"]), F32(quant.get("d_min", 2 ** -12)),
                  F32(quant.get("d_max", 1.0)))
      xmax = np.clip(F32(quant["xmax"]), F32(quant.get("xmax_min", 2 ** -8)),
                     F32(quant.get("xmax_max", 127.0)))
      xc = np.clip(kernel / xmax, F32(-1), F32(1)) * xmax
      self.q = round_half_even(xc / d)
      self.L, self.m = F32(1), F32(d)
      w = F32(d) * self.q
    else:
      raise ValueError(kind)
    w = np.asarray(w, dtype=F32)
    if mask is not None:
      mask = np.asarray(mask, dtype=F32)
      w = w * mask                       # quant.py:491, after quantisation
      if self.q is not None:
        assert np.all((mask == 0) | (mask == 1)), "int mode needs a 0/1 mask"
        self.q = self.q * mask
    self.w_fq = w.astype(F32)

  @property
  def quantised(self):
    return self.q is not None

  def dequant_acc(self, acc):
    """y = fl(fl(acc / L) * m) on an exact integer accumulator."""
    a = np.asarray(acc).astype(F32)       # round-to-nearest-even like v_cvt_f32_i32
    return ((a / self.L) * self.m).astype(F32)


# ---------------------------------------------------------------------------
# Contractions
# ---------------------------------------------------------------------------


def _exact_int_matmul(x, q):
  """Exact integer matmul using BLAS on floats (guards exactness)."""
  x = np.asarray(x)
  q = np.asarray(q)
  bound = float(np.max(np.abs(x), initial=0)) * float(
      np.max(np.abs(q), initial=0)) * x.shape[-1]
  if bound < 2 ** 24:
    acc = np.asarray(x, dtype=F32) @ np.asarray(q, dtype=F32)
  else:
    assert bound < 2 ** 53
    acc = np.asarray(x, dtype=np.float64) @ np.asarray(q, dtype=np.float64)
  return np.rint(acc).astype(np.int64)


def _is_integer_valued(x):
  x = np.asarray(x)
  return np.issubdtype(x.dtype, np.integer) or x.dtype == np.bool_ or bool(
      np.all(x == np.rint(x)))


def dense_acc(x, qw: QWeight):
  """Integer accumulator of QuantDense in 'int' mode ([..., N] int64)."""
  x = np.asarray(x)
  lead = x.shape[:-1]
  acc = _exact_int_matmul(x.reshape(-1, x.shape[-1]), qw.q)
  return acc.reshape(lead + (qw.q.shape[-1],))


# The summation order of the 'float' mode.  The reference's order is whatever XLA's CPU backend
# picks (flax_qconv.py:158-168, flax_qdense.py:87-89) and cannot be observed here; the default
# stand-in is BLAS.  oracle/int_vs_float.py installs other orders (K permuted, strictly sequential,
# pairwise tree) to bound how much ANY order can move a potential or flip a spike.
FLOAT_MATMUL = None


def float_matmul(a, w):
  return (a @ w) if FLOAT_MATMUL is None else FLOAT_MATMUL(a, w)


def quant_dense(x, qw: QWeight, mode: str = "int"):
  """QuantDense.__call__ without bias, flax_qdense.py:58-89."""
  x = np.asarray(x)
  lead = x.shape[:-1]
  x2 = x.reshape(-1, x.shape[-1])
  if mode == "int":
    assert qw.quantised and _is_integer_valued(x2)
    y = qw.dequant_acc(_exact_int_matmul(x2, qw.q))
  elif mode == "fseq":
    y = fseq_matmul(x2.astype(F32), qw.w_fq)
  elif mode == "float":
    y = float_matmul(x2.astype(F32), qw.w_fq)
  else:
    raise ValueError(mode)
  return y.reshape(lead + (qw.w_fq.shape[-1],)).astype(F32)


def resolve_padding(in_spatial, k_spatial, strides, padding, rhs_dilation=None):
  """String padding -> explicit pads, flax_qconv.py:131-144 (padtype_to_pads).

  SAME: out = ceil(in / stride); total = max((out-1)*stride + k_eff - in, 0);
  lo = total // 2, hi = total - lo.  VALID: zeros."""
  n = len(in_spatial)
  rhs_dilation = rhs_dilation or (1,) * n
  if isinstance(padding, str):
    pads = []
    for i in range(n):
      k_eff = (k_spatial[i] - 1) * rhs_dilation[i] + 1
      if padding.upper() == "SAME":
        out = -(-in_spatial[i] // strides[i])
        total = max((out - 1) * strides[i] + k_eff - in_spatial[i], 0)
        pads.append((total // 2, total - total // 2))
      elif padding.upper() == "VALID":
        pads.append((0, 0))
      else:
        raise ValueError(padding)
    return tuple(pads)
  return tuple((int(lo), int(hi)) for lo, hi in padding)


def conv_out_spatial(in_spatial, k_spatial, strides, pads, lhs_dilation=None,
                     rhs_dilation=None):
  n = len(in_spatial)
  lhs_dilation = lhs_dilation or (1,) * n
  rhs_dilation = rhs_dilation or (1,) * n
  out = []
  for i in range(n):
    in_d = (in_spatial[i] - 1) * lhs_dilation[i] + 1 if in_spatial[i] > 0 else 0
    k_eff = (k_spatial[i] - 1) * rhs_dilation[i] + 1
    tot = in_d + pads[i][0] + pads[i][1]
    out.append(0 if tot < k_eff else (tot - k_eff) // strides[i] + 1)
  return tuple(out)


def im2col(x, k_spatial, strides, pads, lhs_dilation=None, rhs_dilation=None):
  """Channels-last [B, *spatial, C] (1, 2 or 3 spatial axes) -> [B, *out_spatial, prod(k) * Cin]
  in (k..., cin) order -- the flattening of the reference's kernels (HWIO / DHWIO,
  flax_qconv.py:120-123), which is the order of the `fseq` chain."""
  import itertools
  x = np.asarray(x)
  n = x.ndim - 2
  lhs_dilation = tuple(lhs_dilation or (1,) * n)
  rhs_dilation = tuple(rhs_dilation or (1,) * n)
  assert 1 <= n <= 3, "oracle supports 1-D, 2-D and 3-D convolutions"
  B, C = x.shape[0], x.shape[-1]
  sp = x.shape[1:-1]
  if any(d != 1 for d in lhs_dilation):
    dsp = tuple((sp[i] - 1) * lhs_dilation[i] + 1 if sp[i] > 0 else 0 for i in range(n))
    xd = np.zeros((B,) + dsp + (C,), dtype=x.dtype)
    xd[(slice(None),) + tuple(slice(None, None, lhs_dilation[i]) for i in range(n))] = x
    x, sp = xd, dsp
  xp = np.pad(x, ((0, 0),) + tuple(tuple(p) for p in pads) + ((0, 0),))
  out = conv_out_spatial(sp, k_spatial, strides, pads, None, rhs_dilation)
  cols = np.empty((B,) + tuple(out) + tuple(k_spatial) + (C,), dtype=x.dtype)
  for kpos in itertools.product(*[range(k) for k in k_spatial]):
    src = (slice(None),) + tuple(
        slice(kpos[i] * rhs_dilation[i], kpos[i] * rhs_dilation[i] + (out[i] - 1) * strides[i] + 1, strides[i])
        for i in range(n)) + (slice(None),)
    cols[(slice(None),) * (1 + n) + tuple(kpos) + (slice(None),)] = xp[src]
  K = C
  for k in k_spatial:
    K *= k
  return cols.reshape((B,) + tuple(out) + (K,))


def quant_conv(x, qw: QWeight, strides=None, padding="SAME", input_dilation=None,
               kernel_dilation=None, feature_group_count=1, mode="int",
               return_acc=False):
  """QuantConv.__call__ without bias, flax_qconv.py:93-171.

  x channels-last ([B, W, Cin], [B, H, W, Cin] or [B, D, H, W, Cin]); kernel HWIO / DHWIO in qw."""
  x = np.asarray(x)
  kshape = qw.kernel.shape
  nsp = len(kshape) - 2
  single = False
  if x.ndim == nsp + 1:                              # flax_qconv.py:109-112
    single, x = True, x[None]
  k_spatial = tuple(kshape[:nsp])
  strides = tuple(strides or (1,) * nsp)
  cin_g, cout = kshape[-2], kshape[-1]
  G = feature_group_count
  assert x.shape[-1] % G == 0                        # flax_qconv.py:117
  assert x.shape[-1] // G == cin_g and cout % G == 0
  # flax_qconv.py:128-144 uses rhs_dilation = 1 to resolve string padding
  pads = resolve_padding(x.shape[1:-1], k_spatial, strides, padding, None)
  outs = []
  for g in range(G):
    xg = x[..., g * cin_g:(g + 1) * cin_g]
    cols = im2col(xg, k_spatial, strides, pads, input_dilation, kernel_dilation)
    lead = cols.shape[:-1]
    c2 = cols.reshape(-1, cols.shape[-1])
    og = cout // G
    sl = slice(g * og, (g + 1) * og)
    if mode == "int":
      assert qw.quantised and _is_integer_valued(c2)
      acc = _exact_int_matmul(c2, qw.q.reshape(-1, cout)[:, sl])
      yg = acc if return_acc else qw.dequant_acc(acc)
    elif mode == "fseq":
      yg = fseq_matmul(c2.astype(F32), qw.w_fq.reshape(-1, cout)[:, sl])
    elif mode == "float":
      yg = float_matmul(c2.astype(F32), np.ascontiguousarray(qw.w_fq.reshape(-1, cout)[:, sl]))
    else:
      raise ValueError(mode)
    outs.append(yg.reshape(lead + (og,)))
  y = np.concatenate(outs, axis=-1) if G > 1 else outs[0]
  if not (mode == "int" and return_acc):
    y = y.astype(F32)
  return y[0] if single else y


# ---------------------------------------------------------------------------
# Neurons, norm, SpikingBlock   (spiking_learning.py)
# ---------------------------------------------------------------------------


def gated_conv(s, gate, qw: QWeight, padding=((1, 1), (1, 1))):
  """The 'gint' contraction: QuantConv (flax_qconv.py:93-171, stride 1) on the product of a spike
  raster and a per-(image, channel) gate -- what a TCJA block hands to the next conv block,
  x = s * sigmoid(...)[:, :, None, None, :] (examples/tcja/models.py:95-97 -> :149-187).

  s [NB, H, W, C] in {0, 1}; gate [NB, C] float32 -> currents float32 [NB, OH, OW, Cout]:
    I[p, c, o]  = sum over the taps of code[tap, c, o] * s[pixel of the tap, c]    (exact integer)
    acc[p, o]   = fmaf(gate[c], I[p, c, o], acc[p, o])  for c = 0, 1, ..., C - 1     (start +0)
    current     = fl(fl(acc / L) * m)                                              (quant.py:443,467)
  The reference multiplies float32 fake-quantised weights with the float32 products g * s and
  sums in XLA's order; this contract factors the gate out of the nine taps of its channel -- the
  same real number, C roundings instead of 9 C -- so that the taps are summed as integers
  (oracle/int_vs_float.py reports its distance to the reference-literal float mode)."""
  assert qw.quantised
  s = np.asarray(s)
  gate = np.ascontiguousarray(gate, dtype=F32)
  NB, H, W, C = s.shape
  kh, kw, cin, cout = qw.q.shape
  assert cin == C and gate.shape == (NB, C) and _is_integer_valued(s)
  cols = im2col(s.astype(F32), (kh, kw), (1, 1), resolve_padding((H, W), (kh, kw), (1, 1), padding))
  OH, OW = cols.shape[1], cols.shape[2]
  cols = cols.reshape(NB * OH * OW, kh * kw, C)          # (kh, kw, cin) order of im2col
  codes = qw.q.reshape(kh * kw, C, cout).astype(F32)
  acc = np.zeros((NB * OH * OW, cout), F32)
  for c in range(C):
    I = np.ascontiguousarray(cols[:, :, c]) @ np.ascontiguousarray(codes[:, c, :])   # exact: |I| <= 9 * 127
    g = np.ascontiguousarray(np.repeat(gate[:, c], OH * OW))
    clib().oracle_fma_rows(_fp(acc), _fp(g), _fp(np.ascontiguousarray(I, dtype=F32)), acc.shape[0], cout)
  y = ((acc / qw.L) * qw.m).astype(F32)
  return y.reshape(NB, OH, OW, cout)


def gated_conv_block(spikes, gate, qw: QWeight, bn: Optional[dict], neuron_cfg=None, u0=None):
  """SpikingBlock(QuantConv 3x3 pad 1, BatchNorm, neuron) on gate[T, B, C] x spikes[T, B, H, W, C]
  in the 'gint' mode."""
  norm = None
  if bn is not None:
    norm = lambda x: batchnorm_eval(x, bn["mean"], bn["var"], bn.get("scale"),
                                    bn.get("bias"), bn.get("eps", 1e-5))
  T = spikes.shape[0]
  seq = [(spikes[t], gate[t]) for t in range(T)]
  neuron = _neuron(neuron_cfg or {})
  u = None
  out = []
  for st, gt in seq:
    x = gated_conv(st, gt, qw)
    if norm is not None:
      x = norm(x)
    if u is None:
      u = np.zeros_like(x) if u0 is None else np.asarray(u0, F32)
    u, sp = neuron(u, x)
    out.append(sp)
  return u, np.stack(out)


def gated_dense(s, gate, qw: QWeight):
  """The 'gint' contraction of a dense layer on the CHANNEL-MAJOR flattening of gate x raster
  (examples/tcja/models.py:97 -> :189-190 -> :200-216: the first dense block of CextNet sees
  x[k] = gate[c] * s[c, h, w] with k = (c * H + h) * W + w).

  s [NB, H, W, C] in {0, 1}; gate [NB, C] float32; qw.q [C * H * W, N] -> currents [NB, N]:
    I[c, o] = sum over (h, w) of code[(c, h, w), o] * s[h, w, c]                  (exact integer)
    acc[o]  = fmaf(gate[c], I[c, o], acc[o])   for c = 0 .. C - 1                 (start +0)
    current = fl(fl(acc / L) * m)"""
  assert qw.quantised
  s = np.asarray(s)
  gate = np.ascontiguousarray(gate, dtype=F32)
  NB, H, W, C = s.shape
  K, N = qw.q.shape
  assert K == C * H * W and gate.shape == (NB, C) and _is_integer_valued(s)
  codes = qw.q.reshape(C, H * W, N).astype(F32)
  sp = np.transpose(s.reshape(NB, H * W, C), (0, 2, 1)).astype(F32)       # [NB, C, HW]
  acc = np.zeros((NB, N), F32)
  for c in range(C):
    I = np.ascontiguousarray(sp[:, c, :]) @ np.ascontiguousarray(codes[c])   # exact: |I| <= HW * 127
    clib().oracle_fma_rows(_fp(acc), _fp(np.ascontiguousarray(gate[:, c])), _fp(np.ascontiguousarray(I, dtype=F32)),
                           NB, N)
  return ((acc / qw.L) * qw.m).astype(F32)


def gated_dense_block(spikes, gate, qw: QWeight, neuron_cfg=None, u0=None):
  """SpikingBlock(QuantDense, neuron) on the channel-major flattening of gate[T, B, C] x
  spikes[T, B, H, W, C], 'gint' mode."""
  neuron = _neuron(neuron_cfg or {})
  u = None
  out = []
  for t in range(spikes.shape[0]):
    x = gated_dense(spikes[t], gate[t], qw)
    if u is None:
      u = np.zeros_like(x) if u0 is None else np.asarray(u0, F32)
    u, sp = neuron(u, x)
    out.append(sp)
  return u, np.stack(out)


def heaviside(x):
  """Forward of atan / fast_sigmoid / ... (spiking_learning.py:139-241): x >= 0."""
  return (np.asarray(x, dtype=F32) >= F32(0)).astype(F32)


# Two float32 library functions of the reference's path live in jax / XLA, not under
# /root/reference, and their last bit is not pinned by anything here: the logistic of the TCJA gate
# and of the PLIF / LIF decay (jax.nn.sigmoid -> 1 / (1 + exp(-x)) in float32, XLA's expf), and the
# reciprocal square root of BatchNorm (flax 0.4.0 _normalize: lax.rsqrt(var + eps)).  The oracle's
# choice for each is below; oracle/int_vs_float.py (--choices) installs the other plausible
# evaluations through these two hooks and counts what they change downstream.
SIGMOID = None       # None: float64 expit rounded once to float32
RSQRT = None         # None: fl(1 / fl(sqrt(v))), two float32 roundings
# ... and a third: the ORDER of BatchNorm's three operations.  flax 0.4.0's _normalize is recalled
# (the file is not under /root/reference) as y = (x - mean) * mul + bias; later
# flax versions fold the mean into the bias, y = x * mul + (bias - mean * mul).
BN_FOLDED = False    # False: fl(fl(fl(x - mean) * mul) + bias)
# ... and a fourth: CONTRACTION.  XLA's CPU backend lets LLVM fuse a float32 multiply and the add
# behind it into one fused multiply-add where the host has the instruction (one rounding instead
# of two).  Where that can change a value on this path: BatchNorm's `y * mul + bias`, PLIF's
# `u + d * k` and LIF's `u * k + s_in` (multi_step_LIF with tau = 2 divides by a power of two: the
# product is exact either way).  The oracle rounds every operation by itself, as the source reads.
FMA_CONTRACT = False


def _fma32(a, b, c):
  """fl(a * b + c) for float32 arrays: the product of two float32 values is exact in float64, the sum is
  rounded to float64 and then to float32 (a double rounding that differs from a true fused
  multiply-add about once in 2^29 operations: good enough to COUNT what contraction changes)."""
  return (np.asarray(a, F32).astype(np.float64) * np.asarray(b, F32).astype(np.float64)
          + np.asarray(c, F32).astype(np.float64)).astype(F32)


def sigmoid_f32(x):
  """jax.nn.sigmoid on float32 (TCJA gate models.py:95, PLIF / LIF decay
  spiking_learning.py:381,432): float64 expit rounded once, unless SIGMOID is installed."""
  if SIGMOID is not None:
    return np.asarray(SIGMOID(np.asarray(x, dtype=F32)), dtype=F32)
  return (1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))).astype(F32)


def multi_step_lif(u, s_in, tau=2.0, v_threshold=1.0, v_reset=0.0):
  """multi_step_LIF.__call__, spiking_learning.py:403-416."""
  u = np.asarray(u, dtype=F32)
  s_in = np.asarray(s_in, dtype=F32)
  tau, vth, vr = F32(tau), F32(v_threshold), F32(v_reset)
  u = u + (s_in - (u - vr)) / tau          # :410
  s = heaviside(u - vth)                   # :412
  u = np.where(s != 0, vr, u)              # :414
  return u.astype(F32), s


def parametric_leaky_if(u, s_in, tau_param, v_threshold=1.0, v_reset=0.0):
  """parametric_leaky_IF.__call__, spiking_learning.py:370-387 (tau_param shape (1,))."""
  u = np.asarray(u, dtype=F32)
  s_in = np.asarray(s_in, dtype=F32)
  k = sigmoid_f32(np.asarray(tau_param).reshape(-1)[0])
  vth, vr = F32(v_threshold), F32(v_reset)
  if FMA_CONTRACT:
    u = _fma32((s_in - (u - vr)).astype(F32), k, u)
  else:
    u = u + (s_in - (u - vr)) * k          # :381
  s = heaviside(u - vth)
  u = np.where(s != 0, vr, u)
  return u.astype(F32), s


def lif(u, s_in, tau_vec, v_threshold=1.0, v_reset=0.0):
  """LIF.__call__, spiking_learning.py:426-438 (tau_vec shape (N,))."""
  u = np.asarray(u, dtype=F32)
  s_in = np.asarray(s_in, dtype=F32)
  k = sigmoid_f32(tau_vec)
  vth, vr = F32(v_threshold), F32(v_reset)
  u = _fma32(u, k, s_in) if FMA_CONTRACT else u * k + s_in      # :432
  s = heaviside(u - vth)
  u = np.where(s > F32(0.5), vr, u)        # :436
  return u.astype(F32), s


def bn_coeffs(mean, var, scale=None, bias=None, eps=1e-5):
  """Eval-mode flax 0.4.0 BatchNorm folded to (mean, mul, bias):
  mul = fl(fl(1 / sqrt(var + eps)) * scale)  (models.py:101-107)."""
  mean = np.asarray(mean, dtype=F32)
  var = np.asarray(var, dtype=F32)
  v = (var + F32(eps)).astype(F32)
  mul = F32(1) / np.sqrt(v) if RSQRT is None else np.asarray(RSQRT(v), dtype=F32)
  if scale is not None:
    mul = mul * np.asarray(scale, dtype=F32)
  b = np.zeros_like(mean) if bias is None else np.asarray(bias, dtype=F32)
  return mean, mul.astype(F32), b


def batchnorm_eval(x, mean, var, scale=None, bias=None, eps=1e-5):
  """y = fl(fl(fl(x - mean) * mul) + bias) over the last axis."""
  mean, mul, b = bn_coeffs(mean, var, scale, bias, eps)
  if BN_FOLDED:
    return (np.asarray(x, dtype=F32) * mul + (b - mean * mul).astype(F32)).astype(F32)
  if FMA_CONTRACT:
    return _fma32((np.asarray(x, dtype=F32) - mean).astype(F32), mul, b)
  y = (np.asarray(x, dtype=F32) - mean) * mul
  return (y + b).astype(F32)


def spiking_block(u0, inputs, connection_fn: Callable, neuron_fn: Callable,
                  norm_fn: Optional[Callable] = None):
  """SpikingBlock.__call__, spiking_learning.py:446-462: scan over axis 0."""
  u = None if u0 is None else np.asarray(u0, dtype=F32)
  out = []
  for t in range(inputs.shape[0]):
    x = connection_fn(inputs[t])
    if norm_fn is not None:
      x = norm_fn(x)
    if u is None:                         # initialize_carry, :464-472
      u = np.zeros_like(x, dtype=F32)
    u, s = neuron_fn(u, x)
    out.append(s)
  return u, np.stack(out, axis=0)


def max_pool_2x2(x):
  """reduce_window max (1,1,2,2,1), models.py:145-147, on [T, B, H, W, C]."""
  T, B, H, W, C = x.shape
  x = x[:, :, :H // 2 * 2, :W // 2 * 2]
  return x.reshape(T, B, H // 2, 2, W // 2, 2, C).max(axis=(3, 5))


def flatten_channel_major(x):
  """models.py:189-190: transpose (T,B,C,H,W) then reshape [T, B, C*H*W]."""
  x = np.transpose(x, (0, 1, 4, 2, 3))
  return x.reshape(x.shape[:2] + (-1,))


def vote(spikes, group=10):
  """models.py:253-255: mean over T, then mean over groups of `group`.

  Both means are sequential float32 sums divided by the count."""
  s = np.asarray(spikes, dtype=F32)
  T = s.shape[0]
  acc = np.zeros(s.shape[1:], dtype=F32)
  for t in range(T):
    acc = acc + s[t]
  r = acc / F32(T)
  r = r.reshape(r.shape[0], -1, group)
  acc2 = np.zeros(r.shape[:2], dtype=F32)
  for j in range(group):
    acc2 = acc2 + r[:, :, j]
  return (acc2 / F32(group)).astype(F32)


def onehot(labels, num_classes):
  return (np.asarray(labels)[:, None] == np.arange(num_classes)[None]).astype(F32)


def mse_loss(logits, labels, smoothing=0.0, T=1):
  """examples/train_utils.py:210-217."""
  oh = onehot(labels, logits.shape[1])
  oh = oh * F32(1 - smoothing) + F32(smoothing / oh.shape[1])
  return F32(np.mean(np.square(np.asarray(logits, dtype=F32) / F32(T) - oh),
                     dtype=F32))


def compute_metrics(logits, labels, smoothing=0.0):
  """examples/train_utils.py:220-225."""
  return {"loss": mse_loss(logits, labels, smoothing),
          "accuracy": np.argmax(logits, -1) == np.asarray(labels)}


# ---------------------------------------------------------------------------
# Model-level restatements for the BASELINE.json configs
# ---------------------------------------------------------------------------


def _neuron(cfg):
  kind = cfg.get("kind", "multi_step_LIF")
  vth, vr = cfg.get("v_threshold", 1.0), cfg.get("v_reset", 0.0)
  if kind == "multi_step_LIF":
    return lambda u, x: multi_step_lif(u, x, cfg.get("tau", 2.0), vth, vr)
  if kind == "parametric_leaky_IF":
    return lambda u, x: parametric_leaky_if(u, x, cfg["tau_param"], vth, vr)
  if kind == "LIF":
    return lambda u, x: lif(u, x, cfg["tau_vec"], vth, vr)
  raise ValueError(kind)


def dense_block(inputs, qw: QWeight, neuron_cfg=None, mode="int", u0=None):
  """SpikingBlock(QuantDense, neuron) on [T, B, K] -> (u_T, spikes [T, B, N])."""
  return spiking_block(u0, inputs, lambda x: quant_dense(x, qw, mode),
                       _neuron(neuron_cfg or {}))


def conv_block(inputs, qw: QWeight, bn: Optional[dict], neuron_cfg=None,
               mode="int", padding=((1, 1), (1, 1)), strides=None, u0=None):
  """SpikingBlock(QuantConv, neuron, BatchNorm) on [T, B, H, W, C]."""
  norm = None
  if bn is not None:
    norm = lambda x: batchnorm_eval(x, bn["mean"], bn["var"], bn.get("scale"),
                                    bn.get("bias"), bn.get("eps", 1e-5))
  return spiking_block(
      u0, inputs,
      lambda x: quant_conv(x, qw, strides=strides, padding=padding, mode=mode),
      _neuron(neuron_cfg or {}), norm)


def dense2_forward(inputs, qw1: QWeight, qw2: QWeight, neuron_cfg=None,
                   mode="int", group=10):
  """Configs C1/C2: [T, B, 2048] -> QuantDense(512)+LIF -> QuantDense(110)+LIF -> vote.

  Head of CextNet, models.py:200-255.  Returns dict of intermediates."""
  u1, s1 = dense_block(inputs, qw1, neuron_cfg, mode)
  u2, s2 = dense_block(s1, qw2, neuron_cfg, mode)
  return {"u1": u1, "s1": s1, "u2": u2, "s2": s2, "logits": vote(s2, group)}


def conv3_dense_forward(inputs, conv_qw: Sequence[QWeight], bns: Sequence[dict],
                        dense_qw: QWeight, neuron_cfg=None, mode="int",
                        group=10, keep=False):
  """Config C3: three (QuantConv3x3 + BN + LIF + 2x2 max-pool) blocks
  (models.py:109-147), channel-major flatten (:189-190), QuantDense + LIF
  (:231-246), vote (:253-255).  inputs [B, T, H, W, Cin]."""
  x = np.swapaxes(np.asarray(inputs), 0, 1)          # models.py:109
  out = {}
  for i, (qw, bn) in enumerate(zip(conv_qw, bns)):
    u, s = conv_block(x, qw, bn, neuron_cfg, mode)
    x = max_pool_2x2(s)
    if keep:
      out["conv%d_u" % i], out["conv%d_s" % i] = u, s
    out["pool%d" % i] = x
  xf = flatten_channel_major(x)
  u, s = dense_block(xf, dense_qw, neuron_cfg, mode)
  out["dense_u"], out["dense_s"] = u, s
  out["logits"] = vote(s, group)
  return out


def events_to_frames(x, y, p, T, H, W, scale=1.0):
  """preprocess_data_number, examples/input_pipeline.py:142-219 (split_by 'number'):
  equal-count slices of the time-ordered events, bincount per pixel and polarity."""
  x, y, p = np.asarray(x), np.asarray(y), np.asarray(p)
  n = x.shape[0]
  di = n // T
  frames = np.zeros((T, 2, H * W), np.int32)
  for i in range(T):
    lo = i * di
    hi = lo + di if i < T - 1 else n
    xs = np.floor(x[lo:hi].astype(F32) / F32(scale)).astype(np.int64)
    ys = np.floor(y[lo:hi].astype(F32) / F32(scale)).astype(np.int64)
    ps = p[lo:hi]
    for j, mask in enumerate((ps == 0, ps != 0)):
      pos = ys[mask] * W + xs[mask]
      frames[i, j] += np.bincount(pos, minlength=H * W)[:H * W].astype(np.int32)
  return np.transpose(frames.reshape(T, 2, H, W), (0, 2, 3, 1))


def density(x, lead_dims=2):
  """sparse_nums of examples/tcja/models.py:128-131."""
  x = np.asarray(x)
  lead = x.shape[:lead_dims]
  flat = x.reshape(lead + (-1,))
  return (np.sum(flat != 0, axis=-1) / F32(flat.shape[-1])).astype(F32)


def tcja(x_seq, qw_t: QWeight, qw_c: QWeight, probes: Optional[dict] = None, tag: str = ""):
  """TCJA gate, examples/tcja/models.py:41-99, on x_seq [T, B, H, W, C].

  Channel means are sequential float32 sums / (H*W); the two 1-D QuantConvs
  (k = 4, SAME) run in 'fseq' mode on their real-valued inputs; the logistic is
  evaluated in float64 on the float32 product and rounded once."""
  x_seq = np.asarray(x_seq, dtype=F32)
  T, B, H, W, C = x_seq.shape
  acc = np.zeros((T, B, C), F32)
  flat = x_seq.reshape(T, B, H * W, C)
  for p in range(H * W):
    acc = acc + flat[:, :, p]
  m = acc / F32(H * W)                                   # [T, B, C]
  x = np.ascontiguousarray(np.transpose(m, (1, 0, 2)))   # [B, T, C]
  x_c = np.ascontiguousarray(np.transpose(x, (0, 2, 1)))  # [B, C, T]
  conv_t = quant_conv(x_c, qw_t, None, "SAME", mode="fseq")   # [B, C, T]
  conv_c = quant_conv(x, qw_c, None, "SAME", mode="fseq")     # [B, T, C]
  if probes is not None:                                 # models.py:45-91, per sample
    probes["conv_tcja1_%s_inpt" % tag] = density(x_c, 1)
    probes["conv_tcja1_%s_out" % tag] = density(conv_t, 1)
    probes["conv_tcja2_%s_inpt" % tag] = density(x, 1)
    probes["conv_tcja2_%s_out" % tag] = density(conv_c, 1)
  conv_t = np.transpose(conv_t, (2, 0, 1))               # [T, B, C]
  conv_c = np.transpose(conv_c, (1, 0, 2))               # [T, B, C]
  z = (conv_c * conv_t).astype(F32)
  gate = sigmoid_f32(z)                                  # models.py:95
  return (x_seq * gate[:, :, None, None, :]).astype(F32), gate


def cextnet_forward(inputs, conv_qw: Sequence[QWeight], bns: Sequence[dict],
                    tcja_qw: Sequence[Tuple[QWeight, QWeight]], dense_qw: Sequence[QWeight],
                    neuron_cfg=None, group=10, probes: Optional[dict] = None, gated="gint"):
  """Full CextNet (models.py:31-257), eval.  conv_qw: the five 3x3 kernels;
  tcja_qw: [(conv_t, conv_c)] x 2; dense_qw: the two dense kernels.  Integer
  mode while activations are spikes, 'fseq' once they are real-valued.
  probes: dict that receives the per-slice densities (`sparse_nums`) the model sows
  (models.py:128-142 and the like), keyed `<name>_inpt` / `<name>_out`.
  gated: how the conv block behind the first TCJA gate contracts its gate x raster input --
  "gint" (gated_conv: integer sums per channel, one float32 chain over the gates; the product's
  contract since round 5) or "fseq" (the fmaf chain over (kh, kw, cin) of the float32 product)."""
  x = np.swapaxes(np.asarray(inputs), 0, 1)
  out = {}

  def probe(name, v, lead=2):
    if probes is not None:
      probes[name] = density(v, lead)
  for i in range(3):
    probe("conv_%d_inpt" % i, x)
    _, s = conv_block(x, conv_qw[i], bns[i], neuron_cfg, "int")
    probe("conv_%d_out" % i, s)
    x = max_pool_2x2(s)
    out["pool%d" % i] = x
  mode = "int"
  pooled_s = gate = None
  for i in range(2):
    probe("conv_t_%d_inpt" % i, x)
    if mode == "fseq" and gated == "gint" and conv_qw[3 + i].quantised:
      _, s = gated_conv_block(pooled_s, gate, conv_qw[3 + i], bns[3 + i], neuron_cfg)
    else:
      _, s = conv_block(x, conv_qw[3 + i], bns[3 + i], neuron_cfg, mode)
    probe("conv_t_%d_out" % i, s)
    out["conv_t_%d" % i] = s
    y, gate = tcja(s, *tcja_qw[i], probes=probes, tag=str(i))
    out["gate%d" % i] = gate
    x = max_pool_2x2(y)
    pooled_s = max_pool_2x2(s)           # max(g * s_i) = g * max(s_i): g > 0, s in {0, 1}
    mode = "fseq"
  xf = flatten_channel_major(x)
  probe("dense1_inpt", xf)
  if gated == "gint" and dense_qw[0].quantised and pooled_s.shape[2] * pooled_s.shape[3] <= 32:
    _, s1 = gated_dense_block(pooled_s, gate, dense_qw[0], neuron_cfg)
  else:
    _, s1 = dense_block(xf, dense_qw[0], neuron_cfg, "fseq")
  probe("dense1_out", s1)
  probe("dense2_inpt", s1)
  _, s2 = dense_block(s1, dense_qw[1], neuron_cfg, "int")
  probe("dense2_out", s2)
  out["dense1_s"], out["dense2_s"] = s1, s2
  out["logits"] = vote(s2, group)
  return out


# ---------------------------------------------------------------------------
# Synthetic inputs shared by tests and bench
# ---------------------------------------------------------------------------

WEIGHT_SEED = 203853699      # examples/tcja/configs/prune_quant_joint.py:21
DATA_SEED = 8627169          # quant_test.py:149


def poisson_spikes(shape, lam=0.1, seed=DATA_SEED):
  rng = np.random.Generator(np.random.PCG64(seed))
  return (rng.poisson(lam, size=shape) > 0).astype(np.uint8)


def synth_kernel(shape, gain=1.0, seed=WEIGHT_SEED):
  """N(0, 1/fan_in) * gain (stand-in for lecun_normal, flax_qdense.py:26)."""
  rng = np.random.Generator(np.random.PCG64(seed))
  fan_in = int(np.prod(shape[:-1]))
  return (rng.standard_normal(shape) * (gain / np.sqrt(fan_in))).astype(F32)
