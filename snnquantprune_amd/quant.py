"""Quantisers, pruning mask and calibration functions -- forward pass only.

Host-side mirror of the reference's ``quant.py`` (same class names, fields and
variable names); the arithmetic runs in ``snnqp_quantize`` (csrc/quantize.hip).
Besides the reference's ``Q(bits=, g_scale=)(kernel) -> float tensor`` call,
each quantiser can *describe* itself (`describe(kernel)`), which is what lets
QuantDense / QuantConv load packed integer codes instead of fake-quantised
floats (BASELINE.json: "quant.py fake-quant -> packed int load").

Out of scope (training only): the custom-VJP backward rules of the round_*
functions (quant.py:35-288), gradscale (quant.py:404-418), get_noise (:19-23).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Callable, Optional

import torch

from . import _lib as L
from . import linen as nn
from . import ops
from ._cache import TensorCache

Array = Any


# ---------------------------------------------------------------------------
# round_* : every variant has the same forward, jnp.round (quant.py:26-288).
# They only select a backward rule in the reference; kept as config tokens.
# ---------------------------------------------------------------------------


def _make_round(name):
  def fn(x, scale=0.0, off=False, *unused):
    return x if off else torch.round(x)     # torch.round = round half to even
  fn.__name__ = name
  return fn


round_ste = _make_round("round_ste")
round_gaussian_noise = _make_round("round_gaussian_noise")
round_uniform_noise = _make_round("round_uniform_noise")
round_ewgs = _make_round("round_ewgs")
round_acos = _make_round("round_acos")
round_tanh = _make_round("round_tanh")
round_invtanh = _make_round("round_invtanh")
round_psgd = _make_round("round_psgd")
round_fsig = _make_round("round_fsig")
round_gaussian = _make_round("round_gaussian")
round_multi_gaussian = _make_round("round_multi_gaussian")


# ---------------------------------------------------------------------------
# Calibration functions (quant.py:296-314) -- pack-time, once per model.  The value
# sits inside round(clip(W / a) * L): an `a` one ulp off flips codes at ties, so the
# statistics are evaluated in ONE defined way -- float32 NumPy reductions on the host
# (pairwise sums), the same ops the CPU oracle runs -- rather than with whatever
# reduction order the device library picks for the tensor's size.
# ---------------------------------------------------------------------------


def max_init(x, bits, sign, axis=None):
  x = torch.as_tensor(x, dtype=torch.float32)
  if axis is not None:
    raise NotImplementedError("per-axis calibration")
  if float(x.max()) == 0:
    return torch.tensor(1 / 2 ** bits, dtype=torch.float32, device=x.device)
  return x.abs().max()


def gaussian_init(x, bits, sign, axis=None):
  x = torch.as_tensor(x, dtype=torch.float32)
  if axis is not None:
    raise NotImplementedError("per-axis calibration")
  import numpy as np
  xn = x.detach().cpu().numpy()
  if np.max(xn) == 0:
    return torch.tensor(1 / 2 ** bits, dtype=torch.float32, device=x.device)
  mu = np.mean(xn, dtype=np.float32)
  sigma = np.std(xn, dtype=np.float32)
  three = np.float32(3)
  v = np.float32(np.maximum(np.abs(mu - three * sigma), np.abs(mu + three * sigma)))
  return torch.tensor(float(v), dtype=torch.float32, device=x.device)


def percentile_init(x, bits, sign, perc, axis=None):
  x = torch.as_tensor(x, dtype=torch.float32)
  if axis is not None:
    raise NotImplementedError("per-axis calibration")
  if float(x.max()) == 0:
    return torch.tensor(1 / 2 ** bits, dtype=torch.float32, device=x.device)
  import numpy as np
  return torch.tensor(float(np.percentile(x.abs().cpu().numpy(), perc)),
                      dtype=torch.float32, device=x.device)


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------

_scalar_cache = TensorCache(4096)


def host_scalar(t) -> float:
  """float(t) for a 1-element parameter, cached per tensor version (the one
  device->host readback of the pack step)."""
  if not isinstance(t, torch.Tensor):
    return float(t)
  v = _scalar_cache.get((t,))
  if v is None:
    v = _scalar_cache.put((t,), None, float(t.reshape(-1)[0].item()))
  return v


@dataclass(frozen=True)
class QuantDesc:
  """What snnqp_quantize_ex needs, plus the dequantisation y = fl(fl(acc/L) * m).  `sign` is the
  reference's call argument (quant.py:331): False = the unsigned levels 0 .. 2^bits - 1."""
  kind: int
  bits: int
  p0: float
  p1: float
  L: float
  m: float
  sign: bool = True


def _f32(x: float) -> float:
  import numpy as np
  return float(np.float32(x))          # rounds a Python float to float32 (host only)


def _levels(bits: int, sign: bool) -> int:
  """num_levels / q_pos of quant.py:338-341, :378-384, :532-535."""
  return 2 ** (bits - 1) - 1 if sign else 2 ** bits - 1


def _apply(desc: QuantDesc, x):
  x = torch.as_tensor(x)
  fq, _, _ = ops.quantize(desc.kind, x, None, desc.bits, desc.p0, desc.p1,
                          want_fq=True, want_codes=False, sign=desc.sign)
  return fq.reshape(x.shape)


# ---------------------------------------------------------------------------
# Quantisers
# ---------------------------------------------------------------------------


class uniform_static(nn.Module):
  """quant.py:322-358."""
  bits: int = 8
  act: bool = False
  round_fn: Callable = round_psgd
  init_fn: Callable = max_init
  g_scale: float = 0.
  maxabs_w: float = None

  @nn.compact_method
  def describe(self, x, sign: bool = True) -> Optional[QuantDesc]:
    if type(self.bits) == int:
      assert self.bits > 1, (
          "Bit widths below 2 bits are not supported but got bits: " + str(self.bits))
    num_levels = _levels(self.bits, sign)
    xmax = self.variable("quant_params", "dynamic_range_no_train",
                         lambda: torch.ones((1,), device=torch.as_tensor(x).device))
    if self.is_mutable_collection("quant_params"):
      v = self.init_fn(x, bits=self.bits, sign=sign).reshape(1).to(torch.float32)
      xmax.value = torch.where(v == 0, torch.ones_like(v), v)
    xm = host_scalar(xmax.value)
    return QuantDesc(L.Q_UNIFORM_STATIC, self.bits, xm, 0.0, 1.0,
                     _f32(_f32(xm) / float(num_levels)), bool(sign))

  def __call__(self, x: Array, sign: bool = True) -> Array:
    return _apply(self.describe(x, sign), x)


class parametric_d(nn.Module):
  """quant.py:361-425 (LSQ-style step size), forward."""
  bits: int = 8
  act: bool = False
  round_fn: Callable = round_psgd
  init_fn: Callable = max_init
  g_scale: float = 0.
  clip_quant_grads: bool = True
  maxabs_w: float = None

  @nn.compact_method
  def describe(self, inputs, sign: bool = True) -> Optional[QuantDesc]:
    q_pos = _levels(self.bits, sign)
    dev = torch.as_tensor(inputs).device
    step = self.variable("quant_params", "step_size",
                         lambda: torch.ones((1,), device=dev))
    if self.is_mutable_collection("quant_params"):
      v = self.init_fn(inputs, bits=self.bits, sign=sign).reshape(1).to(torch.float32)
      # init / sqrt(q_pos) as ONE float32 division (quant.py:397-399); a tensor divided by a Python
      # scalar on the device is a multiplication by the reciprocal, one ulp off
      import numpy as np
      sv = np.float32(host_scalar(v)) / np.sqrt(np.float32(q_pos))
      step.value = torch.full((1,), float(sv), device=v.device)
    s = host_scalar(step.value)
    return QuantDesc(L.Q_PARAMETRIC_D, self.bits, s, 0.0, 1.0, _f32(s), bool(sign))

  def __call__(self, inputs: Array, sign: bool = True) -> Array:
    return _apply(self.describe(inputs, sign), inputs)


class DuQ(nn.Module):
  """Differentiable and unified Quantization, quant.py:428-469, forward.

  Params `a`, `c` of shape (1,), init -1; `a == -1` and `bits == -1` pass the
  input through unchanged (:453-454, :469)."""
  bits: int = 4
  act: bool = False
  g_scale: float = 0.
  round_fn: Callable = round_ste
  maxabs_w: float = None

  @nn.compact_method
  def describe(self, inputs, sign: bool = True) -> Optional[QuantDesc]:
    if self.bits == -1:
      return None
    if self.bits < 2:
      raise ValueError("DuQ needs bits >= 2 (n_lv - 1 = 0 divides by zero)")
    a = self.param("a", nn.constant(-1), (1,))
    c = self.param("c", nn.constant(-1), (1,))
    a_h, c_h = host_scalar(a), host_scalar(c)
    if a_h == -1.0:
      return None
    n_lv = 2 ** (self.bits - 1) if sign else 2 ** self.bits        # quant.py:458-461
    return QuantDesc(L.Q_DUQ, self.bits, a_h, c_h, float(n_lv - 1), _f32(c_h), bool(sign))

  def __call__(self, inputs: Array, sign: bool = True) -> Array:
    desc = self.describe(inputs, sign)
    if desc is None:
      return inputs
    return _apply(desc, inputs)


class prune(nn.Module):
  """quant.py:472-491: multiplies by the `mask` parameter (init ones)."""

  @nn.compact_method
  def get_mask(self, shape, device=None):
    return self.param("mask", nn.constant(1), tuple(shape))

  def __call__(self, inputs: Array, sign: bool = True) -> Array:
    mask = self.get_mask(inputs.shape)
    x = torch.as_tensor(inputs)
    return x * mask.to(x.device, x.dtype)


class parametric_d_xmax(nn.Module):
  """Parametric heterogeneous quantisation, quant.py:494-625, forward."""
  bits: int = 4
  act: bool = False
  xmax_min: float = 2 ** -8
  xmax_max: float = 127
  d_min: float = 2 ** -12
  d_max: float = 1
  round_fn: Callable = round_ste
  init_fn: Callable = None
  g_scale: float = 0.
  ceil_tolerance: float = 0.0
  maxabs_w: float = None
  bitwidth_min: int = 2

  @nn.compact_method
  def describe(self, inputs, sign: bool = True) -> Optional[QuantDesc]:
    x = torch.as_tensor(inputs)
    dev = x.device
    num_levels = _levels(self.bits, sign)
    one = lambda v: torch.full((1,), float(v), device=dev)  # noqa: E731
    self.variable("quant_config", "max_xmax", one, self.xmax_max)
    self.variable("quant_config", "min_xmax", one, self.xmax_min)
    self.variable("quant_config", "max_d", one, self.d_max)
    self.variable("quant_config", "min_d", one, self.d_min)
    d = self.variable("quant_params", "step_size", one, 1.0)
    xmax = self.variable("quant_params", "dynamic_range", one, 1.0)
    act_mb = self.variable("act_size", "act_mb", one, 1.0)
    weight_mb = self.variable("weight_size", "weight_mb", one, 1.0)
    bw = self.bits
    if self.is_mutable_collection("quant_params"):
      if self.init_fn is None:               # quant.py:553-565
        if self.act:
          xmax.value = one(2 ** -3 * (2. ** bw - 1))
          d.value = one(2 ** -3)
        else:
          maxabs_w = float(self.maxabs_w) if self.maxabs_w is not None else float(
              x.abs().max())
          lg = math.log2(_f32(maxabs_w / (2 ** (bw - 1) - 1)))
          dv = 2.0 ** (math.ceil(lg) if bw > 4 else math.floor(lg))
          d.value = one(dv)
          xmax.value = one(dv * (2 ** (bw - 1) - 1))
      else:                                  # quant.py:566-570
        v = self.init_fn(x, bits=self.bits, sign=sign).reshape(1).to(torch.float32)
        v = torch.where(v == 0, torch.ones_like(v), v)
        xmax.value = v
        import numpy as np
        d.value = one(float(np.float32(host_scalar(v)) / np.float32(num_levels)))   # one float32 division (:570)
    d_h = min(max(host_scalar(d.value), float(self.d_min)), float(self.d_max))
    xmax_h = min(max(host_scalar(xmax.value), float(self.xmax_min)),
                 float(self.xmax_max))
    d_h, xmax_h = _f32(d_h), _f32(xmax_h)
    # size bookkeeping, quant.py:585-614
    real_xmax = round(xmax_h / d_h) * d_h
    n_wf = 1
    for s in (x.shape[1:] if self.act else x.shape):
      n_wf *= int(s)
    nbits = math.ceil(math.log2(real_xmax / d_h + 1) - self.ceil_tolerance)
    nbits = max(nbits + (1 if sign else 0), self.bitwidth_min)
    if self.is_mutable_collection("act_size"):
      act_mb.value = one(n_wf * nbits if self.act else 0.0)
    if self.is_mutable_collection("weight_size"):
      weight_mb.value = one(0.0 if self.act else n_wf * nbits)
    return QuantDesc(L.Q_PARAMETRIC_D_XMAX, self.bits, d_h, xmax_h, 1.0, d_h, bool(sign))

  def __call__(self, inputs: Array, sign: bool = True) -> Array:
    return _apply(self.describe(inputs, sign), inputs)
