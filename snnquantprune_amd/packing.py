"""The pack step: reference parameter leaves -> what the kernels load.

`{'kernel', 'DuQ_0': {a, c}, 'prune_0': {mask}}` becomes either
int8 codes * mask with the dequantisation (L, m) -- the exact-integer path --
or the float32 fake-quantised * mask kernel (unquantised layers, real-valued
inputs).  This replaces the reference's per-timestep fake-quant inside the scan
body (spiking_learning.py:455 -> flax_qdense.py:74-85): it runs once per weight
version and is cached.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import ops
from ._cache import TensorCache
from .quant import QuantDesc

# When True a float32 activation tensor into a quantised layer takes the exact-integer kernels
# SPECULATIVELY -- the reference casts every input to float32 (flax_qdense.py:67,
# flax_qconv.py:101), so its event frames and spike rasters arrive as integer-valued float32
# tensors -- with the check on the device: the integer kernel (or the narrowing pass in front of
# it) verifies every value it reads, and a float32 launch enqueued right behind it redoes the block
# when one is not an integer in [0, 255] (ops.FloatFallback; snnqp.h, x_flags).  No inspection
# pass, no read-back, capturable.  False (or integer_inputs(False) around layers whose inputs
# are real-valued by construction): float32 tensors take the float32 kernels directly.
AUTO_INTEGER_INPUTS = True


class _Speculate:
  """prepare_input's answer for a float32 tensor that MAY be integer-valued: truthy."""

  def __bool__(self):
    return True

  def __repr__(self):
    return "SPECULATE"


SPECULATE = _Speculate()


import contextlib


@contextlib.contextmanager
def integer_inputs(enabled: bool):
  """Scoped override of AUTO_INTEGER_INPUTS (e.g. False around layers whose
  inputs are real-valued by construction, so the arithmetic path never depends
  on the data)."""
  global AUTO_INTEGER_INPUTS
  old, AUTO_INTEGER_INPUTS = AUTO_INTEGER_INPUTS, enabled
  try:
    yield
  finally:
    AUTO_INTEGER_INPUTS = old


def table_slots(ranges):
  """The event layer's per-channel dequantisation tables (snnqp.h ch_slots / ch_stack_max): channel
  c of a 128-channel group keeps a table over its own accumulator range (range_c = sum |code|) in
  one of the 32 LDS bank columns; the 32 channels of a wave (32 w .. 32 w + 31) must sit in 32
  different columns (a wave's lanes then never collide), four channels stack per column, and the
  tallest column sizes the table.  Greedy balance: wave by wave, the widest channel goes to the
  emptiest column.  Returns (int32 slots [padded Cout]: 4 * column + position in the column, the
  largest column sum of ranges)."""
  import numpy as np
  r = np.asarray(ranges, np.int64)
  n128 = (r.size + 127) // 128 * 128
  r = np.concatenate([r, np.zeros(n128 - r.size, np.int64)])
  slots = np.zeros(n128, np.int32)
  worst = 0
  for g in range(n128 // 128):
    col = np.zeros(32, np.int64)
    for w in range(4):
      ch = 128 * g + 32 * w + np.argsort(-r[128 * g + 32 * w:128 * g + 32 * w + 32], kind="stable")
      banks = np.argsort(col, kind="stable")
      slots[ch] = 4 * banks + w
      col[banks] += r[ch]
    worst = max(worst, int(col.max()))
  return slots, worst


class PackedKernel:
  def __init__(self, kernel: torch.Tensor, desc: Optional[QuantDesc],
               mask: Optional[torch.Tensor]):
    self.kernel = kernel
    self.desc = desc
    self.mask = mask
    self._int = None
    self._int_done = False
    self._float = None
    self._wt = {}

  def int_weight(self) -> Optional[ops.Weight]:
    """int8 codes (None if unquantised, > 8 bits, or the mask is not 0/1)."""
    if self._int_done:
      return self._int
    self._int_done = True
    d = self.desc
    if d is None or d.bits > 8:
      return None
    _, codes, flags = ops.quantize(d.kind, self.kernel, self.mask, d.bits, d.p0, d.p1,
                                   want_fq=False, want_codes=True, sign=d.sign)
    if int(flags.item()) & (L.FLAG_CODE_OVERFLOW | L.FLAG_MASK_NOT_BINARY):
      return None
    c2 = codes.reshape(-1, codes.shape[-1]).to(torch.int32)
    # inputs of the integer kernels are never negative (spikes, event counts), so an
    # accumulator lies in [-sum of |negative codes|, +sum of positive codes] * x_max: the
    # larger one-sided sum over the outputs bounds |acc| (about half of sum |code|)
    side = torch.maximum(c2.clamp(min=0).sum(0).max(), (-c2).clamp(min=0).sum(0).max())
    stats = torch.stack([side, c2.abs().max()]).tolist()             # one readback
    # dense kernels: column sums of the codes, for uint8 rows read as x - 128 (snnqp.h col_sum)
    col = c2.sum(0).to(torch.int32).contiguous() if self.kernel.ndim == 2 else None
    slots, stack = None, 0
    if self.kernel.ndim == 4 and self.kernel.shape[2] == 2:
      slots, stack = table_slots(c2.abs().sum(0).cpu().numpy())
      slots = torch.from_numpy(slots).to(codes.device)
    self._int = ops.Weight(L.W_I8, codes, d.L, d.m, abs_sum_max=int(stats[0]),
                           code_max=int(stats[1]), col_sum=col, ch_stack_max=stack, ch_slots=slots)
    return self._int

  def gated_codes(self):
    """The codes in the operand layout of ops.conv_gated_forward (3x3 kernels: e2m3 for codes up to 7,
    two e3m2 digits per code up to 127), or None."""
    w = self.int_weight()
    if w is None or self.kernel.ndim != 4 or tuple(self.kernel.shape[:2]) != (3, 3) or not (0 < w.code_max <= 127):
      return None
    if "_gated" not in self._wt:
      self._wt["_gated"] = ops.pack_codes_gated(w.w, w.code_max)
    return self._wt["_gated"]

  def gated_dense_codes(self, C: int, HW: int):
    """The codes of a dense kernel [C * HW, N] in the operand layout of ops.dense_gated_forward
    (codes that fit fp6, HW <= 16), or None."""
    w = self.int_weight()
    if (w is None or self.kernel.ndim != 2 or self.kernel.shape[0] != C * HW or not (0 < w.code_max <= 127)
        or HW > 16 or C not in (32, 64, 96, 128)):
      return None
    key = ("_dgated", C, HW)
    if key not in self._wt:
      self._wt[key] = ops.pack_codes_dense_gated(w.w, C, HW, w.code_max)
    return self._wt[key]

  def float_weight(self) -> ops.Weight:
    """float32 kernel_fwd of flax_qdense.py:74-85 (fake-quant, then * mask)."""
    if self._float is None:
      d = self.desc
      k = self.kernel.to(torch.float32).contiguous()
      if d is not None:
        fq, _, _ = ops.quantize(d.kind, k, self.mask, d.bits, d.p0, d.p1,
                                want_fq=True, want_codes=False, sign=d.sign)
      elif self.mask is not None:
        fq = k * self.mask.to(k.device, torch.float32)   # quant.py:491
      else:
        fq = k
      self._float = ops.Weight(L.W_F32, fq.contiguous())
    return self._float

  def int_weight_mfma(self, n_pad: int, row_perm: Optional[torch.Tensor] = None,
                            perm_key=None) -> Optional[ops.Weight]:
    """int weight with `wt` = MFMA-tiled codes (None if K % 32 != 0).  `row_perm`
    re-orders the K rows first (an exact re-indexing of the integer sum, used
    to absorb the channel-major flatten of models.py:189-190)."""
    base = self.int_weight()
    if base is None:
      return None
    key = (n_pad, perm_key)
    w = self._wt.get(key)
    if w is None:
      codes = base.w.reshape(-1, base.w.shape[-1])
      if row_perm is not None:
        codes = codes.index_select(0, row_perm)
      codes = codes.contiguous()
      if self.kernel.ndim == 2 and codes.shape[0] % 32:
        # dense layers: zero rows up to a multiple of 32 (the packed input rows carry
        # zero bits there), so K = 784 etc. stay on the MFMA kernel
        pad = 32 - codes.shape[0] % 32
        tiles_src = torch.cat([codes, codes.new_zeros((pad, codes.shape[1]))], 0)
      elif (self.kernel.ndim == 4 and row_perm is None and 2 < base.w.shape[2] <= 128
            and base.w.shape[2] not in (64, 128)):
        # convolution over bit-packed spikes: the MFMA kernels walk 64 or 128 input
        # channels per tap; other widths (config.channels = 100, 96, 48 ...) get zero
        # codes up to the next of the two (the spike words carry zero bits there)
        kh, kw, ci, co = base.w.shape
        cpad = 64 if ci <= 64 else 128          # snnqp.h: `wt` of a conv block
        padded = base.w.new_zeros((kh, kw, cpad, co))
        padded[:, :, :ci] = base.w
        tiles_src = padded.reshape(-1, co)
      else:
        tiles_src = codes
      wt = ops.pack_codes_mfma(tiles_src, n_pad) if tiles_src.shape[0] % 32 == 0 else None
      # dense kernels whose codes fit fp6 (DuQ up to 4 bits): also the 6-bit packed tiles of
      # the f8f6f4 kernel (any K: the packer pads with zero codes)
      wt6 = None
      if self.kernel.ndim == 2 and 0 < base.code_max <= 7:
        wt6 = ops.pack_codes_fp6(codes, n_pad)
      w = ops.Weight(L.W_I8, codes, base.L, base.m, wt=wt, abs_sum_max=base.abs_sum_max,
                     code_max=base.code_max, col_sum=base.col_sum, wt6=wt6, ch_stack_max=base.ch_stack_max,
                     ch_slots=base.ch_slots)
      self._wt[key] = w
    return w


_cache = TensorCache(128)


def get_packed(kernel, desc, mask) -> PackedKernel:
  pk = _cache.get((kernel, mask), desc)
  if pk is None:
    pk = _cache.put((kernel, mask), desc, PackedKernel(kernel, desc, mask))
  return pk


def clear_cache():
  _cache.clear()


def prepare_input(x, prefer_bits: Optional[bool] = None):
  """Returns (tensor-or-PackedSpikes, integer_typed): True for PackedSpikes / PackedFrames and
  uint8 tensors as they are, SPECULATE for a float32 tensor under AUTO_INTEGER_INPUTS (still the
  float32 tensor: whether it holds integers is found out on the device, by the kernel that reads
  it), False for a float32 tensor otherwise.  Other integer dtypes are widened to float32 (exact
  below 2^24) and speculate like it; nothing is read back."""
  if isinstance(x, (ops.PackedSpikes, ops.PackedFrames)):
    return x, True
  if isinstance(x, ops.GatedSpikes):
    return x, False            # real-valued: the block takes the gated form or multiplies it out
  if not isinstance(x, torch.Tensor):
    x = torch.as_tensor(x)
  if x.dtype in (torch.uint8, torch.bool):
    return (x if x.dtype == torch.uint8 else x.to(torch.uint8)), True
  if x.dtype != torch.float32:
    x = x.to(torch.float32)
  if not AUTO_INTEGER_INPUTS or x.numel() == 0:
    return x, False
  return x, SPECULATE
