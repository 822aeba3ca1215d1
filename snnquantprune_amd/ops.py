"""Thin functional layer: torch-ROCm tensors in, C-ABI calls (include/snnqp.h) out.

torch is plumbing here (device memory + the current HIP stream); every op below
is one call into libsnnqp.so.  Tensors must live on the GPU: there is no CPU
fallback and nothing in this package imports the oracle.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L


# ---------------------------------------------------------------------------
# Packed spike tensors
# ---------------------------------------------------------------------------


class PackedSpikes:
  """Binary activations, channel-packed: `bits` is int32 [..., ceil(C/32)],
  channel c of a row in bit (c & 31) of word (c >> 5); `shape` is the logical
  shape [..., C] (time-major [T, B, ..., C] inside the model)."""

  def __init__(self, bits: torch.Tensor, channels: int):
    assert bits.dtype == torch.int32 and bits.is_contiguous()
    assert bits.shape[-1] == (channels + 31) // 32
    self.bits = bits
    self.channels = int(channels)
    # (C, H, W) when this is a [T, B, H*W*C] NHWC-ordered flattening of a
    # [T, B, H, W, C] block whose logical order is channel-major
    # (examples/tcja/models.py:189-190); consumers permute weight rows instead.
    self.flat_perm = None

  @property
  def shape(self):
    return tuple(self.bits.shape[:-1]) + (self.channels,)

  @property
  def device(self):
    return self.bits.device

  @property
  def ndim(self):
    return self.bits.ndim

  def __getitem__(self, idx):
    """Indexing over leading (non-channel) axes only."""
    if not isinstance(idx, tuple):
      idx = (idx,)
    assert len(idx) < self.bits.ndim, "cannot index the packed channel axis"
    return PackedSpikes(self.bits[idx].contiguous(), self.channels)

  def reshape_leading(self, *lead):
    return PackedSpikes(self.bits.reshape(*lead, self.bits.shape[-1]), self.channels)

  def to_dense(self) -> torch.Tensor:
    return unpack_bits(self)

  def __repr__(self):
    return "PackedSpikes(shape=%s, device=%s)" % (self.shape, self.device)


class PackedFrames:
  """The model input -- 2-channel event frames [..., H, W, 2], leading axes [B, T] as the
  reference hands them over (examples/tcja/models.py:38, :109) -- in one of the wire
  formats of include/snnqp.h: `data` is int32 [..., ceil(H*W*2/32)] for _lib.EV1 (binary
  frames, 1 bit per element) or uint8 [..., H*W] for _lib.EV4 (counts <= 15, one byte per
  pixel).  What the host feed ships (feed.py): 1/8 resp. 1/2 of the uint8 bytes.  The
  first conv block stages EV1 frames directly; every other consumer unpacks."""

  def __init__(self, data: torch.Tensor, H: int, W: int, fmt: int):
    assert fmt in (L.EV1, L.EV4), fmt
    unit = frame_units(H, W, fmt)
    assert data.dtype == (torch.int32 if fmt == L.EV1 else torch.uint8), data.dtype
    assert data.shape[-1] == unit and data.is_contiguous(), (tuple(data.shape), unit)
    self.data, self.H, self.W, self.fmt = data, int(H), int(W), int(fmt)

  @property
  def shape(self):
    return tuple(self.data.shape[:-1]) + (self.H, self.W, 2)

  @property
  def device(self):
    return self.data.device

  @property
  def ndim(self):
    return self.data.ndim + 2

  @property
  def dtype(self):
    return torch.uint8

  @property
  def is_cuda(self):
    return self.data.is_cuda

  def __getitem__(self, idx):
    """Indexing over the leading (batch / time) axes only."""
    if not isinstance(idx, tuple):
      idx = (idx,)
    assert len(idx) < self.data.ndim, "cannot index inside a packed frame"
    return PackedFrames(self.data[idx].contiguous(), self.H, self.W, self.fmt)

  def narrow(self, dim, start, length):
    assert dim < self.data.ndim - 1
    return PackedFrames(self.data.narrow(dim, start, length).contiguous(), self.H, self.W, self.fmt)

  def to(self, device, non_blocking=False):
    return PackedFrames(self.data.to(device, non_blocking=non_blocking), self.H, self.W, self.fmt)

  def to_u8(self) -> torch.Tensor:
    return unpack_frames(self)

  def __repr__(self):
    return "PackedFrames(%s, shape=%s, device=%s)" % (
        "EV1" if self.fmt == L.EV1 else "EV4", self.shape, self.device)


class GatedSpikes:
  """gate[T, B, C] x spikes[T, B, H, W, C], NOT multiplied out: what a TCJA block hands to the next
  block (x_seq * out[:, :, None, None, :], examples/tcja/models.py:97).  A quantised 3x3 conv block
  contracts it in the 'gint' form (conv_gated_forward: the nine taps of a channel as an integer sum,
  the gates in one float32 chain); every other consumer multiplies it out (to_dense)."""

  def __init__(self, spikes: "PackedSpikes", gate: torch.Tensor, flat: bool = False):
    assert isinstance(spikes, PackedSpikes) and spikes.ndim == 5, "spikes [T, B, H, W, C], bit-packed"
    assert gate.dtype == torch.float32 and tuple(gate.shape) == (spikes.shape[0], spikes.shape[1], spikes.channels)
    self.spikes, self.gate = spikes, gate.contiguous()
    # flat: stands for the channel-major flattening [T, B, C H W] of the product (the transpose +
    # reshape of examples/tcja/models.py:189-190) -- nothing is moved, a dense block reads the
    # raster through its own weight layout (dense_gated_forward)
    self.flat = bool(flat)

  @property
  def shape(self):
    if self.flat:
      T, B, H, W, C = self.spikes.shape
      return (T, B, C * H * W)
    return self.spikes.shape

  @property
  def ndim(self):
    return 3 if self.flat else 5

  @property
  def device(self):
    return self.spikes.device

  def flattened(self) -> "GatedSpikes":
    return GatedSpikes(self.spikes, self.gate, flat=True)

  def to_dense(self) -> torch.Tensor:
    x = apply_gate(self.spikes, self.gate)
    if self.flat:
      x = x.permute(0, 1, 4, 2, 3)
      x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3] * x.shape[4]).contiguous()
    return x


def frame_units(H: int, W: int, fmt: int) -> int:
  """Words (EV1) / bytes (EV4) of one packed frame."""
  return (H * W * 2 + 31) // 32 if fmt == L.EV1 else H * W


def pack_frames_host(x, fmt: int) -> PackedFrames:
  """uint8 frames [..., H, W, 2] in host memory (numpy array or CPU tensor) -> PackedFrames
  in host memory, with numpy -- what a data pipeline does once per sample before the feed.
  Raises if a value does not fit the format (EV1: > 1, EV4: > 15)."""
  import numpy as np
  a = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
  if a.dtype != np.uint8 or a.ndim < 3 or a.shape[-1] != 2:
    raise ValueError("event frames must be uint8 [..., H, W, 2], got %s %s" % (a.dtype, a.shape))
  H, W = a.shape[-3], a.shape[-2]
  lead = a.shape[:-3]
  limit = 1 if fmt == L.EV1 else 15
  if a.size and int(a.max()) > limit:
    raise ValueError("event count %d does not fit the %s frame format (max %d)"
                     % (int(a.max()), "EV1" if fmt == L.EV1 else "EV4", limit))
  if fmt == L.EV1:
    flat = np.ascontiguousarray(a).reshape(lead + (H * W * 2,))
    b = np.packbits(flat, axis=-1, bitorder="little")
    nbytes = frame_units(H, W, fmt) * 4
    if b.shape[-1] != nbytes:
      b = np.concatenate([b, np.zeros(lead + (nbytes - b.shape[-1],), np.uint8)], -1)
    data = torch.from_numpy(np.ascontiguousarray(b).view(np.int32))
  else:
    data = torch.from_numpy(np.ascontiguousarray(a[..., 0] | (a[..., 1] << 4)).reshape(lead + (H * W,)))
  return PackedFrames(data, H, W, fmt)


def pack_frames(x: torch.Tensor, fmt: int, flags: Optional[torch.Tensor] = None) -> PackedFrames:
  """uint8 frames [..., H, W, 2] on the GPU -> PackedFrames (snnqp_pack_frames).  Values the
  format cannot hold are saturated and flagged into the int32 device word `flags`
  (_lib.FLAG_GT_ONE / FLAG_GT_15) when one is given; nothing is read back here."""
  _require_gpu(x, flags)
  assert x.dtype == torch.uint8 and x.shape[-1] == 2 and x.ndim >= 3, (x.dtype, tuple(x.shape))
  x = x.contiguous()
  H, W = x.shape[-3], x.shape[-2]
  lead = tuple(x.shape[:-3])
  frames = 1
  for d in lead:
    frames *= d
  data = torch.empty(lead + (frame_units(H, W, fmt),),
                     dtype=torch.int32 if fmt == L.EV1 else torch.uint8, device=x.device)
  L.check(L.lib().snnqp_pack_frames(_ptr(x), frames, H, W, fmt, _ptr(data), _ptr(flags), _stream()))
  return PackedFrames(data, H, W, fmt)


def pack_frames_checked(x: torch.Tensor) -> Tuple[PackedFrames, torch.Tensor]:
  """uint8 or float32 frames [..., H, W, 2] on the GPU -> (bit-packed EV1 frames, one int32 device
  word) in one pass (snnqp_pack_frames_checked): the word is zero iff every value is 0 or 1, i.e.
  iff the packed frames ARE the tensor -- the speculative half of conv_lif_forward(binary_first=True)."""
  _require_gpu(x)
  assert x.dtype in (torch.uint8, torch.float32) and x.shape[-1] == 2 and x.ndim >= 3, (x.dtype, tuple(x.shape))
  x = x.contiguous()
  H, W = x.shape[-3], x.shape[-2]
  lead = tuple(x.shape[:-3])
  frames = 1
  for d in lead:
    frames *= d
  data = torch.empty(lead + (frame_units(H, W, L.EV1),), dtype=torch.int32, device=x.device)
  flags = torch.empty(1, dtype=torch.int32, device=x.device)
  L.check(L.lib().snnqp_pack_frames_checked(_ptr(x), L.U8 if x.dtype == torch.uint8 else L.F32, frames, H, W,
                                            _ptr(data), _ptr(flags), _stream()))
  return PackedFrames(data, H, W, L.EV1), flags


def unpack_frames(p: PackedFrames) -> torch.Tensor:
  """PackedFrames -> uint8 [..., H, W, 2] (snnqp_unpack_frames)."""
  _require_gpu(p.data)
  lead = tuple(p.data.shape[:-1])
  frames = 1
  for d in lead:
    frames *= d
  y = torch.empty(lead + (p.H, p.W, 2), dtype=torch.uint8, device=p.device)
  L.check(L.lib().snnqp_unpack_frames(_ptr(p.data), p.fmt, frames, p.H, p.W, _ptr(y), _stream()))
  return y


# ---------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
  return None if t is None else ctypes.c_void_p(t.data_ptr())


def _require_gpu(*ts):
  for t in ts:
    if t is None:
      continue
    if not t.is_cuda:
      raise RuntimeError(
          "snnquantprune_amd ops run on the GPU only (tensor on %s); there is no "
          "CPU fallback" % t.device)


def _f32c(t: torch.Tensor) -> torch.Tensor:
  if t.dtype != torch.float32:
    t = t.to(torch.float32)
  return t.contiguous()


def is_pow2(x: float) -> bool:
  import math
  if not (x > 0) or math.isinf(x):
    return False
  m, _ = math.frexp(x)
  return m == 0.5


@dataclass
class Weight:
  """Kernel after the weight transforms (flax_qdense.py:74-85).

  wtype W_I8: `w` = int8 codes * mask in the reference's layout, current =
  fl(fl(acc / L) * m); `wt` = optional MFMA-tiled codes (pack_codes_mfma).
  wtype W_F32: `w` = float32 fake-quantised * mask kernel."""
  wtype: int
  w: torch.Tensor
  L: float = 1.0
  m: float = 1.0
  wt: Optional[torch.Tensor] = None
  abs_sum_max: int = 0      # |acc| <= abs_sum_max * x_max (max one-sided code sum over the outputs)
  code_max: int = 0         # max |code| (<= 7: exact in fp6)
  min_current_bits: int = 0  # smallest non-zero |BN(dequant(acc))| as float bits (current_min)
  col_sum: Optional[torch.Tensor] = None   # dense: int32 [N] column sums of the codes (uint8 input)
  wt6: Optional[torch.Tensor] = None       # dense, code_max <= 7: fp6 MFMA tiles (pack_codes_fp6)
  ch_stack_max: int = 0     # event layer: largest stacked per-channel code range (snnqp.h), 0 = unknown
  ch_slots: Optional[torch.Tensor] = None  # event layer: int32 [padded Cout] table slots (packing.table_slots)

  def struct(self) -> L.WeightT:
    # (built once per object: a Weight is not modified after the pack step made it --
    # dataclasses.replace makes a new one -- and it keeps its tensors alive)
    st = self.__dict__.get("_cstruct")
    if st is None:
      st = L.WeightT(self.wtype, self.w.data_ptr(), float(self.L), float(self.m),
                     int(self.abs_sum_max), int(self.code_max),
                     None if self.col_sum is None else self.col_sum.data_ptr(),
                     None if self.wt6 is None else self.wt6.data_ptr(),
                     int(self.min_current_bits), int(self.ch_stack_max),
                     None if self.ch_slots is None else self.ch_slots.data_ptr())
      self.__dict__["_cstruct"] = st
    return st

  @property
  def is_int(self):
    return self.wtype == L.W_I8


@dataclass
class Neuron:
  kind: int
  k: float = 2.0
  v_threshold: float = 1.0
  v_reset: float = 0.0
  decay: Optional[torch.Tensor] = None   # LIF: sigmoid(tau) per feature (device)

  def struct(self) -> L.NeuronT:
    return L.NeuronT(self.kind, float(self.k), float(self.v_threshold),
                     float(self.v_reset),
                     None if self.decay is None else self.decay.data_ptr())


@dataclass
class BnCoeffs:
  mean: torch.Tensor
  mul: torch.Tensor
  bias: torch.Tensor
  flags: int = 0      # _lib.BN_MEAN_ZERO | BN_BIAS_ZERO | BN_MUL_UNIFORM: known from the host-side fold

  def struct(self) -> L.BnT:
    return L.BnT(self.mean.data_ptr(), self.mul.data_ptr(), self.bias.data_ptr(), int(self.flags))


@dataclass
class ConvGeom:
  H: int
  W: int
  Cin: int
  Cout: int
  KH: int
  KW: int
  stride: Tuple[int, int] = (1, 1)
  pad: Tuple[Tuple[int, int], Tuple[int, int]] = ((0, 0), (0, 0))
  in_dil: Tuple[int, int] = (1, 1)
  k_dil: Tuple[int, int] = (1, 1)
  groups: int = 1

  def struct(self) -> L.ConvGeomT:
    return L.ConvGeomT(self.H, self.W, self.Cin, self.Cout, self.KH, self.KW,
                       self.stride[0], self.stride[1], self.pad[0][0],
                       self.pad[0][1], self.pad[1][0], self.pad[1][1],
                       self.in_dil[0], self.in_dil[1], self.k_dil[0],
                       self.k_dil[1], self.groups)

  def tag(self) -> str:
    return "%dx%dx%d->%d" % (self.H, self.W, self.Cin, self.Cout)

  def out_hw(self) -> Tuple[int, int]:
    oh, ow = ctypes.c_int32(), ctypes.c_int32()
    g = self.struct()
    L.check(L.lib().snnqp_conv_out_shape(ctypes.byref(g), ctypes.byref(oh),
                                         ctypes.byref(ow)))
    return oh.value, ow.value


@dataclass(frozen=True)
class Conv3dGeom:
  """Geometry of a 3-D QuantConv (snnqp_conv3d_geom_t): axis 0 = depth, 1 = height, 2 = width."""
  D: int
  H: int
  W: int
  Cin: int
  Cout: int
  KD: int
  KH: int
  KW: int
  stride: Tuple[int, int, int] = (1, 1, 1)
  pad: Tuple[Tuple[int, int], Tuple[int, int], Tuple[int, int]] = ((0, 0), (0, 0), (0, 0))
  in_dil: Tuple[int, int, int] = (1, 1, 1)
  k_dil: Tuple[int, int, int] = (1, 1, 1)
  groups: int = 1

  def struct(self) -> L.Conv3dGeomT:
    i3 = ctypes.c_int32 * 3
    return L.Conv3dGeomT(self.D, self.H, self.W, self.Cin, self.Cout, self.KD, self.KH, self.KW,
                         i3(*self.stride), i3(*[p[0] for p in self.pad]), i3(*[p[1] for p in self.pad]),
                         i3(*self.in_dil), i3(*self.k_dil), self.groups)

  def tag(self) -> str:
    return "%dx%dx%dx%d->%d" % (self.D, self.H, self.W, self.Cin, self.Cout)

  def out_dhw(self) -> Tuple[int, int, int]:
    od, oh, ow = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    g = self.struct()
    L.check(L.lib().snnqp_conv3d_out_shape(ctypes.byref(g), ctypes.byref(od), ctypes.byref(oh), ctypes.byref(ow)))
    return od.value, oh.value, ow.value


def _in_desc(x):
  """(pointer tensor, in_type, words/elements per pixel-row unit)."""
  if isinstance(x, PackedSpikes):
    return x.bits, L.BITS
  if isinstance(x, PackedFrames):
    return x.data, x.fmt
  if x.dtype == torch.uint8:
    return x, L.U8
  if x.dtype == torch.float32:
    return x, L.F32
  raise TypeError("activation dtype %s not supported (float32, uint8 or "
                  "PackedSpikes)" % x.dtype)


# ---------------------------------------------------------------------------
# weight transforms
# ---------------------------------------------------------------------------


def quantize(kind: int, w: torch.Tensor, mask: Optional[torch.Tensor], bits: int,
             p0: float, p1: float = 0.0, want_fq: bool = True,
             want_codes: bool = False, sign: bool = True):
  """snnqp_quantize_ex: returns (fq | None, codes | None, flags tensor | None); sign = False is
  the reference's unsigned form (levels 0 .. 2^bits - 1, quant.py:338-341)."""
  w = _f32c(w)
  _require_gpu(w, mask)
  if mask is not None:
    mask = _f32c(mask)
    assert mask.shape == w.shape
  fq = torch.empty_like(w) if want_fq else None
  codes = torch.empty(w.shape, dtype=torch.int8, device=w.device) if want_codes else None
  flags = torch.zeros(1, dtype=torch.int32, device=w.device) if (
      want_codes or mask is not None) else None
  L.check(L.lib().snnqp_quantize_ex(kind, _ptr(w), _ptr(mask), w.numel(), int(bits), 1 if sign else 0,
                                    float(p0), float(p1), _ptr(fq), _ptr(codes),
                                    _ptr(flags), _stream()))
  return fq, codes, flags


def pack_codes_mfma(codes: torch.Tensor, n_pad: Optional[int] = None) -> torch.Tensor:
  """[K, N] int8 codes -> MFMA B-operand tiles [Npad/32, K/32, 64, 16]."""
  _require_gpu(codes)
  assert codes.dtype == torch.int8
  c2 = codes.reshape(-1, codes.shape[-1]).contiguous()
  K, N = c2.shape
  n_pad = (N + 31) // 32 * 32 if n_pad is None else n_pad
  if K % 32:
    raise ValueError("MFMA tiling needs K % 32 == 0 (K = %d)" % K)
  wt = torch.empty((n_pad // 32, K // 32, 64, 16), dtype=torch.int8, device=codes.device)
  L.check(L.lib().snnqp_pack_codes_mfma(_ptr(c2), K, N, n_pad, _ptr(wt), _stream()))
  return wt


def pack_codes_fp6(codes: torch.Tensor, n_pad: Optional[int] = None) -> torch.Tensor:
  """[K, N] int8 codes of magnitude <= 7 -> fp6 MFMA tiles, uint8 [Npad/32, ceil(K/64), 1536]
  (snnqp_pack_codes_fp6): the packed form the f8f6f4 dense kernel streams."""
  _require_gpu(codes)
  assert codes.dtype == torch.int8
  c2 = codes.reshape(-1, codes.shape[-1]).contiguous()
  K, N = c2.shape
  n_pad = (N + 31) // 32 * 32 if n_pad is None else n_pad
  wt6 = torch.empty((n_pad // 32, (K + 63) // 64, 1536), dtype=torch.uint8, device=codes.device)
  L.check(L.lib().snnqp_pack_codes_fp6(_ptr(c2), K, N, n_pad, _ptr(wt6), _stream()))
  return wt6


# ---------------------------------------------------------------------------
# activation formats
# ---------------------------------------------------------------------------


class NotCapturable(RuntimeError):
  """The step needs a value on the host and is being recorded into a hipGraph, which cannot wait
  for one.  (Nothing in this package does any more -- float32 inputs are checked on the device,
  snnqp.h x_flags -- the class stays for callers that catch it.)"""


@dataclass
class FloatFallback:
  """What redoes an integer block whose float32 input turns out not to be integer-valued
  (the reference casts every input to float32, flax_qdense.py:67 / flax_qconv.py:101: a tensor of
  integers in [0, 255] takes the exact-integer kernels, anything else the float32 ones -- decided
  per tensor on the device, with no read-back: snnqp.h, x_flags and the *_if entry points).

  weight  the float32 fake-quantised kernel of the same layer (Weight, W_F32)
  x       the float32 tensor, when the integer launch reads a narrowed copy of it (narrow_f32_async /
          pack_bits_checked); None when the integer kernel stages the float32 tensor itself
  pred    the device word (int32 tensor of one element) the narrowing pass reported into; None when
          the integer kernel reports into a word of its own"""
  weight: "Weight"
  x: Optional[torch.Tensor] = None
  pred: Optional[torch.Tensor] = None


def forget_inputs():
  """Kept for callers of earlier versions (bench.py called it each step to drop cached facts about
  the last batch): nothing about an activation tensor is cached on the host any more."""


class CountHint:
  """What the first layer's kernel should size its tables for: the largest event count MOST of
  its staged chunks will hold on this device (1 = binary frames).  The kernel never trusts it
  -- it checks every chunk of input it stages and falls back per chunk (snnqp.h, x_max) -- and
  reports into `word` (eight int32 words: the largest value it saw and the number of chunks by
  their largest value, <= 1, 2, <= 7, <= 31, above, exactly 3); the words are copied back with a
  non-blocking copy and looked at on a later call, when the copy has long finished, so the host
  never waits for the device to learn about its input.

  The hint is the bucket bound that minimises the estimated time of the next launch: chunks
  within the hint run the table sized for it (the larger the table, the slower: COST), chunks
  above it the general path.  One hot pixel -- real DVS sensors have them -- therefore costs the
  few chunks that hold it, not the batch (a hint that followed the maximum would put every
  chunk on the slowest path)."""

  BOUNDS = (1, 2, 3, 7, 31)                 # the buckets' upper bounds; above 31: no table
  # relative time of a chunk in the table mode of a bound (headline layer, tools/conv0_hint_time.py:
  # per-channel tables 5.96 / 5.82 / 5.82 ms for hints 1 / 2 / 3, shared table 7.02) and on the
  # general path (x - 128, arithmetic: 10.9)
  COST = (1.0, 1.0, 1.0, 1.19, 1.19)
  GENERAL = 1.85

  def __init__(self, device):
    self.value = 1
    self.word = torch.zeros(8, dtype=torch.int32, device=device)
    self._host = torch.zeros(8, dtype=torch.int32).pin_memory()
    self._event = None
    self.max_seen = 0
    self.saw_counts = False      # a value above 1 has been reported on this device (sticky)

  def binary_so_far(self) -> bool:
    """Every launch that reported so far met binary frames only: byte / float32 frames are then
    worth packing to bits in front of the event layer (conv_lif_forward(binary_first=True)).
    Sticky the other way: one count above 1 -- a hot pixel -- and the frames go in as they are
    from then on, a failed speculation costs the batch twice."""
    return self.current() == 1 and not self.saw_counts

  @staticmethod
  def capturing() -> bool:
    """True while the current stream is being captured into a hipGraph: the read-back
    (a copy, a memset and an event) must stay out of the graph -- an event recorded on a
    capturing stream cannot be queried afterwards, and every replay would re-zero the
    shared word -- so a captured launch gets no `x_seen` word and the last known hint."""
    return torch.cuda.is_current_stream_capturing()

  @classmethod
  def choose(cls, words) -> int:
    """words = x_seen[1..6]: chunks with largest value <= 1, 2, <= 7, <= 31, above, and (part of
    the third) exactly 3 -> the hint."""
    w = list(words) + [0] * (6 - len(words))
    hist = [w[0], w[1], w[5], w[2] - w[5], w[3], w[4]]           # <= 1, 2, 3, 4..7, <= 31, above
    total = sum(hist)
    if total == 0:
      return 1
    best, best_cost = 255, cls.GENERAL * total          # no table fits: every chunk general
    below = 0
    for k, (bound, cost) in enumerate(zip(cls.BOUNDS, cls.COST)):
      below += hist[k]
      c = cost * below + cls.GENERAL * (total - below)
      if c < best_cost - 1e-9:
        best, best_cost = bound, c
    return best

  def current(self) -> int:
    if self.capturing():
      return self.value
    if self._event is not None and self._event.query():
      self._event = None
      words = [int(v) for v in self._host]
      self.max_seen = words[0]
      if self.max_seen > 1:
        self.saw_counts = True
      if sum(words[1:6]) > 0:
        # (never above what was seen: the tables are sized by abs_sum_max * hint, and a bucket
        # bound of 7 where the data stops at 4 can push 8-bit codes past the table's capacity)
        self.value = max(1, min(self.choose(words[1:7]), self.max_seen))
    return self.value

  def seen_word(self):
    """The device words a launch reports into, or None under graph capture."""
    return None if self.capturing() else self.word

  def launched(self):
    """After a launch that was given `word`: start its read-back (at most one in flight)."""
    if self.capturing():
      return
    if self._event is None:
      self._host.copy_(self.word, non_blocking=True)
      self.word.zero_()
      self._event = torch.cuda.Event()
      self._event.record()


_count_hints = {}


def count_hint(device) -> CountHint:
  key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
  h = _count_hints.get(key)
  if h is None:
    h = _count_hints[key] = CountHint(device)
  return h


def reset_count_hints(device=None):
  """Forget what the event layer has reported about its input on `device` (all devices: None): the
  next launch starts from "binary frames, nothing seen" again.  For a caller that switches to
  another data stream (bench.py between its legs); a model fed by one stream never needs it."""
  if device is None:
    _count_hints.clear()
  else:
    d = torch.device(device)
    _count_hints.pop(d.index if d.index is not None else torch.cuda.current_device(), None)


def f32_to_u8(x: torch.Tensor) -> torch.Tensor:
  x = _f32c(x)
  _require_gpu(x)
  y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
  L.check(L.lib().snnqp_f32_to_u8(_ptr(x), _ptr(y), x.numel(), _stream()))
  return y


def narrow_f32_async(x: torch.Tensor):
  """float32 activations -> (uint8 copy, pred): one device pass (snnqp_narrow_f32) and NO
  read-back.  `pred` is a one-element int32 device tensor, non-zero when some element is not an
  integer in [0, 255] (the copy is then meaningless): the predicate of the float32 launch that
  follows the integer one (FloatFallback)."""
  x = _f32c(x)
  _require_gpu(x)
  y = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
  flags = torch.zeros(2, dtype=torch.int32, device=x.device)
  L.check(L.lib().snnqp_narrow_f32(_ptr(x), _ptr(y), x.numel(), _ptr(flags), _stream()))
  return y, flags[1:2]


def pack_bits_checked(x: torch.Tensor):
  """float32 / uint8 [..., C] -> (PackedSpikes, pred): `pred` (one int32 device word) is non-zero
  when some element is neither 0 nor 1 -- the raster then is not the tensor.  No read-back."""
  _require_gpu(x)
  if x.dtype not in (torch.float32, torch.uint8):
    x = x.to(torch.float32)
  x = x.contiguous()
  C = x.shape[-1]
  rows = x.numel() // C if C else 0
  bits = torch.empty(tuple(x.shape[:-1]) + ((C + 31) // 32,), dtype=torch.int32, device=x.device)
  flags = torch.zeros(1, dtype=torch.int32, device=x.device)
  L.check(L.lib().snnqp_pack_bits_checked(_ptr(x), L.F32 if x.dtype == torch.float32 else L.U8,
                                          rows, C, _ptr(bits), _ptr(flags), _stream()))
  return PackedSpikes(bits, C), flags


def pack_bits(x: torch.Tensor) -> PackedSpikes:
  """float32 / uint8 [..., C] (nonzero = spike) -> PackedSpikes."""
  _require_gpu(x)
  if x.dtype not in (torch.float32, torch.uint8):
    x = x.to(torch.float32)
  x = x.contiguous()
  C = x.shape[-1]
  rows = x.numel() // C if C else 0
  bits = torch.empty(tuple(x.shape[:-1]) + ((C + 31) // 32,), dtype=torch.int32,
                     device=x.device)
  L.check(L.lib().snnqp_pack_bits(_ptr(x), L.F32 if x.dtype == torch.float32 else L.U8,
                                  rows, C, _ptr(bits), _stream()))
  return PackedSpikes(bits, C)


def unpack_bits(s: PackedSpikes) -> torch.Tensor:
  _require_gpu(s.bits)
  y = torch.empty(s.shape, dtype=torch.float32, device=s.device)
  rows = y.numel() // s.channels
  L.check(L.lib().snnqp_unpack_bits(_ptr(s.bits), rows, s.channels, _ptr(y), _stream()))
  return y


# ---------------------------------------------------------------------------
# connection only
# ---------------------------------------------------------------------------


def fseq_gemm_supported(geom: ConvGeom) -> bool:
  """Shapes the float32-MFMA connection kernel serves (fseq_gemm.hip): stride 1, no
  dilation or groups, output size == input size (dense = 1x1 on a 1x1 image)."""
  (pt, pb), (pl, pr) = geom.pad
  return (geom.groups == 1 and tuple(geom.stride) == (1, 1) and tuple(geom.in_dil) == (1, 1)
          and tuple(geom.k_dil) == (1, 1)
          and pt + pb == geom.KH - 1 and pl + pr == geom.KW - 1)


def conv_forward(x, geom: ConvGeom, weight: Weight, want_acc: bool = False):
  """x [NB, H, W, Cin] -> float32 [NB, OH, OW, Cout] (+ int32 accumulators)."""
  xt, in_type = _in_desc(x)
  xt = xt.contiguous()
  _require_gpu(xt, weight.w)
  NB = xt.shape[0]
  OH, OW = geom.out_hw()
  y = torch.empty((NB, OH, OW, geom.Cout), dtype=torch.float32, device=xt.device)
  acc = torch.empty(y.shape, dtype=torch.int32, device=xt.device) if want_acc else None
  g, w = geom.struct(), weight.struct()
  L.check(L.lib().snnqp_conv_forward(_ptr(xt), in_type, NB, ctypes.byref(g),
                                     ctypes.byref(w), _ptr(y), _ptr(acc), _stream()))
  return (y, acc) if want_acc else y


def conv3d_lif_forward(x, geom: Conv3dGeom, weight: Weight, neuron: Optional[Neuron] = None,
                       bn: Optional[BnCoeffs] = None, u0: Optional[torch.Tensor] = None, want_u: bool = True,
                       packed_out: bool = False, pred: Optional[torch.Tensor] = None, out=None):
  """3-D QuantConv (snnqp_conv3d_lif_forward).  With a neuron: x [T, B, D, H, W, Cin] ->
  (u_T [B, OD, OH, OW, Cout] | None, spikes [T, B, OD, OH, OW, Cout]); without: x [NB, D, H, W, Cin]
  -> float32 currents [NB, OD, OH, OW, Cout].  pred / out: the predicated second launch of a
  speculative pair writes into the outputs `out` of the first, only if *pred != 0."""
  xt, in_type = _in_desc(x)
  xt = xt.contiguous()
  _require_gpu(xt, weight.w, u0)
  OD, OH, OW = geom.out_dhw()
  dev = xt.device
  unit = geom.D * geom.H * geom.W * xt.shape[-1]
  g, w = geom.struct(), weight.struct()
  b = bn.struct() if bn is not None else None
  if neuron is None:
    NB = xt.shape[0]
    y = out if out is not None else torch.empty((NB, OD, OH, OW, geom.Cout), dtype=torch.float32, device=dev)
    with _timed("conv3d[%s]" % geom.tag()):
      L.check(L.lib().snnqp_conv3d_lif_forward(_ptr(pred), _ptr(xt), in_type, 0, unit, 1, NB, ctypes.byref(g),
                                               ctypes.byref(w), None, None, None, None, _ptr(y), L.F32, _stream()))
    return y
  T, B = xt.shape[0], xt.shape[1]
  if out is not None:
    u_out, s = out
    s = s.bits if isinstance(s, PackedSpikes) else s
  else:
    u_out = torch.empty((B, OD, OH, OW, geom.Cout), dtype=torch.float32, device=dev) if want_u else None
    oshape = (T, B, OD, OH, OW)
    s = torch.empty(oshape + (((geom.Cout + 31) // 32,) if packed_out else (geom.Cout,)),
                    dtype=torch.int32 if packed_out else torch.float32, device=dev)
  if u0 is not None:
    u0 = _f32c(u0)
    assert tuple(u0.shape) == (B, OD, OH, OW, geom.Cout), (u0.shape, (B, OD, OH, OW, geom.Cout))
  n = neuron.struct()
  with _timed("conv3d[%s]" % geom.tag()):
    L.check(L.lib().snnqp_conv3d_lif_forward(_ptr(pred), _ptr(xt), in_type, B * unit, unit, T, B, ctypes.byref(g),
                                             ctypes.byref(w), ctypes.byref(b) if b is not None else None,
                                             ctypes.byref(n), _ptr(u0), _ptr(u_out), _ptr(s),
                                             L.BITS if packed_out else L.F32, _stream()))
  return u_out, (PackedSpikes(s, geom.Cout) if packed_out else s)


def conv_forward_speculative(x: torch.Tensor, geom: ConvGeom, int_weight: Weight, float_weight: Weight):
  """The connection alone on a float32 tensor [NB, H, W, Cin] that MAY hold integers in [0, 255]
  (QuantDense / QuantConv called outside a SpikingBlock): narrowed to uint8 and checked in one
  device pass, the integer connection on the copy, then the float32 connection into the same
  output, executed only if the check failed (snnqp_conv_forward_if).  Nothing is read back."""
  x = _f32c(x)
  x8, pred = narrow_f32_async(x)
  y = conv_forward(x8, geom, int_weight)
  g, w = geom.struct(), float_weight.struct()
  L.check(L.lib().snnqp_conv_forward_if(_ptr(pred), _ptr(x), L.F32, x.shape[0], ctypes.byref(g), ctypes.byref(w),
                                        _ptr(y), _stream()))
  return y


def pack_codes_gated(codes: torch.Tensor, code_max: int = 7) -> torch.Tensor:
  """int8 HWIO codes [3, 3, Cin, Cout] -> the operand layout of conv_gated_forward
  (snnqp_pack_codes_gated_ex): e2m3 for code_max <= 7, two e3m2 digits per code up to 127."""
  _require_gpu(codes)
  assert codes.dtype == torch.int8 and codes.ndim == 4 and tuple(codes.shape[:2]) == (3, 3)
  codes = codes.contiguous()
  cin, cout = codes.shape[2], codes.shape[3]
  out = torch.empty(int(L.lib().snnqp_conv_gated_packed_bytes_ex(cin, cout, int(code_max))), dtype=torch.uint8,
                    device=codes.device)
  L.check(L.lib().snnqp_pack_codes_gated_ex(_ptr(codes), cin, cout, int(code_max), _ptr(out), _stream()))
  return out


def conv_gated_forward(x: GatedSpikes, geom: ConvGeom, weight: Weight, packed: torch.Tensor) -> torch.Tensor:
  """gate x raster [T, B, H, W, Cin] -> float32 currents [T, B, H, W, Cout] in the 'gint' form
  (snnqp_conv_gated_forward); raises SnnqpError(EUNSUPPORTED) for shapes it does not serve."""
  bits, gate = x.spikes.bits.contiguous(), x.gate
  _require_gpu(bits, gate, weight.w, packed)
  T, B, H, W, _ = x.shape
  assert (H, W, x.spikes.channels) == (geom.H, geom.W, geom.Cin), (x.shape, geom)
  y = torch.empty((T, B, H, W, geom.Cout), dtype=torch.float32, device=bits.device)
  g, w = geom.struct(), weight.struct()
  with _timed("conv[gated %s]" % geom.tag()):
    L.check(L.lib().snnqp_conv_gated_forward(_ptr(bits), _ptr(gate), T * B, ctypes.byref(g), ctypes.byref(w),
                                             _ptr(packed), _ptr(y), _stream()))
  return y


def pack_codes_dense_gated(codes: torch.Tensor, C: int, HW: int, code_max: int = 7) -> torch.Tensor:
  """int8 codes [C * HW, N] (rows channel-major) -> the operand layout of dense_gated_forward
  (snnqp_pack_codes_dense_gated_ex): e2m3 for code_max <= 7, two e3m2 digits per code up to 127."""
  _require_gpu(codes)
  assert codes.dtype == torch.int8 and codes.ndim == 2 and codes.shape[0] == C * HW
  codes = codes.contiguous()
  N = codes.shape[1]
  out = torch.empty(int(L.lib().snnqp_dense_gated_packed_bytes_ex(C, N, int(code_max))), dtype=torch.uint8,
                    device=codes.device)
  L.check(L.lib().snnqp_pack_codes_dense_gated_ex(_ptr(codes), C, HW, N, int(code_max), _ptr(out), _stream()))
  return out


def dense_gated_forward(x: GatedSpikes, weight: Weight, packed: torch.Tensor) -> torch.Tensor:
  """channel-major flattening of gate x raster [T, B, (C, H, W)] -> float32 currents [T, B, N] in
  the 'gint' form (snnqp_dense_gated_forward); raises SnnqpError(EUNSUPPORTED) for shapes it does
  not serve."""
  bits, gate = x.spikes.bits.contiguous(), x.gate
  _require_gpu(bits, gate, weight.w, packed)
  T, B, H, W, C = x.spikes.shape
  N = weight.w.shape[-1]
  assert weight.w.shape[0] == C * H * W, (tuple(weight.w.shape), x.spikes.shape)
  y = torch.empty((T, B, N), dtype=torch.float32, device=bits.device)
  w = weight.struct()
  with _timed("dense[gated K=%d N=%d]" % (C * H * W, N)):
    L.check(L.lib().snnqp_dense_gated_forward(_ptr(bits), _ptr(gate), T * B, H * W, C, N, ctypes.byref(w),
                                              _ptr(packed), _ptr(y), _stream()))
  return y


# ---------------------------------------------------------------------------
# per-launch timing with HIP events on the launch stream (bench.py roofline)
# ---------------------------------------------------------------------------

_PROFILE = None
PROFILE_NOTES = {}      # tag -> facts about the launches of the last profiled region


def profile_start():
  global _PROFILE
  _PROFILE = {}
  PROFILE_NOTES.clear()


def profile_stop():
  """{tag: (launches, total milliseconds)} since profile_start()."""
  global _PROFILE
  prof, _PROFILE = _PROFILE or {}, None
  torch.cuda.synchronize()
  return {tag: (len(evs), sum(a.elapsed_time(b) for a, b in evs))
          for tag, evs in prof.items()}


class _timed:
  def __init__(self, tag):
    self.tag = tag

  def __enter__(self):
    if _PROFILE is not None:
      self.a = torch.cuda.Event(enable_timing=True)
      self.b = torch.cuda.Event(enable_timing=True)
      self.a.record()

  def __exit__(self, *exc):
    if _PROFILE is not None and exc[0] is None:
      self.b.record()
      _PROFILE.setdefault(self.tag, []).append((self.a, self.b))
    return False


# ---------------------------------------------------------------------------
# fused SpikingBlock
# ---------------------------------------------------------------------------


_current_min_cache = None


def current_min_bits(weight: Weight, bn: Optional[BnCoeffs], bound: int, cout: int) -> int:
  """float32 bits of the smallest non-zero |BatchNorm(dequant(acc))| over |acc| <= bound and
  the channels: what Weight.min_current_bits wants (one device pass + a 4-byte read-back per
  (weights, BatchNorm) version, cached)."""
  global _current_min_cache
  if _current_min_cache is None:
    from ._cache import TensorCache
    _current_min_cache = TensorCache(64)
  key = (weight.w,) + ((bn.mean, bn.mul, bn.bias) if bn is not None else (None, None, None))
  extra = (int(bound), int(cout), float(weight.L), float(weight.m))
  v = _current_min_cache.get(key, extra)
  if v is None:
    _require_gpu(weight.w)
    out = torch.full((1,), 0x7F800000, dtype=torch.int32, device=weight.w.device)
    w = weight.struct()
    b = bn.struct() if bn is not None else None
    L.check(L.lib().snnqp_current_min(ctypes.byref(w), ctypes.byref(b) if b is not None else None,
                                      int(bound), int(cout), _ptr(out), _stream()))
    v = int(out.item()) & 0xFFFFFFFF
    _current_min_cache.put(key, extra, v)
  return v


def _tb_strides(x, T, B, time_major: bool, unit: int):
  """Element (word) strides of t and b for a contiguous tensor."""
  if time_major:
    return B * unit, unit
  return unit, T * unit


def conv_lif_forward(x, geom: ConvGeom, weight: Weight, neuron: Neuron,
                     bn: Optional[BnCoeffs] = None, u0: Optional[torch.Tensor] = None,
                     want_u: bool = True, packed_out: bool = False, pool: int = 1,
                     impl: int = L.IMPL_AUTO, time_major: bool = True, x_max: int = 0,
                     x_seen: Optional[torch.Tensor] = None,
                     fallback: Optional[FloatFallback] = None, binary_first: bool = False):
  """x [T, B, H, W, Cin] (or [B, T, ...] with time_major=False) ->
  (u_T [B, OH, OW, Cout] | None, spikes [T, B, OH/pool, OW/pool, Cout]).
  x_max: the largest input value expected (a hint, snnqp.h); x_seen: eight int32 device words
  that receive the largest uint8 input value the launch met and the chunk counts by maximum.
  fallback: float32 input into integer codes (FloatFallback): the integer launch is followed by
  the predicated float32 one into the same outputs.
  binary_first: uint8 / float32 frames of the 2-channel event layer that are EXPECTED to be binary
  (ops.CountHint.binary_so_far): one checked pass packs them to bits, the event layer runs its
  bit-packed variant on 1/8 (1/32) of the bytes, and a predicated launch on the frames as they are
  redoes the block iff a value was not 0 or 1 (snnqp_pack_frames_checked,
  snnqp_conv_lif_forward_pred) -- same results whatever the frames hold, nothing read back."""
  xt, in_type = _in_desc(x)
  xt = xt.contiguous()
  _require_gpu(xt, weight.w, u0)
  x_flags = None
  if in_type == L.F32 and weight.is_int:
    if fallback is None:
      raise ValueError("float32 input into integer codes needs a FloatFallback (the float32 kernel)")
    x_flags = torch.empty(1, dtype=torch.int32, device=xt.device)
  if x_seen is not None:
    assert x_seen.dtype == torch.int32 and x_seen.numel() >= 8 and x_seen.is_cuda, \
        "x_seen: eight int32 device words (snnqp.h)"
  T, B = (xt.shape[0], xt.shape[1]) if time_major else (xt.shape[1], xt.shape[0])
  if isinstance(x, PackedFrames):        # [T, B, words / bytes of a frame]
    assert (x.H, x.W, 2) == (geom.H, geom.W, geom.Cin) and xt.ndim == 3, (x.shape, geom)
    unit = xt.shape[-1]
  else:
    unit = geom.H * geom.W * (xt.shape[-1])
  xs_t, xs_b = _tb_strides(xt, T, B, time_major, unit)
  OH, OW = geom.out_hw()
  dev = xt.device
  u_out = torch.empty((B, OH, OW, geom.Cout), dtype=torch.float32, device=dev) \
      if want_u else None
  if u0 is not None:
    u0 = _f32c(u0)
    assert tuple(u0.shape) == (B, OH, OW, geom.Cout), (u0.shape, (B, OH, OW, geom.Cout))
  oshape = (T, B, OH // pool, OW // pool)
  if packed_out:
    s = torch.empty(oshape + ((geom.Cout + 31) // 32,), dtype=torch.int32, device=dev)
  else:
    s = torch.empty(oshape + (geom.Cout,), dtype=torch.float32, device=dev)
  g, w, n = geom.struct(), weight.struct(), neuron.struct()
  b = bn.struct() if bn is not None else None
  tag = "conv%dx%d[%dx%dx%d->%d]" % (geom.KH, geom.KW, geom.H, geom.W, geom.Cin, geom.Cout)
  if _PROFILE is not None and in_type == L.BITS and weight.is_int and tag not in PROFILE_NOTES:
    PROFILE_NOTES[tag] = {"dequant": conv_dequant_form(weight, neuron)}
  speculate = (binary_first and not isinstance(x, (PackedFrames, PackedSpikes)) and in_type in (L.U8, L.F32)
               and geom.Cin == 2 and weight.is_int and impl != L.IMPL_GENERIC and xt.ndim == 5)
  with _timed(tag):
    if speculate:
      pf, not_binary = pack_frames_checked(xt)
      ps_t, ps_b = _tb_strides(pf.data, T, B, time_major, pf.data.shape[-1])
      L.check(L.lib().snnqp_conv_lif_forward(
          _ptr(pf.data), L.EV1, ps_t, ps_b, T, B, ctypes.byref(g), ctypes.byref(w),
          _ptr(weight.wt), ctypes.byref(b) if b is not None else None, ctypes.byref(n),
          _ptr(u0), _ptr(u_out), _ptr(s), L.BITS if packed_out else L.F32, pool, impl,
          1, None, None, _stream()))
      # ... and the frames as they are, iff a value was not 0 or 1
      L.check(L.lib().snnqp_conv_lif_forward_pred(
          _ptr(not_binary), _ptr(xt), in_type, xs_t, xs_b, T, B, ctypes.byref(g), ctypes.byref(w),
          _ptr(weight.wt), ctypes.byref(b) if b is not None else None, ctypes.byref(n),
          _ptr(u0), _ptr(u_out), _ptr(s), L.BITS if packed_out else L.F32, pool,
          int(x_max), _ptr(x_seen), _ptr(x_flags), _stream()))
    else:
      L.check(L.lib().snnqp_conv_lif_forward(
          _ptr(xt), in_type, xs_t, xs_b, T, B, ctypes.byref(g), ctypes.byref(w),
          _ptr(weight.wt), ctypes.byref(b) if b is not None else None, ctypes.byref(n),
          _ptr(u0), _ptr(u_out), _ptr(s), L.BITS if packed_out else L.F32, pool, impl,
          int(x_max), _ptr(x_seen), _ptr(x_flags), _stream()))
  if fallback is not None:
    # the same block on the float32 kernel, executed only if the word says so (snnqp.h)
    xf = _f32c(xt if fallback.x is None else fallback.x)
    pred = x_flags if fallback.pred is None else fallback.pred
    fw = fallback.weight.struct()
    fs_t, fs_b = _tb_strides(xf, T, B, time_major, geom.H * geom.W * geom.Cin)
    L.check(L.lib().snnqp_conv_lif_forward_if(
        _ptr(pred), _ptr(xf), L.F32, fs_t, fs_b, T, B, ctypes.byref(g), ctypes.byref(fw),
        ctypes.byref(b) if b is not None else None, ctypes.byref(n), _ptr(u0), _ptr(u_out), _ptr(s),
        L.BITS if packed_out else L.F32, pool, _stream()))
  return u_out, (PackedSpikes(s, geom.Cout) if packed_out else s)


def dense_lif_forward(x, weight: Weight, K: int, N: int, neuron: Neuron,
                      bn: Optional[BnCoeffs] = None, u0: Optional[torch.Tensor] = None,
                      want_u: bool = True, packed_out: bool = False,
                      impl: int = L.IMPL_AUTO, time_major: bool = True,
                      fallback: Optional[FloatFallback] = None):
  """x [T, B, K] -> (u_T [B, N] | None, spikes [T, B, N]).  fallback: as conv_lif_forward."""
  xt, in_type = _in_desc(x)
  xt = xt.contiguous()
  _require_gpu(xt, weight.w, u0)
  x_flags = None
  if in_type == L.F32 and weight.is_int:
    if fallback is None:
      raise ValueError("float32 rows into integer codes need a FloatFallback (the float32 kernel)")
    x_flags = torch.empty(1, dtype=torch.int32, device=xt.device)
  T, B = (xt.shape[0], xt.shape[1]) if time_major else (xt.shape[1], xt.shape[0])
  xs_t, xs_b = _tb_strides(xt, T, B, time_major, xt.shape[-1])
  dev = xt.device
  u_out = torch.empty((B, N), dtype=torch.float32, device=dev) if want_u else None
  if u0 is not None:
    u0 = _f32c(u0)
    assert tuple(u0.shape) == (B, N)
  if packed_out:
    s = torch.empty((T, B, (N + 31) // 32), dtype=torch.int32, device=dev)
  else:
    s = torch.empty((T, B, N), dtype=torch.float32, device=dev)
  w, n = weight.struct(), neuron.struct()
  b = bn.struct() if bn is not None else None
  # a long contraction over few rows (the read-out) splits K over workgroups through a workspace
  ws = _dense_workspace(dev, int(L.lib().snnqp_dense_workspace_bytes(in_type, T, B, K, N, ctypes.byref(w)))) \
      if (packed_out and impl != L.IMPL_GENERIC) else None
  with _timed("dense[%d->%d]" % (K, N)):
    L.check(L.lib().snnqp_dense_lif_forward_ws(
        _ptr(xt), in_type, xs_t, xs_b, T, B, K, N, ctypes.byref(w), _ptr(weight.wt),
        ctypes.byref(b) if b is not None else None, ctypes.byref(n), _ptr(u0),
        _ptr(u_out), _ptr(s), L.BITS if packed_out else L.F32, impl, _ptr(x_flags),
        _ptr(ws), 0 if ws is None else ws.numel(), _stream()))
  if fallback is not None:
    xf = _f32c(xt if fallback.x is None else fallback.x)
    pred = x_flags if fallback.pred is None else fallback.pred
    fw = fallback.weight.struct()
    L.check(L.lib().snnqp_dense_lif_forward_if(
        _ptr(pred), _ptr(xf), L.F32, xs_t, xs_b, T, B, K, N, ctypes.byref(fw),
        ctypes.byref(b) if b is not None else None, ctypes.byref(n), _ptr(u0), _ptr(u_out), _ptr(s),
        L.BITS if packed_out else L.F32, _stream()))
  return u_out, (PackedSpikes(s, N) if packed_out else s)


_dense_ws = {}
# workspaces allocated inside the capture that is being recorded: linen.CapturedApply installs a
# list here and keeps it, so that the block is not handed to a later allocation of the same capture
_CAPTURE_KEEP = None


def _dense_workspace(dev, nbytes: int):
  """The workspace of a dense launch that hands partial results over between workgroups
  (snnqp.h: used by one launch at a time; the call zeroes its tickets on the stream), or None.

  Eager launches share one per (device, stream) -- launches of one stream run one after the
  other -- grown as needed.  A launch that is being captured into a hipGraph gets a workspace of
  its OWN, allocated inside the capture, i.e. from the graph's private memory pool: the address
  baked into the graph lives exactly as long as the graph, is never handed to an eager launch and
  never to another graph (two live captures replayed on two streams cannot meet in one).  The
  capture object keeps the tensor (_CAPTURE_KEEP), so no later allocation of the same capture gets
  the block either."""
  if nbytes <= 0:
    return None
  if torch.cuda.is_current_stream_capturing():
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if _CAPTURE_KEEP is not None:
      _CAPTURE_KEEP.append(ws)
    return ws
  key = (torch.device(dev).index, torch.cuda.current_stream(dev).cuda_stream)
  ws = _dense_ws.get(key)
  if ws is None or ws.numel() < nbytes:
    ws = _dense_ws[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
  return ws


def dense_head_forward(x, w1: Weight, K: int, N1: int, nrn1: Neuron, w2: Weight, N2: int,
                       nrn2: Neuron, group: int = 10, want_s1: bool = False,
                       want_s2: bool = False, time_major: bool = True,
                       fallback: Optional[FloatFallback] = None):
  """The dense head as ONE launch (snnqp_dense_head_forward; examples/tcja/models.py:200-255):
  x [T, B, K] uint8, PackedSpikes or float32 (the rows as the reference holds them, staged in
  place; `fallback` = the float32 kernel of the FIRST block) -> (logits [B, N2 // group], hidden
  raster | None, output raster | None).  Raises SnnqpError(EUNSUPPORTED) when the head does not fit the fused
  kernel; the caller then runs the two blocks and the vote one by one."""
  xt, in_type = _in_desc(x)
  xt = xt.contiguous()
  _require_gpu(xt, w1.w, w2.w)
  T, B = (xt.shape[0], xt.shape[1]) if time_major else (xt.shape[1], xt.shape[0])
  xs_t, xs_b = _tb_strides(xt, T, B, time_major, xt.shape[-1])
  dev = xt.device
  if N2 % group:
    raise ValueError("vote: %d features not divisible by group %d" % (N2, group))
  logits = torch.empty((B, N2 // group), dtype=torch.float32, device=dev)
  x_flags = None
  if in_type == L.F32:
    if fallback is None or fallback.x is not None:
      raise ValueError("float32 rows into the fused head need a FloatFallback of the first block")
    x_flags = torch.empty(1, dtype=torch.int32, device=dev)
  elif fallback is not None:
    raise ValueError("a FloatFallback goes with float32 rows; uint8 / bit-packed rows have nothing to redo")
  redo = in_type == L.F32              # the predicated launches go through both rasters
  s1 = torch.empty((T, B, (N1 + 31) // 32), dtype=torch.int32, device=dev) if (want_s1 or redo) else None
  s2 = torch.empty((T, B, (N2 + 31) // 32), dtype=torch.int32, device=dev) if (want_s2 or redo) else None
  a, b, n1, n2 = w1.struct(), w2.struct(), nrn1.struct(), nrn2.struct()
  # a batch that fills at most half the chip: two workgroups per tile through a workspace
  nws = _head_ws_bytes.get((T, B, N1))
  if nws is None:
    nws = _head_ws_bytes[(T, B, N1)] = int(L.lib().snnqp_dense_head_workspace_bytes(T, B, N1))
  ws = _dense_workspace(dev, nws)
  with _timed("dense_head[%d->%d->%d]" % (K, N1, N2)):
    L.check(L.lib().snnqp_dense_head_forward(
        _ptr(xt), in_type, xs_t, xs_b, T, B, K, N1, ctypes.byref(a), _ptr(w1.wt), ctypes.byref(n1),
        N2, ctypes.byref(b), _ptr(w2.wt), ctypes.byref(n2), int(group),
        _ptr(s1) if (want_s1 or redo) else None, _ptr(s2) if (want_s2 or redo) else None,
        _ptr(logits), _ptr(x_flags), _ptr(ws), 0 if ws is None else ws.numel(), _stream()))
  if redo:
    # not integer-valued after all: the first block on the float32 kernel, the second (its input
    # is a spike raster whatever produced it) on the integer direct form, the vote -- each
    # executed only if the word says so
    fw = fallback.weight.struct()
    lib = L.lib()
    L.check(lib.snnqp_dense_lif_forward_if(_ptr(x_flags), _ptr(xt), L.F32, xs_t, xs_b, T, B, K, N1,
                                           ctypes.byref(fw), None, ctypes.byref(n1), None, None, _ptr(s1),
                                           L.BITS, _stream()))
    cw1 = (N1 + 31) // 32
    L.check(lib.snnqp_dense_lif_forward_if(_ptr(x_flags), _ptr(s1), L.BITS, B * cw1, cw1, T, B, N1, N2,
                                           ctypes.byref(b), None, ctypes.byref(n2), None, None, _ptr(s2),
                                           L.BITS, _stream()))
    L.check(lib.snnqp_vote_if(_ptr(x_flags), _ptr(s2), L.BITS, T, B, N2, int(group), _ptr(logits), _stream()))
  return (logits, PackedSpikes(s1, N1) if want_s1 else None,
          PackedSpikes(s2, N2) if want_s2 else None)


_head_ws_bytes = {}


def fallback_counts(reset: bool = False) -> dict:
  """Blocks that IMPL_AUTO handed to the direct-form kernel (20-25 x slower than the MFMA
  kernels) since the library was loaded / the last reset, with the last reason."""
  conv, dense = ctypes.c_int64(), ctypes.c_int64()
  buf = ctypes.create_string_buffer(256)
  L.check(L.lib().snnqp_fallback_counts(ctypes.byref(conv), ctypes.byref(dense), buf, 256,
                                        1 if reset else 0))
  return {"conv_blocks": conv.value, "dense_blocks": dense.value,
          "last_reason": buf.value.decode("utf-8", "replace")}


def device_status(device=None, reset: bool = False) -> int:
  """Codes the kernels of `device` have reported into its status word (_lib.STATUS_*; 0: none) --
  a launch whose bookkeeping did not add up.  While it is not zero the fused block calls raise
  SnnqpError(EHIP); reset=True clears it (after the caller has dealt with the suspect results)."""
  dev = torch.cuda.current_device() if device is None else torch.device(device).index
  code = ctypes.c_uint32()
  L.check(L.lib().snnqp_device_status(int(dev or 0), ctypes.byref(code), 1 if reset else 0))
  return int(code.value)


def workqueue_stats(reset: bool = False) -> dict:
  """Conv launches that walked their patches statically (no capture slot left / eager slot still
  in flight) and bit-input conv launches that ran the arithmetic dequantisation because the
  device did not pass (or, under capture, had not yet run) the denormal probe of the table form."""
  a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
  L.check(L.lib().snnqp_workqueue_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), 1 if reset else 0))
  return {"captured_static_walks": a.value, "busy_static_walks": b.value, "dequant_table_fallbacks": c.value}


def workqueue_capture_mark(device) -> int:
  m = ctypes.c_int64()
  L.check(L.lib().snnqp_workqueue_capture_mark(int(torch.device(device).index or 0), ctypes.byref(m)))
  return int(m.value)


def workqueue_capture_release(device, begin: int, end: int):
  L.check(L.lib().snnqp_workqueue_capture_release(int(torch.device(device).index or 0), int(begin), int(end)))


DQ_FORMS = {1: "arith", 2: "one", 3: "table"}


def conv_dequant_form(w: Weight, nrn: Neuron) -> str:
  """How the bit-input 3x3 MFMA kernels dequantise with these weights and this neuron
  (snnqp_conv_dequant_form): "arith" (three instructions per value), "one" (L == 1: a
  multiply) or "table" (the accumulator's bit pattern addresses an LDS table)."""
  ws, ns = w.struct(), nrn.struct()
  rc = L.lib().snnqp_conv_dequant_form(ctypes.byref(ws), ctypes.byref(ns))
  if rc < 0:
    L.check(rc)
  return DQ_FORMS[rc]


# ---------------------------------------------------------------------------
# element-wise pieces
# ---------------------------------------------------------------------------


def lif_forward(x: torch.Tensor, neuron: Neuron, bn: Optional[BnCoeffs] = None,
                u0: Optional[torch.Tensor] = None, want_u: bool = True,
                packed_out: bool = False):
  """x float32 [T, ..., C] currents -> (u_T [..., C] | None, spikes [T, ..., C])."""
  x = _f32c(x)
  _require_gpu(x, u0)
  T, C = x.shape[0], x.shape[-1]
  R = (x.numel() // (T * C)) if T * C else 0
  u_out = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device) if want_u else None
  if u0 is not None:
    u0 = _f32c(u0)
    assert tuple(u0.shape) == tuple(x.shape[1:])
  if packed_out:
    s = torch.empty(tuple(x.shape[:-1]) + ((C + 31) // 32,), dtype=torch.int32,
                    device=x.device)
  else:
    s = torch.empty_like(x)
  n = neuron.struct()
  b = bn.struct() if bn is not None else None
  L.check(L.lib().snnqp_lif_forward(
      _ptr(x), T, R, C, ctypes.byref(b) if b is not None else None, ctypes.byref(n),
      _ptr(u0), _ptr(u_out), _ptr(s), L.BITS if packed_out else L.F32, _stream()))
  return u_out, (PackedSpikes(s, C) if packed_out else s)


def batchnorm_forward(x: torch.Tensor, bn: BnCoeffs) -> torch.Tensor:
  x = _f32c(x)
  _require_gpu(x)
  C = x.shape[-1]
  y = torch.empty_like(x)
  b = bn.struct()
  L.check(L.lib().snnqp_batchnorm_forward(_ptr(x), x.numel() // C if C else 0, C,
                                          ctypes.byref(b), _ptr(y), _stream()))
  return y


def maxpool2x2(x):
  """[..., H, W, C] -> [..., H/2, W/2, C] for float32 tensors or PackedSpikes."""
  if isinstance(x, PackedSpikes):
    bits = x.bits
    _require_gpu(bits)
    H, W = bits.shape[-3], bits.shape[-2]
    lead = tuple(bits.shape[:-3])
    NB = 1
    for d in lead:
      NB *= d
    y = torch.empty(lead + (H // 2, W // 2, bits.shape[-1]), dtype=torch.int32,
                    device=bits.device)
    L.check(L.lib().snnqp_maxpool2x2(_ptr(bits), L.BITS, NB, H, W, x.channels, _ptr(y),
                                     _stream()))
    return PackedSpikes(y, x.channels)
  x = _f32c(x)
  _require_gpu(x)
  H, W, C = x.shape[-3:]
  lead = tuple(x.shape[:-3])
  NB = 1
  for d in lead:
    NB *= d
  y = torch.empty(lead + (H // 2, W // 2, C), dtype=torch.float32, device=x.device)
  L.check(L.lib().snnqp_maxpool2x2(_ptr(x), L.F32, NB, H, W, C, _ptr(y), _stream()))
  return y


def spatial_mean(x) -> torch.Tensor:
  """[..., H, W, C] (float32 or PackedSpikes) -> float32 [..., C] (mean over H, W)."""
  if isinstance(x, PackedSpikes):
    t, typ, C = x.bits, L.BITS, x.channels
  else:
    t, typ, C = _f32c(x), L.F32, x.shape[-1]
  _require_gpu(t)
  lead = tuple(t.shape[:-3])
  HW = t.shape[-3] * t.shape[-2]
  NB = 1
  for d in lead:
    NB *= d
  y = torch.empty(lead + (C,), dtype=torch.float32, device=t.device)
  L.check(L.lib().snnqp_spatial_mean(_ptr(t), typ, NB, HW, C, _ptr(y), _stream()))
  return y


def sigmoid_gate(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
  a, b = _f32c(a), _f32c(b)
  _require_gpu(a, b)
  assert a.shape == b.shape
  g = torch.empty_like(a)
  L.check(L.lib().snnqp_sigmoid_gate(_ptr(a), _ptr(b), a.numel(), _ptr(g), _stream()))
  return g


def apply_gate(x, g: torch.Tensor) -> torch.Tensor:
  """x [..., H, W, C] * g [..., C] broadcast over H, W -> float32."""
  if isinstance(x, PackedSpikes):
    t, typ, C = x.bits, L.BITS, x.channels
  else:
    t, typ, C = _f32c(x), L.F32, x.shape[-1]
  g = _f32c(g)
  _require_gpu(t, g)
  lead = tuple(t.shape[:-3])
  H, W = t.shape[-3], t.shape[-2]
  assert tuple(g.shape) == lead + (C,), (g.shape, lead, C)
  NB = 1
  for d in lead:
    NB *= d
  y = torch.empty(lead + (H, W, C), dtype=torch.float32, device=t.device)
  L.check(L.lib().snnqp_apply_gate(_ptr(t), typ, _ptr(g), NB, H * W, C, _ptr(y), _stream()))
  return y


def events_to_frames(x, y, p, T: int, H: int, W: int, scale: float = 1.0, as_u8=True):
  """N time-ordered DVS events -> [T, H, W, 2] count frames (input_pipeline.py:142-219)."""
  x = x.to(torch.int32).contiguous()
  y = y.to(torch.int32).contiguous()
  p = p.to(torch.int32).contiguous()
  _require_gpu(x, y, p)
  counts = torch.empty((T, H, W, 2), dtype=torch.int32, device=x.device)
  u8 = torch.empty((T, H, W, 2), dtype=torch.uint8, device=x.device) if as_u8 else None
  L.check(L.lib().snnqp_events_to_frames(_ptr(x), _ptr(y), _ptr(p), x.numel(), T, H, W,
                                         float(scale), _ptr(counts), _ptr(u8), _stream()))
  return u8 if as_u8 else counts


def density(x, lead_dims: int = 2, counts: bool = False) -> torch.Tensor:
  """Fraction of non-zero activations of each leading slice (default per [T, B]),
  the probe of examples/tcja/models.py:128-142; float32 of shape x.shape[:lead_dims]
  (or the exact int32 non-zero counts with counts=True)."""
  if isinstance(x, PackedFrames):
    x = unpack_frames(x)
  if isinstance(x, GatedSpikes):
    x = x.to_dense()
  if isinstance(x, PackedSpikes):
    t, typ, C, shape = x.bits, L.BITS, x.channels, x.shape
  elif x.dtype == torch.uint8:
    t, typ, C, shape = x.contiguous(), L.U8, x.shape[-1], tuple(x.shape)
  else:
    t, typ, C, shape = _f32c(x), L.F32, x.shape[-1], tuple(x.shape)
  _require_gpu(t)
  lead = tuple(shape[:lead_dims])
  NB = 1
  for d in lead:
    NB *= d
  n = 1
  for d in shape[lead_dims:]:
    n *= d
  nnz = torch.empty(lead, dtype=torch.int32, device=t.device)
  L.check(L.lib().snnqp_density(_ptr(t), typ, NB, n, C, _ptr(nnz), _stream()))
  if counts:
    return nnz
  return nnz.to(torch.float32) / float(n)


def vote(s, group: int = 10) -> torch.Tensor:
  """spikes [T, B, N] -> logits float32 [B, N // group] (models.py:253-255)."""
  if isinstance(s, PackedSpikes):
    t, typ, N = s.bits, L.BITS, s.channels
  else:
    t, typ, N = _f32c(s), L.F32, s.shape[-1]
  _require_gpu(t)
  T, B = t.shape[0], t.shape[1]
  if N % group:
    raise ValueError("vote: %d features not divisible by group %d" % (N, group))
  out = torch.empty((B, N // group), dtype=torch.float32, device=t.device)
  L.check(L.lib().snnqp_vote(_ptr(t), typ, T, B, N, group, _ptr(out), _stream()))
  return out
