"""Neuron models and SpikingBlock -- host-side mirror of the reference's
``spiking_learning.py`` (forward pass).

The reference scans ``connection -> norm -> neuron`` over T with ``nn.scan``
(spiking_learning.py:446-462).  Here one SpikingBlock call is ONE fused HIP
launch (snnqp_conv_lif_forward / snnqp_dense_lif_forward): the connection is
stateless across t, so only the neuron update is sequential and it runs inside
the kernel with the membrane potential in registers.

Out of scope (training / unused): surrogate-gradient backward rules
(:139-241), gsis, debug, the DECOLLE neuron and block (:244-354).
"""

from __future__ import annotations

import math
from typing import Any, Callable, Optional

import numpy as np

import torch

from . import _lib as L
from . import linen as nn
from . import ops
from . import packing
from ._cache import TensorCache
from .flax_qconv import QuantConv
from .flax_qdense import QuantDense

Array = Any


# ---------------------------------------------------------------------------
# initialisers, spiking_learning.py:24-77 (host side)
# ---------------------------------------------------------------------------


def uniform(scale=1e-2, dtype=torch.float32):
  def init(key, shape, dtype=dtype):
    return (torch.rand(tuple(shape), generator=key) * scale * 2 - scale).to(
        dtype).to(nn._default_device())
  return init


def static_init(val=1.0, dtype=torch.float32):
  def init(key, shape, dtype=dtype):
    return torch.full(tuple(shape), float(val), dtype=dtype, device=nn._default_device())
  return init


def normal_shift(bias=0, scale=1e-2, no_sign_flip=True, dtype=torch.float32):
  def init(key, shape, dtype=dtype):
    x = torch.randn(tuple(shape), generator=key) * scale + bias
    if no_sign_flip:
      x = x.abs()
    return x.to(dtype).to(nn._default_device())
  return init


# ---------------------------------------------------------------------------
# spike functions: forward = Heaviside(x >= 0), spiking_learning.py:139-241.
# They differ only in their surrogate gradient; kept as config tokens.
# ---------------------------------------------------------------------------


def _make_spike_fn(name):
  def fn(x):
    return (x >= 0.0).to(x.dtype)
  fn.__name__ = name
  return fn


fast_sigmoid = _make_spike_fn("fast_sigmoid")
slayer = _make_spike_fn("slayer")
smooth_step = _make_spike_fn("smooth_step")
piecewise_linear = _make_spike_fn("piecewise_linear")
atan = _make_spike_fn("atan")


def _sigmoid_f32(x):
  """float64 expit rounded once to float32 (PLIF / LIF decay factor)."""
  x = np.asarray(x, dtype=np.float64)
  return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


_decay_cache = TensorCache(256)


class _NeuronBase(nn.Module):

  def _step(self, u, s_in):
    """One time step (u, s_in) -> (u, s), via the T = 1 scan kernel."""
    s_in = torch.as_tensor(s_in, dtype=torch.float32)
    nrn = self.neuron(s_in.shape[-1])
    u0 = None if u is None or isinstance(u, ZeroCarry) else u
    x = s_in.unsqueeze(0)
    if x.ndim == 2:                      # 1-D state: [T=1, C]
      x = x.unsqueeze(1)
      u0 = None if u0 is None else u0.unsqueeze(0)
      u_out, s = ops.lif_forward(x, nrn, u0=u0)
      return u_out[0], s[0, 0]
    u_out, s = ops.lif_forward(x, nrn, u0=u0)
    return u_out, s[0]


class parametric_leaky_IF(_NeuronBase):
  """spiking_learning.py:357-387: u += (s_in - (u - v_reset)) * sigmoid(tau)."""
  init_tau: float
  spike_fn: Callable
  v_threshold: float = 1.0
  v_reset: float = 0.0
  pre_spike_fn: Callable = None
  dtype: Any = torch.float32

  @nn.compact_method
  def neuron(self, features: int) -> ops.Neuron:
    tau = self.param("tau", static_init(-math.log(self.init_tau - 1)), (1,))
    k = float(_sigmoid_f32(tau.detach().cpu().numpy().reshape(-1)[0]))
    return ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, k, self.v_threshold, self.v_reset)

  def __call__(self, u, s_in):
    return self._step(u, s_in)


class multi_step_LIF(_NeuronBase):
  """spiking_learning.py:390-416: u += (s_in - (u - v_reset)) / tau; s = u >= v_th;
  hard reset.  No parameters."""
  tau: float
  spike_fn: Callable
  v_threshold: float = 1.0
  v_reset: float = 0.0
  pre_spike_fn: Callable = None
  dtype: Any = torch.float32

  @nn.compact_method
  def neuron(self, features: int) -> ops.Neuron:
    return ops.Neuron(L.NEURON_MULTI_STEP_LIF, float(np.float32(self.tau)),
                      self.v_threshold, self.v_reset)

  def __call__(self, u, s_in):
    return self._step(u, s_in)


class LIF(_NeuronBase):
  """spiking_learning.py:419-438: u = u * sigmoid(tau[N]) + s_in; reset where s > .5."""
  init_tau: float
  spike_fn: Callable
  v_threshold: float = 1.0
  v_reset: float = 0.0
  dtype: Any = torch.float32

  @nn.compact_method
  def neuron(self, features: int) -> ops.Neuron:
    tau = self.param("tau", uniform(self.init_tau), (int(features),))
    dec = _decay_cache.get((tau,))
    if dec is None:
      dec = _decay_cache.put(
          (tau,), None,
          torch.from_numpy(_sigmoid_f32(tau.detach().cpu().numpy())).to(tau.device))
    return ops.Neuron(L.NEURON_LIF, 0.0, self.v_threshold, self.v_reset, decay=dec)

  def __call__(self, u, s_in):
    return self._step(u, s_in)


class ZeroCarry:
  """The all-zero initial membrane state of initialize_carry
  (spiking_learning.py:464-472) without allocating it: the kernels start from
  zero registers when given no u0."""

  def __init__(self, shape, dtype=torch.float32):
    self.shape = tuple(shape)
    self.dtype = dtype

  def materialize(self, device=None):
    return torch.zeros(self.shape, dtype=self.dtype, device=device or nn._default_device())

  def __repr__(self):
    return "ZeroCarry(shape=%s)" % (self.shape,)


_flat_perm_cache = {}


def _flat_perm(c, h, w, device):
  """row_perm[k_nhwc] = k_channel_major for a [h, w, c] block flattened NHWC (built once per
  shape and device: the read-out asks for it on every step)."""
  key = (int(c), int(h), int(w), str(device))
  perm = _flat_perm_cache.get(key)
  if perm is None:
    idx = torch.arange(c * h * w, device=device).reshape(c, h, w)   # value = k_cm
    perm = _flat_perm_cache[key] = idx.permute(1, 2, 0).reshape(-1).contiguous()   # position = k_nhwc
  return perm


def fused_dense_head(block1, block2, inputs, group: int = 10, want_s1: bool = False,
                     want_s2: bool = False):
  """SpikingBlock(QuantDense) -> SpikingBlock(QuantDense) -> vote of examples/tcja/models.py:
  200-255 as ONE launch (snnqp_dense_head_forward) when both blocks are plain fusable dense blocks
  over integer inputs with int8 codes; returns (logits, s1 | None, s2 | None), or None when the
  head does not fit (the caller then runs the blocks one by one: same numbers).  The parameters
  are created in the order the two blocks would create them."""
  for blk in (block1, block2):
    if (not blk._fusable() or not isinstance(blk.connection_fn, QuantDense) or blk.norm_fn is not None
        or blk.pool != 1 or blk.impl != L.IMPL_AUTO or blk.return_state):
      return None
  if block2.batch_major_input or getattr(inputs, "flat_perm", None) is not None:
    return None
  f32 = isinstance(inputs, torch.Tensor) and inputs.dtype == torch.float32
  if f32 and not packing.AUTO_INTEGER_INPUTS:
    return None
  if not (isinstance(inputs, ops.PackedSpikes) or f32
          or (isinstance(inputs, torch.Tensor) and inputs.dtype == torch.uint8)):
    return None
  x = inputs
  if x.ndim != 3:
    return None
  tm = not block1.batch_major_input
  T = x.shape[0] if tm else x.shape[1]
  c1, c2 = block1.connection_fn, block2.connection_fn
  K, N1, N2 = x.shape[-1], c1.features, c2.features
  if not (1 <= T <= 64 and 128 < N1 <= 512 and N2 <= 128 and N2 % group == 0):
    return None
  if isinstance(x, torch.Tensor) and K % 16:
    return None
  pk1 = c1.packed_kernel(K)
  w1 = pk1.int_weight_mfma((N1 + 31) // 32 * 32)
  nrn1 = block1.neural_dynamics.neuron(N1)
  pk2 = c2.packed_kernel(N1)
  w2 = pk2.int_weight_mfma((N2 + 31) // 32 * 32)
  nrn2 = block2.neural_dynamics.neuron(N2)
  if w1 is None or w2 is None or w1.wt is None or w2.wt is None or w1.col_sum is None:
    return None
  # float32 rows (the reference's own input format) are staged in place and checked by the kernel;
  # the float32 kernel of the first block stands by (ops.FloatFallback)
  fb = ops.FloatFallback(pk1.float_weight()) if f32 else None
  try:
    out = ops.dense_head_forward(x, w1, K, N1, nrn1, w2, N2, nrn2, group=group, want_s1=want_s1,
                                 want_s2=want_s2, time_major=tm, fallback=fb)
  except L.SnnqpError as e:
    if e.code != L.EUNSUPPORTED:
      raise
    return None
  fused_dense_head.last_plan = (w1, K, N1, nrn1, w2, N2, nrn2, group, tm, fb)    # models.DenseSNN caches it
  return out


fused_dense_head.last_plan = None


class SpikingBlock(nn.Module):
  """connection -> [norm] -> neuron over the leading (time) axis.

  Reference surface (spiking_learning.py:441-472):
    SpikingBlock(connection_fn, neural_dynamics, norm_fn=None)(u, inputs[T, B, ...])
      -> (u_T, spikes[T, B, ...])
  Extensions used by this package's models (defaults keep the reference's
  behaviour):
    pool          2 fuses the 2x2 max-pool that follows the block in
                  examples/tcja/models.py:145-147 (spikes come back pooled)
    return_state  False skips materialising u_T (the models discard it)
    packed        True / False forces bit-packed / float32 spikes; None: packed
                  iff the input was integer-typed (uint8 or PackedSpikes)
    impl          kernel choice, _lib.IMPL_*
    batch_major_input  inputs are [B, T, ...] (the model's input layout,
                  models.py:109 swaps axes first); the kernels read it by strides
  """
  connection_fn: Callable
  neural_dynamics: Callable
  norm_fn: Callable = None
  pool: int = 1
  return_state: bool = True
  packed: Optional[bool] = None
  impl: int = L.IMPL_AUTO
  batch_major_input: bool = False

  def _fusable(self):
    conn, nrn, norm = self.connection_fn, self.neural_dynamics, self.norm_fn
    if not isinstance(conn, (QuantDense, QuantConv)) or conn.use_bias:
      return False
    if not isinstance(nrn, _NeuronBase):
      return False
    if norm is not None:
      if not isinstance(norm, nn.BatchNorm) or not norm.use_running_average:
        return False
    return True

  @staticmethod
  def _event_layer_geometry(g: ops.ConvGeom) -> bool:
    """What conv3x3_u8c2_kernel serves (snnqp.h, IMPL_MFMA): 3x3, stride 1, pad 1, Cin = 2."""
    return (g.Cin == 2 and (g.KH, g.KW) == (3, 3) and tuple(g.stride) == (1, 1)
            and tuple(map(tuple, g.pad)) == ((1, 1), (1, 1)) and tuple(g.in_dil) == (1, 1)
            and tuple(g.k_dil) == (1, 1) and g.groups == 1)

  def __call__(self, u, inputs):
    if self.pool not in (1, 2):
      raise ValueError("pool must be 1 or 2")
    if self._fusable():
      return self._fused(u, inputs)
    return self._composed(u, inputs)

  # -- one launch ---------------------------------------------------------------
  def _fused(self, u, inputs):
    conn, norm = self.connection_fn, self.norm_fn
    if isinstance(inputs, ops.GatedSpikes):
      out = self._gated_block(u, inputs)
      if out is not None:
        return out
      inputs = inputs.to_dense()            # not a shape the gated form serves: float32 product
    if isinstance(conn, QuantConv) and len(conn._ksize()) == 3:
      return self._conv3d_block(u, inputs)
    flat = getattr(inputs, "flat_perm", None)
    x, integer = packing.prepare_input(inputs)
    spec = integer is packing.SPECULATE      # float32 that may hold integers: decided on the device
    if flat is not None:
      x.flat_perm = flat
    u0 = None if (u is None or isinstance(u, ZeroCarry)) else u
    tm = not self.batch_major_input
    cin = x.shape[-1]
    pk = conn.packed_kernel(cin)
    packed_out = bool(integer) if self.packed is None else bool(self.packed)
    is_dense = isinstance(conn, QuantDense)
    w = None
    if integer:
      if is_dense:
        n_pad = (conn.features + 31) // 32 * 32
        perm = None
        if flat is not None:
          perm = _flat_perm(*flat, device=pk.kernel.device)
        w = pk.int_weight_mfma(n_pad, perm, perm_key=flat)
        flat = None if w is not None else flat
      else:
        w = pk.int_weight_mfma((conn.features + 31) // 32 * 32)
    if w is None:
      w = pk.float_weight()
      spec = False                 # no integer codes for this layer: float32 all the way
    fb = None
    if spec:
      # Speculate on the integer kernels, with the float32 kernel of the same layer standing by
      # (ops.FloatFallback): either the integer kernel stages the float32 tensor itself -- the
      # 2-channel event layer, the wide dense kernel: the hot shapes -- or one device pass narrows
      # it (to spike bits when the layer is wide enough for the bit kernels, else to uint8 counts)
      # and reports into the predicate of the float32 launch.  Nothing is read back.
      fbw = pk.float_weight()
      T_ = x.shape[0] if tm else x.shape[1]
      if is_dense:
        # what the wide kernel stages in place (dense_wide.hip, dense_wide_unsupported): K a multiple
        # of 16 and at most 65536 (int32 sums of x - 128), more than 128 features, T <= 64, aligned
        # rows, and 32-bit byte offsets within a workgroup's 128 samples
        B_ = x.shape[1] if tm else x.shape[0]
        xs_t, xs_b = (B_ * cin, cin) if tm else (cin, T_ * cin)
        direct = (x.ndim == 3 and w.wt is not None and w.col_sum is not None and cin % 16 == 0
                  and cin <= 65536 and conn.features > 128 and T_ <= 64 and self.impl == L.IMPL_AUTO
                  and packed_out and (x.data_ptr() & 15) == 0 and x.is_contiguous()
                  and (max(T_ - 1, 0) * xs_t + 128 * xs_b + cin) * 4 < (1 << 31))
      else:
        direct = (x.ndim == 5 and self.impl != L.IMPL_GENERIC and packed_out
                  and self._event_layer_geometry(conn.geometry(tuple(x.shape[2:-1]), cin)))
      x_f32 = x

      def narrowed():
        """One device pass in front of the integer kernel (spike bits for layers wide enough for
        the bit kernels, else uint8 counts), reporting into the predicate of the float32 launch."""
        if not is_dense and cin >= 32:
          xn, pred = ops.pack_bits_checked(x_f32)
        else:
          xn, pred = ops.narrow_f32_async(x_f32)
        return xn, ops.FloatFallback(fbw, x=x_f32, pred=pred)
      if direct:
        fb = ops.FloatFallback(fbw)
      else:
        x, fb = narrowed()
        spec = False               # nothing left to retry: the integer kernels take the narrowed tensor
    if flat is not None:
      # float path: the fmaf order is the channel-major one -> reorder the data
      c, h, ww = flat
      d = x.to_dense() if isinstance(x, ops.PackedSpikes) else x.to(torch.float32)
      d = d.reshape(d.shape[0], d.shape[1], h, ww, c).permute(0, 1, 4, 2, 3)
      x = d.reshape(d.shape[0], d.shape[1], c * h * ww).contiguous()
    nrn = self.neural_dynamics.neuron(conn.features)
    bn = norm.coeffs(conn.features) if norm is not None else None

    # Packed event frames (the host feed's wire formats): the fused 3x3 event layer stages
    # them directly, bit-packed binary frames (EV1) and nibble-packed count frames (EV4); every
    # other consumer -- other geometries, the float32 and direct-form kernels -- gets the uint8
    # frames back with one device pass.
    if isinstance(x, ops.PackedFrames):
      direct = (not is_dense and w.wtype == L.W_I8
                and self.impl != L.IMPL_GENERIC and x.ndim == 5
                and self._event_layer_geometry(conn.geometry(tuple(x.shape[2:-1]), cin)))
      if not direct:
        x = ops.unpack_frames(x)

    # uint8 activations into a dense block (config C2's first layer).  The MFMA dense kernel
    # reads uint8 rows in place, any count 0..255 as x - 128 (snnqp.h, col_sum): no packing
    # pass, no inspection, nothing for the host to wait for.
    # (Rows it cannot take that way -- K not a multiple of 16, more than 64 timesteps -- run on the
    # direct-form kernel, whatever their values: nothing is inspected.)
    # float32 kernels -- the real-valued TCJA-gated blocks, and unquantised layers whatever
    # feeds them (float32, uint8 counts or packed spikes: widened on the fly) -- run the
    # connection on the f32 MFMA (the same fmaf chain as the direct-form kernel), then the
    # BatchNorm + neuron scan.  The float32 currents go through HBM, so the batch is
    # walked in slices that keep them under ~1 GiB.
    if w.wtype == L.W_F32 and self.impl == L.IMPL_AUTO and (
        isinstance(x, ops.PackedSpikes) or x.dtype in (torch.float32, torch.uint8)):
      geom = None
      if is_dense and x.ndim == 3:
        geom = ops.ConvGeom(1, 1, cin, conn.features, 1, 1)
      elif not is_dense and len(conn._ksize()) == 2 and x.ndim == 5:
        geom = conn.geometry(tuple(x.shape[2:-1]), cin)
      if geom is not None and ops.fseq_gemm_supported(geom):
        if self.pool == 2 and is_dense:
          raise ValueError("pool=2 needs a convolutional block")
        return self._float_block(x, tm, is_dense, geom, w, nrn, bn, u0, packed_out)

    if is_dense:
      if x.ndim != 3:
        raise ValueError("QuantDense block expects [T, B, K] inputs, got %s" % (x.shape,))
      try:
        u_out, s = ops.dense_lif_forward(x, w, cin, conn.features, nrn, bn=bn, u0=u0,
                                         want_u=self.return_state, packed_out=packed_out,
                                         impl=self.impl, time_major=tm, fallback=fb)
      except L.SnnqpError as e:
        # a float32 tensor the integer kernel turned out not to stage in place (a shape the
        # predicate above does not know): narrow it and launch again -- nothing has run yet
        if e.code != L.EUNSUPPORTED or not spec:
          raise
        x, fb = narrowed()
        u_out, s = ops.dense_lif_forward(x, w, cin, conn.features, nrn, bn=bn, u0=u0,
                                         want_u=self.return_state, packed_out=packed_out,
                                         impl=self.impl, time_major=tm, fallback=fb)
      if self.pool == 2:
        raise ValueError("pool=2 needs a convolutional block")
      return u_out, s

    nsp = len(conn._ksize())
    if x.ndim != nsp + 3:
      raise ValueError("QuantConv block expects [T, B, spatial..., C] inputs, got %s"
                       % (x.shape,))
    geom = conn.geometry(tuple(x.shape[2:-1]), cin)
    hint = None
    binary_first = False
    if (integer and cin == 2 and nsp == 2 and
        ((isinstance(x, torch.Tensor) and x.dtype in (torch.uint8, torch.float32)) or
         (isinstance(x, ops.PackedFrames) and x.fmt == L.EV4))):
      # the 2-channel event layer: the kernel checks its input itself and only wants to know
      # what to expect -- no inspection pass, no read-back in front of the launch
      hint = ops.count_hint(x.device)
      x_max = hint.current()
      # byte / float32 frames that have been binary so far: packed to bits in one checked pass, the
      # event layer on its bit-packed variant, the frames as they are behind it iff the check fails
      binary_first = (isinstance(x, torch.Tensor) and self.impl != L.IMPL_GENERIC and hint.binary_so_far()
                      and self._event_layer_geometry(geom) and w.is_int)
    elif isinstance(x, ops.PackedFrames):
      x_max = 1                          # EV1: binary by construction
    else:
      # spikes are 1; uint8 counts into a layer with more than two channels go to the direct-form
      # kernel whatever their values (the C side decides and counts the fallback): nothing to
      # inspect, nothing for the host to wait for
      x_max = 1 if isinstance(x, ops.PackedSpikes) else 0
    impl = self.impl
    T, B = (x.shape[0], x.shape[1]) if tm else (x.shape[1], x.shape[0])
    if nsp == 1 and not tm:
      raise NotImplementedError("batch-major input for 1-D convolution blocks")
    if nsp == 1:     # 1-D: H = 1
      if isinstance(x, ops.PackedSpikes):
        x = x.reshape_leading(T, B, 1, geom.W)
      else:
        x = x.reshape(T, B, 1, geom.W, cin)
      if u0 is not None:
        u0 = u0.unsqueeze(1)
    if (isinstance(x, ops.PackedSpikes) and w.is_int and w.min_current_bits == 0
        and 0 < w.abs_sum_max < (1 << 22) and impl != L.IMPL_GENERIC and u0 is None):
      # let the kernel fuse the membrane update where BatchNorm of every reachable
      # dequantised accumulator value proves that exact (snnqp.h, min_current_bits)
      import dataclasses
      w = dataclasses.replace(w, min_current_bits=ops.current_min_bits(
          w, bn, int(w.abs_sum_max), geom.Cout))
    x_seen = hint.seen_word() if hint is not None else None
    for attempt in (0, 1):
      try:
        u_out, s = ops.conv_lif_forward(x, geom, w, nrn, bn=bn, u0=u0,
                                        want_u=self.return_state, packed_out=packed_out,
                                        pool=self.pool, impl=impl, time_major=tm,
                                        x_max=x_max, x_seen=x_seen, fallback=fb, binary_first=binary_first)
        break
      except L.SnnqpError as e:
        if (e.code == L.EUNSUPPORTED and spec and attempt == 0 and isinstance(x, torch.Tensor)
            and x.dtype == torch.float32):
          # float32 frames the event layer does not stage in place after all (a shape the predicate
          # above does not know): narrow them and launch again -- nothing has run yet
          x, fb = narrowed()
          if nsp == 1:
            x = x.reshape(T, B, 1, geom.W, cin)
          spec, hint, x_seen, x_max, binary_first = False, None, None, 0, False
          continue
        if e.code != L.EUNSUPPORTED or self.pool != 2 or impl == L.IMPL_MFMA:
          raise
        u_out, s = ops.conv_lif_forward(x, geom, w, nrn, bn=bn, u0=u0,
                                        want_u=self.return_state, packed_out=packed_out,
                                        pool=1, impl=impl, time_major=tm, x_max=x_max,
                                        x_seen=x_seen, fallback=fb, binary_first=binary_first)
        s = ops.maxpool2x2(s)
        break
    if hint is not None:
      hint.launched()
    return self._finish_conv((u_out, s), nsp, T, B)

  @staticmethod
  def _finish_conv(out, nsp, T, B):
    u_out, s = out
    if nsp == 1:
      if isinstance(s, ops.PackedSpikes):
        s = s.reshape_leading(T, B, s.shape[3])
      else:
        s = s.reshape(T, B, s.shape[3], s.shape[4])
      if u_out is not None:
        u_out = u_out.squeeze(1)
    return u_out, s

  # -- gate x raster into a quantised 3x3 conv block: the 'gint' connection + neuron scan ----
  def _gated_block(self, u, x):
    """examples/tcja/models.py:95-97 -> :149-187 without multiplying the gate out: the nine taps of
    a channel are summed as integers, the gates applied in one float32 chain
    (ops.conv_gated_forward; oracle gated_conv); BatchNorm and the neuron follow as the scan.
    None when the block is not a shape that kernel serves."""
    conn, norm = self.connection_fn, self.norm_fn
    if self.batch_major_input or self.impl == L.IMPL_GENERIC:
      return None
    if x.flat:
      return self._gated_dense_block(u, x)
    if not isinstance(conn, QuantConv) or len(conn._ksize()) != 2:
      return None
    cin = x.shape[-1]
    geom = conn.geometry(tuple(x.shape[2:-1]), cin)
    pk = conn.packed_kernel(cin)
    w = pk.int_weight()
    if w is None or not (geom.KH == geom.KW == 3 and tuple(geom.stride) == (1, 1)
                         and tuple(map(tuple, geom.pad)) == ((1, 1), (1, 1)) and tuple(geom.in_dil) == (1, 1)
                         and tuple(geom.k_dil) == (1, 1) and geom.groups == 1 and cin in (32, 64, 96, 128)):
      return None
    packed = pk.gated_codes()
    if packed is None:
      return None
    T, B = x.shape[0], x.shape[1]
    N = geom.Cout
    nrn = self.neural_dynamics.neuron(N)
    bn = norm.coeffs(N) if norm is not None else None
    u0 = None if (u is None or isinstance(u, ZeroCarry)) else u
    packed_out = True if self.packed is None else bool(self.packed)
    per_sample = 4 * T * geom.H * geom.W * N              # bytes of float32 currents
    step = max(1, min(B, (1 << 30) // max(per_sample, 1)))
    us, ss = [], []
    for b0 in range(0, max(B, 1), step):      # (an empty batch: one empty slice)
      b1 = min(B, b0 + step)
      xs = x if (b0 == 0 and b1 == B) else ops.GatedSpikes(
          ops.PackedSpikes(x.spikes.bits[:, b0:b1].contiguous(), x.spikes.channels), x.gate[:, b0:b1].contiguous())
      y = ops.conv_gated_forward(xs, geom, w, packed)
      u_out, s = ops.lif_forward(y, nrn, bn=bn, u0=None if u0 is None else u0[b0:b1],
                                 want_u=self.return_state, packed_out=packed_out)
      del y
      if self.pool == 2:
        s = ops.maxpool2x2(s)
      us.append(u_out)
      ss.append(s)
    if len(ss) == 1:
      return us[0], ss[0]
    if isinstance(ss[0], ops.PackedSpikes):
      s = ops.PackedSpikes(torch.cat([p.bits for p in ss], 1), ss[0].channels)
    else:
      s = torch.cat(ss, 1)
    return (None if us[0] is None else torch.cat(us, 0)), s

  # -- 3-D convolution: the direct-form kernel with a depth axis -----------------------------------
  def _conv3d_block(self, u, inputs):
    """SpikingBlock(QuantConv with three spatial axes, [BatchNorm], neuron) on [T, B, D, H, W, C]
    (flax_qconv.py:93-171 inside spiking_learning.py:446-462): one launch of the direct-form
    kernel (snnqp_conv3d_lif_forward); float32 inputs speculate on the integer codes like every
    other block (narrowing pass + predicated float32 launch into the same outputs)."""
    conn, norm = self.connection_fn, self.norm_fn
    if self.pool != 1:
      raise ValueError("the fused 2x2 pool is 2-D")
    if isinstance(inputs, (ops.PackedFrames, ops.GatedSpikes)):
      raise ValueError("a 3-D QuantConv block takes tensors or PackedSpikes")
    x, integer = packing.prepare_input(inputs)
    if x.ndim != 6:
      raise ValueError("QuantConv block expects [T, B, spatial..., C] inputs, got %s" % (x.shape,))
    if self.batch_major_input:
      x = ops.PackedSpikes(x.bits.transpose(0, 1).contiguous(), x.channels) if isinstance(x, ops.PackedSpikes) \
          else x.transpose(0, 1).contiguous()
    cin = x.shape[-1]
    pk = conn.packed_kernel(cin)
    geom = conn.geometry(tuple(x.shape[2:-1]), cin)
    w = pk.int_weight() if integer else None
    spec = integer is packing.SPECULATE and w is not None
    if w is None:
      w = pk.float_weight()
    nrn = self.neural_dynamics.neuron(conn.features)
    bn = norm.coeffs(conn.features) if norm is not None else None
    u0 = None if (u is None or isinstance(u, ZeroCarry)) else u
    packed_out = bool(integer) if self.packed is None else bool(self.packed)
    if spec:
      x8, pred = ops.narrow_f32_async(x)
      out = ops.conv3d_lif_forward(x8, geom, w, nrn, bn=bn, u0=u0, want_u=self.return_state, packed_out=packed_out)
      return ops.conv3d_lif_forward(x, geom, pk.float_weight(), nrn, bn=bn, u0=u0, want_u=self.return_state,
                                    packed_out=packed_out, pred=pred, out=out)
    return ops.conv3d_lif_forward(x, geom, w, nrn, bn=bn, u0=u0, want_u=self.return_state, packed_out=packed_out)

  def _gated_dense_block(self, u, x):
    """examples/tcja/models.py:97 -> :189-190 -> :200-216: QuantDense on the channel-major
    flattening of gate x raster, the positions of a channel summed as integers and the gates
    applied in one float32 chain (ops.dense_gated_forward; oracle gated_dense).  None when the
    block is not a shape that kernel serves."""
    conn, norm = self.connection_fn, self.norm_fn
    if not isinstance(conn, QuantDense) or self.pool != 1:
      return None
    T, B, H, W, C = x.spikes.shape
    pk = conn.packed_kernel(C * H * W)
    w = pk.int_weight()
    packed = pk.gated_dense_codes(C, H * W) if w is not None else None
    if packed is None:
      return None
    N = conn.features
    nrn = self.neural_dynamics.neuron(N)
    bn = norm.coeffs(N) if norm is not None else None
    u0 = None if (u is None or isinstance(u, ZeroCarry)) else u
    packed_out = True if self.packed is None else bool(self.packed)
    y = ops.dense_gated_forward(x, w, packed)
    return ops.lif_forward(y, nrn, bn=bn, u0=u0, want_u=self.return_state, packed_out=packed_out)

  # -- float32 kernel: f32-MFMA connection + neuron scan, batch slice by batch slice ----
  def _float_block(self, x, tm, is_dense, geom, w, nrn, bn, u0, packed_out):
    T, B = (x.shape[0], x.shape[1]) if tm else (x.shape[1], x.shape[0])
    N = geom.Cout
    per_sample = 4 * T * geom.H * geom.W * N              # bytes of float32 currents
    step = max(1, min(B, (1 << 30) // max(per_sample, 1)))
    tag = "%s[f32 %s]" % ("dense" if is_dense else "conv", geom.tag())
    us, ss = [], []
    for b0 in range(0, max(B, 1), step):      # (an empty batch: one empty slice)
      b1 = min(B, b0 + step)
      if tm:
        xs = x[(slice(None), slice(b0, b1))]
      else:                                               # [B, T, ...] -> time-major slice
        xs = x[b0:b1]
        if isinstance(xs, ops.PackedSpikes):
          xs = ops.PackedSpikes(xs.bits.transpose(0, 1).contiguous(), xs.channels)
        else:
          xs = xs.transpose(0, 1)
      if isinstance(xs, torch.Tensor):
        xs = xs.contiguous()
      nb = b1 - b0
      if isinstance(xs, ops.PackedSpikes):
        xi = xs.reshape_leading(*((T * nb, 1, 1) if is_dense else (T * nb, geom.H, geom.W)))
      else:
        xi = xs.reshape((T * nb,) + ((1, 1, geom.Cin) if is_dense else (geom.H, geom.W, geom.Cin)))
      with ops._timed(tag):
        y = ops.conv_forward(xi, geom, w)
      y = y.reshape((T, nb) + ((N,) if is_dense else (geom.H, geom.W, N)))
      u0s = None if u0 is None else u0[b0:b1]
      u_out, s = ops.lif_forward(y, nrn, bn=bn, u0=u0s, want_u=self.return_state,
                                 packed_out=packed_out)
      del y
      if self.pool == 2:
        s = ops.maxpool2x2(s)
      us.append(u_out)
      ss.append(s)
    if len(ss) == 1:
      return us[0], ss[0]
    if isinstance(ss[0], ops.PackedSpikes):
      s = ops.PackedSpikes(torch.cat([p.bits for p in ss], 1), ss[0].channels)
    else:
      s = torch.cat(ss, 1)
    u_out = None if us[0] is None else torch.cat(us, 0)
    return u_out, s

  # -- arbitrary connection / norm / neuron: compose the stand-alone ops -----------
  def _composed(self, u, inputs):
    conn, norm, nrn_mod = self.connection_fn, self.norm_fn, self.neural_dynamics
    if isinstance(inputs, ops.PackedFrames):
      inputs = inputs.to_u8()
    if isinstance(inputs, ops.GatedSpikes):
      inputs = inputs.to_dense()
    if self.batch_major_input:
      inputs = inputs.transpose(0, 1)
    T = inputs.shape[0]
    xs = []
    for t in range(T):
      x = conn(inputs[t])
      if norm is not None:
        x = norm(x)
      xs.append(x)
    x = torch.stack(xs, 0)
    if isinstance(nrn_mod, _NeuronBase):
      u0 = None if (u is None or isinstance(u, ZeroCarry)) else u
      packed_out = bool(self.packed)
      u_out, s = ops.lif_forward(x, nrn_mod.neuron(x.shape[-1]), u0=u0,
                                 want_u=self.return_state, packed_out=packed_out)
    else:                       # foreign neuron callable: explicit time loop
      if isinstance(u, ZeroCarry) or u is None:
        u = torch.zeros_like(x[0])
      out = []
      for t in range(T):
        u, st = nrn_mod(u, x[t])
        out.append(st)
      u_out, s = u, torch.stack(out, 0)
    if self.pool == 2:
      s = ops.maxpool2x2(s)
    return u_out, s

  @staticmethod
  def initialize_carry(inputs, connection_fn, norm_fn=None, dtype=torch.float32):
    """Zero state shaped like norm(conn(inputs[0])) (spiking_learning.py:464-472);
    returned lazily, the connection is not run for its shape."""
    shape = tuple(inputs.shape[1:])
    if hasattr(connection_fn, "out_shape"):
      return ZeroCarry(connection_fn.out_shape(shape), dtype)
    x = connection_fn(inputs[0])
    if norm_fn is not None:
      x = norm_fn(x)
    return ZeroCarry(tuple(x.shape), dtype)
