"""Models for the BASELINE.json configurations, built from the same blocks and
with the same call signature as the reference's CextNet
(examples/tcja/models.py:31-257):

  CextNet       the reference's full DVS128 model: 3 conv blocks, 2 conv blocks each
                followed by a TCJA attention gate, channel-major flatten, two
                dense blocks, vote (models.py:31-257)
  DenseSNN      configs C1 / C2: the two-layer head of CextNet,
                QuantDense(hidden)+LIF -> QuantDense(num_classes*10)+LIF -> vote
                (models.py:200-255)
  ConvDenseSNN  config C3: the three `for i in range(3)` conv blocks
                (QuantConv 3x3 + BatchNorm + LIF + 2x2 max-pool, models.py:111-147)
                -> channel-major flatten (:189-190) -> QuantDense + LIF (:231-246)
                -> vote (:253-255)

Variable names follow Flax auto-naming in construction order (QuantConv_i,
BatchNorm_i, QuantDense_i; tcja_load_pretrained_weights.py:19-36).
"""

from __future__ import annotations

from typing import Any, Callable

import torch

from . import linen as nn
from . import ops
from . import packing
from .flax_qconv import QuantConv
from .flax_qdense import QuantDense
from .spiking_learning import SpikingBlock, fused_dense_head


def flatten_channel_major(x):
  """transpose (T,B,C,H,W) + reshape [T, B, C*H*W]  (models.py:189-190).

  Packed spikes are not moved: the NHWC words are re-labelled and the consumer
  re-orders its weight rows (an exact re-indexing of an integer sum)."""
  if isinstance(x, ops.GatedSpikes):
    return x.flattened()
  if isinstance(x, ops.PackedSpikes):
    T, B, H, W, C = x.shape
    if C % 32:
      d = x.to_dense().permute(0, 1, 4, 2, 3).reshape(T, B, C * H * W)
      return ops.pack_bits(d.contiguous())
    out = ops.PackedSpikes(x.bits.reshape(T, B, H * W * (C // 32)), H * W * C)
    out.flat_perm = (C, H, W)
    return out
  x = x.permute(0, 1, 4, 2, 3)
  return x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3] * x.shape[4]).contiguous()   # (no -1: B may be 0)


def _layer_bits(cfg, i):
  """Bit width of the i-th quantised layer: `config.quant.layer_bits[i]` when
  given (mixed precision, BASELINE config C5), else `config.quant.bits`; every
  layer has its own `bits` attribute in the reference (flax_qconv.py:89)."""
  q = cfg.quant
  if "layer_bits" in q and q.layer_bits is not None:
    return int(q.layer_bits[i])
  return q.bits


def _require_eval(train):
  if train:
    raise NotImplementedError(
        "training (dropout, batch statistics, gradients) is out of scope: this "
        "package implements the eval forward pass (train=False)")


def _probing(mod, cfg):
  """Density probes are sown under the reference's names (models.py:45-51,128-142) when the
  caller both asks for them (config.density_probes) and made 'intermediates' mutable.  The
  reference sows them unconditionally -- under jit an unread probe costs nothing; here each
  probe is a kernel launch, and the `*_out_*` probes need the UNPOOLED raster, so the blocks
  then run with the 2x2 max-pool as a separate pass."""
  return bool(cfg.get("density_probes", False)) and mod.is_mutable_collection("intermediates")


def _sow_density(mod, name, x, lead_dims=2):
  """sparse_nums = nnz / size per leading slice; `<name>_min` is its MAXIMUM (the reference's
  naming, models.py:131-133) and `<name>_mean` its mean, float32 device scalars."""
  d = ops.density(x, lead_dims=lead_dims).reshape(-1)
  mod.sow("intermediates", name + "_min", d.max())
  mod.sow("intermediates", name + "_mean", d.mean())


def _as_input(inputs):
  if isinstance(inputs, (ops.PackedSpikes, ops.PackedFrames, ops.GatedSpikes)):
    return inputs
  x = torch.as_tensor(inputs)
  if x.dtype not in (torch.uint8, torch.float32):
    x = x.to(torch.float32)
  return x


# The dense head of a given set of parameters, prepared once: what `fused_dense_head` hands to
# the kernel depends on the parameter tensors (identity and version), the two bit widths, the
# quantiser and neuron factories of the config and the input's layout -- nothing else.  A step
# whose keys match skips the module machinery (ten modules bound and ~400 Python calls to arrive
# at the same two descriptors): config C2 is a 0.03 ms kernel, the host must not cost more.
_head_plans = None


def _head_key(mod, cfg, x, hidden, nout):
  """(tensors, extra) identifying the head's prepared launch, or None when the step is not the
  plain one (initialising, mutable collections, probes, parameters missing or of another shape)."""
  root = mod._root
  if root.initializing or root.mutable or mod._path != ():
    return None
  if isinstance(x, ops.PackedSpikes):
    if not (x.bits.is_cuda and x.ndim == 3 and x.flat_perm is None):
      return None
  elif not (isinstance(x, torch.Tensor) and x.dtype in (torch.uint8, torch.float32) and x.is_cuda
            and x.ndim == 3 and x.is_contiguous()):
    return None
  if isinstance(x, torch.Tensor) and x.dtype == torch.float32 and not packing.AUTO_INTEGER_INPUTS:
    return None
  p = root.variables.get("params")
  try:
    l0, l1 = p["QuantDense_0"], p["QuantDense_1"]
    ts = (l0["kernel"], l0["DuQ_0"]["a"], l0["DuQ_0"]["c"], l0["prune_0"]["mask"],
          l1["kernel"], l1["DuQ_0"]["a"], l1["DuQ_0"]["c"], l1["prune_0"]["mask"])
  except (KeyError, TypeError):
    return None
  q = cfg.quant
  extra = (type(x).__name__, str(getattr(x, "dtype", "")), tuple(x.shape), hidden, nout, _layer_bits(cfg, 0), _layer_bits(cfg, 1), id(q.get("weight")),
           float(q.prune_percentage) >= 0.0, id(cfg.neuron_dynamics))
  return ts, extra


class DenseSNN(nn.Module):
  """inputs [B, T, K] -> logits [B, num_classes] (configs C1 / C2)."""
  num_classes: int = 11
  dtype: Any = torch.float32
  config: dict = nn.FrozenConfigDict({})

  def __call__(self, inputs, trgt=None, train: bool = False, rng: Any = None,
               u_state=None, online=False):
    _require_eval(train)
    cfg = self.config
    x = _as_input(inputs)
    hidden = cfg.hidden if "hidden" in cfg else cfg.channels * 2 * 2
    probe = _probing(self, cfg)
    global _head_plans
    key = None if probe else _head_key(self, cfg, x, hidden, self.num_classes * 10)
    if key is not None:
      if _head_plans is None:
        from ._cache import TensorCache
        _head_plans = TensorCache(32)
      plan = _head_plans.get(*key)
      if plan is not None:
        w1, K, N1, nrn1, w2, N2, nrn2, group, tm, fb, _refs = plan
        return ops.dense_head_forward(x, w1, K, N1, nrn1, w2, N2, nrn2, group=group, time_major=tm,
                                      fallback=fb)[0], None
    layer = SpikingBlock(
        connection_fn=QuantDense(hidden, use_bias=False, dtype=self.dtype,
                                 config=cfg.quant, bits=_layer_bits(cfg, 0),
                                 g_scale=cfg.quant.g_scale),
        neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype),
        return_state=False, batch_major_input=True)
    layer2 = SpikingBlock(
        connection_fn=QuantDense(self.num_classes * 10, use_bias=False,
                                 dtype=self.dtype, config=cfg.quant,
                                 bits=_layer_bits(cfg, 1), g_scale=cfg.quant.g_scale),
        neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype),
        return_state=False)
    if not probe:
      # both blocks and the vote as one launch when the head fits it (the hidden raster then
      # never leaves the CU); the output raster only when somebody collects the intermediates
      want_s2 = self.is_mutable_collection("intermediates")
      head = fused_dense_head(layer, layer2, x, 10, want_s2=want_s2)
      if head is not None:
        if want_s2:
          self.sow("intermediates", "dense2_out", head[2])
        if key is not None and fused_dense_head.last_plan is not None:
          # (the factories are kept alive with the plan: their ids are part of its key)
          _head_plans.put(key[0], key[1], fused_dense_head.last_plan + ((cfg.quant.get("weight"), cfg.neuron_dynamics),))
        return head[0], None
    if probe:
      _sow_density(self, "dense1_inpt", x)
    _, x = layer(None, x)
    if probe:
      _sow_density(self, "dense1_out", x)
      _sow_density(self, "dense2_inpt", x)
    _, x = layer2(None, x)
    self.sow("intermediates", "dense2_out", x)
    if probe:
      _sow_density(self, "dense2_out", x)
    return ops.vote(x, 10), None                       # models.py:253-255


class ConvDenseSNN(nn.Module):
  """inputs [B, T, H, W, 2] -> logits [B, num_classes] (config C3)."""
  num_classes: int = 11
  dtype: Any = torch.float32
  config: dict = nn.FrozenConfigDict({})

  def __call__(self, inputs, trgt=None, train: bool = False, rng: Any = None,
               u_state=None, online=False):
    _require_eval(train)
    cfg = self.config
    x = _as_input(inputs)
    nblocks = cfg.num_conv_blocks if "num_conv_blocks" in cfg else 3
    norm = lambda: nn.BatchNorm(use_running_average=not train, momentum=0.9,  # noqa: E731
                                epsilon=1e-5, use_bias=True, use_scale=True,
                                dtype=self.dtype)
    probe = _probing(self, cfg)
    for i in range(nblocks):
      layer = SpikingBlock(
          connection_fn=QuantConv(features=cfg.channels, kernel_size=(3, 3),
                                  padding=((1, 1), (1, 1)), use_bias=False,
                                  dtype=self.dtype, config=cfg.quant,
                                  bits=_layer_bits(cfg, i), g_scale=cfg.quant.g_scale),
          neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype),
          norm_fn=norm(),
          pool=1 if probe else 2,       # the reduce_window max of models.py:145-147
          return_state=False,
          batch_major_input=(i == 0))   # models.py:109 swapaxes, done by strides
      if probe:
        _sow_density(self, "conv_%d_inpt" % i, x)
      _, x = layer(None, x)
      if probe:
        _sow_density(self, "conv_%d_out" % i, x)
        x = ops.maxpool2x2(x)
      self.sow("intermediates", "pool%d" % i, x)
    x = flatten_channel_major(x)
    if probe:
      _sow_density(self, "dense1_inpt", x)
    layer = SpikingBlock(
        connection_fn=QuantDense(self.num_classes * 10, use_bias=False,
                                 dtype=self.dtype, config=cfg.quant,
                                 bits=_layer_bits(cfg, nblocks), g_scale=cfg.quant.g_scale),
        neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype),
        return_state=False)
    _, x = layer(None, x)
    self.sow("intermediates", "dense_out", x)
    if probe:
      _sow_density(self, "dense1_out", x)
    return ops.vote(x, 10), None


class CextNet(nn.Module):
  """TCJA-SNN, examples/tcja/models.py:31-257 (eval forward).

  inputs [B, T, H, W, 2] -> (logits [B, num_classes], None).  Variables:
  QuantConv_0..8, BatchNorm_0..4, QuantDense_0..1 in the reference's order
  (tcja_load_pretrained_weights.py:19-36).  After the first TCJA gate the
  activations are real-valued: those layers use the float (fmaf-chain) kernels
  with the fake-quantised weights, as the reference does."""
  num_classes: int = 11
  dtype: Any = torch.float32
  config: dict = nn.FrozenConfigDict({})

  def __call__(self, inputs, trgt=None, train: bool = False, rng: Any = None,
               u_state=None, online=False):
    _require_eval(train)
    cfg = self.config

    def qconv1d(features, x):
      return QuantConv(features=features, kernel_size=[4], padding="SAME", use_bias=False,
                       dtype=self.dtype, config=cfg.quant, bits=cfg.quant.bits,
                       g_scale=cfg.quant.g_scale)(x)

    probe = _probing(self, cfg)

    def TCJA(x_seq, i=0, gated=False):                  # models.py:41-99
      T, C = x_seq.shape[0], x_seq.shape[-1]
      m = ops.spatial_mean(x_seq)                       # [T, B, C]
      x = m.transpose(0, 1).contiguous()                # [B, T, C]
      x_c = x.transpose(1, 2).contiguous()              # [B, C, T]
      with packing.integer_inputs(False):               # channel means are real-valued
        conv_t_out = qconv1d(T, x_c)                    # [B, C, T]
        conv_c_out = qconv1d(C, x)                      # [B, T, C]
      if probe:                                         # per sample, models.py:45-91
        _sow_density(self, "conv_tcja1_%d_inpt" % i, x_c, 1)
        _sow_density(self, "conv_tcja1_%d_out" % i, conv_t_out, 1)
        _sow_density(self, "conv_tcja2_%d_inpt" % i, x, 1)
        _sow_density(self, "conv_tcja2_%d_out" % i, conv_c_out, 1)
      conv_t_out = conv_t_out.permute(2, 0, 1).contiguous()   # [T, B, C]
      conv_c_out = conv_c_out.transpose(0, 1).contiguous()    # [T, B, C]
      gate = ops.sigmoid_gate(conv_c_out, conv_t_out)
      self.sow("intermediates", "tcja_gate_%d" % i, gate)
      # The reference gates the raster and then max-pools it 2x2 (models.py:99, 145-147).
      # The gate is a sigmoid (>= 0) and constant over H, W, the raster is 0/1, so
      # max(g * s_i) == g * max(s_i) exactly: pool the spikes first (an OR of bits) and
      # gate the pooled raster -- a quarter of the float32 traffic, same numbers.
      pooled = ops.maxpool2x2(x_seq)
      if gated and isinstance(pooled, ops.PackedSpikes):
        # the next block is a quantised 3x3 conv (i = 0) or the flatten + dense block (i = 1): it
        # contracts gate x raster without the product being written (SpikingBlock._gated_block;
        # config.gated_int = False multiplies it out)
        return ops.GatedSpikes(pooled, gate)
      return ops.apply_gate(pooled, gate)   # [T, B, H/2, W/2, C] float32

    def conv_block(x, first, pool, packed=None):
      layer = SpikingBlock(
          connection_fn=QuantConv(features=cfg.channels, kernel_size=(3, 3),
                                  padding=((1, 1), (1, 1)), use_bias=False,
                                  dtype=self.dtype, config=cfg.quant, bits=cfg.quant.bits,
                                  g_scale=cfg.quant.g_scale),
          neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype),
          norm_fn=nn.BatchNorm(use_running_average=not train, momentum=0.9, epsilon=1e-5,
                               use_bias=True, use_scale=True, dtype=self.dtype),
          pool=pool, return_state=False, batch_major_input=first, packed=packed)
      return layer(None, x)[1]

    def dense_block(x, features, packed=None):
      layer = SpikingBlock(
          connection_fn=QuantDense(features, use_bias=False, dtype=self.dtype,
                                   config=cfg.quant, bits=cfg.quant.bits,
                                   g_scale=cfg.quant.g_scale),
          neural_dynamics=cfg.neuron_dynamics(dtype=self.dtype), return_state=False, packed=packed)
      return layer(None, x)[1]

    x = _as_input(inputs)
    for i in range(3):                                  # models.py:111-147
      if probe:
        _sow_density(self, "conv_%d_inpt" % i, x)
      x = conv_block(x, first=(i == 0), pool=1 if probe else 2)
      if probe:
        _sow_density(self, "conv_%d_out" % i, x)
        x = ops.maxpool2x2(x)
      self.sow("intermediates", "pool%d" % i, x)
    real_valued = False
    for i in range(2):                                  # models.py:149-187
      if probe:
        _sow_density(self, "conv_t_%d_inpt" % i, x)
      with packing.integer_inputs(not real_valued):
        # TCJA needs the unpooled raster; bit-packed also when the input is real-valued
        x = conv_block(x, first=False, pool=1, packed=True)
      self.sow("intermediates", "conv_t_%d" % i, x)
      if probe:
        _sow_density(self, "conv_t_%d_out" % i, x)
      # gated and pooled; the blocks behind the gates may take the product unmultiplied
      x = TCJA(x, i, gated=bool(cfg.get("gated_int", True)))
      real_valued = True
    x = flatten_channel_major(x)                        # models.py:189-190
    if probe:
      _sow_density(self, "dense1_inpt", x)
    with packing.integer_inputs(False):
      # real-valued inputs, but what comes out are spikes: bit-packed, so that dense2 runs on the
      # integer kernels without anybody having to look at the values first
      x = dense_block(x, cfg.channels * 2 * 2, packed=True)          # models.py:200-216
    self.sow("intermediates", "dense1_out", x)
    if probe:
      _sow_density(self, "dense1_out", x)
      _sow_density(self, "dense2_inpt", x)
    x = dense_block(x, self.num_classes * 10)           # models.py:231-246
    self.sow("intermediates", "dense2_out", x)
    if probe:
      _sow_density(self, "dense2_out", x)
    return ops.vote(x, 10), None
