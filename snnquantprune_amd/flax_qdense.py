"""QuantDense -- host-side mirror of the reference's ``flax_qdense.py:47-106``.

Same fields, variable names (`kernel` [in, features], `DuQ_0/{a,c}`,
`prune_0/mask`, `bias`) and call signature; the contraction runs in libsnnqp
(dense = the 1x1 convolution on a 1x1 image, csrc/generic_block.hip; fused with
the neuron in SpikingBlock via snnqp_dense_lif_forward).
"""

from __future__ import annotations

from typing import Any, Callable

import torch

from . import linen as nn
from . import ops
from . import packing
from .quant import prune

default_kernel_init = nn.lecun_normal()


def weight_quantizer(module):
  """cfg.weight(bits=, g_scale=) or None, flax_qdense.py:74-82."""
  cfg = module.config
  if "weight" not in cfg:
    return None
  if module.bits is not None:
    return cfg.weight(bits=module.bits, g_scale=module.g_scale)
  return cfg.weight(g_scale=module.g_scale)


def packed_kernel_of(module, kernel):
  """Applies the module's config to `kernel`: quantiser description + prune mask
  (flax_qdense.py:74-85 / flax_qconv.py:147-156), packed once per version."""
  q = weight_quantizer(module)
  desc = q.describe(kernel) if q is not None else None
  mask = None
  if module.config.prune_percentage >= 0.:      # AttributeError if missing, as the reference
    mask = prune().get_mask(kernel.shape)
  return packing.get_packed(kernel, desc, mask)


def quantized_bias(module, kernel):
  bias = module.param("bias", module.bias_init, (module.features,))
  bias = bias.to(torch.float32)
  if "bias" in module.config:                    # flax_qdense.py:95-103
    maxabs_w = float(kernel.abs().max())
    bias = module.config.bias(bits=module.bits, g_scale=module.g_scale,
                              maxabs_w=maxabs_w)(bias)
  return bias


def add_bias(y, bias):
  """y + bias as one HIP pass (x - 0) * 1 + b, exact."""
  z = torch.zeros_like(bias)
  return ops.batchnorm_forward(y, ops.BnCoeffs(z, torch.ones_like(bias), bias.contiguous()))


class QuantDense(nn.Module):
  """A linear transformation applied over the last dimension of the input."""
  features: int
  use_bias: bool = True
  dtype: Any = torch.float32
  precision: Any = None
  kernel_init: Callable = default_kernel_init
  bias_init: Callable = nn.zeros
  config: dict = nn.FrozenConfigDict({})
  bits: int = 8
  quant_act_sign: bool = True
  g_scale: float = 0.

  def _check_dtype(self):
    nn.check_compute_dtype(self.dtype, "QuantDense")

  @nn.compact_method
  def packed_kernel(self, in_features: int) -> packing.PackedKernel:
    self._check_dtype()
    kernel = self.param("kernel", self.kernel_init, (int(in_features), self.features))
    return packed_kernel_of(self, kernel)

  def out_shape(self, in_shape):
    return tuple(in_shape[:-1]) + (self.features,)

  def __call__(self, inputs, rng: Any = None):
    x, integer = packing.prepare_input(inputs)
    K = x.shape[-1]
    pk = self.packed_kernel(K)
    lead = tuple(x.shape[:-1])
    nb = 1
    for d in lead:
      nb *= d
    w = pk.int_weight() if integer else None
    if w is None:
      w = pk.float_weight()
    geom = ops.ConvGeom(1, 1, K, self.features, 1, 1)
    x4 = x.reshape_leading(nb, 1, 1) if isinstance(x, ops.PackedSpikes) \
        else x.reshape(nb, 1, 1, K)
    if integer is packing.SPECULATE and w.is_int:     # float32 that may hold integers: decided on the device
      y = ops.conv_forward_speculative(x4, geom, w, pk.float_weight())
    else:
      y = ops.conv_forward(x4, geom, w)
    y = y.reshape(lead + (self.features,))
    if self.use_bias:
      y = add_bias(y, quantized_bias(self, pk.kernel))
    return y
