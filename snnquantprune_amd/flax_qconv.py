"""QuantConv -- host-side mirror of the reference's ``flax_qconv.py:46-188``.

N-D (here 1-D, 2-D and 3-D) convolution, channels-last inputs, `kernel` with the spatial axes
first (HWIO / DHWIO), string or explicit padding, strides, input / kernel dilation, feature groups;
weights pass through the configured quantiser and the prune mask.  The arithmetic runs
in libsnnqp (csrc/generic_block.hip; csrc/conv3x3_bits.hip / conv3x3_u8c2.hip when fused with the
neuron in SpikingBlock; 3-D kernels on the direct-form kernel only: no shipped model has one).
"""

from __future__ import annotations

from typing import Any, Callable, Iterable, Optional, Tuple, Union

import torch

from . import linen as nn
from . import ops
from . import packing
from .flax_qdense import (add_bias, default_kernel_init, packed_kernel_of,
                          quantized_bias)


def padtype_to_pads(in_shape, window_shape, window_strides, padding):
  """String padding -> explicit (lo, hi) pairs (jax.lax.padtype_to_pads, used at
  flax_qconv.py:131-144).  SAME: out = ceil(in / stride), total padding split
  lo = total // 2, hi = total - lo."""
  p = padding.upper()
  if p == "SAME":
    pads = []
    for i, k, s in zip(in_shape, window_shape, window_strides):
      out = -(-i // s)
      total = max((out - 1) * s + k - i, 0)
      pads.append((total // 2, total - total // 2))
    return pads
  if p == "VALID":
    return [(0, 0)] * len(in_shape)
  raise ValueError("Unknown padding type: %r" % (padding,))


class QuantConv(nn.Module):
  """Convolution module (see flax_qconv.py:46-75 for the field meanings)."""
  features: int
  kernel_size: Union[int, Iterable[int]]
  strides: Optional[Iterable[int]] = None
  padding: Union[str, Iterable[Tuple[int, int]]] = "SAME"
  input_dilation: Optional[Iterable[int]] = None
  kernel_dilation: Optional[Iterable[int]] = None
  feature_group_count: int = 1
  use_bias: bool = True
  dtype: Any = torch.float32
  precision: Any = None
  kernel_init: Callable = default_kernel_init
  bias_init: Callable = nn.zeros
  config: dict = None
  bits: int = 8
  quant_act_sign: bool = True
  g_scale: float = 0.

  def _ksize(self):
    if isinstance(self.kernel_size, int):
      return (self.kernel_size,)
    return tuple(int(k) for k in self.kernel_size)

  def geometry(self, spatial, in_features) -> ops.ConvGeom:
    """Resolves strides / padding / dilations for an input of spatial shape
    `spatial` (flax_qconv.py:114-144)."""
    ks = self._ksize()
    nsp = len(ks)
    if nsp not in (1, 2, 3):
      raise NotImplementedError("QuantConv supports 1-D, 2-D and 3-D convolutions")
    if len(spatial) != nsp:
      raise ValueError("input has %d spatial dims, kernel has %d" % (len(spatial), nsp))
    strides = tuple(self.strides) if self.strides else (1,) * nsp
    assert in_features % self.feature_group_count == 0     # flax_qconv.py:117
    in_dil = tuple(self.input_dilation) if self.input_dilation else (1,) * nsp
    k_dil = tuple(self.kernel_dilation) if self.kernel_dilation else (1,) * nsp
    if isinstance(self.padding, str):
      # the reference resolves string padding with rhs_dilation = 1 (:128-144)
      pads = padtype_to_pads(spatial, ks, strides, self.padding)
    else:
      pads = [(int(lo), int(hi)) for lo, hi in self.padding]
    if nsp == 3:
      return ops.Conv3dGeom(spatial[0], spatial[1], spatial[2], in_features, self.features, ks[0], ks[1], ks[2],
                            tuple(strides), tuple(tuple(p) for p in pads), tuple(in_dil), tuple(k_dil),
                            self.feature_group_count)
    if nsp == 1:
      return ops.ConvGeom(1, spatial[0], in_features, self.features, 1, ks[0],
                          (1, strides[0]), ((0, 0), tuple(pads[0])),
                          (1, in_dil[0]), (1, k_dil[0]), self.feature_group_count)
    return ops.ConvGeom(spatial[0], spatial[1], in_features, self.features, ks[0],
                        ks[1], strides, (tuple(pads[0]), tuple(pads[1])), in_dil,
                        k_dil, self.feature_group_count)

  @nn.compact_method
  def packed_kernel(self, in_features: int) -> packing.PackedKernel:
    nn.check_compute_dtype(self.dtype, "QuantConv")
    assert in_features % self.feature_group_count == 0     # flax_qconv.py:117
    kshape = self._ksize() + (in_features // self.feature_group_count, self.features)
    kernel = self.param("kernel", self.kernel_init, kshape)
    return packed_kernel_of(self, kernel)

  def out_shape(self, in_shape):
    """Output shape for an input [B, spatial..., Cin] (no batch-less inputs)."""
    nsp = len(self._ksize())
    g = self.geometry(tuple(in_shape[-nsp - 1:-1]), in_shape[-1])
    if nsp == 3:
      return tuple(in_shape[:-nsp - 1]) + tuple(g.out_dhw()) + (self.features,)
    oh, ow = g.out_hw()
    sp = (ow,) if nsp == 1 else (oh, ow)
    return tuple(in_shape[:-nsp - 1]) + sp + (self.features,)

  def __call__(self, inputs, rng: Any = None):
    x, integer = packing.prepare_input(inputs)
    if isinstance(x, ops.PackedFrames):                   # packed model input: the connection
      x = x.to_u8()                                       # alone reads plain uint8 frames
    nsp = len(self._ksize())
    is_single = False
    if x.ndim == nsp + 1:                                 # flax_qconv.py:109-112
      is_single = True
      x = x.reshape_leading(1, *x.shape[:-1]) if isinstance(x, ops.PackedSpikes) \
          else x.unsqueeze(0)
    cin = x.shape[-1]
    pk = self.packed_kernel(cin)
    g = self.geometry(tuple(x.shape[1:-1]), cin)
    w = pk.int_weight() if integer else None
    if w is None:
      w = pk.float_weight()
    nb = x.shape[0]
    if nsp == 3:
      y = self._call3d(x, integer, g, w, pk)
      if is_single:
        y = y.squeeze(0)
      if self.use_bias:
        y = add_bias(y, quantized_bias(self, pk.kernel))
      return y
    if isinstance(x, ops.PackedSpikes):
      x4 = x.reshape_leading(nb, g.H, g.W)
    else:
      x4 = x.reshape(nb, g.H, g.W, cin)
    if integer is packing.SPECULATE and w.is_int:     # float32 that may hold integers: decided on the device
      y = ops.conv_forward_speculative(x4, g, w, pk.float_weight())
    else:
      y = ops.conv_forward(x4, g, w)
    if nsp == 1:
      y = y.reshape(nb, y.shape[2], self.features)
    if is_single:
      y = y.squeeze(0)
    if self.use_bias:
      y = add_bias(y, quantized_bias(self, pk.kernel))
    return y

  def _call3d(self, x, integer, g, w, pk):
    """[NB, D, H, W, Cin] -> float32 [NB, OD, OH, OW, Cout] on the direct-form kernel."""
    if integer is packing.SPECULATE and w.is_int:     # float32 that may hold integers: decided on the device
      x8, pred = ops.narrow_f32_async(x)
      y = ops.conv3d_lif_forward(x8, g, w)
      return ops.conv3d_lif_forward(x, g, pk.float_weight(), pred=pred, out=y)
    return ops.conv3d_lif_forward(x, g, w)
