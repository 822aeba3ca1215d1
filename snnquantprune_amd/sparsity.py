"""Per-layer weight / activation densities and the workload tables the reference exports
for its accelerator model -- mirror of examples/sparsity.py:109-122 (weight density),
:143-170 (accumulating the sown probes over the evaluation batches) and :172-300 (the two
`workload_*_{mean,min}.txt` files).

  model.apply(variables, x, ..., mutable=['intermediates'])   with config.density_probes = True
      sows `conv_<i>_{inpt,out}_{min,mean}`, `conv_t_<i>_*`, `conv_tcja{1,2}_<i>_*`,
      `dense{1,2}_*` (models.py; one kernel launch per probe: snnqp_density)
  weight_density(params, config)      {layer: fraction of non-zero fake-quantised weights}
  ProbeAccumulator                    stacks the sown scalars of successive batches
  workload_tables(...)                the rows of both files; write_workload(...) writes them

Columns: name, weights, inputs, outputs, T, C, M, P, Q, R, S, HS, WS -- the geometry columns
follow the reference's own conventions (its literals for DVS128 at sparsity.py:172-230 are
reproduced exactly when frames = 20, channels = 128 and the input is 128x128x2).
"""

from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _lib as L
from . import ops
from .prune_utils import check_quant_obj

HEADER = "name,weights,inputs,outputs,T,C,M,P,Q,R,S,HS,WS\n"


def weight_density(params, config, layer_bits: Dict[str, int] = None) -> Dict[str, float]:
  """sparsity_compute (examples/sparsity.py:109-122) for every quantised layer of `params`:
  kernel * mask, DuQ with `config.quant.bits` (or `layer_bits[name]`), fraction != 0.
  Evaluated by the library's quantiser (codes * mask != 0 is the same set)."""
  out = {}
  for name, leaf in params.items():
    if not (isinstance(leaf, dict) and check_quant_obj(leaf)):
      continue
    k = leaf["kernel"]
    mask = leaf.get("prune_0", {}).get("mask")
    a = float(leaf["DuQ_0"]["a"].reshape(-1)[0]) if "DuQ_0" in leaf else -1.0
    bits = int((layer_bits or {}).get(name, config.quant.bits))
    if a == -1.0 or bits == -1:                       # pass-through quantiser (quant.py:453,469)
      fq = k.to(torch.float32) * (mask if mask is not None else 1.0)
    else:
      c = float(leaf["DuQ_0"]["c"].reshape(-1)[0])
      fq, _, _ = ops.quantize(L.Q_DUQ, k, mask, bits, a, c if c != 0.0 else a, want_fq=True)
    nnz = ops.density(fq.reshape(1, -1), lead_dims=1, counts=True)
    out[name] = float(int(nnz.item()) / fq.numel())
  return out


class ProbeAccumulator:
  """`track_intermediates.append(...)` + `stack_forest` of examples/sparsity.py:143-170."""

  def __init__(self):
    self.steps: List[Dict[str, float]] = []

  def append(self, intermediates: dict):
    row = {}
    for name, val in intermediates.items():
      if name.endswith(("_min", "_mean")):
        v = val[0] if isinstance(val, (tuple, list)) else val
        row[name] = float(v)
    self.steps.append(row)

  def stacked(self) -> Dict[str, np.ndarray]:
    names = sorted(set().union(*[set(r) for r in self.steps])) if self.steps else []
    return {n: np.array([r[n] for r in self.steps if n in r], np.float32) for n in names}


def _rows(frames: int, cin: int, channels: int, hw: Sequence[int], hidden: int, nout: int,
          full: bool):
  """(row name, layer, probe prefix, geometry columns) in the reference's order."""
  H, W = hw
  rows = [("Conv1", "QuantConv_0", "conv_0", (frames, cin, channels, H, W, 3, 3, 1, 1)),
          ("Conv2", "QuantConv_1", "conv_1", (frames, channels, channels, H // 2, W // 2, 3, 3, 1, 1)),
          ("Conv3", "QuantConv_2", "conv_2", (frames, channels, channels, H // 4, W // 4, 3, 3, 1, 1))]
  if not full:
    flat = (H // 8) * (W // 8) * channels
    rows.append(("Dense1", "QuantDense_0", "dense1", (frames, flat, nout, 1, 1, 1, 1, 1, 1)))
    return rows
  h3, w3, h4, w4 = H // 8, W // 8, H // 16, W // 16
  rows += [
      ("Conv4", "QuantConv_3", "conv_t_0", (frames, channels, channels, h3, w3, 3, 3, 1, 1)),
      ("TCJA11", "QuantConv_4", "conv_tcja1_0", (frames, frames, frames, 1, h3 * w3, 1, 4, 1, 1)),
      ("TCJA12", "QuantConv_5", "conv_tcja2_0", (channels, channels, frames, 1, h3 * w3, 1, 4, 1, 1)),
      ("Conv5", "QuantConv_6", "conv_t_1", (frames, channels, channels, h4, w4, 3, 3, 1, 1)),
      ("TCJA21", "QuantConv_7", "conv_tcja1_1", (frames, frames, frames, 1, h4 * w4, 1, 4, 1, 1)),
      ("TCJA22", "QuantConv_8", "conv_tcja2_1", (channels, channels, frames, 1, h4 * w4, 1, 4, 1, 1)),
      ("Dense1", "QuantDense_0", "dense1",
       (frames, (H // 32) * (W // 32) * channels, hidden, 1, 1, 1, 1, 1, 1)),
      ("Dense2", "QuantDense_1", "dense2", (frames, hidden, nout, 1, 1, 1, 1, 1, 1)),
  ]
  return rows


def workload_tables(layer_sparse: Dict[str, float], acc: Dict[str, np.ndarray], *, frames: int,
                    channels: int, hw=(128, 128), cin: int = 2, num_classes: int = 11,
                    full: bool = True):
  """The lines of `workload_<run>_mean.txt` and `workload_<run>_min.txt`
  (examples/sparsity.py:172-300): mean of the `*_mean` probes, max of the `*_min` probes."""
  rows = _rows(frames, cin, channels, hw, channels * 4, num_classes * 10, full)
  mean_lines, min_lines = [HEADER], [HEADER]
  for name, layer, probe, geom in rows:
    g = ",".join(str(int(v)) for v in geom)
    w = str(float(layer_sparse[layer]))
    mean_lines.append("%s,%s,%s,%s,%s\n" % (
        name, w, str(np.mean(acc[probe + "_inpt_mean"])), str(np.mean(acc[probe + "_out_mean"])), g))
    min_lines.append("%s,%s,%s,%s,%s\n" % (
        name, w, str(np.max(acc[probe + "_inpt_min"])), str(np.max(acc[probe + "_out_min"])), g))
  return mean_lines, min_lines


def write_workload(prefix: str, layer_sparse, acc, **geometry):
  """Writes `<prefix>_mean.txt` and `<prefix>_min.txt`; returns the two paths."""
  mean_lines, min_lines = workload_tables(layer_sparse, acc, **geometry)
  paths = (prefix + "_mean.txt", prefix + "_min.txt")
  for path, lines in zip(paths, (mean_lines, min_lines)):
    with open(path, "w") as f:
      f.writelines(lines)
  return paths
