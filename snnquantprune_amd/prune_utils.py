"""Pack-time inputs of the hot path: prune masks and DuQ `a`, `c` -- mirror of
examples/train_inpt_spikingjelly.py:147-223 (host side, run once per model).

  update_prune_mask     per-layer magnitude mask (:147-157)
  global masks          one magnitude threshold over all kernels in tree order
                        (:174-223, the path the shipped configs reach:
                        `prune_global = True`)
  update_quant_params   a = c = init_fn(kernel, bits, sign=True) (:159-172)

They operate on the `params` tree ({'QuantConv_0': {'kernel', 'DuQ_0', 'prune_0'},
...}) and return a new tree; leaves stay torch tensors on their device.
"""

from __future__ import annotations

import numpy as np
import torch


def check_quant_obj(x) -> bool:
  """A quantised layer's leaf dict (has 'kernel'), train_inpt_spikingjelly.py:199-204."""
  return isinstance(x, dict) and "kernel" in x


def _map_layers(fn, tree):
  if check_quant_obj(tree):
    return fn(tree)
  if isinstance(tree, dict):
    return {k: _map_layers(fn, v) for k, v in tree.items()}
  return tree


def _layers(tree, out=None):
  out = [] if out is None else out
  if check_quant_obj(tree):
    out.append(tree)
  elif isinstance(tree, dict):
    for v in tree.values():
      _layers(v, out)
  return out


def _like(arr: np.ndarray, ref: torch.Tensor):
  return torch.from_numpy(np.ascontiguousarray(arr)).to(ref.device, ref.dtype)


def update_prune_mask(params, prune_percentage: float):
  """Local (layer-wise) pruning: zero the int(numel * p) smallest |kernel|."""
  def f(x):
    k = x["kernel"].detach().cpu().numpy()
    mask = np.ones(k.shape)
    n = int(np.prod(k.shape) * prune_percentage)
    idx = np.argpartition(np.abs(k).reshape(-1), n)[:n]
    mask.reshape(-1)[idx] = 0
    y = dict(x)
    y["prune_0"] = {"mask": _like(mask, x["kernel"])}
    return y
  return _map_layers(f, params)


def update_global_prune_mask(params, prune_percentage: float):
  """Global pruning: one magnitude cut over the concatenation of all kernels."""
  layers = _layers(params)
  flat = np.concatenate([l["kernel"].detach().cpu().numpy().reshape(-1) for l in layers])
  gm = np.ones(flat.shape)
  n = int(np.prod(flat.shape) * prune_percentage)
  idx = np.argpartition(np.abs(flat), n)[:n]
  gm[idx] = 0
  off = [0]

  def f(x):
    sz = int(np.prod(x["kernel"].shape))
    local = gm[off[0]:off[0] + sz].reshape(tuple(x["kernel"].shape))
    off[0] += sz
    y = dict(x)
    y["prune_0"] = {"mask": _like(local, x["kernel"])}
    return y
  return _map_layers(f, params)


def update_quant_params(params, init_fn, bits: int):
  """DuQ_0/{a, c} = init_fn(kernel, bits=bits, sign=True), shape (1,)."""
  def f(x):
    v = init_fn(x["kernel"], bits=bits, sign=True).reshape(1).to(torch.float32)
    y = dict(x)
    y["DuQ_0"] = {"a": v.clone(), "c": v.clone()}
    return y
  return _map_layers(f, params)


def prepare_params(params, config):
  """The sequence of train_inpt_spikingjelly.py:206-230 for a `config.quant`."""
  q = config.quant
  if q.prune_percentage > 0.:
    if "prune_global" in q and q.prune_global is False:
      params = update_prune_mask(params, q.prune_percentage)
    else:
      params = update_global_prune_mask(params, q.prune_percentage)
  if "start_epoch" not in q or q.start_epoch == -1:
    params = update_quant_params(params, q.init_fn, q.bits)
  return params
