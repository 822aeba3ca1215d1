"""Checkpoint import: reference parameter trees -> this package's variable tree.

Mirrors the two sources the reference loads from
(examples/tcja/tcja_load_pretrained_weights.py:39-167, examples/train_utils.py:30-41):

  * a PyTorch TCJA `.pth` state dict (`torch_state['net']`): layer-name map
    (:19-36), conv kernels OIHW -> HWIO (`transpose (2,3,1,0)`, :117), 1-D conv
    kernels reversed axes (:127), fc weights transposed (:137-139), BatchNorm
    weight/bias -> scale/bias and running_mean/var -> batch_stats mean/var (:68-107);
  * a Flax checkpoint (`flax.training.checkpoints`: a msgpack file of the
    TrainState dict): parsed directly with `msgpack` -- flax is not needed.  The
    ndarray extension format is flax.serialization's: ext code 1 wrapping
    msgpack((shape, dtype_name, raw_bytes)).

Both return ({'params': ..., 'batch_stats': ...}) with NumPy leaves; move to the
GPU with `linen.tree_from_numpy`.  Quantiser / prune leaves a checkpoint lacks are
filled with the reference's initial values (a = c = -1, mask = 1).
"""

from __future__ import annotations

from typing import Any, Dict

import msgpack
import numpy as np

# tcja_load_pretrained_weights.py:19-36
TORCH_MAP = {
    "conv.0.0": "QuantConv_0", "conv.0.1": "BatchNorm_0",
    "conv.3.0": "QuantConv_1", "conv.3.1": "BatchNorm_1",
    "conv.6.0": "QuantConv_2", "conv.6.1": "BatchNorm_2",
    "conv.9.0": "QuantConv_3", "conv.9.1": "BatchNorm_3",
    "conv.11.conv": "QuantConv_4", "conv.11.conv_c": "QuantConv_5",
    "conv.13.0": "QuantConv_6", "conv.13.1": "BatchNorm_4",
    "conv.15.conv": "QuantConv_7", "conv.15.conv_c": "QuantConv_8",
    "fc.2.0": "QuantDense_0", "fc.5.0": "QuantDense_1",
}

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3


def _ext_hook(code, data):
  if code == _EXT_NDARRAY:
    shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
    dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    if dtype_name == "bfloat16":
      raw = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
      return raw.view(np.float32).reshape(shape)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()
  if code == _EXT_NPSCALAR:
    dtype_name, buf = msgpack.unpackb(data, raw=True)
    dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    return np.frombuffer(buf, dtype=np.dtype(dtype_name))[0]
  if code == _EXT_COMPLEX:
    re, im = msgpack.unpackb(data)
    return complex(re, im)
  return msgpack.ExtType(code, data)


def msgpack_restore(blob: bytes) -> Any:
  """flax.serialization.msgpack_restore without flax."""
  def strkeys(x):
    if isinstance(x, dict):
      return {(k.decode() if isinstance(k, bytes) else k): strkeys(v) for k, v in x.items()}
    return x
  return strkeys(msgpack.unpackb(blob, ext_hook=_ext_hook, raw=True, strict_map_key=False))


def msgpack_serialize(tree: Any) -> bytes:
  """Inverse of msgpack_restore (same wire format), for tests and export."""
  def default(o):
    if isinstance(o, np.ndarray):
      payload = msgpack.packb((list(o.shape), o.dtype.name, o.tobytes()), use_bin_type=True)
      return msgpack.ExtType(_EXT_NDARRAY, payload)
    if isinstance(o, np.generic):
      payload = msgpack.packb((o.dtype.name, o.tobytes()), use_bin_type=True)
      return msgpack.ExtType(_EXT_NPSCALAR, payload)
    raise TypeError(type(o))
  return msgpack.packb(tree, default=default, use_bin_type=True)


def _fill_quant_leaves(params: Dict[str, Any]) -> Dict[str, Any]:
  out = {}
  for name, leaf in params.items():
    leaf = dict(leaf)
    if "kernel" in leaf:
      k = np.asarray(leaf["kernel"], dtype=np.float32)
      leaf["kernel"] = k
      if "DuQ_0" not in leaf:                       # quant.py:463-464, init -1
        leaf["DuQ_0"] = {"a": np.full((1,), -1, np.float32), "c": np.full((1,), -1, np.float32)}
      if "prune_0" not in leaf:                     # quant.py:489, init ones
        leaf["prune_0"] = {"mask": np.ones(k.shape, np.float32)}
    out[name] = leaf
  return out


def load_flax_checkpoint(path: str) -> Dict[str, Any]:
  """Reads a `flax.training.checkpoints` file written by the reference
  (train_utils.py:34-41): {'params': {'params': ...}, 'batch_stats': ..., ...}."""
  with open(path, "rb") as f:
    state = msgpack_restore(f.read())
  params = state["params"]
  if "params" in params and isinstance(params["params"], dict):
    params = params["params"]
  as32 = lambda t: {k: (as32(v) if isinstance(v, dict) else np.asarray(v, np.float32))  # noqa: E731
                    for k, v in t.items()}
  return {"params": _fill_quant_leaves(as32(params)),
          "batch_stats": as32(state.get("batch_stats", {}))}


def from_torch_state_dict(net_state: Dict[str, Any]) -> Dict[str, Any]:
  """TCJA PyTorch weights (`torch.load(path)['net']`) -> variable tree."""
  params: Dict[str, Dict[str, Any]] = {}
  stats: Dict[str, Dict[str, Any]] = {}
  for key, value in net_state.items():
    if "num_batches_tracked" in key:
      continue
    value = np.asarray(value.detach().cpu().numpy() if hasattr(value, "detach") else value,
                       dtype=np.float32)
    parts = key.split(".")
    name = TORCH_MAP.get(".".join(parts[:3]))
    if name is None:
      continue
    if "BatchNorm" in name:
      tgt = {"weight": (params, "scale"), "bias": (params, "bias"),
             "running_mean": (stats, "mean"), "running_var": (stats, "var")}.get(parts[-1])
      if tgt is not None:
        tgt[0].setdefault(name, {})[tgt[1]] = value
      continue
    if parts[-1] != "weight":
      continue
    if "conv" in key:
      if value.ndim == 4:
        kernel = np.transpose(value, (2, 3, 1, 0))          # OIHW -> HWIO
      elif value.ndim == 3:
        kernel = np.transpose(value)                        # (O, I, K) -> (K, I, O)
      else:
        raise Exception("Unknown weight dimensions...")
    else:                                                   # fc
      kernel = value.transpose()
    params.setdefault(name, {})["kernel"] = np.ascontiguousarray(kernel)
  return {"params": _fill_quant_leaves(params), "batch_stats": stats}


def load_torch_checkpoint(path: str) -> Dict[str, Any]:
  import torch
  state = torch.load(path, map_location="cpu")
  return from_torch_state_dict(state["net"] if "net" in state else state)
