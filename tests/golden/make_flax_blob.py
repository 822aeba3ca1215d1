"""Writes tests/golden/flax_checkpoint_tiny.msgpack: a checkpoint file in the wire format of
`flax.training.checkpoints.save_checkpoint` (what the reference writes,
examples/train_utils.py:34-41), assembled BYTE BY BYTE here -- neither the `msgpack`
package nor snnquantprune_amd.checkpoint is used to produce it, so the importer
(snnquantprune_amd/checkpoint.py) is tested against independent bytes.

    python tests/golden/make_flax_blob.py

Format (flax.serialization, flax 0.4.0): the TrainState's state dict as one msgpack map with
str keys; every ndarray is msgpack ext type 1 whose payload is itself the msgpack of the
3-tuple (shape as array of ints, dtype name as str, C-order bytes as bin); NumPy scalars are
ext type 3 with payload msgpack((dtype name, bytes)).  msgpack primitives used (spec):
  fixmap 0x80|n, map16 0xde;  fixstr 0xa0|n, str8 0xd9;  fixarray 0x90|n;
  positive fixint, uint8 0xcc, uint16 0xcd, uint32 0xce;  bin8 0xc4, bin16 0xc5, bin32 0xc6;
  fixext/ext8 0xc7, ext16 0xc8, ext32 0xc9 (length, then the type byte, then the payload).

The tree is a small CextNet (examples/tcja/models.py:31-257; channels = 32, 32x32 input,
T = 4) as the reference's train loop would checkpoint it: TrainState.params =
{'params': {QuantConv_i / BatchNorm_i / QuantDense_i ...}} with learnt-looking DuQ a != c,
prune masks absent (an unpruned run), batch_stats, step, and an optimiser leaf the importer
must ignore.  `tree()` returns the same arrays for the tests.
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flax_checkpoint_tiny.msgpack")
CHANNELS, HW, FRAMES, CLASSES = 32, 32, 4, 11
F32 = np.float32


# ---- a minimal msgpack writer (spec-level, independent of the msgpack package) ----------

def p_uint(n):
  if n < 128:
    return bytes([n])
  if n < 1 << 8:
    return b"\xcc" + struct.pack(">B", n)
  if n < 1 << 16:
    return b"\xcd" + struct.pack(">H", n)
  return b"\xce" + struct.pack(">I", n)


def p_str(s):
  b = s.encode("utf-8")
  if len(b) < 32:
    return bytes([0xA0 | len(b)]) + b
  assert len(b) < 256
  return b"\xd9" + struct.pack(">B", len(b)) + b


def p_bin(b):
  if len(b) < 1 << 8:
    return b"\xc4" + struct.pack(">B", len(b)) + b
  if len(b) < 1 << 16:
    return b"\xc5" + struct.pack(">H", len(b)) + b
  return b"\xc6" + struct.pack(">I", len(b)) + b


def p_array_header(n):
  assert n < 16
  return bytes([0x90 | n])


def p_ext(code, payload):
  n = len(payload)
  if n < 1 << 8:
    return b"\xc7" + struct.pack(">B", n) + bytes([code]) + payload
  if n < 1 << 16:
    return b"\xc8" + struct.pack(">H", n) + bytes([code]) + payload
  return b"\xc9" + struct.pack(">I", n) + bytes([code]) + payload


def p_ndarray(a):                      # flax.serialization._ndarray_to_bytes
  a = np.ascontiguousarray(a)
  payload = (p_array_header(3) + p_array_header(a.ndim) + b"".join(p_uint(d) for d in a.shape) +
             p_str(a.dtype.name) + p_bin(a.tobytes("C")))
  return p_ext(1, payload)


def p_npscalar(x):                     # flax.serialization: _MsgpackExtType.npscalar = 3
  x = np.asarray(x)
  return p_ext(3, p_array_header(2) + p_str(x.dtype.name) + p_bin(x.tobytes()))


def p_map(d):
  n = len(d)
  head = bytes([0x80 | n]) if n < 16 else b"\xde" + struct.pack(">H", n)
  out = [head]
  for k, v in d.items():
    out.append(p_str(k))
    if isinstance(v, dict):
      out.append(p_map(v))
    elif isinstance(v, np.ndarray):
      out.append(p_ndarray(v))
    elif isinstance(v, np.generic):
      out.append(p_npscalar(v))
    elif isinstance(v, int):
      out.append(p_uint(v))
    else:
      raise TypeError(type(v))
  return b"".join(out)


# ---- the tree -------------------------------------------------------------------------------

def tree():
  rng = np.random.Generator(np.random.PCG64(20261004))
  C = CHANNELS

  def kern(shape, gain):
    fan_in = int(np.prod(shape[:-1]))
    return (rng.standard_normal(shape) * (gain / np.sqrt(fan_in))).astype(F32)

  def qleaf(shape, gain):
    k = kern(shape, gain)
    s = float(np.std(k))
    return {"kernel": k,
            "DuQ_0": {"a": np.array([2.7 * s], F32), "c": np.array([2.9 * s], F32)}}

  def bn():
    return ({"scale": (1 + 0.2 * rng.standard_normal(C)).astype(F32),
             "bias": (0.1 * rng.standard_normal(C)).astype(F32)},
            {"mean": (0.1 * rng.standard_normal(C)).astype(F32),
             "var": (1 + 0.3 * rng.random(C)).astype(F32)})

  shapes = {0: ((3, 3, 2, C), 4.0), 1: ((3, 3, C, C), 5.0), 2: ((3, 3, C, C), 5.0),
            3: ((3, 3, C, C), 6.0), 4: ((4, FRAMES, FRAMES), 6.0), 5: ((4, C, C), 6.0),
            6: ((3, 3, C, C), 12.0), 7: ((4, FRAMES, FRAMES), 6.0), 8: ((4, C, C), 6.0)}
  params, stats = {}, {}
  for i in range(9):
    params["QuantConv_%d" % i] = qleaf(*shapes[i])
  for i in range(5):
    params["BatchNorm_%d" % i], stats["BatchNorm_%d" % i] = bn()
  flat = (HW // 32) * (HW // 32) * C
  params["QuantDense_0"] = qleaf((flat, 4 * C), 16.0)
  params["QuantDense_1"] = qleaf((4 * C, CLASSES * 10), 8.0)
  # insertion order of a flax state dict is alphabetical within a level
  params = {k: params[k] for k in sorted(params)}
  stats = {k: stats[k] for k in sorted(stats)}
  return {"params": params, "batch_stats": stats}


def state_dict():
  t = tree()
  return {
      "batch_stats": t["batch_stats"],
      "opt_state": {"0": {"count": np.int32(1234)},
                    "1": {"mu": {"params": {"QuantDense_1": {"kernel": np.zeros((2, 2), F32)}}}}},
      "params": {"params": t["params"]},
      "step": 1234,
      "weight_size": np.float32(0.5),
  }


def main():
  blob = p_map(state_dict())
  with open(OUT, "wb") as f:
    f.write(blob)
  print("%s: %d bytes" % (OUT, len(blob)))


if __name__ == "__main__":
  sys.exit(main())
