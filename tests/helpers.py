"""Shared by the CPU and GPU tests: oracle-side views of the synthetic trees."""
import numpy as np


def qweight_of(o, leaf, bits, quantized=True):
  """oracle.QWeight from a reference-style parameter leaf."""
  a = float(leaf["DuQ_0"]["a"][0])
  quant = None
  if quantized and a != -1.0:
    quant = {"kind": "duq", "bits": bits, "a": a, "c": float(leaf["DuQ_0"]["c"][0])}
  mask = leaf.get("prune_0", {}).get("mask")
  return o.QWeight(leaf["kernel"], quant, mask)


def bn_of(variables, i):
  p = variables["params"]["BatchNorm_%d" % i]
  s = variables["batch_stats"]["BatchNorm_%d" % i]
  return dict(mean=s["mean"], var=s["var"], scale=p["scale"], bias=p["bias"])


def packbits_lastaxis(s):
  """uint8/float 0-1 array [..., C] -> uint32 words [..., ceil(C/32)] in the
  library's bit order (channel c -> bit c & 31 of word c >> 5)."""
  s = (np.asarray(s) != 0)
  C = s.shape[-1]
  CW = (C + 31) // 32
  pad = CW * 32 - C
  if pad:
    s = np.concatenate([s, np.zeros(s.shape[:-1] + (pad,), bool)], -1)
  b = np.packbits(s.reshape(s.shape[:-1] + (CW, 32)), axis=-1, bitorder="little")
  return b.view(np.uint32).reshape(s.shape[:-1] + (CW,))


def rate(s):
  return float(np.mean(np.asarray(s, dtype=np.float64)))


def pack_ev1(x):
  """uint8 0/1 frames [..., H, W, 2] -> uint32 words [..., ceil(H*W*2/32)]: the SNNQP_EV1
  wire format of include/snnqp.h restated with a plain loop-free numpy expression (element
  i = (y*W + x)*2 + p in bit i & 31 of word i >> 5)."""
  x = np.asarray(x)
  lead, n = x.shape[:-3], int(np.prod(x.shape[-3:]))
  nw = (n + 31) // 32
  flat = np.zeros(lead + (nw * 32,), np.uint64)
  flat[..., :n] = x.reshape(lead + (n,))
  w = (flat.reshape(lead + (nw, 32)) << np.arange(32, dtype=np.uint64)).sum(-1)
  return w.astype(np.uint32)


def pack_ev4(x):
  """uint8 counts <= 15 [..., H, W, 2] -> uint8 [..., H*W] (SNNQP_EV4: polarity 0 low nibble)."""
  x = np.asarray(x).astype(np.uint8)
  return (x[..., 0] + 16 * x[..., 1]).astype(np.uint8).reshape(x.shape[:-3] + (-1,))


def fp6_tiles(codes, n_pad=None):
  """int8 codes [K, N] with |c| <= 7 -> the fp6 (e2m3) MFMA tiles of snnqp_pack_codes_fp6,
  uint8 [Npad/32, ceil(K/64), 1536], restated from the layout in include/snnqp.h: lane
  l = (n & 31) + 32 h of tile (nb, ks) holds k = 64 ks + 32 h + j, value j at bits [6j, 6j+6)
  of six little-endian dwords; dwords 0..3 of the 64 lanes first (1 KiB), then dwords 4..5."""
  codes = np.asarray(codes, np.int64)
  K, N = codes.shape
  n_pad = (N + 31) // 32 * 32 if n_pad is None else n_pad
  KS = (K + 63) // 64
  full = np.zeros((KS * 64, n_pad), np.int64)
  full[:K, :N] = codes
  mag = np.array([0x00, 0x08, 0x10, 0x14, 0x18, 0x1A, 0x1C, 0x1E], np.uint64)   # e2m3 of 0..7
  e = (mag[np.abs(full)] | np.where(full < 0, 0x20, 0).astype(np.uint64))      # [KS*64, n_pad]
  out = np.zeros((n_pad // 32, KS, 1536), np.uint8)
  for nb in range(n_pad // 32):
    blk = e[:, nb * 32:(nb + 1) * 32].reshape(KS, 2, 32, 32)       # [ks, h, j, n]
    bits = np.zeros((KS, 2, 32, 192), np.uint8)                    # [ks, h, n, bit]
    for j in range(32):
      for b in range(6):
        bits[:, :, :, 6 * j + b] = ((blk[:, :, j, :] >> np.uint64(b)) & np.uint64(1)).astype(np.uint8)
    by = np.packbits(bits, axis=-1, bitorder="little")             # [ks, h, n, 24 bytes]
    lanes = by.transpose(0, 1, 2, 3).reshape(KS, 64, 24)           # lane = n + 32 h
    out[nb, :, :1024] = lanes[:, :, :16].reshape(KS, 1024)
    out[nb, :, 1024:] = lanes[:, :, 16:].reshape(KS, 512)
  return out


def input_max_bound(x):
  """The x_max hint a test hands to ops.conv_lif_forward: 1 for spikes / binary frames, 15 for
  nibble-packed counts, the maximum (at least 1) of a uint8 tensor (read on the host: tests may)."""
  from snnquantprune_amd import _lib as L, ops
  if isinstance(x, ops.PackedSpikes):
    return 1
  if isinstance(x, ops.PackedFrames):
    return 1 if x.fmt == L.EV1 else 15
  if x.dtype.is_floating_point:
    return max(1, int(x.max().item())) if x.numel() else 1
  return max(1, int(x.max().item())) if x.numel() else 1
