"""Diagnostic: the gated 3x3 connection alone (CextNet's conv_t_1: 8 x 8 x 128 -> 128, T = 20) timed with
HIP events:  [SNNQP_DIAG_LIB=...] python tools/gated_time.py [B] [bits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T, H, W, C, N = 20, 8, 8, 128, 128
leaf = syn.quant_leaf((3, 3, C, N), 5.0, 971, True, 0.9 if bits == 4 else 0.3)
a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, bits, a, c, float(2 ** (bits - 1) - 1), float(np.float32(c)), True),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight()
packed = pk.gated_codes()
s = ops.pack_bits((torch.rand((T, B, H, W, C), device=dev) < 0.2).to(torch.uint8))
gate = torch.sigmoid(torch.randn((T, B, C), device=dev))
x = ops.GatedSpikes(s, gate)
geom = ops.ConvGeom(H, W, C, N, 3, 3, pad=((1, 1), (1, 1)))
for _ in range(3):
  y = ops.conv_gated_forward(x, geom, w, packed)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
  y = ops.conv_gated_forward(x, geom, w, packed)
e1.record()
torch.cuda.synchronize()
print("%s conv_gated %dx%dx%d->%d B=%d T=%d %d-bit: %.4f ms per launch (checksum %.6g)" % (
    os.environ.get("SNNQP_DIAG_LIB", "product"), H, W, C, N, B, T, bits, e0.elapsed_time(e1) / n, float(y.double().sum())))
