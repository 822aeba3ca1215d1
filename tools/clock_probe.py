"""Diagnostic: builds libsnnqp with in-kernel clock stamps (SNNQP_PROBE=1), runs
conv1 of config C3 and prints the shader clock the chip sustains inside the
kernel (delta s_memtime / delta s_memrealtime x 100 MHz).  Rebuild normally after."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SNNQP_PROBE"] = "1"
from snnquantprune_amd.csrc import build as b
b.build(force=True, verbose=False)
import numpy as np, torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
CONV0 = len(sys.argv) > 2 and sys.argv[2] == "conv0"
CIN, HW = (2, 128) if CONV0 else (128, 64)
leaf = syn.quant_leaf((3, 3, CIN, 128), 5.0, 1, True, 0.9)
a = float(leaf["DuQ_0"]["a"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight_mfma(128)
x = (torch.rand((T, B, HW, HW, CIN), device=dev) < 0.15).to(torch.uint8)
if not CONV0:
  x = ops.pack_bits(x)
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
g = ops.ConvGeom(HW, HW, CIN, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
for _ in range(3):
  ops.conv_lif_forward(x, g, w, nrn, packed_out=True, pool=2, want_u=False, x_max=1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 4)()
L.lib().snnqp_debug_read_probe.argtypes = [ctypes.c_void_p]
rc = L.lib().snnqp_debug_read_probe(out)
if CONV0:
  print("rc", rc, "total cycles %d; staging %d (%.0f%%), compute %d (%.0f%%), flush %d (%.0f%%)" % (
      out[0], out[1], 100.0 * out[1] / out[0], out[2], 100.0 * out[2] / out[0], out[3],
      100.0 * out[3] / out[0]))
else:
  print("rc", rc, "shader cycles", out[0], "realtime ticks (100 MHz)", out[1],
        "-> clock %.3f GHz, kernel %.3f ms" % (out[0] / out[1] * 0.1, out[1] / 1e5))
