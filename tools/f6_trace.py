"""Diagnostic: in-kernel shader-clock stamps of one leader and one follower wave of
the fp6 conv kernel inside one timestep (library built with -DSNNQP_F6_TRACE)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B, T, HW = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 20, 64
leaf = syn.quant_leaf((3, 3, 128, 128), 5.0, 1, True, 0.9)
a = float(leaf["DuQ_0"]["a"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight_mfma(128)
x = ops.pack_bits((torch.rand((T, B, HW, HW, 128), device=dev) < 0.15).to(torch.uint8))
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
g = ops.ConvGeom(HW, HW, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
for _ in range(2):
  ops.conv_lif_forward(x, g, w, nrn, packed_out=True, pool=2, want_u=False, x_max=1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
L.lib().snnqp_debug_read_f6_trace.argtypes = [ctypes.c_void_p]
print("rc", L.lib().snnqp_debug_read_f6_trace(out))
t0 = min(out[0], out[8])
names = ["start", "staged-begin", "fused", "staged-end", "barrier"]
for r, nm in ((0, "wave 0"), (1, "wave 4")):
  print(nm, " ".join("%s=%d" % (names[i], out[r * 8 + i] - t0) for i in range(5)))
