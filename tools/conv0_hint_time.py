"""conv0 (128x128x2 -> 128, B = 1024, T = 20) on count frames under every hint, and on binary
frames: python tools/conv0_hint_time.py [u8|ev4]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import _lib as L, linen as nn, ops, synthetic as syn, packing
from snnquantprune_amd.quant import QuantDesc
fmt = sys.argv[1] if len(sys.argv) > 1 else "u8"
dev = torch.device("cuda:0")
B, T = 1024, 20
v = syn.conv_net_variables(prune_p=0.9, out=110, random_bn=True)
leaf = v["params"]["QuantConv_0"]
a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, c, 7.0, c),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight_mfma(128)
print("abs_sum_max", w.abs_sum_max, "code_max", w.code_max, "ch_stack_max", w.ch_stack_max)
from tests.helpers import bn_of
bn = bn_of(v, 0)
mul = (np.float32(1) / np.sqrt(bn["var"] + np.float32(1e-5))) * bn["scale"]
bnc = ops.BnCoeffs(torch.from_numpy(bn["mean"]).to(dev), torch.from_numpy(mul.astype(np.float32)).to(dev),
                   torch.from_numpy(bn["bias"]).to(dev))
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
g = ops.ConvGeom(128, 128, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
gen = torch.Generator(device=dev); gen.manual_seed(1)
counts = torch.poisson(torch.full((B, T, 128, 128, 2), 0.1, device=dev), generator=gen).clamp_(max=15).to(torch.uint8)
binary = (counts > 0).to(torch.uint8)
def run(x, hint, n=5):
  xin = ops.pack_frames(x, L.EV4) if fmt == "ev4" else x
  seen = torch.zeros(8, dtype=torch.int32, device=dev)
  for _ in range(2):
    ops.conv_lif_forward(xin, g, w, nrn, bn=bnc, want_u=False, packed_out=True, pool=2, impl=L.IMPL_MFMA,
                         time_major=False, x_max=hint, x_seen=seen)
  torch.cuda.synchronize()
  seen.zero_()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    ops.conv_lif_forward(xin, g, w, nrn, bn=bnc, want_u=False, packed_out=True, pool=2, impl=L.IMPL_MFMA,
                         time_major=False, x_max=hint, x_seen=seen)
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n, (seen.cpu().numpy() // n).tolist()
for name, x in (("binary", binary), ("counts", counts)):
  for hint in (1, 2, 3, 4, 5, 7, 15):
    ms, st = run(x, hint)
    print("%-7s hint %2d: %.3f ms   images by max (<=1, 2, <=7, <=31, >31): %s" % (name, hint, ms, st[1:6]))
