"""Diagnostic: time of the fused conv blocks vs T (per-step slope and per-patch
intercept) at B samples.  python tools/conv_scaling.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)

def weight(cin):
  leaf = syn.quant_leaf((3, 3, cin, 128), 4.0 if cin == 2 else 5.0, 1, True, 0.9)
  a = float(leaf["DuQ_0"]["a"][0])
  pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                            torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
  return pk.int_weight_mfma(128)

def timeit(fn, n=3):
  fn(); torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n

CASES = ((2, 128),) if os.environ.get('ONLY_CONV0') else ((128, 64),) if os.environ.get('ONLY_CONV1') else ((2, 128), (128, 64))
for cin, hw in CASES:
  w = weight(cin)
  g = ops.ConvGeom(hw, hw, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  for T in ((20,) if (os.environ.get('ONLY_CONV0') or os.environ.get('ONLY_CONV1')) else (1, 2, 4, 10, 20, 40)):
    if cin == 2:
      x = (torch.rand((T, B, hw, hw, 2), device=dev) < 0.1).to(torch.uint8)
    else:
      x = ops.pack_bits((torch.rand((T, B, hw, hw, 128), device=dev) < 0.15).to(torch.uint8))
    ms = timeit(lambda: ops.conv_lif_forward(x, g, w, nrn, packed_out=True, pool=2, want_u=False, x_max=int(os.environ.get('X_MAX', '1'))))
    print("cin %3d hw %3d T %2d: %.3f ms  (%.3f ms/step)" % (cin, hw, T, ms, ms / T))
    del x
