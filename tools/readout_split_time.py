"""The read-out (32768 -> 110, B = 1024, T = 20) with and without the K split over workgroups:
  python tools/readout_split_time.py"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import _lib as L, ops, synthetic as syn, packing
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B, T, K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 20, 32768, 110
leaf = syn.quant_leaf((K, N), 4.0, 77, True, 0.9)
a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, c, 7.0, c),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight_mfma(128)
x = ops.pack_bits((torch.rand((T, B, K), device=dev) < 0.1).to(torch.uint8))
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
print("workspace bytes", L.lib().snnqp_dense_workspace_bytes(L.BITS, T, B, K, N, ctypes.byref(w.struct())),
      "plan override", os.environ.get("SNNQP_DENSE_FP6_PLAN"))
if os.environ.get("SNNQP_DENSE_FP6_PLAN"):
  _big = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
  ops._dense_workspace = lambda d, n: _big
def run(n=20):
  for _ in range(3):
    ops.dense_lif_forward(x, w, K, N, nrn, want_u=False, packed_out=True)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n):
    _, s = ops.dense_lif_forward(x, w, K, N, nrn, want_u=False, packed_out=True)
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n * 1e3, s
t1, s1 = run()
orig = ops._dense_workspace
ops._dense_workspace = lambda dev, n: None
t0, s0 = run()
ops._dense_workspace = orig
print("split %.1f us   unsplit %.1f us   equal %s" % (t1, t0, torch.equal(s0.bits, s1.bits)))
