"""Times the fp6 dense block against K (fixed cost versus per-chunk cost): python tools/dense_fp6_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc

dev = torch.device("cuda:0")
T, B, N = 20, 1024, int(os.environ.get("DENSE_N", "110"))
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
for K in (256, 1024, 4096, 16384, 32768):
  leaf = syn.quant_leaf((K, N), 4.0, 5, True, 0.9)
  a = float(leaf["DuQ_0"]["a"][0])
  pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                            torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
  w = pk.int_weight_mfma((N + 31) // 32 * 32)
  x = ops.pack_bits((torch.rand((T, B, K), device=dev) < 0.1).to(torch.uint8))
  for _ in range(3):
    ops.dense_lif_forward(x, w, K, N, nrn, want_u=False, packed_out=True)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    ops.dense_lif_forward(x, w, K, N, nrn, want_u=False, packed_out=True)
  e1.record(); torch.cuda.synchronize()
  print("K %6d  %.4f ms per launch (rt env %s)" % (K, e0.elapsed_time(e1) / 20, os.environ.get("SNNQP_DENSE_FP6_RT")))
