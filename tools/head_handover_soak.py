"""Soak of the fused dense head's column split (two workgroups per tile hand halves of the hidden
raster over through the workspace): N random batches, the head's logits against the two blocks and
the vote run one by one on the same inputs.   python tools/head_handover_soak.py [N] [B]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import _lib as L, ops, synthetic as syn, packing
from snnquantprune_amd.quant import QuantDesc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
T, K, N1, N2 = 20, 2048, 512, 110
def weight(shape, seed):
  leaf = syn.quant_leaf(shape, 4.0, seed, True, 0.5)
  a = float(leaf["DuQ_0"]["a"][0])
  pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 8, a, a, 127.0, a),
                            torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
  return pk.int_weight_mfma((shape[1] + 31) // 32 * 32)
w1, w2 = weight((K, N1), 11), weight((N1, N2), 12)
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
assert L.lib().snnqp_dense_head_workspace_bytes(T, B, N1) > 0, "this batch does not split"
gen = torch.Generator(device=dev); gen.manual_seed(5)
bad = 0
for i in range(N):
  x = (torch.rand((T, B, K), device=dev, generator=gen) < 0.1).to(torch.uint8)
  logits, _, _ = ops.dense_head_forward(x, w1, K, N1, nrn, w2, N2, nrn)
  _, s1 = ops.dense_lif_forward(x, w1, K, N1, nrn, want_u=False, packed_out=True)
  _, s2 = ops.dense_lif_forward(s1, w2, N1, N2, nrn, want_u=False, packed_out=True)
  want = ops.vote(s2, 10)
  if not torch.equal(logits, want):
    bad += 1
    print("mismatch at batch", i, int((logits != want).sum()), flush=True)
  if i % 500 == 0:
    print("batch", i, "mismatches so far", bad, "status", ops.device_status(), flush=True)
print("done:", N, "batches,", bad, "mismatches, device status", ops.device_status())
sys.exit(1 if bad else 0)
