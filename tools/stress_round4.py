"""Wider sweep of the round-4 dense paths than the test suite runs: random dense blocks (the wide
kernel, the fp6 kernel with its K split, the 128-column kernel) against the direct-form kernel,
and random dense heads against the blocks run one by one.   python tools/stress_round4.py [N]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.stress import dense_block_random
from snnquantprune_amd import _lib as L, ops, synthetic as syn, packing
from snnquantprune_amd.quant import QuantDesc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
bad = dense_block_random(dev, N, 40004)
print("dense blocks:", N, "failures", bad)
rng = np.random.Generator(np.random.PCG64(77))
def weight(shape, bits, seed):
  leaf = syn.quant_leaf(shape, float(rng.uniform(3, 7)), seed, True, float(rng.choice([0.0, 0.5, 0.9])))
  a = float(leaf["DuQ_0"]["a"][0])
  pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, bits, a, a, float(2 ** (bits - 1) - 1), a),
                            torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
  return pk.int_weight_mfma((shape[1] + 31) // 32 * 32)
fails = []
for it in range(N):
  T, B = int(rng.integers(1, 65)), int(rng.integers(1, 400))
  K = int(16 * rng.integers(1, 200))
  N1 = int(rng.integers(129, 513))
  group = int(rng.choice([1, 2, 5, 10]))
  N2 = group * int(rng.integers(1, 128 // group + 1))
  bits = int(rng.choice([3, 4, 8]))
  w1, w2 = weight((K, N1), bits, 1000 + it), weight((N1, N2), bits, 5000 + it)
  kind = rng.choice(["ms2", "plif", "ms3", "vr"])
  nrn = {"ms2": ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0),
         "plif": ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, 0.4375, 1.0, 0.0),
         "ms3": ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 1.0, 0.0),
         "vr": ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 0.9, 0.1)}[kind]
  lam = float(rng.choice([0.1, 0.3]))
  xu = torch.from_numpy(np.minimum(rng.poisson(lam, (T, B, K)), 255).astype(np.uint8)).to(dev)
  x = xu if rng.random() < 0.6 else ops.pack_bits(xu.clamp(max=1))
  try:
    logits, s1, s2 = ops.dense_head_forward(x, w1, K, N1, nrn, w2, N2, nrn, group=group, want_s1=True, want_s2=True)
  except L.SnnqpError as e:
    print("skip", T, B, K, N1, N2, str(e)[:60]); continue
  _, r1 = ops.dense_lif_forward(x, w1, K, N1, nrn, want_u=False, packed_out=True, impl=L.IMPL_GENERIC)
  _, r2 = ops.dense_lif_forward(r1, w2, N1, N2, nrn, want_u=False, packed_out=True, impl=L.IMPL_GENERIC)
  want = ops.vote(r2, group)
  ok = torch.equal(s1.bits, r1.bits) and torch.equal(s2.bits, r2.bits) and torch.equal(logits, want)
  if not ok:
    fails.append((T, B, K, N1, N2, group, bits, kind, type(x).__name__))
    print("MISMATCH", fails[-1], flush=True)
print("dense heads:", N, "failures", fails, "device status", ops.device_status())
sys.exit(1 if (bad or fails) else 0)
