import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
dev = torch.device('cuda:0')
B, T = int(sys.argv[1]), 20
cfg = syn.make_config(bits=4, prune_percentage=0.9)
model = models.CextNet(num_classes=11, config=cfg)
variables = nn.tree_from_numpy(syn.cextnet_variables(prune_p=0.9), dev)
x = (torch.rand((B, T, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
for _ in range(2):
  out = model.apply(variables, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
ops.profile_start()
t0 = time.perf_counter()
for _ in range(3):
  out = model.apply(variables, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
prof = ops.profile_stop()
print("CextNet B=%d: %.2f ms/step, %.0f samples/s" % (B, dt * 1e3, B / dt))
for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
  print("  %-40s x%d  %.3f ms each" % (k, n // 3, ms / n))
