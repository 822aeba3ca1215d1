"""Diagnostic: per-step wall time of the first steps of a fresh process (C3, B = 1024)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
dev = torch.device("cuda:0")
cfg = syn.make_config(bits=4, prune_percentage=0.9)
model = models.ConvDenseSNN(num_classes=11, config=cfg)
variables = nn.tree_from_numpy(syn.conv_net_variables(prune_p=0.9), dev)
x = (torch.rand((1024, 20, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
torch.cuda.synchronize()
for i in range(14):
  t0 = time.perf_counter()
  ops.forget_inputs()
  (logits, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  torch.cuda.synchronize()
  print("step %2d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
