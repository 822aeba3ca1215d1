"""Diagnostic: time of the 2-layer dense SNN (BASELINE configs C1 / C2)."""
import sys, time, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
dev = torch.device('cuda:0')
for name, B, T, bits, p, quant in (("C1 f32 weights", 32, 10, 8, -1.0, False), ("C2 8-bit 50%", 256, 20, 8, 0.5, True),
                                   ("C2 at B=4096", 4096, 20, 8, 0.5, True)):
  cfg = syn.make_config(bits=bits, prune_percentage=p, quantized=quant) if "quantized" in syn.make_config.__code__.co_varnames \
      else syn.make_config(bits=bits, prune_percentage=p)
  model = models.DenseSNN(num_classes=11, config=cfg)
  variables = nn.tree_from_numpy(syn.dense_net_variables(prune_p=max(p, 0.0), quantized=quant), dev)
  x = (torch.rand((B, T, 2048), device=dev) < 0.15).to(torch.uint8)   # [B, T, K] as eval.py hands it over
  for _ in range(2):
    out = model.apply(variables, x, trgt=None, train=False, rng=None)
  torch.cuda.synchronize()
  ops.profile_start()
  t0 = time.perf_counter()
  for _ in range(5):
    out = model.apply(variables, x, trgt=None, train=False, rng=None)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 5
  prof = ops.profile_stop()
  print("%s B=%d T=%d: %.3f ms/step, %.0f samples/s" % (name, B, T, dt * 1e3, B / dt),
        {k: round(ms / n, 4) for k, (n, ms) in prof.items()})
