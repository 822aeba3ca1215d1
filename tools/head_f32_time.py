"""Diagnostic: the fused dense head (config C2) at a batch, per input format, as a hipGraph:
   python tools/head_f32_time.py [B]      (SNNQP_DENSE_WIDE_RT=1..4 forces the row tiles)"""
import sys, time, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 20
model = models.DenseSNN(num_classes=11, config=syn.make_config(bits=8, prune_percentage=0.5, hidden=512))
variables = nn.tree_from_numpy(syn.dense_net_variables(2048, 512, 110, True, 0.5), dev)
xu = (torch.rand((B, T, 2048), device=dev) < 0.1).to(torch.uint8)
for name, x in (("u8", xu), ("f32", xu.to(torch.float32)), ("bits", ops.pack_bits(xu))):
  step = nn.capture(model, variables, x, trgt=None, train=False, rng=None)
  for _ in range(5):
    step()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  n = 100
  a.record()
  for _ in range(n):
    step()
  b.record()
  torch.cuda.synchronize()
  ms = a.elapsed_time(b) / n
  nbytes = x.bits.numel() * 4 if name == "bits" else x.numel() * x.element_size()
  print("%-5s B=%d: %.4f ms per step  %.1f M samples/s  input %.0f MB -> %.2f TB/s" %
        (name, B, ms, B / ms / 1e3, nbytes / 1e6, nbytes / ms / 1e9), flush=True)
  step.close()
