"""Host-side cost of one eager model.apply (cProfile): python tools/host_profile.py [dense|c3] [B]"""
import cProfile, pstats, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import linen as nn, models, synthetic as syn, ops, parallel
which = sys.argv[1] if len(sys.argv) > 1 else "dense"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
if which == "dense":
  cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
  model = models.DenseSNN(num_classes=11, config=cfg)
  v = nn.tree_from_numpy(syn.dense_net_variables(2048, 512, 110, True, 0.5), dev)
  x = (torch.rand((B, 20, 2048), device=dev) < 0.095).to(torch.uint8)
else:
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  v = nn.tree_from_numpy(syn.conv_net_variables(prune_p=0.9, out=110), dev)
  x = (torch.rand((B, 20, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
def step():
  ops.forget_inputs()
  (logits, _) = model.apply(v, x, trgt=None, train=False, rng=None)
  return parallel.all_gather_rows(logits)
for _ in range(20):
  step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(300):
  step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step, with drain %.1f us/step" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
  step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
