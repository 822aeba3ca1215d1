"""Where does a model step wait for the device?  Patches Tensor.item / tolist / cpu / __bool__ /
__int__ / __float__ and prints every distinct call site hit during one warm step:
  python tools/find_syncs.py [cextnet|c3|dense]"""
import sys, os, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import linen as nn, models, synthetic as syn, ops
which = sys.argv[1] if len(sys.argv) > 1 else "cextnet"
dev = torch.device("cuda:0")
B = 8
cfg = syn.make_config(bits=4, prune_percentage=0.9)
if which == "cextnet":
  model = models.CextNet(num_classes=11, config=cfg)
  v = nn.tree_from_numpy(syn.cextnet_variables(prune_p=0.9, out=110), dev)
else:
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  v = nn.tree_from_numpy(syn.conv_net_variables(prune_p=0.9, out=110), dev)
x = (torch.rand((B, 20, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
for _ in range(3):
  model.apply(v, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
seen = {}
def wrap(name):
  orig = getattr(torch.Tensor, name)
  def f(self, *a, **k):
    if self.is_cuda:
      st = [fr for fr in traceback.extract_stack()[:-1] if "snnquantprune_amd" in fr.filename]
      key = (name,) + tuple((os.path.basename(fr.filename), fr.lineno) for fr in st[-3:])
      seen[key] = seen.get(key, 0) + 1
    return orig(self, *a, **k)
  setattr(torch.Tensor, name, f)
for n in ("item", "tolist", "cpu", "__bool__", "__int__", "__float__", "numpy"):
  wrap(n)
ops.forget_inputs()
model.apply(v, x, trgt=None, train=False, rng=None)
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
  print(n, k)
print("distinct sites:", len(seen))
