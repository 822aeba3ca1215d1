"""In-kernel phase times of dense_wide_kernel from a -DW_STAMP=1 diagnostic build:
  SNNQP_DIAG_LIB=diag_build/wide_stamp/libsnnqp.so python tools/wide_stamps.py [B]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import _lib, linen as nn, models, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
model = models.DenseSNN(num_classes=11, config=cfg)
v = nn.tree_from_numpy(syn.dense_net_variables(2048, 512, 110, True, 0.5), dev)
x = (torch.rand((B, 20, 2048), device=dev) < 0.095).to(torch.uint8)
for _ in range(5):
  model.apply(v, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
lib = _lib.lib()
n = 16 * 8192
buf = (ctypes.c_ulonglong * n)()
lib.snnqp_debug_wide_stamps.restype = ctypes.c_int
lib.snnqp_debug_wide_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.snnqp_debug_wide_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.int64)
live = s[:, 0] > 0
s = s[live]
# (workgroups that handed their half raster over and left have no stamps behind the hand-over)
full = s[:, 10] > s[:, 0]
print("workgroups", len(s), "of which ran to the vote", int(full.sum()))
s = s[full]
print("workgroups", len(s), " (s_memtime ticks = shader cycles)")
names = ["prologue", "K loop", "barrier", "walk1", "flush words+s_out", "layer2 mfma", "barrier", "walk2", "barrier", "vote"]
d = np.diff(s[:, :11], axis=1)
for i, nme in enumerate(names):
  print("%-20s mean %8.1f ticks  p50 %8.1f  max %8.1f" % (nme, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = s[:, 10] - s[:, 0]
print("workgroup total mean %.1f ticks; launch span %.1f ticks" % (tot.mean(), s[:, 10].max() - s[:, 0].min()))
