"""In-kernel phase times of dense_wide_kernel from a -DW_STAMP=1 diagnostic build
(tools/diag/dense_wide_stamps.patch):
  python tools/diag_build.py wide_stamp --patch tools/diag/dense_wide_stamps.patch -- -DW_STAMP=1
  SNNQP_DIAG_LIB=diag_build/wide_stamp/libsnnqp.so python tools/wide_stamps.py [B] [u8|f32|bits]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snnquantprune_amd import _lib, linen as nn, models, ops, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fmt = sys.argv[2] if len(sys.argv) > 2 else "u8"
dev = torch.device("cuda:0")
cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
model = models.DenseSNN(num_classes=11, config=cfg)
v = nn.tree_from_numpy(syn.dense_net_variables(2048, 512, 110, True, 0.5), dev)
x = (torch.rand((B, 20, 2048), device=dev) < 0.095).to(torch.uint8)
x = x.to(torch.float32) if fmt == "f32" else ops.pack_bits(x) if fmt == "bits" else x
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
s = s[s[:, 0] > 0]
s = s[s[:, 5] > s[:, 0]]
print(fmt, "B", B, "workgroups that ran to the vote:", len(s), " (s_memtime ticks)")
names = ["prologue", "K loop", "barrier", "walk 1 + raster", "second block + vote"]
d = np.diff(s[:, :6], axis=1)
for i, nme in enumerate(names):
  print("%-22s mean %8.1f  p50 %8.1f  max %8.1f" % (nme, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = s[:, 5] - s[:, 0]
print("workgroup total mean %.1f; launch span %.1f; starts spread over %.1f" % (
    tot.mean(), s[:, 5].max() - s[:, 0].min(), s[:, 0].max() - s[:, 0].min()))
