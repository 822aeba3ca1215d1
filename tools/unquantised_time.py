import sys, time, torch
sys.path.insert(0, '/root/repo')
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
dev = torch.device('cuda:0')
B, T = int(sys.argv[1]), 20
cfg = syn.make_config(bits=4, prune_percentage=0.5)
model = models.ConvDenseSNN(num_classes=11, config=cfg)
variables = nn.tree_from_numpy(syn.conv_net_variables(quantized=False, prune_p=0.5, gains=(5.0, 7.0, 8.0, 8.0)), dev)
x = (torch.rand((B, T, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
out = model.apply(variables, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
ops.profile_start()
t0 = time.perf_counter()
out = model.apply(variables, x, trgt=None, train=False, rng=None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = ops.profile_stop()
print("unquantised C3 B=%d: %.1f ms/step, %.0f samples/s" % (B, dt * 1e3, B / dt), {k: (n, round(ms, 2)) for k, (n, ms) in prof.items()})
