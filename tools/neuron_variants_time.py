"""Diagnostic: C3 forward time with the other neuron kinds / time constants (general
epilogue of the MFMA kernels)."""
import sys, time, os, torch
from functools import partial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snnquantprune_amd import linen as nn, models, ops, synthetic as syn
from snnquantprune_amd import spiking_learning as sl
dev = torch.device('cuda:0')
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
x = (torch.rand((B, T, 128, 128, 2), device=dev) < 0.095).to(torch.uint8)
variants = {
    "multi_step_LIF tau=2 (fast path)": partial(sl.multi_step_LIF, spike_fn=sl.atan, tau=2.0),
    "multi_step_LIF tau=3": partial(sl.multi_step_LIF, spike_fn=sl.atan, tau=3.0),
    "multi_step_LIF v_reset=0.1": partial(sl.multi_step_LIF, spike_fn=sl.atan, tau=2.0, v_reset=0.1),
    "parametric_leaky_IF": partial(sl.parametric_leaky_IF, spike_fn=sl.atan, init_tau=2.0),
}
for name, nd in variants.items():
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  cfg.neuron_dynamics = nd
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  variables = syn.conv_net_variables(prune_p=0.9)
  try:
    v0 = model.init(0, x[:1], trgt=None, train=False, rng=None)     # neuron params (PLIF tau)
    tree = nn.tree_from_numpy(variables, dev)
    for k in v0.get("params", {}):
      if k not in tree["params"]:
        tree["params"][k] = v0["params"][k]
    for _ in range(2):
      out = model.apply(tree, x, trgt=None, train=False, rng=None)
    torch.cuda.synchronize()
    ops.profile_start()
    t0 = time.perf_counter()
    for _ in range(3):
      out = model.apply(tree, x, trgt=None, train=False, rng=None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    prof = ops.profile_stop()
    print("%-34s B=%d: %.2f ms/step" % (name, B, dt * 1e3), {k: round(ms / n, 3) for k, (n, ms) in prof.items()})
  except Exception as e:
    print("%-34s FAILED: %s: %s" % (name, type(e).__name__, str(e)[:200]))
