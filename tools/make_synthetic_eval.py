"""Writes a self-contained eval work directory -- a Flax-format checkpoint of a randomly
initialised C3 model, a `.npz` of synthetic DVS-shaped frames with labels and a config file --
for `python -m snnquantprune_amd.eval --workdir DIR --config DIR/config.py` (the harness that
mirrors examples/eval.py; under `python -m torch.distributed.run --nproc-per-node N ...` every
rank takes its slice of the split).

  python tools/make_synthetic_eval.py DIR [--samples 64] [--hw 128] [--frames 20] [--feed ev1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIG = '''"""Eval config in the shape of examples/tcja/configs/prune_quant_joint.py."""
from functools import partial

from snnquantprune_amd import linen as nn
from snnquantprune_amd.quant import DuQ, gaussian_init, round_ewgs
from snnquantprune_amd.spiking_learning import atan, multi_step_LIF
from snnquantprune_amd.train_utils import mse_loss


def get_config():
  config = nn.ConfigDict()
  config.seed = 203853699
  config.model = "ConvDenseSNN"
  config.dataset = %(data)r            # .npz with dvs_matrix [N, T, H, W, 2] uint8, label [N]
  config.feed_format = %(feed)r        # wire format of the frames: "u8" | "ev1" | "ev4"
  config.cache = True
  config.num_classes = 11
  config.num_frames = %(frames)d
  config.channels = 128
  config.neuron_dynamics = partial(multi_step_LIF, spike_fn=atan, tau=2.0)
  config.loss_fn = partial(mse_loss, T=1)
  config.smoothing = 0.0
  config.batch_size = %(batch)d
  config.eval_batch_size = %(batch)d
  config.steps_per_eval = -1
  config.quant = nn.ConfigDict()
  config.quant.bits = 4
  config.quant.g_scale = 5e-3
  config.quant.init_fn = gaussian_init
  config.quant.weight = partial(DuQ, round_fn=round_ewgs)
  config.quant.prune_percentage = 0.9
  return config
'''


def main(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument("workdir")
  ap.add_argument("--samples", type=int, default=64)
  ap.add_argument("--hw", type=int, default=128)
  ap.add_argument("--frames", type=int, default=20)
  ap.add_argument("--batch", type=int, default=16)
  ap.add_argument("--feed", choices=("u8", "ev1", "ev4"), default="ev1")
  args = ap.parse_args(argv)
  from snnquantprune_amd import checkpoint, synthetic as syn
  os.makedirs(args.workdir, exist_ok=True)
  v = syn.conv_net_variables(hw=args.hw, prune_p=0.9, random_bn=True,
                             gains=(4.0, 5.0, 4.0, 4.0) if args.hw >= 64 else (4.0, 5.0, 6.0, 10.0))
  state = {"step": np.int32(1), "params": {"params": v["params"]}, "batch_stats": v["batch_stats"]}
  with open(os.path.join(args.workdir, "checkpoint_1"), "wb") as f:
    f.write(checkpoint.msgpack_serialize(state))
  x = syn.poisson_spikes((args.samples, args.frames, args.hw, args.hw, 2), 0.1, seed=8627169)
  rng = np.random.Generator(np.random.PCG64(8627170))
  data = os.path.join(args.workdir, "frames.npz")
  np.savez(data, dvs_matrix=x, label=rng.integers(0, 11, args.samples).astype(np.int8))
  with open(os.path.join(args.workdir, "config.py"), "w") as f:
    f.write(CONFIG % dict(data=data, feed=args.feed, frames=args.frames, batch=args.batch))
  print("wrote", args.workdir)


if __name__ == "__main__":
  main()
