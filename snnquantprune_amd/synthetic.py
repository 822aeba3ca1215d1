"""Synthetic inputs and parameter trees for the BASELINE.json configurations
(SURVEY.md section 8d).  NumPy only, shared by tests, bench.py and smoke():
"identical random inputs" means these same arrays on both sides.

Seeds: weights 203853699 (examples/tcja/configs/prune_quant_joint.py:21),
data 8627169 (quant_test.py:149).  Spikes: (Poisson(lam) > 0), lam = 0.1.
Weights: N(0, 1/fan_in) * gain (stand-in for lecun_normal, flax_qdense.py:26),
gain chosen per layer so firing rates stay in a sane band; a = c =
gaussian_init(W) (quant.py:305-309); mask = per-layer magnitude prune
(examples/train_inpt_spikingjelly.py:147-157).
"""

from __future__ import annotations

from functools import partial

import numpy as np

WEIGHT_SEED = 203853699
DATA_SEED = 8627169
F32 = np.float32


def poisson_spikes(shape, lam=0.1, seed=DATA_SEED):
  rng = np.random.Generator(np.random.PCG64(seed))
  return (rng.poisson(lam, size=shape) > 0).astype(np.uint8)


def poisson_counts(shape, lam=0.3, seed=DATA_SEED):
  """Integer event counts like real DVS frames (input_pipeline.py:195-218)."""
  rng = np.random.Generator(np.random.PCG64(seed))
  return np.minimum(rng.poisson(lam, size=shape), 127).astype(np.uint8)


def kernel(shape, gain=1.0, seed=WEIGHT_SEED):
  rng = np.random.Generator(np.random.PCG64(seed))
  fan_in = int(np.prod(shape[:-1]))
  return (rng.standard_normal(shape) * (gain / np.sqrt(fan_in))).astype(F32)


def magnitude_mask(w, p):
  """update_prune_mask: zero the int(numel * p) smallest |w|."""
  mask = np.ones(w.shape, dtype=F32)
  k = int(np.prod(w.shape) * p)
  if k > 0:
    idx = np.argpartition(np.abs(w).reshape(-1), k)[:k]
    mask.reshape(-1)[idx] = 0
  return mask


def gaussian_ac(w):
  """gaussian_init(kernel): max(|mu - 3 sigma|, |mu + 3 sigma|) in float32."""
  w = np.asarray(w, dtype=F32)
  mu = np.mean(w, dtype=F32)
  sigma = np.std(w, dtype=F32)
  return F32(max(abs(mu - F32(3) * sigma), abs(mu + F32(3) * sigma)))


def quant_leaf(shape, gain, seed, quantized=True, prune_p=-1.0):
  """One layer's parameter leaf in the reference's naming."""
  w = kernel(shape, gain, seed)
  leaf = {"kernel": w}
  ac = gaussian_ac(w) if quantized else F32(-1)     # a == -1: pass-through
  leaf["DuQ_0"] = {"a": np.array([ac], F32), "c": np.array([ac], F32)}
  if prune_p >= 0:
    leaf["prune_0"] = {"mask": magnitude_mask(w, prune_p)}
  return leaf


def bn_leaf(c, randomize=False, seed=WEIGHT_SEED + 77):
  if not randomize:
    return ({"scale": np.ones(c, F32), "bias": np.zeros(c, F32)},
            {"mean": np.zeros(c, F32), "var": np.ones(c, F32)})
  rng = np.random.Generator(np.random.PCG64(seed))
  return ({"scale": (1 + 0.2 * rng.standard_normal(c)).astype(F32),
           "bias": (0.1 * rng.standard_normal(c)).astype(F32)},
          {"mean": (0.1 * rng.standard_normal(c)).astype(F32),
           "var": (1 + 0.3 * rng.random(c)).astype(F32)})


def dense_net_variables(K=2048, hidden=512, out=110, quantized=True, prune_p=0.5,
                        gains=(4.0, 6.0), seed=WEIGHT_SEED):
  """Variables of models.DenseSNN (configs C1 / C2)."""
  return {"params": {
      "QuantDense_0": quant_leaf((K, hidden), gains[0], seed, quantized, prune_p),
      "QuantDense_1": quant_leaf((hidden, out), gains[1], seed + 1, quantized, prune_p),
  }, "batch_stats": {}}


def conv_net_variables(channels=128, cin=2, nblocks=3, hw=128, out=110,
                       quantized=True, prune_p=0.9, gains=(4.0, 5.0, 4.0, 4.0),
                       random_bn=False, seed=WEIGHT_SEED):
  """Variables of models.ConvDenseSNN (config C3)."""
  params, stats = {}, {}
  c_in = cin
  for i in range(nblocks):
    params["QuantConv_%d" % i] = quant_leaf((3, 3, c_in, channels), gains[i],
                                            seed + i, quantized, prune_p)
    p, s = bn_leaf(channels, random_bn, seed + 100 + i)
    params["BatchNorm_%d" % i], stats["BatchNorm_%d" % i] = p, s
    c_in = channels
    hw //= 2
  params["QuantDense_0"] = quant_leaf((hw * hw * channels, out), gains[nblocks],
                                      seed + 50, quantized, prune_p)
  return {"params": params, "batch_stats": stats}


def cextnet_variables(channels=128, frames=20, hw=128, out=110, prune_p=0.9,
                      gains=(4.0, 5.0, 5.0, 6.0, 12.0, 6.0, 6.0, 16.0, 8.0), random_bn=False,
                      seed=WEIGHT_SEED):
  """Variables of models.CextNet in the reference's naming and order
  (tcja_load_pretrained_weights.py:19-36): QuantConv_0..2 (3x3), QuantConv_3 (3x3),
  QuantConv_4 / 5 (TCJA over T / over C), QuantConv_6 (3x3), QuantConv_7 / 8 (TCJA),
  BatchNorm_0..4, QuantDense_0 (flatten -> 4 * channels), QuantDense_1 (-> out).
  gains: conv0..4, tcja (both), dense0, dense1."""
  params, stats = {}, {}
  shapes = {0: (3, 3, 2, channels), 1: (3, 3, channels, channels), 2: (3, 3, channels, channels),
            3: (3, 3, channels, channels), 4: (4, frames, frames), 5: (4, channels, channels),
            6: (3, 3, channels, channels), 7: (4, frames, frames), 8: (4, channels, channels)}
  g = {0: gains[0], 1: gains[1], 2: gains[2], 3: gains[3], 6: gains[4], 4: gains[5], 5: gains[6],
       7: gains[5], 8: gains[6]}
  for i in range(9):
    params["QuantConv_%d" % i] = quant_leaf(shapes[i], g[i], seed + i, True, prune_p)
  for i in range(5):
    p, s = bn_leaf(channels, random_bn, seed + 100 + i)
    params["BatchNorm_%d" % i], stats["BatchNorm_%d" % i] = p, s
  flat = (hw // 32) * (hw // 32) * channels
  params["QuantDense_0"] = quant_leaf((flat, channels * 4), gains[7], seed + 50, True, prune_p)
  params["QuantDense_1"] = quant_leaf((channels * 4, out), gains[8], seed + 51, True, prune_p)
  return {"params": params, "batch_stats": stats}


def make_config(bits=4, prune_percentage=0.9, channels=128, tau=2.0, quantized=True,
                **extra):
  """ConfigDict shaped like examples/tcja/configs/prune_quant_joint.py."""
  from . import linen as nn
  from .quant import DuQ, gaussian_init, round_ewgs
  from .spiking_learning import atan, multi_step_LIF
  cfg = nn.ConfigDict()
  cfg.seed = WEIGHT_SEED
  cfg.neuron_dynamics = partial(multi_step_LIF, spike_fn=atan, tau=tau)
  cfg.num_frames = 20
  cfg.num_classes = 11
  cfg.channels = channels
  cfg.smoothing = 0.0
  cfg.quant = nn.ConfigDict()
  cfg.quant.bits = bits
  cfg.quant.init_fn = gaussian_init
  cfg.quant.g_scale = 5e-3
  if quantized:
    cfg.quant.weight = partial(DuQ, round_fn=round_ewgs)
  cfg.quant.prune_percentage = prune_percentage
  for k, v in extra.items():
    cfg[k] = v
  return cfg
