"""bench.py -- throughput of the quantized / pruned SNN forward pass on MI355X.

Workload (BASELINE.json configs[2], "C3"): 3 x (QuantConv 3x3 + BatchNorm + LIF +
2x2 max-pool) -> flatten -> QuantDense(110) + LIF -> vote, DVS128-shaped input
[B, T=20, 128, 128, 2], 4-bit DuQ weights, 90 % magnitude-pruned, B = 1024 per
GPU.  One "step" = one model.apply() on one resident batch (uint8 event frames
already in HBM) + the all-gather of the logits.  Weak scaling: every rank owns
its own B samples; value = N * B * K / max-over-ranks time.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed
inside the timed region) and, at N = 1, `cpu_baseline` (the CPU oracle in its
reference-literal float mode on a bounded sample, host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

INT8_MFMA_PEAK_TOPS = 5000.0   # dense int8 MFMA, 2x the ~2.5 PF bf16 dense peak
HBM_PEAK_GBS = 8000.0          # HBM3E spec (MI355X_MICROARCH.md)


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--batch", type=int, default=1024, help="samples per GPU")
  ap.add_argument("--frames", type=int, default=20)
  ap.add_argument("--bits", type=int, default=4)
  ap.add_argument("--prune", type=float, default=0.9)
  ap.add_argument("--cpu-samples", type=int, default=8)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  return ap.parse_args()


def cpu_baseline(args, variables_np):
  """Reference-literal float mode of the CPU oracle (dense float32 fake-quantised
  weights, BLAS matmul on im2col, float32 spike tensors between layers, T
  sequential LIF steps) on a bounded sample of the same workload."""
  from oracle import snn_oracle as o
  from snnquantprune_amd import synthetic as syn
  from tests.helpers import bn_of, qweight_of
  p = variables_np["params"]
  cq = [qweight_of(o, p["QuantConv_%d" % i], args.bits) for i in range(3)]
  bns = [bn_of(variables_np, i) for i in range(3)]
  dq = qweight_of(o, p["QuantDense_0"], args.bits)
  n = max(1, args.cpu_samples)
  x = syn.poisson_spikes((n, args.frames, 128, 128, 2), 0.1, seed=4242).astype(np.float32)
  o.conv3_dense_forward(x[:1, :2], cq, bns, dq, mode="float")      # warm-up, discarded
  t0 = time.perf_counter()
  o.conv3_dense_forward(x, cq, bns, dq, mode="float")
  dt = time.perf_counter() - t0
  cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
  return {"value": n / dt, "unit": "samples/s", "cores": int(cores), "kind": "port",
          "sample": "%d samples of the same C3 workload (T=%d, 128x128x2), oracle float "
                    "mode, numpy/BLAS threads = host cores, %.1f s" % (n, args.frames, dt)}


def main():
  args = parse()
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, parallel, synthetic as syn

  rank, world, local = parallel.init_from_env("nccl")
  assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
  assert torch.cuda.is_available(), "bench.py needs a GPU"
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)

  B, T = args.batch, args.frames
  cfg = syn.make_config(bits=args.bits, prune_percentage=args.prune)
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  variables_np = syn.conv_net_variables(prune_p=args.prune)
  variables = nn.tree_from_numpy(variables_np, dev)

  # synthetic Poisson-spike DVS frames, resident in HBM before the timed region
  gen = torch.Generator(device=dev)
  gen.manual_seed(8627169 + rank)
  p_spike = 1.0 - float(np.exp(-0.1))            # P(Poisson(0.1) > 0)
  x = (torch.rand((B, T, 128, 128, 2), device=dev, generator=gen) < p_spike).to(torch.uint8)

  def step():
    (logits, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
    return parallel.all_gather_rows(logits)

  def fence():
    if world > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    out = step()
  fence()
  ops.profile_start()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    out = step()
  fence()
  dt = time.perf_counter() - t0
  prof = ops.profile_stop()                       # {tag: (launches, total ms)}
  assert out.shape == (world * B, 11)

  tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
  if world > 1:
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
  dt = float(tmax.item())

  if rank != 0:
    return
  value = world * B * args.steps / dt

  # ---- roofline of the dominant kernel (HIP events inside the timed region) ----
  kern = {}
  for tag, (n, ms) in prof.items():
    kern[tag] = {"launches": n, "avg_ms": ms / max(n, 1)}
  dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["launches"])
  macs = {"conv3x3[128x128x2->128]": B * T * 128 * 128 * 128 * 18,
          "conv3x3[64x64x128->128]": B * T * 64 * 64 * 128 * 1152,
          "conv3x3[32x32x128->128]": B * T * 32 * 32 * 128 * 1152,
          "dense[32768->110]": B * T * 32768 * 110}
  # algorithmic HBM bytes per launch, formats the kernels really read / write
  hbm = {"conv3x3[128x128x2->128]": B * T * (128 * 128 * 2 + 64 * 64 * 16),
         "conv3x3[64x64x128->128]": B * T * (64 * 64 * 16 + 32 * 32 * 16),
         "conv3x3[32x32x128->128]": B * T * (32 * 32 * 16 + 16 * 16 * 16),
         "dense[32768->110]": B * T * (4096 + 16) + 32768 * 128}
  for tag, k in kern.items():
    if tag in macs:
      k["tops"] = 2.0 * macs[tag] / (k["avg_ms"] * 1e-3) / 1e12
      k["hbm_gbs"] = hbm[tag] / (k["avg_ms"] * 1e-3) / 1e9
  d = kern[dom]
  roofline = {"kernel": dom, "bound": "mfma", "achieved": d.get("tops"),
              "peak": INT8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
              "frac": (d.get("tops") or 0.0) / INT8_MFMA_PEAK_TOPS, "traffic": None,
              "avg_launch_ms": d["avg_ms"]}
  dn = kern.get("dense[32768->110]")
  roofline_dense = None
  if dn is not None:
    roofline_dense = {"kernel": "dense[32768->110]", "bound": "hbm",
                      "achieved": dn["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": dn["hbm_gbs"] / HBM_PEAK_GBS, "traffic": None,
                      "avg_launch_ms": dn["avg_ms"]}

  line = {
      "metric": "samples/sec/node (DVS128 T=20, 4-bit/90%-pruned)",
      "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "int8 codes x binary spikes -> int32 acc, f32 membrane",
      "data": "synthetic Poisson(0.1)>0 spikes, N(0,1/fan_in) weights, random seeds fixed",
      "config": {"workload": "C3: 3x(qconv3x3+BN+LIF+pool2) + qdense(32768->110)+LIF + vote, "
                             "DVS128 128x128x2, T=%d, %d-bit, %.0f%% pruned" %
                             (T, args.bits, args.prune * 100),
                 "batch_per_gpu": B, "global_batch": world * B, "frames": T,
                 "parallelism": "dp%d (batch-sharded, all-gather logits)" % world},
      "roofline": roofline,
      "roofline_dense": roofline_dense,
      "kernels": kern,
  }
  if world == 1 and not args.no_cpu_baseline:
    line["cpu_baseline"] = cpu_baseline(args, variables_np)
  print(json.dumps(line))


if __name__ == "__main__":
  main()
