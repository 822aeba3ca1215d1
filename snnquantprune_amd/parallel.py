"""Data-parallel eval: one process per GPU, batch sharded across ranks, logits
joined by one all-gather (RCCL over xGMI on MI355X; `nccl` backend = RCCL).

The reference shards the eval batch over local devices with jax.pmap and no
collective (examples/eval.py:106-129; input reshaped [num_devices, B/dev, ...] at
examples/input_pipeline.py:38-46); samples never interact in eval (BatchNorm
uses running statistics), so the only exchange is the gather of the per-sample
outputs.  Message: B/world x num_classes float32 per rank (45 KB at 1024 x 11),
latency-bound.
"""

from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False):
  """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).
  `single_rank_group`: create the process group even for one rank, so that the collectives
  below really run through the backend (RCCL on one GPU: the communicator, the all-gather
  kernel and the device-bound barrier execute; what a one-GPU box can rehearse of C4)."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", str(rank)))
  if single_rank_group and world == 1:
    os.environ.setdefault("MASTER_PORT", "29599")
  if (world > 1 or single_rank_group) and not dist.is_initialized():
    if backend is None:
      backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
      torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, world, local


def shard_bounds(batch_size: int, rank: int, world: int):
  """Contiguous split; the batch must divide evenly (examples/eval.py:64-72)."""
  if batch_size % world > 0:
    raise ValueError("Batch size (" + str(batch_size) + ") must be divisible by the "
                     "number of devices (" + str(world) + ").")
  per = batch_size // world
  return rank * per, (rank + 1) * per


def shard_batch(x, rank: int, world: int, batch_axis: int = 0):
  lo, hi = shard_bounds(x.shape[batch_axis], rank, world)
  return x.narrow(batch_axis, lo, hi - lo)


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
  """[b, ...] on every rank -> [world * b, ...] (rank-major), one collective."""
  if not dist.is_initialized():
    return x
  x = x.contiguous()
  world = dist.get_world_size(group)
  out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype,
                    device=x.device)
  dist.all_gather_into_tensor(out, x, group=group)
  return out


def sharded_logits(apply_fn, variables, inputs, rank: int, world: int, group=None,
                   **apply_kwargs):
  """Runs the model on this rank's slice of `inputs` [B, ...] and returns the
  logits of the WHOLE batch on every rank."""
  local = shard_batch(inputs, rank, world)
  out = apply_fn(variables, local, **apply_kwargs)
  logits = out[0] if isinstance(out, tuple) else out
  if isinstance(logits, tuple):
    logits = logits[0]
  return all_gather_rows(logits, group)
