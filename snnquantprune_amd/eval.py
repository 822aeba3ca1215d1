"""Eval harness -- mirror of examples/eval.py:53-139 (`evaluate(config, workdir)`), the
loop the kernels drop in under: restore the checkpoint, shard the data over the
processes, run `eval_step` per batch, stack the metrics, mean loss / accuracy.

Differences forced by the platform, none in the numbers:
  * one PROCESS per GPU (`torch.distributed`, RCCL) instead of `jax.pmap` over local
    devices: `jax_utils.replicate` is each rank loading the checkpoint, the per-device
    metrics that pmap stacks on the host (eval.py:125-129) are all-gathered;
  * the dataset is a `.npz` of event-count frames (`config.dataset`), see
    input_pipeline.py; batches are prefetched two deep over PCIe (feed.DeviceFeeder),
    in the wire format `config.feed_format`;
  * `config.prepare_params = True` re-derives the prune masks and DuQ (a, c) from the
    restored kernels the way train_inpt_spikingjelly.py:206-230 does before training
    (a freshly converted TCJA checkpoint has none).

  python -m snnquantprune_amd.eval --workdir DIR --config CONFIG.py
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      -m snnquantprune_amd.eval --workdir DIR --config CONFIG.py
CONFIG.py defines get_config() returning a snnquantprune_amd.linen.ConfigDict shaped like
examples/tcja/configs/prune_quant_joint.py.
"""

from __future__ import annotations

import argparse
import glob
import logging
import os
import re
import sys
from typing import Any, Callable, Optional

import torch

from . import checkpoint, input_pipeline, parallel
from . import linen as nn
from .train_utils import EvalState, create_model, eval_step, mse_loss


def latest_checkpoint(workdir: str, prefix: str = "checkpoint_") -> Optional[str]:
  """flax.training.checkpoints.latest_checkpoint: the file `<prefix><step>` with the
  largest step (natural order), or None."""
  best, best_step = None, None
  for p in glob.glob(os.path.join(workdir, prefix + "*")):
    m = re.fullmatch(re.escape(prefix) + r"(\d+(?:\.\d+)?)", os.path.basename(p))
    if m and os.path.isfile(p):
      step = float(m.group(1))
      if best_step is None or step > best_step:
        best, best_step = p, step
  return best


def initialized(key, image_size, model, t, device):
  """train_utils.py:137-158: variables of a model from a dummy [1, T, S, S, 2] batch."""
  shape = (1, image_size, image_size, 2) if t == -1 else (1, t, image_size, image_size, 2)
  variables = model.init({"params": key, "dropout": key},
                         torch.zeros(shape, dtype=torch.float32, device=device), rng=key,
                         trgt=torch.ones((1,)), online=t == -1, train=False)
  return variables["params"], variables.get("batch_stats", {})


def create_eval_state(rng, config, model, image_size, device) -> EvalState:
  """create_train_state (train_utils.py:161-192) without the optimiser."""
  params, batch_stats = initialized(rng, image_size, model,
                                    -1 if "online" in config else config.num_frames, device)
  return EvalState(apply_fn=model.apply, params={"params": params}, batch_stats=batch_stats)


def restore_checkpoint(state: Optional[EvalState], workdir: str, apply_fn: Callable,
                       device) -> Optional[EvalState]:
  """train_utils.py:30-31: the latest checkpoint of `workdir` over `state`; `state` itself
  when the directory holds none (flax's behaviour)."""
  path = latest_checkpoint(workdir) if workdir else None
  if path is None:
    return state
  tree = nn.tree_from_numpy(checkpoint.load_flax_checkpoint(path), device)
  return EvalState(apply_fn=apply_fn, params={"params": tree["params"]},
                   batch_stats=tree["batch_stats"])


def evaluate_metrics(config, workdir: str, *, model=None, source=None, device=None):
  """The body of evaluate(): returns (state, summary, per_step) with summary =
  {'loss', 'accuracy', 'samples', 'steps'} over ALL ranks' shards."""
  rank, world, local = parallel.init_from_env(config.get("backend", None))
  if config.eval_batch_size % world > 0:                          # eval.py:64-72
    raise ValueError("Batch size (" + str(config.eval_batch_size) + ") must be divisible by "
                     "the number of devices (" + str(world) + ").")
  if device is None:
    device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
  device = torch.device(device)
  if device.type == "cuda":
    torch.cuda.set_device(device)

  source = input_pipeline.load_source(config.dataset if source is None else source)
  eval_iter = input_pipeline.create_input_iter(source, config, train=False,
                                               cache=config.get("cache", True), rank=rank,
                                               world=world, device=device)
  n = source["label"].shape[0]
  steps_per_eval = n // config.eval_batch_size if config.steps_per_eval == -1 \
      else config.steps_per_eval                                  # eval.py:85-91

  if model is None:
    from . import models
    model = create_model(model_cls=getattr(models, config.model),
                         num_classes=config.num_classes, config=config)
  image_size = source["dvs_matrix"].shape[-2]                      # eval.py:98
  state = restore_checkpoint(None, workdir, model.apply, device)   # eval.py:104
  if state is None:
    key = torch.Generator()
    key.manual_seed(int(config.seed))
    state = create_eval_state(key, config, model, image_size, device)
  if config.get("prepare_params", False):
    from . import prune_utils
    state = EvalState(state.apply_fn, {"params": prune_utils.prepare_params(state.params["params"], config)},
                      state.batch_stats)

  loss_fn = config.get("loss_fn", mse_loss)
  smoothing = config.get("smoothing", 0.0)
  losses, accs = [], []
  for _ in range(steps_per_eval):                                  # eval.py:120-126
    batch = next(eval_iter)
    m = eval_step(state, batch, None, smoothing, loss_fn)
    losses.append(m["loss"].reshape(1).to(torch.float32))
    accs.append(m["accuracy"].to(torch.float32))
  # stack_forest + the all-gather that stands in for pmap's stacked outputs (eval.py:128):
  # loss [steps, world], accuracy [steps, world, B / world]
  loss = parallel.all_gather_rows(torch.stack(losses, 1)) if losses else torch.zeros((world, 0))
  acc = parallel.all_gather_rows(torch.stack(accs, 0).unsqueeze(0)) if accs else torch.zeros((world, 0, 0))
  summary = {"loss": float(loss.mean()) if loss.numel() else float("nan"),
             "accuracy": float(acc.mean()) if acc.numel() else float("nan"),
             "samples": int(acc.numel()), "steps": int(steps_per_eval), "world": world}
  if rank == 0:
    logging.info("Eval loss: %.4f, accuracy: %.2f", summary["loss"], summary["accuracy"] * 100)
  if device.type == "cuda":
    torch.cuda.synchronize()                                       # eval.py:136-137
  per_step = {"loss": loss.transpose(0, 1).cpu(), "accuracy": acc.transpose(0, 1).cpu()}
  return state, summary, per_step


def evaluate(config, workdir: str, **kw) -> EvalState:
  """examples/eval.py:53-139: returns the restored state, logs mean loss and accuracy."""
  return evaluate_metrics(config, workdir, **kw)[0]


def load_config(path: str):
  """A config file is a Python file with get_config() (ml_collections' convention)."""
  ns: dict = {"__file__": path, "__name__": "snnqp_config"}
  with open(path) as f:
    exec(compile(f.read(), path, "exec"), ns)
  return ns["get_config"]()


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
  ap.add_argument("--workdir", required=True, help="directory of the checkpoint (checkpoint_<step>)")
  ap.add_argument("--config", required=True, help="Python file with get_config()")
  args = ap.parse_args(argv)
  logging.basicConfig(level=logging.INFO)
  _, summary, _ = evaluate_metrics(load_config(args.config), args.workdir)
  if int(os.environ.get("RANK", "0")) == 0:
    import json
    print(json.dumps(summary))
  if torch.distributed.is_available() and torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
