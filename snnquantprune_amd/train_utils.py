"""Eval-side step functions -- mirror of examples/train_utils.py (eval_step
:370-390, compute_metrics :220-225, mse_loss :210-217, cross_entropy_loss
:196-207, create_model :133-134).  Training (train_step, optimisers, LR
schedules, checkpoint writing) is out of scope.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable

import torch


@dataclass
class EvalState:
  """The fields of the reference's TrainState that eval_step reads."""
  apply_fn: Callable
  params: dict          # {'params': ...} as in train_utils.py:187-192
  batch_stats: dict


def create_model(*, model_cls, num_classes, model_dtype=torch.float32, **kwargs):
  return model_cls(num_classes=num_classes, dtype=model_dtype, **kwargs)


def onehot(labels, num_classes):
  labels = torch.as_tensor(labels).to(torch.int64)
  return torch.nn.functional.one_hot(labels, num_classes).to(torch.float32)


def cross_entropy_loss(logits, labels, smoothing=0):
  oh = onehot(labels, logits.shape[1]).to(logits.device)
  oh = oh * (1 - smoothing) + smoothing / oh.shape[1]
  return torch.mean(-(oh * torch.log_softmax(logits, -1)).sum(-1))


def mse_loss(logits, labels, smoothing=0, T=1):
  oh = onehot(labels, logits.shape[1]).to(logits.device)
  oh = oh * (1 - smoothing) + smoothing / oh.shape[1]
  return torch.mean(torch.square(logits / T - oh))


def compute_metrics(logits, labels, smoothing, loss_fn):
  loss = loss_fn(logits, labels, smoothing)
  accuracy = torch.argmax(logits, -1) == torch.as_tensor(labels).to(logits.device)
  return {"loss": loss, "accuracy": accuracy}


def eval_step(state, batch, rng, smoothing, loss_type, burnin=0):
  variables = {"params": state.params["params"], "batch_stats": state.batch_stats}
  (logits, _), _ = state.apply_fn(
      variables, batch["dvs_matrix"], trgt=batch["label"], train=False,
      online=False, rng=rng, mutable=["batch_stats"], rngs={"dropout": rng})
  return compute_metrics(logits, batch["label"], smoothing, loss_type)
