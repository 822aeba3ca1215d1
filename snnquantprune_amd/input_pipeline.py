"""Eval-side input pipeline -- mirror of examples/input_pipeline.py for datasets that are
already event-count frames in host memory (a `.npz` with `dvs_matrix` [N, T, H, W, 2]
uint8 and `label` [N], e.g. synthetic Poisson frames: there is no tfds and no network on
the boxes this runs on; turning raw events into such frames is `ops.events_to_frames`,
input_pipeline.py:142-219).

What is kept from the reference:
  * every PROCESS takes a contiguous slice of the split, `split_size = N // processes`,
    `start = process_index * split_size` (input_pipeline.py:245-254) -- one process per
    GPU here, so that slice is also the device shard (:38-46);
  * eval batches of `config.eval_batch_size // processes`, remainder dropped, repeated
    (:321-329);
  * two batches prefetched towards the device (:17-27) -- `feed.DeviceFeeder`.
What is new: `config.feed_format` ("u8" | "ev1" | "ev4") picks the wire format of the
frames (include/snnqp.h); with `cache` the slice is packed once into page-locked memory.
"""

from __future__ import annotations

from typing import Any, Dict, Iterator

import numpy as np
import torch

from . import _lib as L
from . import feed, ops

FEED_FORMATS = {"u8": None, "ev1": L.EV1, "ev4": L.EV4}


def load_source(source) -> Dict[str, np.ndarray]:
  if isinstance(source, str):
    with np.load(source) as z:
      source = {k: z[k] for k in z.files}
  x, y = np.asarray(source["dvs_matrix"]), np.asarray(source["label"])
  if x.ndim != 5 or x.shape[-1] != 2 or x.shape[0] != y.shape[0]:
    raise ValueError("dvs_matrix must be [N, T, H, W, 2] with one label per sample, got %s / %s"
                     % (x.shape, y.shape))
  return {"dvs_matrix": x, "label": y.astype(np.int8)}            # :271 casts labels to int8


def create_split(source, config, train: bool, cache: bool, rank: int = 0, world: int = 1,
                 pin: bool = False) -> Iterator[Dict[str, Any]]:
  """Host-side iterator of this process's batches (input_pipeline.py:222-345)."""
  if train:
    raise NotImplementedError("the training split (shuffle, augmentation) is out of scope")
  data = load_source(source)
  n = data["label"].shape[0]
  split = n // world                                               # :250-254
  lo = rank * split
  x, y = data["dvs_matrix"][lo:lo + split], data["label"][lo:lo + split]
  per = config.eval_batch_size // world                            # :326
  if per < 1 or split < per:
    raise ValueError("eval_batch_size %d over %d processes needs at least %d samples per "
                     "process, the split has %d" % (config.eval_batch_size, world, max(per, 1), split))
  fmt = FEED_FORMATS[config.get("feed_format", "u8") if hasattr(config, "get") else "u8"]

  def wire(frames):
    if fmt is None:
      return torch.from_numpy(np.ascontiguousarray(frames))
    return ops.pack_frames_host(frames.astype(np.uint8, copy=False), fmt)

  nb = split // per                                                # drop_remainder=True
  if cache:                 # ds.cache(): decode (here: pack) the slice once
    wx = wire(x[:nb * per])
    wy = torch.from_numpy(np.ascontiguousarray(y[:nb * per]))
    if pin:
      wx, wy = feed.pinned_like(wx), feed.pinned_like(wy)

  def gen():
    while True:                                                    # ds.repeat()
      for i in range(nb):
        if cache:
          bx = wx.narrow(0, i * per, per) if isinstance(wx, ops.PackedFrames) else wx[i * per:(i + 1) * per]
          by = wy[i * per:(i + 1) * per]
        else:
          bx = wire(x[i * per:(i + 1) * per])
          by = torch.from_numpy(np.ascontiguousarray(y[i * per:(i + 1) * per]))
        yield {"dvs_matrix": bx, "label": by}
  return gen()


def create_input_iter(source, config, train: bool, cache: bool, rank: int = 0, world: int = 1,
                      device=None):
  """create_input_iter of input_pipeline.py:17-27: the split, prefetched two batches deep
  to this process's device."""
  device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else device
  ds = create_split(source, config, train=train, cache=cache, rank=rank, world=world,
                    pin=torch.device(device).type == "cuda")
  return feed.DeviceFeeder(ds, device, 2)
