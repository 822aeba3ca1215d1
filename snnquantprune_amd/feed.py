"""Host -> device feed of eval batches: the counterpart of
`flax.jax_utils.prefetch_to_device(it, 2)` at examples/input_pipeline.py:17-27.

The reference keeps two batches in flight towards the devices while the current one
computes.  Here: a ring of depth + 2 device buffers per GPU process, filled by
`hipMemcpyAsync` from page-locked host memory on a dedicated copy stream, handed to
the compute stream through events -- batch k + 1 and k + 2 cross PCIe while batch k
runs, and nothing on the host waits for a copy.

What crosses the link is the iterator's business: uint8 frames [B, T, H, W, 2]
(655 360 B per DVS128 sample -- at the kernels' 76 k samples/s that is the whole of a
PCIe gen5 x16 link), or `ops.PackedFrames` in one of the wire formats of include/snnqp.h
(bit-packed binary frames 81 920 B, nibble-packed counts 327 680 B), which the first
conv block reads as they are.
"""

from __future__ import annotations

import collections
from typing import Any, Dict, Iterable, Iterator, Optional

import numpy as np
import torch

from . import ops


def _leaf(v):
  """(host tensor, rebuild(device tensor) -> what the consumer gets)."""
  if isinstance(v, ops.PackedFrames):
    return v.data, (lambda d, v=v: ops.PackedFrames(d, v.H, v.W, v.fmt))
  if isinstance(v, np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(v)), (lambda d: d)
  if isinstance(v, torch.Tensor):
    return v.contiguous(), (lambda d: d)
  raise TypeError("batch leaves must be numpy arrays, torch tensors or PackedFrames, got %s"
                  % type(v).__name__)


class DeviceFeeder:
  """Iterates `it` (dict batches in host memory) `depth` batches ahead, returning the same
  dicts with every leaf resident on `device`.

  * a leaf already in page-locked memory (`tensor.pin_memory()`, `pinned_like`) is DMA-ed
    as it is; any other leaf is first copied into the slot's own page-locked staging
    buffer (a host memcpy on the calling thread: a loader that wants the link's full rate
    writes its batches into pinned memory itself);
  * a device buffer is overwritten only after the compute stream has passed the point
    where the consumer asked for the batch AFTER the one that lived in it (event recorded
    in __next__), so a batch stays valid until the next `next()` call returns;
  * on a CPU `device` (tests) the feeder degenerates to the plain iterator.
  """

  def __init__(self, it: Iterable[Dict[str, Any]], device, depth: int = 2):
    self._it: Iterator = iter(it)
    self.device = torch.device(device)
    self.depth = int(depth)
    assert self.depth >= 1
    self._gpu = self.device.type == "cuda"
    self._queue = collections.deque()
    self._nslots = self.depth + 2
    self._slots = [dict() for _ in range(self._nslots)]      # key -> [pinned staging | None, device buffer]
    self._free = [None] * self._nslots                        # compute-stream event: slot reusable
    self._copied = [None] * self._nslots                      # copy-stream event: slot's H2D done
    self._next_slot = 0
    self._last_slot: Optional[int] = None
    self._done = False
    self.bytes_copied = 0
    self.batches = 0
    self._stream = torch.cuda.Stream(device=self.device) if self._gpu else None
    for _ in range(self.depth):
      self._enqueue()

  def _enqueue(self):
    if self._done:
      return
    try:
      host = next(self._it)
    except StopIteration:
      self._done = True
      return
    if not self._gpu:
      self._queue.append((host, None, None))
      return
    slot = self._next_slot
    self._next_slot = (slot + 1) % self._nslots
    bufs = self._slots[slot]
    out = {}
    leaves = {key: _leaf(val) for key, val in host.items()}
    # device buffers are allocated on the CONSUMER's stream (the caching allocator ties a block
    # to the stream current at allocation: freed later, it must not be handed to a new user
    # while the consumer's kernels still read it); the copy stream only ever writes into them,
    # ordered against the consumer by the events below
    fresh = False
    for key, (src, _) in leaves.items():
      ent = bufs.get(key)
      if ent is None or ent[1].shape != src.shape or ent[1].dtype != src.dtype:
        bufs[key] = [None, torch.empty(src.shape, dtype=src.dtype, device=self.device)]
        fresh = True
    with torch.cuda.stream(self._stream):
      if self._free[slot] is not None:
        self._stream.wait_event(self._free[slot])
      if fresh:                            # memory the consumer's stream may just have released
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
      for key, (src, rebuild) in leaves.items():
        ent = bufs[key]
        if not src.is_pinned():
          if ent[0] is None:
            ent[0] = torch.empty(src.shape, dtype=src.dtype).pin_memory()
          if self._copied[slot] is not None:
            self._copied[slot].synchronize()      # the staging buffer's last DMA (long done)
          ent[0].copy_(src)
          src = ent[0]
        ent[1].copy_(src, non_blocking=True)
        self.bytes_copied += src.numel() * src.element_size()
        out[key] = rebuild(ent[1])
      ev = torch.cuda.Event()
      ev.record(self._stream)
    self._copied[slot] = ev
    self._queue.append((out, ev, slot))

  def __iter__(self):
    return self

  def __next__(self):
    if not self._queue:
      raise StopIteration
    batch, ev, slot = self._queue.popleft()
    if self._gpu:
      cur = torch.cuda.current_stream(self.device)
      cur.wait_event(ev)
      if self._last_slot is not None:
        # everything the consumer enqueued for the previous batch lies before this point
        done = torch.cuda.Event()
        done.record(cur)
        self._free[self._last_slot] = done
      self._last_slot = slot
    self.batches += 1
    self._enqueue()
    return batch


def pinned_like(v):
  """A page-locked copy of a host leaf (numpy array, CPU tensor or PackedFrames): what a
  loader hands to DeviceFeeder so that the H2D copy is one DMA with no staging memcpy."""
  if isinstance(v, ops.PackedFrames):
    return ops.PackedFrames(v.data.pin_memory(), v.H, v.W, v.fmt)
  if isinstance(v, np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(v)).pin_memory()
  return v.contiguous().pin_memory()
