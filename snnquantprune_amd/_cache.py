"""Caches keyed by tensor identity + version (never by data_ptr, which the
allocator reuses): entries hold weak references and are validated on lookup."""
import weakref
from collections import OrderedDict


class TensorCache:
  def __init__(self, maxsize=256):
    self._d = OrderedDict()
    self._max = maxsize

  @staticmethod
  def _key(tensors, extra):
    return tuple(None if t is None else id(t) for t in tensors) + (extra,)

  def get(self, tensors, extra=None):
    key = self._key(tensors, extra)
    e = self._d.get(key)
    if e is None:
      return None
    refs, versions, value = e
    for t, r, v in zip(tensors, refs, versions):
      if t is None:
        if r is not None:
          return None
      elif r is None or r() is not t or t._version != v:
        del self._d[key]
        return None
    self._d.move_to_end(key)
    return value

  def put(self, tensors, extra, value):
    key = self._key(tensors, extra)
    refs = tuple(None if t is None else weakref.ref(t) for t in tensors)
    versions = tuple(None if t is None else t._version for t in tensors)
    self._d[key] = (refs, versions, value)
    while len(self._d) > self._max:
      self._d.popitem(last=False)
    return value

  def clear(self):
    self._d.clear()
