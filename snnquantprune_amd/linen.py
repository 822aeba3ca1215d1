"""A small Flax-linen-shaped module protocol on torch tensors.

The reference's layers are ``flax.linen`` modules (flax 0.4.0, not installed
here and not a dependency of this package).  This file provides just the part
of that protocol the hot path uses, so the layers keep the reference's surface
(SURVEY.md section 8b):

  * dataclass-style modules with ``__call__`` as the compact method;
  * ``Module.init(rngs, *args, **kw) -> variables`` and
    ``Module.apply(variables, *args, mutable=..., rngs=..., **kw)``
    (call sites: examples/train_utils.py:144-158 and :377-386);
  * ``self.param`` / ``self.variable`` / ``self.sow`` /
    ``self.is_mutable_collection``;
  * the variable tree ``{collection: {ModuleName_i: {...}}}`` with Flax's
    auto-naming (``QuantConv_0``, ``DuQ_0``, ``prune_0``, ``BatchNorm_0`` ...),
    so trees written by the reference
    (examples/tcja/tcja_load_pretrained_weights.py:19-36) map one to one.

Arrays are torch tensors; forward only (no autodiff, no jit, no lifting).
"""

from __future__ import annotations

import dataclasses
import functools
import hashlib
import math
import threading
from collections.abc import MutableMapping
from typing import Any, Callable, Dict, Iterable, Optional, Sequence

import torch

_tls = threading.local()


def _stack():
  if not hasattr(_tls, "stack"):
    _tls.stack = []
  return _tls.stack


class ConfigDict(MutableMapping):
  """Attribute-style config (stand-in for ml_collections.ConfigDict).

  Missing keys raise AttributeError, like the reference's layers rely on
  (``config.prune_percentage`` at flax_qdense.py:84).  Not a dict subclass so
  that an empty config can be a dataclass field default, as in the reference
  (``config: dict = ml_collections.FrozenConfigDict({})``, flax_qdense.py:53)."""

  def __init__(self, *args, **kwargs):
    object.__setattr__(self, "_d", {})
    for k, v in dict(*args, **kwargs).items():
      self[k] = v

  def __getitem__(self, k):
    return self._d[k]

  def __setitem__(self, k, v):
    if isinstance(v, dict):
      v = ConfigDict(v)
    self._d[k] = v

  def __delitem__(self, k):
    del self._d[k]

  def __iter__(self):
    return iter(self._d)

  def __len__(self):
    return len(self._d)

  def __contains__(self, k):
    return k in self._d

  def __getattr__(self, k):
    try:
      return self._d[k]
    except KeyError:
      raise AttributeError(k)

  def __setattr__(self, k, v):
    self[k] = v

  def __delattr__(self, k):
    del self._d[k]

  def __hash__(self):
    return id(self)

  def __eq__(self, other):
    return self is other

  def __repr__(self):
    return "ConfigDict(%r)" % (self._d,)

  def to_dict(self):
    return {k: (v.to_dict() if isinstance(v, ConfigDict) else v)
            for k, v in self._d.items()}


FrozenConfigDict = ConfigDict


# ---------------------------------------------------------------------------
# compute dtype policy
# ---------------------------------------------------------------------------

# The kernels compute in exact integers and float32: the modules' default dtype
# (flax_qdense.py:49, flax_qconv.py:78) and the parity target.  The reference's shipped configs
# ask for bfloat16 (examples/tcja/configs/prune_quant_joint.py:71), which the layers refuse by
# default -- running a bfloat16 request in float32 silently would be a different computation than
# the one asked for.  A caller who wants those configs to load unedited says so:
#   nn.set_compute_dtype_policy("float32")   # any requested layer dtype is computed in float32
_DTYPE_POLICY = "strict"
_dtype_warned = False


def set_compute_dtype_policy(policy: str):
  """"strict" (default): layers refuse a dtype other than float32.  "float32": layers accept any
  requested dtype and compute in float32 (more precise than the bfloat16 the reference would run;
  said once on stderr)."""
  global _DTYPE_POLICY
  if policy not in ("strict", "float32"):
    raise ValueError("compute dtype policy must be 'strict' or 'float32'")
  _DTYPE_POLICY = policy


def check_compute_dtype(dtype, who: str):
  global _dtype_warned
  if dtype in (torch.float32, None, "float32"):
    return
  if _DTYPE_POLICY == "float32":
    if not _dtype_warned:
      import sys
      print("snnquantprune_amd: %s asked for %r; computing in float32 (nn.set_compute_dtype_policy)"
            % (who, dtype), file=sys.stderr)
      _dtype_warned = True
    return
  raise NotImplementedError(
      "%s computes in float32 (got %r); nn.set_compute_dtype_policy('float32') runs such a "
      "request in float32 instead of refusing it" % (who, dtype))


# ---------------------------------------------------------------------------
# initialisers (host side; never on the hot path)
# ---------------------------------------------------------------------------


def _default_device():
  return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def zeros(key, shape, dtype=torch.float32):
  return torch.zeros(tuple(shape), dtype=dtype, device=_default_device())


def ones(key, shape, dtype=torch.float32):
  return torch.ones(tuple(shape), dtype=dtype, device=_default_device())


def constant(value):
  def init(key, shape, dtype=torch.float32):
    return torch.full(tuple(shape), float(value), dtype=dtype, device=_default_device())
  return init


def lecun_normal():
  """Truncated normal, variance 1 / fan_in (flax default kernel_init)."""
  def init(key, shape, dtype=torch.float32):
    shape = tuple(shape)
    fan_in = 1
    for d in shape[:-1]:
      fan_in *= d
    std = math.sqrt(1.0 / max(fan_in, 1)) / 0.87962566103423978
    w = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std,
                                generator=key)
    return w.to(dtype).to(_default_device())
  return init


class Variable:
  def __init__(self, module, col, name):
    self._m, self._col, self._name = module, col, name

  @property
  def value(self):
    return self._m._get(self._col, self._name)

  @value.setter
  def value(self, v):
    root = self._m._root
    if not (root.initializing or self._col in root.mutable):
      raise RuntimeError("collection %r is immutable" % self._col)
    self._m._set(self._col, self._name, v)


class _Root:
  def __init__(self, variables, mutable, initializing, rngs):
    self.variables = variables
    self.mutable = set(mutable)
    self.initializing = initializing
    self.rngs = rngs
    self.rng_count = 0


def _seed_of(rngs, name):
  key = None
  if isinstance(rngs, dict):
    key = rngs.get(name, rngs.get("params"))
  else:
    key = rngs
  if key is None:
    return 0
  if isinstance(key, torch.Generator):
    return key.initial_seed()
  if isinstance(key, torch.Tensor):
    return int(key.reshape(-1)[0].item())
  return int(key)


class Module:
  """Base class.  Subclasses declare dataclass fields and define __call__."""

  name: Optional[str] = dataclasses.field(default=None, kw_only=True)

  def __init_subclass__(cls, **kw):
    super().__init_subclass__(**kw)
    if "__call__" in cls.__dict__:
      cls.__call__ = _wrap_call(cls.__dict__["__call__"])
    dataclasses.dataclass(cls, eq=False, repr=False)

  def __post_init__(self):
    object.__setattr__(self, "_parent", None)
    object.__setattr__(self, "_root", None)
    object.__setattr__(self, "_path", None)
    object.__setattr__(self, "_autoname", {})
    st = _stack()
    if st:
      self._adopt(st[-1], None)

  # -- binding -------------------------------------------------------------
  def _adopt(self, parent, attr_name):
    name = self.name or attr_name
    if name is None:
      base = type(self).__name__
      idx = parent._autoname.get(base, 0)
      parent._autoname[base] = idx + 1
      name = "%s_%d" % (base, idx)
    object.__setattr__(self, "_parent", parent)
    object.__setattr__(self, "_root", parent._root)
    object.__setattr__(self, "_path", parent._path + (name,))

  def _bind_root(self, root):
    object.__setattr__(self, "_parent", None)
    object.__setattr__(self, "_root", root)
    object.__setattr__(self, "_path", ())

  @property
  def is_bound(self):
    return self._root is not None

  # -- variable access -------------------------------------------------------
  def _node(self, col, create):
    d = self._root.variables
    if col not in d:
      if not create:
        return None
      d[col] = {}
    d = d[col]
    for p in self._path:
      if p not in d:
        if not create:
          return None
        d[p] = {}
      d = d[p]
    return d

  def _get(self, col, name):
    node = self._node(col, False)
    if node is None or name not in node:
      raise KeyError("no variable %s/%s/%s" % (col, "/".join(self._path), name))
    return node[name]

  def _set(self, col, name, value):
    self._node(col, True)[name] = value

  def has_variable(self, col, name):
    node = self._node(col, False)
    return node is not None and name in node

  def make_rng(self, name="params"):
    root = self._root
    root.rng_count += 1
    h = hashlib.sha256(("%d/%s/%s/%d" % (_seed_of(root.rngs, name), name,
                                         "/".join(self._path), root.rng_count)
                        ).encode()).digest()
    g = torch.Generator()
    g.manual_seed(int.from_bytes(h[:8], "little") & ((1 << 63) - 1))
    return g

  def param(self, name, init_fn, *init_args):
    if not self.is_bound:
      raise RuntimeError("module %s is not bound; use init()/apply()" % type(self).__name__)
    if not self.has_variable("params", name):
      if not self._root.initializing:
        raise KeyError("parameter %s/%s not found in variables" %
                       ("/".join(self._path), name))
      self._set("params", name, init_fn(self.make_rng("params"), *init_args))
    return self._get("params", name)

  def variable(self, col, name, init_fn=None, *init_args):
    if not self.has_variable(col, name):
      if not (self._root.initializing or col in self._root.mutable):
        raise KeyError("variable %s/%s/%s not found" % (col, "/".join(self._path), name))
      self._set(col, name, init_fn(*init_args) if init_fn is not None else None)
    return Variable(self, col, name)

  def is_mutable_collection(self, col):
    return self._root.initializing or col in self._root.mutable

  def sow(self, col, name, value):
    if not self.is_bound or not (col in self._root.mutable):
      return False
    node = self._node(col, True)
    node[name] = tuple(node.get(name, ())) + (value,)
    return True

  # -- entry points ------------------------------------------------------------
  def init(self, rngs, *args, **kwargs):
    """Flax Module.init: runs __call__ creating every variable; returns them."""
    root = _Root({}, mutable=(), initializing=True, rngs=rngs)
    return self._run(root, args, kwargs)[1]

  def init_with_output(self, rngs, *args, **kwargs):
    root = _Root({}, mutable=(), initializing=True, rngs=rngs)
    return self._run(root, args, kwargs)

  def apply(self, variables, *args, rngs=None, mutable=False, **kwargs):
    """Flax Module.apply.  With `mutable` returns (output, mutated collections)."""
    if mutable is True:
      mcols = set(variables.keys()) | {"intermediates"}
    elif not mutable:
      mcols = set()
    elif isinstance(mutable, str):
      mcols = {mutable}
    else:
      mcols = set(mutable)
    vars_copy = {c: (_copy_tree(v) if c in mcols else v) for c, v in variables.items()}
    root = _Root(vars_copy, mutable=mcols, initializing=False, rngs=rngs)
    out, allvars = self._run(root, args, kwargs)
    if mutable:
      return out, {c: allvars.get(c, {}) for c in mcols if c in allvars}
    return out

  def _run(self, root, args, kwargs):
    saved = (self._parent, self._root, self._path)
    st = _stack()
    saved_stack = list(st)
    del st[:]
    self._bind_root(root)
    try:
      out = self(*args, **kwargs)
    finally:
      object.__setattr__(self, "_parent", saved[0])
      object.__setattr__(self, "_root", saved[1])
      object.__setattr__(self, "_path", saved[2])
      st[:] = saved_stack
    return out, root.variables

  def __repr__(self):
    fs = ", ".join("%s=%r" % (f.name, getattr(self, f.name))
                   for f in dataclasses.fields(self)
                   if f.name != "name" and not isinstance(getattr(self, f.name), Module))
    return "%s(%s)" % (type(self).__name__, fs)


dataclasses.dataclass(Module, eq=False, repr=False)


def _copy_tree(t):
  if isinstance(t, dict):
    return {k: _copy_tree(v) for k, v in t.items()}
  return t


def _wrap_call(fn):
  @functools.wraps(fn)
  def wrapped(self, *args, **kwargs):
    st = _stack()
    if not self.is_bound:
      if not st:
        raise RuntimeError(
            "%s called outside init()/apply(): modules are stateless, use "
            "module.apply(variables, ...)" % type(self).__name__)
      parent = st[-1]
      attr = None
      for f in dataclasses.fields(parent):
        if getattr(parent, f.name, None) is self:
          attr = f.name
          break
      self._adopt(parent, attr)
    elif st and self._root is not st[-1]._root:
      # constructed under another (finished) init/apply: rebind under the caller
      parent = st[-1]
      object.__setattr__(self, "_root", parent._root)
    self._autoname.clear()           # compact methods restart their auto-names
    st.append(self)
    try:
      return fn(self, *args, **kwargs)
    finally:
      st.pop()
  return wrapped


def compact(fn):
  """No-op marker kept for source compatibility with `@nn.compact`."""
  return fn


# Methods other than __call__ that declare variables / submodules.
compact_method = _wrap_call


# ---------------------------------------------------------------------------
# BatchNorm (flax.linen.BatchNorm, eval mode) -- examples/tcja/models.py:101-107
# ---------------------------------------------------------------------------

from ._cache import TensorCache  # noqa: E402

_bn_cache = TensorCache(256)


class BatchNorm(Module):
  """Eval-mode batch normalisation over the last axis with running statistics:
  y = (x - mean) * (rsqrt(var + eps) * scale) + bias, each op rounded in
  float32.  params: scale, bias; batch_stats: mean, var.  Training-mode
  statistics are out of scope (forward / eval only)."""
  use_running_average: Optional[bool] = None
  axis: int = -1
  momentum: float = 0.99
  epsilon: float = 1e-5
  dtype: Any = torch.float32
  use_bias: bool = True
  use_scale: bool = True

  @compact_method
  def coeffs(self, features: int):
    """Folds (mean, var, scale, bias) to (mean, mul, bias) on the host in
    float32 (IEEE sqrt and divide), cached per tensor version."""
    from . import ops
    import numpy as np
    feat = (int(features),)
    mean = self.variable("batch_stats", "mean", lambda: zeros(None, feat)).value
    var = self.variable("batch_stats", "var", lambda: ones(None, feat)).value
    scale = self.param("scale", ones, feat) if self.use_scale else None
    bias = self.param("bias", zeros, feat) if self.use_bias else None
    hit = _bn_cache.get((mean, var, scale, bias), float(self.epsilon))
    if hit is not None:
      return hit
    dev = mean.device
    m = mean.detach().cpu().numpy().astype(np.float32)
    v = var.detach().cpu().numpy().astype(np.float32)
    mul = np.float32(1) / np.sqrt(v + np.float32(self.epsilon))
    if scale is not None:
      mul = mul * scale.detach().cpu().numpy().astype(np.float32)
    b = (bias.detach().cpu().numpy().astype(np.float32) if bias is not None
         else np.zeros_like(m))
    from . import _lib as L
    # a freshly initialised BatchNorm has zero running mean and zero bias: fl(x - 0) and
    # fl(x + 0) are x, and the fused kernels may skip those two instructions
    flags = (0 if m.any() else L.BN_MEAN_ZERO) | (0 if b.any() else L.BN_BIAS_ZERO)
    # ... and one multiplier for every channel (the same bits): a kernel whose channels share a
    # dequantisation table folds it into the entries (snnqp.h, SNNQP_BN_MUL_UNIFORM)
    mul32 = mul.astype(np.float32)
    if mul32.size and np.all(mul32.view(np.uint32) == mul32.view(np.uint32).flat[0]) and np.isfinite(mul32.flat[0]):
      flags |= L.BN_MUL_UNIFORM
    out = ops.BnCoeffs(torch.from_numpy(m).to(dev),
                       torch.from_numpy(mul.astype(np.float32)).to(dev),
                       torch.from_numpy(b).to(dev), flags)
    return _bn_cache.put((mean, var, scale, bias), float(self.epsilon), out)

  def __call__(self, x, use_running_average: Optional[bool] = None):
    from . import ops
    ura = self.use_running_average if use_running_average is None else use_running_average
    if ura is None:
      raise ValueError("BatchNorm needs use_running_average")
    if not ura:
      raise NotImplementedError(
          "BatchNorm batch statistics (training mode) are out of scope: this "
          "package implements the eval forward pass")
    if self.axis not in (-1, x.ndim - 1):
      raise NotImplementedError("BatchNorm over a non-last axis")
    return ops.batchnorm_forward(x, self.coeffs(x.shape[-1]))


# ---------------------------------------------------------------------------
# hipGraph capture of a whole apply(): for launch-bound models
# ---------------------------------------------------------------------------


class CapturedApply:
  """`module.apply(variables, inputs, **kwargs)` recorded once into a hipGraph and replayed.

  A small model (config C2: two dense blocks and the vote, 0.04 ms of kernels) is bound by
  the host's launch rate when every kernel is launched from Python; one graph launch per
  step removes that (1.4 M -> 3.8 M samples/s at B = 256).  The reference gets the same
  effect from `jax.jit` around the step (examples/eval.py:108-116).

    step = nn.capture(model, variables, example_inputs, trgt=None, train=False, rng=None)
    logits, _ = step(batch)          # copies `batch` into the captured input, replays

  The captured launch reads the input buffer, the packed weights and every intermediate at
  fixed addresses: `inputs` passed to a call must have the example's shape, dtype and format
  (torch tensor, ops.PackedSpikes or ops.PackedFrames), outputs are the same tensors on
  every call (copy what must survive the next call), and new weights need a new capture.
  Kernels that adapt to the data outside the graph (the event layer's count hint,
  ops.CountHint) run with the hint as it was at capture time -- still exact, see snnqp.h.
  No step of this package needs a value on the host any more (float32 inputs are checked on the
  device, snnqp.h x_flags): every model captures, float32 frames included.  Launches that hand
  partial results over between workgroups get a workspace allocated INSIDE the capture (the
  graph's private pool: ops._dense_workspace), so the addresses baked into the graph live and
  die with it.
  """

  def __init__(self, module, variables, example, **kwargs):
    from . import ops
    self._ops = ops
    dev = example.device if not isinstance(example, torch.Tensor) else example.device
    if torch.device(dev).type != "cuda":
      raise RuntimeError("capture needs the inputs on the GPU")
    self.static_input = self._clone(example)
    self._stream = torch.cuda.Stream(device=dev)
    self._graph = torch.cuda.CUDAGraph()
    cur = torch.cuda.current_stream(dev)
    self._stream.wait_stream(cur)
    self._device = torch.device(dev)
    self._slots = None
    with torch.cuda.stream(self._stream):
      for _ in range(2):               # packs, caches and the allocator's pools, outside the capture
        module.apply(variables, self.static_input, **kwargs)
      self._stream.synchronize()
      # the conv launches recorded below take a work-queue slot each (snnqp.h): remember which,
      # so that they go back to the pool with this object
      mark = ops.workqueue_capture_mark(self._device)
      # hand-over workspaces allocated inside the capture stay referenced by this object
      self._workspaces, outer = [], ops._CAPTURE_KEEP
      ops._CAPTURE_KEEP = self._workspaces
      try:
        with torch.cuda.graph(self._graph, stream=self._stream):
          self.static_output = module.apply(variables, self.static_input, **kwargs)
      finally:
        ops._CAPTURE_KEEP = outer
        self._slots = (mark, ops.workqueue_capture_mark(self._device))
    cur.wait_stream(self._stream)

  def close(self):
    """Destroys the graph and hands its work-queue slots back (after its last replay has
    completed).  Called by the destructor; idempotent."""
    slots, self._slots = self._slots, None
    if slots is None or slots[0] == slots[1]:
      self._graph = None
      return
    try:
      torch.cuda.synchronize(self._device)
      self._graph = None
      self._ops.workqueue_capture_release(self._device, slots[0], slots[1])
    except Exception:                   # interpreter shutdown: the pool dies with the process
      pass

  def __del__(self):
    self.close()

  def _clone(self, x):
    ops = self._ops
    if isinstance(x, ops.PackedFrames):
      return ops.PackedFrames(x.data.clone(), x.H, x.W, x.fmt)
    if isinstance(x, ops.PackedSpikes):
      return ops.PackedSpikes(x.bits.clone(), x.channels)
    return x.clone()

  def _raw(self, x):
    ops = self._ops
    return x.data if isinstance(x, ops.PackedFrames) else x.bits if isinstance(x, ops.PackedSpikes) else x

  def __call__(self, inputs=None):
    if inputs is not None and inputs is not self.static_input:
      src, dst = self._raw(inputs), self._raw(self.static_input)
      if src.shape != dst.shape or src.dtype != dst.dtype:
        raise ValueError("captured for inputs %s %s, got %s %s"
                         % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
      dst.copy_(src, non_blocking=True)
    if self._graph is None:
      raise RuntimeError("this CapturedApply has been closed")
    self._graph.replay()
    return self.static_output


def capture(module, variables, example, **kwargs) -> CapturedApply:
  """Records module.apply(variables, example, **kwargs) into a replayable hipGraph."""
  return CapturedApply(module, variables, example, **kwargs)


# ---------------------------------------------------------------------------
# tree helpers
# ---------------------------------------------------------------------------


def tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: tree_map(fn, v) for k, v in tree.items()}
  if isinstance(tree, (tuple, list)):
    return type(tree)(tree_map(fn, v) for v in tree)
  return fn(tree)


def tree_to_device(tree, device):
  return tree_map(lambda x: x.to(device) if isinstance(x, torch.Tensor) else x, tree)


def tree_from_numpy(tree, device=None):
  import numpy as np
  device = device or _default_device()
  return tree_map(lambda x: torch.as_tensor(np.asarray(x)).to(device)
                  if not isinstance(x, torch.Tensor) else x.to(device), tree)
