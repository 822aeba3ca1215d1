"""CPU suite, part 2: host logic -- the C-ABI library loads and exports every
symbol include/snnqp.h declares (no compute without a GPU), the module
protocol produces the reference's variable tree, argument errors surface as the
reference's exceptions, and the data-parallel path works on gloo."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
  from snnquantprune_amd import _lib as L
  if not os.path.exists(L.LIB_PATH):
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
  return L


def test_library_exports_every_declared_symbol():
  L = _lib()
  hdr = open(os.path.join(ROOT, "include", "snnqp.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  declared = set(re.findall(r"\b(snnqp_[a-z0-9_]+)\s*\(", hdr))
  assert declared, "no declarations parsed"
  lib = L.lib()
  for name in sorted(declared):
    assert hasattr(lib, name), "libsnnqp.so does not export %s" % name
  assert declared == set(L.EXPORTED_SYMBOLS)
  assert lib.snnqp_version() == L.ABI_VERSION
  # the in-tree library is the product build: no diagnostic switch compiled in
  assert L.build_flags() == ""
  L.require_product_build()


def test_argument_errors_are_reported_without_a_gpu():
  L = _lib()
  lib = L.lib()
  # quant.py:332-336: bit widths below 2 are rejected
  rc = lib.snnqp_quantize(L.Q_UNIFORM_STATIC, ctypes.c_void_p(8), None, 4, 1, 1.0, 0.0,
                          None, None, None, None)
  assert rc == L.EINVAL and b"bits" in lib.snnqp_last_error()
  with pytest.raises(L.SnnqpError):
    L.check(rc)
  g = L.ConvGeomT(28, 28, 3, 8, 2, 2, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 2)
  w = L.WeightT(L.W_F32, 8, 1.0, 1.0)
  rc = lib.snnqp_conv_forward(ctypes.c_void_p(8), L.F32, 1, ctypes.byref(g), ctypes.byref(w),
                              ctypes.c_void_p(8), None, None)
  assert rc == L.EINVAL and b"feature_group_count" in lib.snnqp_last_error()  # flax_qconv.py:117
  rc = lib.snnqp_vote(ctypes.c_void_p(8), L.F32, 4, 2, 110, 7, ctypes.c_void_p(8), None)
  assert rc == L.EINVAL
  # a BatchNorm descriptor with flag bits this build does not know (a caller built against
  # an older snnqp.h leaves `flags` uninitialised) is refused, not read as "mean and bias zero"
  g3 = L.ConvGeomT(8, 8, 2, 128, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1)
  wi = L.WeightT(L.W_I8, 8, 7.0, 1.0, 10, 7, 0)
  nrn = L.NeuronT(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0, None)
  bad = L.BnT(8, 8, 8, 0x7F3)
  rc = lib.snnqp_conv_lif_forward(ctypes.c_void_p(8), L.U8, 128, 128 * 4, 4, 1, ctypes.byref(g3),
                                  ctypes.byref(wi), None, ctypes.byref(bad), ctypes.byref(nrn),
                                  None, None, ctypes.c_void_p(8), L.BITS, 2, L.IMPL_MFMA, 1, None, None, None)
  assert rc == L.EINVAL and b"unknown flag bits" in lib.snnqp_last_error()
  # float32 input into integer codes without the word its check reports into (snnqp.h, x_flags)
  rc = lib.snnqp_conv_lif_forward(ctypes.c_void_p(8), L.F32, 128, 128 * 4, 4, 1, ctypes.byref(g3),
                                  ctypes.byref(wi), None, None, ctypes.byref(nrn),
                                  None, None, ctypes.c_void_p(8), L.BITS, 2, L.IMPL_MFMA, 1, None, None, None)
  assert rc == L.EINVAL and b"x_flags" in lib.snnqp_last_error()


def test_conv_out_shape_matches_reference_table():
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd.flax_qconv import QuantConv
  from tests import cases
  _lib()
  for name, H, W, ks, st, pad, OH, OW in cases.REF_CONV_GEOMS:
    conv = QuantConv(features=10, kernel_size=ks, strides=st, padding=pad,
                     config=nn.ConfigDict({"prune_percentage": -1.0}))
    assert conv.geometry((H, W), 1).out_hw() == (OH, OW), name
    conv2 = QuantConv(features=20, kernel_size=ks, strides=st, padding=pad,
                      config=nn.ConfigDict({"prune_percentage": -1.0}))
    assert conv2.geometry((OH, OW), 10).out_hw() == cases.REF_CONV_TWICE[name], name
  c1 = QuantConv(features=8, kernel_size=[4], padding="SAME",
                 config=nn.ConfigDict({"prune_percentage": -1.0}))
  g = c1.geometry((20,), 128)
  assert g.pad == ((0, 0), (1, 2)) and g.out_hw() == (1, 20)      # TCJA convs, k = 4


def test_ops_refuse_cpu_tensors():
  from snnquantprune_amd import ops
  _lib()
  with pytest.raises(RuntimeError, match="GPU only"):
    ops.pack_bits(torch.zeros(4, 32))
  with pytest.raises(RuntimeError, match="GPU only"):
    ops.vote(torch.zeros(2, 2, 20))


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
  from snnquantprune_amd import _lib as L
  monkeypatch.setattr(L, "_lib", None)
  monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
  with pytest.raises(ImportError, match="no CPU fallback"):
    L.lib()


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "snnquantprune_amd")
  for dp, _, fs in os.walk(pkg):
    for f in fs:
      if f.endswith((".py", ".hip", ".h", ".cpp")):
        src = open(os.path.join(dp, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
        assert "snn_oracle" not in src, f


# ---- module protocol ---------------------------------------------------------------

def test_module_protocol_names_and_mutability():
  from snnquantprune_amd import linen as nn

  class Leaf(nn.Module):
    feats: int

    def __call__(self, x):
      w = self.param("kernel", nn.lecun_normal(), (x.shape[-1], self.feats))
      cnt = self.variable("batch_stats", "count", lambda: torch.zeros(()))
      if self.is_mutable_collection("batch_stats"):
        cnt.value = cnt.value + 1
      return x @ w.cpu()

  class Holder(nn.Module):
    inner: object

    def __call__(self, x):
      return self.inner(x)

  class Top(nn.Module):
    config: dict = nn.FrozenConfigDict({})

    def __call__(self, x, train=False):
      a, b = Leaf(4), Leaf(3)
      y = Holder(inner=a)(x)      # `a` keeps the name it got in Top's scope
      y = a(x) + y                # second call shares the same parameters
      return b(y)

  top = Top()
  v = top.init({"params": 1}, torch.ones(2, 5))
  assert sorted(v["params"]) == ["Leaf_0", "Leaf_1"]
  assert v["params"]["Leaf_0"]["kernel"].shape == (5, 4)
  out = top.apply(v, torch.ones(2, 5))
  out2, mut = top.apply(v, torch.ones(2, 5), mutable=["batch_stats"])
  assert torch.equal(out, out2)
  assert float(mut["batch_stats"]["Leaf_0"]["count"]) == float(
      v["batch_stats"]["Leaf_0"]["count"]) + 2
  assert float(v["batch_stats"]["Leaf_0"]["count"]) == 2      # input tree untouched
  with pytest.raises(KeyError):
    top.apply({"params": {}}, torch.ones(2, 5))
  with pytest.raises(RuntimeError):
    Leaf(3)(torch.ones(2, 5))                                  # unbound call


def test_config_dict_behaves_like_the_reference_expects():
  from snnquantprune_amd import linen as nn
  cfg = nn.ConfigDict({"quant": {"bits": 4}})
  assert cfg.quant.bits == 4 and "weight" not in cfg.quant
  with pytest.raises(AttributeError):          # flax_qdense.py:84 on stale configs
    cfg.quant.prune_percentage
  cfg.quant.prune_percentage = 0.9
  assert cfg.quant["prune_percentage"] == 0.9


def test_synthetic_trees_use_reference_names():
  from snnquantprune_amd import synthetic as syn
  v = syn.conv_net_variables(hw=16)
  assert sorted(v["params"]) == ["BatchNorm_0", "BatchNorm_1", "BatchNorm_2", "QuantConv_0",
                                 "QuantConv_1", "QuantConv_2", "QuantDense_0"]
  leaf = v["params"]["QuantConv_1"]
  assert sorted(leaf) == ["DuQ_0", "kernel", "prune_0"]
  assert leaf["kernel"].shape == (3, 3, 128, 128) and leaf["DuQ_0"]["a"].shape == (1,)
  assert abs(1 - leaf["prune_0"]["mask"].mean() - 0.9) < 1e-3
  assert sorted(v["batch_stats"]["BatchNorm_0"]) == ["mean", "var"]


def test_shard_bounds_error_matches_eval_py():
  from snnquantprune_amd import parallel
  assert parallel.shard_bounds(16, 3, 8) == (6, 8)
  with pytest.raises(ValueError, match="divisible"):
    parallel.shard_bounds(10, 0, 8)


_GLOO_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from snnquantprune_amd import parallel
rank, world, _ = parallel.init_from_env("gloo")
assert world == 2
x = torch.arange(8 * 3, dtype=torch.float32).reshape(8, 3)
def apply_fn(variables, inp, **kw):        # stand-in for model.apply
  return (inp.sum(-1, keepdim=True) * variables["s"], None), {}
full = parallel.sharded_logits(apply_fn, {"s": 2.0}, x, rank, world)
assert full.shape == (8, 1)
assert torch.equal(full, x.sum(-1, keepdim=True) * 2.0), full
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_eval_on_gloo_world_size_2(tmp_path):
  script = tmp_path / "worker.py"
  script.write_text(_GLOO_WORKER % ROOT)
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
           for r in range(2)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for p, o in zip(procs, outs):
    assert p.returncode == 0, o
    assert "ok" in o


def test_checkpoint_import_roundtrip_and_torch_layouts(tmp_path):
  """F2: the Flax msgpack wire format (ext 1 = ndarray) and the PyTorch layout
  contract of tcja_load_pretrained_weights.py:19-36,117,127,137-139."""
  from snnquantprune_amd import checkpoint, synthetic as syn
  v = syn.cextnet_variables(frames=4, hw=32)
  state = {"step": np.int32(7), "params": {"params": v["params"]},
           "batch_stats": v["batch_stats"], "opt_state": {"count": np.float32(1.5)}}
  path = tmp_path / "checkpoint_7"
  path.write_bytes(checkpoint.msgpack_serialize(state))
  got = checkpoint.load_flax_checkpoint(str(path))
  assert sorted(got["params"]) == sorted(v["params"])
  for name, leaf in v["params"].items():
    for k, val in leaf.items():
      if isinstance(val, dict):
        for kk, vv in val.items():
          np.testing.assert_array_equal(got["params"][name][k][kk], vv)
      else:
        np.testing.assert_array_equal(got["params"][name][k], val)
  np.testing.assert_array_equal(got["batch_stats"]["BatchNorm_2"]["var"],
                                v["batch_stats"]["BatchNorm_2"]["var"])
  # torch state dict -> tree
  rng = np.random.default_rng(0)
  net = {"conv.0.0.weight": torch.from_numpy(rng.standard_normal((128, 2, 3, 3)).astype(np.float32)),
         "conv.0.1.weight": torch.ones(128), "conv.0.1.bias": torch.zeros(128),
         "conv.0.1.running_mean": torch.full((128,), 0.5), "conv.0.1.running_var": torch.ones(128),
         "conv.0.1.num_batches_tracked": torch.tensor(3),
         "conv.11.conv.weight": torch.from_numpy(rng.standard_normal((20, 20, 4)).astype(np.float32)),
         "fc.5.0.weight": torch.from_numpy(rng.standard_normal((110, 512)).astype(np.float32))}
  tree = checkpoint.from_torch_state_dict(net)
  k0 = tree["params"]["QuantConv_0"]["kernel"]
  assert k0.shape == (3, 3, 2, 128)
  assert k0[1, 2, 0, 5] == net["conv.0.0.weight"][5, 0, 1, 2].item()
  assert tree["params"]["QuantConv_4"]["kernel"].shape == (4, 20, 20)
  assert tree["params"]["QuantDense_1"]["kernel"].shape == (512, 110)
  assert tree["params"]["QuantDense_1"]["kernel"][3, 7] == net["fc.5.0.weight"][7, 3].item()
  assert float(tree["params"]["QuantConv_0"]["DuQ_0"]["a"][0]) == -1.0
  assert tree["params"]["QuantConv_0"]["prune_0"]["mask"].min() == 1.0
  assert tree["batch_stats"]["BatchNorm_0"]["mean"][0] == 0.5
  assert sorted(tree["params"]["BatchNorm_0"]) == ["bias", "scale"]


def test_prune_utils_on_cpu_tensors():
  """Mask / a, c builders are host-side: they run on CPU tensors too."""
  from oracle import snn_oracle as o
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import prune_utils, synthetic as syn
  from snnquantprune_amd.quant import gaussian_init
  v = syn.dense_net_variables(K=64, hidden=32, out=20, prune_p=-1.0)
  params = nn.tree_from_numpy(v["params"], torch.device("cpu"))
  g = prune_utils.update_global_prune_mask(params, 0.5)
  ks = [v["params"]["QuantDense_0"]["kernel"], v["params"]["QuantDense_1"]["kernel"]]
  for name, m in zip(("QuantDense_0", "QuantDense_1"), o.global_prune_masks(ks, 0.5)):
    np.testing.assert_array_equal(g[name]["prune_0"]["mask"].numpy(), m)
  q = prune_utils.update_quant_params(params, gaussian_init, 4)
  # bit-equal: `a` sits inside round(), a one-ulp difference flips codes at ties
  for name, k in zip(("QuantDense_0", "QuantDense_1"), ks):
    a = q[name]["DuQ_0"]["a"].numpy()
    assert a.dtype == np.float32 and a.shape == (1,)
    np.testing.assert_array_equal(a[0], np.float32(o.gaussian_init(k, 4)))
    np.testing.assert_array_equal(q[name]["DuQ_0"]["c"].numpy(), a)


# ---------------------------------------------------------------------------
# bench.py plumbing: self-launch, rank environment, metric label
# ---------------------------------------------------------------------------


def _bench(args, env=None, timeout=300, detail=None):
  extra = ["--detail", detail] if detail else ["--detail", os.devnull]
  p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + extra, env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
  return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_bench_gpus_2_starts_its_own_ranks_on_gloo(tmp_path):
  """`python bench.py --gpus 2` with no launcher around it: the parent spawns the ranks, the
  ranks run bench.py's own step / fence / all-gather / all_reduce(MAX) code (gloo, stand-in
  for model.apply, which also checks the gathered rows rank by rank) and rank 0's line
  comes back."""
  env = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  detail = str(tmp_path / "detail.json")
  rc, out, err = _bench(["--gpus", "2", "--backend", "gloo", "--stand-in", "--batch", "6",
                         "--frames", "3", "--steps", "3", "--warmup", "1"], env, detail=detail)
  assert rc == 0, err
  lines = [l for l in out.splitlines() if l.startswith("{")]
  assert len(lines) == 1 and out.rstrip("\n").splitlines()[-1] == lines[0], out
  short = json.loads(lines[0])
  # stdout carries the compact line; the per-rank accounts are in the detail file it names
  assert len(lines[0]) < 6000 and short["detail"] == detail and "rank_detail" not in short
  assert short["ranks_seen"] == 2 and len(short["rank_seconds"]) == 2
  d = json.load(open(detail))
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
    assert short[k] == d[k]
  assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
  assert d["config"]["global_batch"] == 12 and d["config"]["batch_per_gpu"] == 6
  assert d["scaling"] == "weak" and d["unit"] == "samples/s"
  assert abs(d["value"] - 12 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
  # the line shows that the collective spanned both ranks, and each rank's own time
  assert d["ranks_seen"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
  assert len({r["pid"] for r in d["ranks"]}) == 2
  assert len(d["rank_seconds"]) == 2 and max(d["rank_seconds"]) == pytest.approx(d["ms_per_step"] * 3e-3)
  # ... and each rank's own account (seconds, per-layer kernel times, fallback counters: the
  # stand-in launches no kernel), gathered into rank 0's line in rank order
  assert [r["rank"] for r in d["rank_detail"]] == [0, 1]
  assert [r["seconds"] for r in d["rank_detail"]] == d["rank_seconds"]
  assert all(r["kernels"] == {} for r in d["rank_detail"])
  # every rank announced itself on stderr
  assert err.count("bench.py rank {") >= 1


def test_bench_strong_scaling_splits_a_fixed_global_batch():
  """--scaling strong: the global batch (BASELINE config C4: 8192) is fixed and split over the
  ranks; an indivisible one is refused."""
  env = {k: v for k, v in os.environ.items()
         if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
  rc, out, err = _bench(["--gpus", "2", "--backend", "gloo", "--stand-in", "--scaling", "strong",
                         "--global-batch", "10", "--frames", "3", "--steps", "2", "--warmup", "1"], env)
  assert rc == 0, err
  d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
  assert d["scaling"] == "strong" and d["n_gpus"] == 2
  assert d["config"]["global_batch"] == 10 and d["config"]["batch_per_gpu"] == 5
  assert abs(d["value"] - 10 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
  rc, out, err = _bench(["--gpus", "2", "--backend", "gloo", "--stand-in", "--scaling", "strong",
                         "--global-batch", "9", "--frames", "3", "--steps", "1", "--warmup", "0"], env)
  assert rc != 0 and "divisible" in err


def test_bench_under_an_external_launcher_env():
  """The driver's form: WORLD_SIZE / RANK / MASTER_* come from torch.distributed.run."""
  env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
  args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
          "--stand-in", "--batch", "4", "--frames", "2", "--steps", "2", "--warmup", "1", "--detail", os.devnull]
  procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
  outs = [p.communicate(timeout=300) for p in procs]
  for p, (o, e) in zip(procs, outs):
    assert p.returncode == 0, e.decode()
  d = json.loads([l for l in outs[0][0].decode().splitlines() if l.startswith("{")][0])
  assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8
  assert not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]   # rank 0 only


def test_bench_launcher_fails_when_a_rank_fails():
  """No GPU here: the nccl ranks die on their first assert and the parent must say so."""
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
  env["SNNQP_BENCH_LAUNCH_TIMEOUT"] = "240"
  import torch
  if torch.cuda.is_available():
    pytest.skip("needs a box without a GPU")
  rc, out, err = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2"], env)
  assert rc != 0
  assert "failed" in err and not out.strip()


def test_bench_world_size_mismatch_is_an_error():
  env = dict(os.environ, WORLD_SIZE="1", RANK="0")
  rc, out, err = _bench(["--gpus", "2", "--backend", "gloo", "--stand-in"], env)
  assert rc != 0 and "WORLD_SIZE" in err


def test_bench_metric_label_follows_the_arguments():
  sys.path.insert(0, ROOT)
  import bench
  assert bench.metric_name(bench.parse([])) == "samples/sec/node (DVS128 T=20, 4-bit/90%-pruned)"
  assert bench.metric_name(bench.parse(["--bits", "8", "--prune", "0.3"])) == \
      "samples/sec/node (DVS128 T=20, 8-bit/30%-pruned)"
  assert bench.metric_name(bench.parse(["--frames", "50", "--layer-bits", "2,4,2,4", "--prune",
                                        "0.95"])) == \
      "samples/sec/node (DVS128 T=50, mixed 2/4-bit/95%-pruned)"
  c2 = bench.parse(["--model", "dense", "--batch", "256", "--bits", "8", "--prune", "0.5", "--graph"])
  assert c2.graph and bench.metric_name(c2) == \
      "samples/sec/node (2-layer qdense 2048-512-110, T=20, 8-bit/50%-pruned)"
  # the roofline block of the C2 line: the first dense block is the dense layer the metric names
  r = bench.rooflines_of(c2, {"dense[2048->512]": (10, 0.2), "dense[512->110]": (10, 0.2)}, 256, 20, [8] * 4)
  assert r["roofline_dense"]["kernel"] == "dense[2048->512]" and r["roofline"]["kernel"] == "dense kernel"
  # (the bytes of the format the kernel really reads: uint8 rows in place, 2048 B per sample-step)
  nbytes = 256 * 20 * (2048 + 64) + 2048 * 512
  assert abs(r["roofline_dense"]["achieved"] - nbytes / 0.02e-3 / 1e9) < 1e-6
  # ... and as one launch (snnqp_dense_head_forward), which is what the model runs when it fits
  r = bench.rooflines_of(c2, {"dense_head[2048->512->110]": (10, 0.3)}, 256, 20, [8] * 4)
  assert r["roofline_dense"]["kernel"] == "dense_head[2048->512->110]"
  nbytes = 256 * 20 * 2048 + 2048 * 512 + 512 * 128 + 256 * 4 * 11
  assert abs(r["roofline_dense"]["achieved"] - nbytes / 0.03e-3 / 1e9) < 1e-6


def _flax_blob_module():
  import importlib.util
  path = os.path.join(ROOT, "tests", "golden", "make_flax_blob.py")
  spec = importlib.util.spec_from_file_location("make_flax_blob", path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_flax_checkpoint_importer_against_independent_bytes():
  """F2: tests/golden/flax_checkpoint_tiny.msgpack was assembled byte by byte from the
  msgpack spec and flax.serialization's ext layout (tests/golden/make_flax_blob.py uses
  neither the msgpack package nor this package's writer).  The committed file is what the
  generator produces, and the importer reads every leaf back bit for bit, fills the leaves
  an unpruned run does not save (mask = 1) and ignores the optimiser state."""
  from snnquantprune_amd import checkpoint
  mk = _flax_blob_module()
  with open(mk.OUT, "rb") as f:
    blob = f.read()
  assert blob == mk.p_map(mk.state_dict()), "committed blob is stale: run tests/golden/make_flax_blob.py"
  got = checkpoint.load_flax_checkpoint(mk.OUT)
  want = mk.tree()
  assert list(got["params"]) == sorted(want["params"]) and sorted(got) == ["batch_stats", "params"]
  for name, leaf in want["params"].items():
    for k, v in leaf.items():
      if isinstance(v, dict):
        for kk, vv in v.items():
          np.testing.assert_array_equal(got["params"][name][k][kk], vv)
      else:
        np.testing.assert_array_equal(got["params"][name][k], v)
        assert got["params"][name][k].dtype == np.float32
  for name, leaf in want["batch_stats"].items():
    for k, v in leaf.items():
      np.testing.assert_array_equal(got["batch_stats"][name][k], v)
  q1 = got["params"]["QuantConv_1"]
  assert float(q1["DuQ_0"]["a"][0]) != float(q1["DuQ_0"]["c"][0]) > 0      # learnt a != c survive
  assert q1["prune_0"]["mask"].shape == (3, 3, 32, 32) and q1["prune_0"]["mask"].min() == 1.0
  # the msgpack package, a third implementation, reads the same framing
  import msgpack
  raw = msgpack.unpackb(blob, raw=False, strict_map_key=False,
                        ext_hook=lambda code, data: (code, len(data)))
  assert raw["step"] == 1234 and raw["weight_size"][0] == 3
  assert raw["params"]["params"]["QuantDense_1"]["kernel"][0] == 1


def test_workload_tables_reproduce_the_reference_geometry_columns(tmp_path):
  """F4: the geometry columns of examples/sparsity.py:172-230 (literals for DVS128, T = 20,
  128 channels) come out of the parametric rows; values are formatted with str() like there."""
  from snnquantprune_amd import sparsity
  ref_geometry = {                                   # data: the reference's own literals
      "Conv1": "20,2,128,128,128,3,3,1,1", "Conv2": "20,128,128,64,64,3,3,1,1",
      "Conv3": "20,128,128,32,32,3,3,1,1", "Conv4": "20,128,128,16,16,3,3,1,1",
      "TCJA11": "20,20,20,1,256,1,4,1,1", "TCJA12": "128,128,20,1,256,1,4,1,1",
      "Conv5": "20,128,128,8,8,3,3,1,1", "TCJA21": "20,20,20,1,64,1,4,1,1",
      "TCJA22": "128,128,20,1,64,1,4,1,1", "Dense1": "20,2048,512,1,1,1,1,1,1",
      "Dense2": "20,512,110,1,1,1,1,1,1"}
  layers = ["QuantConv_%d" % i for i in range(9)] + ["QuantDense_0", "QuantDense_1"]
  ls = {n: 0.1 + 0.01 * i for i, n in enumerate(layers)}
  probes = ["conv_0", "conv_1", "conv_2", "conv_t_0", "conv_t_1", "conv_tcja1_0", "conv_tcja2_0",
            "conv_tcja1_1", "conv_tcja2_1", "dense1", "dense2"]
  acc = {}
  for i, p in enumerate(probes):
    for io in ("inpt", "out"):
      acc["%s_%s_mean" % (p, io)] = np.array([0.1 * (i + 1), 0.3 * (i + 1)], np.float32)
      acc["%s_%s_min" % (p, io)] = np.array([0.2, 0.5 + 0.01 * i], np.float32)
  mean_lines, min_lines = sparsity.workload_tables(ls, acc, frames=20, channels=128)
  assert mean_lines[0] == "name,weights,inputs,outputs,T,C,M,P,Q,R,S,HS,WS\n" == min_lines[0]
  assert [l.split(",")[0] for l in mean_lines[1:]] == list(ref_geometry)
  for line in mean_lines[1:] + min_lines[1:]:
    name, w, i, o, geom = line.rstrip("\n").split(",", 4)
    assert geom == ref_geometry[name], (name, geom)
  assert mean_lines[1].split(",")[1:4] == [str(0.1), str(np.mean(acc["conv_0_inpt_mean"])),
                                           str(np.mean(acc["conv_0_out_mean"]))]
  assert min_lines[11].split(",")[2] == str(np.max(acc["dense2_inpt_min"]))
  paths = sparsity.write_workload(str(tmp_path / "workload_run"), ls, acc, frames=20, channels=128)
  assert open(paths[0]).readlines() == mean_lines and open(paths[1]).readlines() == min_lines
  # the C3 topology: three conv rows and the read-out
  m3, _ = sparsity.workload_tables(ls, acc, frames=20, channels=128, full=False)
  assert [l.split(",")[0] for l in m3[1:]] == ["Conv1", "Conv2", "Conv3", "Dense1"]
  assert m3[4].rstrip("\n").split(",", 4)[4] == "20,32768,110,1,1,1,1,1,1"


def test_bench_roofline_block_from_recorded_launch_times():
  """bench.py's roofline arithmetic on fixed per-kernel times (no GPU): achieved = algorithmic
  ops / average launch duration of the dominant device function, frac against the peak of the
  instruction it issues, the read-out's HBM ceiling, conv0's stated VALU-issue ceiling."""
  sys.path.insert(0, ROOT)
  import bench
  args = bench.parse([])
  prof = {"conv3x3[128x128x2->128]": (5, 5 * 6.0), "conv3x3[64x64x128->128]": (5, 5 * 5.5),
          "conv3x3[32x32x128->128]": (5, 5 * 1.4), "dense[32768->110]": (5, 5 * 0.13)}
  out = bench.rooflines_of(args, prof, 1024, 20, [4, 4, 4, 4])
  r = out["roofline"]
  assert r["kernel"] == "conv3x3_bits_kernel" and r["launches_per_step"] == 2 and r["bound"] == "mfma"
  ops = (2 * 1024 * 20 * 64 * 64 * 128 * 1152 + 2 * 1024 * 20 * 32 * 32 * 128 * 1152) / 2
  assert abs(r["achieved"] - ops / (3.45e-3) / 1e12) < 1e-6 * r["achieved"]
  assert abs(r["frac"] - r["achieved"] / 10000.0) < 1e-12 and r["unit"] == "TFLOP/s"
  assert r["algorithmic_bytes"] == (1024 * 20 * (65536 + 16384) + 1024 * 20 * (16384 + 4096)) / 2
  d = out["roofline_dense"]
  # 4-bit codes: the read-out runs on the fp6 instruction (ceiling = ops at the 10 POP/s peak)
  assert d["bound"] == "hbm" and 0.65 < d["ceiling_hbm_frac"] < 0.8 and d["frac"] < d["ceiling_hbm_frac"]
  assert "valu_issue" in r and r["valu_issue"]["measured_mix_cycles_per_tile"] > 0
  c0 = [x for x in out["rooflines"] if x["kernel"].startswith("conv3x3[128x128x2")][0]
  # (one wave64 vector instruction per SIMD every 2 cycles: 3.5 per update at 6.0 ms is a third of it)
  assert 0.25 < c0["valu_issue"]["frac"] < 0.4 and c0["valu_issue"]["peak_ginstr_per_s"] == 1228.8
  assert c0["valu_issue"]["frac"] < c0["valu_issue"]["measured_mix_frac"] and "note" in c0
  # 8-bit codes run on the int8 instruction: its peak, not the fp6 one
  out8 = bench.rooflines_of(bench.parse(["--bits", "8"]), prof, 1024, 20, [8, 8, 8, 8])
  assert out8["roofline"]["peak"] == 5000.0
  assert 0.3 < out8["roofline_dense"]["ceiling_hbm_frac"] < 0.4          # int8 read-out
  # mixed precision: the peak is the time-weighted one of the launches
  outm = bench.rooflines_of(bench.parse(["--layer-bits", "2,4,8,4"]), prof, 1024, 20, [2, 4, 8, 4])
  assert 5000.0 < outm["roofline"]["peak"] < 10000.0


def test_bench_stdout_line_stays_parseable_by_the_driver():
  """The driver keeps the last 8 000 bytes of stdout and parses the final line from them: round 5's
  29 KB record (seven legs with their own roofline blocks) left BENCH_r05.parsed null.  The full
  record of that very run, pushed through bench.compact, must come out under 6 000 bytes with
  every key of the contract, and the rest must be reachable through the detail file."""
  sys.path.insert(0, ROOT)
  import bench
  with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
    full = json.load(f)
  assert len(json.dumps(full)) > 20000 and len(full["legs"]) == 7
  full["detail"] = "bench_detail.json"
  short = bench.compact(full)
  text = json.dumps(short)
  assert len(text) < bench.LINE_LIMIT == 6000 and "\n" not in text
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_dense",
            "cpu_baseline", "ranks_seen", "fallbacks", "device_status", "legs", "detail"):
    assert k in short, k
    if k not in ("legs", "config", "roofline", "roofline_dense", "cpu_baseline", "detail"):
      assert short[k] == full[k]
  assert set(short["config"]) == {"workload", "batch_per_gpu", "global_batch", "frames", "parallelism"}
  assert "model" not in short["config"]
  for k in ("kernel", "launches_per_step", "avg_launch_ms", "algorithmic_bytes", "traffic", "bound",
            "achieved", "peak", "unit", "frac"):
    assert short["roofline"][k] == full["roofline"][k]
  assert abs(short["roofline"]["frac"] - short["roofline"]["achieved"] / short["roofline"]["peak"]) < 1e-12
  # where `traffic` comes from is in the line: the committed file (round 5's record) or this run's own counters
  assert short["roofline"]["traffic_measured"] == "committed file"
  live = dict(full, roofline=dict(full["roofline"], traffic_source="measured in this run: rocprofv3 ..."))
  assert bench.compact(live)["roofline"]["traffic_measured"] == "live"
  assert set(short["cpu_baseline"]) == {"value", "unit", "cores", "kind", "cpu", "sample"}
  assert set(short["legs"]) == {"fed", "resident_u8", "captured", "general"} | set(full["legs"])
  for name, leg in short["legs"].items():
    assert set(leg) <= {"value", "ms_per_step", "frac", "bound", "captured_value"} and leg["value"] > 0
  # a dense leg reports the dense layer's HBM fraction, a conv leg the dominant kernel's
  assert short["legs"]["c2_b4096_f32"]["frac"] == full["legs"]["c2_b4096_f32"]["roofline_dense"]["frac"]
  assert short["legs"]["c3_f32"]["frac"] == full["legs"]["c3_f32"]["roofline"]["frac"]
  # emit() writes the full record where --detail says and prints the compact line, last
  import contextlib, io, tempfile
  with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "d.json")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
      bench.emit(bench.parse(["--detail", path]), dict(full))
    assert json.loads(buf.getvalue()) == dict(short, detail=path)
    assert json.load(open(path))["legs"]["c5"] == full["legs"]["c5"]
  # a record that is too long however it is cut still comes out as one parseable line
  bloated = dict(full, legs={"leg%d" % i: full["legs"]["c5"] for i in range(400)})
  buf = io.StringIO()
  with contextlib.redirect_stdout(buf):
    bench.emit(bench.parse(["--detail", os.devnull]), bloated)
  assert len(buf.getvalue()) < 6000 and json.loads(buf.getvalue())["value"] == full["value"]


def test_profile_tools_split_launches_and_count_the_pack_pass(tmp_path):
  """tools/kernel_trace_stats.py splits the launches of one device function by their place in a
  step (conv1 / conv2 share a kernel and a grid), and tools/pmc_summary.py adds the checked pack pass
  to the event layer's bytes only when it runs once per step (a single launch is the bench's own
  preparation of a resident bit-packed batch)."""
  trace = tmp_path / "1_kernel_trace.csv"
  hdr = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name",
         "Correlation_Id", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
         "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z",
         "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]
  rows, t, d = [], 1000, 0
  for step in range(3):
    for name, dur in (("void snnqp::conv3x3_u8c2_kernel<1, true>(snnqp::ConvMfmaArgs)", 4800),
                      ("void snnqp::conv3x3_bits_kernel<0, 128>(snnqp::ConvMfmaArgs)", 5100),
                      ("void snnqp::conv3x3_bits_kernel<0, 128>(snnqp::ConvMfmaArgs)", 1300),
                      ("void at::native::something<float>(int)", 7)):
      d += 1
      rows.append(["KERNEL_DISPATCH", "Agent 2", 1, 0, 1, d, 1, name, d, t, t + dur + step, 0, 0, 64, 0, 16, 256, 1, 1,
                   131072, 1, 1])
      t += dur + 100
  import csv
  with open(trace, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(hdr)
    w.writerows(rows)
  out = tmp_path / "stats.csv"
  p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_trace_stats.py"), str(trace), str(out),
                      "--steps", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
  assert p.returncode == 0, p.stderr.decode()
  got = {(r["Name"], r["Slot"]): r for r in csv.DictReader(open(out))}
  assert set(got) == {("snnqp::conv3x3_u8c2_kernel<1, true>", ""), ("snnqp::conv3x3_bits_kernel<0, 128>", "0"),
                      ("snnqp::conv3x3_bits_kernel<0, 128>", "1")}
  assert float(got[("snnqp::conv3x3_bits_kernel<0, 128>", "0")]["AverageNs"]) == 5101.0
  assert float(got[("snnqp::conv3x3_bits_kernel<0, 128>", "1")]["AverageNs"]) == 1301.0
  assert got[("snnqp::conv3x3_u8c2_kernel<1, true>", "")]["Calls"] == "3"
  # counters: FETCH_SIZE / WRITE_SIZE in KiB per dispatch, two passes in two directories
  for setup_only in (True, False):
    root = tmp_path / ("pmc_%d" % setup_only)
    disp = []
    if setup_only:
      disp.append(("snnqp::pack_ev1_kernel(unsigned char const*)", 1000, 100))          # one launch: preparation
    for step in range(3):
      if not setup_only:
        disp.append(("snnqp::pack_ev1_kernel(unsigned char const*)", 1000, 100))        # once per step
      disp.append(("void snnqp::conv3x3_u8c2_kernel<1, true>(snnqp::ConvMfmaArgs)", 500, 300))
    for counter, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
      os.makedirs(root / counter.lower() / "box")
      with open(root / counter.lower() / "box" / "1_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, dd in enumerate(disp):
          w.writerow([i + 1, dd[0], counter, dd[col]])
    tj = tmp_path / ("traffic_%d.json" % setup_only)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(root), "--steps", "3",
                        "--traffic", str(tj), "--input-format", "u8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()
    conv0 = json.load(open(tj))["bytes_per_launch"]["conv3x3[128x128x2->128]"]
    assert conv0 == (2 * 500 + 300) * 1024 + (0 if setup_only else (2 * 1000 + 100) * 1024)


def test_bench_labels_the_unquantised_dense_net_as_c1():
  sys.path.insert(0, ROOT)
  import bench
  c1 = bench.parse(["--model", "dense", "--batch", "32", "--frames", "10", "--bits", "-1", "--prune", "-1"])
  assert bench.metric_name(c1) == "samples/sec/node (2-layer qdense 2048-512-110, T=10, f32 weights/unpruned)"


# ---------------------------------------------------------------------------
# eval harness (examples/eval.py:53-139): sharding, feed, metric reduction
# ---------------------------------------------------------------------------

_EVAL_WORKER = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from snnquantprune_amd import eval as ev, linen as nn, parallel
from snnquantprune_amd.train_utils import mse_loss

class StandIn:      # stand-in for CextNet: logits a per-sample function of the frames
  dtype = torch.float32
  def init(self, rngs, x, **kw):
    return {"params": {"scale": torch.tensor(1.0)}, "batch_stats": {}}
  def apply(self, variables, x, trgt=None, train=False, online=False, rng=None, mutable=False,
            rngs=None):
    assert not train and mutable == ["batch_stats"]
    if hasattr(x, "to_u8"):
      raise AssertionError("CPU test feeds plain frames")
    f = x.reshape(x.shape[0], -1).to(torch.float32)
    logits = torch.stack([(f[:, k::11].sum(1) * variables["params"]["scale"]) for k in range(11)], 1)
    return (logits / logits.sum(1, keepdim=True).clamp(min=1), None), {}

cfg = nn.ConfigDict()
cfg.seed, cfg.num_frames, cfg.num_classes = 1, 3, 11
cfg.eval_batch_size, cfg.batch_size, cfg.steps_per_eval = %(bs)d, %(bs)d, %(steps)d
cfg.smoothing, cfg.loss_fn, cfg.backend = 0.0, mse_loss, "gloo"
cfg.dataset = %(data)r
state, summary, per_step = ev.evaluate_metrics(cfg, %(workdir)r, model=StandIn(), device="cpu")
rank = int(os.environ.get("RANK", "0"))
if rank == 0:
  print("SUMMARY " + json.dumps({"summary": summary,
                                 "loss": per_step["loss"].tolist(),
                                 "acc": per_step["accuracy"].tolist()}))
if torch.distributed.is_initialized():
  torch.distributed.barrier()
  torch.distributed.destroy_process_group()
"""


def _eval_dataset(path, n=24, T=3, hw=6):
  rng = np.random.Generator(np.random.PCG64(31))
  x = (rng.random((n, T, hw, hw, 2)) < 0.3).astype(np.uint8)
  y = rng.integers(0, 11, n).astype(np.int8)
  np.savez(path, dvs_matrix=x, label=y)
  return x, y


def _eval_reference(x, y, world, bs, steps):
  """What evaluate() must report, restated: process r owns samples [r * n // world, ...)
  (input_pipeline.py:245-254), batches of bs // world, repeated; per-step per-rank mse loss
  and per-sample accuracy, then the mean of each."""
  from oracle import snn_oracle as o
  n = len(y)
  split, per = n // world, bs // world
  nb = split // per
  loss = np.zeros((steps, world)); acc = np.zeros((steps, world, per))
  for r in range(world):
    xs, ys = x[r * split:(r + 1) * split], y[r * split:(r + 1) * split]
    for s in range(steps):
      i = s % nb
      xb, yb = xs[i * per:(i + 1) * per], ys[i * per:(i + 1) * per]
      f = xb.reshape(per, -1).astype(np.float32)
      logits = np.stack([f[:, k::11].sum(1) for k in range(11)], 1)
      logits = logits / np.maximum(logits.sum(1, keepdims=True), 1)
      m = o.compute_metrics(logits.astype(np.float32), yb)
      loss[s, r], acc[s, r] = m["loss"], m["accuracy"]
  return loss, acc


@pytest.mark.parametrize("world", [1, 2])
def test_evaluate_shards_feeds_and_reduces_like_eval_py(tmp_path, world):
  """evaluate() under 1 and 2 gloo processes on CPU: every rank restores / initialises, takes
  its contiguous slice of the split, runs eval_step per batch through the (degenerate, CPU)
  feeder, and the all-gathered per-step losses and per-sample accuracies equal the restated
  reference loop; steps_per_eval = -1 covers the split once, a larger count wraps around."""
  data = str(tmp_path / "frames.npz")
  x, y = _eval_dataset(data)
  for bs, steps in ((8, -1), (4, 7)):
    script = tmp_path / ("worker_%d_%d.py" % (world, bs))
    script.write_text(_EVAL_WORKER % dict(root=ROOT, bs=bs, steps=steps, data=data,
                                          workdir=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29640 + world * 4 + (bs == 4)),
               WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o_ in zip(procs, outs):
      assert p.returncode == 0, o_
    line = [l for l in outs[0].splitlines() if l.startswith("SUMMARY ")][0]
    got = json.loads(line[len("SUMMARY "):])
    nsteps = len(y) // bs if steps == -1 else steps
    eloss, eacc = _eval_reference(x, y, world, bs, nsteps)
    np.testing.assert_allclose(np.asarray(got["loss"]), eloss, rtol=1e-6, atol=1e-7)   # float32 means
    np.testing.assert_array_equal(np.asarray(got["acc"]), eacc)
    assert got["summary"]["steps"] == nsteps and got["summary"]["world"] == world
    assert got["summary"]["samples"] == nsteps * bs
    assert abs(got["summary"]["accuracy"] - eacc.mean()) < 1e-6
    assert abs(got["summary"]["loss"] - eloss.mean()) < 1e-6


def test_evaluate_refuses_an_indivisible_batch(tmp_path):
  from snnquantprune_amd import eval as ev, linen as nn
  cfg = nn.ConfigDict()
  cfg.eval_batch_size = 7
  env_world = os.environ.get("WORLD_SIZE")
  os.environ["WORLD_SIZE"] = "1"
  try:
    cfg.eval_batch_size = 0
    data = str(tmp_path / "d.npz")
    _eval_dataset(data, n=4)
    cfg.dataset, cfg.steps_per_eval, cfg.seed, cfg.num_frames = data, 1, 0, 3
    with pytest.raises(ValueError):
      ev.evaluate_metrics(cfg, str(tmp_path), model=object(), device="cpu")
  finally:
    if env_world is None:
      del os.environ["WORLD_SIZE"]
    else:
      os.environ["WORLD_SIZE"] = env_world
  from snnquantprune_amd import parallel
  with pytest.raises(ValueError, match="divisible"):
    parallel.shard_bounds(10, 0, 4)


def test_latest_checkpoint_picks_the_largest_step(tmp_path):
  from snnquantprune_amd import eval as ev
  assert ev.latest_checkpoint(str(tmp_path)) is None
  for s in (2, 10, 9):
    (tmp_path / ("checkpoint_%d" % s)).write_bytes(b"x")
  (tmp_path / "checkpoint_11.tmp").write_bytes(b"x")
  assert os.path.basename(ev.latest_checkpoint(str(tmp_path))) == "checkpoint_10"


def test_packed_frame_formats_on_the_host():
  """The host packers against the formats restated in tests/helpers.py, shapes, slicing and
  the refusal of values a format cannot hold (no GPU involved)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  from tests.helpers import pack_ev1, pack_ev4
  rng = np.random.Generator(np.random.PCG64(3))
  for H, W in ((128, 128), (13, 17), (5, 3)):
    x = (rng.random((2, 3, H, W, 2)) < 0.2).astype(np.uint8)
    p = ops.pack_frames_host(x, L.EV1)
    assert p.shape == x.shape and p.data.shape == (2, 3, (H * W * 2 + 31) // 32)
    np.testing.assert_array_equal(p.data.numpy().view(np.uint32), pack_ev1(x))
    np.testing.assert_array_equal(p[1].data.numpy().view(np.uint32), pack_ev1(x[1]))
    np.testing.assert_array_equal(p.narrow(0, 1, 1).data.numpy().view(np.uint32), pack_ev1(x[1:2]))
    c = np.minimum(rng.poisson(2.0, (2, 3, H, W, 2)), 15).astype(np.uint8)
    q = ops.pack_frames_host(c, L.EV4)
    assert q.data.shape == (2, 3, H * W)
    np.testing.assert_array_equal(q.data.numpy(), pack_ev4(c))
    with pytest.raises(ValueError):
      ops.pack_frames_host(c + 1 if c.max() == 15 else np.full_like(c, 16), L.EV4)
    with pytest.raises(ValueError):
      ops.pack_frames_host(np.full_like(x, 2), L.EV1)
  assert ops.frame_units(128, 128, L.EV1) * 4 == 4096 and ops.frame_units(128, 128, L.EV4) == 16384


def test_feeder_on_cpu_is_the_plain_iterator():
  from snnquantprune_amd import feed
  batches = [{"dvs_matrix": np.full((2, 3), i, np.uint8), "label": torch.tensor([i, i])} for i in range(5)]
  f = feed.DeviceFeeder(iter(batches), "cpu", 2)
  got = list(f)
  assert len(got) == 5 and f.batches == 5
  for i, b in enumerate(got):
    assert int(np.asarray(b["dvs_matrix"]).max()) == i


def test_conv_dequant_form_query():
  """snnqp_conv_dequant_form (host-side, no device work): which of the three bit-equal
  dequantisation forms the bit-input conv kernels run -- the table addressed by the accumulator
  needs fp6 codes, |acc| <= 2047 and the multi-step LIF / PLIF form with v_reset = 0."""
  import dataclasses
  import torch
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  codes = torch.zeros((3, 3, 4, 32), dtype=torch.int8)
  w = ops.Weight(L.W_I8, codes, 7.0, 0.37, abs_sum_max=700, code_max=7)
  mslif = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
  assert ops.conv_dequant_form(w, mslif) == "table"
  assert ops.conv_dequant_form(dataclasses.replace(w, abs_sum_max=2047), mslif) == "table"
  assert ops.conv_dequant_form(dataclasses.replace(w, abs_sum_max=2048), mslif) == "arith"
  assert ops.conv_dequant_form(dataclasses.replace(w, abs_sum_max=0), mslif) == "arith"      # unknown bound
  assert ops.conv_dequant_form(dataclasses.replace(w, code_max=127, L=127.0), mslif) == "arith"
  assert ops.conv_dequant_form(dataclasses.replace(w, L=1.0), mslif) == "one"
  assert ops.conv_dequant_form(w, ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.25)) == "arith"
  lif = ops.Neuron(L.NEURON_LIF, 2.0, 1.0, 0.0, decay=torch.full((32,), 0.5))
  assert ops.conv_dequant_form(w, lif) == "arith"
  wf = ops.Weight(L.W_F32, torch.zeros((3, 3, 4, 32)), 1.0, 1.0)
  with pytest.raises(L.SnnqpError):
    ops.conv_dequant_form(wf, mslif)


def test_compute_dtype_policy():
  """The reference's shipped configs ask for bfloat16 (examples/tcja/configs/prune_quant_joint.py:71);
  the layers refuse that by default and compute it in float32 when the caller opts in."""
  from snnquantprune_amd import linen as nn
  nn.check_compute_dtype(torch.float32, "QuantDense")
  with pytest.raises(NotImplementedError, match="float32"):
    nn.check_compute_dtype(torch.bfloat16, "QuantDense")
  nn.set_compute_dtype_policy("float32")
  try:
    nn.check_compute_dtype(torch.bfloat16, "QuantDense")      # accepted, computed in float32
  finally:
    nn.set_compute_dtype_policy("strict")
  with pytest.raises(ValueError):
    nn.set_compute_dtype_policy("bf16")


def test_table_slots_balance_the_bank_columns():
  """packing.table_slots (snnqp_weight_t.ch_slots): every wave's 32 channels in 32 different bank
  columns, every slot once, the reported stack = the tallest column; balanced far below the
  default stacking (column = c mod 32) on ranges as skewed as a 90 %-pruned event layer's."""
  import numpy as np
  from snnquantprune_amd import packing
  rng = np.random.Generator(np.random.PCG64(12))
  for cout in (128, 200, 32):
    ranges = (rng.poisson(1.8, cout) * rng.integers(1, 8, cout)).astype(np.int64)
    slots, stack = packing.table_slots(ranges)
    n128 = (cout + 127) // 128 * 128
    assert slots.shape == (n128,) and slots.dtype == np.int32
    r = np.concatenate([ranges, np.zeros(n128 - cout, np.int64)])
    worst = 0
    for g in range(n128 // 128):
      sl = slots[128 * g:128 * g + 128]
      assert sorted(sl.tolist()) == list(range(128))                       # every slot once
      for w in range(4):
        assert sorted((sl[32 * w:32 * w + 32] >> 2).tolist()) == list(range(32))   # a wave: 32 columns
        assert set((sl[32 * w:32 * w + 32] & 3).tolist()) == {w}
      col = np.zeros(32, np.int64)
      np.add.at(col, sl >> 2, r[128 * g:128 * g + 128])
      worst = max(worst, int(col.max()))
      default = r[128 * g:128 * g + 128].reshape(4, 32).sum(0).max()
      assert col.max() <= default
    assert stack == worst
  ranges = (rng.poisson(1.8, 128) * rng.integers(1, 8, 128)).astype(np.int64)
  _, stack = packing.table_slots(ranges)
  assert stack <= 1.35 * ranges.sum() / 32 + ranges.max() / 2 < ranges.reshape(4, 32).sum(0).max()
