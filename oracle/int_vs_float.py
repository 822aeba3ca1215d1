"""How far is the product's integer contract from the reference's literal float32 arithmetic?

TEST INFRASTRUCTURE (like the rest of oracle/): run by tests/test_oracle_cpu.py and by hand,

  python -m oracle.int_vs_float [--samples 8] [--out profiles/r02_int_vs_float.json]

The HIP kernels are bit-exact against the oracle's 'int' mode: integer codes x spikes summed
exactly, current = fl(fl(acc / L) * m).  The reference itself multiplies float32 spikes with
float32 fake-quantised weights w = fl(fl(q / L) * c) and sums the products in float32 in
whatever order XLA's CPU convolution picks (flax_qconv.py:158-168, flax_qdense.py:87-89,
quant.py:443,467) -- the oracle's 'float' mode, with BLAS summation order standing in for
XLA's.  The two differ by float32 rounding of the contraction only (everything after the
current -- BatchNorm, spiking_learning.py:410-414 -- is the same float32 op sequence), so a
membrane potential can differ by a few ulp of the current and a spike can flip only where the
potential sits within that distance of the threshold.

One summation order with no flips does not bound another, and XLA's order is
unknown.  The forced comparison is therefore repeated under a family of float32 accumulation
orders, installed through snn_oracle.FLOAT_MATMUL: BLAS on the natural K order, BLAS on
`--perms` random permutations of K (different blockings and SIMD-lane groupings of the same
products), one strictly sequential chain (k ascending: what a naive loop or an fma chain does)
and a leaf-1 pairwise tree (the other extreme: the most parallel order).  `summary` reports the
maxima OVER ALL ORDERS; the expensive orders (sequential, tree) run on fewer samples, stated.

Per layer, at BASELINE size (C3: [B, 20, 128, 128, 2], 4-bit, 90 % pruned; C2: 2048 -> 512
-> 110, 8-bit, 50 % pruned, B = 256) this reports
  forced   the layer run in both modes on the SAME input raster (the int-mode one):
           raster flip rate, max |du| and max relative du of the final membrane potentials
           over neurons whose rasters agree ("away from ties")
  free     both modes run end to end, flips propagate: raster disagreement per layer, logits
           agreement, arg-max agreement
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import snn_oracle as o  # noqa: E402

F32 = np.float32


# ---- float32 accumulation orders (a [M, K] float32, w [K, N] float32 -> [M, N] float32) --------

def order_perm(seed):
  """BLAS over a random permutation of K."""
  def f(a, w):
    p = np.random.Generator(np.random.PCG64(seed * 1000003 + a.shape[1])).permutation(a.shape[1])
    return np.ascontiguousarray(a[:, p]) @ np.ascontiguousarray(w[p])
  return f


def order_sequential(a, w, budget=96 << 20):
  """acc = fl(acc + fl(a[:, k] * w[k])), k ascending: one chain per output."""
  M, K = a.shape
  N = w.shape[1]
  out = np.empty((M, N), F32)
  step = max(1, budget // (4 * N))
  for r0 in range(0, M, step):
    ac = a[r0:r0 + step]
    acc = np.zeros((ac.shape[0], N), F32)
    live = np.flatnonzero(ac.any(axis=0))        # x + fl(0 * w) = x: columns of zeros change nothing
    for k in live:
      acc += ac[:, k, None] * w[k][None, :]
    out[r0:r0 + step] = acc
  return out


def order_tree(a, w, budget=192 << 20):
  """Leaf-1 pairwise tree over K (zero-padded to a power of two): (p0 + p1) + (p2 + p3) ..."""
  M, K = a.shape
  N = w.shape[1]
  K2 = 1 << max(0, (K - 1).bit_length())
  out = np.empty((M, N), F32)
  step = max(1, budget // (4 * N * K2))
  for r0 in range(0, M, step):
    ac = a[r0:r0 + step]
    prod = np.zeros((ac.shape[0], K2, N), F32)
    np.multiply(ac[:, :, None], w[None, :, :], out=prod[:, :K])
    while prod.shape[1] > 1:
      prod = prod[:, 0::2] + prod[:, 1::2]
    out[r0:r0 + step] = prod[:, 0]
  return out


def make_orders(perms=8, cheap_samples=16, seq_samples=2, tree_samples=1):
  """name -> (matmul | None for plain BLAS, C3 samples it runs on)."""
  orders = {"blas": (None, cheap_samples)}
  for i in range(perms):
    orders["blas_perm%d" % i] = (order_perm(i + 1), cheap_samples)
  if seq_samples > 0:
    orders["sequential"] = (order_sequential, seq_samples)
  if tree_samples > 0:
    orders["pairwise_tree"] = (order_tree, tree_samples)
  return orders


class _float_order:
  def __init__(self, fn):
    self.fn = fn

  def __enter__(self):
    self.old, o.FLOAT_MATMUL = o.FLOAT_MATMUL, self.fn

  def __exit__(self, *exc):
    o.FLOAT_MATMUL = self.old
    return False


def _u_error(u_int, u_flt, s_int, s_flt):
  """Errors of the final potentials over neurons whose whole rasters agree."""
  same = np.all(s_int == s_flt, axis=0)
  ui, uf = u_int[same].astype(np.float64), u_flt[same].astype(np.float64)
  d = np.abs(ui - uf)
  if d.size == 0:
    return {"neurons": 0}
  scale = np.maximum(np.abs(ui), np.abs(uf))
  big = scale >= 1e-2                       # relative error proper, away from cancellation
  rel = d[big] / scale[big]
  return {"neurons": int(d.size),
          "max_abs": float(d.max()),
          # relative to the threshold scale (v_th = 1): what decides a spike
          "max_rel_to_threshold": float((d / np.maximum(scale, 1.0)).max()),
          "max_rel": float(rel.max()) if rel.size else 0.0,
          "p999_rel": float(np.quantile(rel, 0.999)) if rel.size else 0.0}


def _flip(s_a, s_b):
  n = s_a.size
  k = int(np.count_nonzero(s_a != s_b))
  return {"flips": k, "of": int(n), "rate": k / n if n else 0.0}


def _merge_order(acc, layer, f, ue):
  d = acc.setdefault(layer, {"flips": 0, "neuron_steps": 0, "u_max_abs": 0.0, "u_max_rel": 0.0,
                             "u_max_rel_to_threshold": 0.0, "u_p999_rel": 0.0})
  d["flips"] += f["flips"]
  d["neuron_steps"] += f["of"]
  if ue.get("neurons"):
    for k_src, k_dst in (("max_abs", "u_max_abs"), ("max_rel", "u_max_rel"),
                         ("max_rel_to_threshold", "u_max_rel_to_threshold"), ("p999_rel", "u_p999_rel")):
      d[k_dst] = max(d[k_dst], ue[k_src])


def c3_report(samples=8, frames=20, hw=128, bits=4, prune=0.9, chunk=4, lam=0.1, seed=4242, orders=None):
  from snnquantprune_amd import synthetic as syn
  from tests.helpers import bn_of, qweight_of
  v = syn.conv_net_variables(hw=hw, prune_p=prune) if hw == 128 else \
      syn.conv_net_variables(hw=hw, prune_p=prune, gains=(4.0, 5.0, 6.0, 10.0))
  p = v["params"]
  cq = [qweight_of(o, p["QuantConv_%d" % i], bits) for i in range(3)]
  bns = [bn_of(v, i) for i in range(3)]
  dq = qweight_of(o, p["QuantDense_0"], bits)
  names = ["conv0", "conv1", "conv2", "dense"]
  forced = {n: {"flips": 0, "of": 0, "u": []} for n in names}
  free = {n: {"flips": 0, "of": 0} for n in names}
  logits_equal = argmax_equal = 0
  logit_max_diff = 0.0
  rates = {n: [] for n in names}
  by_order = {}
  for b0 in range(0, samples, chunk):
    nb = min(chunk, samples - b0)
    x = syn.poisson_spikes((nb, frames, hw, hw, 2), lam, seed=seed + b0)
    # free-running float pass
    rf = o.conv3_dense_forward(x, cq, bns, dq, mode="float", keep=True)
    # int pass, layer by layer, with the float-mode layer forced onto the same input
    xi = np.swapaxes(x, 0, 1)
    for i in range(3):
      ui, si = o.conv_block(xi, cq[i], bns[i], None, "int")
      for oname, (ofn, onum) in (orders or {}).items():
        if ofn is None or b0 >= onum:
          continue
        k = min(nb, onum - b0)                    # the first `onum` samples, in time-major layout
        with _float_order(ofn):
          uo, so = o.conv_block(xi[:, :k], cq[i], bns[i], None, "float")
        _merge_order(by_order.setdefault(oname, {}), names[i], _flip(si[:, :k], so), _u_error(ui[:k], uo, si[:, :k], so))
        del uo, so
      uf, sf = o.conv_block(xi, cq[i], bns[i], None, "float")
      _merge_order(by_order.setdefault("blas", {}), names[i], _flip(si, sf), _u_error(ui, uf, si, sf))
      f = _flip(si, sf)
      forced[names[i]]["flips"] += f["flips"]; forced[names[i]]["of"] += f["of"]
      forced[names[i]]["u"].append(_u_error(ui, uf, si, sf))
      g = _flip(si, rf["conv%d_s" % i])
      free[names[i]]["flips"] += g["flips"]; free[names[i]]["of"] += g["of"]
      rates[names[i]].append(float(si.mean()))
      xi = o.max_pool_2x2(si)
      del ui, uf, sf
    xf = o.flatten_channel_major(xi)
    ui, si = o.dense_block(xf, dq, None, "int")
    for oname, (ofn, onum) in (orders or {}).items():
      if ofn is None or b0 >= onum:
        continue
      k = min(nb, onum - b0)
      with _float_order(ofn):
        uo, so = o.dense_block(xf[:, :k], dq, None, "float")
      _merge_order(by_order.setdefault(oname, {}), "dense", _flip(si[:, :k], so), _u_error(ui[:k], uo, si[:, :k], so))
    uf, sf = o.dense_block(xf, dq, None, "float")
    _merge_order(by_order.setdefault("blas", {}), "dense", _flip(si, sf), _u_error(ui, uf, si, sf))
    f = _flip(si, sf)
    forced["dense"]["flips"] += f["flips"]; forced["dense"]["of"] += f["of"]
    forced["dense"]["u"].append(_u_error(ui, uf, si, sf))
    g = _flip(si, rf["dense_s"])
    free["dense"]["flips"] += g["flips"]; free["dense"]["of"] += g["of"]
    rates["dense"].append(float(si.mean()))
    li, lf = o.vote(si), rf["logits"]
    logits_equal += int(np.count_nonzero(np.all(li == lf, axis=1)))
    argmax_equal += int(np.count_nonzero(np.argmax(li, 1) == np.argmax(lf, 1)))
    logit_max_diff = max(logit_max_diff, float(np.abs(li - lf).max()))
    del rf
  out = {"config": "C3: 3x(qconv3x3+BN+LIF+pool)+qdense(%d->110), %dx%dx2, T=%d, %d-bit, %g%% pruned, "
                   "Poisson(%g)>0 spikes" % (dq.kernel.shape[0], hw, hw, frames, bits, prune * 100, lam),
         "samples": samples, "layers": {}}
  for n in names:
    us = [u for u in forced[n]["u"] if u.get("neurons")]
    out["layers"][n] = {
        "firing_rate": float(np.mean(rates[n])),
        "forced_flip_rate": forced[n]["flips"] / forced[n]["of"],
        "forced_flips": forced[n]["flips"], "neuron_steps": forced[n]["of"],
        "free_flip_rate": free[n]["flips"] / free[n]["of"],
        "u_max_abs": max(u["max_abs"] for u in us),
        "u_max_rel_to_threshold": max(u["max_rel_to_threshold"] for u in us),
        "u_max_rel": max(u["max_rel"] for u in us),
        "u_p999_rel": max(u["p999_rel"] for u in us)}
  out["logits_bit_equal"] = "%d/%d" % (logits_equal, samples)
  out["argmax_equal"] = "%d/%d" % (argmax_equal, samples)
  out["logits_max_abs_diff"] = logit_max_diff
  out["forced_by_order"] = by_order
  return out


def c2_report(B=256, T=20, K=2048, hidden=512, nout=110, bits=8, prune=0.5, seed=4343, orders=None):
  from snnquantprune_amd import synthetic as syn
  from tests.helpers import qweight_of
  v = syn.dense_net_variables(K, hidden, nout, True, prune)
  q1 = qweight_of(o, v["params"]["QuantDense_0"], bits)
  q2 = qweight_of(o, v["params"]["QuantDense_1"], bits)
  x = np.swapaxes(syn.poisson_spikes((B, T, K), 0.1, seed=seed), 0, 1)
  ri = o.dense2_forward(x, q1, q2, mode="int")
  rf = o.dense2_forward(x, q1, q2, mode="float")
  u2f, s2f = o.dense_block(ri["s1"], q2, None, "float")      # layer 2 forced onto int input
  by_order = {}
  for oname, (ofn, _) in (orders or {"blas": (None, 0)}).items():
    with _float_order(ofn):
      u1o, s1o = o.dense_block(x, q1, None, "float")
      u2o, s2o = o.dense_block(ri["s1"], q2, None, "float")
    _merge_order(by_order.setdefault(oname, {}), "dense1", _flip(ri["s1"], s1o), _u_error(ri["u1"], u1o, ri["s1"], s1o))
    _merge_order(by_order.setdefault(oname, {}), "dense2", _flip(ri["s2"], s2o), _u_error(ri["u2"], u2o, ri["s2"], s2o))
  out = {"config": "C2: qdense(%d->%d)+LIF -> qdense(%d->%d)+LIF, T=%d, B=%d, %d-bit, %g%% pruned"
                   % (K, hidden, hidden, nout, T, B, bits, prune * 100),
         "layers": {
             "dense1": dict(forced_flip_rate=_flip(ri["s1"], rf["s1"])["rate"],
                            free_flip_rate=_flip(ri["s1"], rf["s1"])["rate"],
                            firing_rate=float(ri["s1"].mean()),
                            **{"u_" + k: v_ for k, v_ in _u_error(ri["u1"], rf["u1"], ri["s1"], rf["s1"]).items()}),
             "dense2": dict(forced_flip_rate=_flip(ri["s2"], s2f)["rate"],
                            free_flip_rate=_flip(ri["s2"], rf["s2"])["rate"],
                            firing_rate=float(ri["s2"].mean()),
                            **{"u_" + k: v_ for k, v_ in _u_error(ri["u2"], u2f, ri["s2"], s2f).items()})},
         "logits_bit_equal": "%d/%d" % (int(np.count_nonzero(np.all(ri["logits"] == rf["logits"], 1))), B),
         "argmax_equal": "%d/%d" % (int(np.count_nonzero(np.argmax(ri["logits"], 1) ==
                                                         np.argmax(rf["logits"], 1))), B),
         "logits_max_abs_diff": float(np.abs(ri["logits"] - rf["logits"]).max()),
         "forced_by_order": by_order}
  return out


def summarize(c3, c2):
  lay = {**{"C3." + k: v for k, v in c3["layers"].items()},
         **{"C2." + k: v for k, v in c2["layers"].items()}}
  ob = {}
  for cfg, rep in (("C3", c3), ("C2", c2)):
    for oname, layers in (rep.get("forced_by_order") or {}).items():
      for lname, d in layers.items():
        ob.setdefault(oname, {})[cfg + "." + lname] = d
  every = [d for layers in ob.values() for d in layers.values()]
  over_orders = {}
  if every:
    over_orders = {
        "orders": sorted(ob),
        "neuron_steps_by_order": {k: int(sum(d["neuron_steps"] for d in v.values())) for k, v in sorted(ob.items())},
        "max_flip_rate_over_orders": max(d["flips"] / max(d["neuron_steps"], 1) for d in every),
        "total_flips_over_orders": int(sum(d["flips"] for d in every)),
        # the two scales, named: error / max(|u|, v_th = 1) -- what decides a spike -- and the pure
        # relative error error / |u| over potentials with |u| >= 0.01
        "max_u_err_rel_to_max_absu_vth_over_orders": max(d["u_max_rel_to_threshold"] for d in every),
        "max_u_err_pure_rel_over_orders": max(d["u_max_rel"] for d in every),
        "p999_u_err_pure_rel_over_orders": max(d["u_p999_rel"] for d in every),
        "max_u_err_abs_over_orders": max(d["u_max_abs"] for d in every),
        "worst_order_by_pure_rel": max(ob, key=lambda k: max(d["u_max_rel"] for d in ob[k].values()))}
  return {"what": "oracle 'int' mode (the kernels' contract, bit-exact on the GPU) against the "
                  "reference-literal float32 mode, CPU, BASELINE sizes; forced = same input raster",
          "north_star_tolerance": "1e-5 relative for membrane potentials: met on the scale "
                                  "max(|u|, v_th) (max_u_err_rel_to_max_absu_vth_*); the PURE relative "
                                  "error |du| / |u| (max_u_err_pure_rel_*) exceeds it on potentials of a "
                                  "few 1e-2 whose absolute error is below 1e-6",
          "over_orders": over_orders,
          "samples_c3": c3["samples"],
          "max_forced_flip_rate": max(v["forced_flip_rate"] for v in lay.values()),
          "max_free_flip_rate": max(v["free_flip_rate"] for v in lay.values()),
          "max_u_rel_to_threshold": max(v["u_max_rel_to_threshold"] for v in lay.values()),
          "max_u_rel": max(v["u_max_rel"] for v in lay.values()),
          "c3_logits_bit_equal": c3["logits_bit_equal"], "c3_argmax_equal": c3["argmax_equal"],
          "c2_logits_bit_equal": c2["logits_bit_equal"], "c2_argmax_equal": c2["argmax_equal"],
          "per_layer_forced_flip_rate": {k: v["forced_flip_rate"] for k, v in lay.items()}}


# ---- the oracle's two free library choices: float32 logistic and reciprocal square root ---------
#
# jax.nn.sigmoid (TCJA gate examples/tcja/models.py:95; PLIF / LIF decay spiking_learning.py:381,432)
# is 1 / (1 + exp(-x)) evaluated in float32 by XLA's CPU expf; flax 0.4.0's BatchNorm multiplies by
# lax.rsqrt(var + eps) (configured examples/tcja/models.py:101-107).  Neither library is here, so
# the oracle picked one evaluation of each (float64 expit rounded once; fl(1 / fl(sqrt(v)))).  The
# families below are the other evaluations a float32 library can plausibly produce, including the
# two adversarial ones -- EVERY value one ulp above / below the correctly rounded one -- which bound
# any implementation that is accurate to an ulp.

def _ulps(x, k):
  x = np.asarray(x, F32)
  to = F32(np.inf) if k > 0 else F32(-np.inf)
  for _ in range(abs(k)):
    x = np.nextafter(x, to)
  return x.astype(F32)


def _expf(x, off=0):
  """Correctly rounded float32 exp (float64 exp rounded once), optionally `off` ulps away."""
  return _ulps(np.exp(np.asarray(x, np.float64)).astype(F32), off)


def _sig_cr(x):
  return (1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(F32)


SIGMOID_CHOICES = {
    "f64_rounded_once": None,                                        # the oracle's choice
    # jax 0.2.27 expit: lax.div(1, lax.add(1, lax.exp(lax.neg(x)))), every op rounded to float32
    "f32_1_over_1p_exp": lambda x: (F32(1) / (F32(1) + _expf(-x))).astype(F32),
    "f32_1_over_1p_exp_expf+1ulp": lambda x: (F32(1) / (F32(1) + _expf(-x, +1))).astype(F32),
    "f32_1_over_1p_exp_expf-1ulp": lambda x: (F32(1) / (F32(1) + _expf(-x, -1))).astype(F32),
    "f32_exp_over_1p_exp": lambda x: (_expf(x) / (F32(1) + _expf(x))).astype(F32),
    "f32_half_tanh": lambda x: (F32(0.5) * np.tanh(F32(0.5) * np.asarray(x, np.float64)).astype(F32)
                                + F32(0.5)).astype(F32),
    "all+1ulp": lambda x: _ulps(_sig_cr(x), +1),
    "all-1ulp": lambda x: _ulps(_sig_cr(x), -1),
}


def _rsqrt_cr(v):
  return (1.0 / np.sqrt(np.asarray(v, np.float64))).astype(F32)


def _rsqrt_newton(v):
  """A 12-bit estimate refined by one float32 Newton step (the shape of a vectorised rsqrt)."""
  v = np.asarray(v, F32)
  y = _rsqrt_cr(v)
  y = (y.view(np.uint32) & np.uint32(0xFFFFF000)).view(F32)        # keep 11 mantissa bits
  return (y * (F32(1.5) - (F32(0.5) * v) * y * y)).astype(F32)


RSQRT_CHOICES = {
    "1_over_sqrt": None,                                             # the oracle's choice
    "rsqrt_correctly_rounded": _rsqrt_cr,
    "rsqrt_estimate_newton": _rsqrt_newton,
    "all+1ulp": lambda v: _ulps(_rsqrt_cr(v), +1),
    "all-1ulp": lambda v: _ulps(_rsqrt_cr(v), -1),
}


class _choice:
  def __init__(self, sigmoid=None, rsqrt=None, bn_folded=False, fma=False):
    self.new = (sigmoid, rsqrt, bn_folded, fma)

  def __enter__(self):
    self.old = (o.SIGMOID, o.RSQRT, o.BN_FOLDED, o.FMA_CONTRACT)
    o.SIGMOID, o.RSQRT, o.BN_FOLDED, o.FMA_CONTRACT = self.new

  def __exit__(self, *exc):
    o.SIGMOID, o.RSQRT, o.BN_FOLDED, o.FMA_CONTRACT = self.old
    return False


def _ulp_distance(a, b):
  """Largest distance in float32 ulps between two positive float32 arrays."""
  a = np.asarray(a, F32).view(np.int32).astype(np.int64)
  b = np.asarray(b, F32).view(np.int32).astype(np.int64)
  return int(np.abs(a - b).max()) if a.size else 0


_CEXT_RASTERS = ("pool0", "pool1", "pool2", "conv_t_0", "conv_t_1", "dense1_s", "dense2_s")


def _cextnet_run(v, x, bits, neuron_cfg=None):
  from tests.helpers import bn_of, qweight_of
  p = v["params"]
  return o.cextnet_forward(
      x, [qweight_of(o, p["QuantConv_%d" % i], bits) for i in (0, 1, 2, 3, 6)],
      [bn_of(v, i) for i in range(5)],
      [(qweight_of(o, p["QuantConv_4"], bits), qweight_of(o, p["QuantConv_5"], bits)),
       (qweight_of(o, p["QuantConv_7"], bits), qweight_of(o, p["QuantConv_8"], bits))],
      [qweight_of(o, p["QuantDense_0"], bits), qweight_of(o, p["QuantDense_1"], bits)], neuron_cfg)


def _against(base, other, acc):
  for name in _CEXT_RASTERS:
    d = acc["raster_flips"].setdefault(name, {"flips": 0, "neuron_steps": 0})
    d["flips"] += int(np.count_nonzero(base[name] != other[name]))
    d["neuron_steps"] += int(base[name].size)
  for g in ("gate0", "gate1"):
    acc["gate_max_ulps"][g] = max(acc["gate_max_ulps"].get(g, 0), _ulp_distance(base[g], other[g]))
    acc["gate_values_changed"][g] = acc["gate_values_changed"].get(g, 0) + \
        int(np.count_nonzero(base[g] != other[g]))
    acc["gate_values"][g] = acc["gate_values"].get(g, 0) + int(base[g].size)
  eq = np.all(base["logits"] == other["logits"], axis=1)
  acc["logits_bit_equal"] += int(np.count_nonzero(eq))
  acc["argmax_equal"] += int(np.count_nonzero(np.argmax(base["logits"], 1) == np.argmax(other["logits"], 1)))
  acc["samples"] += int(eq.size)
  acc["logits_max_abs_diff"] = max(acc["logits_max_abs_diff"], float(np.abs(base["logits"] - other["logits"]).max()))


def _new_acc():
  return {"raster_flips": {}, "gate_max_ulps": {}, "gate_values_changed": {}, "gate_values": {},
          "logits_bit_equal": 0, "argmax_equal": 0, "samples": 0, "logits_max_abs_diff": 0.0}


def cextnet_choices(samples=8, frames=20, hw=128, bits=4, prune=0.9, chunk=2, lam=0.1, seed=991,
                    neuron_cfg=None, sigmoids=None, rsqrts=None, with_fma=False):
  """Full CextNet (5 conv blocks + 2 TCJA gates + 2 dense, random BatchNorm statistics), run end to
  end under every alternative evaluation of the logistic (gates; decay when `neuron_cfg` names a
  PLIF / LIF neuron) and of BatchNorm's reciprocal square root, each against the oracle's own
  choice on the same inputs: raster flips per layer (free-running: a flip propagates), how far
  the gates move, and whether the logits stay bit-equal."""
  from snnquantprune_amd import synthetic as syn
  v = syn.cextnet_variables(frames=frames, hw=hw, prune_p=prune, random_bn=True)
  sig = SIGMOID_CHOICES if sigmoids is None else sigmoids
  rsq = RSQRT_CHOICES if rsqrts is None else rsqrts
  runs = [("sigmoid", k, f, None, False) for k, f in sig.items() if f is not None] + \
         [("rsqrt", k, None, f, False) for k, f in rsq.items() if f is not None]
  if rsqrts is None:
    # the OTHER operation order of BatchNorm (mean folded into the bias), with either rsqrt
    runs += [("bn_order", "x*mul+(bias-mean*mul)", None, None, True),
             ("bn_order", "x*mul+(bias-mean*mul), rsqrt correctly rounded", None, _rsqrt_cr, True)]
    # multiply-add pairs contracted into fused multiply-adds (BatchNorm; the PLIF update when the
    # neuron is one), alone and on top of the other order / the other rsqrt / jax's logistic
    runs += [("fma", "contracted", None, None, False, True),
             ("fma", "contracted, mean folded into the bias", None, None, True, True),
             ("fma", "contracted, rsqrt correctly rounded, float32 logistic", SIGMOID_CHOICES["f32_1_over_1p_exp"],
              _rsqrt_cr, False, True)]
  if rsqrts is not None and with_fma:
    runs += [("fma", "contracted", None, None, False, True)]
  runs = [r if len(r) == 6 else r + (False,) for r in runs]
  accs = {(r[0], r[1]): _new_acc() for r in runs}
  rates = {n: [] for n in _CEXT_RASTERS}
  for b0 in range(0, samples, chunk):
    nb = min(chunk, samples - b0)
    x = syn.poisson_spikes((nb, frames, hw, hw, 2), lam, seed=seed + b0)
    base = _cextnet_run(v, x, bits, neuron_cfg)
    for n in _CEXT_RASTERS:
      rates[n].append(float(np.mean(base[n])))
    for kind, name, fs, fr, folded, fma in runs:
      with _choice(fs, fr, folded, fma):
        other = _cextnet_run(v, x, bits, neuron_cfg)
      _against(base, other, accs[(kind, name)])
  # how many of the 5 x C BatchNorm multipliers each rsqrt evaluation changes, and by how much
  muls = {}
  for name, f in rsq.items():
    with _choice(None, f):
      muls[name] = np.concatenate([o.bn_coeffs(**{k: a for k, a in _bn(v, i).items()})[1] for i in range(5)])
  base_name = [k for k, f in rsq.items() if f is None][0]
  out = {"config": "CextNet: 5x qconv3x3 + 2 TCJA gates + qdense(2048->512->110), %dx%dx2, T=%d, %d-bit, %g%% "
                   "pruned, random BatchNorm statistics, Poisson(%g)>0 spikes%s"
                   % (hw, hw, frames, bits, prune * 100, lam,
                      "" if not neuron_cfg else ", neuron %s" % neuron_cfg.get("kind")),
         "samples": samples, "firing_rate": {n: float(np.mean(r)) for n, r in rates.items()},
         "sigmoid": {}, "rsqrt": {}, "bn_order": {}, "fma": {},
         "bn_multipliers": {name: {"changed": int(np.count_nonzero(m != muls[base_name])), "of": int(m.size),
                                   "max_ulps": _ulp_distance(m / np.sign(m), muls[base_name] / np.sign(muls[base_name]))}
                            for name, m in muls.items() if name != base_name}}
  for (kind, name), a in accs.items():
    for d in a["raster_flips"].values():
      d["rate"] = d["flips"] / max(d["neuron_steps"], 1)
    a["logits_bit_equal"] = "%d/%d" % (a["logits_bit_equal"], a["samples"])
    a["argmax_equal"] = "%d/%d" % (a["argmax_equal"], a["samples"])
    out[kind][name] = a
  return out


def _bn(v, i):
  from tests.helpers import bn_of
  return bn_of(v, i)


def decay_choices(B=256, T=20, K=2048, hidden=512, nout=110, bits=8, prune=0.5, seed=4343):
  """PLIF (one learnt scalar, spiking_learning.py:381) and LIF (one per neuron, :432) decays
  k = sigmoid(tau) under every logistic, on the C2 head at full size: how many decays change,
  the rasters' flips and the logits."""
  from snnquantprune_amd import synthetic as syn
  from tests.helpers import qweight_of
  v = syn.dense_net_variables(K, hidden, nout, True, prune)
  q1 = qweight_of(o, v["params"]["QuantDense_0"], bits)
  q2 = qweight_of(o, v["params"]["QuantDense_1"], bits)
  x = np.swapaxes(syn.poisson_spikes((B, T, K), 0.1, seed=seed), 0, 1)
  rng = np.random.Generator(np.random.PCG64(seed + 1))
  taus = {"parametric_leaky_IF": [{"kind": "parametric_leaky_IF", "tau_param": np.array([t], F32)}
                                  for t in (0.3, -0.7, 1.1, 2.3)],
          "LIF": [None]}
  out = {"config": "C2 head: qdense(%d->%d) -> qdense(%d->%d), T=%d, B=%d, %d-bit, %g%% pruned; PLIF with "
                   "tau_param in {0.3, -0.7, 1.1, 2.3}; LIF with tau ~ N(0, 1) per neuron"
                   % (K, hidden, hidden, nout, T, B, bits, prune * 100), "kinds": {}}
  tv1, tv2 = rng.standard_normal(hidden).astype(F32), rng.standard_normal(nout).astype(F32)

  def run(cfgs):
    u1, s1 = o.dense_block(x, q1, cfgs[0], "int")
    u2, s2 = o.dense_block(s1, q2, cfgs[1], "int")
    return s1, s2, o.vote(s2)
  for kind, plist in taus.items():
    res = {}
    for name, f in list(SIGMOID_CHOICES.items()) + [("fma_contracted", "fma")]:
      if f is None:
        continue
      fma = f == "fma"
      f = None if fma else f
      a = {"decays_changed": 0, "decays": 0, "flips": {"dense1": 0, "dense2": 0},
           "neuron_steps": {"dense1": 0, "dense2": 0}, "logits_bit_equal": 0, "samples": 0}
      for pc in plist:
        cfgs = (pc, pc) if pc is not None else ({"kind": "LIF", "tau_vec": tv1}, {"kind": "LIF", "tau_vec": tv2})
        tau_all = np.concatenate([np.ravel(c.get("tau_param", c.get("tau_vec"))) for c in
                                  (cfgs if pc is None else cfgs[:1])])
        base = run(cfgs)
        k0 = o.sigmoid_f32(tau_all)
        with _choice(f, None, False, fma):
          other = run(cfgs)
          k1 = o.sigmoid_f32(tau_all)
        a["decays_changed"] += int(np.count_nonzero(k0 != k1)); a["decays"] += int(k0.size)
        for li, ln in ((0, "dense1"), (1, "dense2")):
          a["flips"][ln] += int(np.count_nonzero(base[li] != other[li]))
          a["neuron_steps"][ln] += int(base[li].size)
        a["logits_bit_equal"] += int(np.count_nonzero(np.all(base[2] == other[2], 1)))
        a["samples"] += int(base[2].shape[0])
      a["logits_bit_equal"] = "%d/%d" % (a["logits_bit_equal"], a["samples"])
      res[name] = a
    out["kinds"][kind] = res
  return out


def summarize_choices(cext, decay, cext_plif=None):
  def worst(block):
    w = {}
    for name, a in block.items():
      for layer, d in a["raster_flips"].items():
        if d["rate"] >= w.get(layer, (-1.0, ""))[0]:
          w[layer] = (d["rate"], name)
    return {layer: {"max_flip_rate": r, "under": n} for layer, (r, n) in w.items()}
  plausible = lambda block: {k: a for k, a in block.items() if not k.startswith("all")}
  s = {"what": "the oracle's own evaluation of the float32 logistic (float64 expit rounded once) and of "
               "BatchNorm's reciprocal square root (fl(1 / fl(sqrt(v)))) against the other evaluations a "
               "float32 library can produce, full CextNet geometry end to end, CPU.  'all+-1ulp' move "
               "EVERY value one ulp: a bound for any implementation accurate to one ulp, not a candidate",
       "samples": cext["samples"],
       "sigmoid_plausible_total_flips": int(sum(d["flips"] for a in plausible(cext["sigmoid"]).values()
                                                for d in a["raster_flips"].values())),
       "rsqrt_plausible_total_flips": int(sum(d["flips"] for a in plausible(cext["rsqrt"]).values()
                                              for d in a["raster_flips"].values())),
       "sigmoid_worst_per_layer": worst(cext["sigmoid"]), "rsqrt_worst_per_layer": worst(cext["rsqrt"]),
       "sigmoid_logits_bit_equal": {k: a["logits_bit_equal"] for k, a in cext["sigmoid"].items()},
       "rsqrt_logits_bit_equal": {k: a["logits_bit_equal"] for k, a in cext["rsqrt"].items()},
       "sigmoid_argmax_equal": {k: a["argmax_equal"] for k, a in cext["sigmoid"].items()},
       "rsqrt_argmax_equal": {k: a["argmax_equal"] for k, a in cext["rsqrt"].items()},
       "gate_max_ulps": {k: a["gate_max_ulps"] for k, a in cext["sigmoid"].items()},
       "bn_order": {k: {"flips": {l: d["flips"] for l, d in a["raster_flips"].items()},
                        "logits_bit_equal": a["logits_bit_equal"], "argmax_equal": a["argmax_equal"]}
                    for k, a in cext.get("bn_order", {}).items()},
       "fma": {k: {"flips": {l: d["flips"] for l, d in a["raster_flips"].items()},
                   "logits_bit_equal": a["logits_bit_equal"], "argmax_equal": a["argmax_equal"]}
               for k, a in cext.get("fma", {}).items()},
       "bn_multipliers": cext["bn_multipliers"],
       "decay": {kind: {k: {"decays_changed": "%d/%d" % (a["decays_changed"], a["decays"]),
                            "flips": a["flips"], "logits_bit_equal": a["logits_bit_equal"]}
                        for k, a in res.items()} for kind, res in decay["kinds"].items()}}
  if cext_plif is not None:
    both = dict(cext_plif["sigmoid"], **{"fma: " + k: a for k, a in cext_plif.get("fma", {}).items()})
    s["cextnet_plif_sigmoid_logits_bit_equal"] = {k: a["logits_bit_equal"] for k, a in both.items()}
    s["cextnet_plif_total_flips"] = {k: int(sum(d["flips"] for d in a["raster_flips"].values()))
                                     for k, a in both.items()}
  return s


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--choices", action="store_true",
                  help="instead of int-vs-float: the oracle's logistic / rsqrt choices against the "
                       "alternatives (profiles/r06_oracle_choices.json)")
  ap.add_argument("--plif-samples", type=int, default=4)
  ap.add_argument("--samples", type=int, default=8)
  ap.add_argument("--perms", type=int, default=8, help="random K permutations under BLAS")
  ap.add_argument("--seq-samples", type=int, default=2, help="C3 samples under the sequential order")
  ap.add_argument("--tree-samples", type=int, default=1, help="C3 samples under the pairwise-tree order")
  ap.add_argument("--out", default=None)
  args = ap.parse_args()
  t0 = time.time()
  if args.choices:
    cext = cextnet_choices(args.samples)
    decay = decay_choices()
    plif = cextnet_choices(args.plif_samples, neuron_cfg={"kind": "parametric_leaky_IF",
                                                          "tau_param": np.array([0.3], F32)},
                           rsqrts={"1_over_sqrt": None}, with_fma=True) if args.plif_samples > 0 else None
    rep = {"summary": summarize_choices(cext, decay, plif), "cextnet": cext, "decay": decay,
           "cextnet_plif": plif, "seconds": round(time.time() - t0, 1),
           "generated_by": "python -m oracle.int_vs_float --choices --samples %d --plif-samples %d"
                           % (args.samples, args.plif_samples)}
    txt = json.dumps(rep, indent=1, sort_keys=True)
    print(txt)
    if args.out:
      with open(args.out, "w") as f:
        f.write(txt + "\n")
    return
  orders = make_orders(args.perms, args.samples, args.seq_samples, args.tree_samples)
  c3 = c3_report(args.samples, orders=orders)
  c2 = c2_report(orders=orders)
  rep = {"summary": summarize(c3, c2), "c3": c3, "c2": c2, "seconds": round(time.time() - t0, 1),
         "generated_by": "python -m oracle.int_vs_float --samples %d --perms %d --seq-samples %d --tree-samples %d"
                         % (args.samples, args.perms, args.seq_samples, args.tree_samples)}
  txt = json.dumps(rep, indent=1, sort_keys=True)
  print(txt)
  if args.out:
    with open(args.out, "w") as f:
      f.write(txt + "\n")


if __name__ == "__main__":
  main()
