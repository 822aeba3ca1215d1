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


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--samples", type=int, default=8)
  ap.add_argument("--perms", type=int, default=8, help="random K permutations under BLAS")
  ap.add_argument("--seq-samples", type=int, default=2, help="C3 samples under the sequential order")
  ap.add_argument("--tree-samples", type=int, default=1, help="C3 samples under the pairwise-tree order")
  ap.add_argument("--out", default=None)
  args = ap.parse_args()
  t0 = time.time()
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
