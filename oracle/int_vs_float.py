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


def c3_report(samples=8, frames=20, hw=128, bits=4, prune=0.9, chunk=4, lam=0.1, seed=4242):
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
  for b0 in range(0, samples, chunk):
    nb = min(chunk, samples - b0)
    x = syn.poisson_spikes((nb, frames, hw, hw, 2), lam, seed=seed + b0)
    # free-running float pass
    rf = o.conv3_dense_forward(x, cq, bns, dq, mode="float", keep=True)
    # int pass, layer by layer, with the float-mode layer forced onto the same input
    xi = np.swapaxes(x, 0, 1)
    for i in range(3):
      ui, si = o.conv_block(xi, cq[i], bns[i], None, "int")
      uf, sf = o.conv_block(xi, cq[i], bns[i], None, "float")
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
    uf, sf = o.dense_block(xf, dq, None, "float")
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
  return out


def c2_report(B=256, T=20, K=2048, hidden=512, nout=110, bits=8, prune=0.5, seed=4343):
  from snnquantprune_amd import synthetic as syn
  from tests.helpers import qweight_of
  v = syn.dense_net_variables(K, hidden, nout, True, prune)
  q1 = qweight_of(o, v["params"]["QuantDense_0"], bits)
  q2 = qweight_of(o, v["params"]["QuantDense_1"], bits)
  x = np.swapaxes(syn.poisson_spikes((B, T, K), 0.1, seed=seed), 0, 1)
  ri = o.dense2_forward(x, q1, q2, mode="int")
  rf = o.dense2_forward(x, q1, q2, mode="float")
  u2f, s2f = o.dense_block(ri["s1"], q2, None, "float")      # layer 2 forced onto int input
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
         "logits_max_abs_diff": float(np.abs(ri["logits"] - rf["logits"]).max())}
  return out


def summarize(c3, c2):
  lay = {**{"C3." + k: v for k, v in c3["layers"].items()},
         **{"C2." + k: v for k, v in c2["layers"].items()}}
  return {"what": "oracle 'int' mode (the kernels' contract, bit-exact on the GPU) against the "
                  "reference-literal float32 mode, CPU, BASELINE sizes; forced = same input raster",
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
  ap.add_argument("--out", default=None)
  args = ap.parse_args()
  t0 = time.time()
  c3 = c3_report(args.samples)
  c2 = c2_report()
  rep = {"summary": summarize(c3, c2), "c3": c3, "c2": c2, "seconds": round(time.time() - t0, 1),
         "generated_by": "python -m oracle.int_vs_float --samples %d" % args.samples}
  txt = json.dumps(rep, indent=1, sort_keys=True)
  print(txt)
  if args.out:
    with open(args.out, "w") as f:
      f.write(txt + "\n")


if __name__ == "__main__":
  main()
