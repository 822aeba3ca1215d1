"""Deterministic parity cases: inputs (from seeds) and the oracle's expected
outputs.  tests/golden/make_golden.py stores `expected(...)` as .npz fixtures;
the CPU suite checks the oracle still reproduces them; the GPU suite checks the
HIP path against both the live oracle and the committed fixtures.
"""
import numpy as np

from snnquantprune_amd import synthetic as syn
from tests.helpers import bn_of, packbits_lastaxis, qweight_of

F32 = np.float32

# The nine geometries of the reference's flax_qconv_test.py:148-285
# (name, H, W, kernel, strides, padding, expected OH, OW after ONE conv).
REF_CONV_GEOMS = [
    ("base_case", 28, 28, (2, 2), (1, 1), "SAME", 28, 28),
    ("1x1_input", 1, 1, (2, 2), (1, 1), "SAME", 1, 1),
    ("13x17_input", 13, 17, (2, 2), (1, 1), "SAME", 13, 17),
    ("1x1_kernel", 28, 28, (1, 1), (1, 1), "SAME", 28, 28),
    ("3x7_kernel", 28, 28, (3, 7), (1, 1), "SAME", 28, 28),
    ("valid_padding", 28, 28, (2, 2), (1, 1), "VALID", 27, 27),
    ("3715_padding", 28, 28, (2, 2), (1, 1), ((3, 7), (1, 5)), 37, 33),
    ("2_2_stride", 28, 28, (2, 2), (2, 2), "SAME", 14, 14),
    ("3_7_stride", 28, 28, (2, 2), (3, 7), "SAME", 10, 4),
]
# The reference applies two such convs in sequence; its table lists the size
# after both (e.g. VALID: 28 -> 27 -> 26; pad ((3,7),(1,5)): 28 -> 37 -> 46 and
# 28 -> 33 -> 38; stride (3,7): 28 -> 10 -> 4 and 28 -> 4 -> 1).
REF_CONV_TWICE = {"base_case": (28, 28), "1x1_input": (1, 1), "13x17_input": (13, 17),
                  "1x1_kernel": (28, 28), "3x7_kernel": (28, 28),
                  "valid_padding": (26, 26), "3715_padding": (46, 38),
                  "2_2_stride": (7, 7), "3_7_stride": (4, 1)}


def _rng(seed):
  return np.random.Generator(np.random.PCG64(seed))


# ---------------------------------------------------------------------------
# quantiser known answers
# ---------------------------------------------------------------------------


def quant_case():
  r = _rng(11)
  w = (r.standard_normal((48, 40)) * 0.3).astype(F32)
  # exact ties of x * (n_lv - 1) for a = 1: 0.5/7, 1.5/7, 2.5/7 round to even
  w[0, :6] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5], F32) / F32(7)
  w[1, :4] = np.array([5.0, -5.0, 1.0, -1.0], F32)        # clipping
  mask = (r.random(w.shape) > 0.5).astype(F32)
  return {"w": w, "mask": mask}


def quant_expected(o):
  c = quant_case()
  w, mask = c["w"], c["mask"]
  out = {}
  for bits in (2, 3, 4, 8):
    for a, cc in ((1.0, 1.0), (0.73, 0.41)):
      key = "duq_b%d_a%g" % (bits, a)
      out[key + "_codes"] = o.duq_codes(w, a, bits).astype(np.int8)
      out[key + "_fq"] = o.duq_forward(w, a, cc, bits)
      out[key + "_fq_masked"] = o.prune_forward(o.duq_forward(w, a, cc, bits), mask)
    out["us_b%d" % bits] = o.uniform_static_forward(w, 0.9, bits)
    out["pd_b%d" % bits] = o.parametric_d_forward(w, 0.05, bits)
    out["pdx_b%d" % bits] = o.parametric_d_xmax_forward(w, 2 ** -4, 0.8)
  return out


# ---------------------------------------------------------------------------
# dense block, integer path (ragged K and N)
# ---------------------------------------------------------------------------


def dense_block_case(T=6, B=5, K=200, N=70, bits=8, p=0.5, counts=False):
  leaf = syn.quant_leaf((K, N), 4.0, 901, True, p)
  if counts:
    x = syn.poisson_counts((T, B, K), 0.15, seed=902)
  else:
    x = syn.poisson_spikes((T, B, K), 0.1, seed=902)
  u0 = (_rng(903).random((B, N)) * 0.5).astype(F32)
  return {"leaf": leaf, "x": x, "u0": u0, "bits": bits}


def dense_block_expected(o, c):
  qw = qweight_of(o, c["leaf"], c["bits"])
  acc = o.dense_acc(c["x"], qw)
  u, s = o.dense_block(c["x"], qw, None, "int", u0=c["u0"])
  return {"acc": acc.astype(np.int32), "u": u, "s": s.astype(np.uint8)}


# ---------------------------------------------------------------------------
# dense block, float (fseq) path: unquantised weights, real-valued input
# ---------------------------------------------------------------------------


def dense_fseq_case(T=4, B=3, K=96, N=40):
  leaf = syn.quant_leaf((K, N), 3.0, 911, False, -1)
  x = (_rng(912).random((T, B, K)) < 0.2) * _rng(913).random((T, B, K))
  return {"leaf": leaf, "x": x.astype(F32)}


def dense_fseq_expected(o, c):
  qw = qweight_of(o, c["leaf"], 8, quantized=False)
  y = o.quant_dense(c["x"], qw, "fseq")
  u, s = o.dense_block(c["x"], qw, None, "fseq")
  return {"y": y, "u": u, "s": s.astype(np.uint8)}


# ---------------------------------------------------------------------------
# conv blocks (the MFMA shapes, small)
# ---------------------------------------------------------------------------


def conv_block_case(T=3, B=2, hw=8, cin=128, cout=128, bits=4, p=0.9,
                    random_bn=True, seed=921, lam=0.15, gain=5.0):
  leaf = syn.quant_leaf((3, 3, cin, cout), gain, seed, True, p)
  bp, bs = syn.bn_leaf(cout, random_bn, seed + 1)
  if cin == 2:
    x = syn.poisson_counts((T, B, hw, hw, cin), 0.3, seed=seed + 2)
  else:
    x = syn.poisson_spikes((T, B, hw, hw, cin), lam, seed=seed + 2)
  return {"leaf": leaf, "bn": dict(mean=bs["mean"], var=bs["var"], scale=bp["scale"],
                                   bias=bp["bias"]),
          "x": x, "bits": bits}


def conv_block_expected(o, c):
  qw = qweight_of(o, c["leaf"], c["bits"])
  u, s = o.conv_block(c["x"], qw, c["bn"], None, "int")
  pooled = o.max_pool_2x2(s)
  return {"u": u, "s_bits": packbits_lastaxis(s), "pooled_bits": packbits_lastaxis(pooled),
          "rate": np.array([s.mean()], F32)}


# ---------------------------------------------------------------------------
# reference conv geometries, integer-valued data (exact in any order)
# ---------------------------------------------------------------------------


def conv_geom_case(name):
  g = [t for t in REF_CONV_GEOMS if t[0] == name][0]
  _, H, W, ks, st, pad, _, _ = g
  r = _rng(17 + len(name) * 7 + H)
  x = r.integers(-3, 4, size=(2, H, W, 1)).astype(F32)
  k1 = r.integers(-2, 3, size=ks + (1, 10)).astype(F32)
  k2 = r.integers(-2, 3, size=ks + (10, 20)).astype(F32)
  return {"x": x, "k1": k1, "k2": k2, "strides": st, "padding": pad}


def conv_geom_expected(o, c):
  y1 = o.quant_conv(c["x"], o.QWeight(c["k1"]), c["strides"], c["padding"], mode="fseq")
  y2 = o.quant_conv(y1, o.QWeight(c["k2"]), c["strides"], c["padding"], mode="fseq")
  return {"y1": y1, "y2": y2}


# ---------------------------------------------------------------------------
# neuron variants
# ---------------------------------------------------------------------------


def neuron_case(T=12, R=7, C=40):
  x = (_rng(931).standard_normal((T, R, C)) * 0.9 + 0.3).astype(F32)
  tau_vec = (_rng(932).standard_normal(C) * 0.8).astype(F32)
  u0 = (_rng(933).random((R, C)) * 0.7).astype(F32)
  return {"x": x, "tau_vec": tau_vec, "u0": u0, "tau_param": F32(-np.log(3.0 - 1))}


def neuron_expected(o, c):
  out = {}
  x = c["x"]
  for name, fn in (
      ("mslif_tau2", lambda u, v: o.multi_step_lif(u, v, 2.0)),
      ("mslif_tau3_vr", lambda u, v: o.multi_step_lif(u, v, 3.0, 0.8, 0.1)),
      ("plif", lambda u, v: o.parametric_leaky_if(u, v, c["tau_param"])),
      ("lif", lambda u, v: o.lif(u, v, c["tau_vec"])),
  ):
    u = c["u0"].copy()
    ss = []
    for t in range(x.shape[0]):
      u, s = fn(u, x[t])
      ss.append(s)
    out[name + "_u"] = u
    out[name + "_s"] = np.stack(ss).astype(np.uint8)
  return out


# ---------------------------------------------------------------------------
# whole models
# ---------------------------------------------------------------------------


def dense_net_case(quantized, T=6, B=4, K=256, hidden=96, out=110):
  v = syn.dense_net_variables(K, hidden, out, quantized, 0.5 if quantized else -1.0)
  x = syn.poisson_spikes((B, T, K), 0.1, seed=941)
  return {"vars": v, "x": x, "bits": 8, "quantized": quantized}


def dense_net_expected(o, c):
  p = c["vars"]["params"]
  q = c["quantized"]
  r = o.dense2_forward(np.swapaxes(c["x"], 0, 1),
                       qweight_of(o, p["QuantDense_0"], c["bits"], q),
                       qweight_of(o, p["QuantDense_1"], c["bits"], q),
                       mode="int" if q else "fseq")
  return {"s1": r["s1"].astype(np.uint8), "s2": r["s2"].astype(np.uint8),
          "logits": r["logits"]}


def conv_net_case(T=4, B=2, hw=16, bits=4, p=0.9, random_bn=True, counts=False,
                  gains=(4.0, 5.0, 6.0, 10.0), layer_bits=None, out=110):
  # gains for the tiny 16x16 topology (fan-in of the read-out is only 512)
  v = syn.conv_net_variables(hw=hw, prune_p=p, random_bn=random_bn, gains=gains, out=out)
  if counts:
    x = syn.poisson_counts((B, T, hw, hw, 2), 0.2, seed=951)
  else:
    x = syn.poisson_spikes((B, T, hw, hw, 2), 0.1, seed=951)
  return {"vars": v, "x": x, "bits": bits, "layer_bits": layer_bits}


def conv_net_expected(o, c):
  p = c["vars"]["params"]
  lb = c.get("layer_bits") or [c["bits"]] * 4
  r = o.conv3_dense_forward(
      c["x"], [qweight_of(o, p["QuantConv_%d" % i], lb[i]) for i in range(3)],
      [bn_of(c["vars"], i) for i in range(3)],
      qweight_of(o, p["QuantDense_0"], lb[3]), mode="int")
  out = {"pool%d_bits" % i: packbits_lastaxis(r["pool%d" % i]) for i in range(3)}
  out["dense_s"] = r["dense_s"].astype(np.uint8)
  out["logits"] = r["logits"]
  out["rates"] = np.array([r["pool%d" % i].mean() for i in range(3)] +
                          [r["dense_s"].mean()], F32)
  return out


def cextnet_case(T=4, B=2, hw=64, bits=4, p=0.9):
  v = syn.cextnet_variables(frames=T, hw=hw, prune_p=p, random_bn=True)
  x = syn.poisson_spikes((B, T, hw, hw, 2), 0.1, seed=991)
  return {"vars": v, "x": x, "bits": bits}


def cextnet_expected(o, c):
  p, b = c["vars"]["params"], c["bits"]
  r = o.cextnet_forward(
      c["x"], [qweight_of(o, p["QuantConv_%d" % i], b) for i in (0, 1, 2, 3, 6)],
      [bn_of(c["vars"], i) for i in range(5)],
      [(qweight_of(o, p["QuantConv_4"], b), qweight_of(o, p["QuantConv_5"], b)),
       (qweight_of(o, p["QuantConv_7"], b), qweight_of(o, p["QuantConv_8"], b))],
      [qweight_of(o, p["QuantDense_0"], b), qweight_of(o, p["QuantDense_1"], b)])
  out = {"pool%d_bits" % i: packbits_lastaxis(r["pool%d" % i]) for i in range(3)}
  for i in range(2):
    out["conv_t_%d_bits" % i] = packbits_lastaxis(r["conv_t_%d" % i])
    out["gate%d" % i] = r["gate%d" % i]
  out["dense1_s"] = r["dense1_s"].astype(np.uint8)
  out["dense2_s"] = r["dense2_s"].astype(np.uint8)
  out["logits"] = r["logits"]
  return out


GOLDEN = {
    "quant": lambda o: quant_expected(o),
    "dense_block": lambda o: dense_block_expected(o, dense_block_case()),
    "dense_block_counts": lambda o: dense_block_expected(o, dense_block_case(counts=True)),
    "dense_fseq": lambda o: dense_fseq_expected(o, dense_fseq_case()),
    "conv_block_c128": lambda o: conv_block_expected(o, conv_block_case()),
    "conv_block_c2": lambda o: conv_block_expected(
        o, conv_block_case(hw=16, cin=2, seed=961, gain=4.0)),
    "neurons": lambda o: neuron_expected(o, neuron_case()),
    "dense_net_c1": lambda o: dense_net_expected(o, dense_net_case(False)),
    "dense_net_c2": lambda o: dense_net_expected(o, dense_net_case(True)),
    "conv_net_c3_tiny": lambda o: conv_net_expected(o, conv_net_case()),
    "cextnet_tiny": lambda o: cextnet_expected(o, cextnet_case()),
}
for _name in [g[0] for g in REF_CONV_GEOMS]:
  GOLDEN["conv_geom_" + _name] = (
      lambda o, _n=_name: conv_geom_expected(o, conv_geom_case(_n)))
