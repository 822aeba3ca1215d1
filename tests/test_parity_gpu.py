"""GPU suite: the HIP path (through the C ABI) against the oracle on the same
seeded inputs and against the committed golden fixtures.  Bit-exact for spike
rasters, integer accumulators and -- because the float paths are defined as
fmaf chains in a fixed order -- float membrane potentials too; where a looser
bound applies (any-order float reference) the tolerance is written in the test.
"""
import os
from functools import partial

import numpy as np
import pytest
import torch

from tests import cases
from tests.helpers import input_max_bound, bn_of, packbits_lastaxis, qweight_of

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
  assert torch.cuda.is_available(), "GPU tests need a GPU"
  from snnquantprune_amd import _lib
  _lib.lib()                      # fails loudly if the HIP extension is missing
  return torch.device("cuda:0")


def _t(a, dev):
  return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _np(x):
  from snnquantprune_amd import ops
  if isinstance(x, ops.PackedSpikes):
    return x.bits.cpu().numpy().view(np.uint32)
  return x.cpu().numpy()


def _golden(golden_dir, name):
  with np.load(os.path.join(golden_dir, name + ".npz")) as g:
    return {k: g[k] for k in g.files}


def _weight(leaf, bits, dev, quantized=True, transposed=False):
  """Product-side packing of a reference-style leaf."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import packing
  from snnquantprune_amd.quant import QuantDesc
  a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
  desc = None
  if quantized and a != -1.0:
    desc = QuantDesc(L.Q_DUQ, bits, a, c, float(2 ** (bits - 1) - 1), c)
  mask = leaf.get("prune_0", {}).get("mask")
  pk = packing.PackedKernel(_t(leaf["kernel"], dev), desc,
                            None if mask is None else _t(mask, dev))
  if desc is None:
    return pk.float_weight()
  if transposed:
    n = leaf["kernel"].shape[-1]
    return pk.int_weight_mfma((n + 31) // 32 * 32)
  return pk.int_weight()


def _mslif():
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  return ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)


def _bn(bn, dev):
  from snnquantprune_amd import ops
  mul = (F32(1) / np.sqrt(bn["var"] + F32(1e-5))) * bn["scale"]
  return ops.BnCoeffs(_t(bn["mean"], dev), _t(mul.astype(F32), dev), _t(bn["bias"], dev))


# ---------------------------------------------------------------------------
# weight transforms
# ---------------------------------------------------------------------------

def test_quantizers_bit_exact(dev, oracle, golden_dir):
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.quant_case()
  g = _golden(golden_dir, "quant")
  live = cases.quant_expected(oracle)
  w, mask = _t(c["w"], dev), _t(c["mask"], dev)
  for bits in (2, 3, 4, 8):
    for a, cc in ((1.0, 1.0), (0.73, 0.41)):
      key = "duq_b%d_a%g" % (bits, a)
      fq, codes, flags = ops.quantize(L.Q_DUQ, w, None, bits, a, cc, True, True)
      assert int(flags.item()) == 0
      for ref in (g, live):
        np.testing.assert_array_equal(_np(fq), ref[key + "_fq"])
        np.testing.assert_array_equal(_np(codes), ref[key + "_codes"])
      fqm, codes_m, _ = ops.quantize(L.Q_DUQ, w, mask, bits, a, cc, True, True)
      np.testing.assert_array_equal(_np(fqm), g[key + "_fq_masked"])
      np.testing.assert_array_equal(_np(codes_m), g[key + "_codes"] * c["mask"].astype(np.int8))
    fq, _, _ = ops.quantize(L.Q_UNIFORM_STATIC, w, None, bits, 0.9)
    np.testing.assert_array_equal(_np(fq), g["us_b%d" % bits])
    fq, _, _ = ops.quantize(L.Q_PARAMETRIC_D, w, None, bits, 0.05)
    np.testing.assert_array_equal(_np(fq), g["pd_b%d" % bits])
    fq, _, _ = ops.quantize(L.Q_PARAMETRIC_D_XMAX, w, None, bits, 2 ** -4, 0.8)
    np.testing.assert_array_equal(_np(fq), g["pdx_b%d" % bits])
  # flags: non-binary mask, code overflow (parametric_d with a tiny step)
  _, _, fl = ops.quantize(L.Q_DUQ, w, mask * 0.5, 4, 1.0, 1.0, False, True)
  assert int(fl.item()) & L.FLAG_MASK_NOT_BINARY
  _, _, fl = ops.quantize(L.Q_PARAMETRIC_D, w, None, 12, 1e-3, 0.0, False, True)
  assert int(fl.item()) & L.FLAG_CODE_OVERFLOW


def test_quantizer_modules_match_reference_tests(dev, oracle):
  """quant_test.py invariants through the module surface (init + apply)."""
  from snnquantprune_amd import quant
  rng = np.random.Generator(np.random.PCG64(8627169))
  data8 = np.clip(rng.integers(-128, 127, size=(60, 50)), -127, 127).astype(F32)
  for q in (quant.uniform_static, quant.parametric_d):
    m = q(8)
    x = _t(data8, dev)
    variables = m.init(0, x)
    scale = 1.0
    if "step_size" in variables["quant_params"]:
      scale = float(variables["quant_params"]["step_size"])
    out = m.apply(variables, x * scale)
    np.testing.assert_allclose(_np(out) / scale, data8, rtol=1e-6)
  m = quant.parametric_d_xmax(8, xmax_max=127)
  variables = m.init(0, _t(data8, dev))
  np.testing.assert_allclose(_np(m.apply(variables, _t(data8, dev))), data8)
  # unique values
  data = (rng.uniform(-1, 1, size=(300, 200)) * 23).astype(F32)
  data[0, 0] = 23
  for bits in (2, 3, 5, 8, 11):
    m = quant.uniform_static(bits)
    v = m.init(0, _t(data, dev))
    assert len(np.unique(_np(m.apply(v, _t(data, dev))))) == 2 ** bits - 1
    d = quant.DuQ(bits)
    v = d.init(0, _t(data, dev))
    np.testing.assert_array_equal(_np(d.apply(v, _t(data, dev))), data)   # a == -1
    v["params"]["a"] = torch.full((1,), 23.0, device=dev)
    v["params"]["c"] = torch.full((1,), 23.0, device=dev)
    out = _np(d.apply(v, _t(data, dev)))
    assert len(np.unique(out)) == 2 ** bits - 1
    np.testing.assert_array_equal(out, oracle.duq_forward(data, 23.0, 23.0, bits))
  with pytest.raises(AssertionError):
    quant.uniform_static(1).init(0, _t(data, dev))


# ---------------------------------------------------------------------------
# formats and element-wise pieces
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("C", [1, 31, 32, 33, 64, 110, 130])
def test_pack_unpack_roundtrip(dev, C):
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(C))
  x = (rng.random((5, 7, C)) < 0.3)
  for arr in (x.astype(F32), x.astype(np.uint8)):
    p = ops.pack_bits(_t(arr, dev))
    assert p.shape == (5, 7, C)
    np.testing.assert_array_equal(_np(p), packbits_lastaxis(x))
    np.testing.assert_array_equal(_np(p.to_dense()), x.astype(F32))
  e = ops.pack_bits(torch.zeros((0, C), device=dev))
  assert e.bits.shape[0] == 0


def test_narrowing_passes_report_instead_of_being_read_back(dev):
  """float32 -> uint8 / spike bits in one device pass each, with what they found in a device word
  (the predicate of the float32 launch that follows the integer one): nothing returns to the host."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  x = torch.tensor([0., 1., 3., 127.], device=dev)
  np.testing.assert_array_equal(_np(ops.f32_to_u8(x)), [0, 1, 3, 127])
  xf = torch.zeros(100007, dtype=torch.float32, device=dev)
  xf[3], xf[99990], xf[100006] = 7.0, 201.0, 3.0
  y, pred = ops.narrow_f32_async(xf)
  assert y.dtype == torch.uint8 and int(pred.item()) == 0
  np.testing.assert_array_equal(_np(y), _np(xf).astype(np.uint8))
  for bad in (0.5, -1.0, 256.0, float("nan"), float("inf")):
    xb = xf.clone()
    xb[5] = bad
    assert int(ops.narrow_f32_async(xb)[1].item()) == L.FLAG_NOT_INTEGER, bad
  xs = (torch.rand((37, 70), device=dev) < 0.3).to(torch.float32)
  p, pred = ops.pack_bits_checked(xs)
  assert int(pred.item()) == 0
  np.testing.assert_array_equal(_np(p), packbits_lastaxis(_np(xs)))
  for bad in (2.0, 0.5, -1.0, float("nan")):
    xb = xs.clone()
    xb[36, 69] = bad
    assert int(ops.pack_bits_checked(xb)[1].item()) == L.FLAG_GT_ONE, bad


def test_neurons_bit_exact(dev, oracle, golden_dir):
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.neuron_case()
  g = _golden(golden_dir, "neurons")
  x, u0 = _t(c["x"], dev), _t(c["u0"], dev)
  dec = _t(oracle.sigmoid_f32(c["tau_vec"]), dev)
  k_plif = float(oracle.sigmoid_f32(c["tau_param"]))
  kinds = {
      "mslif_tau2": ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0),
      "mslif_tau3_vr": ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 0.8, 0.1),
      "plif": ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, k_plif, 1.0, 0.0),
      "lif": ops.Neuron(L.NEURON_LIF, 0.0, 1.0, 0.0, decay=dec),
  }
  for name, nrn in kinds.items():
    u, s = ops.lif_forward(x, nrn, u0=u0)
    np.testing.assert_array_equal(_np(s).astype(np.uint8), g[name + "_s"], err_msg=name)
    np.testing.assert_array_equal(_np(u), g[name + "_u"], err_msg=name)
    _, sp = ops.lif_forward(x, nrn, u0=u0, packed_out=True, want_u=False)
    np.testing.assert_array_equal(_np(sp), packbits_lastaxis(g[name + "_s"]))
    assert 0.01 < g[name + "_s"].mean() < 0.6


def test_neuron_modules_single_step(dev, oracle):
  from snnquantprune_amd.spiking_learning import LIF, atan, multi_step_LIF, parametric_leaky_IF
  c = cases.neuron_case()
  x0, u0 = c["x"][0], c["u0"]
  m = multi_step_LIF(tau=2.0, spike_fn=atan)
  (u, s) = m.apply({}, _t(u0, dev), _t(x0, dev))
  eu, es = oracle.multi_step_lif(u0, x0, 2.0)
  np.testing.assert_array_equal(_np(u), eu)
  np.testing.assert_array_equal(_np(s), es)
  p = parametric_leaky_IF(init_tau=3.0, spike_fn=atan)
  v = p.init(0, _t(u0, dev), _t(x0, dev))
  assert v["params"]["tau"].shape == (1,)
  (u, s) = p.apply(v, _t(u0, dev), _t(x0, dev))
  eu, es = oracle.parametric_leaky_if(u0, x0, _np(v["params"]["tau"]))
  np.testing.assert_array_equal(_np(u), eu)
  l = LIF(init_tau=0.5, spike_fn=atan)
  v = l.init(0, _t(u0, dev), _t(x0, dev))
  assert v["params"]["tau"].shape == (x0.shape[-1],)
  (u, s) = l.apply(v, _t(u0, dev), _t(x0, dev))
  eu, es = oracle.lif(u0, x0, _np(v["params"]["tau"]))
  np.testing.assert_array_equal(_np(u), eu)
  np.testing.assert_array_equal(_np(s), es)


def test_bn_pool_vote(dev, oracle):
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(4))
  x = rng.standard_normal((3, 2, 6, 8, 40)).astype(F32)
  bn = dict(mean=rng.standard_normal(40).astype(F32), var=(rng.random(40) + 0.5).astype(F32),
            scale=rng.standard_normal(40).astype(F32), bias=rng.standard_normal(40).astype(F32))
  y = ops.batchnorm_forward(_t(x, dev), _bn(bn, dev))
  np.testing.assert_array_equal(_np(y), oracle.batchnorm_eval(x, **bn))
  np.testing.assert_array_equal(_np(ops.maxpool2x2(_t(x, dev))), oracle.max_pool_2x2(x))
  s = (rng.random((3, 2, 6, 8, 40)) < 0.2).astype(F32)
  ps = ops.pack_bits(_t(s, dev))
  np.testing.assert_array_equal(_np(ops.maxpool2x2(ps)),
                                packbits_lastaxis(oracle.max_pool_2x2(s)))
  sp = (rng.random((9, 5, 110)) < 0.3).astype(F32)
  np.testing.assert_array_equal(_np(ops.vote(_t(sp, dev))), oracle.vote(sp))
  np.testing.assert_array_equal(_np(ops.vote(ops.pack_bits(_t(sp, dev)))), oracle.vote(sp))
  with pytest.raises(ValueError):
    ops.vote(_t(sp, dev), group=7)


# ---------------------------------------------------------------------------
# dense blocks
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("counts", [False, True])
def test_dense_block_int_path(dev, oracle, golden_dir, counts):
  """Ragged K = 200, N = 70; integer accumulators, rasters and u bit-exact."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.dense_block_case(counts=counts)
  g = _golden(golden_dir, "dense_block_counts" if counts else "dense_block")
  live = cases.dense_block_expected(oracle, c)
  T, B, K = c["x"].shape
  N = c["leaf"]["kernel"].shape[1]
  w = _weight(c["leaf"], c["bits"], dev)
  x = _t(c["x"], dev)
  inputs = [x] if counts else [x, ops.pack_bits(x)]
  for xin in inputs:
    x4 = xin.reshape(T * B, 1, 1, K) if isinstance(xin, torch.Tensor) \
        else xin.reshape_leading(T * B, 1, 1)
    y, acc = ops.conv_forward(x4, ops.ConvGeom(1, 1, K, N, 1, 1), w, want_acc=True)
    np.testing.assert_array_equal(_np(acc).reshape(T, B, N), g["acc"])
    qw = qweight_of(oracle, c["leaf"], c["bits"])
    np.testing.assert_array_equal(_np(y).reshape(T, B, N), qw.dequant_acc(g["acc"]))
    for packed in (False, True):
      u, s = ops.dense_lif_forward(xin, w, K, N, _mslif(), u0=_t(c["u0"], dev),
                                   packed_out=packed, impl=L.IMPL_GENERIC)
      for ref in (g, live):
        np.testing.assert_array_equal(_np(u), ref["u"])
        exp_s = packbits_lastaxis(ref["s"]) if packed else ref["s"].astype(F32)
        np.testing.assert_array_equal(_np(s), exp_s)
  assert 0.02 < g["s"].mean() < 0.4
  # batch-major input read by strides gives the same result
  xb = _t(np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)), dev)
  u, s = ops.dense_lif_forward(xb, w, K, N, _mslif(), u0=_t(c["u0"], dev),
                               impl=L.IMPL_GENERIC, time_major=False)
  np.testing.assert_array_equal(_np(s), g["s"].astype(F32))
  # int8 codes with float32 input are refused (the caller passes the fake-quant kernel)
  with pytest.raises(ValueError):        # ... unless the float32 kernel stands by (ops.FloatFallback)
    ops.dense_lif_forward(x.to(torch.float32), w, K, N, _mslif(), impl=L.IMPL_GENERIC)


@pytest.mark.parametrize("shape", [(6, 5, 256, 70), (20, 37, 2048, 512), (10, 3, 96, 160),
                                   (33, 9, 64, 33)],
                         ids=["small", "c2_layer1", "ragged_n", "long_t"])
def test_dense_block_mfma(dev, oracle, shape):
  """int8 MFMA dense block vs the oracle and vs the direct-form kernel:
  rasters and final u bit-exact, incl. a non-zero carry and batch-major input."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N)
  e = cases.dense_block_expected(oracle, c)
  assert 0.01 < e["s"].mean() < 0.5
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert w.wt is not None
  x = ops.pack_bits(_t(c["x"], dev))
  u0 = _t(c["u0"], dev)
  u, s = ops.dense_lif_forward(x, w, K, N, _mslif(), u0=u0, packed_out=True, impl=L.IMPL_MFMA)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(e["s"]))
  np.testing.assert_array_equal(_np(u), e["u"])
  ug, sg = ops.dense_lif_forward(x, w, K, N, _mslif(), u0=u0, packed_out=True,
                                 impl=L.IMPL_GENERIC)
  np.testing.assert_array_equal(_np(s), _np(sg))
  np.testing.assert_array_equal(_np(u), _np(ug))
  xb = ops.pack_bits(_t(np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)), dev))
  ub, sb = ops.dense_lif_forward(xb, w, K, N, _mslif(), u0=u0, packed_out=True,
                                 impl=L.IMPL_MFMA, time_major=False)
  np.testing.assert_array_equal(_np(sb), _np(s))
  for nrn in (ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 0.8, 0.1),
              ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, 0.3, 1.0, 0.0),
              ops.Neuron(L.NEURON_LIF, 0.0, 1.0, 0.0,
                         decay=_t(np.linspace(0.2, 0.9, N).astype(F32), dev))):
    ua, sa = ops.dense_lif_forward(x, w, K, N, nrn, packed_out=True, impl=L.IMPL_MFMA)
    ub, sb = ops.dense_lif_forward(x, w, K, N, nrn, packed_out=True, impl=L.IMPL_GENERIC)
    np.testing.assert_array_equal(_np(sa), _np(sb))
    np.testing.assert_array_equal(_np(ua), _np(ub))


@pytest.mark.parametrize("shape", [(20, 40, 4100, 110, False), (7, 100, 8192, 200, True),
                                   (90, 4, 4096, 33, False), (20, 64, 32768, 110, True)],
                         ids=["odd_k_odd_chunks", "ragged_rows_two_col_blocks_bn", "long_t",
                              "readout_shape_bn"])
def test_dense_block_long_contractions(dev, oracle, shape):
  """Long contractions on the fused MFMA dense kernel (register rings several chunks deep,
  staging interleaved with the MFMAs) against the oracle and the direct-form kernel --
  rasters and final potentials bit-exact; K that is neither a multiple of 32 nor of the
  chunk groups, rows that do not fill the row tile, N beyond one 128-feature block, a sample
  that nearly fills the tile (T = 90), BatchNorm, a carry, batch-major input, every neuron."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N, with_bn = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N, bits=4, p=0.8)
  qw = qweight_of(oracle, c["leaf"], 4)
  rng = np.random.Generator(np.random.PCG64(4100 + T))
  bn = None
  if with_bn:
    bn = {"scale": (1 + 0.2 * rng.standard_normal(N)).astype(F32), "bias": (0.1 * rng.standard_normal(N)).astype(F32),
          "mean": (0.1 * rng.standard_normal(N)).astype(F32), "var": (1 + 0.3 * rng.random(N)).astype(F32)}
  if bn:      # the oracle's dense block has no norm: the same layer as a 1x1 convolution block
    leaf4 = {"kernel": c["leaf"]["kernel"].reshape(1, 1, K, N), "DuQ_0": c["leaf"]["DuQ_0"],
             "prune_0": {"mask": c["leaf"]["prune_0"]["mask"].reshape(1, 1, K, N)}}
    eu, es = oracle.conv_block(c["x"].reshape(T, B, 1, 1, K), qweight_of(oracle, leaf4, 4), bn, None,
                               "int", padding=((0, 0), (0, 0)), u0=c["u0"].reshape(B, 1, 1, N))
    eu, es = eu.reshape(B, N), es.reshape(T, B, N)
  else:
    eu, es = oracle.dense_block(c["x"], qw, None, "int", u0=c["u0"])
  assert 0.01 < es.mean() < 0.6
  w = _weight(c["leaf"], 4, dev, transposed=True)
  x = ops.pack_bits(_t(c["x"], dev))
  u0 = _t(c["u0"], dev)
  bnc = _bn(bn, dev) if bn else None
  u, s = ops.dense_lif_forward(x, w, K, N, _mslif(), bn=bnc, u0=u0, packed_out=True, impl=L.IMPL_MFMA)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  ug, sg = ops.dense_lif_forward(x, w, K, N, _mslif(), bn=bnc, u0=u0, packed_out=True, impl=L.IMPL_GENERIC)
  np.testing.assert_array_equal(_np(s), _np(sg))
  np.testing.assert_array_equal(_np(u), _np(ug))
  xb = ops.pack_bits(_t(np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)), dev))
  ub, sb = ops.dense_lif_forward(xb, w, K, N, _mslif(), bn=bnc, u0=u0, packed_out=True,
                                 impl=L.IMPL_MFMA, time_major=False)
  np.testing.assert_array_equal(_np(sb), _np(s))
  np.testing.assert_array_equal(_np(ub), _np(u))
  for nrn in (ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 0.8, 0.1),
              ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, 0.3, 1.0, 0.0),
              ops.Neuron(L.NEURON_LIF, 0.0, 1.0, 0.0,
                         decay=_t(np.linspace(0.2, 0.9, N).astype(F32), dev))):
    ua, sa = ops.dense_lif_forward(x, w, K, N, nrn, packed_out=True, impl=L.IMPL_MFMA)
    ub, sb = ops.dense_lif_forward(x, w, K, N, nrn, packed_out=True, impl=L.IMPL_GENERIC)
    np.testing.assert_array_equal(_np(sa), _np(sb))
    np.testing.assert_array_equal(_np(ua), _np(ub))


def test_dense_block_long_T_falls_back(dev, oracle):
  """More timesteps than the MFMA dense kernel's row tile holds (T > 96): the block
  runs on the direct-form kernel instead of failing."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(77))
  T, B, K, N = 170, 2, 64, 32
  leaf = {"kernel": (rng.standard_normal((K, N)) * 0.6).astype(F32),
          "DuQ_0": {"a": F32([1.0]), "c": F32([0.9])},
          "prune_0": {"mask": (rng.random((K, N)) > 0.5).astype(F32)}}
  x = (rng.random((T, B, K)) < 0.2).astype(np.uint8)
  w = _weight(leaf, 4, dev, transposed=True)
  u, s = ops.dense_lif_forward(ops.pack_bits(_t(x, dev)), w, K, N, _mslif(), packed_out=True)
  eu, es = oracle.dense_block(x, qweight_of(oracle, leaf, 4), None, "int")
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)


def test_dense_block_k_not_multiple_of_32(dev, oracle):
  """K = 784 (a flattened 28x28 image) and other sizes that are not multiples of 32: the
  codes are tiled with zero rows up to the next k-step, the block stays on the MFMA kernel."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(78))
  for K, N in ((784, 100), (50, 33)):
    T, B = 6, 5
    std = 8.0 / np.sqrt(K)
    leaf = {"kernel": (rng.standard_normal((K, N)) * std).astype(F32),
            "DuQ_0": {"a": F32([3 * std]), "c": F32([3 * std])},
            "prune_0": {"mask": (rng.random((K, N)) > 0.5).astype(F32)}}
    x = (rng.random((T, B, K)) < 0.2).astype(np.uint8)
    w = _weight(leaf, 4, dev, transposed=True)
    assert w.wt is not None and w.wt.shape[1] == (K + 31) // 32
    u, s = ops.dense_lif_forward(ops.pack_bits(_t(x, dev)), w, K, N, _mslif(), packed_out=True,
                                 impl=L.IMPL_MFMA)
    eu, es = oracle.dense_block(x, qweight_of(oracle, leaf, 4), None, "int")
    assert es.mean() > 0.01
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(u), eu)


def test_dense_block_fseq_path(dev, oracle, golden_dir):
  """Unquantised float32 weights, real-valued input: k-ascending fmaf chain,
  bit-exact vs the oracle's fseq mode and within 1e-5 relative of float64."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.dense_fseq_case()
  g = _golden(golden_dir, "dense_fseq")
  T, B, K = c["x"].shape
  N = c["leaf"]["kernel"].shape[1]
  w = _weight(c["leaf"], 8, dev, quantized=False)
  assert w.wtype == L.W_F32
  y = ops.conv_forward(_t(c["x"], dev).reshape(T * B, 1, 1, K), ops.ConvGeom(1, 1, K, N, 1, 1), w)
  np.testing.assert_array_equal(_np(y).reshape(T, B, N), g["y"])
  ref64 = c["x"].astype(np.float64) @ c["leaf"]["kernel"].astype(np.float64)
  np.testing.assert_allclose(_np(y).reshape(T, B, N), ref64, rtol=1e-5, atol=1e-6)
  u, s = ops.dense_lif_forward(_t(c["x"], dev), w, K, N, _mslif())
  np.testing.assert_array_equal(_np(u), g["u"])
  np.testing.assert_array_equal(_np(s), g["s"].astype(F32))


def test_quant_dense_module_matches_plain_matmul(dev, oracle):
  """flax_qdense_test.py: QuantDense with an empty config == x @ W (+ bias)."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd.flax_qdense import QuantDense
  rng = np.random.Generator(np.random.PCG64(12))
  cfg = nn.ConfigDict({"prune_percentage": -1.0})
  for (m, k, n) in ((512, 100, 20), (64, 1, 1), (256, 1, 200)):
    x = rng.standard_normal((m, k)).astype(F32)
    layer = QuantDense(n, config=cfg)
    v = layer.init(3, _t(x, dev))
    assert sorted(v["params"]) == ["bias", "kernel"]
    v["params"]["bias"] = _t(rng.standard_normal(n).astype(F32), dev)
    y = _np(layer.apply(v, _t(x, dev)))
    wk, b = _np(v["params"]["kernel"]), _np(v["params"]["bias"])
    np.testing.assert_array_equal(y, oracle.fseq_matmul(x, wk) + b)
    np.testing.assert_allclose(y, x.astype(np.float64) @ wk + b, rtol=1e-5, atol=1e-5)
  with pytest.raises(AttributeError):        # stale configs fail as in the reference
    QuantDense(4, config=nn.ConfigDict({})).init(0, _t(x, dev))


# ---------------------------------------------------------------------------
# convolutions
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("geom", cases.REF_CONV_GEOMS, ids=[g[0] for g in cases.REF_CONV_GEOMS])
def test_quant_conv_reference_geometries(dev, oracle, golden_dir, geom):
  """flax_qconv_test.py:148-285 through the QuantConv module, tolerance 0.0."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd.flax_qconv import QuantConv
  name, H, W, ks, st, pad, OH, OW = geom
  c = cases.conv_geom_case(name)
  g = _golden(golden_dir, "conv_geom_" + name)
  cfg = nn.ConfigDict({"prune_percentage": -1.0})
  c1 = QuantConv(features=10, kernel_size=ks, strides=st, padding=pad, use_bias=False, config=cfg)
  c2 = QuantConv(features=20, kernel_size=ks, strides=st, padding=pad, use_bias=False, config=cfg)
  y1 = c1.apply({"params": {"kernel": _t(c["k1"], dev)}}, _t(c["x"], dev))
  assert tuple(y1.shape) == (2, OH, OW, 10)
  y2 = c2.apply({"params": {"kernel": _t(c["k2"], dev)}}, y1)
  np.testing.assert_array_equal(_np(y1), g["y1"])
  np.testing.assert_array_equal(_np(y2), g["y2"])
  # single (batch-less) input, flax_qconv.py:109-112
  ys = c1.apply({"params": {"kernel": _t(c["k1"], dev)}}, _t(c["x"][0], dev))
  np.testing.assert_array_equal(_np(ys), g["y1"][0])


def test_conv_generic_dilation_groups_1d_and_int_paths(dev, oracle):
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops
  from snnquantprune_amd.flax_qconv import QuantConv
  rng = np.random.Generator(np.random.PCG64(5))
  cfg = nn.ConfigDict({"prune_percentage": -1.0})
  x = rng.standard_normal((2, 9, 11, 4)).astype(F32)
  k = rng.standard_normal((3, 2, 2, 6)).astype(F32)
  conv = QuantConv(features=6, kernel_size=(3, 2), strides=(2, 1), padding=((1, 2), (0, 1)),
                   input_dilation=(1, 2), kernel_dilation=(2, 1), feature_group_count=2,
                   use_bias=False, config=cfg)
  y = conv.apply({"params": {"kernel": _t(k, dev)}}, _t(x, dev))
  e = oracle.quant_conv(x, oracle.QWeight(k), (2, 1), ((1, 2), (0, 1)), (1, 2), (2, 1), 2, "fseq")
  np.testing.assert_array_equal(_np(y), e)
  with pytest.raises(AssertionError):          # flax_qconv.py:117
    QuantConv(features=6, kernel_size=(3, 2), feature_group_count=3, config=cfg).init(
        0, _t(x, dev))
  # 1-D SAME k = 4 (the TCJA convolutions)
  x1 = rng.standard_normal((3, 10, 5)).astype(F32)
  k1 = rng.standard_normal((4, 5, 7)).astype(F32)
  c1 = QuantConv(features=7, kernel_size=[4], padding="SAME", use_bias=False, config=cfg)
  y1 = c1.apply({"params": {"kernel": _t(k1, dev)}}, _t(x1, dev))
  np.testing.assert_array_equal(_np(y1), oracle.quant_conv(x1, oracle.QWeight(k1), None, "SAME",
                                                            mode="fseq"))
  # integer path of the direct-form kernel on a strided, padded, grouped conv
  leaf = {"kernel": (rng.standard_normal((3, 3, 8, 24)) * 0.2).astype(F32),
          "DuQ_0": {"a": F32([0.5]), "c": F32([0.45])},
          "prune_0": {"mask": (rng.random((3, 3, 8, 24)) > 0.6).astype(F32)}}
  xs = (rng.random((3, 13, 10, 16)) < 0.3).astype(np.uint8)
  qw = qweight_of(oracle, leaf, 4)
  w = _weight(leaf, 4, dev)
  geom = ops.ConvGeom(13, 10, 16, 24, 3, 3, (2, 1), ((1, 1), (0, 2)), groups=2)
  for xin in (_t(xs, dev), ops.pack_bits(_t(xs, dev))):
    yy, acc = ops.conv_forward(xin, geom, w, want_acc=True)
    eacc = oracle.quant_conv(xs, qw, (2, 1), ((1, 1), (0, 2)), feature_group_count=2,
                             mode="int", return_acc=True)
    np.testing.assert_array_equal(_np(acc), eacc)
    np.testing.assert_array_equal(_np(yy), qw.dequant_acc(eacc))


@pytest.mark.parametrize("which", ["conv_block_c128", "conv_block_c2"])
def test_conv_block_mfma_and_generic(dev, oracle, golden_dir, which):
  """The fused conv + BN + LIF (+ pool) block: MFMA kernel and direct-form
  kernel both bit-exact vs the oracle (rasters, pooled rasters, final u)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case() if which.endswith("c128") else \
      cases.conv_block_case(hw=16, cin=2, seed=961, gain=4.0)
  g = _golden(golden_dir, which)
  live = cases.conv_block_expected(oracle, c)
  T, B, H, W, Cin = c["x"].shape
  Cout = c["leaf"]["kernel"].shape[-1]
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn = _bn(c["bn"], dev)
  geom = ops.ConvGeom(H, W, Cin, Cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
  x = _t(c["x"], dev)
  xin = x if Cin == 2 else ops.pack_bits(x)
  for impl in (L.IMPL_GENERIC, L.IMPL_MFMA):
    u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=bn, packed_out=True, pool=1, impl=impl)
    for ref in (g, live):
      np.testing.assert_array_equal(_np(s), ref["s_bits"], err_msg="impl %d" % impl)
      np.testing.assert_array_equal(_np(u), ref["u"], err_msg="impl %d" % impl)
  _, sp = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=bn, packed_out=True, pool=2,
                               impl=L.IMPL_MFMA, want_u=False)
  np.testing.assert_array_equal(_np(sp), g["pooled_bits"])
  # continuing from a non-zero carry: run T in two halves
  u1, s1 = ops.conv_lif_forward(xin[:2], geom, w, _mslif(), bn=bn, packed_out=True,
                                impl=L.IMPL_MFMA)
  u2, s2 = ops.conv_lif_forward(xin[2:], geom, w, _mslif(), bn=bn, u0=u1, packed_out=True,
                                impl=L.IMPL_MFMA)
  np.testing.assert_array_equal(np.concatenate([_np(s1), _np(s2)]), g["s_bits"])
  np.testing.assert_array_equal(_np(u2), g["u"])
  # batch-major input (the model's [B, T, ...] layout) read by strides
  xb = _t(np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)), dev)
  xbin = xb if Cin == 2 else ops.pack_bits(xb)
  _, sb = ops.conv_lif_forward(xbin, geom, w, _mslif(), bn=bn, packed_out=True, pool=2,
                               impl=L.IMPL_MFMA, want_u=False, time_major=False)
  np.testing.assert_array_equal(_np(sb), g["pooled_bits"])
  # other neuron kinds take the general epilogue
  nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 0.8, 0.1)
  ua, sa = ops.conv_lif_forward(xin, geom, w, nrn, bn=bn, packed_out=True, impl=L.IMPL_MFMA)
  ub, sb = ops.conv_lif_forward(xin, geom, w, nrn, bn=bn, packed_out=True, impl=L.IMPL_GENERIC)
  np.testing.assert_array_equal(_np(sa), _np(sb))
  np.testing.assert_array_equal(_np(ua), _np(ub))
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  eu, es = oracle.conv_block(c["x"], qw, c["bn"],
                             {"tau": 3.0, "v_threshold": 0.8, "v_reset": 0.1}, "int")
  np.testing.assert_array_equal(_np(sa), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(ua), eu)


@pytest.mark.parametrize("T", [1, 2, 5, 33])
def test_conv_block_pipeline_tails(dev, oracle, T):
  """The t-pipelined MFMA kernels at odd / even / single step counts and across
  the flush period of the LDS-staged spike words (32 steps): pooled rasters and
  final membrane potentials bit-exact."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  for cin, hw in ((128, 8), (2, 16)):
    c = cases.conv_block_case(T=T, B=2, hw=hw, cin=cin, seed=970 + T, gain=5.0 if cin > 2 else 4.0)
    e = cases.conv_block_expected(oracle, c)
    w = _weight(c["leaf"], c["bits"], dev, transposed=True)
    geom = ops.ConvGeom(hw, hw, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
    x = _t(c["x"], dev)
    xin = x if cin == 2 else ops.pack_bits(x)
    for pool, key in ((2, "pooled_bits"), (1, "s_bits")):
      u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                  pool=pool, impl=L.IMPL_MFMA, x_max=input_max_bound(xin))
      np.testing.assert_array_equal(_np(s), e[key], err_msg="T=%d cin=%d pool=%d" % (T, cin, pool))
      np.testing.assert_array_equal(_np(u), e["u"])


@pytest.mark.parametrize("mode", ["channel", "channel_nobn", "shared_counts", "shared_8bit",
                                  "none_xmax", "c128_none", "channel_cout160",
                                  "c128_i8_5bit", "c128_i8_8bit", "c128_fp6_cout160",
                                  "channel_tiny_currents", "channel_T40", "none_u8_255",
                                  "shared_tiny_currents", "shared_T40"])
def test_conv_block_table_modes(dev, oracle, mode):
  """The MFMA kernels dequantise through LDS tables when the accumulator bound
  allows (per-channel tables with BatchNorm folded in, one shared table, or plain
  arithmetic); every mode gives the oracle's rasters and membrane potentials."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  cin, hw, bits, cout, lam, x_hint = 2, 16, 4, 128, None, None
  if mode in ("shared_8bit", "shared_tiny_currents", "shared_T40"):
    bits = 8
  if mode.startswith("c128"):
    cin, hw = 128, 8
  if mode == "c128_i8_5bit":          # codes up to 15: int8 MFMA kernel, shared table
    bits = 5
  if mode == "c128_i8_8bit":          # int8 MFMA kernel, arithmetic dequant
    bits = 8
  if mode in ("channel_cout160", "c128_fp6_cout160"):
    cout = 160
  # conv0 runs the membrane update as one fused multiply-add when the table proves no
  # subnormal can arise within T steps; channel_tiny_currents (currents ~2^-126: subnormal membrane potentials) and
  # a carried-in u0 (below) must take the two-step form, channel_T40 the fused one
  # (the shared-table kernel proves the same from BatchNorm of every table entry)
  T = 40 if mode.endswith("_T40") else 4
  c = cases.conv_block_case(T=T, B=3, hw=hw, cin=cin, cout=cout, bits=bits, seed=1201,
                            gain=5.0 if cin > 2 else 4.0, random_bn=mode != "channel_nobn")
  if mode.endswith("tiny_currents"):
    z, one = np.zeros(cout, F32), np.ones(cout, F32)
    c["bn"] = dict(mean=z, var=one, scale=(one * F32(2.0 ** -126)).astype(F32), bias=z, eps=0.0)
  x = c["x"]
  if cin == 2:
    if mode.startswith("channel") or mode in ("shared_8bit", "shared_tiny_currents", "shared_T40"):
      x = np.minimum(x, 1).astype(np.uint8)              # binary events: bound = sum |code|
    elif mode == "shared_counts":
      x = (x * 5).astype(np.uint8)                        # counts up to ~20
    elif mode == "none_xmax":
      x = (x * 30).astype(np.uint8)                       # counts > 31: no table
    elif mode == "none_u8_255":
      x = np.minimum(x.astype(np.int32) * 85, 255).astype(np.uint8)   # counts up to 255
    c["x"] = x
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(hw, hw, cin, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xt = _t(c["x"], dev)
  xin = xt if cin == 2 else ops.pack_bits(xt)
  x_max = input_max_bound(xin)
  bound = int(w.abs_sum_max) * x_max
  if mode.startswith("channel"):
    assert 0 < bound <= 40, bound
  elif mode.startswith("shared"):
    assert 40 < bound <= 4095 and x_max <= 31, (bound, x_max)
  elif mode == "none_xmax":
    assert 31 < x_max <= 127
  elif mode == "none_u8_255":
    assert x_max == 255
  if mode == "c128_none":
    x_max = 0                                             # no bound given: arithmetic dequant
  bn = _bn(c["bn"], dev) if mode != "channel_nobn" else None
  if mode.endswith("tiny_currents"):
    bn = ops.BnCoeffs(_t(c["bn"]["mean"], dev), _t(c["bn"]["scale"], dev), _t(c["bn"]["bias"], dev))
  if bn is None:
    qw = qweight_of(oracle, c["leaf"], c["bits"])
    eu, es = oracle.conv_block(c["x"], qw, None, None, "int")
    e = {"u": eu, "s_bits": packbits_lastaxis(es),
         "pooled_bits": packbits_lastaxis(oracle.max_pool_2x2(es))}
  for pool, key in ((2, "pooled_bits"), (1, "s_bits")):
    u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=bn, packed_out=True, pool=pool,
                                impl=L.IMPL_MFMA, x_max=x_max)
    np.testing.assert_array_equal(_np(s), e[key], err_msg="%s pool=%d" % (mode, pool))
    np.testing.assert_array_equal(_np(u), e["u"])


def test_conv_fp6_and_int8_mfma_kernels_agree(dev, oracle):
  """Codes of magnitude <= 7 take the fp4 x fp6 MFMA kernel; the same weights with
  code_max withheld take the int8 MFMA kernel: identical rasters and potentials
  (and both equal the oracle), with and without the dequant table."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=7, B=9, hw=16, cin=128, seed=1301, gain=5.0)
  e = cases.conv_block_expected(oracle, c)
  w6 = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert 0 < w6.code_max <= 7
  w8 = dataclasses.replace(w6, code_max=0)
  geom = ops.ConvGeom(16, 16, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  bn = _bn(c["bn"], dev)
  for x_max in (1, 0):
    for pool, key in ((2, "pooled_bits"), (1, "s_bits")):
      outs = [ops.conv_lif_forward(xin, geom, w, _mslif(), bn=bn, packed_out=True, pool=pool,
                                   impl=L.IMPL_MFMA, x_max=x_max) for w in (w6, w8)]
      for u, s in outs:
        np.testing.assert_array_equal(_np(s), e[key])
        np.testing.assert_array_equal(_np(u), e["u"])


@pytest.mark.parametrize("shape", ["dense_2048_512", "dense_odd", "conv3x3_c128", "conv3x3_c16",
                                   "conv5x5", "conv1d_k4_same", "conv_1x1"])
def test_float_connection_f32_mfma(dev, oracle, shape):
  """float32 activations x float32 (fake-quantised) kernels run on the f32 MFMA
  (fseq_gemm.hip); the result is the oracle's k-ascending fmaf chain bit for bit,
  for sizes that are not multiples of the 128 x 128 tile as well."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(4401))
  mixed = lambda shp: (rng.standard_normal(shp) * np.exp2(rng.integers(-6, 7, shp))).astype(F32)
  if shape.startswith("dense"):
    M, K, N = (1280, 2048, 512) if shape == "dense_2048_512" else (37, 20, 7)
    x, k = mixed((M, K)), mixed((K, N))
    geom = ops.ConvGeom(1, 1, K, N, 1, 1)
    assert ops.fseq_gemm_supported(geom)
    y = ops.conv_forward(_t(x.reshape(M, 1, 1, K), dev), geom, ops.Weight(L.W_F32, _t(k, dev)))
    np.testing.assert_array_equal(_np(y).reshape(M, N), oracle.fseq_matmul(x, k))
    return
  NB, H, W, Cin, Cout, KH, KW, pad = {
      "conv3x3_c128": (5, 8, 8, 128, 128, 3, 3, ((1, 1), (1, 1))),
      "conv3x3_c16": (3, 7, 9, 16, 40, 3, 3, ((1, 1), (1, 1))),
      "conv5x5": (2, 6, 5, 8, 33, 5, 5, ((2, 2), (2, 2))),
      "conv1d_k4_same": (4, 1, 20, 128, 20, 1, 4, ((0, 0), (1, 2))),
      "conv_1x1": (3, 4, 4, 12, 130, 1, 1, ((0, 0), (0, 0))),
  }[shape]
  x, k = mixed((NB, H, W, Cin)), mixed((KH, KW, Cin, Cout))
  geom = ops.ConvGeom(H, W, Cin, Cout, KH, KW, (1, 1), pad)
  assert ops.fseq_gemm_supported(geom)
  y = ops.conv_forward(_t(x, dev), geom, ops.Weight(L.W_F32, _t(k, dev)))
  e = oracle.quant_conv(x, oracle.QWeight(k), None, pad, mode="fseq")
  np.testing.assert_array_equal(_np(y), e)


@pytest.mark.parametrize("cin", [128, 2])
@pytest.mark.parametrize("kind", ["plif", "plif_vr", "lif", "lif_vr", "mslif_tau3", "mslif_tau4_vr"])
def test_conv_block_neuron_forms(dev, oracle, cin, kind):
  """Every neuron of spiking_learning.py has a straight-line epilogue on the MFMA conv
  kernels (fp6 / conv0): parametric_leaky_IF (:381), LIF with per-feature decay (:432),
  multi_step_LIF with a non-power-of-two tau (a true division, :410) and non-zero
  v_reset -- bit-exact with the oracle and with the direct-form kernel."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  hw = 8 if cin == 128 else 16
  c = cases.conv_block_case(T=6, B=4, hw=hw, cin=cin, seed=1501, gain=5.0 if cin > 2 else 4.0)
  if cin == 2:
    c["x"] = np.minimum(c["x"], 1).astype(np.uint8)
  vr = 0.15 if kind.endswith("_vr") else 0.0
  sig = lambda v: (1.0 / (1.0 + np.exp(-np.asarray(v, np.float64)))).astype(F32)
  if kind.startswith("plif"):
    tau_param = F32(-0.35)
    nrn = ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, float(sig(tau_param)), 1.0, vr)
    ocfg = {"kind": "parametric_leaky_IF", "tau_param": tau_param}
  elif kind.startswith("lif"):
    tau_vec = np.random.Generator(np.random.PCG64(9)).uniform(-1.0, 2.0, 128).astype(F32)
    nrn = ops.Neuron(L.NEURON_LIF, 1.0, 1.0, vr, decay=_t(sig(tau_vec), dev))
    ocfg = {"kind": "LIF", "tau_vec": tau_vec}
  else:
    tau = 3.0 if "tau3" in kind else 4.0
    nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, tau, 1.0, vr)
    ocfg = {"kind": "multi_step_LIF", "tau": tau}
  ocfg.update(v_threshold=1.0, v_reset=vr)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  eu, es = oracle.conv_block(c["x"], qw, c["bn"], ocfg, "int")
  assert 0.01 < es.mean() < 0.6, es.mean()
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(hw, hw, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xt = _t(c["x"], dev)
  xin = xt if cin == 2 else ops.pack_bits(xt)
  for impl in (L.IMPL_MFMA, L.IMPL_GENERIC):
    for x_max in (1, 0):                  # with and without the dequant tables
      u, s = ops.conv_lif_forward(xin, geom, w, nrn, bn=_bn(c["bn"], dev), packed_out=True,
                                  impl=impl, x_max=x_max)
      np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg="impl %d" % impl)
      np.testing.assert_array_equal(_np(u), eu)
  _, sp = ops.conv_lif_forward(xin, geom, w, nrn, bn=_bn(c["bn"], dev), packed_out=True, pool=2,
                               impl=L.IMPL_MFMA, want_u=False, x_max=1)
  np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es)))
  if cin == 128:                          # the int8 kernel (codes wider than fp6 holds)
    import dataclasses
    w8 = dataclasses.replace(w, code_max=0)
    u, s = ops.conv_lif_forward(xin, geom, w8, nrn, bn=_bn(c["bn"], dev), packed_out=True,
                                impl=L.IMPL_MFMA, x_max=1)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(u), eu)


def test_unquantised_conv_net_on_f32_mfma(dev, oracle):
  """A network whose kernels pass through unquantised (DuQ a = -1: float32 kernels),
  fed by event counts / spikes: every block takes the f32-MFMA connection (inputs
  widened on the fly, the batch walked in slices) and equals the oracle's fseq mode bit
  for bit, logits included."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  from snnquantprune_amd import spiking_learning as sl
  v = syn.conv_net_variables(hw=16, quantized=False, prune_p=0.5, random_bn=True,
                             gains=(5.0, 7.0, 8.0, 12.0))
  x = syn.poisson_counts((5, 4, 16, 16, 2), 0.25, seed=977)          # [B, T, H, W, 2] counts
  p = v["params"]
  r = oracle.conv3_dense_forward(
      x, [qweight_of(oracle, p["QuantConv_%d" % i], 4) for i in range(3)],
      [bn_of(v, i) for i in range(3)], qweight_of(oracle, p["QuantDense_0"], 4), mode="fseq")
  rates = [r["pool%d" % i].mean() for i in range(3)]
  assert all(0.01 < q < 0.7 for q in rates), rates
  cfg = syn.make_config(bits=4, prune_percentage=0.5)
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  old = sl.SpikingBlock._float_block
  slices = []
  def spy(self, x_, tm, is_dense, geom, *a, **k):       # the route really is taken
    slices.append(geom.tag())
    return old(self, x_, tm, is_dense, geom, *a, **k)
  sl.SpikingBlock._float_block = spy
  try:
    (logits, _), mut = model.apply(nn.tree_from_numpy(v, dev), _t(x, dev), trgt=None,
                                   train=False, rng=None, mutable=["intermediates"])
  finally:
    sl.SpikingBlock._float_block = old
  assert len(slices) == 4, slices
  for i in range(3):
    np.testing.assert_array_equal(_np(mut["intermediates"]["pool%d" % i][0]),
                                  packbits_lastaxis(r["pool%d" % i]))
  np.testing.assert_array_equal(_np(logits), r["logits"])


def test_conv_block_xcd_split_schedule(dev, oracle):
  """Batches of 8 or more samples take the XCD-aware patch schedule (samples
  b = xcd mod 8 per XCD); uneven B = 19 leaves XCDs with different sample counts."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  for cin, hw in ((128, 16), (2, 16)):
    c = cases.conv_block_case(T=3, B=19, hw=hw, cin=cin, seed=985, gain=5.0 if cin > 2 else 4.0)
    e = cases.conv_block_expected(oracle, c)
    w = _weight(c["leaf"], c["bits"], dev, transposed=True)
    geom = ops.ConvGeom(hw, hw, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
    x = _t(c["x"], dev)
    xin = x if cin == 2 else ops.pack_bits(x)
    u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                pool=2, impl=L.IMPL_MFMA, x_max=input_max_bound(xin))
    np.testing.assert_array_equal(_np(s), e["pooled_bits"])
    np.testing.assert_array_equal(_np(u), e["u"])


@pytest.mark.parametrize("shape", ["c128_34x34", "c128_17x17", "c128_12x20", "c2_34x34", "c2_28x28",
                                   "c2_5x3"])
def test_conv_block_any_image_size(dev, oracle, shape):
  """Image sizes that are not multiples of the 8x8 patch (N-MNIST 34x34, MNIST 28x28,
  the 17x17 a pooled 34x34 leaves, odd sizes): the MFMA kernels clip the edge patches;
  rasters, pooled rasters (VALID 2x2: the odd last row / column is dropped), potentials
  and the carry-in path bit-exact."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  cin = 128 if shape.startswith("c128") else 2
  H, W = [int(v) for v in shape.split("_")[1].split("x")]
  rng = np.random.Generator(np.random.PCG64(3000 + H * 64 + W))
  c = cases.conv_block_case(T=4, B=3, hw=8, cin=cin, seed=1601, gain=5.0 if cin > 2 else 4.0)
  if cin == 2:
    x = np.minimum(rng.poisson(0.3, (4, 3, H, W, 2)), 3).astype(np.uint8)
  else:
    x = (rng.random((4, 3, H, W, 128)) < 0.15).astype(np.uint8)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  eu, es = oracle.conv_block(x, qw, c["bn"], None, "int")
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(H, W, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xt = _t(x, dev)
  xin = xt if cin == 2 else ops.pack_bits(xt)
  x_max = input_max_bound(xin)
  u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                              impl=L.IMPL_MFMA, x_max=x_max)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  if H >= 2 and W >= 2:
    _, sp = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                 pool=2, impl=L.IMPL_MFMA, want_u=False, x_max=x_max)
    np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es)))
  # carry-in: two halves of T
  u1, s1 = ops.conv_lif_forward(xin[:2], geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                impl=L.IMPL_MFMA, x_max=x_max)
  u2, s2 = ops.conv_lif_forward(xin[2:], geom, w, _mslif(), bn=_bn(c["bn"], dev), u0=u1,
                                packed_out=True, impl=L.IMPL_MFMA, x_max=x_max)
  np.testing.assert_array_equal(np.concatenate([_np(s1), _np(s2)]), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u2), eu)


@pytest.mark.parametrize("cin", [128, 64, 2])
@pytest.mark.parametrize("cout", [100, 40, 200])
def test_conv_block_any_output_channel_count(dev, oracle, cin, cout):
  """Cout that is not a multiple of 32 (config.channels = 100, ...): the last spike word
  of a pixel is masked, per-channel parameters clamped, potentials of the channels that do
  not exist neither read nor written."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  hw = 8 if cin > 2 else 16
  c = cases.conv_block_case(T=4, B=3, hw=hw, cin=cin, cout=cout, seed=1801,
                            gain=5.0 if cin > 2 else 4.0)
  if cin == 2:
    c["x"] = np.minimum(c["x"], 1).astype(np.uint8)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(hw, hw, cin, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xt = _t(c["x"], dev)
  xin = xt if cin == 2 else ops.pack_bits(xt)
  for pool, key in ((1, "s_bits"), (2, "pooled_bits")):
    u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                pool=pool, impl=L.IMPL_MFMA, x_max=1)
    np.testing.assert_array_equal(_np(s), e[key])
    np.testing.assert_array_equal(_np(u), e["u"])
  u1, s1 = ops.conv_lif_forward(xin[:2], geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                impl=L.IMPL_MFMA, x_max=1)
  u2, s2 = ops.conv_lif_forward(xin[2:], geom, w, _mslif(), bn=_bn(c["bn"], dev), u0=u1,
                                packed_out=True, impl=L.IMPL_MFMA, x_max=1)
  np.testing.assert_array_equal(np.concatenate([_np(s1), _np(s2)]), e["s_bits"])
  np.testing.assert_array_equal(_np(u2), e["u"])


@pytest.mark.parametrize("cout", [64, 128, 160])
def test_conv_block_64_input_channels(dev, oracle, cout):
  """Blocks of 64 input channels (config.channels = 64) run on the fp6 MFMA kernel too
  (one 64-channel plane, 9 k-steps); codes wider than fp6 holds stay on the direct-form
  kernel.  Bit-exact rasters / pooled rasters / potentials."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=5, B=3, hw=8, cin=64, cout=cout, seed=1701, gain=5.0)
  rng = np.random.Generator(np.random.PCG64(88))
  x = (rng.random((5, 3, 12, 20, 64)) < 0.2).astype(np.uint8)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  eu, es = oracle.conv_block(x, qw, c["bn"], None, "int")
  assert 0.01 < es.mean() < 0.6
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(12, 20, 64, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(x, dev))
  for x_max in (1, 0):
    u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                                impl=L.IMPL_MFMA, x_max=x_max)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(u), eu)
  _, sp = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                               pool=2, impl=L.IMPL_MFMA, want_u=False, x_max=1)
  np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es)))
  # the same codes through the int8 MFMA kernel (what wider codes take): two 32-channel
  # planes per tap at Cin <= 64
  u, s = ops.conv_lif_forward(xin, geom, dataclasses.replace(w, code_max=0), _mslif(),
                              bn=_bn(c["bn"], dev), packed_out=True, impl=L.IMPL_MFMA)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("cin", [3, 32, 48, 65, 96, 100])
def test_conv_block_any_input_channel_count(dev, oracle, cin, bits):
  """Blocks whose input width is not 64 or 128 (config.channels = 100, 96, 48, ...) stay on
  the MFMA kernels: the tiled codes are zero-padded along Cin to 64 / 128 and the kernels
  read the ceil(Cin / 32) spike words a pixel really has.  fp6 kernel (4-bit codes) and
  int8 kernel (8-bit codes), rasters / pooled rasters / potentials bit-exact."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=4, B=3, hw=8, cin=cin, cout=96, seed=1901 + cin, gain=5.0, bits=bits)
  rng = np.random.Generator(np.random.PCG64(cin))
  x = (rng.random((4, 3, 9, 14, cin)) < 0.25).astype(np.uint8)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  eu, es = oracle.conv_block(x, qw, c["bn"], None, "int")
  assert 0.005 < es.mean() < 0.7, es.mean()
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert (w.code_max <= 7) == (bits == 4)
  geom = ops.ConvGeom(9, 14, cin, 96, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(x, dev))
  u, s = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                              impl=L.IMPL_MFMA)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  _, sp = ops.conv_lif_forward(xin, geom, w, _mslif(), bn=_bn(c["bn"], dev), packed_out=True,
                               pool=2, impl=L.IMPL_MFMA, want_u=False)
  np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es)))


def test_mfma_kernel_refuses_unsupported_shapes(dev):
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(hw=8)
  w = _weight(c["leaf"], 4, dev, transposed=True)
  x = ops.pack_bits(_t(c["x"], dev))
  geom = ops.ConvGeom(8, 8, 128, 128, 3, 3, (2, 2), ((1, 1), (1, 1)))
  with pytest.raises(L.SnnqpError) as ei:
    ops.conv_lif_forward(x, geom, w, _mslif(), packed_out=True, impl=L.IMPL_MFMA)
  assert ei.value.code == L.EUNSUPPORTED and "stride" in str(ei.value)


# ---------------------------------------------------------------------------
# whole models through the Module.init / Module.apply surface
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("quantized", [False, True], ids=["c1_fp32", "c2_8bit_50pct"])
def test_dense_snn_model(dev, oracle, golden_dir, quantized):
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.dense_net_case(quantized)
  g = _golden(golden_dir, "dense_net_c2" if quantized else "dense_net_c1")
  cfg = syn.make_config(bits=8, prune_percentage=0.5 if quantized else -1.0,
                        hidden=96)
  model = models.DenseSNN(num_classes=11, config=cfg)
  x = _t(c["x"], dev)
  init_vars = model.init({"params": 0, "dropout": 1}, x, rng=None, trgt=None, train=False)
  want = {"QuantDense_0": ["DuQ_0", "kernel"] + (["prune_0"] if quantized else []),
          "QuantDense_1": ["DuQ_0", "kernel"] + (["prune_0"] if quantized else [])}
  assert {k: sorted(v) for k, v in init_vars["params"].items()} == want
  assert float(init_vars["params"]["QuantDense_0"]["DuQ_0"]["a"]) == -1.0
  variables = nn.tree_from_numpy(c["vars"], dev)
  for inp in (x, x.to(torch.float32)):       # typed uint8 and the reference's float32
    (logits, _), mut = model.apply(variables, inp, trgt=None, train=False, rng=None,
                                   mutable=["intermediates"])
    np.testing.assert_array_equal(_np(logits), g["logits"])
    s2 = mut["intermediates"]["dense2_out"][0]
    s2 = s2.to_dense() if hasattr(s2, "to_dense") else s2
    np.testing.assert_array_equal(_np(s2).astype(np.uint8), g["s2"])
  with pytest.raises(NotImplementedError):
    model.apply(variables, x, trgt=None, train=True, rng=None)


def test_conv_dense_snn_model_tiny(dev, oracle, golden_dir):
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.conv_net_case()
  g = _golden(golden_dir, "conv_net_c3_tiny")
  live = cases.conv_net_expected(oracle, c)
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  model = models.ConvDenseSNN(num_classes=11, config=cfg)
  x = _t(c["x"], dev)
  iv = model.init({"params": 0}, x, rng=None, trgt=None, train=False)
  assert sorted(iv["params"]) == ["BatchNorm_0", "BatchNorm_1", "BatchNorm_2", "QuantConv_0",
                                  "QuantConv_1", "QuantConv_2", "QuantDense_0"]
  assert sorted(iv["batch_stats"]) == ["BatchNorm_0", "BatchNorm_1", "BatchNorm_2"]
  assert sorted(iv["params"]["QuantConv_1"]) == ["DuQ_0", "kernel", "prune_0"]
  assert tuple(iv["params"]["QuantConv_1"]["kernel"].shape) == (3, 3, 128, 128)
  variables = nn.tree_from_numpy(c["vars"], dev)
  for inp in (x, x.to(torch.float32)):
    (logits, _), mut = model.apply(variables, inp, trgt=None, train=False, rng=None,
                                   mutable=["intermediates", "batch_stats"])
    for ref in (g, live):
      np.testing.assert_array_equal(_np(logits), ref["logits"])
      for i in range(3):
        np.testing.assert_array_equal(_np(mut["intermediates"]["pool%d" % i][0]),
                                      ref["pool%d_bits" % i])
      np.testing.assert_array_equal(
          _np(mut["intermediates"]["dense_out"][0].to_dense()).astype(np.uint8), ref["dense_s"])


def test_conv_dense_snn_with_event_counts(dev, oracle):
  """Integer event counts (input_pipeline.py:195-218) into conv0."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.conv_net_case(counts=True)
  assert c["x"].max() > 1
  e = cases.conv_net_expected(oracle, c)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  (logits, _) = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None,
                            train=False, rng=None)
  np.testing.assert_array_equal(_np(logits), e["logits"])


def test_full_cextnet_with_tcja(dev, oracle, golden_dir):
  """The reference's full DVS128 model (5 conv blocks, 2 TCJA gates, 2 dense
  blocks, models.py:31-257) at 64x64 input: spike rasters bit-exact throughout,
  including the layers fed by real-valued (gated) activations; gate values equal
  to the oracle's float64-logistic to the last bit."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.cextnet_case()
  g = _golden(golden_dir, "cextnet_tiny")
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  model = models.CextNet(num_classes=11, config=cfg)
  x = _t(c["x"], dev)
  iv = model.init({"params": 0}, x, rng=None, trgt=None, train=False)
  assert sorted(iv["params"]) == sorted(["QuantConv_%d" % i for i in range(9)] +
                                        ["BatchNorm_%d" % i for i in range(5)] +
                                        ["QuantDense_0", "QuantDense_1"])
  assert tuple(iv["params"]["QuantConv_4"]["kernel"].shape) == (4, 4, 4)        # over T
  assert tuple(iv["params"]["QuantConv_5"]["kernel"].shape) == (4, 128, 128)    # over C
  assert tuple(iv["params"]["QuantDense_0"]["kernel"].shape) == (2 * 2 * 128, 512)
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), x, trgt=None, train=False,
                                 rng=None, mutable=["intermediates"])
  im = mut["intermediates"]
  for i in range(3):
    np.testing.assert_array_equal(_np(im["pool%d" % i][0]), g["pool%d_bits" % i])
  for i in range(2):
    np.testing.assert_array_equal(_np(im["tcja_gate_%d" % i][0]), g["gate%d" % i])
    s = im["conv_t_%d" % i][0]
    s = _np(s) if hasattr(s, "bits") else packbits_lastaxis(_np(s))
    np.testing.assert_array_equal(s, g["conv_t_%d_bits" % i])
  d1 = im["dense1_out"][0]
  d1 = d1.to_dense() if hasattr(d1, "to_dense") else d1
  np.testing.assert_array_equal(_np(d1).astype(np.uint8), g["dense1_s"])
  d2 = im["dense2_out"][0]
  d2 = d2.to_dense() if hasattr(d2, "to_dense") else d2
  np.testing.assert_array_equal(_np(d2).astype(np.uint8), g["dense2_s"])
  np.testing.assert_array_equal(_np(logits), g["logits"])
  assert 0.02 < g["dense1_s"].mean() < 0.5 and 0.02 < g["dense2_s"].mean() < 0.5


def test_full_size_cextnet_against_oracle(dev, oracle):
  """The reference's full DVS128 model at ITS geometry -- 128x128x2 events, T = 20 (the TCJA
  convolutions over T are 20 -> 20), 4-bit / 90 % pruned, random BatchNorm statistics -- against
  the oracle computed here: every raster, both gates and the logits bit-exact."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.cextnet_case(T=20, B=2, hw=128)
  g = cases.cextnet_expected(oracle, c)
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  model = models.CextNet(num_classes=11, config=cfg)
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None,
                                 train=False, rng=None, mutable=["intermediates"])
  im = mut["intermediates"]
  for i in range(3):
    np.testing.assert_array_equal(_np(im["pool%d" % i][0]), g["pool%d_bits" % i])
  for i in range(2):
    np.testing.assert_array_equal(_np(im["tcja_gate_%d" % i][0]), g["gate%d" % i])
    s = im["conv_t_%d" % i][0]
    s = _np(s) if hasattr(s, "bits") else packbits_lastaxis(_np(s))
    np.testing.assert_array_equal(s, g["conv_t_%d_bits" % i])
  for name in ("dense1", "dense2"):
    d = im[name + "_out"][0]
    d = d.to_dense() if hasattr(d, "to_dense") else d
    np.testing.assert_array_equal(_np(d).astype(np.uint8), g[name + "_s"])
  np.testing.assert_array_equal(_np(logits), g["logits"])
  assert 0.02 < g["dense1_s"].mean() < 0.5 and 0.02 < g["dense2_s"].mean() < 0.5


def test_mixed_precision_c5_like_model(dev, oracle):
  """BASELINE config C5 composed from the same blocks: per-layer 2/4-bit weights,
  95 % unstructured prune, 10 classes (read-out 100), odd T."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  lb = [2, 4, 2, 4]
  c = cases.conv_net_case(T=7, B=3, hw=16, p=0.95, layer_bits=lb, out=100,
                          gains=(8.0, 12.0, 14.0, 20.0))
  e = cases.conv_net_expected(oracle, c)
  assert np.all(e["rates"] > 0.005), e["rates"]
  cfg = syn.make_config(bits=4, prune_percentage=0.95)
  cfg.quant.layer_bits = lb
  model = models.ConvDenseSNN(num_classes=10, config=cfg)
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None,
                                 train=False, rng=None, mutable=["intermediates"])
  assert tuple(logits.shape) == (3, 10)
  for i in range(3):
    np.testing.assert_array_equal(_np(mut["intermediates"]["pool%d" % i][0]), e["pool%d_bits" % i])
  np.testing.assert_array_equal(_np(logits), e["logits"])


def test_full_size_c2_against_oracle(dev, oracle):
  """BASELINE config 2 at full size: 2048 -> 512 -> 110, 8-bit, 50 % magnitude
  prune, T = 20, B = 256: both rasters and the logits bit-exact."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.dense_net_case(True, T=20, B=256, K=2048, hidden=512)
  e = cases.dense_net_expected(oracle, c)
  assert 0.01 < e["s1"].mean() < 0.3 and 0.01 < e["s2"].mean() < 0.3
  cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
  model = models.DenseSNN(num_classes=11, config=cfg)
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None,
                                 train=False, rng=None, mutable=["intermediates"])
  np.testing.assert_array_equal(_np(mut["intermediates"]["dense2_out"][0].to_dense()).astype(np.uint8),
                                e["s2"])
  np.testing.assert_array_equal(_np(logits), e["logits"])


def test_prepare_params_masks_and_ac(dev, oracle):
  """Mask + a, c construction (train_inpt_spikingjelly.py:147-223) vs the oracle."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import prune_utils, synthetic as syn
  from snnquantprune_amd.quant import gaussian_init
  v = syn.conv_net_variables(hw=16, prune_p=-1.0)
  params = nn.tree_from_numpy(v["params"], dev)
  names = [k for k in params if k.startswith("Quant")]
  kernels = [v["params"][k]["kernel"] for k in names]
  loc = prune_utils.update_prune_mask(params, 0.75)
  for k in names:
    np.testing.assert_array_equal(_np(loc[k]["prune_0"]["mask"]),
                                  oracle.local_prune_mask(v["params"][k]["kernel"], 0.75))
  glob = prune_utils.update_global_prune_mask(params, 0.6)
  for k, m in zip(names, oracle.global_prune_masks(kernels, 0.6)):
    np.testing.assert_array_equal(_np(glob[k]["prune_0"]["mask"]), m)
  q = prune_utils.update_quant_params(params, gaussian_init, 4)
  for k in names:
    a = float(q[k]["DuQ_0"]["a"])
    assert q[k]["DuQ_0"]["a"].shape == (1,)
    np.testing.assert_array_equal(np.float32(a), np.float32(oracle.gaussian_init(v["params"][k]["kernel"], 4)))
    assert float(q[k]["DuQ_0"]["c"]) == a
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  cfg.quant.prune_global = True
  cfg.quant.start_epoch = -1
  full = prune_utils.prepare_params(params, cfg)
  tot = sum(int(np.prod(k.shape)) for k in kernels)
  kept = sum(float(full[k]["prune_0"]["mask"].sum()) for k in names)
  assert kept == tot - int(tot * 0.9)
  assert "BatchNorm_0" in full and "scale" in full["BatchNorm_0"]


@pytest.mark.parametrize("bits,prune,counts", [(4, 0.9, False), (8, 0.3, False), (8, 0.3, True)])
def test_full_size_c3_layers_against_oracle(dev, oracle, bits, prune, counts):
  """BASELINE config 3 geometry (128x128x2 input, 128 channels, 32768 -> 110
  read-out, T = 20) at B = 1: every pooled raster and the logits bit-exact -- on the
  headline 4-bit / 90 % pruned weights (fp6 instruction) and on the reference's shipped
  8-bit / 30 % pruned configuration (int8 instruction, arithmetic dequantisation), the
  latter also on event-count frames; plus the size-independent property that samples
  are independent (a batch equals its samples run one by one)."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  gains = (4.0, 5.0, 4.0, 4.0) if bits == 4 else (3.0, 2.5, 2.5, 3.0)
  c = cases.conv_net_case(T=20, B=1, hw=128, bits=bits, p=prune, random_bn=False, counts=counts,
                          gains=gains)
  e = cases.conv_net_expected(oracle, c)
  assert np.all(e["rates"] > 0.004) and np.all(e["rates"] < 0.5), e["rates"]
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=bits, prune_percentage=prune))
  variables = nn.tree_from_numpy(c["vars"], dev)
  (logits, _), mut = model.apply(variables, _t(c["x"], dev), trgt=None, train=False, rng=None,
                                 mutable=["intermediates"])
  for i in range(3):
    np.testing.assert_array_equal(_np(mut["intermediates"]["pool%d" % i][0]),
                                  e["pool%d_bits" % i])
  np.testing.assert_array_equal(_np(logits), e["logits"])
  xb = (syn.poisson_counts if counts else syn.poisson_spikes)((3, 20, 128, 128, 2), 0.1, seed=77)
  (lb, _) = model.apply(variables, _t(xb, dev), trgt=None, train=False, rng=None)
  for i in range(3):
    (li, _) = model.apply(variables, _t(xb[i:i + 1], dev), trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(lb)[i:i + 1], _np(li))


@pytest.mark.parametrize("bits,prune", [(4, 0.9), (8, 0.3)])
def test_full_batch_c3_properties(dev, oracle, bits, prune):
  """BASELINE config 3 at the bench size (B = 1024, T = 20, 128x128x2), through
  size-independent properties: a sample's logits do not depend on the batch it is in
  (persistent patch schedule, XCD split, every table mode), permuting the batch
  permutes the logits, and a repeated run is bit-identical."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  B, T = 1024, 20
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=bits, prune_percentage=prune))
  variables = nn.tree_from_numpy(syn.conv_net_variables(prune_p=prune), dev)
  gen = torch.Generator(device=dev)
  gen.manual_seed(20261003)
  x = (torch.rand((B, T, 128, 128, 2), device=dev, generator=gen) < 0.095).to(torch.uint8)
  (full, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  full = _np(full)
  assert full.shape == (B, 11) and np.isfinite(full).all() and full.std() > 0
  (again, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(again), full)
  for b in (0, 7, 513, 1023):
    (one, _) = model.apply(variables, x[b:b + 1], trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(one), full[b:b + 1])
  perm = torch.randperm(B, device=dev, generator=gen)
  (pl, _) = model.apply(variables, x[perm].contiguous(), trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(pl), full[perm.cpu().numpy()])
  # a sub-batch that is not a multiple of 8 samples (uneven XCD shares)
  (sub, _) = model.apply(variables, x[100:137].contiguous(), trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(sub), full[100:137])


@pytest.mark.parametrize("random_bn", [False, True], ids=["bn_default", "bn_random"])
def test_full_size_c5_against_oracle(dev, oracle, random_bn):
  """BASELINE config 5 at its own geometry: 128x128x2 input, T = 50 (conv0 stages its halo in
  chunks of fewer timesteps than T, the spike words are flushed several times per patch),
  per-layer 2/4/2/4-bit weights (flax_qconv.py:89; 2-bit DuQ codes are {-1, 0, 1}, L = 1,
  quant.py:458-461), 95 % unstructured prune, 10 classes (read-out 100): every pooled raster,
  the read-out raster and the logits bit-exact against the int oracle, firing rates in band."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  lb = [2, 4, 2, 4]
  c = cases.conv_net_case(T=50, B=1, hw=128, p=0.95, layer_bits=lb, out=100, random_bn=random_bn,
                          gains=(4.0, 5.0, 4.0, 4.0))
  e = cases.conv_net_expected(oracle, c)
  assert np.all(e["rates"] > 0.01) and np.all(e["rates"] < 0.5), e["rates"]
  cfg = syn.make_config(bits=4, prune_percentage=0.95)
  cfg.quant.layer_bits = lb
  model = models.ConvDenseSNN(num_classes=10, config=cfg)
  variables = nn.tree_from_numpy(c["vars"], dev)
  for inp in (_t(c["x"], dev), _t(c["x"], dev).to(torch.float32)):
    (logits, _), mut = model.apply(variables, inp, trgt=None, train=False, rng=None,
                                   mutable=["intermediates"])
    assert tuple(logits.shape) == (1, 10)
    for i in range(3):
      np.testing.assert_array_equal(_np(mut["intermediates"]["pool%d" % i][0]),
                                    e["pool%d_bits" % i])
    np.testing.assert_array_equal(
        _np(mut["intermediates"]["dense_out"][0].to_dense()).astype(np.uint8), e["dense_s"])
    np.testing.assert_array_equal(_np(logits), e["logits"])


def test_full_batch_c5_properties(dev, oracle):
  """BASELINE config 5 at its per-GPU bench size (B = 4096 / 8 = 512, T = 50, mixed 2/4-bit,
  95 % pruned) through size-independent properties: batch independence (incl. a sample the
  oracle also computes), permutation equivariance, repeatability."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  from tests.helpers import bn_of, qweight_of
  B, T, lb = 512, 50, [2, 4, 2, 4]
  cfg = syn.make_config(bits=4, prune_percentage=0.95)
  cfg.quant.layer_bits = lb
  model = models.ConvDenseSNN(num_classes=10, config=cfg)
  v = syn.conv_net_variables(prune_p=0.95, out=100)
  variables = nn.tree_from_numpy(v, dev)
  gen = torch.Generator(device=dev)
  gen.manual_seed(20261004)
  x = (torch.rand((B, T, 128, 128, 2), device=dev, generator=gen) < 0.095).to(torch.uint8)
  (full, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  full = _np(full)
  assert full.shape == (B, 10) and np.isfinite(full).all() and full.std() > 0
  (again, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(again), full)
  for b in (0, 5, 300, 511):
    (one, _) = model.apply(variables, x[b:b + 1], trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(one), full[b:b + 1])
  perm = torch.randperm(B, device=dev, generator=gen)
  (pl, _) = model.apply(variables, x[perm].contiguous(), trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(pl), full[perm.cpu().numpy()])
  # one sample of the batch through the oracle as well
  p = v["params"]
  r = oracle.conv3_dense_forward(
      _np(x[300:301]), [qweight_of(oracle, p["QuantConv_%d" % i], lb[i]) for i in range(3)],
      [bn_of(v, i) for i in range(3)], qweight_of(oracle, p["QuantDense_0"], lb[3]), mode="int")
  np.testing.assert_array_equal(full[300:301], r["logits"])


@pytest.mark.parametrize("prepare", [False, True], ids=["as_saved", "prune_quant_joint"])
def test_flax_checkpoint_through_the_kernels(dev, oracle, golden_dir, prepare):
  """F2 end to end: the independent-bytes Flax checkpoint (tests/golden/make_flax_blob.py) ->
  load_flax_checkpoint -> [prepare_params as configs/prune_quant_joint.py does: global
  magnitude mask, a = c = gaussian_init] -> CextNet.apply, against the oracle fed the same
  tree (tcja_load_pretrained_weights.py:39-167, train_inpt_spikingjelly.py:159-223,
  train_utils.py:30-41).  as_saved keeps the checkpoint's learnt a != c and no pruning."""
  from snnquantprune_amd import checkpoint, linen as nn
  from snnquantprune_amd import models, prune_utils, synthetic as syn
  from snnquantprune_amd.quant import gaussian_init
  path = os.path.join(golden_dir, "flax_checkpoint_tiny.msgpack")
  tree = checkpoint.load_flax_checkpoint(path)
  cfg = syn.make_config(bits=4, prune_percentage=0.6 if prepare else -1.0, channels=32)
  cfg.quant.prune_global = True
  cfg.quant.start_epoch = -1
  cfg.quant.init_fn = gaussian_init
  variables = nn.tree_from_numpy(tree, dev)
  np_tree = {"params": {k: dict(v) for k, v in tree["params"].items()},
             "batch_stats": tree["batch_stats"]}
  if prepare:
    variables = dict(variables, params=prune_utils.prepare_params(variables["params"], cfg))
    names = [k for k in tree["params"] if k.startswith("Quant")]
    masks = oracle.global_prune_masks([tree["params"][k]["kernel"] for k in names], 0.6)
    for k, m in zip(names, masks):
      a = np.array([oracle.gaussian_init(tree["params"][k]["kernel"], 4)], F32)
      np_tree["params"][k] = dict(tree["params"][k], prune_0={"mask": m}, DuQ_0={"a": a, "c": a})
      np.testing.assert_array_equal(_np(variables["params"][k]["prune_0"]["mask"]), m)
      np.testing.assert_array_equal(_np(variables["params"][k]["DuQ_0"]["a"]), a)
  x = syn.poisson_spikes((2, 4, 32, 32, 2), 0.15, seed=424242)
  e = cases.cextnet_expected(oracle, {"vars": np_tree, "x": x, "bits": 4})
  model = models.CextNet(num_classes=11, config=cfg)
  (logits, _), mut = model.apply(variables, _t(x, dev), trgt=None, train=False, rng=None,
                                 mutable=["intermediates"])
  im = mut["intermediates"]
  for i in range(3):
    np.testing.assert_array_equal(_np(im["pool%d" % i][0]), e["pool%d_bits" % i])
  for i in range(2):
    np.testing.assert_array_equal(_np(im["tcja_gate_%d" % i][0]), e["gate%d" % i])
  d2 = im["dense2_out"][0]
  d2 = d2.to_dense() if hasattr(d2, "to_dense") else d2
  np.testing.assert_array_equal(_np(d2).astype(np.uint8), e["dense2_s"])
  np.testing.assert_array_equal(_np(logits), e["logits"])
  assert e["dense2_s"].mean() > 0.005, e["dense2_s"].mean()


def test_density_probes_and_workload_tables(dev, oracle, tmp_path):
  """F4: with config.density_probes the models sow the reference's probe names
  (examples/tcja/models.py:45-91,128-142; `<name>_min` is the MAXIMUM of the per-slice
  densities, as there) from snnqp_density on the packed rasters; checked against the
  oracle's densities of the same tensors, with the logits unchanged by probing; then the
  weight densities (sparsity.py:109-122) and both workload files."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, sparsity, synthetic as syn
  c = cases.cextnet_case()
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  cfg.density_probes = True
  model = models.CextNet(num_classes=11, config=cfg)
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  (plain, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  (logits, _), mut = model.apply(variables, x, trgt=None, train=False, rng=None,
                                 mutable=["intermediates"])
  np.testing.assert_array_equal(_np(logits), _np(plain))           # probing changes nothing
  p, b = c["vars"]["params"], c["bits"]
  probes = {}
  oracle.cextnet_forward(
      c["x"], [qweight_of(oracle, p["QuantConv_%d" % i], b) for i in (0, 1, 2, 3, 6)],
      [bn_of(c["vars"], i) for i in range(5)],
      [(qweight_of(oracle, p["QuantConv_4"], b), qweight_of(oracle, p["QuantConv_5"], b)),
       (qweight_of(oracle, p["QuantConv_7"], b), qweight_of(oracle, p["QuantConv_8"], b))],
      [qweight_of(oracle, p["QuantDense_0"], b), qweight_of(oracle, p["QuantDense_1"], b)],
      probes=probes)
  im = mut["intermediates"]
  assert len(probes) == 22
  for name, d in probes.items():
    got_max, got_mean = float(im[name + "_min"][0]), float(im[name + "_mean"][0])
    assert abs(got_max - float(d.max())) <= 1e-7 * max(1.0, float(d.max())), name
    assert abs(got_mean - float(np.mean(d, dtype=np.float64))) <= 2e-7, name
  assert 0.0 < float(im["conv_1_out_mean"][0]) < 0.5 and float(im["conv_0_inpt_min"][0]) < 0.2
  # exact non-zero counts of a uint8 tensor (new input type of snnqp_density)
  from snnquantprune_amd import ops
  cnt = _np(ops.density(x, lead_dims=2, counts=True))
  np.testing.assert_array_equal(cnt, (c["x"] != 0).reshape(c["x"].shape[:2] + (-1,)).sum(-1))
  # weight densities: kernel * mask through DuQ, fraction of non-zeros
  ls = sparsity.weight_density(variables["params"], cfg)
  for name in ("QuantConv_1", "QuantConv_4", "QuantDense_0"):
    q = qweight_of(oracle, p[name], b)
    assert ls[name] == float(np.count_nonzero(q.w_fq) / q.w_fq.size), name
  acc = sparsity.ProbeAccumulator()
  acc.append(im)
  (_, _), mut2 = model.apply(variables, _t(c["x"][::-1].copy(), dev), trgt=None, train=False,
                             rng=None, mutable=["intermediates"])
  acc.append(mut2["intermediates"])
  st = acc.stacked()
  assert st["dense2_out_mean"].shape == (2,)
  paths = sparsity.write_workload(str(tmp_path / "workload_tiny"), ls, st, frames=4, channels=128,
                                  hw=(64, 64))
  lines = open(paths[0]).read().splitlines()
  assert lines[0] == "name,weights,inputs,outputs,T,C,M,P,Q,R,S,HS,WS" and len(lines) == 12
  assert lines[1].startswith("Conv1,%s," % str(ls["QuantConv_0"]))
  assert lines[1].endswith(",4,2,128,64,64,3,3,1,1") and lines[11].endswith(",4,512,110,1,1,1,1,1,1")
  # the C3 model sows its four layers under the same names
  c3 = cases.conv_net_case()
  cfg3 = syn.make_config(bits=4, prune_percentage=0.9)
  cfg3.density_probes = True
  m3 = models.ConvDenseSNN(num_classes=11, config=cfg3)
  (l3, _), mut3 = m3.apply(nn.tree_from_numpy(c3["vars"], dev), _t(c3["x"], dev), trgt=None,
                           train=False, rng=None, mutable=["intermediates"])
  e3 = cases.conv_net_expected(oracle, c3)
  np.testing.assert_array_equal(_np(l3), e3["logits"])
  p3 = c3["vars"]["params"]
  r3 = oracle.conv3_dense_forward(
      c3["x"], [qweight_of(oracle, p3["QuantConv_%d" % i], 4) for i in range(3)],
      [bn_of(c3["vars"], i) for i in range(3)], qweight_of(oracle, p3["QuantDense_0"], 4),
      mode="int", keep=True)
  for i in range(3):
    d = oracle.density(r3["conv%d_s" % i])
    assert abs(float(mut3["intermediates"]["conv_%d_out_min" % i][0]) - float(d.max())) < 1e-7
    np.testing.assert_array_equal(_np(mut3["intermediates"]["pool%d" % i][0]), e3["pool%d_bits" % i])
  d = oracle.density(r3["dense_s"])
  assert abs(float(mut3["intermediates"]["dense1_out_mean"][0]) - float(d.mean())) < 2e-7


def test_work_queue_slots_with_many_launches_in_flight_and_graph_capture(dev, oracle):
  """The fused conv kernels claim patches from per-launch work-queue slots (64 per device,
  snnqp.h).  96 launches enqueued on 8 streams without a synchronisation in between (more
  than there are slots: the launches that find their slot busy walk statically), both conv
  kernels, must all produce the result of a lone launch; and a launch captured into a HIP
  graph (a queue slot of its own, which every launch leaves zeroed for the next replay) replays
  bit-identically."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  outs = {}
  jobs = []
  for name, c in (("bits", cases.conv_block_case(T=6, B=16, hw=16)),
                  ("u8c2", cases.conv_block_case(T=6, B=16, hw=16, cin=2, seed=961, gain=4.0))):
    e = cases.conv_block_expected(oracle, c)
    w = _weight(c["leaf"], c["bits"], dev, transposed=True)
    bn = _bn(c["bn"], dev)
    nrn = _mslif()
    g = ops.ConvGeom(16, 16, c["x"].shape[-1], 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
    x = _t(c["x"], dev)
    xin = ops.pack_bits(x) if name == "bits" else x
    x_max = 1 if name == "bits" else input_max_bound(x)
    jobs.append((name, xin, g, w, nrn, bn, x_max, e))
  torch.cuda.synchronize()
  streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
  results = []
  for rep in range(6):
    for st in streams:
      for (name, xin, g, w, nrn, bn, x_max, e) in jobs:
        with torch.cuda.stream(st):
          _, s = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                      impl=L.IMPL_MFMA, x_max=x_max)
        results.append((name, s, e))
  torch.cuda.synchronize()
  assert len(results) == 96
  for name, s, e in results:
    np.testing.assert_array_equal(_np(s), e["pooled_bits"], err_msg=name)
  # graph capture: the launch inside the capture takes one of the capture-only slots
  name, xin, g, w, nrn, bn, x_max, e = jobs[0]
  out = torch.zeros_like(results[0][1].bits)
  graph = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream(device=dev)
  with torch.cuda.stream(cap):
    ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                         impl=L.IMPL_MFMA, x_max=x_max)          # warm (allocations) outside capture
    cap.synchronize()
    with torch.cuda.graph(graph, stream=cap):
      _, s = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                  impl=L.IMPL_MFMA, x_max=x_max)
      out.copy_(s.bits)
  for _ in range(3):
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(out).view(np.uint32), e["pooled_bits"])


def test_event_layer_checks_its_input_instead_of_trusting_the_hint(dev, oracle):
  """conv0 (uint8 event frames, Cin = 2): `x_max` only says what the tables are sized for.
  Binary frames with one hot pixel (count 200) and real count frames, launched with the
  hint 1 (per-channel tables), with a hint in the shared-table range and with an exact one:
  every result equals the oracle's (the chunks that exceed the hint run the general path),
  and x_seen reports the largest value met.  Then through the model: the adaptive hint
  (ops.count_hint) never makes the host wait and never changes a result."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.conv_block_case(T=9, B=5, hw=24, cin=2, seed=961, gain=4.0)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(24, 24, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  rng = np.random.Generator(np.random.PCG64(5))
  binary = (rng.random((9, 5, 24, 24, 2)) < 0.1).astype(np.uint8)
  hot = binary.copy()
  hot[3, 2, 7, 11, 1] = 200                                   # one hot pixel in one patch
  counts = np.minimum(rng.poisson(0.4, (9, 5, 24, 24, 2)), 255).astype(np.uint8)
  for name, x in (("binary", binary), ("hot", hot), ("counts", counts)):
    cc = dict(c, x=x)
    e = cases.conv_block_expected(oracle, cc)
    for hint in (1, 4, int(x.max())):
      seen = torch.zeros(8, dtype=torch.int32, device=dev)
      _, s = ops.conv_lif_forward(_t(x, dev), g, w, nrn, bn=bn, want_u=False, packed_out=True,
                                  pool=2, impl=L.IMPL_MFMA, x_max=hint, x_seen=seen)
      np.testing.assert_array_equal(_np(s), e["pooled_bits"], err_msg="%s hint %d" % (name, hint))
      st = seen.cpu().numpy()
      assert int(st[0]) == int(x.max()), (name, hint)
      # every staged chunk (5 samples x 3 x 3 patches, one chunk of 9 timesteps each) is counted
      # once, by its largest value; the hot pixel (200) sits in the halo of few of them
      assert int(st[1:6].sum()) == 5 * 9 and int(st[7]) == 0 and int(st[6]) <= int(st[3]), st   # [6]: exactly 3
      if name == "hot":
        assert 1 <= int(st[5]) <= 4 and int(st[1]) == 5 * 9 - int(st[5]), st
  # the model path: hints adapt from what the kernel reports, results never change
  cm = cases.conv_net_case(counts=True)
  em = cases.conv_net_expected(oracle, cm)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(cm["vars"], dev)
  hint = ops.count_hint(dev)
  for i in range(4):
    (logits, _) = model.apply(variables, _t(cm["x"], dev), trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(logits), em["logits"])
    torch.cuda.synchronize()
  # learnt without a blocking read: the bound of the bucket that holds most chunks' maxima
  # (never above the largest value seen)
  assert hint.max_seen == int(cm["x"].max()) > 1 and 1 < hint.current() <= hint.max_seen
  cb = cases.conv_net_case()
  eb = cases.conv_net_expected(oracle, cb)
  vb = nn.tree_from_numpy(cb["vars"], dev)
  for i in range(3):                                         # back to binary frames
    (logits, _) = model.apply(vb, _t(cb["x"], dev), trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(logits), eb["logits"])
    torch.cuda.synchronize()
  assert hint.current() == 1


def test_random_blocks_against_the_oracle(dev, oracle):
  """Randomised geometries against the ORACLE (tests/stress.py compares kernels with each
  other; this one compares them with the restatement of the reference): 28 fused conv blocks
  -- image sizes that clip patches, any Cin / Cout, event counts or spikes, 2..8-bit codes,
  random BatchNorm (incl. zero mean and bias: the multiply-only form), every neuron form,
  carried-in potentials, with and without the fused pool -- and 16 dense blocks; potentials
  and rasters bit-exact."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(20261004))
  kinds = ("ms2", "ms4", "ms3", "plif", "lif", "vr")

  def neuron(kind, n):
    if kind == "plif":
      tp = F32(-0.35)
      return (ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, float(oracle.sigmoid_f32(tp)), 1.0, 0.0),
              {"kind": "parametric_leaky_IF", "tau_param": tp})
    if kind == "lif":
      tv = rng.uniform(-1, 2, n).astype(F32)
      return (ops.Neuron(L.NEURON_LIF, 1.0, 1.0, 0.0, decay=_t(oracle.sigmoid_f32(tv), dev)),
              {"kind": "LIF", "tau_vec": tv})
    tau = {"ms2": 2.0, "ms4": 4.0, "ms3": 3.0, "vr": 2.0}[kind]
    vr = 0.1 if kind == "vr" else 0.0
    return (ops.Neuron(L.NEURON_MULTI_STEP_LIF, tau, 1.0, vr),
            {"kind": "multi_step_LIF", "tau": tau, "v_reset": vr})

  for it in range(28):
    first = it % 3 == 0
    cin = 2 if first else int(rng.integers(3, 129))
    cout = int(rng.choice([32, 64, 100, 128, 160]))
    H, W = int(rng.integers(3, 22)), int(rng.integers(3, 22))
    T, B = int(rng.integers(1, 8)), int(rng.integers(1, 4))
    bits = int(rng.choice([2, 3, 4, 8]))
    pool = int(rng.choice([1, 2]))
    if pool == 2:
      H, W = H + (H & 1), W + (W & 1)
    leaf = syn_leaf = None
    from snnquantprune_amd import synthetic as syn
    leaf = syn.quant_leaf((3, 3, cin, cout), float(rng.uniform(3, 7)), int(rng.integers(1 << 30)), True,
                          float(rng.choice([0.0, 0.5, 0.9])))
    zero_bn = rng.random() < 0.3
    bn = dict(mean=np.zeros(cout, F32) if zero_bn else rng.normal(0, 0.2, cout).astype(F32),
              var=rng.uniform(0.5, 1.5, cout).astype(F32), scale=rng.uniform(0.5, 1.5, cout).astype(F32),
              bias=np.zeros(cout, F32) if zero_bn else rng.normal(0, 0.2, cout).astype(F32))
    if first:
      x = np.minimum(rng.poisson(float(rng.choice([0.15, 0.6])), (T, B, H, W, 2)), 255).astype(np.uint8)
    else:
      x = (rng.random((T, B, H, W, cin)) < 0.2).astype(np.uint8)
    kind = kinds[it % len(kinds)]
    nrn, ncfg = neuron(kind, cout)
    u0 = (rng.normal(0, 0.3, (B, H, W, cout)).astype(F32)) if rng.random() < 0.3 else None
    qw = qweight_of(oracle, leaf, bits)
    ue, se = oracle.conv_block(x, qw, bn, ncfg, "int", u0=u0)
    if pool == 2:
      se = oracle.max_pool_2x2(se)
    w = _weight(leaf, bits, dev, transposed=True)
    mean, mul, bias = oracle.bn_coeffs(bn["mean"], bn["var"], bn["scale"], bn["bias"])
    flags = (L.BN_MEAN_ZERO | L.BN_BIAS_ZERO) if zero_bn else 0
    bnc = ops.BnCoeffs(_t(mean, dev), _t(mul, dev), _t(bias, dev), flags)
    g = ops.ConvGeom(H, W, cin, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
    xin = _t(x, dev) if first else ops.pack_bits(_t(x, dev))
    tag = "it %d cin %d cout %d %dx%d T %d B %d bits %d pool %d %s bn0 %s u0 %s" % (
        it, cin, cout, H, W, T, B, bits, pool, kind, zero_bn, u0 is not None)
    u, s = ops.conv_lif_forward(xin, g, w, nrn, bn=bnc, u0=None if u0 is None else _t(u0, dev),
                                packed_out=True, pool=pool, impl=L.IMPL_MFMA,
                                x_max=int(x.max()) if first else 1)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(se), err_msg=tag)
    np.testing.assert_array_equal(_np(u), ue, err_msg=tag)
  for it in range(16):
    K = int(rng.choice([rng.integers(1, 200), rng.integers(200, 3000), 512, 784]))
    N = int(rng.choice([rng.integers(1, 40), 110, 100, rng.integers(100, 300)]))
    T, B = int(rng.integers(1, 30)), int(rng.integers(1, 9))
    bits = int(rng.choice([2, 3, 4, 8]))
    from snnquantprune_amd import synthetic as syn
    leaf = syn.quant_leaf((K, N), float(rng.uniform(2, 8)), int(rng.integers(1 << 30)), True,
                          float(rng.choice([0.0, 0.5, 0.9])))
    x = (rng.random((T, B, K)) < rng.uniform(0.02, 0.4)).astype(np.uint8)
    kind = kinds[it % len(kinds)]
    nrn, ncfg = neuron(kind, N)
    u0 = (rng.normal(0, 0.3, (B, N)).astype(F32)) if rng.random() < 0.3 else None
    ue, se = oracle.dense_block(x, qweight_of(oracle, leaf, bits), ncfg, "int", u0=u0)
    w = _weight(leaf, bits, dev, transposed=True)
    u, s = ops.dense_lif_forward(ops.pack_bits(_t(x, dev)), w, K, N, nrn,
                                 u0=None if u0 is None else _t(u0, dev), packed_out=True,
                                 impl=L.IMPL_MFMA)
    tag = "dense it %d K %d N %d T %d B %d bits %d %s" % (it, K, N, T, B, bits, kind)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(se), err_msg=tag)
    np.testing.assert_array_equal(_np(u), ue, err_msg=tag)


def test_eval_step_metrics(dev, oracle):
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn, train_utils
  c = cases.dense_net_case(True)
  cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=96)
  model = train_utils.create_model(model_cls=models.DenseSNN, num_classes=11, config=cfg)
  variables = nn.tree_from_numpy(c["vars"], dev)
  state = train_utils.EvalState(model.apply, {"params": variables["params"]},
                                variables["batch_stats"])
  labels = np.array([3, 1, 0, 7])
  m = train_utils.eval_step(state, {"dvs_matrix": _t(c["x"], dev), "label": _t(labels, dev)},
                            None, 0.0, partial(train_utils.mse_loss, T=1))
  e = cases.dense_net_expected(oracle, c)
  em = oracle.compute_metrics(e["logits"], labels)
  assert abs(float(m["loss"]) - float(em["loss"])) < 1e-7
  np.testing.assert_array_equal(_np(m["accuracy"]), em["accuracy"])


def test_event_front_end_and_density_probes(dev, oracle):
  """F3 / F4: event -> frame histogram (input_pipeline.py:142-219) and the density
  probes (models.py:128-142), exact integer results."""
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(31))
  n, T, H, W = 100003, 20, 128, 128
  x = rng.integers(0, W, n)
  y = rng.integers(0, H, n)
  p = rng.integers(0, 2, n)
  x[:500] = 7; y[:500] = 9; p[:500] = 1               # a hot pixel: counts > 255 saturate in u8
  e = oracle.events_to_frames(x, y, p, T, H, W)
  c = ops.events_to_frames(_t(x, dev), _t(y, dev), _t(p, dev), T, H, W, as_u8=False)
  np.testing.assert_array_equal(_np(c), e)
  u8 = ops.events_to_frames(_t(x, dev), _t(y, dev), _t(p, dev), T, H, W)
  np.testing.assert_array_equal(_np(u8), np.minimum(e, 255).astype(np.uint8))
  assert e.sum() == n and e[0, 9, 7, 1] >= 500
  e2 = oracle.events_to_frames(x, y, p, T, 64, 64, scale=2.0)
  c2 = ops.events_to_frames(_t(x, dev), _t(y, dev), _t(p, dev), T, 64, 64, scale=2.0, as_u8=False)
  np.testing.assert_array_equal(_np(c2), e2)
  s = (rng.random((5, 3, 6, 6, 70)) < 0.2).astype(F32)
  nnz = (s != 0).reshape(5, 3, -1).sum(-1)
  for inp in (_t(s, dev), ops.pack_bits(_t(s, dev))):
    np.testing.assert_array_equal(_np(ops.density(inp, counts=True)), nnz)     # exact counts
    np.testing.assert_allclose(_np(ops.density(inp)), oracle.density(s), rtol=2e-7)


def test_conv_block_random_shapes(dev):
  """40 random geometries (image sizes 3..40, Cin 2..128, Cout 32..300, T, B, pool, 3/4/5/8-bit
  codes, binary / count / large-count events, carried-in potentials): the MFMA kernels --
  work-queue schedule, clipped edge patches, masked channel words, padded Cin, fp6 and int8
  formats, every conv0 table mode -- equal the direct-form kernel bit for bit.
  A PROPERTY test (two implementations of this package agree), not parity evidence: the oracle
  comparisons of the same shapes are test_random_blocks_against_the_oracle and the layer tests."""
  from tests.stress import conv_block_random
  assert conv_block_random(dev, 40, 20261004) == []


def test_dense_block_random_shapes(dev):
  """40 random dense blocks (any K and N, T up to 60, every neuron form, carried-in potentials):
  the MFMA kernels (128-column, wide, fp6) equal the direct-form kernel bit for bit.  A PROPERTY
  test (two implementations of this package agree), not parity evidence."""
  from tests.stress import dense_block_random
  assert dense_block_random(dev, 40, 20261005) == []


@pytest.mark.parametrize("bits", [4, 5, 8])
@pytest.mark.parametrize("tiny", [False, True])
def test_conv_bits_kernel_fused_membrane_update(dev, oracle, bits, tiny):
  """The bits kernel runs u + (x - u) / tau as one fused multiply-add when
  min_current_bits (snnqp_current_min over BatchNorm of every table entry) proves it exact:
  identical rasters and potentials with and without the hint, on fp6 (4-bit) and int8
  codes (5-bit: table; 8-bit: arithmetic dequantisation); BatchNorm that scales the
  currents to ~2^-126 must be refused."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=6, B=3, hw=8, cin=128, cout=128, bits=bits, seed=2101, gain=5.0)
  if tiny:
    z, one = np.zeros(128, F32), np.ones(128, F32)
    c["bn"] = dict(mean=z, var=one, scale=(one * F32(2.0 ** -126)).astype(F32), bias=z, eps=0.0)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn = _bn(c["bn"], dev)
  if tiny:
    bn = ops.BnCoeffs(_t(c["bn"]["mean"], dev), _t(c["bn"]["scale"], dev), _t(c["bn"]["bias"], dev))
  geom = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  mb = ops.current_min_bits(w, bn, int(w.abs_sum_max), 128)
  x_min = np.array([mb], np.uint32).view(F32)[0]
  assert (x_min < 2.0 ** -100) == tiny, x_min
  wf = dataclasses.replace(w, min_current_bits=mb)
  for pool, key in ((2, "pooled_bits"), (1, "s_bits")):
    for ww in (w, wf):
      u, s = ops.conv_lif_forward(xin, geom, ww, _mslif(), bn=bn, packed_out=True, pool=pool,
                                  impl=L.IMPL_MFMA, x_max=1)
      np.testing.assert_array_equal(_np(s), e[key])
      np.testing.assert_array_equal(_np(u), e["u"])


@pytest.mark.parametrize("cin,table", [(32, True), (64, False), (128, False)])
def test_conv_bits_kernel_dequantises_through_the_accumulator_addressed_table(dev, oracle, cin, table):
  """fp6 kernel, |acc| <= abs_sum_max <= 2047: the current is read from an LDS table at the
  address the accumulator's (denormal) bit pattern spells (conv3x3_bits.hip, DQ_TABLE); larger
  bounds keep the three-instruction form.  Codes of +7 / -7 over whole output channels and
  frames of all ones drive the accumulator to both ends of the table (+-288 * 7 at Cin = 32)
  and through every partial sum on the way; the other channels and frames are random.
  Rasters, pooled rasters and potentials bit-exact, with the fused and the two-rounding update."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops, synthetic as syn
  rng = np.random.Generator(np.random.PCG64(7700 + cin))
  T, B, H, W, cout = 6, 2, 8, 12, 64
  a = F32(0.5)
  k = (rng.integers(-7, 8, size=(3, 3, cin, cout)) * (a / F32(7))).astype(F32)
  k[..., 0] = a; k[..., 1] = -a; k[..., 2] = a; k[:, :, ::2, 2] = -a       # codes +7, -7, alternating
  mask = (rng.random((3, 3, cin, cout)) < (0.9 if cin == 32 else 0.6)).astype(F32)
  mask[..., :3] = 1
  leaf = {"kernel": k, "DuQ_0": {"a": np.array([a], F32), "c": np.array([0.37], F32)},
          "prune_0": {"mask": mask}}
  x = (rng.random((T, B, H, W, cin)) < 0.3).astype(np.uint8)
  x[1] = 1; x[4, 0] = 1; x[5, 1, :, :, ::2] = 1
  bp, bs = syn.bn_leaf(cout, True, 7701)
  bn = dict(mean=bs["mean"], var=bs["var"], scale=bp["scale"], bias=bp["bias"])
  qw = qweight_of(oracle, leaf, 4)
  eu, es = oracle.conv_block(x, qw, bn, None, "int")
  assert 0.02 < es.mean() < 0.9, es.mean()
  w = _weight(leaf, 4, dev, transposed=True)
  assert w.code_max == 7 and (int(w.abs_sum_max) <= 2047) == table, w.abs_sum_max
  assert ops.conv_dequant_form(w, _mslif()) == ("table" if table else "arith")
  if cin == 32:
    assert int(w.abs_sum_max) == 288 * 7
  bnc = _bn(bn, dev)
  geom = ops.ConvGeom(H, W, cin, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(x, dev))
  mb = ops.current_min_bits(w, bnc, int(w.abs_sum_max), cout)
  for ww in (w, dataclasses.replace(w, min_current_bits=mb)):
    u, s = ops.conv_lif_forward(xin, geom, ww, _mslif(), bn=bnc, packed_out=True, impl=L.IMPL_MFMA)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(u), eu)
    _, sp = ops.conv_lif_forward(xin, geom, ww, _mslif(), bn=bnc, packed_out=True, pool=2,
                                 impl=L.IMPL_MFMA, want_u=False)
    np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es)))


@pytest.mark.parametrize("cin", [64, 128])
def test_conv_bits_kernel_folds_a_uniform_batchnorm_multiplier_into_its_table(dev, oracle, cin):
  """BatchNorm with zero means and biases and ONE multiplier for every channel (a freshly initialised
  BatchNorm: rsqrt(1 + eps); here also 0.83 and none at all) on the table form of the bits kernel with
  the fused membrane update: the entries of the table every channel shares are fl(current * mul)
  (snnqp.h, SNNQP_BN_MUL_UNIFORM) and the epilogue has no BatchNorm instruction left.  Rasters,
  pooled rasters and potentials bit-exact against the oracle; the same launch without the flag (the
  multiply per update) gives the same bits; a carried-in state keeps the fold out (no fused update)
  and stays exact."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=7, B=3, hw=8, cin=cin, cout=128, bits=4, seed=2301 + cin, gain=5.0)
  w = _weight(c["leaf"], 4, dev, transposed=True)
  assert ops.conv_dequant_form(w, _mslif()) == "table"
  geom = ops.ConvGeom(8, 8, cin, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  z, one = np.zeros(128, F32), np.ones(128, F32)
  u0 = (np.random.Generator(np.random.PCG64(5)).random((3, 8, 8, 128)) * 0.6).astype(F32)
  for var, scale in ((one, one), (one * F32(1.7), one * F32(1.0823)), (None, None)):
    if var is None:
      bnd, bn_plain = None, None
    else:
      bnd = dict(mean=z, var=var, scale=scale, bias=z)
      bn_plain = _bn(bnd, dev)
      assert len(set(_np(bn_plain.mul).view(np.uint32).tolist())) == 1
    qw = qweight_of(oracle, c["leaf"], 4)
    for carry in (None, u0):
      eu, es = oracle.conv_block(c["x"], qw, bnd, None, "int", u0=carry)
      assert 0.01 < es.mean() < 0.7
      outs = []
      for flags in ((L.BN_MEAN_ZERO | L.BN_BIAS_ZERO | L.BN_MUL_UNIFORM, L.BN_MEAN_ZERO | L.BN_BIAS_ZERO, 0)
                    if bn_plain is not None else (0,)):
        bn = None if bn_plain is None else dataclasses.replace(bn_plain, flags=flags)
        mb = ops.current_min_bits(w, bn, int(w.abs_sum_max), 128)
        ww = dataclasses.replace(w, min_current_bits=mb)
        for pool in (1, 2):
          u, s = ops.conv_lif_forward(xin, geom, ww, _mslif(), bn=bn, u0=None if carry is None else _t(carry, dev),
                                      packed_out=True, pool=pool, impl=L.IMPL_MFMA, x_max=1)
          np.testing.assert_array_equal(_np(s), packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es))
          np.testing.assert_array_equal(_np(u), eu)
  # the module computes the flag when it folds the statistics on the host
  from snnquantprune_amd import linen as nn

  class Probe(nn.Module):
    def __call__(self, n):
      return nn.BatchNorm(use_running_average=True, momentum=0.9, epsilon=1e-5).coeffs(n)

  def flags_of(scale):
    v = {"params": {"BatchNorm_0": {"scale": scale, "bias": z}}, "batch_stats": {"BatchNorm_0": {"mean": z, "var": one}}}
    return Probe().apply(nn.tree_from_numpy(v, dev), 128).flags
  assert flags_of(one) == L.BN_MEAN_ZERO | L.BN_BIAS_ZERO | L.BN_MUL_UNIFORM
  assert flags_of(np.linspace(0.8, 1.2, 128).astype(F32)) == L.BN_MEAN_ZERO | L.BN_BIAS_ZERO
  assert ops.device_status() == 0


def test_conv_bits_kernel_forms_agree_at_the_headline_shape(dev):
  """The dequantisation forms of the bits kernel are bit-equal by construction; at conv1's own
  shape (64 x 64 x 128 -> 128, T = 20, 4-bit / 90 % pruned, B = 32: 4096 patches, every
  workgroup walks several) the table form (what AUTO picks) and the arithmetic form (forced by
  hiding the accumulator bound) give the same pooled rasters and the same potentials, with and
  without a carried-in state."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=2, B=1, hw=8, cin=128, cout=128, seed=5150, gain=5.0)
  w = _weight(c["leaf"], 4, dev, transposed=True)
  assert ops.conv_dequant_form(w, _mslif()) == "table"
  wa = dataclasses.replace(w, abs_sum_max=0, min_current_bits=0)
  assert ops.conv_dequant_form(wa, _mslif()) == "arith"
  bn = _bn(c["bn"], dev)
  g = ops.ConvGeom(64, 64, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  gen = torch.Generator(device=dev); gen.manual_seed(5151)
  x = ops.pack_bits((torch.rand((20, 32, 64, 64, 128), device=dev, generator=gen) < 0.15).to(torch.uint8))
  u0 = torch.rand((32, 64, 64, 128), device=dev, generator=gen) * 0.7
  for carry in (None, u0):
    ut, st = ops.conv_lif_forward(x, g, w, _mslif(), bn=bn, u0=carry, packed_out=True, pool=2, impl=L.IMPL_MFMA)
    ua, sa = ops.conv_lif_forward(x, g, wa, _mslif(), bn=bn, u0=carry, packed_out=True, pool=2, impl=L.IMPL_MFMA)
    assert torch.equal(st.bits, sa.bits) and torch.equal(ut, ua)
    assert 0.01 < float(st.to_dense().float().mean()) < 0.9


def test_conv_work_queues_on_concurrent_streams(dev, oracle):
  """The patch work queues come from a per-device pool of slots: launches that overlap on
  different streams (and more than 64 launches in a row, so slots are reused) still give
  the oracle's rasters."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=5, B=24, hw=16, cin=128, seed=2301, gain=5.0)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  geom = ops.ConvGeom(16, 16, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  bn = _bn(c["bn"], dev)
  torch.cuda.synchronize()
  streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
  outs = []
  for rep in range(30):                       # 90 launches: every slot of the pool reused
    for st in streams:
      with torch.cuda.stream(st):
        outs.append(ops.conv_lif_forward(xin, geom, w, _mslif(), bn=bn, packed_out=True, pool=2,
                                         impl=L.IMPL_MFMA, x_max=1))
  torch.cuda.synchronize()
  for u, s in outs:
    np.testing.assert_array_equal(_np(s), e["pooled_bits"])
    np.testing.assert_array_equal(_np(u), e["u"])


def test_event_layer_fallback_in_a_later_chunk(dev, oracle):
  """conv0 stages its halo in chunks of at most 32 timesteps and decides the path (tables /
  fused membrane update / general) per chunk.  T = 40 is two chunks; the hot pixel (count
  200) sits in the SECOND chunk of one patch only, so that patch runs a table + fused
  chunk and then a general chunk, carrying its potentials across (and, with u0, starts from
  a carried-in state, where the fused form is never taken).  Rasters and the final membrane
  potentials against the oracle, hints 1 and 4, with and without u0."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, hw = 40, 3, 16
  c = cases.conv_block_case(T=T, B=B, hw=hw, cin=2, seed=961, gain=4.0)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(hw, hw, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  rng = np.random.Generator(np.random.PCG64(77))
  x = (rng.random((T, B, hw, hw, 2)) < 0.1).astype(np.uint8)
  x[35, 1, 9, 4, 0] = 200                               # second chunk of sample 1, patch (1, 0)
  x[36, 1, 9, 4, 1] = 3
  u0 = (rng.random((B, hw, hw, 128)) * 0.6).astype(F32)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  for carry in (None, u0):
    eu, es = oracle.conv_block(x, qw, c["bn"], None, "int", u0=carry)
    for hint in (1, 4):
      seen = torch.zeros(8, dtype=torch.int32, device=dev)
      for pool in (1, 2):
        u, s = ops.conv_lif_forward(_t(x, dev), g, w, nrn, bn=bn,
                                    u0=None if carry is None else _t(carry, dev), want_u=True,
                                    packed_out=True, pool=pool, impl=L.IMPL_MFMA, x_max=hint,
                                    x_seen=seen)
        exp = oracle.max_pool_2x2(es) if pool == 2 else es
        tag = "hint %d pool %d carry %s" % (hint, pool, carry is not None)
        np.testing.assert_array_equal(_np(s), packbits_lastaxis(exp), err_msg=tag)
        np.testing.assert_array_equal(_np(u), eu, err_msg=tag)
      assert int(seen[0].item()) == 200


def test_model_captured_into_a_graph_then_called_eagerly(dev, oracle):
  """ConvDenseSNN.apply captured into a hipGraph (the uint8 event layer's count hint keeps
  its read-back -- copy, memset, event -- out of the capture: ops.CountHint), replayed, and
  then called eagerly on the same device: all three give the oracle's logits, and the eager
  call after the capture neither raises on a captured event nor finds the hint's word
  re-zeroed by a replay."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.conv_net_case()
  e = cases.conv_net_expected(oracle, c)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)

  def apply():
    return model.apply(variables, x, trgt=None, train=False, rng=None)[0]
  np.testing.assert_array_equal(_np(apply()), e["logits"])       # warm: packs, allocations
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream(device=dev)
  with torch.cuda.stream(cap):
    apply()
    cap.synchronize()
    with torch.cuda.graph(graph, stream=cap):
      static = apply()
  for _ in range(3):
    static.zero_()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(static), e["logits"])
  hint = ops.count_hint(dev)
  for _ in range(3):                                             # eager again, hint alive
    np.testing.assert_array_equal(_np(apply()), e["logits"])
    torch.cuda.synchronize()
    assert hint.current() == 1
  graph.replay()
  torch.cuda.synchronize()
  np.testing.assert_array_equal(_np(static), e["logits"])


# ---------------------------------------------------------------------------
# packed event frames (the host feed's wire formats, snnqp.h SNNQP_EV1 / SNNQP_EV4)
# ---------------------------------------------------------------------------


@pytest.mark.parametrize("hw", [(16, 16), (13, 17), (34, 34), (5, 3)])
def test_packed_frames_round_trip(dev, hw):
  """EV1 / EV4: the host packer (numpy), the device packer and the format restated in
  tests/helpers.py agree word for word; unpack inverts pack; values a format cannot hold
  are saturated and flagged on the device, refused on the host.  Frame sizes whose bit
  count is not a multiple of 32 included."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  from tests.helpers import pack_ev1, pack_ev4
  H, W = hw
  rng = np.random.Generator(np.random.PCG64(H * 100 + W))
  binary = (rng.random((3, 4, H, W, 2)) < 0.3).astype(np.uint8)
  counts = np.minimum(rng.poisson(1.5, (3, 4, H, W, 2)), 15).astype(np.uint8)
  for fmt, x, ref in ((L.EV1, binary, pack_ev1(binary)), (L.EV4, counts, pack_ev4(counts)),
                      (L.EV4, binary, pack_ev4(binary))):
    host = ops.pack_frames_host(x, fmt)
    devp = ops.pack_frames(_t(x, dev), fmt)
    view = (lambda t: t.cpu().numpy().view(np.uint32)) if fmt == L.EV1 else (lambda t: t.cpu().numpy())
    np.testing.assert_array_equal(view(host.data), ref)
    np.testing.assert_array_equal(view(devp.data), ref)
    assert host.shape == devp.shape == x.shape
    np.testing.assert_array_equal(_np(ops.unpack_frames(devp)), x)
    np.testing.assert_array_equal(_np(ops.unpack_frames(host.to(dev))), x)
    np.testing.assert_array_equal(_np(devp[1].to_u8()), x[1])          # leading-axis indexing
  # overflow: saturate + flag on the device, raise on the host
  big = counts.copy()
  big[1, 2, H // 2, W // 2, 1] = 77
  for fmt, flag, sat in ((L.EV1, L.FLAG_GT_ONE, 1), (L.EV4, L.FLAG_GT_15, 15)):
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    p = ops.pack_frames(_t(big, dev), fmt, flags)
    assert int(flags.item()) == flag
    np.testing.assert_array_equal(_np(ops.unpack_frames(p)), np.minimum(big, sat))
    with pytest.raises(ValueError):
      ops.pack_frames_host(big, fmt)
  flags = torch.zeros(1, dtype=torch.int32, device=dev)
  ops.pack_frames(_t(binary, dev), L.EV1, flags)
  assert int(flags.item()) == 0


@pytest.mark.parametrize("hw,T", [((16, 16), 9), ((24, 24), 40), ((13, 17), 5), ((34, 34), 20)])
def test_event_layer_stages_bit_packed_frames(dev, oracle, hw, T):
  """conv0 on EV1 frames (1 bit per element, staged directly: one halo row = 20 bits out of
  two words) against the oracle and against the same launch on uint8 frames: rasters and
  final potentials bit-equal, with and without the 2x2 pool, with a carried-in state, with
  4-bit codes (per-channel tables), 8-bit codes (shared table) and with no table at all
  (unknown accumulator bound: the general x - 128 path); image sizes that are not multiples
  of 8 or whose rows do not start on a word boundary; batch-major and time-major."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  H, W = hw
  B = 3
  rng = np.random.Generator(np.random.PCG64(T * 1000 + H))
  x = (rng.random((T, B, H, W, 2)) < 0.12).astype(np.uint8)
  x[:, :, 0, 0, :] = 1                         # image corners: the masks of edge patches
  x[:, :, H - 1, W - 1, :] = 1
  x[:, :, 0, W - 1, 0] = 1
  u0 = (rng.random((B, H, W, 128)) * 0.6).astype(F32)
  g = ops.ConvGeom(H, W, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  nrn = _mslif()
  for bits, notable in ((4, False), (8, False), (8, True)):
    c = cases.conv_block_case(T=2, B=1, hw=8, cin=2, bits=bits, seed=961 + bits, gain=4.0)
    qw = qweight_of(oracle, c["leaf"], bits)
    w = _weight(c["leaf"], bits, dev, transposed=True)
    if notable:
      w = dataclasses.replace(w, abs_sum_max=0, min_current_bits=0)
    bn = _bn(c["bn"], dev)
    for carry in (None, u0):
      eu, es = oracle.conv_block(x, qw, c["bn"], None, "int", u0=carry)
      assert 0.01 < es.mean() < 0.6
      for pool in (1, 2) if H % 2 == 0 and W % 2 == 0 else (1,):
        exp = packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es)
        kw = dict(bn=bn, u0=None if carry is None else _t(carry, dev), want_u=True,
                  packed_out=True, pool=pool, impl=L.IMPL_MFMA, x_max=1)
        tag = "bits %d notable %s carry %s pool %d" % (bits, notable, carry is not None, pool)
        u8_u, u8_s = ops.conv_lif_forward(_t(x, dev), g, w, nrn, **kw)
        pf = ops.pack_frames(_t(x, dev), L.EV1)
        ev_u, ev_s = ops.conv_lif_forward(pf, g, w, nrn, **kw)
        np.testing.assert_array_equal(_np(ev_s), exp, err_msg=tag)
        np.testing.assert_array_equal(_np(ev_u), eu, err_msg=tag)
        np.testing.assert_array_equal(_np(u8_s), exp, err_msg=tag)
        # batch-major frames [B, T, words], as the model's input arrives
        pfb = ops.pack_frames(_t(np.ascontiguousarray(np.swapaxes(x, 0, 1)), dev), L.EV1)
        bm_u, bm_s = ops.conv_lif_forward(pfb, g, w, nrn, time_major=False, **kw)
        np.testing.assert_array_equal(_np(bm_s), exp, err_msg=tag + " batch-major")
        np.testing.assert_array_equal(_np(bm_u), eu, err_msg=tag + " batch-major")


@pytest.mark.parametrize("hw", [(16, 16), (13, 17), (34, 30)])
def test_event_layer_stages_nibble_packed_counts(dev, oracle, hw):
  """conv0 on EV4 frames (counts <= 15, one byte per pixel, staged directly: a byte becomes
  the two bytes the uint8 frame would hold) against the oracle and against the same launch on
  uint8 frames: rasters and potentials bit-equal for hints below, at and above the real
  maximum (tables / general path per chunk), 4- and 8-bit codes, pooled or not, time- and
  batch-major; the count-hint words see the same maxima as on uint8 frames."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  H, W = hw
  T, B = 7, 3
  rng = np.random.Generator(np.random.PCG64(4400 + H))
  x = np.minimum(rng.poisson(0.35, (T, B, H, W, 2)), 15).astype(np.uint8)
  x[2, 1, H // 2, W // 2, 1] = 15                # one pixel at the format's limit
  x[:, :, 0, 0, 0] = 3
  g = ops.ConvGeom(H, W, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  nrn = _mslif()
  for bits in (4, 8):
    c = cases.conv_block_case(T=2, B=1, hw=8, cin=2, bits=bits, seed=961 + bits, gain=4.0)
    qw = qweight_of(oracle, c["leaf"], bits)
    w = _weight(c["leaf"], bits, dev, transposed=True)
    bn = _bn(c["bn"], dev)
    eu, es = oracle.conv_block(x, qw, c["bn"], None, "int")
    assert 0.01 < es.mean() < 0.8
    pf = ops.pack_frames(_t(x, dev), L.EV4)
    for hint in (1, 3, 15):
      for pool in (1, 2) if H % 2 == 0 and W % 2 == 0 else (1,):
        exp = packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es)
        seen_a = torch.zeros(8, dtype=torch.int32, device=dev)
        seen_b = torch.zeros(8, dtype=torch.int32, device=dev)
        kw = dict(bn=bn, want_u=True, packed_out=True, pool=pool, impl=L.IMPL_MFMA, x_max=hint)
        tag = "bits %d hint %d pool %d" % (bits, hint, pool)
        ev_u, ev_s = ops.conv_lif_forward(pf, g, w, nrn, x_seen=seen_a, **kw)
        u8_u, u8_s = ops.conv_lif_forward(_t(x, dev), g, w, nrn, x_seen=seen_b, **kw)
        np.testing.assert_array_equal(_np(ev_s), exp, err_msg=tag)
        np.testing.assert_array_equal(_np(ev_u), eu, err_msg=tag)
        np.testing.assert_array_equal(_np(u8_s), exp, err_msg=tag)
        assert seen_a.tolist() == seen_b.tolist() and seen_a[0].item() == 15, (tag, seen_a.tolist())
    pfb = ops.pack_frames(_t(np.ascontiguousarray(np.swapaxes(x, 0, 1)), dev), L.EV4)
    bm_u, bm_s = ops.conv_lif_forward(pfb, g, w, nrn, bn=bn, want_u=True, packed_out=True,
                                      impl=L.IMPL_MFMA, x_max=3, time_major=False)
    np.testing.assert_array_equal(_np(bm_s), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(bm_u), eu)


def test_models_take_packed_frames(dev, oracle):
  """model.apply on PackedFrames -- EV1 (binary, staged directly by the event layer) and EV4
  (counts, unpacked on the device first) -- gives the logits of the uint8 input and of the
  oracle: the C3 topology and the full CextNet."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.conv_net_case()
  e = cases.conv_net_expected(oracle, c)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  host = ops.pack_frames_host(c["x"], L.EV1)             # packed on the host, shipped as is
  for inp in (host.to(dev), ops.pack_frames(_t(c["x"], dev), L.EV1),
              ops.pack_frames(_t(c["x"], dev), L.EV4), _t(c["x"], dev)):
    (logits, _), mut = model.apply(variables, inp, trgt=None, train=False, rng=None,
                                   mutable=["intermediates"])
    np.testing.assert_array_equal(_np(logits), e["logits"])
    np.testing.assert_array_equal(_np(mut["intermediates"]["pool0"][0]), e["pool0_bits"])
  cc = cases.conv_net_case(counts=True)
  ec = cases.conv_net_expected(oracle, cc)
  vc = nn.tree_from_numpy(cc["vars"], dev)
  assert 1 < int(cc["x"].max()) <= 15
  (logits, _) = model.apply(vc, ops.pack_frames_host(cc["x"], L.EV4).to(dev), trgt=None,
                            train=False, rng=None)
  np.testing.assert_array_equal(_np(logits), ec["logits"])
  ct = cases.cextnet_case()
  et = cases.cextnet_expected(oracle, ct)
  net = models.CextNet(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  vt = nn.tree_from_numpy(ct["vars"], dev)
  (logits, _) = net.apply(vt, ops.pack_frames_host(ct["x"], L.EV1).to(dev), trgt=None, train=False,
                          rng=None)
  np.testing.assert_array_equal(_np(logits), et["logits"])


def test_feeder_hands_over_every_batch_in_order(dev):
  """feed.DeviceFeeder on the GPU: 12 batches through a ring of 4 device buffers (depth 2),
  pinned and unpinned sources, a consumer that keeps using batch k while k + 1 and k + 2
  are copied: every batch arrives intact and in order, and a batch stays valid until the
  next one has been asked for."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import feed, ops
  rng = np.random.Generator(np.random.PCG64(9))
  frames = [(rng.random((4, 3, 16, 16, 2)) < 0.3).astype(np.uint8) for _ in range(12)]

  def source(pin, packed):
    for i, f in enumerate(frames):
      v = ops.pack_frames_host(f, L.EV1) if packed else f
      yield {"dvs_matrix": feed.pinned_like(v) if pin else v, "label": np.full((4,), i, np.int8)}
  for pin in (True, False):
    for packed in (True, False):
      f = feed.DeviceFeeder(source(pin, packed), dev, 2)
      sums = []
      for i, b in enumerate(f):
        x = b["dvs_matrix"]
        u8 = x.to_u8() if packed else x
        # a long-running consumer of this batch: the copies of the next two overlap it
        acc = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(20):
          acc = acc + u8.to(torch.int64).sum()
        sums.append((acc, int(b["label"][0].item())))
        np.testing.assert_array_equal(_np(u8), frames[i])
      assert [s[1] for s in sums] == list(range(12))
      for i, (acc, _) in enumerate(sums):
        assert int(acc.item()) == 20 * int(frames[i].sum())
      assert f.batches == 12 and f.bytes_copied > 0


def test_evaluate_restores_feeds_and_scores_like_the_oracle(dev, oracle, tmp_path, golden_dir):
  """eval.evaluate_metrics (examples/eval.py:53-139) end to end on the GPU: a Flax checkpoint
  file in the workdir -> restore -> the split through the feeder (uint8 and bit-packed
  frames) -> eval_step per batch -> mean loss / accuracy, against the oracle's logits and
  metrics on the same samples."""
  from snnquantprune_amd import checkpoint, eval as ev, linen as nn
  from snnquantprune_amd import synthetic as syn
  from snnquantprune_amd.train_utils import mse_loss
  c = cases.conv_net_case(B=8)
  rng = np.random.Generator(np.random.PCG64(12))
  labels = rng.integers(0, 11, 8).astype(np.int8)
  state = {"step": np.int32(3), "params": {"params": c["vars"]["params"]},
           "batch_stats": c["vars"]["batch_stats"]}
  (tmp_path / "checkpoint_3").write_bytes(checkpoint.msgpack_serialize(state))
  (tmp_path / "checkpoint_1").write_bytes(b"stale")
  data = str(tmp_path / "frames.npz")
  np.savez(data, dvs_matrix=c["x"], label=labels)
  e = cases.conv_net_expected(oracle, c)
  want = [oracle.compute_metrics(e["logits"][i * 4:(i + 1) * 4], labels[i * 4:(i + 1) * 4])
          for i in range(2)]
  for fmt in ("u8", "ev1"):
    cfg = syn.make_config(bits=4, prune_percentage=0.9)
    cfg.model, cfg.dataset, cfg.feed_format = "ConvDenseSNN", data, fmt
    cfg.eval_batch_size, cfg.batch_size, cfg.steps_per_eval = 4, 4, -1
    cfg.loss_fn, cfg.smoothing = mse_loss, 0.0
    state, summary, per_step = ev.evaluate_metrics(cfg, str(tmp_path), device=dev)
    assert summary["steps"] == 2 and summary["samples"] == 8
    np.testing.assert_array_equal(per_step["accuracy"].numpy().reshape(2, 4),
                                  np.stack([w["accuracy"] for w in want]).astype(np.float32))
    np.testing.assert_allclose(per_step["loss"].numpy().reshape(2), [w["loss"] for w in want],
                               rtol=1e-6)
    assert abs(summary["accuracy"] - np.mean([w["accuracy"].mean() for w in want])) < 1e-6
    assert "QuantConv_0" in state.params["params"]


# ---------------------------------------------------------------------------
# dense blocks on uint8 rows, config C2 without a host synchronisation, fallbacks, capture
# ---------------------------------------------------------------------------


@pytest.mark.parametrize("shape", [(20, 37, 2048, 512), (6, 5, 208, 70), (64, 3, 96, 33),
                                   (9, 130, 4096, 110)],
                         ids=["c2_layer1", "ragged_n_k16", "longest_t", "many_rows"])
def test_dense_block_reads_uint8_rows_in_place(dev, oracle, shape):
  """The MFMA dense kernel on uint8 rows (IN = U8: x - 128 against the codes, 128 * col_sum
  added back): binary rows, small counts and rows holding every value up to 255 give the
  oracle's rasters and potentials, batch-major and time-major, with a carried-in state --
  no packing pass and no inspection of the values in front of the launch."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert w.wt is not None and w.col_sum is not None
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  rng = np.random.Generator(np.random.PCG64(K + N))
  small = np.minimum(rng.poisson(0.15, (T, B, K)), 255).astype(np.uint8)
  wide = small.copy()
  wide[rng.random(wide.shape) < 0.002] = 255
  wide[0, 0, :3] = (128, 127, 200)
  u0 = _t(c["u0"], dev)
  before = ops.fallback_counts()["dense_blocks"]
  for name, x in (("binary", c["x"]), ("counts", small), ("to_255", wide)):
    eu, es = oracle.dense_block(x, qw, None, "int", u0=c["u0"])
    u, s = ops.dense_lif_forward(_t(x, dev), w, K, N, _mslif(), u0=u0, packed_out=True,
                                 impl=L.IMPL_MFMA)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg=name)
    np.testing.assert_array_equal(_np(u), eu, err_msg=name)
    xb = _t(np.ascontiguousarray(np.swapaxes(x, 0, 1)), dev)
    ub, sb = ops.dense_lif_forward(xb, w, K, N, _mslif(), u0=u0, packed_out=True,
                                   impl=L.IMPL_AUTO, time_major=False)
    np.testing.assert_array_equal(_np(sb), packbits_lastaxis(es), err_msg=name + " batch-major")
    np.testing.assert_array_equal(_np(ub), eu, err_msg=name + " batch-major")
  assert ops.fallback_counts()["dense_blocks"] == before          # AUTO stayed on the MFMA kernel


def test_c2_steps_without_a_host_synchronisation(dev, oracle, monkeypatch):
  """Config C2 (uint8 [B, T, 2048] -> qdense 512 -> qdense 110 -> vote) at its own size,
  B = 256, T = 20: bit-exact logits, and a step neither inspects its input on the host
  (Tensor.item and Tensor.tolist are made to raise) nor falls back to
  the direct-form kernel; a fresh batch every step."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.dense_net_case(True, T=20, B=256, K=2048, hidden=512)
  e = cases.dense_net_expected(oracle, c)
  model = models.DenseSNN(num_classes=11, config=syn.make_config(bits=8, prune_percentage=0.5, hidden=512))
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  (logits, _) = model.apply(variables, x, trgt=None, train=False, rng=None)     # packs the weights
  np.testing.assert_array_equal(_np(logits), e["logits"])
  ops.fallback_counts(reset=True)

  def boom(*a, **k):
    raise AssertionError("host synchronisation inside a C2 step")
  monkeypatch.setattr(torch.Tensor, "item", boom)
  monkeypatch.setattr(torch.Tensor, "tolist", boom)
  outs = []
  for i in range(3):
    xi = torch.roll(x, i, 0)                                         # a new tensor each step
    outs.append(model.apply(variables, xi, trgt=None, train=False, rng=None)[0])
  monkeypatch.undo()
  for i, o_ in enumerate(outs):
    np.testing.assert_array_equal(_np(o_), np.roll(e["logits"], i, 0))
  assert ops.fallback_counts() == {"conv_blocks": 0, "dense_blocks": 0, "last_reason": ""}


def test_c1_at_its_own_size(dev, oracle):
  """BASELINE config C1 at full size: 2048 -> 512 -> 110, float32 weights (no `weight` key:
  flax_qdense.py:74-82 pass-through), no pruning, T = 10, B = 32 -- spikes and logits against
  the oracle's `fseq` mode (the f32-MFMA connection runs that fmaf chain)."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.dense_net_case(False, T=10, B=32, K=2048, hidden=512)
  e = cases.dense_net_expected(oracle, c)
  assert 0.01 < e["s1"].mean() < 0.6 and 0.01 < e["s2"].mean() < 0.6
  cfg = syn.make_config(bits=-1, prune_percentage=-1.0, hidden=512, quantized=False)
  model = models.DenseSNN(num_classes=11, config=cfg)
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None,
                                 train=False, rng=None, mutable=["intermediates"])
  np.testing.assert_array_equal(_np(logits), e["logits"])
  s2 = mut["intermediates"]["dense2_out"][0]
  s2 = s2.to_dense() if hasattr(s2, "to_dense") else s2
  np.testing.assert_array_equal(_np(s2).astype(np.uint8), e["s2"])


def test_fallbacks_to_the_direct_form_kernel_are_counted(dev, oracle):
  """IMPL_AUTO handing a block to the direct-form kernel (20-25 x slower) is visible: the
  library counts such blocks and keeps the reason (snnqp_fallback_counts)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  ops.fallback_counts(reset=True)
  # dense, T > 96 on bit-packed rows
  c = cases.dense_block_case(T=100, B=2, K=64, N=32)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  ops.dense_lif_forward(ops.pack_bits(_t(c["x"], dev)), w, 64, 32, _mslif(), packed_out=True)
  f = ops.fallback_counts()
  assert f["dense_blocks"] == 1 and f["conv_blocks"] == 0 and "96" in f["last_reason"]
  # conv, 5x5 kernel
  rng = np.random.Generator(np.random.PCG64(4))
  leaf = {"kernel": (rng.standard_normal((5, 5, 32, 32)) * 0.2).astype(F32),
          "DuQ_0": {"a": F32([1.0]), "c": F32([0.9])}}
  w5 = _weight(leaf, 4, dev)
  x = ops.pack_bits(_t((rng.random((3, 2, 8, 8, 32)) < 0.2).astype(np.uint8), dev))
  g = ops.ConvGeom(8, 8, 32, 32, 5, 5, (1, 1), ((2, 2), (2, 2)))
  ops.conv_lif_forward(x, g, w5, _mslif(), packed_out=True)
  f = ops.fallback_counts()
  assert f["conv_blocks"] == 1 and f["last_reason"].startswith("conv: ") and "3x3" in f["last_reason"]
  # an explicit request for the direct-form kernel is not a fallback
  ops.conv_lif_forward(x, g, w5, _mslif(), packed_out=True, impl=L.IMPL_GENERIC)
  assert ops.fallback_counts(reset=True)["conv_blocks"] == 1
  assert ops.fallback_counts() == {"conv_blocks": 0, "dense_blocks": 0, "last_reason": ""}


def test_capture_replays_the_model_as_one_graph(dev, oracle):
  """nn.capture: model.apply recorded into a hipGraph -- C2 (uint8 rows), C3 on uint8 frames
  and on bit-packed frames -- replayed on new inputs gives the oracle's logits; a wrong input
  shape is refused."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.dense_net_case(True)
  e = cases.dense_net_expected(oracle, c)
  model = models.DenseSNN(num_classes=11, config=syn.make_config(bits=8, prune_percentage=0.5, hidden=96))
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  step = nn.capture(model, variables, torch.zeros_like(x), trgt=None, train=False, rng=None)
  for i in range(3):
    logits, _ = step(torch.roll(x, i, 0))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(logits), np.roll(e["logits"], i, 0))
  with pytest.raises(ValueError):
    step(x[:1])
  c3 = cases.conv_net_case()
  e3 = cases.conv_net_expected(oracle, c3)
  m3 = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  v3 = nn.tree_from_numpy(c3["vars"], dev)
  for inp in (_t(c3["x"], dev), ops.pack_frames(_t(c3["x"], dev), L.EV1)):
    zero = ops.pack_frames(torch.zeros_like(_t(c3["x"], dev)), L.EV1) if isinstance(inp, ops.PackedFrames) \
        else torch.zeros_like(inp)
    step3 = nn.capture(m3, v3, zero, trgt=None, train=False, rng=None)
    for _ in range(2):
      logits, _ = step3(inp)
      torch.cuda.synchronize()
      np.testing.assert_array_equal(_np(logits), e3["logits"])


# ---------------------------------------------------------------------------
# dense blocks on the f8f6f4 MFMA (codes of magnitude <= 7, packed as fp6)
# ---------------------------------------------------------------------------


def test_fp6_tiles_match_the_documented_layout(dev):
  """snnqp_pack_codes_fp6 against the layout of include/snnqp.h restated in tests/helpers.py:
  K not a multiple of 64, N not a multiple of 32, every code value -7..7."""
  from snnquantprune_amd import ops
  from tests.helpers import fp6_tiles
  rng = np.random.Generator(np.random.PCG64(66))
  for K, N in ((64, 32), (200, 70), (4100, 110)):
    codes = rng.integers(-7, 8, (K, N)).astype(np.int8)
    codes[rng.random((K, N)) < 0.5] = 0
    got = ops.pack_codes_fp6(_t(codes, dev)).cpu().numpy()
    np.testing.assert_array_equal(got, fp6_tiles(codes))


@pytest.mark.parametrize("shape", [(20, 64, 32768, 110, 4), (6, 5, 200, 70, 4), (33, 9, 64, 33, 3),
                                   (10, 3, 96, 160, 2), (100, 3, 4100, 110, 4), (160, 1, 256, 40, 4),
                                   (20, 300, 2048, 512, 4)],
                         ids=["readout", "ragged_k_n", "t33_3bit", "two_col_blocks_2bit", "t100_odd_k",
                              "t160", "four_col_blocks_many_rows"])
def test_dense_block_fp6_mfma(dev, oracle, shape):
  """Dense block whose codes fit fp6 (DuQ 2, 3 and 4 bits) on the f8f6f4 kernel: rasters and
  potentials against the oracle, against the int8 kernel and the direct-form kernel on the
  same inputs; carried-in state, BatchNorm, batch-major rows, every neuron form; K that is
  not a multiple of 32 / 64 / 256, N beyond one 128-feature block, T up to 160."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N, bits = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N, bits=bits, p=0.9 if K > 1000 else 0.5)
  qw = qweight_of(oracle, c["leaf"], bits)
  w = _weight(c["leaf"], bits, dev, transposed=True)
  assert w.wt6 is not None and 0 < w.code_max <= 7
  x = ops.pack_bits(_t(c["x"], dev))
  u0 = _t(c["u0"], dev)
  ops.fallback_counts(reset=True)
  eu, es = oracle.dense_block(c["x"], qw, None, "int", u0=c["u0"])
  assert 0.005 < es.mean() < 0.6, es.mean()
  u, s = ops.dense_lif_forward(x, w, K, N, _mslif(), u0=u0, packed_out=True)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  w8 = dataclasses.replace(w, wt6=None)                           # the int8 kernel, same codes
  if w8.wt is not None and T <= 96:
    u8, s8 = ops.dense_lif_forward(x, w8, K, N, _mslif(), u0=u0, packed_out=True, impl=L.IMPL_MFMA)
    np.testing.assert_array_equal(_np(s8), _np(s))
    np.testing.assert_array_equal(_np(u8), _np(u))
  xb = ops.pack_bits(_t(np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)), dev))
  ub, sb = ops.dense_lif_forward(xb, w, K, N, _mslif(), u0=u0, packed_out=True, time_major=False)
  np.testing.assert_array_equal(_np(sb), _np(s))
  np.testing.assert_array_equal(_np(ub), _np(u))
  if B * T * K * N < 5e9:
    rng = np.random.Generator(np.random.PCG64(N))
    bnd = dict(mean=(0.1 * rng.standard_normal(N)).astype(F32), var=(1 + 0.3 * rng.random(N)).astype(F32),
               scale=(1 + 0.2 * rng.standard_normal(N)).astype(F32), bias=(0.1 * rng.standard_normal(N)).astype(F32))
    for nrn in (ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, 0.8, 0.1),
                ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, 0.3, 1.0, 0.0),
                ops.Neuron(L.NEURON_LIF, 0.0, 1.0, 0.0, decay=_t(np.linspace(0.2, 0.9, N).astype(F32), dev))):
      ua, sa = ops.dense_lif_forward(x, w, K, N, nrn, bn=_bn(bnd, dev), packed_out=True)
      ug, sg = ops.dense_lif_forward(x, w, K, N, nrn, bn=_bn(bnd, dev), packed_out=True, impl=L.IMPL_GENERIC)
      np.testing.assert_array_equal(_np(sa), _np(sg))
      np.testing.assert_array_equal(_np(ua), _np(ug))
  assert ops.fallback_counts()["dense_blocks"] == 0


def test_bench_runs_its_collectives_through_rccl_on_one_gpu(dev, tmp_path):
  """The multi-GPU path's moving parts on the one GPU a test box has: bench.py with
  --single-rank-collective creates the RCCL process group (backend nccl), runs the step's
  all-gather of the logits, the device-bound barrier and the all-reduce of the timing through
  it, and says so in its line (RCCL version, PCI address of the device)."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
  env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
  p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "16", "--steps", "3",
                      "--warmup", "1", "--no-cpu-baseline", "--no-fed-leg", "--single-rank-collective",
                      "--detail", str(tmp_path / "detail.json")],
                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
  assert p.returncode == 0, p.stderr.decode()[-2000:]
  last = p.stdout.decode().rstrip("\n").splitlines()[-1]
  short = json.loads(last)                       # the compact line the driver parses, last on stdout
  assert len(last) < 6000 and short["roofline"]["frac"] > 0 and short["device_status"] == 0
  d = json.load(open(str(tmp_path / "detail.json")))
  assert d["value"] == short["value"] and d["roofline"]["frac"] == short["roofline"]["frac"]
  assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["collective"].startswith("nccl process group")
  r = d["ranks"][0]
  assert r["rccl"] not in (None, "", "unknown") and r["pci"].count(":") == 2
  assert d["fallbacks"]["conv_blocks"] == 0 and d["value"] > 0


def test_eval_command_line_on_a_synthetic_workdir(dev, tmp_path):
  """`python -m snnquantprune_amd.eval --workdir DIR --config DIR/config.py` on a work directory
  written by tools/make_synthetic_eval.py (checkpoint file, frames, config file): the harness
  restores, feeds bit-packed frames, evaluates every sample once and prints one JSON summary;
  run again on uint8 frames it reports the same metrics."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
  out = {}
  for feed in ("ev1", "u8"):
    wd = str(tmp_path / feed)
    subprocess.run([sys.executable, os.path.join(root, "tools", "make_synthetic_eval.py"), wd, "--samples",
                    "12", "--hw", "16", "--frames", "4", "--batch", "4", "--feed", feed], check=True, cwd=root)
    p = subprocess.run([sys.executable, "-m", "snnquantprune_amd.eval", "--workdir", wd, "--config",
                        os.path.join(wd, "config.py")], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out[feed] = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out[feed]["steps"] == 3 and out[feed]["samples"] == 12 and out[feed]["world"] == 1
    assert 0.0 <= out[feed]["accuracy"] <= 1.0 and out[feed]["loss"] > 0
  assert out["ev1"] == out["u8"]


def test_count_hint_follows_the_chunks_not_the_hot_pixel(dev, oracle):
  """Binary frames with one hot pixel (count 200): the kernel reports how many staged chunks
  held which maximum, and the hint the model path derives from it stays at 1 -- the chunks
  around the hot pixel take the general path, everything else keeps the fast tables -- while
  every result stays the oracle's.  Count frames move the hint to the bucket that covers them."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.conv_net_case(B=4, hw=32)
  x = c["x"].copy()
  x[2, :, 11, 17, 1] = 200                                    # a stuck pixel, every frame of sample 2
  c = dict(c, x=x)
  e = cases.conv_net_expected(oracle, c)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  hint = ops.count_hint(dev)
  for i in range(4):
    (logits, _) = model.apply(variables, _t(x, dev), trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(logits), e["logits"])
    torch.cuda.synchronize()
  assert hint.max_seen == 200 and hint.current() == 1
  assert ops.CountHint.choose([100, 0, 0, 0, 3]) == 1 and ops.CountHint.choose([0, 0, 90, 10, 0]) == 31
  assert ops.CountHint.choose([10, 5, 80, 0, 5]) == 7 and ops.CountHint.choose([0, 0, 0, 0, 9]) == 255
  # Poisson(0.1) count frames: most chunks stop at 2 or 3 -- the per-channel tables reach 3
  assert ops.CountHint.choose([0, 55, 45, 0, 0, 43]) == 3 and ops.CountHint.choose([0, 55, 45, 0, 0, 10]) == 7


def test_float32_frames_step_without_a_read_back_and_capture(dev, oracle, monkeypatch):
  """The reference's own input format -- float32 frames (flax_qconv.py:101 casts every input) --
  through config C3's topology: the event layer stages them in place and checks them on the
  device, so a step reads nothing back (Tensor.item / tolist raise) and captures into a hipGraph
  (rounds 1-4 inspected and narrowed every batch on the host and refused the capture); logits
  equal the oracle's on binary frames and on count frames."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  for counts in (False, True):
    c3 = cases.conv_net_case(counts=counts)
    e3 = cases.conv_net_expected(oracle, c3)
    m3 = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
    v3 = nn.tree_from_numpy(c3["vars"], dev)
    x = _t(c3["x"], dev)
    xf = x.to(torch.float32)
    m3.apply(v3, xf, trgt=None, train=False, rng=None)              # packs the weights

    def boom(*a, **k):
      raise AssertionError("host synchronisation inside a step on float32 frames")
    with monkeypatch.context() as m:
      m.setattr(torch.Tensor, "item", boom)
      m.setattr(torch.Tensor, "tolist", boom)
      logits = m3.apply(v3, xf, trgt=None, train=False, rng=None)[0]
    np.testing.assert_array_equal(_np(logits), e3["logits"])
    step = nn.capture(m3, v3, xf, trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(step(xf)[0]), e3["logits"])
    np.testing.assert_array_equal(_np(step(torch.roll(xf, 1, 0))[0]), np.roll(e3["logits"], 1, 0))
    del step
  assert ops.device_status() == 0


def test_zz_captured_launches_beyond_the_capture_slots_walk_statically(dev, oracle):
  """A device has 960 work-queue slots for launches captured into graphs, each taken for good
  (snnqp.h).  One graph of 1000 conv launches uses them up: the launches beyond take the static
  walk, and every one of the 1000 outputs is the oracle's, on the first replay and on the
  second.  (Last in the file: captures of later tests in this process would find no slot.)"""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=3, B=8, hw=8)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  cap = torch.cuda.Stream(device=dev)
  graph = torch.cuda.CUDAGraph()
  n = 1000
  static0 = ops.workqueue_stats()["captured_static_walks"]
  with torch.cuda.stream(cap):
    _, s0 = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                 impl=L.IMPL_MFMA, x_max=1)
    outs = torch.zeros((n,) + tuple(s0.bits.shape), dtype=s0.bits.dtype, device=dev)
    cap.synchronize()
    mark0 = ops.workqueue_capture_mark(dev)
    with torch.cuda.graph(graph, stream=cap):
      for i in range(n):
        _, s = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                    impl=L.IMPL_MFMA, x_max=1)
        outs[i].copy_(s.bits)
  for _ in range(2):
    outs.zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = _np(outs).view(np.uint32)
    assert (got == e["pooled_bits"][None]).all()
  # the launches that found no slot were counted, and the slots come back with the graph
  assert ops.workqueue_stats()["captured_static_walks"] - static0 >= n - 960
  mark1 = ops.workqueue_capture_mark(dev)
  assert 0 < mark1 - mark0 <= 960
  del graph
  torch.cuda.synchronize()
  ops.workqueue_capture_release(dev, mark0, mark1)
  static1 = ops.workqueue_stats()["captured_static_walks"]
  graph2 = torch.cuda.CUDAGraph()
  with torch.cuda.stream(cap):
    with torch.cuda.graph(graph2, stream=cap):
      _, s = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                  impl=L.IMPL_MFMA, x_max=1)
  graph2.replay()
  torch.cuda.synchronize()
  np.testing.assert_array_equal(_np(s), e["pooled_bits"])
  assert ops.workqueue_stats()["captured_static_walks"] == static1      # it got a slot again
  mark2 = ops.workqueue_capture_mark(dev)
  del graph2
  ops.workqueue_capture_release(dev, mark1, mark2)


# ---------------------------------------------------------------------------
# round 4: the wide dense kernel (a workgroup per 256 / 512-column block, the neuron from the
# accumulator registers) and the dense head as one launch
# ---------------------------------------------------------------------------


@pytest.mark.parametrize("shape", [(20, 37, 2048, 512), (1, 9, 64, 129), (3, 70, 160, 256),
                                   (64, 3, 96, 300), (17, 11, 1040, 700), (33, 6, 48, 512),
                                   (5, 300, 272, 384)],
                         ids=["c2_layer1", "one_step", "ct1_full", "longest_t", "two_col_blocks",
                              "t_over_32", "many_samples"])
@pytest.mark.parametrize("fmt", ["u8", "bits"])
def test_dense_wide_kernel(dev, oracle, shape, fmt):
  """Dense blocks with more than 128 features (dense_wide.hip): uint8 rows (binary, counts, every
  value up to 255) and bit-packed rows, time-major and batch-major, with a carried-in state, all
  row-tile counts the launcher may pick: rasters and potentials equal the oracle's."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  rng = np.random.Generator(np.random.PCG64(K + N))
  u0 = _t(c["u0"], dev)
  before = ops.fallback_counts()["dense_blocks"]
  inputs = [("binary", c["x"])]
  if fmt == "u8":
    wide = np.minimum(rng.poisson(0.15, (T, B, K)), 255).astype(np.uint8)
    wide[rng.random(wide.shape) < 0.002] = 255
    wide[0, 0, :3] = (128, 127, 200)
    inputs.append(("to_255", wide))
  for name, x in inputs:
    eu, es = oracle.dense_block(x, qw, None, "int", u0=c["u0"])
    xd = _t(x, dev) if fmt == "u8" else ops.pack_bits(_t(x, dev))
    u, s = ops.dense_lif_forward(xd, w, K, N, _mslif(), u0=u0, packed_out=True, impl=L.IMPL_MFMA)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg=name)
    np.testing.assert_array_equal(_np(u), eu, err_msg=name)
    xb = np.ascontiguousarray(np.swapaxes(x, 0, 1))
    xbd = _t(xb, dev) if fmt == "u8" else ops.pack_bits(_t(xb, dev))
    ub, sb = ops.dense_lif_forward(xbd, w, K, N, _mslif(), u0=None, packed_out=True,
                                   impl=L.IMPL_AUTO, time_major=False)
    eu0, es0 = oracle.dense_block(x, qw, None, "int")
    np.testing.assert_array_equal(_np(sb), packbits_lastaxis(es0), err_msg=name + " batch-major")
    np.testing.assert_array_equal(_np(ub), eu0, err_msg=name + " batch-major")
  assert ops.fallback_counts()["dense_blocks"] == before


@pytest.mark.parametrize("kind", ["plif", "lif", "mslif_tau3", "mslif_vreset"])
def test_dense_wide_kernel_neurons_and_batchnorm(dev, oracle, kind):
  """The neuron variants of spiking_learning.py:357-438 and a BatchNorm in front of them on the
  wide dense kernel (the fast form u += (x - u) m and the general one)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N = 11, 13, 320, 200
  c = cases.dense_block_case(T=T, B=B, K=K, N=N)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  rng = np.random.Generator(np.random.PCG64(77))
  bn = {"mean": rng.normal(0, 0.1, N).astype(F32), "var": (0.5 + rng.random(N)).astype(F32),
        "scale": (0.8 + 0.4 * rng.random(N)).astype(F32), "bias": rng.normal(0, 0.1, N).astype(F32)}
  sig = lambda v: (1.0 / (1.0 + np.exp(-np.asarray(v, np.float64)))).astype(F32)
  vth, vr = 1.0, 0.0
  if kind == "plif":
    tau_param = F32(-0.35)
    nrn = ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, float(sig(tau_param)), vth, vr)
    ocfg = {"kind": "parametric_leaky_IF", "tau_param": tau_param}
  elif kind == "lif":
    tau_vec = rng.uniform(-1.0, 2.0, N).astype(F32)
    nrn = ops.Neuron(L.NEURON_LIF, 1.0, vth, vr, decay=_t(sig(tau_vec), dev))
    ocfg = {"kind": "LIF", "tau_vec": tau_vec}
  elif kind == "mslif_tau3":
    nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, vth, vr)
    ocfg = {"kind": "multi_step_LIF", "tau": 3.0}
  else:
    vth, vr = 0.8, 0.1
    nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, vth, vr)
    ocfg = {"kind": "multi_step_LIF", "tau": 2.0}
  ocfg.update(v_threshold=vth, v_reset=vr)
  for use_bn in (False, True):
    norm = (lambda y: oracle.batchnorm_eval(y, bn["mean"], bn["var"], bn["scale"], bn["bias"], 1e-5)) \
        if use_bn else None
    eu, es = oracle.spiking_block(c["u0"], c["x"], lambda xx: oracle.quant_dense(xx, qw, "int"),
                                  oracle._neuron(ocfg), norm)
    assert 0.005 < es.mean() < 0.7, es.mean()
    u, s = ops.dense_lif_forward(_t(c["x"], dev), w, K, N, nrn, bn=_bn(bn, dev) if use_bn else None,
                                 u0=_t(c["u0"], dev), packed_out=True, impl=L.IMPL_MFMA)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg="%s bn=%s" % (kind, use_bn))
    np.testing.assert_array_equal(_np(u), eu, err_msg="%s bn=%s" % (kind, use_bn))


@pytest.mark.parametrize("shape", [(20, 256, 2048, 512, 110), (6, 5, 208, 200, 70), (64, 3, 96, 300, 30),
                                   (1, 40, 64, 512, 120), (9, 131, 1040, 384, 110), (3, 77, 320, 160, 50),
                                   (20, 3000, 96, 300, 30)],
                         ids=["c2", "small", "longest_t", "one_step", "ragged_batch", "many_per_group",
                              "full_grid_unsplit"])
@pytest.mark.parametrize("fmt", ["u8", "bits"])
def test_dense_head_as_one_launch(dev, oracle, shape, fmt):
  """snnqp_dense_head_forward -- QuantDense + LIF -> QuantDense + LIF -> vote
  (examples/tcja/models.py:200-255) in one launch -- against the oracle's two blocks and vote:
  both rasters and the logits bit-exact, on uint8 rows (counts up to 255) and bit-packed rows.
  Batches that fill at most half the chip run as two workgroups per tile of samples (the hidden
  columns split, the halves of the hidden raster handed over through the workspace; every shape
  here with more than 256 hidden features but the last), larger ones as one."""
  from snnquantprune_amd import ops
  T, B, K, N1, N2 = shape
  c = cases.dense_net_case(True, T=T, B=B, K=K, hidden=N1, out=N2)
  p = c["vars"]["params"]
  x = np.ascontiguousarray(np.swapaxes(c["x"], 0, 1))                # [T, B, K]
  if fmt == "u8":
    rng = np.random.Generator(np.random.PCG64(K))
    x = x.copy()
    x[rng.random(x.shape) < 0.003] = 3
    x[0, 0, :2] = (255, 128)
  q1, q2 = qweight_of(oracle, p["QuantDense_0"], 8), qweight_of(oracle, p["QuantDense_1"], 8)
  e = oracle.dense2_forward(x, q1, q2, mode="int")
  w1 = _weight(p["QuantDense_0"], 8, dev, transposed=True)
  w2 = _weight(p["QuantDense_1"], 8, dev, transposed=True)
  xd = _t(x, dev) if fmt == "u8" else ops.pack_bits(_t(x, dev))
  logits, s1, s2 = ops.dense_head_forward(xd, w1, K, N1, _mslif(), w2, N2, _mslif(), group=10,
                                          want_s1=True, want_s2=True)
  np.testing.assert_array_equal(_np(s1), packbits_lastaxis(e["s1"].astype(np.uint8)))
  np.testing.assert_array_equal(_np(s2), packbits_lastaxis(e["s2"].astype(np.uint8)))
  np.testing.assert_array_equal(_np(logits), e["logits"])
  # without the rasters (what a step asks for), batch-major rows
  xb = np.ascontiguousarray(np.swapaxes(x, 0, 1))
  xbd = _t(xb, dev) if fmt == "u8" else ops.pack_bits(_t(xb, dev))
  l2, n1, n2 = ops.dense_head_forward(xbd, w1, K, N1, _mslif(), w2, N2, _mslif(), time_major=False)
  assert n1 is None and n2 is None
  np.testing.assert_array_equal(_np(l2), e["logits"])


def test_dense_snn_model_takes_the_fused_head(dev, oracle):
  """DenseSNN (config C2) calls the fused head when it fits and the two blocks + vote when it
  does not (hidden > 512): same logits as the oracle either way, and the fused path is ONE
  dense launch per step."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  for hidden, fused in ((512, True), (544, False)):
    c = cases.dense_net_case(True, T=20, B=24, K=512, hidden=hidden)
    e = cases.dense_net_expected(oracle, c)
    model = models.DenseSNN(num_classes=11, config=syn.make_config(bits=8, prune_percentage=0.5, hidden=hidden))
    variables = nn.tree_from_numpy(c["vars"], dev)
    model.apply(variables, _t(c["x"], dev), trgt=None, train=False, rng=None)   # packs the weights
    ops.profile_start()
    (logits, _) = model.apply(variables, _t(c["x"], dev), trgt=None, train=False, rng=None)
    prof = ops.profile_stop()
    np.testing.assert_array_equal(_np(logits), e["logits"])
    tags = sorted(prof)
    assert (tags == ["dense_head[512->512->110]"]) == fused, tags


def test_a_dirty_work_queue_is_reported_not_skipped_silently(dev, oracle):
  """A captured conv launch relies on its work-queue words being zero at every replay (the last
  workgroup of a launch leaves them so).  A word that is not -- here poked into the slot between
  two replays; in the field an aborted launch, or a graph replayed concurrently with itself --
  makes workgroups skip patches.  That must not pass as a result: the launch's patch tally does
  not add up, the device's status word says so, the next call into the library raises, and after
  the reset (the launch has left its slot clean again) replays are the oracle's once more.
  (1280 patches for at most 512 workgroups: the queue is really walked.)"""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=2, B=40, hw=32)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(32, 32, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))

  def launch():
    return ops.conv_lif_forward(xin, g, w, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                                impl=L.IMPL_MFMA, x_max=1)[1]
  assert ops.device_status() == 0
  cap = torch.cuda.Stream(device=dev)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.stream(cap):
    launch()
    cap.synchronize()
    mark = ops.workqueue_capture_mark(dev)
    with torch.cuda.graph(graph, stream=cap):
      s = launch()
  assert ops.workqueue_capture_mark(dev) - mark == 1
  graph.replay()
  torch.cuda.synchronize()
  np.testing.assert_array_equal(_np(s), e["pooled_bits"])
  assert ops.device_status() == 0
  # the queue of XCD 0 (blockIdx.y = 0: word 0) claims to be five patches in
  L.check(L.lib().snnqp_debug_workqueue_poke(0, mark, 0, 5))
  graph.replay()
  torch.cuda.synchronize()
  assert ops.device_status() == L.STATUS_QUEUE_CORRUPT
  with pytest.raises(L.SnnqpError) as err:
    launch()
  assert err.value.code == L.EHIP and "work queue" in str(err.value)
  assert ops.device_status(reset=True) == L.STATUS_QUEUE_CORRUPT and ops.device_status() == 0
  graph.replay()                                                       # the slot is clean again
  torch.cuda.synchronize()
  np.testing.assert_array_equal(_np(s), e["pooled_bits"])
  np.testing.assert_array_equal(_np(launch()), e["pooled_bits"])
  assert ops.device_status() == 0
  del graph
  ops.workqueue_capture_release(dev, mark, mark + 1)


def test_captured_apply_hands_its_slots_back(dev, oracle):
  """nn.capture of a conv model takes a work-queue slot per conv launch; destroying the
  CapturedApply gives them back, so a process that re-captures (new weights, new shapes) never
  runs out: 400 capture / destroy cycles of a three-block model (1200 slots' worth against a pool
  of 960) and not one captured launch falls to the static walk."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.conv_net_case(T=2, B=1)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  (want, _) = model.apply(variables, x, trgt=None, train=False, rng=None)
  before = ops.workqueue_stats()["captured_static_walks"]
  for i in range(400):
    step = nn.capture(model, variables, x, trgt=None, train=False, rng=None)
    if i % 100 == 0:
      logits, _ = step(x)
      assert torch.equal(logits, want)
    del step
  assert ops.workqueue_stats()["captured_static_walks"] == before


def test_table_dequantisation_is_probed_per_device(dev, oracle, monkeypatch):
  """DQ_TABLE rests on the f8f6f4 MFMA adding float32 denormals exactly; the library probes that
  once per device before the first table launch (runtime.hip).  Here: the probe has run on this
  device by now and passed (no arithmetic fallback was counted for conv1's headline weights), and
  a process in which the probe is made to fail (SNNQP_FORCE_DENORM_PROBE_FAIL) gives the same
  rasters from the arithmetic form and counts the fallback."""
  import subprocess, sys, json
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=3, B=2, hw=8)
  e = cases.conv_block_expected(oracle, c)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert ops.conv_dequant_form(w, _mslif()) == "table"
  g = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  before = ops.workqueue_stats()["dequant_table_fallbacks"]
  _, s = ops.conv_lif_forward(ops.pack_bits(_t(c["x"], dev)), g, w, _mslif(), bn=_bn(c["bn"], dev),
                              want_u=False, packed_out=True, pool=2, impl=L.IMPL_MFMA, x_max=1)
  np.testing.assert_array_equal(_np(s), e["pooled_bits"])
  assert ops.workqueue_stats()["dequant_table_fallbacks"] == before
  code = (
      "import numpy as np, torch, json, sys\n"
      "from tests import cases\n"
      "from tests.test_parity_gpu import _weight, _bn, _mslif, _t, _np\n"
      "from snnquantprune_amd import _lib as L, ops\n"
      "dev = torch.device('cuda:0')\n"
      "c = cases.conv_block_case(T=3, B=2, hw=8)\n"
      "w = _weight(c['leaf'], c['bits'], dev, transposed=True)\n"
      "g = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))\n"
      "_, s = ops.conv_lif_forward(ops.pack_bits(_t(c['x'], dev)), g, w, _mslif(), bn=_bn(c['bn'], dev),"
      " want_u=False, packed_out=True, pool=2, impl=L.IMPL_MFMA, x_max=1)\n"
      "print(json.dumps({'bits': _np(s).ravel().tolist(), 'fallbacks': ops.workqueue_stats()['dequant_table_fallbacks']}))\n")
  env = dict(os.environ, SNNQP_FORCE_DENORM_PROBE_FAIL="1")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  got = json.loads(out.stdout.strip().splitlines()[-1])
  assert got["fallbacks"] >= 1
  np.testing.assert_array_equal(np.asarray(got["bits"], np.uint32), e["pooled_bits"].ravel())


def test_an_abs_sum_max_below_the_codes_is_reported(dev, oracle):
  """snnqp_weight_t.abs_sum_max sizes the LDS tables the conv kernels dequantise through, and the
  accumulator addresses them as it is: a caller that understates it gets wrong currents.  The
  library checks the bound against the codes once per weights (a column-sum pass after the first
  launch that used them) and reports a violation through the device status word."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=2, B=2, hw=8)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  g = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  xin = ops.pack_bits(_t(c["x"], dev))
  assert ops.device_status() == 0
  ops.conv_lif_forward(xin, g, w, _mslif(), bn=_bn(c["bn"], dev), want_u=False, packed_out=True,
                       impl=L.IMPL_MFMA, x_max=1)
  torch.cuda.synchronize()
  assert ops.device_status() == 0                       # the honest bound passes
  bad = dataclasses.replace(w, abs_sum_max=max(1, w.abs_sum_max // 4))
  ops.conv_lif_forward(xin, g, bad, _mslif(), bn=_bn(c["bn"], dev), want_u=False, packed_out=True,
                       impl=L.IMPL_MFMA, x_max=1)
  torch.cuda.synchronize()
  assert ops.device_status() == L.STATUS_BOUND
  with pytest.raises(L.SnnqpError) as err:
    ops.conv_lif_forward(xin, g, w, _mslif(), bn=_bn(c["bn"], dev), want_u=False, packed_out=True,
                         impl=L.IMPL_MFMA, x_max=1)
  assert "abs_sum_max" in str(err.value)
  assert ops.device_status(reset=True) == L.STATUS_BOUND


def test_hip_against_the_reference_literal_float_arithmetic(dev, oracle):
  """BASELINE.json: "bit-exact for spike rasters and integer accumulators, within 1e-5 relative
  for float membrane potentials" -- against the reference's own float32 arithmetic, which the
  oracle restates as its `float` mode (float32 fake-quantised weights x float32 spikes summed in
  float32, BLAS order standing in for XLA's: flax_qconv.py:158-168, flax_qdense.py:87-89), not
  the `int` mode the kernels are bit-exact against.  Config C3's layers at their own geometry
  (128 x 128 x 2, T = 20, 4-bit, 90 % pruned), B = 16, each layer on the same input raster:
  the HIP rasters may differ from the float-mode rasters only where a potential sat within
  rounding of the threshold (asserted: fewer than one spike in a million), and the final
  membrane potentials of all neurons whose rasters agree differ by at most 1e-5 relative to
  max(|u|, v_th) (asserted; the measured figure is ~5e-7)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops, synthetic as syn
  B, T = 16, 20
  v = syn.conv_net_variables(prune_p=0.9, out=110)
  p = v["params"]
  x = np.swapaxes(syn.poisson_spikes((B, T, 128, 128, 2), 0.1, seed=4711), 0, 1)   # [T, B, ...]
  nrn = _mslif()
  report = {}
  cur = x
  for i, hw in enumerate((128, 64, 32)):
    leaf = p["QuantConv_%d" % i]
    qw = qweight_of(oracle, leaf, 4)
    bn = bn_of(v, i)
    uf, sf = oracle.conv_block(cur, qw, bn, None, "float")
    w = _weight(leaf, 4, dev, transposed=True)
    g = ops.ConvGeom(hw, hw, cur.shape[-1], 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
    xin = _t(cur.astype(np.uint8), dev) if i == 0 else ops.pack_bits(_t(cur.astype(np.uint8), dev))
    u, s = ops.conv_lif_forward(xin, g, w, nrn, bn=_bn(bn, dev), want_u=True, packed_out=True, pool=1,
                                impl=L.IMPL_MFMA, x_max=1)
    sh = _np(s.to_dense()).astype(np.uint8)
    report["conv%d" % i] = _float_deviation(_np(u), uf, sh, sf.astype(np.uint8))
    cur = oracle.max_pool_2x2(sh.astype(F32))            # the HIP raster feeds the next layer
  xf = oracle.flatten_channel_major(cur)
  qd = qweight_of(oracle, p["QuantDense_0"], 4)
  uf, sf = oracle.dense_block(xf, qd, None, "float")
  wd = _weight(p["QuantDense_0"], 4, dev, transposed=True)
  u, s = ops.dense_lif_forward(ops.pack_bits(_t(xf.astype(np.uint8), dev)), wd, xf.shape[-1], 110, nrn,
                               want_u=True, packed_out=True, impl=L.IMPL_MFMA)
  report["readout"] = _float_deviation(_np(u), uf, _np(s.to_dense()).astype(np.uint8), sf.astype(np.uint8))
  print("HIP vs reference-literal float arithmetic:", report)
  for name, r in report.items():
    assert r["flip_rate"] < 1e-6, (name, r)
    assert r["max_rel_to_max_u_vth"] <= 1e-5, (name, r)


def _float_deviation(u_hip, u_flt, s_hip, s_flt):
  flips = int((s_hip != s_flt).sum())
  same = np.all(s_hip == s_flt, axis=0)
  a, b = u_hip[same].astype(np.float64), u_flt[same].astype(np.float64)
  d = np.abs(a - b)
  scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1.0)        # v_th = 1
  return {"neuron_steps": int(s_hip.size), "flips": flips, "flip_rate": flips / max(s_hip.size, 1),
          "max_abs_u": float(d.max()) if d.size else 0.0,
          "max_rel_to_max_u_vth": float((d / scale).max()) if d.size else 0.0}


@pytest.mark.parametrize("shape", [(20, 64, 32768, 110), (20, 100, 32768, 110), (7, 20, 32768, 200),
                                   (33, 9, 40000, 70)],
                         ids=["readout_small_batch", "readout_100", "two_col_blocks", "ragged_k"])
def test_dense_fp6_splits_k_over_workgroups(dev, oracle, shape):
  """The fp4 x fp6 dense kernel with a workspace: long contractions over too few rows to fill the
  chip (the read-out of config C3 at small batches) split K over two or four workgroups per tile,
  partial tiles and tickets in the workspace, the last arriver runs the neuron.  Rasters and potentials equal the oracle's and the
  unsplit launch's, twice in a row (the tickets are zero again after a launch), with a carried-in
  state."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  import ctypes
  T, B, K, N = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N, bits=4, p=0.9)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert w.wt6 is not None and 0 < w.code_max <= 7
  ws_bytes = int(L.lib().snnqp_dense_workspace_bytes(L.BITS, T, B, K, N, ctypes.byref(w.struct())))
  assert ws_bytes > 0, "this shape is meant to split"
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  xin = ops.pack_bits(_t(c["x"], dev))
  u0 = _t(c["u0"], dev)
  eu, es = oracle.dense_block(c["x"], qw, None, "int", u0=c["u0"])
  assert ops.device_status() == 0
  for rep in range(2):
    u, s = ops.dense_lif_forward(xin, w, K, N, _mslif(), u0=u0, packed_out=True, impl=L.IMPL_AUTO)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg="rep %d" % rep)
    np.testing.assert_array_equal(_np(u), eu, err_msg="rep %d" % rep)
  assert ops.device_status() == 0


def test_cextnet_steps_without_a_host_synchronisation_and_captures(dev, oracle, golden_dir, monkeypatch):
  """The reference's full TCJA model (examples/tcja/models.py:31-257) on integer frames: a step
  never waits for the device (Tensor.item / tolist and the inspection helpers raise during it:
  dense1 hands its spikes on bit-packed, so nothing downstream has to look at float values), and
  nn.capture records it -- the replay gives the golden logits."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.cextnet_case()
  g = _golden(golden_dir, "cextnet_tiny")
  model = models.CextNet(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  (logits, _) = model.apply(variables, x, trgt=None, train=False, rng=None)       # packs the weights
  np.testing.assert_array_equal(_np(logits), g["logits"])

  def boom(*a, **k):
    raise AssertionError("host synchronisation inside a CextNet step")
  with monkeypatch.context() as m:
    m.setattr(torch.Tensor, "item", boom)
    m.setattr(torch.Tensor, "tolist", boom)
    ops.forget_inputs()
    (l2, _) = model.apply(variables, torch.roll(x, 1, 0), trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(l2), np.roll(g["logits"], 1, 0))
  step = nn.capture(model, variables, x, trgt=None, train=False, rng=None)
  np.testing.assert_array_equal(_np(step(x)[0]), g["logits"])
  np.testing.assert_array_equal(_np(step(torch.roll(x, 1, 0))[0]), np.roll(g["logits"], 1, 0))
  del step


@pytest.mark.parametrize("kind", ["plif", "lif", "mslif_tau3", "mslif_vreset"])
def test_dense_head_neuron_variants(dev, oracle, kind):
  """The fused dense head with every neuron of spiking_learning.py:357-438 in both blocks (the
  straight-line walk for u += (x - u) m with v_reset = 0, the general walk for the rest), a hidden
  width that takes one column tile per wave (N1 = 200) and one that takes two (N1 = 400)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  sig = lambda v: (1.0 / (1.0 + np.exp(-np.asarray(v, np.float64)))).astype(F32)
  for N1 in (200, 400):
    T, B, K, N2 = 13, 21, 320, 60
    c = cases.dense_net_case(True, T=T, B=B, K=K, hidden=N1, out=N2)
    p = c["vars"]["params"]
    x = np.ascontiguousarray(np.swapaxes(c["x"], 0, 1))
    rng = np.random.Generator(np.random.PCG64(N1))
    vth, vr = 1.0, 0.0
    def neuron(n):
      if kind == "plif":
        tp = F32(-0.35)
        return ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, float(sig(tp)), vth, vr), \
            {"kind": "parametric_leaky_IF", "tau_param": tp, "v_threshold": vth, "v_reset": vr}
      if kind == "lif":
        tv = rng.uniform(-1.0, 2.0, n).astype(F32)
        return ops.Neuron(L.NEURON_LIF, 1.0, vth, vr, decay=_t(sig(tv), dev)), \
            {"kind": "LIF", "tau_vec": tv, "v_threshold": vth, "v_reset": vr}
      if kind == "mslif_tau3":
        return ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0, vth, vr), \
            {"kind": "multi_step_LIF", "tau": 3.0, "v_threshold": vth, "v_reset": vr}
      return ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 0.8, 0.1), \
          {"kind": "multi_step_LIF", "tau": 2.0, "v_threshold": 0.8, "v_reset": 0.1}
    n1, o1 = neuron(N1)
    n2, o2 = neuron(N2)
    q1, q2 = qweight_of(oracle, p["QuantDense_0"], 8), qweight_of(oracle, p["QuantDense_1"], 8)
    _, s1 = oracle.dense_block(x, q1, o1, "int")
    _, s2 = oracle.dense_block(s1, q2, o2, "int")
    want = oracle.vote(s2, 10)
    assert 0.005 < s1.mean() < 0.7, s1.mean()
    w1 = _weight(p["QuantDense_0"], 8, dev, transposed=True)
    w2 = _weight(p["QuantDense_1"], 8, dev, transposed=True)
    logits, g1, g2 = ops.dense_head_forward(_t(x, dev), w1, K, N1, n1, w2, N2, n2, group=10,
                                            want_s1=True, want_s2=True)
    np.testing.assert_array_equal(_np(g1), packbits_lastaxis(s1.astype(np.uint8)), err_msg="%s N1 %d" % (kind, N1))
    np.testing.assert_array_equal(_np(g2), packbits_lastaxis(s2.astype(np.uint8)), err_msg="%s N1 %d" % (kind, N1))
    np.testing.assert_array_equal(_np(logits), want)


def test_dense_snn_prepared_head_follows_its_parameters(dev, oracle):
  """models.DenseSNN keeps the prepared launch of the dense head per parameter set (identity and
  version of the eight leaves, the config's factories, the input's shape).  Changing a leaf in
  place, swapping the tree, another batch size or another config object must each give the
  oracle's logits for THAT state -- never the cached launch of another."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  c = cases.dense_net_case(True, T=20, B=12, K=512, hidden=512)
  e = cases.dense_net_expected(oracle, c)
  cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
  model = models.DenseSNN(num_classes=11, config=cfg)
  variables = nn.tree_from_numpy(c["vars"], dev)
  x = _t(c["x"], dev)
  for _ in range(3):                                   # generic path, then the cached plan twice
    np.testing.assert_array_equal(_np(model.apply(variables, x, trgt=None, train=False, rng=None)[0]), e["logits"])
  # a leaf rewritten in place (a training step would): new codes, new logits
  import copy
  v2 = copy.deepcopy(c["vars"])
  k = v2["params"]["QuantDense_1"]["kernel"]
  k[:] = np.roll(k, 7, axis=1)
  e2 = cases.dense_net_expected(oracle, dict(c, vars=v2))
  assert not np.array_equal(e2["logits"], e["logits"])
  variables["params"]["QuantDense_1"]["kernel"].copy_(_t(k, dev))
  np.testing.assert_array_equal(_np(model.apply(variables, x, trgt=None, train=False, rng=None)[0]), e2["logits"])
  # another tree, another batch size
  variables3 = nn.tree_from_numpy(c["vars"], dev)
  np.testing.assert_array_equal(_np(model.apply(variables3, x, trgt=None, train=False, rng=None)[0]), e["logits"])
  np.testing.assert_array_equal(_np(model.apply(variables3, x[:5], trgt=None, train=False, rng=None)[0]), e["logits"][:5])
  # another config (4-bit codes from the same leaves): its own logits
  cfg4 = syn.make_config(bits=4, prune_percentage=0.5, hidden=512)
  m4 = models.DenseSNN(num_classes=11, config=cfg4)
  e4 = cases.dense_net_expected(oracle, dict(c, bits=4))
  np.testing.assert_array_equal(_np(m4.apply(variables3, x, trgt=None, train=False, rng=None)[0]), e4["logits"])
  np.testing.assert_array_equal(_np(model.apply(variables3, x, trgt=None, train=False, rng=None)[0]), e["logits"])


def test_poisoned_tickets_cannot_reach_a_launch(dev, oracle):
  """The two in-launch hand-overs (the fused head's column split, the read-out's K split) count
  arrivals in ticket words at the head of a workspace.  A word that is not zero when a launch
  begins would make a workgroup believe it is the last arriver before the others have stored
  (value 1: nothing could detect that) or nobody believe it (value 7, 0xFFFFFFFF).  Every call
  therefore zeroes the tickets on its stream in front of the kernel (snnqp.h):
  here every ticket of the cached workspace is poisoned between launches, and the next launch is
  the oracle's, bit for bit, with nothing reported."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  assert ops.device_status() == 0
  # the fused head, column split (B = 40: 20 tiles of two samples, two workgroups each)
  T, B, K, N1, N2 = 20, 40, 512, 512, 110
  c = cases.dense_net_case(True, T=T, B=B, K=K, hidden=N1, out=N2)
  p = c["vars"]["params"]
  x = np.ascontiguousarray(np.swapaxes(c["x"], 0, 1))
  e = oracle.dense2_forward(x, qweight_of(oracle, p["QuantDense_0"], 8), qweight_of(oracle, p["QuantDense_1"], 8),
                            mode="int")
  w1 = _weight(p["QuantDense_0"], 8, dev, transposed=True)
  w2 = _weight(p["QuantDense_1"], 8, dev, transposed=True)
  xd = _t(x, dev)
  assert int(L.lib().snnqp_dense_head_workspace_bytes(T, B, N1)) > 0, "this shape is meant to split"

  def head():
    return _np(ops.dense_head_forward(xd, w1, K, N1, _mslif(), w2, N2, _mslif(), group=10)[0])
  np.testing.assert_array_equal(head(), e["logits"])
  ws = ops._dense_ws[(torch.device(dev).index, torch.cuda.current_stream(dev).cuda_stream)]
  for poison in (1, 7, -1):
    ws[:4096].view(torch.int32).fill_(poison)
    np.testing.assert_array_equal(head(), e["logits"], err_msg="tickets poisoned with %d" % poison)
    assert ops.device_status() == 0
  # the read-out's K split (four workgroups per tile at this batch)
  T, B, K, N = 20, 64, 32768, 110
  c = cases.dense_block_case(T=T, B=B, K=K, N=N, bits=4, p=0.9)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  xin = ops.pack_bits(_t(c["x"], dev))
  eu, es = oracle.dense_block(c["x"], qweight_of(oracle, c["leaf"], c["bits"]), None, "int")
  for poison in (0, 1, 3, -1):
    ws = ops._dense_ws.get((torch.device(dev).index, torch.cuda.current_stream(dev).cuda_stream))
    if ws is not None:
      ws[:4096].view(torch.int32).fill_(poison)
    u, s = ops.dense_lif_forward(xin, w, K, N, _mslif(), packed_out=True, impl=L.IMPL_AUTO)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg="tickets poisoned with %d" % poison)
    np.testing.assert_array_equal(_np(u), eu)
    assert ops.device_status() == 0


def test_two_live_captures_on_two_streams_own_their_workspaces(dev, oracle):
  """Two CapturedApply objects of the split dense head, alive together and replayed on two
  streams at once, many times: each graph's hand-over workspace was allocated inside its own
  capture (the graph's private pool), so the two never share tickets or raster slabs, and an
  eager launch in between uses a third one (the workspace belongs to the
  capture, not to a (device, stream-handle) table)."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  ca = cases.dense_net_case(True, T=20, B=48, K=512, hidden=512)
  ea = cases.dense_net_expected(oracle, ca)
  rng = np.random.Generator(np.random.PCG64(99))
  xb_np = (rng.random(ca["x"].shape) < 0.12).astype(np.uint8)
  cb = dict(ca, x=xb_np)
  eb = cases.dense_net_expected(oracle, cb)
  assert not np.array_equal(ea["logits"], eb["logits"])
  model = models.DenseSNN(num_classes=11, config=syn.make_config(bits=8, prune_percentage=0.5, hidden=512))
  variables = nn.tree_from_numpy(ca["vars"], dev)
  xa, xb = _t(ca["x"], dev), _t(xb_np, dev)
  capa = nn.capture(model, variables, xa, trgt=None, train=False, rng=None)
  capb = nn.capture(model, variables, xb, trgt=None, train=False, rng=None)
  sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
  torch.cuda.synchronize()
  for it in range(50):
    with torch.cuda.stream(sa):
      la = capa()[0]
    with torch.cuda.stream(sb):
      lb = capb()[0]
    if it % 10 == 3:                   # an eager launch on the default stream in between
      (le, _) = model.apply(variables, xa, trgt=None, train=False, rng=None)
      np.testing.assert_array_equal(_np(le), ea["logits"])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(la), ea["logits"], err_msg="replay %d, graph a" % it)
    np.testing.assert_array_equal(_np(lb), eb["logits"], err_msg="replay %d, graph b" % it)
  assert ops.device_status() == 0
  capa.close()
  capb.close()


def _float_weight(leaf, bits, dev):
  """The float32 fake-quantised * mask kernel of a leaf (what stands by behind an integer launch)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import packing
  from snnquantprune_amd.quant import QuantDesc
  a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
  desc = QuantDesc(L.Q_DUQ, bits, a, c, float(2 ** (bits - 1) - 1), c)
  mask = leaf.get("prune_0", {}).get("mask")
  return packing.PackedKernel(_t(leaf["kernel"], dev), desc, None if mask is None else _t(mask, dev)).float_weight()


@pytest.mark.parametrize("shape", [(20, 256, 2048, 512, 110), (6, 5, 208, 200, 70), (64, 3, 96, 300, 30),
                                   (9, 131, 1040, 384, 110), (20, 3000, 96, 300, 30)],
                         ids=["c2", "small", "longest_t", "ragged_batch", "full_grid_unsplit"])
def test_dense_head_on_float32_rows(dev, oracle, shape):
  """The reference's own input format into the fused head: float32 rows [T, B, K]
  (flax_qdense.py:67 casts every input to float32), staged IN PLACE by the kernel and checked on
  the device (VERDICT r04 #1a).  Integer-valued rows -- spikes, counts up to 255, -0.0 -- give the
  integer contract bit for bit, with the status word of the launch clear; one value that is not
  an integer (0.5), one above 255 (300.0), a negative one, a NaN: the predicated float32 launches
  redo the head -- first block as the fmaf chain of the oracle's `fseq` mode on the fake-quantised
  kernel, second block (its input is a spike raster whatever produced it) on the integers, vote --
  and the rasters and logits are the oracle's for THAT contract.  Nothing is read back in between."""
  from snnquantprune_amd import ops
  T, B, K, N1, N2 = shape
  c = cases.dense_net_case(True, T=T, B=B, K=K, hidden=N1, out=N2)
  p = c["vars"]["params"]
  x = np.ascontiguousarray(np.swapaxes(c["x"], 0, 1)).astype(F32)           # [T, B, K]
  rng = np.random.Generator(np.random.PCG64(K + 1))
  x[rng.random(x.shape) < 0.003] = 3.0
  x[0, 0, :3] = (255.0, 128.0, -0.0)
  q1, q2 = qweight_of(oracle, p["QuantDense_0"], 8), qweight_of(oracle, p["QuantDense_1"], 8)
  w1 = _weight(p["QuantDense_0"], 8, dev, transposed=True)
  w2 = _weight(p["QuantDense_1"], 8, dev, transposed=True)
  fb = ops.FloatFallback(_float_weight(p["QuantDense_0"], 8, dev))

  def run(xn, time_major=True):
    xd = _t(xn if time_major else np.ascontiguousarray(np.swapaxes(xn, 0, 1)), dev)
    return ops.dense_head_forward(xd, w1, K, N1, _mslif(), w2, N2, _mslif(), group=10, want_s1=True,
                                  want_s2=True, time_major=time_major, fallback=fb)
  e = oracle.dense2_forward(x, q1, q2, mode="int")
  for tm in (True, False):
    logits, s1, s2 = run(x, tm)
    np.testing.assert_array_equal(_np(s1), packbits_lastaxis(e["s1"].astype(np.uint8)))
    np.testing.assert_array_equal(_np(s2), packbits_lastaxis(e["s2"].astype(np.uint8)))
    np.testing.assert_array_equal(_np(logits), e["logits"])
  for bad, where in ((0.5, (T - 1, B - 1, K - 1)), (300.0, (0, 0, 5)), (float("nan"), (T // 2, B // 2, K // 2)),
                     (-1.0, (0, B - 1, 0))):
    xb = x.copy()
    xb[where] = bad
    logits, s1, s2 = run(xb)
    if np.isnan(bad):
      continue            # (a NaN current: only that it runs; the oracle's rasters would be of NaN potentials)
    _, es1 = oracle.dense_block(xb, q1, None, "fseq")
    _, es2 = oracle.dense_block(es1, q2, None, "int")
    np.testing.assert_array_equal(_np(s1), packbits_lastaxis(es1.astype(np.uint8)), err_msg=str(bad))
    np.testing.assert_array_equal(_np(s2), packbits_lastaxis(es2.astype(np.uint8)), err_msg=str(bad))
    np.testing.assert_array_equal(_np(logits), oracle.vote(es2), err_msg=str(bad))
  assert ops.device_status() == 0


@pytest.mark.parametrize("shape", [(20, 37, 2048, 512), (10, 3, 96, 160), (7, 70, 400, 384)],
                         ids=["c2_layer1", "small", "three_column_blocks"])
def test_dense_wide_block_on_float32_rows(dev, oracle, shape):
  """The stand-alone wide dense block (N > 128) on float32 rows staged in place, with a carried-in
  state and the returned potentials: integer-valued rows bit-exact on the integer contract, a
  non-integer value anywhere flips the whole block to the float32 contract."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B, K, N = shape
  c = cases.dense_block_case(T=T, B=B, K=K, N=N, bits=8, p=0.5, counts=True)
  qw = qweight_of(oracle, c["leaf"], 8)
  w = _weight(c["leaf"], 8, dev, transposed=True)
  fb = ops.FloatFallback(_float_weight(c["leaf"], 8, dev))
  x = c["x"].astype(F32)
  eu, es = oracle.dense_block(x, qw, None, "int", u0=c["u0"])
  u, s = ops.dense_lif_forward(_t(x, dev), w, K, N, _mslif(), u0=_t(c["u0"], dev), packed_out=True,
                               impl=L.IMPL_AUTO, fallback=fb)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  xb = x.copy()
  xb[T - 1, B - 1, K - 3] = 1.25
  fu, fs = oracle.dense_block(xb, qw, None, "fseq", u0=c["u0"])
  u, s = ops.dense_lif_forward(_t(xb, dev), w, K, N, _mslif(), u0=_t(c["u0"], dev), packed_out=True,
                               impl=L.IMPL_AUTO, fallback=fb)
  np.testing.assert_array_equal(_np(s), packbits_lastaxis(fs))
  np.testing.assert_array_equal(_np(u), fu)
  assert ops.device_status() == 0


@pytest.mark.parametrize("hw,pool", [(24, 2), (21, 1), (16, 2)], ids=["24_pool", "21_clipped", "16_pool"])
def test_event_layer_stages_float32_frames(dev, oracle, hw, pool):
  """conv0 on float32 frames (flax_qconv.py:101: the reference's input format), staged in place:
  binary frames, count frames with a hot pixel (255) -- through the per-channel tables, the shared
  table and the general path alike -- equal the oracle's integer contract bit for bit, with and
  without the 2x2 pool, on images that clip patches; a value that is not an integer in [0, 255]
  sends the whole block to the float32 kernel (the oracle's `fseq` mode), pool included."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  T, B = 9, 5
  c = cases.conv_block_case(T=T, B=B, hw=hw, cin=2, seed=961, gain=4.0)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  fb = ops.FloatFallback(_float_weight(c["leaf"], c["bits"], dev))
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(hw, hw, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  rng = np.random.Generator(np.random.PCG64(hw))
  binary = (rng.random((T, B, hw, hw, 2)) < 0.1).astype(F32)
  counts = np.minimum(rng.poisson(0.4, (T, B, hw, hw, 2)), 255).astype(F32)
  counts[3, 2, 7, 11, 1] = 255.0
  counts[0, 0, 0, 0, 0] = -0.0

  def run(xn, hint):
    return ops.conv_lif_forward(_t(xn, dev), g, w, nrn, bn=bn, want_u=True, packed_out=True, pool=pool,
                                impl=L.IMPL_AUTO, x_max=hint, fallback=fb)
  for name, x in (("binary", binary), ("counts", counts)):
    eu, es = oracle.conv_block(x, qw, c["bn"], None, "int")
    exp = packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es)
    for hint in (1, 4, 255):
      u, s = run(x, hint)
      np.testing.assert_array_equal(_np(s), exp, err_msg="%s hint %d" % (name, hint))
      np.testing.assert_array_equal(_np(u), eu, err_msg="%s hint %d" % (name, hint))
  for bad in (0.5, 256.0, -2.0):
    xb = binary.copy()
    xb[T - 1, B - 1, hw - 1, hw - 2, 1] = bad
    fu, fs = oracle.conv_block(xb, qw, c["bn"], None, "fseq")
    u, s = run(xb, 1)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(oracle.max_pool_2x2(fs) if pool == 2 else fs),
                                  err_msg=str(bad))
    np.testing.assert_array_equal(_np(u), fu, err_msg=str(bad))
  assert ops.device_status() == 0


@pytest.mark.parametrize("hw,pool", [(24, 2), (13, 1), (16, 2)], ids=["24_pool", "13_clipped", "16_pool"])
def test_binary_frames_are_packed_in_front_of_the_event_layer(dev, oracle, hw, pool):
  """conv_lif_forward(binary_first=True): uint8 / float32 frames that are expected to be binary are
  packed to bits by one checked pass (snnqp_pack_frames_checked), the event layer runs its
  bit-packed variant, and a predicated launch on the frames as they are (snnqp_conv_lif_forward_pred)
  redoes the block iff a value was not 0 or 1.  Whatever the frames hold the result is the oracle's:
  binary frames (-0.0 included), frames with one count of 3 or a hot pixel (the redo runs, on the
  integer kernel), float32 frames with a value that is not an integer (the redo flags it and the
  float32 kernel behind it runs: the `fseq` contract); time- and batch-major; images that clip
  patches and rows that do not start on a word boundary; with a carried-in state."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  from tests.helpers import pack_ev1
  T, B = 7, 5
  c = cases.conv_block_case(T=T, B=B, hw=hw, cin=2, seed=961, gain=4.0)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  fb = ops.FloatFallback(_float_weight(c["leaf"], c["bits"], dev))
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(hw, hw, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  rng = np.random.Generator(np.random.PCG64(hw + 7))
  binary = (rng.random((T, B, hw, hw, 2)) < 0.1).astype(np.uint8)
  binary[:, :, hw - 1, hw - 1, :] = 1
  one3 = binary.copy(); one3[2, 1, 5, 6, 0] = 3
  hot = binary.copy(); hot[T - 1, B - 1, hw - 1, hw - 2, 1] = 255
  u0 = (rng.random((B, hw, hw, 128)) * 0.6).astype(F32)
  seen = torch.zeros(8, dtype=torch.int32, device=dev)

  def run(xt, carry=None, tm=True, hint=1, f=None):
    return ops.conv_lif_forward(xt, g, w, nrn, bn=bn, u0=None if carry is None else _t(carry, dev), want_u=True,
                                packed_out=True, pool=pool, impl=L.IMPL_AUTO, x_max=hint, x_seen=seen,
                                time_major=tm, fallback=f, binary_first=True)
  for name, x in (("binary", binary), ("one count of 3", one3), ("hot pixel", hot)):
    for carry in (None, u0):
      eu, es = oracle.conv_block(x, qw, c["bn"], None, "int", u0=carry)
      exp = packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es)
      for dt in (np.uint8, F32):
        xn = x.astype(dt)
        if dt is F32:
          xn[0, 0, 0, 0, 0] = -0.0 if x[0, 0, 0, 0, 0] == 0 else xn[0, 0, 0, 0, 0]
        tag = "%s %s carry %s" % (name, np.dtype(dt).name, carry is not None)
        seen.zero_()
        u, s = run(_t(xn, dev), carry, f=fb if dt is F32 else None)
        np.testing.assert_array_equal(_np(s), exp, err_msg=tag)
        np.testing.assert_array_equal(_np(u), eu, err_msg=tag)
        # the redo reports what it met; a launch that was not redone reports nothing
        assert int(seen[0].item()) == (0 if name == "binary" else int(x.max())), tag
        ub, sb = run(_t(np.ascontiguousarray(np.swapaxes(xn, 0, 1)), dev), carry, tm=False, f=fb if dt is F32 else None)
        np.testing.assert_array_equal(_np(sb), exp, err_msg=tag + " batch-major")
        np.testing.assert_array_equal(_np(ub), eu, err_msg=tag + " batch-major")
  # the checked pass by itself: the word is zero iff the packed frames are the tensor
  for x, dt, want in ((binary, np.uint8, 0), (one3, np.uint8, L.FLAG_GT_ONE), (binary, F32, 0),
                      (one3, F32, L.FLAG_GT_ONE)):
    pf, word = ops.pack_frames_checked(_t(x.astype(dt), dev))
    assert int(word.item()) == want
    if want == 0:
      np.testing.assert_array_equal(_np(pf.data).view(np.uint32), pack_ev1(x))
  frac = binary.astype(F32); frac[1, 2, 3, 4, 1] = 0.5
  assert int(ops.pack_frames_checked(_t(frac, dev))[1].item()) == L.FLAG_GT_ONE | L.FLAG_NOT_INTEGER
  # float32 frames with a value that is not an integer: pack flags it, the redo's own check flags
  # it, the float32 kernel behind both runs
  for bad in (0.5, 256.0, -2.0, float("nan")):
    xb = binary.astype(F32)
    xb[T - 1, B - 1, hw - 1, hw - 2, 1] = bad
    if np.isnan(bad):
      u, s = run(_t(xb, dev), f=fb)          # (no oracle for a NaN: the launch must not fault or hang)
      torch.cuda.synchronize()
      continue
    fu, fs = oracle.conv_block(xb, qw, c["bn"], None, "fseq")
    u, s = run(_t(xb, dev), f=fb)
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(oracle.max_pool_2x2(fs) if pool == 2 else fs), err_msg=str(bad))
    np.testing.assert_array_equal(_np(u), fu, err_msg=str(bad))
  assert ops.device_status() == 0


def test_speculative_packing_follows_what_the_device_has_seen(dev, oracle):
  """Through the model, as eval.py feeds it: uint8 / float32 frames take the packed path while every
  report from the device said "binary" (ops.CountHint.binary_so_far), count frames switch it off for
  good after their first batch -- whose results are right all the same -- and a captured step with
  the speculation inside replays to the eager result."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, synthetic as syn
  ops._count_hints.clear()
  c = cases.conv_net_case(T=5, B=4, hw=32, p=0.9, gains=(6.0, 9.0, 11.0, 16.0))
  e = cases.conv_net_expected(oracle, c)
  model = models.ConvDenseSNN(num_classes=11, config=syn.make_config(bits=4, prune_percentage=0.9))
  variables = nn.tree_from_numpy(c["vars"], dev)
  assert c["x"].max() == 1
  calls = []
  orig = ops.pack_frames_checked

  def spy(x):
    calls.append(tuple(x.shape))
    return orig(x)
  ops.pack_frames_checked = spy
  try:
    for dt in (torch.uint8, torch.float32):
      (logits, _), mut = model.apply(variables, _t(c["x"], dev).to(dt), trgt=None, train=False, rng=None,
                                     mutable=["intermediates"])
      np.testing.assert_array_equal(_np(logits), e["logits"])
      np.testing.assert_array_equal(_np(mut["intermediates"]["pool0"][0]), e["pool0_bits"])
    assert len(calls) == 2 and ops.count_hint(dev).binary_so_far()
    # a captured step with the speculation inside
    xs = _t(c["x"], dev)
    cap = nn.capture(model, variables, xs, trgt=None, train=False, rng=None)
    n_cap = len(calls)
    assert n_cap >= 3                                   # warm-up applies and the capture itself packed
    np.testing.assert_array_equal(_np(cap()[0]), e["logits"])
    # ... replayed on count frames: the predicated launch inside the graph redoes the block
    cnt = syn.poisson_counts(c["x"].shape, 0.3, seed=77)
    assert cnt.max() > 1
    ec = cases.conv_net_expected(oracle, dict(c, x=cnt))
    np.testing.assert_array_equal(_np(cap(_t(cnt, dev))[0]), ec["logits"])
    assert len(calls) == n_cap                          # (a replay calls nothing on the host)
    del cap
    # eager count frames: right on the first batch (speculation fails, the redo reports), and once
    # the report has arrived the frames go in as they are
    (logits, _) = model.apply(variables, _t(cnt, dev), trgt=None, train=False, rng=None)
    np.testing.assert_array_equal(_np(logits), ec["logits"])
    torch.cuda.synchronize()
    hint = ops.count_hint(dev)
    hint.current()
    assert hint.saw_counts and not hint.binary_so_far()
    n = len(calls)
    for x in (cnt, c["x"]):                             # binary frames afterwards: still as they are
      (logits, _) = model.apply(variables, _t(x, dev), trgt=None, train=False, rng=None)
      torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(logits), e["logits"])
    assert len(calls) == n
  finally:
    ops.pack_frames_checked = orig
    ops._count_hints.clear()
  assert ops.device_status() == 0


def test_float32_activations_through_the_narrowing_passes(dev, oracle):
  """float32 activations into the blocks that do not stage float32 themselves: a 128-channel conv
  block (spike bits packed by one checked device pass), a narrow dense block (uint8 rows by one
  checked pass).  Integer-valued tensors take the integer kernels, a tensor with a value that is
  not (0.5 among the spikes; 2.0 where spikes are expected: not a bit) takes the float32 kernel
  behind the same launch -- through SpikingBlock, as a model calls it, with nothing read back."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.flax_qdense import QuantDense
  from snnquantprune_amd.spiking_learning import SpikingBlock
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  # conv block, Cin = 128
  c = cases.conv_block_case(T=3, B=2, hw=8)
  qw = qweight_of(oracle, c["leaf"], c["bits"])
  blk = SpikingBlock(connection_fn=QuantConv(features=128, kernel_size=(3, 3), padding=((1, 1), (1, 1)),
                                             use_bias=False, config=cfg.quant, bits=4, g_scale=cfg.quant.g_scale),
                     neural_dynamics=cfg.neuron_dynamics(dtype=torch.float32),
                     norm_fn=nn.BatchNorm(use_running_average=True, momentum=0.9, epsilon=1e-5),
                     pool=2, return_state=True)
  # (a top-level SpikingBlock names its children after its fields, as flax does)
  variables = nn.tree_from_numpy({"params": {"connection_fn": c["leaf"],
                                             "norm_fn": {"scale": c["bn"]["scale"], "bias": c["bn"]["bias"]}},
                                  "batch_stats": {"norm_fn": {"mean": c["bn"]["mean"], "var": c["bn"]["var"]}}}, dev)
  x = c["x"].astype(F32)
  pos = np.arange(x.size).reshape(x.shape)
  for xin, mode in ((x, "int"), (np.where(pos == 77, 0.5, x).astype(F32), "fseq"),
                    (np.where(pos == 1234, 2.0, x).astype(F32), "fseq")):
    eu, es = oracle.conv_block(xin, qw, c["bn"], None, mode)
    u, s = blk.apply(variables, None, _t(xin, dev))
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(oracle.max_pool_2x2(es)), err_msg=mode)
    np.testing.assert_array_equal(_np(u), eu, err_msg=mode)
  # narrow dense block (N <= 128): uint8 rows
  d = cases.dense_block_case(T=6, B=5, K=208, N=70, bits=8, p=0.5, counts=True)
  qd = qweight_of(oracle, d["leaf"], 8)
  cfg8 = syn.make_config(bits=8, prune_percentage=0.5)
  dblk = SpikingBlock(connection_fn=QuantDense(70, use_bias=False, config=cfg8.quant, bits=8,
                                               g_scale=cfg8.quant.g_scale),
                      neural_dynamics=cfg8.neuron_dynamics(dtype=torch.float32), return_state=True)
  dvars = nn.tree_from_numpy({"params": {"connection_fn": d["leaf"]}}, dev)
  xd = d["x"].astype(F32)
  posd = np.arange(xd.size).reshape(xd.shape)
  for xin, mode in ((xd, "int"), (np.where(posd == 99, 7.5, xd).astype(F32), "fseq")):
    eu, es = oracle.dense_block(xin, qd, None, mode)
    u, s = dblk.apply(dvars, None, _t(xin, dev))
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg=mode)
    np.testing.assert_array_equal(_np(u), eu, err_msg=mode)
  assert ops.device_status() == 0


def test_float32_fallback_keeps_the_edge_neurons_of_an_odd_pooled_image(dev, oracle):
  """A pooled conv block on a 7 x 7 image with return_state: the 2x2 pool drops the last row and
  column of neurons from the raster (reduce_window without padding, examples/tcja/models.py:145-147),
  but the carry holds a membrane potential for every neuron.  Integer-valued float32 input (integer
  kernels) and input with a non-integer (the predicated float32 kernel, which walks 2x2 windows:
  ADVICE r05) must both return every potential, the edge ones included, bit-exact."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.spiking_learning import SpikingBlock
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  for hw, cin in ((7, 128), (5, 2), (9, 32)):
    c = cases.conv_block_case(T=3, B=2, hw=hw, cin=cin, seed=931 + hw)
    qw = qweight_of(oracle, c["leaf"], c["bits"])
    blk = SpikingBlock(connection_fn=QuantConv(features=128, kernel_size=(3, 3), padding=((1, 1), (1, 1)),
                                               use_bias=False, config=cfg.quant, bits=4, g_scale=cfg.quant.g_scale),
                       neural_dynamics=cfg.neuron_dynamics(dtype=torch.float32),
                       norm_fn=nn.BatchNorm(use_running_average=True, momentum=0.9, epsilon=1e-5),
                       pool=2, return_state=True)
    variables = nn.tree_from_numpy({"params": {"connection_fn": c["leaf"],
                                               "norm_fn": {"scale": c["bn"]["scale"], "bias": c["bn"]["bias"]}},
                                    "batch_stats": {"norm_fn": {"mean": c["bn"]["mean"], "var": c["bn"]["var"]}}}, dev)
    x = c["x"].astype(F32)
    pos = np.arange(x.size).reshape(x.shape)
    for xin, mode in ((x, "int"), (np.where(pos == 77, 0.5, x).astype(F32), "fseq")):
      eu, es = oracle.conv_block(xin, qw, c["bn"], None, mode)
      assert eu.shape == (2, hw, hw, 128) and np.abs(eu[:, hw - 1]).max() > 0      # the edge row is alive
      u, s = blk.apply(variables, None, _t(xin, dev))
      np.testing.assert_array_equal(_np(s), packbits_lastaxis(oracle.max_pool_2x2(es)), err_msg="%s %d" % (mode, hw))
      np.testing.assert_array_equal(_np(u), eu, err_msg="%s %d" % (mode, hw))
  assert ops.device_status() == 0


def test_float32_rows_the_wide_kernel_cannot_stage_are_narrowed_not_refused(dev, oracle):
  """float32 rows into a wide quantised dense block (more than 128 features) whose shape the wide
  kernel does not stage in place -- K beyond 65536 (int32 sums of x - 128), or a tensor whose rows
  lie beyond 32-bit byte offsets within a workgroup (2 GiB of float32) -- must take the narrowing
  pass like uint8 rows of the same shape take the other kernels, not raise (the host's
  `direct` predicate now mirrors dense_wide_unsupported, and an EUNSUPPORTED from a speculative
  launch narrows and relaunches)."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, synthetic as syn
  from snnquantprune_amd.flax_qdense import QuantDense
  from snnquantprune_amd.spiking_learning import SpikingBlock
  cfg8 = syn.make_config(bits=8, prune_percentage=0.5)

  def block(N):
    return SpikingBlock(connection_fn=QuantDense(N, use_bias=False, config=cfg8.quant, bits=8,
                                                 g_scale=cfg8.quant.g_scale),
                        neural_dynamics=cfg8.neuron_dynamics(dtype=torch.float32), return_state=True)
  # K = 65 552 (a multiple of 16 beyond 65 536), 160 features
  K, N = 65552, 160
  leaf = syn.quant_leaf((K, N), 30.0, 941, True, 0.5)
  qd = qweight_of(oracle, leaf, 8)
  x = syn.poisson_spikes((3, 4, K), 0.1, seed=942).astype(F32)
  pos = np.arange(x.size).reshape(x.shape)
  dvars = nn.tree_from_numpy({"params": {"connection_fn": leaf}}, dev)
  for xin, mode in ((x, "int"), (np.where(pos == 4321, 0.25, x).astype(F32), "fseq")):
    eu, es = oracle.dense_block(xin, qd, None, mode)
    u, s = block(N).apply(dvars, None, _t(xin, dev))
    np.testing.assert_array_equal(_np(s), packbits_lastaxis(es), err_msg=mode)
    np.testing.assert_array_equal(_np(u), eu, err_msg=mode)
  assert 0.01 < es.mean() < 0.6
  # rows beyond 32-bit byte offsets: [T = 64, B = 4200, K = 2048] float32 = 2.2 GB, time-major
  T, B, K, N = 64, 4200, 2048, 160
  leaf = syn.quant_leaf((K, N), 4.0, 943, True, 0.5)
  qd = qweight_of(oracle, leaf, 8)
  dvars = nn.tree_from_numpy({"params": {"connection_fn": leaf}}, dev)
  gen = torch.Generator(device=dev)
  gen.manual_seed(944)
  xu = (torch.rand((T, B, K), device=dev, generator=gen) < 0.1).to(torch.uint8)
  xf = xu.to(torch.float32)
  assert xf.numel() * 4 >= (1 << 31)
  u8, s8 = block(N).apply(dvars, None, xu)
  uf, sf = block(N).apply(dvars, None, xf)
  assert torch.equal(s8.bits, sf.bits) and torch.equal(u8, uf)
  # ... and against the oracle on the first and the last samples of the batch (samples are independent)
  for sl in (slice(0, 6), slice(B - 6, B)):
    eu, es = oracle.dense_block(_np(xu[:, sl]), qd, None, "int")
    np.testing.assert_array_equal(_np(sf.bits[:, sl]).view(np.uint32), packbits_lastaxis(es))
    np.testing.assert_array_equal(_np(uf[sl]), eu)
  # one non-integer value in the last sample: the float32 kernel redoes the block
  xf[T - 1, B - 1, 5] = 0.5
  uf2, sf2 = block(N).apply(dvars, None, xf)
  eu, es = oracle.dense_block(_np(xf[:, B - 2:]), qd, None, "fseq")
  np.testing.assert_array_equal(_np(sf2.bits[:, B - 2:]).view(np.uint32), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(uf2[B - 2:]), eu)
  del xf, xu
  torch.cuda.empty_cache()
  assert ops.device_status() == 0


@pytest.mark.parametrize("shape", [(3, 2, 8, 8, 128, 128), (2, 3, 5, 11, 64, 70), (1, 2, 16, 16, 96, 160),
                                   (2, 1, 4, 8, 32, 32), (3, 2, 8, 8, 128, 128, 8), (2, 3, 5, 11, 64, 70, 6)],
                         ids=["cextnet_conv_t_1", "ragged_image_and_outputs", "96_in_160_out", "one_patch",
                              "cextnet_conv_t_1_8bit", "ragged_6bit"])
def test_gated_conv_in_the_gint_form(dev, oracle, shape):
  """QuantConv 3x3 on gate x raster -- the conv block behind a TCJA gate
  (examples/tcja/models.py:95-97 -> :149-187) -- without multiplying the gate out: the nine taps of
  a channel summed as integers on the matrix pipe, the gates applied by one fmaf chain
  (snnqp_conv_gated_forward) against the oracle's `gint` contraction (gated_conv): currents
  bit-exact, on images that clip patches and output counts that are not multiples of 32; then the
  whole block (BatchNorm + LIF scan + 2x2 pool) through SpikingBlock against gated_conv_block."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, packing, synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.quant import QuantDesc
  from snnquantprune_amd.spiking_learning import SpikingBlock
  T, B, H, W, C, N = shape[:6]
  bits = shape[6] if len(shape) > 6 else 4      # beyond 4 bits: two e3m2 digits per code (code_max up to 127)
  leaf = syn.quant_leaf((3, 3, C, N), 5.0, 971, True, 0.9 if bits == 4 else 0.3)
  bp, bs = syn.bn_leaf(N, True, 972)
  bn = dict(mean=bs["mean"], var=bs["var"], scale=bp["scale"], bias=bp["bias"])
  qw = qweight_of(oracle, leaf, bits)
  rng = np.random.Generator(np.random.PCG64(H * W + C))
  s = (rng.random((T, B, H, W, C)) < 0.2).astype(np.uint8)
  gate = (1.0 / (1.0 + np.exp(-rng.standard_normal((T, B, C)) * 1.5))).astype(F32)
  a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
  pk = packing.PackedKernel(_t(leaf["kernel"], dev), QuantDesc(L.Q_DUQ, bits, a, c, float(2 ** (bits - 1) - 1), c),
                            _t(leaf["prune_0"]["mask"], dev))
  w = pk.int_weight()
  assert (w.code_max > 7) == (bits > 4)
  geom = ops.ConvGeom(H, W, C, N, 3, 3, (1, 1), ((1, 1), (1, 1)))
  x = ops.GatedSpikes(ops.pack_bits(_t(s, dev)), _t(gate, dev))
  y = ops.conv_gated_forward(x, geom, w, pk.gated_codes())
  ey = np.stack([oracle.gated_conv(s[t].astype(F32), gate[t], qw) for t in range(T)])
  np.testing.assert_array_equal(_np(y), ey)
  assert 0.3 < np.abs(ey).max() < 200.0
  # the block, as the model calls it
  cfg = syn.make_config(bits=bits, prune_percentage=0.9)
  for pool in (1, 2) if H % 2 == 0 and W % 2 == 0 else (1,):
    blk = SpikingBlock(connection_fn=QuantConv(features=N, kernel_size=(3, 3), padding=((1, 1), (1, 1)),
                                               use_bias=False, config=cfg.quant, bits=bits, g_scale=cfg.quant.g_scale),
                       neural_dynamics=cfg.neuron_dynamics(dtype=torch.float32),
                       norm_fn=nn.BatchNorm(use_running_average=True, momentum=0.9, epsilon=1e-5),
                       pool=pool, return_state=True)
    variables = nn.tree_from_numpy({"params": {"connection_fn": leaf, "norm_fn": {"scale": bn["scale"], "bias": bn["bias"]}},
                                    "batch_stats": {"norm_fn": {"mean": bn["mean"], "var": bn["var"]}}}, dev)
    u, sp = blk.apply(variables, None, x)
    eu, es = oracle.gated_conv_block(s.astype(F32), gate, qw, bn)
    np.testing.assert_array_equal(_np(sp), packbits_lastaxis(oracle.max_pool_2x2(es) if pool == 2 else es))
    np.testing.assert_array_equal(_np(u), eu)
  assert ops.device_status() == 0


@pytest.mark.parametrize("shape", [(3, 5, 4, 4, 128, 512), (2, 33, 2, 3, 64, 70), (1, 2, 1, 1, 32, 32),
                                   (2, 3, 4, 4, 96, 600), (3, 5, 4, 4, 128, 512, 8), (2, 33, 2, 3, 64, 70, 6)],
                         ids=["cextnet_dense1", "ragged_rows_and_outputs", "one_position", "two_output_groups",
                              "cextnet_dense1_8bit", "ragged_6bit"])
def test_gated_dense_in_the_gint_form(dev, oracle, shape):
  """QuantDense on the channel-major flattening of gate x raster -- the dense block behind the second
  TCJA gate (examples/tcja/models.py:97 -> :189-190 -> :200-216) -- without multiplying the gate
  out: the positions of a channel summed as integers on the matrix pipe, the gates applied by one
  fmaf chain (snnqp_dense_gated_forward) against the oracle's `gint` contraction (gated_dense):
  currents bit-exact, on row counts that are not multiples of 32 and output counts that are not
  multiples of 32 or run past one workgroup's 512; then the block through SpikingBlock against
  gated_dense_block, and the same block with the gate multiplied out against the fseq contract."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, packing, synthetic as syn
  from snnquantprune_amd.flax_qdense import QuantDense
  from snnquantprune_amd.quant import QuantDesc
  from snnquantprune_amd.spiking_learning import SpikingBlock
  T, B, H, W, C, N = shape[:6]
  bits = shape[6] if len(shape) > 6 else 4      # beyond 4 bits: two e3m2 digits per code
  K = C * H * W
  leaf = syn.quant_leaf((K, N), 5.0, 981, True, 0.9 if bits == 4 else 0.5)
  qw = qweight_of(oracle, leaf, bits)
  rng = np.random.Generator(np.random.PCG64(H * W + C + N))
  s = (rng.random((T, B, H, W, C)) < 0.3).astype(np.uint8)
  gate = (1.0 / (1.0 + np.exp(-rng.standard_normal((T, B, C)) * 1.5))).astype(F32)
  a, c = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
  pk = packing.PackedKernel(_t(leaf["kernel"], dev), QuantDesc(L.Q_DUQ, bits, a, c, float(2 ** (bits - 1) - 1), c),
                            _t(leaf["prune_0"]["mask"], dev))
  w = pk.int_weight()
  assert (w.code_max > 7) == (bits > 4)
  x = ops.GatedSpikes(ops.pack_bits(_t(s, dev)), _t(gate, dev)).flattened()
  assert tuple(x.shape) == (T, B, K)
  y = ops.dense_gated_forward(x, w, pk.gated_dense_codes(C, H * W))
  ey = np.stack([oracle.gated_dense(s[t].astype(F32), gate[t], qw) for t in range(T)])
  np.testing.assert_array_equal(_np(y), ey)
  assert 0.3 < np.abs(ey).max() < 2000.0
  # what the flattened product is, for every other consumer
  dense = (gate[:, :, None, None, :] * s).astype(F32).transpose(0, 1, 4, 2, 3).reshape(T, B, K)
  np.testing.assert_array_equal(_np(x.to_dense()), dense)
  # the block, as the model calls it
  cfg = syn.make_config(bits=bits, prune_percentage=0.9)
  blk = SpikingBlock(connection_fn=QuantDense(N, use_bias=False, config=cfg.quant, bits=bits, g_scale=cfg.quant.g_scale),
                     neural_dynamics=cfg.neuron_dynamics(dtype=torch.float32), return_state=True)
  variables = nn.tree_from_numpy({"params": {"connection_fn": leaf}}, dev)
  u, sp = blk.apply(variables, None, x)
  eu, es = oracle.gated_dense_block(s.astype(F32), gate, qw)
  np.testing.assert_array_equal(_np(sp), packbits_lastaxis(es))
  np.testing.assert_array_equal(_np(u), eu)
  with packing.integer_inputs(False):
    u2, sp2 = blk.apply(variables, None, _t(dense, dev))
  eu2, es2 = oracle.dense_block(dense, qw, None, "fseq")
  np.testing.assert_array_equal(_np(sp2), es2)
  np.testing.assert_array_equal(_np(u2), eu2)
  assert ops.device_status() == 0


def test_unsigned_quantisers_bit_exact(dev, oracle):
  """The reference's quantisers with sign=False (quant.py:331-358, :374-425, :439-469, :512-625:
  levels 0 .. 2^bits - 1, lower clip bound 0; DuQ keeps hard_tanh and only changes n_lv) --
  snnqp_quantize_ex against the oracle element by element, the module surface (init with
  calibration + apply) against the oracle's init + forward, and the unsigned level count."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops, quant
  rng = np.random.Generator(np.random.PCG64(20261))
  w = (rng.standard_normal((64, 96)) * 0.6).astype(F32)
  w[0, :8] = [0.0, -0.0, 0.45, -0.45, 0.9, 1.7, -3.0, 0.8999999]
  x = _t(w, dev)
  for bits in (2, 3, 4, 8, 11):
    for sign in (False, True):
      fq, _, _ = ops.quantize(L.Q_DUQ, x, None, bits, 0.73, 0.41, sign=sign)
      np.testing.assert_array_equal(_np(fq), oracle.duq_forward(w, 0.73, 0.41, bits, sign))
      fq, _, _ = ops.quantize(L.Q_UNIFORM_STATIC, x, None, bits, 0.9, sign=sign)
      np.testing.assert_array_equal(_np(fq), oracle.uniform_static_forward(w, 0.9, bits, sign))
      fq, _, _ = ops.quantize(L.Q_PARAMETRIC_D, x, None, bits, 0.05, sign=sign)
      np.testing.assert_array_equal(_np(fq), oracle.parametric_d_forward(w, 0.05, bits, sign))
      fq, _, _ = ops.quantize(L.Q_PARAMETRIC_D_XMAX, x, None, bits, 2 ** -4, 0.8, sign=sign)
      np.testing.assert_array_equal(_np(fq), oracle.parametric_d_xmax_forward(w, 2 ** -4, 0.8, sign))
    # unsigned: nothing below zero survives (DuQ excepted: hard_tanh clips at -1)
    fq, _, _ = ops.quantize(L.Q_UNIFORM_STATIC, x, None, bits, 0.9, sign=False)
    assert float(fq.min()) == 0.0 and len(np.unique(_np(fq))) <= 2 ** bits
  # int8 codes of the unsigned form exist up to 7 bits, are flagged beyond
  _, codes, fl = ops.quantize(L.Q_UNIFORM_STATIC, x, None, 7, 0.9, want_fq=False, want_codes=True, sign=False)
  assert int(fl.item()) == 0 and int(codes.max()) == 127 and int(codes.min()) == 0
  _, _, fl = ops.quantize(L.Q_UNIFORM_STATIC, x, None, 8, 0.9, want_fq=False, want_codes=True, sign=False)
  assert int(fl.item()) & L.FLAG_CODE_OVERFLOW
  # the modules: init (calibration) + apply, sign=False throughout
  data = np.abs(rng.uniform(-1, 1, size=(120, 50)) * 23).astype(F32)
  data[0, 0] = 23
  xd = _t(data, dev)
  for bits in (2, 4, 8):
    m = quant.uniform_static(bits)
    v = m.init(0, xd, sign=False)
    out = _np(m.apply(v, xd, sign=False))
    np.testing.assert_array_equal(out, oracle.uniform_static_forward(
        data, oracle.uniform_static_init(data, bits, False), bits, False))
    assert len(np.unique(out)) == 2 ** bits                      # 0 .. 2^bits - 1
    m = quant.parametric_d(bits)
    v = m.init(0, xd, sign=False)
    step = oracle.parametric_d_init(data, bits, False)
    assert float(v["quant_params"]["step_size"]) == float(step)
    np.testing.assert_array_equal(_np(m.apply(v, xd, sign=False)), oracle.parametric_d_forward(data, step, bits, False))
    m = quant.parametric_d_xmax(bits, init_fn=quant.max_init)
    v = m.init(0, xd, sign=False)
    d, xmax = oracle.parametric_d_xmax_init(data, bits, False, init_fn=oracle.max_init)
    np.testing.assert_array_equal(_np(m.apply(v, xd, sign=False)), oracle.parametric_d_xmax_forward(data, d, xmax, False))
    dq = quant.DuQ(bits)
    v = dq.init(0, xd, sign=False)
    v["params"]["a"] = torch.full((1,), 23.0, device=dev)
    v["params"]["c"] = torch.full((1,), 23.0, device=dev)
    out = _np(dq.apply(v, xd, sign=False))
    np.testing.assert_array_equal(out, oracle.duq_forward(data, 23.0, 23.0, bits, False))
    assert len(np.unique(out)) == 2 ** bits


def test_connections_alone_on_float32_inputs(dev, oracle, monkeypatch):
  """QuantDense / QuantConv called on their own (not inside a SpikingBlock) with the float32 tensors
  the reference hands them (flax_qdense.py:67, flax_qconv.py:101), quantised kernels: integer-valued
  inputs give the `int` contract, a non-integer or a value beyond 255 the `fseq` one -- decided on
  the device (narrowing pass + predicated float32 connection), with nothing read back."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.flax_qdense import QuantDense
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  rng = np.random.Generator(np.random.PCG64(4411))
  leaf = syn.quant_leaf((70, 45), 5.0, 31, True, 0.9)
  qw = qweight_of(oracle, leaf, 4)
  m = QuantDense(45, use_bias=False, config=cfg.quant, bits=4, g_scale=cfg.quant.g_scale)
  v = nn.tree_from_numpy({"params": leaf}, dev)
  x = rng.integers(0, 4, size=(3, 9, 70)).astype(F32)
  leafc = syn.quant_leaf((3, 3, 6, 20), 5.0, 32, True, 0.9)
  qc = qweight_of(oracle, leafc, 4)
  mc = QuantConv(20, (3, 3), strides=(2, 1), padding="SAME", use_bias=False, config=cfg.quant, bits=4,
                 g_scale=cfg.quant.g_scale)
  vc = nn.tree_from_numpy({"params": leafc}, dev)
  xc = rng.integers(0, 3, size=(4, 7, 9, 6)).astype(F32)

  def no_readback(*a, **k):
    raise AssertionError("host read-back inside a connection")
  m.apply(v, _t(x, dev)); mc.apply(vc, _t(xc, dev))          # pack once (the pack step reads scalars back)
  monkeypatch.setattr(torch.Tensor, "item", no_readback)
  monkeypatch.setattr(torch.Tensor, "tolist", no_readback)
  for bad, mode in ((None, "int"), (0.5, "fseq"), (300.0, "fseq"), (-1.0, "fseq")):
    xi, xci = x.copy(), xc.copy()
    if bad is not None:
      xi[1, 4, 33] = bad
      xci[2, 3, 5, 1] = bad
    y = m.apply(v, _t(xi, dev))
    yc = mc.apply(vc, _t(xci, dev))
    monkeypatch.undo()
    np.testing.assert_array_equal(_np(y), oracle.quant_dense(xi, qw, mode), err_msg=str(bad))
    np.testing.assert_array_equal(_np(yc), oracle.quant_conv(xci, qc, strides=(2, 1), padding="SAME", mode=mode),
                                  err_msg=str(bad))
    monkeypatch.setattr(torch.Tensor, "item", no_readback)
    monkeypatch.setattr(torch.Tensor, "tolist", no_readback)
  monkeypatch.undo()


@pytest.mark.parametrize("case", ["same_stride2", "explicit_dilated_grouped", "valid_depth_only"])
def test_quant_conv_3d(dev, oracle, case):
  """QuantConv with three spatial axes (flax_qconv.py:93-171: lax.conv_general_dilated over
  [B, D, H, W, C] with a DHWIO kernel) -- the connection alone in the `int` contract (uint8 counts,
  packed spikes, integer-valued float32) and the `fseq` contract (real-valued float32, and the same
  tensor with one non-integer: decided on the device), a batch-less input, then the whole
  SpikingBlock (BatchNorm + neuron over T, u0 carried) against the oracle's conv_block."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import ops, packing, synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.spiking_learning import SpikingBlock
  rng = np.random.Generator(np.random.PCG64(len(case) * 97))
  if case == "same_stride2":
    D, H, W, C, N, ks = 5, 6, 7, 4, 40, (3, 3, 3)
    kw = dict(strides=(2, 1, 2), padding="SAME")
  elif case == "explicit_dilated_grouped":
    D, H, W, C, N, ks = 4, 5, 6, 6, 64, (2, 3, 2)
    kw = dict(strides=(1, 2, 1), padding=((1, 0), (1, 2), (0, 1)), input_dilation=(1, 2, 1),
              kernel_dilation=(2, 1, 2), feature_group_count=2)
  else:
    D, H, W, C, N, ks = 6, 3, 3, 32, 33, (3, 1, 1)
    kw = dict(padding="VALID")
  G = kw.get("feature_group_count", 1)
  cfg = syn.make_config(bits=4, prune_percentage=0.9)
  leaf = syn.quant_leaf(ks + (C // G, N), 5.0, 77, True, 0.9)
  qw = qweight_of(oracle, leaf, 4)
  okw = dict(strides=kw.get("strides"), padding=kw["padding"], input_dilation=kw.get("input_dilation"),
             kernel_dilation=kw.get("kernel_dilation"), feature_group_count=G)
  m = QuantConv(N, ks, use_bias=False, config=cfg.quant, bits=4, g_scale=cfg.quant.g_scale, **kw)
  v = nn.tree_from_numpy({"params": leaf}, dev)
  B = 3
  xi = rng.integers(0, 3, size=(B, D, H, W, C)).astype(np.uint8)
  e_int = oracle.quant_conv(xi.astype(F32), qw, mode="int", **okw)
  assert m.out_shape(xi.shape) == e_int.shape
  np.testing.assert_array_equal(_np(m.apply(v, _t(xi, dev))), e_int)
  np.testing.assert_array_equal(_np(m.apply(v, _t(xi.astype(F32), dev))), e_int)
  np.testing.assert_array_equal(_np(m.apply(v, _t(xi[0].astype(F32), dev))), e_int[0])       # batch-less
  xb = (xi > 1).astype(np.uint8)
  np.testing.assert_array_equal(_np(m.apply(v, ops.pack_bits(_t(xb, dev)))),
                                oracle.quant_conv(xb.astype(F32), qw, mode="int", **okw))
  xr = rng.standard_normal((B, D, H, W, C)).astype(F32)
  with packing.integer_inputs(False):
    np.testing.assert_array_equal(_np(m.apply(v, _t(xr, dev))), oracle.quant_conv(xr, qw, mode="fseq", **okw))
  xn = xi.astype(F32)
  xn[1, 2, 1, 1, 0] = 0.25
  np.testing.assert_array_equal(_np(m.apply(v, _t(xn, dev))), oracle.quant_conv(xn, qw, mode="fseq", **okw))
  # the block
  T = 4
  bp, bs = syn.bn_leaf(N, True, 78)
  bn = dict(mean=bs["mean"], var=bs["var"], scale=bp["scale"], bias=bp["bias"])
  blk = SpikingBlock(connection_fn=QuantConv(N, ks, use_bias=False, config=cfg.quant, bits=4,
                                             g_scale=cfg.quant.g_scale, **kw),
                     neural_dynamics=cfg.neuron_dynamics(dtype=torch.float32),
                     norm_fn=nn.BatchNorm(use_running_average=True, momentum=0.9, epsilon=1e-5), return_state=True)
  variables = nn.tree_from_numpy({"params": {"connection_fn": leaf, "norm_fn": {"scale": bn["scale"], "bias": bn["bias"]}},
                                  "batch_stats": {"norm_fn": {"mean": bn["mean"], "var": bn["var"]}}}, dev)
  xs = rng.integers(0, 2, size=(T, B, D, H, W, C)).astype(np.uint8)
  u0 = (rng.standard_normal(e_int.shape) * 0.3).astype(F32)
  ckw = dict(padding=kw["padding"], strides=kw.get("strides"))
  if G == 1 and "input_dilation" not in kw:
    eu, es = oracle.conv_block(xs.astype(F32), qw, bn, None, "int", u0=u0, **ckw)
    for xin in (_t(xs, dev), _t(xs.astype(F32), dev), ops.pack_bits(_t(xs, dev))):
      u, sp = blk.apply(variables, _t(u0, dev), xin)
      got = _np(sp) if not isinstance(sp, ops.PackedSpikes) else _np(sp.to_dense())
      np.testing.assert_array_equal(got.astype(np.uint8), es.astype(np.uint8))
      np.testing.assert_array_equal(_np(u), eu)
    xf = xs.astype(F32)
    xf[2, 1, 0, 1, 1, 0] = 1.5
    eu, es = oracle.conv_block(xf, qw, bn, None, "fseq", u0=u0, **ckw)
    u, sp = blk.apply(variables, _t(u0, dev), _t(xf, dev))
    got = _np(sp) if not isinstance(sp, ops.PackedSpikes) else _np(sp.to_dense())
    np.testing.assert_array_equal(got.astype(np.uint8), es.astype(np.uint8))
    np.testing.assert_array_equal(_np(u), eu)
  assert ops.device_status() == 0


def test_per_channel_tables_cover_count_frames_and_report_a_short_stack(dev, oracle):
  """conv0's per-channel dequantisation tables (BatchNorm folded in) hold every channel over its OWN
  accumulator range, the four channels of an LDS bank stacked (snnqp_weight_t.ch_stack_max): event
  COUNTS up to 7 with 4-bit codes stay on them (round 4: binary frames only; counts went to the
  shared table and a separate BatchNorm).  Bit-exact against the oracle for hints 1..7 on count
  frames, with the caller's stack bound, without one (0: 8 x abs_sum_max) and -- understated -- the
  report through the device status word."""
  import dataclasses
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import ops
  c = cases.conv_block_case(T=7, B=3, hw=16, cin=2, seed=977, gain=4.0)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  assert w.ch_stack_max > 0 and w.ch_slots is not None and w.code_max <= 7
  bn, nrn = _bn(c["bn"], dev), _mslif()
  g = ops.ConvGeom(16, 16, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  rng = np.random.Generator(np.random.PCG64(6))
  for xmax in (1, 2, 4, 7):
    x = np.minimum(rng.poisson(0.5, (7, 3, 16, 16, 2)), xmax).astype(np.uint8)
    x[0, 0, 0, 0, 0] = xmax
    e = cases.conv_block_expected(oracle, dict(c, x=x))
    for wv in (w, dataclasses.replace(w, ch_stack_max=0)):
      for pool in (1, 2):
        u, s = ops.conv_lif_forward(_t(x, dev), g, wv, nrn, bn=bn, want_u=True, packed_out=True,
                                    pool=pool, impl=L.IMPL_MFMA, x_max=xmax)
        np.testing.assert_array_equal(_np(s), e["pooled_bits"] if pool == 2 else e["s_bits"], err_msg="x_max %d" % xmax)
        np.testing.assert_array_equal(_np(u), e["u"])
  assert ops.device_status() == 0
  bad = dataclasses.replace(w, ch_stack_max=max(1, w.ch_stack_max // 3))
  ops.conv_lif_forward(_t(x, dev), g, bad, nrn, bn=bn, want_u=False, packed_out=True, pool=2,
                       impl=L.IMPL_MFMA, x_max=7)
  torch.cuda.synchronize()
  assert ops.device_status(reset=True) == L.STATUS_BOUND


def test_scan_with_a_ragged_feature_count_replays_in_a_graph(dev, oracle):
  """snnqp_lif_forward assembles a packed raster whose feature count is no multiple of 32 with
  atomicOr into zeroed words; the zeroing is a kernel of the library (a kernel node under capture,
  kernels.h zero_words_async), so a captured scan starts from zeros at every replay."""
  from snnquantprune_amd import ops
  rng = np.random.Generator(np.random.PCG64(909))
  T, R, C = 6, 9, 40
  xs = torch.zeros((T, R, C), dtype=torch.float32, device=dev)
  nrn = _mslif()
  side = torch.cuda.Stream(device=dev)
  with torch.cuda.stream(side):
    ops.lif_forward(xs, nrn, want_u=True, packed_out=True)            # warm up outside the capture
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    u_g, s_g = ops.lif_forward(xs, nrn, want_u=True, packed_out=True)
  for k in range(4):
    x = (rng.standard_normal((T, R, C)) * 1.5).astype(F32)
    xs.copy_(_t(x, dev))
    g.replay()
    torch.cuda.synchronize()
    u = np.zeros((R, C), F32)
    es = []
    for t in range(T):
      u, sp = oracle.multi_step_lif(u, x[t])
      es.append(sp)
    np.testing.assert_array_equal(_np(s_g), packbits_lastaxis(np.stack(es)), err_msg="replay %d" % k)
    np.testing.assert_array_equal(_np(u_g), u)


def test_empty_batches_and_zero_timesteps(dev, oracle):
  """B = 0 and T = 0 through the fused blocks, the dense head, the gated connections, the 3-D
  convolution and the stand-alone connections: nothing is launched out of range, shapes are the
  reference's (a scan over zero steps returns the carry, spiking_learning.py:446-462), nothing is
  reported."""
  from snnquantprune_amd import _lib as L
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, ops, packing, synthetic as syn
  from snnquantprune_amd.flax_qconv import QuantConv
  from snnquantprune_amd.flax_qdense import QuantDense
  from snnquantprune_amd.quant import QuantDesc
  c = cases.conv_block_case(T=3, B=2, hw=8)
  w = _weight(c["leaf"], c["bits"], dev, transposed=True)
  g = ops.ConvGeom(8, 8, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
  nrn, bn = _mslif(), _bn(c["bn"], dev)
  for T, B in ((0, 2), (3, 0), (0, 0)):
    x = ops.pack_bits(torch.zeros((T, B, 8, 8, 128), dtype=torch.uint8, device=dev))
    u, s = ops.conv_lif_forward(x, g, w, nrn, bn=bn, want_u=True, packed_out=True, pool=2)
    assert tuple(s.shape) == (T, B, 4, 4, 128) and tuple(u.shape) == (B, 8, 8, 128)
    xf = torch.zeros((T, B, 8, 8, 128), dtype=torch.float32, device=dev)
    y = ops.conv_forward(xf.reshape(T * B, 8, 8, 128), g, packing.PackedKernel(
        _t(c["leaf"]["kernel"], dev), None, None).float_weight())
    assert tuple(y.shape) == (T * B, 8, 8, 128)
  # dense head and blocks
  cfg = syn.make_config(bits=8, prune_percentage=0.5, hidden=512)
  model = models.DenseSNN(num_classes=11, config=cfg)
  variables = nn.tree_from_numpy(syn.dense_net_variables(2048, 512, 110, True, 0.5), dev)
  for dt in (torch.uint8, torch.float32):
    logits, _ = model.apply(variables, torch.zeros((0, 20, 2048), dtype=dt, device=dev), trgt=None, train=False,
                            rng=None)
    assert tuple(logits.shape) == (0, 11)
  # gated connections
  leaf = syn.quant_leaf((3, 3, 128, 128), 5.0, 971, True, 0.9)
  a, cc = float(leaf["DuQ_0"]["a"][0]), float(leaf["DuQ_0"]["c"][0])
  pk = packing.PackedKernel(_t(leaf["kernel"], dev), QuantDesc(L.Q_DUQ, 4, a, cc, 7.0, cc), _t(leaf["prune_0"]["mask"], dev))
  xg = ops.GatedSpikes(ops.pack_bits(torch.zeros((2, 0, 8, 8, 128), dtype=torch.uint8, device=dev)),
                       torch.zeros((2, 0, 128), dtype=torch.float32, device=dev))
  assert tuple(ops.conv_gated_forward(xg, g, pk.int_weight(), pk.gated_codes()).shape) == (2, 0, 8, 8, 128)
  leafd = syn.quant_leaf((128 * 16, 512), 5.0, 981, True, 0.9)
  a, cc = float(leafd["DuQ_0"]["a"][0]), float(leafd["DuQ_0"]["c"][0])
  pkd = packing.PackedKernel(_t(leafd["kernel"], dev), QuantDesc(L.Q_DUQ, 4, a, cc, 7.0, cc), _t(leafd["prune_0"]["mask"], dev))
  xd = ops.GatedSpikes(ops.pack_bits(torch.zeros((2, 0, 4, 4, 128), dtype=torch.uint8, device=dev)),
                       torch.zeros((2, 0, 128), dtype=torch.float32, device=dev)).flattened()
  assert tuple(ops.dense_gated_forward(xd, pkd.int_weight(), pkd.gated_dense_codes(128, 16)).shape) == (2, 0, 512)
  # 3-D convolution and the stand-alone connections on float32 inputs
  cfg4 = syn.make_config(bits=4, prune_percentage=0.9)
  leaf3 = syn.quant_leaf((2, 3, 3, 4, 8), 5.0, 77, True, 0.9)
  m3 = QuantConv(8, (2, 3, 3), padding="SAME", use_bias=False, config=cfg4.quant, bits=4, g_scale=cfg4.quant.g_scale)
  v3 = nn.tree_from_numpy({"params": leaf3}, dev)
  assert tuple(m3.apply(v3, torch.zeros((0, 3, 5, 5, 4), dtype=torch.float32, device=dev)).shape) == (0, 3, 5, 5, 8)
  leaf2 = syn.quant_leaf((70, 45), 5.0, 31, True, 0.9)
  md = QuantDense(45, use_bias=False, config=cfg4.quant, bits=4, g_scale=cfg4.quant.g_scale)
  vd = nn.tree_from_numpy({"params": leaf2}, dev)
  assert tuple(md.apply(vd, torch.zeros((0, 70), dtype=torch.float32, device=dev)).shape) == (0, 45)
  # whole models on an empty batch (the tail of a sharded eval split)
  mc3 = models.ConvDenseSNN(num_classes=11, config=cfg4)
  vc3 = nn.tree_from_numpy(syn.conv_net_variables(prune_p=0.9, out=110), dev)
  for dt in (torch.uint8, torch.float32):
    out = mc3.apply(vc3, torch.zeros((0, 4, 128, 128, 2), dtype=dt, device=dev), trgt=None, train=False, rng=None)
    assert tuple(out[0].shape) == (0, 11)
  mcx = models.CextNet(num_classes=11, config=cfg4)
  vcx = nn.tree_from_numpy(syn.cextnet_variables(frames=4, prune_p=0.9), dev)
  out = mcx.apply(vcx, torch.zeros((0, 4, 128, 128, 2), dtype=torch.uint8, device=dev), trgt=None, train=False, rng=None)
  assert tuple(out[0].shape) == (0, 11)
  assert ops.device_status() == 0


@pytest.mark.parametrize("bits,prune", [(8, 0.3), (3, 0.7)], ids=["shipped_8bit_30pct", "shipped_3bit_70pct"])
def test_full_cextnet_at_the_shipped_bit_widths(dev, oracle, bits, prune):
  """The reference's TCJA model at the bit widths and pruning of its shipped configurations
  (examples/tcja/configs/prune_quant_joint.py:53,60 -- 8 bits, 30 %; prune_quant_seq.py:53,60 -- 3
  bits, 70 %) against the live oracle at 64 x 64: at 8 bits the blocks behind the gates run the
  two-digit form of the `gint` kernels (codes up to 127), the conv blocks the int8 instruction; every
  raster, both gates and the logits bit-exact."""
  from snnquantprune_amd import linen as nn
  from snnquantprune_amd import models, synthetic as syn
  c = cases.cextnet_case(bits=bits, p=prune)
  e = cases.cextnet_expected(oracle, c)
  model = models.CextNet(num_classes=11, config=syn.make_config(bits=bits, prune_percentage=prune))
  (logits, _), mut = model.apply(nn.tree_from_numpy(c["vars"], dev), _t(c["x"], dev), trgt=None, train=False,
                                 rng=None, mutable=["intermediates"])
  im = mut["intermediates"]
  for i in range(3):
    np.testing.assert_array_equal(_np(im["pool%d" % i][0]), e["pool%d_bits" % i])
  for i in range(2):
    np.testing.assert_array_equal(_np(im["tcja_gate_%d" % i][0]), e["gate%d" % i])
    s = im["conv_t_%d" % i][0]
    s = _np(s) if hasattr(s, "bits") else packbits_lastaxis(_np(s))
    np.testing.assert_array_equal(s, e["conv_t_%d_bits" % i])
  for key, name in (("dense1_out", "dense1_s"), ("dense2_out", "dense2_s")):
    d = im[key][0]
    d = d.to_dense() if hasattr(d, "to_dense") else d
    np.testing.assert_array_equal(_np(d).astype(np.uint8), e[name])
  np.testing.assert_array_equal(_np(logits), e["logits"])
  assert 0.005 < e["dense1_s"].mean() < 0.7
