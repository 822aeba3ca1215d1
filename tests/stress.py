"""Randomised shapes through the fused blocks: the MFMA kernels (work queue, clipped edge
patches, masked channel words, padded Cin, both operand formats, conv0 variants on uint8 and
on bit-packed frames; dense blocks on the int8 and the fp6 instruction, on bit-packed and on
uint8 rows) against the direct-form kernel on the same inputs, bit for bit."""
import numpy as np
import torch


def conv_block_random(dev, N, seed, verbose=False):
  """Runs N random geometries; returns the descriptions of the mismatching ones."""
  from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
  from snnquantprune_amd.quant import QuantDesc
  rng = np.random.Generator(np.random.PCG64(seed))
  failures = []
  for it in range(N):
    first = rng.random() < 0.35                      # a 2-channel event layer (conv0 kernel)
    cin = 2 if first else int(rng.integers(3, 129))
    cout = int(rng.choice([32, 64, 96, 100, 128, 160, 256, 300, 1056]))   # 1056: > 8 channel blocks, static patch walk
    H, W = int(rng.integers(3, 41)), int(rng.integers(3, 41))
    T, B = int(rng.integers(1, 10)), int(rng.integers(1, 21))
    bits = int(rng.choice([3, 4, 5, 8]))
    pool = int(rng.choice([1, 2]))
    if pool == 2:
      H, W = H + (H & 1), W + (W & 1)
    leaf = syn.quant_leaf((3, 3, cin, cout), float(rng.uniform(3, 7)), int(rng.integers(1 << 30)), True,
                          float(rng.choice([0.0, 0.5, 0.9])))
    a = float(leaf["DuQ_0"]["a"][0])
    Lq = float(2 ** (bits - 1) - 1)
    pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, bits, a, a, Lq, a),
                              torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
    w = pk.int_weight_mfma((cout + 31) // 32 * 32)
    if first:
      kind = rng.choice(["binary", "counts", "big"])
      lam = {"binary": 0.2, "counts": 0.6, "big": 20.0}[kind]
      x = torch.from_numpy(np.minimum(rng.poisson(lam, (T, B, H, W, cin)), 255).astype(np.uint8)).to(dev)
      if kind == "binary":
        x = x.clamp(max=1)
      xin, x_max = x, max(1, int(x.max().item()))
    else:
      x = torch.from_numpy((rng.random((T, B, H, W, cin)) < 0.2).astype(np.uint8)).to(dev)
      xin, x_max = ops.pack_bits(x), 1
    bn = ops.BnCoeffs(torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32)).to(dev),
                      torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev),
                      torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32)).to(dev))
    nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, float(rng.choice([2.0, 4.0, 3.0])), 1.0, float(rng.choice([0.0, 0.1])))
    g = ops.ConvGeom(H, W, cin, cout, 3, 3, (1, 1), ((1, 1), (1, 1)))
    u0 = None
    if rng.random() < 0.3:
      u0 = torch.from_numpy(rng.normal(0, 0.3, (B, H, W, cout)).astype(np.float32)).to(dev)
    tag = "cin %d cout %d %dx%d T %d B %d bits %d pool %d x_max %d u0 %s" % (
        cin, cout, H, W, T, B, bits, pool, x_max, u0 is not None)
    try:
      um, sm = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, u0=u0, packed_out=True, pool=pool,
                                    impl=L.IMPL_MFMA, x_max=x_max)
    except L.SnnqpError as e:
      if verbose:
        print("skip (%s): %s" % (tag, str(e)[:60]))
      continue
    ug, sg = ops.conv_lif_forward(xin, g, w, nrn, bn=bn, u0=u0, packed_out=True, pool=1,
                                  impl=L.IMPL_GENERIC, x_max=x_max)
    if pool == 2:
      sg = ops.maxpool2x2(sg)
    ok = torch.equal(sm.bits, sg.bits) and torch.equal(um, ug)
    if first and kind == "binary":          # the same launch on bit-packed (EV1) frames
      ue, se = ops.conv_lif_forward(ops.pack_frames(x, L.EV1), g, w, nrn, bn=bn, u0=u0, packed_out=True,
                                    pool=pool, impl=L.IMPL_MFMA, x_max=1)
      ok = ok and torch.equal(se.bits, sg.bits) and torch.equal(ue, ug)
      tag += " +ev1"
    if first:
      # round 6: the frames packed to bits by a checked pass in front of the event layer, the frames
      # as they are behind it (predicated): the same result whatever they hold, uint8 and float32
      for xt in (x, x.to(torch.float32)):
        fb = ops.FloatFallback(pk.float_weight()) if xt.dtype == torch.float32 else None
        ub, sb = ops.conv_lif_forward(xt, g, w, nrn, bn=bn, u0=u0, packed_out=True, pool=pool,
                                      impl=L.IMPL_AUTO, x_max=1, fallback=fb, binary_first=True)
        ok = ok and torch.equal(sb.bits, sg.bits) and torch.equal(ub, ug)
      tag += " +packed-first"
    if not ok:
      failures.append(tag)
    if verbose:
      print("%s  %s  rate %.3f" % ("ok  " if ok else "FAIL", tag, float(sg.to_dense().float().mean())))
  return failures


def dense_block_random(dev, N, seed, verbose=False):
  """Random dense SpikingBlocks (K 1..6000 not a multiple of anything, any N, T up to 60,
  3..8-bit codes, every neuron form): the MFMA kernel against the direct-form kernel."""
  from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
  from snnquantprune_amd.quant import QuantDesc
  rng = np.random.Generator(np.random.PCG64(seed))
  failures = []
  for it in range(N):
    K = int(rng.choice([rng.integers(1, 200), rng.integers(200, 6000), 2048, 512, 784, 16 * rng.integers(1, 300)]))
    n_out = int(rng.choice([rng.integers(1, 40), 110, 100, 512, rng.integers(100, 600)]))
    T, B = int(rng.integers(1, 61)), int(rng.integers(1, 40))
    bits = int(rng.choice([3, 4, 5, 8]))
    leaf = syn.quant_leaf((K, n_out), float(rng.uniform(2, 8)), int(rng.integers(1 << 30)), True,
                          float(rng.choice([0.0, 0.5, 0.9])))
    a = float(leaf["DuQ_0"]["a"][0])
    pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev),
                              QuantDesc(L.Q_DUQ, bits, a, a, float(2 ** (bits - 1) - 1), a),
                              torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
    w = pk.int_weight_mfma((n_out + 31) // 32 * 32)
    x = ops.pack_bits(torch.from_numpy((rng.random((T, B, K)) < rng.uniform(0.02, 0.4)).astype(np.uint8)).to(dev))
    kind = rng.choice(["ms2", "ms3", "plif", "lif", "vr"])
    if kind == "plif":
      nrn = ops.Neuron(L.NEURON_PARAMETRIC_LEAKY_IF, float(1.0 / (1.0 + np.exp(0.35))), 1.0, 0.0)
    elif kind == "lif":
      dec = torch.from_numpy((1.0 / (1.0 + np.exp(-rng.uniform(-1, 2, n_out)))).astype(np.float32)).to(dev)
      nrn = ops.Neuron(L.NEURON_LIF, 1.0, 1.0, 0.0, decay=dec)
    else:
      nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 3.0 if kind == "ms3" else 2.0, 1.0, 0.1 if kind == "vr" else 0.0)
    u0 = None
    if rng.random() < 0.3:
      u0 = torch.from_numpy(rng.normal(0, 0.3, (B, n_out)).astype(np.float32)).to(dev)
    tag = "K %d N %d T %d B %d bits %d %s u0 %s" % (K, n_out, T, B, bits, kind, u0 is not None)
    try:
      um, sm = ops.dense_lif_forward(x, w, K, n_out, nrn, u0=u0, packed_out=True, impl=L.IMPL_MFMA)
    except L.SnnqpError as e:
      if verbose:
        print("skip (%s): %s" % (tag, str(e)[:60]))
      continue
    ug, sg = ops.dense_lif_forward(x, w, K, n_out, nrn, u0=u0, packed_out=True, impl=L.IMPL_GENERIC)
    ok = torch.equal(sm.bits, sg.bits) and torch.equal(um, ug)
    if K % 16 == 0 and w.wt is not None and w.col_sum is not None:
      # uint8 rows read in place (x - 128 operand), binary rows and rows with counts up to 255
      xu = x.to_dense().to(torch.uint8)
      uu, su = ops.dense_lif_forward(xu, w, K, n_out, nrn, u0=u0, packed_out=True, impl=L.IMPL_MFMA)
      ok = ok and torch.equal(su.bits, sg.bits) and torch.equal(uu, ug)
      xc = torch.from_numpy(np.minimum(rng.poisson(0.3, (T, B, K)) * rng.integers(1, 60), 255).astype(np.uint8)).to(dev)
      uc, sc = ops.dense_lif_forward(xc, w, K, n_out, nrn, u0=u0, packed_out=True, impl=L.IMPL_MFMA)
      ugc, sgc = ops.dense_lif_forward(xc, w, K, n_out, nrn, u0=u0, packed_out=True, impl=L.IMPL_GENERIC)
      ok = ok and torch.equal(sc.bits, sgc.bits) and torch.equal(uc, ugc)
      tag += " +u8"
    if not ok:
      failures.append(tag)
    if verbose:
      print("%s  %s  rate %.3f" % ("ok  " if ok else "FAIL", tag, float(sg.to_dense().float().mean())))
  return failures
