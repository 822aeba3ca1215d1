"""Probe: do the event layer (vector-bound) and a 128-channel conv block (matrix + LDS bound) run
faster side by side on the same CUs than one after the other?  Two streams, half a batch each:
  python tools/corun_probe.py            (product library: persistent grids fill the CUs, so the
                                          two launches serialise -- the reference point)
  SNNQP_DIAG_LIB=diag_build/<occ1>/libsnnqp.so python tools/corun_probe.py
                                         (a diagnostic build that launches ONE workgroup per CU:
                                          the two kernels can then share every CU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc

dev = torch.device("cuda:0")
T, B = 20, 512
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)


def weight(shape, gain, seed):
  leaf = syn.quant_leaf(shape, gain, seed, True, 0.9)
  a = float(leaf["DuQ_0"]["a"][0])
  pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                            torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
  return pk.int_weight_mfma(128)


w0, w1 = weight((3, 3, 2, 128), 4.0, 11), weight((3, 3, 128, 128), 5.0, 12)
g0 = ops.ConvGeom(128, 128, 2, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
g1 = ops.ConvGeom(64, 64, 128, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
x0 = ops.pack_frames((torch.rand((T, B, 128, 128, 2), device=dev) < 0.095).to(torch.uint8), L.EV1)
x1 = ops.pack_bits((torch.rand((T, B, 64, 64, 128), device=dev) < 0.15).to(torch.uint8))
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def conv0():
  return ops.conv_lif_forward(x0, g0, w0, nrn, want_u=False, packed_out=True, pool=2, x_max=1)


def conv1():
  return ops.conv_lif_forward(x1, g1, w1, nrn, want_u=False, packed_out=True, pool=2, x_max=1)


def timed(fn, reps=6):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def serial():
  conv0(); conv1()


def parallel():
  cur = torch.cuda.current_stream(dev)
  sa.wait_stream(cur); sb.wait_stream(cur)
  with torch.cuda.stream(sa):
    conv1()
  with torch.cuda.stream(sb):
    conv0()
  cur.wait_stream(sa); cur.wait_stream(sb)


print("library:", os.environ.get("SNNQP_DIAG_LIB", "product"))
print("conv0 alone   %.3f ms" % timed(conv0))
print("conv1 alone   %.3f ms" % timed(conv1))
print("one after the other %.3f ms" % timed(serial))
print("two streams         %.3f ms" % timed(parallel))
