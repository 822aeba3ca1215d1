"""Diagnostic: builds libsnnqp with in-kernel clock stamps (SNNQP_PROBE=1), runs conv0 of
config C3 and prints the phase split of workgroup 0 (staging / compute / flush) and the
shader clock the chip sustains inside the kernel (delta s_memtime / delta s_memrealtime x
100 MHz).  Rebuild normally after.  usage: clock_probe.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SNNQP_PROBE"] = "1"
from snnquantprune_amd.csrc import build as b
b.build(force=True, verbose=False)
import numpy as np, torch
from snnquantprune_amd import _lib as L, ops, packing, synthetic as syn
from snnquantprune_amd.quant import QuantDesc
dev = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 20
CONV0 = True          # the probes live in the conv0 kernel
CIN, HW = (2, 128) if CONV0 else (128, 64)
leaf = syn.quant_leaf((3, 3, CIN, 128), 5.0, 1, True, 0.9)
a = float(leaf["DuQ_0"]["a"][0])
pk = packing.PackedKernel(torch.from_numpy(leaf["kernel"]).to(dev), QuantDesc(L.Q_DUQ, 4, a, a, 7.0, a),
                          torch.from_numpy(leaf["prune_0"]["mask"]).to(dev))
w = pk.int_weight_mfma(128)
x = (torch.rand((T, B, HW, HW, CIN), device=dev) < 0.15).to(torch.uint8)
if not CONV0:
  x = ops.pack_bits(x)
nrn = ops.Neuron(L.NEURON_MULTI_STEP_LIF, 2.0, 1.0, 0.0)
g = ops.ConvGeom(HW, HW, CIN, 128, 3, 3, (1, 1), ((1, 1), (1, 1)))
for _ in range(3):
  ops.conv_lif_forward(x, g, w, nrn, packed_out=True, pool=2, want_u=False, x_max=1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 5)()
L.lib().snnqp_debug_read_probe.argtypes = [ctypes.c_void_p]
rc = L.lib().snnqp_debug_read_probe(out)
print("rc", rc, "total cycles %d; staging %d (%.0f%%), compute %d (%.0f%%), flush %d (%.0f%%)" % (
    out[0], out[1], 100.0 * out[1] / out[0], out[2], 100.0 * out[2] / out[0], out[3],
    100.0 * out[3] / out[0]))
print("shader cycles", out[0], "realtime ticks (100 MHz)", out[4],
      "-> clock %.3f GHz, workgroup 0 ran %.3f ms" % (out[0] / out[4] * 0.1, out[4] / 1e5))

span = (ctypes.c_uint * 8192)()
L.lib().snnqp_debug_read_wg_span.argtypes = [ctypes.c_void_p]
L.lib().snnqp_debug_read_wg_span(span)
sp = np.array(span, dtype=np.int64).reshape(4096, 2)
sp = sp[(sp[:, 1] != 0)]
t0 = sp[:, 0].min()
st, en = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0          # microseconds
print("%d workgroups: start min/median/max %.0f / %.0f / %.0f us; end min/median/max %.0f / %.0f / %.0f us; "
      "duration min/median/max %.0f / %.0f / %.0f us" % (
          len(sp), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(),
          (en - st).min(), np.median(en - st), (en - st).max()))
for x in range(8):
  m = np.arange(len(sp)) % 8 == x
  print("  blockIdx.x %% 8 == %d: end median %.0f max %.0f us, duration median %.0f" % (
      x, np.median(en[m]), en[m].max(), np.median((en - st)[m])))
