"""Registers, scratch and LDS of every kernel instance of one csrc file (device-only compile):
  python tools/kernel_regs.py dense_wide.hip [-DFOO=1 ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snnquantprune_amd.csrc import build as product
src = os.path.join(product.HERE, sys.argv[1])
with tempfile.TemporaryDirectory() as tmp:
  out = os.path.join(tmp, "k.s")
  subprocess.check_call(["/opt/rocm/bin/hipcc", *product.FLAGS, *sys.argv[2:], "--cuda-device-only", "-S", src, "-o", out])
  s = open(out).read()
  if os.environ.get("KEEP_S"):
    open("/tmp/k.s", "w").write(s)
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
  name, body = m.group(1), m.group(2)
  g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, None])[1]
  dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
  dn = re.sub(r"\(.*", "", dn).replace("snnqp::", "").replace("void ", "")
  print("%-60s vgpr+agpr %4s (acc at %4s) sgpr %3s scratch %5s lds %6s" % (
      dn[:60], g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"),
      g("private_segment_fixed_size"), g("group_segment_fixed_size")))
