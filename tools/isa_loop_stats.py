"""Instruction mix of the MFMA-bearing basic blocks of one kernel in an assembly dump
(tools/kernel_regs.py with KEEP_S=1 leaves /tmp/k.s):  python tools/isa_loop_stats.py /tmp/k.s 'ILi4ELi2ELi1ELb1E'"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
start = None
for m in re.finditer(r"^(_Z\S+):\s*;.*$", s, re.M):
  if pat in m.group(1):
    start = m
    break
end = s.index("s_endpgm", start.end())
body = s[start.end():end]
print(start.group(1)[:100], "lines", body.count("\n"))
blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
for b in blocks:
  n = len(re.findall(r"v_mfma", b))
  if n >= int(sys.argv[3]) if len(sys.argv) > 3 else n > 10:
    name = b.split("\n")[0]
    wc = Counter(re.findall(r"s_waitcnt[^\n]*", b))
    print(name, "mfma", n, "lines", b.count("\n"))
    print("  waits", wc.most_common(14))
    print("  vmem loads", len(re.findall(r"global_load|buffer_load", b)), "stores", len(re.findall(r"global_store", b)),
          "ds_read", len(re.findall(r"ds_read", b)), "ds_write", len(re.findall(r"ds_write", b)),
          "valu", len(re.findall(r"\n\s+v_(?!mfma|accvgpr)", b)), "salu", len(re.findall(r"\n\s+s_(?!waitcnt|barrier|nop)", b)),
          "s_nop", len(re.findall(r"s_nop", b)), "scratch", len(re.findall(r"scratch_", b)),
          "accvgpr moves", len(re.findall(r"v_accvgpr", b)), "smem", len(re.findall(r"s_load", b)))
