"""Per-kernel launch statistics from a rocprofv3 --kernel-trace CSV, with launches of ONE device
function split by where they stand in a step: conv1 and conv2 are two launches of
conv3x3_bits_kernel with the same persistent grid, which rocprofv3 --stats averages into one row.

  python tools/kernel_trace_stats.py <..._kernel_trace.csv> <out.csv> [--steps N]

A kernel launched k x N times in the trace (N = --steps: warm-up + timed + profiled steps of the
bench command, i.e. every launch of the process) is split into k slots, launch i belonging to slot
i mod k; rows: name, slot, grid, workgroup, calls, total / average / min / max ns, registers,
LDS, scratch.  Library kernels only (snnqp::)."""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
rows = defaultdict(list)
with open(src) as f:
  for r in csv.DictReader(f):
    name = r["Kernel_Name"]
    if "snnqp::" not in name:
      continue
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:100]
    key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    rows[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                      r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
out = []
for (name, gx, gy, wg), ls in rows.items():
  ls.sort()
  k = 1
  if steps and len(ls) % steps == 0:
    k = len(ls) // steps
  for slot in range(k):
    d = [l[1] for i, l in enumerate(ls) if i % k == slot]
    out.append((sum(d), name, slot if k > 1 else "", "%sx%s" % (gx, gy), wg, len(d), sum(d), sum(d) / len(d), min(d), max(d),
                ls[0][2], ls[0][3], ls[0][4], ls[0][5]))
out.sort(reverse=True)
with open(dst, "w", newline="") as g:
  w = csv.writer(g)
  w.writerow(["Name", "Slot", "Grid", "Workgroup", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "VGPR",
              "AccumVGPR", "LDS", "Scratch"])
  for o in out:
    w.writerow([o[1], o[2], o[3], o[4], o[5], o[6], "%.1f" % o[7], o[8], o[9], o[10], o[11], o[12], o[13]])
for o in out[:12]:
  print("%-70s slot %-2s grid %-10s calls %4d avg %10.1f ns  min %9d max %9d" % (o[1][:70], o[2], o[3], o[5], o[7], o[8], o[9]))
