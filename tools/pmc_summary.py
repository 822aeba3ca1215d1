"""Per-kernel sums of the rocprofv3 --pmc passes written by tools/pmc_profile.sh.

Kernels launched several times per step under one name (the conv3x3 bits kernel
runs conv1 and conv2) are split by their position in the step: the i-th launch
of a name belongs to slot i % launches_per_step.  Usage:
  python tools/pmc_summary.py <dir> [--json out.json] [--steps N]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2

rows = defaultdict(list)          # (pass dir, kernel, counter) -> [(dispatch id, value)]
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
  tag = path.split(os.sep)[-3]
  with open(path) as f:
    for row in csv.DictReader(f):
      k = row["Kernel_Name"]
      if "snnqp" not in k:
        continue
      k = k.split("(")[0].replace("void ", "")
      rows[(tag, k, row["Counter_Name"])].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))

summary = defaultdict(dict)
for (tag, k, c), vals in rows.items():
  per_disp = defaultdict(float)
  for d, v in vals:
    per_disp[d] += v                                  # counters come per XCC / SE: sum
  disp = sorted(per_disp)
  per_step = max(len(disp) // steps, 1)
  for slot in range(per_step):
    sel = [per_disp[d] for i, d in enumerate(disp) if i % per_step == slot]
    name = k if per_step == 1 else "%s#%d" % (k, slot)
    summary[name][c] = sum(sel) / len(sel)
for k in sorted(summary):
  print(k)
  for c in sorted(summary[k]):
    print("   %-28s per launch %.6g" % (c, summary[k][c]))
if out_json:
  with open(out_json, "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)
