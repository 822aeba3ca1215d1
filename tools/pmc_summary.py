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
launches = {}
for (tag, k, c), vals in rows.items():
  per_disp = defaultdict(float)
  for d, v in vals:
    per_disp[d] += v                                  # counters come per XCC / SE: sum
  disp = sorted(per_disp)
  launches[k] = len(disp)
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

# HBM bytes per launch of the bench.py kernels, keyed by bench.py's kernel tags:
# FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts half of the bytes of wide
# coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section) -> doubled.
if "--traffic" in sys.argv:
  tags = {}
  for k, c in summary.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
      continue
    tag = None
    if "conv3x3_u8c2_kernel" in k:
      tag = "conv3x3[128x128x2->128]"
    elif "conv3x3_bits_kernel" in k or "conv3x3_fp6_kernel" in k:
      tag = "conv3x3[64x64x128->128]" if k.endswith("#0") else "conv3x3[32x32x128->128]"
    elif "dense_wide_kernel" in k:
      tag = "dense_head[2048->512->110]"
    elif "dense_mfma_kernel" in k or "dense_fp6_kernel" in k:
      tag = "dense[32768->110]"
    elif "pack_ev1" in k and launches.get(k.split("#")[0], 0) >= steps:
      # the checked pass in front of the event layer (round 6): once per step -- a single launch is
      # the bench's own preparation of a bit-packed resident batch, not part of a step
      tag = "conv3x3[128x128x2->128]"
    if tag:
      # (byte / float32 frames: the pass, the bit-packed launch and the predicated one add up)
      tags[tag] = tags.get(tag, 0) + int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
  with open(sys.argv[sys.argv.index("--traffic") + 1], "w") as f:
    fmt = sys.argv[sys.argv.index("--input-format") + 1] if "--input-format" in sys.argv else "u8"
    json.dump({"input": fmt,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py "
                         "--steps 1 --warmup 1` (tools/pmc_profile.sh), bytes = 2 * FETCH_SIZE "
                         "KiB + WRITE_SIZE KiB", "bytes_per_launch": tags}, f, indent=1, sort_keys=True)
