"""Copies one tools/r0N_measure.sh result directory into profiles/ under the round's prefix:
  python tools/refresh_profiles.py gpurun_out/s3x r03
(gpurun_out/ is scratch; profiles/ is what the record cites)."""
import glob
import json
import os
import shutil
import sys

src, pre = sys.argv[1], sys.argv[2]
names = {"bench.json": "bench.json", "bench_under_rocprof.json": "bench_under_rocprof.json",
         "kernel_stats.csv": "kernel_stats.csv", "pmc_summary.json": "pmc_summary.json",
         "pmc_traffic.json": "pmc_traffic.json", "pmc_product_summary.json": "pmc_lds_summary.json"}
names.update({n: n for n in ("kernel_stats_c2.csv", "pmc_c2_summary.json", "pmc_c2_b4096_summary.json", "pmc_c2_traffic.json",
                             "bench_line.json", "kernel_trace_stats.csv", "pmc_c2_b4096_f32_traffic.json", "pmc_f32_traffic.json", "pmc_u8_traffic.json")
              if os.path.exists(os.path.join(src, n))})
for a, b in names.items():
  shutil.copy(os.path.join(src, a), os.path.join("profiles", "%s_%s" % (pre, b)))
lines = []
for f in sorted(glob.glob(os.path.join(src, "bench*.json"))):
  name = os.path.basename(f)
  if name not in names:
    shutil.copy(f, os.path.join("profiles", "%s_%s" % (pre, name)))
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    if "kernels" not in d and "kernel_ms" in d:       # the compact line (round 6)
      d["kernels"] = {k: {"avg_ms": v} for k, v in d["kernel_ms"].items()}
    kern = {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}
    lines.append("%-26s %9d samples/s  %8.3f ms/step  %s" % (name[:-5], round(d["value"]), d["ms_per_step"], kern))
  except Exception as e:       # a leg that failed stays visible
    lines.append("%s unreadable: %s" % (name, e))
open(os.path.join("profiles", "%s_configs.txt" % pre), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
