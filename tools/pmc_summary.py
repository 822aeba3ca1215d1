"""Per-kernel sums of the rocprofv3 --pmc passes written by tools/pmc_profile.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
  with open(path) as f:
    for row in csv.DictReader(f):
      k = row["Kernel_Name"][:60]
      if "snnqp" not in k:
        continue
      c = row["Counter_Name"]
      agg[k][c] += float(row["Counter_Value"])
      calls[k][c].add(row["Dispatch_Id"])
for k in sorted(agg):
  print(k)
  for c in sorted(agg[k]):
    n = max(len(calls[k][c]), 1)
    print("   %-28s total %.6g   per launch %.6g   (%d launches)" % (c, agg[k][c], agg[k][c] / n, n))
