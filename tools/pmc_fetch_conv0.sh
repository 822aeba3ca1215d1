#!/bin/bash
# HBM read / write bytes of the event-layer kernel alone (tools/conv_scaling.py, B = 1024, T = 20)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r02n; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp; ONLY_CONV0=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python $GRAFT_REPO_ROOT/tools/conv_scaling.py 1024 > $O/$c.log 2>&1 )
done
python - $O <<'PY'
import csv, glob, sys
for c in ("FETCH_SIZE", "WRITE_SIZE"):
  for p in glob.glob(sys.argv[1] + "/%s/*/*counter_collection.csv" % c):
    per = {}
    for r in csv.DictReader(open(p)):
      if "u8c2" in r["Kernel_Name"]:
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0) + float(r["Counter_Value"])
    print(c, [round(v * 1024 / 1e9 * (2 if c == "FETCH_SIZE" else 1), 3) for v in per.values()], "GB per launch (FETCH doubled)")
PY
