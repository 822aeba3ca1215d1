#!/bin/bash
# Row-tile sweep of the fp6 read-out kernel on the headline shape (tuning knob SNNQP_DENSE_FP6_RT).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-s3k}; mkdir -p $O
for rt in 0 2 3 4 5; do
  SNNQP_DENSE_FP6_RT=$rt timeout -k 10 300 python bench.py --no-cpu-baseline --no-fed-leg --input ev1 --steps 10 --warmup 3 > $O/bench_rt$rt.json 2> $O/bench_rt$rt.err || echo "rt $rt failed"
done
python - $O <<'PY' | tee $O/summary.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_rt*.json")):
  d = json.loads(open(f).read().strip().splitlines()[-1])
  k = d["kernels"]["dense[32768->110]"]
  print(os.path.basename(f), "readout %.4f ms  %.0f GB/s" % (k["avg_ms"], k["hbm_gbs"]), round(d["value"]))
PY
