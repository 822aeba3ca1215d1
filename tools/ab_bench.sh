#!/bin/bash
# A/B of diagnostic builds under the bench: tools/ab_bench.sh OUT name1 name2 ...   (name = product | diag_build/<name>)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
for n in "$@"; do
  if [ "$n" = product ]; then lib=""; else lib=$GRAFT_REPO_ROOT/diag_build/$n/libsnnqp.so; fi
  SNNQP_DIAG_LIB=$lib python bench.py --allow-diag --no-cpu-baseline --no-fed-leg --steps 6 --warmup 2 $BENCH_ARGS > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; continue; }
  python - "$n" "$O/$n.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-22s %8.0f/s %7.3f ms  %s" % (sys.argv[1], d["value"], d["ms_per_step"], " ".join("%.3f" % v["avg_ms"] for v in d["kernels"].values())))
PY
done
