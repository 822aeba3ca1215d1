#!/bin/bash
# bench.py per-kernel times of diagnostic variants: tools/variants_bench.sh OUTDIR name1 name2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  if [ "$v" = product ]; then lib=""; else lib=diag_build/$v/libsnnqp.so; fi
  SNNQP_DIAG_LIB=$lib timeout -k 10 300 python bench.py --allow-diag --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $O/bench_$v.json 2> $O/bench_$v.err || echo "variant $v failed"
done
python - $O <<'PY' | tee $O/summary.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s %7d " % (os.path.basename(f)[6:-5], round(d["value"])), {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
