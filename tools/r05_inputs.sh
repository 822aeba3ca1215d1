#!/bin/bash
# conv0 on every input format, same box: tools/r05_inputs.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
for cfg in "ev1:--input ev1" "u8:--input u8" "f32:--input f32" "ev4:--input ev4" "counts_u8:--counts --input u8" "counts_f32:--counts --input f32" "counts_ev4:--counts --input ev4"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-fed-leg $flags > $O/bench_$name.json 2> $O/bench_$name.err || echo "$name failed"
done
python - $O <<'PY' | tee $O/summary.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-14s %7d " % (os.path.basename(f)[6:-5], round(d["value"])), {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
