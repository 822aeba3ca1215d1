#!/bin/bash
# Round 3: threshold-form micro-benchmark, config C2 eager and as a graph, the default line.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3e}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 120 ./build/ubench/threshold_forms > $O/threshold_forms.txt; cat $O/threshold_forms.txt
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name failed"; tail -5 $O/bench_$name.err; }; }
run c2 --model dense --batch 256 --bits 8 --prune 0.5 --steps 200 --warmup 20
run c2_graph --model dense --batch 256 --bits 8 --prune 0.5 --steps 200 --warmup 20 --graph
run c2_b4096 --model dense --batch 4096 --bits 8 --prune 0.5 --steps 100 --warmup 10
run c1 --model dense --batch 32 --frames 10 --bits -1 --prune -1 --steps 100 --warmup 10
run default --steps 8 --warmup 3
run strong1 --steps 4 --warmup 2 --scaling strong --global-batch 8192 --input ev1
python - $O <<'PY' | tee $O/configs.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-10s %9d samples/s %8.4f ms/step %s fallbacks=%s" % (
        os.path.basename(f)[6:-5], round(d["value"]), d["ms_per_step"],
        {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 4) for k, v in d["kernels"].items()},
        d.get("fallbacks")))
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
