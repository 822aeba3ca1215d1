#!/bin/bash
# Round 3, first GPU call: the GPU suite, then the bench line in its input formats and feeds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03a}; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 3 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "$name failed"; }
run u8
run ev1 --input ev1
run ev4 --input ev4
run u8_fed --feed host --no-fed-leg
run ev1_fed --input ev1 --feed host
run ev4_fed --input ev4 --feed host
run counts --counts
run cextnet_ev1 --model cextnet --input ev1
python - $O <<'PY' | tee $O/configs.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fed = d.get("fed") or {}
    print("%-14s %8d samples/s  %7.3f ms/step  frac %.3f  %s  feed=%s fed_leg=%s" % (
        os.path.basename(f)[6:-5], round(d["value"]), d["ms_per_step"], d["roofline"]["frac"],
        {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()},
        d.get("feed"), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in fed.items() if k in ("samples_per_s_per_gpu", "h2d_GBps", "vs_resident")}))
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
