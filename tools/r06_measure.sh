#!/bin/bash
# Round-6 measurements on the GPU box (one gpurun call): the bench line with its CPU
# baseline, rocprofv3 --kernel-trace --stats of the same command, the PMC passes
# (tools/pmc_profile.sh), the other configurations.  Everything lands in gpurun_out/${1:-s6m}/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s6m}; mkdir -p $O
export TMPDIR=/tmp
# the driver's command: ONE compact line on stdout (bench_line.json), the full record in the detail file
timeout -k 10 600 python bench.py --detail $O/bench.json > $O/bench_line.json 2> $O/bench.err || exit 1
echo "bench done: $(python -c "import json;d=json.load(open('$O/bench_line.json'));print(round(d['value']), d['roofline']['frac'], len(open('$O/bench_line.json').read()), 'bytes')")"
# (the legs launch the same device functions on other shapes: the trace of the headline command runs without them)
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-legs --no-fed-leg --full-line --detail /dev/null > $O/bench_under_rocprof.json 2> $O/prof.err )
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); python tools/trim_rocprof_stats.py $f $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
# ... and the same trace with conv1 / conv2 (two launches of one device function, same grid) apart:
# 5 warm-up + 10 timed + 10 event-carrying steps = 25 steps of the process
f=$(ls $O/prof/*/*kernel_trace.csv | head -1); python tools/kernel_trace_stats.py $f $O/kernel_trace_stats.csv --steps 25
bash tools/pmc_profile.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/summary.json $O/pmc_summary.json; cp gpurun_out/pmc/traffic.json $O/pmc_traffic.json; cp gpurun_out/pmc/summary.txt $O/pmc_summary.txt
bash tools/pmc_variant.sh ${1:-s6m} product > $O/pmc_lds.log 2>&1
run() { name=$1; shift; timeout -k 10 300 python bench.py --full-line --detail /dev/null --no-cpu-baseline --steps 6 --warmup 4 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "$name failed"; }
run random_bn --random-bn
run c5 --frames 50 --batch 512 --layer-bits 2,4,2,4 --prune 0.95 --classes 10
run 8bit --bits 8 --prune 0.3
run 8bit_counts --bits 8 --prune 0.3 --counts
run counts --counts
run counts_ev4 --counts --input ev4
run cextnet --model cextnet
run cextnet_8bit --model cextnet --bits 8 --prune 0.3
run f32 --input f32
run u8 --input u8
run u8_graph --input u8 --graph
run ev1_fed --input ev1 --feed host
run u8_fed --input u8 --feed host --no-fed-leg
run ev4_fed --input ev4 --feed host
run strong8192 --scaling strong --global-batch 8192 --steps 4 --warmup 2
run c1 --model dense --batch 32 --frames 10 --bits -1 --prune -1 --steps 100 --warmup 10
run c2 --model dense --batch 256 --bits 8 --prune 0.5 --steps 200 --warmup 20
run c2_graph --model dense --batch 256 --bits 8 --prune 0.5 --steps 200 --warmup 20 --graph
run c2_b4096 --model dense --batch 4096 --bits 8 --prune 0.5 --steps 100 --warmup 10
run c2_b4096_graph --model dense --batch 4096 --bits 8 --prune 0.5 --steps 100 --warmup 10 --graph
run c2_b4096_f32 --model dense --batch 4096 --bits 8 --prune 0.5 --input f32 --steps 100 --warmup 10
run c2_b4096_f32_graph --model dense --batch 4096 --bits 8 --prune 0.5 --input f32 --steps 100 --warmup 10 --graph
run c2_f32_graph --model dense --batch 256 --bits 8 --prune 0.5 --input f32 --steps 200 --warmup 20 --graph
run f32_graph --input f32 --graph
run counts_f32 --counts --input f32
run cextnet_graph --model cextnet --graph
run counts_random_bn --counts --random-bn
# the C2 kernel under rocprofv3 and its counters
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python $GRAFT_REPO_ROOT/bench.py --full-line --detail /dev/null --no-cpu-baseline --model dense --batch 256 --bits 8 --prune 0.5 --steps 200 --warmup 20 > $O/bench_c2_under_rocprof.json 2> $O/prof_c2.err )
f=$(ls $O/prof_c2/*/*kernel_stats.csv | head -1); python tools/trim_rocprof_stats.py $f $O/kernel_stats_c2.csv; head -4 $O/kernel_stats_c2.csv
PMC_INPUT=u8 bash tools/pmc_profile.sh --model dense --batch 256 --bits 8 --prune 0.5 > $O/pmc_c2_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_c2_traffic.json; cp gpurun_out/pmc/summary.json $O/pmc_c2_fetch_summary.json
BENCH_ARGS="--model dense --batch 256 --bits 8 --prune 0.5" bash tools/pmc_variant.sh ${1:-s6m}_c2 product > $O/pmc_c2.log 2>&1; cp gpurun_out/${1:-s6m}_c2/pmc_product_summary.json $O/pmc_c2_summary.json
BENCH_ARGS="--model dense --batch 4096 --bits 8 --prune 0.5" bash tools/pmc_variant.sh ${1:-s6m}_c2b product > $O/pmc_c2b.log 2>&1; cp gpurun_out/${1:-s6m}_c2b/pmc_product_summary.json $O/pmc_c2_b4096_summary.json
# the dense head on float32 rows at B = 4096: HBM bytes by the counters
PMC_INPUT=f32 bash tools/pmc_profile.sh --model dense --batch 4096 --bits 8 --prune 0.5 --input f32 > $O/pmc_c2_f32_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_c2_b4096_f32_traffic.json
PMC_INPUT=f32 bash tools/pmc_profile.sh --input f32 > $O/pmc_f32_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_f32_traffic.json
PMC_INPUT=u8 bash tools/pmc_profile.sh --input u8 > $O/pmc_u8_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_u8_traffic.json
# (tools/pmc_profile.sh clears gpurun_out/pmc at every call since round 6: the first version of this
#  script let the later calls' summaries mix with the earlier runs' files -- the committed r06 traffic
#  files were re-made by tools/r06_pmc_traffic.sh)
python - $O <<'PY' | tee $O/configs.txt
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-26s %8d samples/s  %7.3f ms/step  frac %.3f  %s" % (os.path.basename(f)[:-5], round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], {k.split("[")[1][:-1] if "[" in k else k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}))
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
