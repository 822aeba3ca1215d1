#!/bin/bash
# GPU batch 1 of round 2: refresh the micro-benchmarks, product bench, diagnostic variants of the
# bits kernel (bench per-kernel times), LDS counters per variant, the GPU test suite.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
( $H tools/ubench/mfma_overlap_classes.hip -o /tmp/ovl && timeout -k 10 120 /tmp/ovl ) > $O/ubench_overlap.txt 2>&1
( $H tools/ubench/mfma_valu_2waves.hip -o /tmp/v2w && timeout -k 10 120 /tmp/v2w ) > $O/ubench_valu2w.txt 2>&1
echo "ubench done"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_product.json 2> $O/bench_product.err || exit 1
echo "product bench done"
for v in notab bnmul nolut nolut2 nostage nolut2_bnmul; do
  SNNQP_DIAG_LIB=build/diag/$v/libsnnqp.so timeout -k 10 300 python bench.py --allow-diag --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || echo "variant $v failed"
  echo "variant $v done"
done
python - <<'PY' > gpurun_out/r02a/summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r02a/bench_*.json")):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["value"]), {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
  except Exception as e:
    print(os.path.basename(f), "unreadable", e)
PY
cat $O/summary.txt
pmc() {  # name, lib ("" = product)
  name=$1; lib=$2
  ( cd /tmp; SNNQP_DIAG_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$name/lds -- python $GRAFT_REPO_ROOT/bench.py --allow-diag --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$name.json 2> $GRAFT_REPO_ROOT/$O/pmc_$name.err )
  python tools/pmc_summary.py $O/pmc_$name --json $O/pmc_${name}_summary.json > $O/pmc_${name}_summary.txt 2>&1
  echo "pmc $name done"
}
pmc product ""
pmc notab $GRAFT_REPO_ROOT/build/diag/notab/libsnnqp.so
pmc nostage $GRAFT_REPO_ROOT/build/diag/nostage/libsnnqp.so
pmc nolut $GRAFT_REPO_ROOT/build/diag/nolut/libsnnqp.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
