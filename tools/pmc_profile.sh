#!/bin/bash
# Collects rocprofv3 PMC counters for one bench.py step (B = 1024) in separate
# passes (SQ has 8 slots, FETCH_SIZE / WRITE_SIZE do not fit one TCC pass) and
# writes per-kernel summaries to gpurun_out/pmc/.  Run on the GPU box:
#   bash tools/pmc_profile.sh [extra bench args]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
rm -rf $OUT          # (rocprofv3 adds files beside those of an earlier call: the summary would mix the runs)
mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-fed-leg $@"
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- \
      python bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "pass $name done"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
python tools/pmc_summary.py $OUT --steps 3 --json $OUT/summary.json --traffic $OUT/traffic.json --input-format ${PMC_INPUT:-ev1} > $OUT/summary.txt
cat $OUT/summary.txt
