#!/bin/bash
# rocprofv3 counter passes of one bench step for the product library or a diagnostic variant:
#   tools/pmc_variant.sh OUTDIR NAME [bench args...]      (NAME = product | diag_build/<NAME>)
# Separate --pmc passes with --kernel-trace only (no other trace domain), per the guide.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/$1; name=$2; shift 2; mkdir -p $O
if [ "$name" = product ]; then lib=""; else lib=$GRAFT_REPO_ROOT/diag_build/$name/libsnnqp.so; fi
export TMPDIR=/tmp
pass() {  # pass name, counters...
  p=$1; shift
  ( cd /tmp; SNNQP_DIAG_LIB=$lib rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name/$p -- \
      python $GRAFT_REPO_ROOT/bench.py --allow-diag --steps 1 --warmup 1 --no-cpu-baseline --no-fed-leg $BENCH_ARGS > $O/pmc_${name}_$p.json 2> $O/pmc_${name}_$p.err )
}
pass w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass i SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA
pass l SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM
python tools/pmc_summary.py $O/pmc_$name --steps 3 --json $O/pmc_${name}_summary.json > $O/pmc_${name}_summary.txt 2>&1
grep -A26 "bits_kernel.*#0\|bits_kernel[^#]*$" $O/pmc_${name}_summary.txt | head -30
