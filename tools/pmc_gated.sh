#!/bin/bash
# rocprofv3 counter passes of the gated connection alone (tools/gated_time.py):  tools/pmc_gated.sh OUTDIR [diag lib]
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
export SNNQP_DIAG_LIB=$2
pass() { p=$1; shift
  ( cd /tmp; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$p -- python3 $GRAFT_REPO_ROOT/tools/gated_time.py > $O/pmc_$p.log 2>&1 )
}
pass w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass i SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA
pass l SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES
python3 tools/pmc_summary.py $O/pmc --steps 13 > $O/pmc_summary.txt 2>&1
grep -A30 "conv_gated_kernel" $O/pmc_summary.txt | head -40
