#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s6p; mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_profile.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/summary.json $O/pmc_summary.json; cp gpurun_out/pmc/traffic.json $O/pmc_traffic.json
PMC_INPUT=u8 bash tools/pmc_profile.sh --model dense --batch 256 --bits 8 --prune 0.5 > $O/pmc_c2_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_c2_traffic.json
PMC_INPUT=f32 bash tools/pmc_profile.sh --model dense --batch 4096 --bits 8 --prune 0.5 --input f32 > $O/pmc_c2_f32_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_c2_b4096_f32_traffic.json
PMC_INPUT=f32 bash tools/pmc_profile.sh --input f32 > $O/pmc_f32_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_f32_traffic.json; cp gpurun_out/pmc/summary.txt $O/pmc_f32_summary.txt
PMC_INPUT=u8 bash tools/pmc_profile.sh --input u8 > $O/pmc_u8_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/pmc_u8_traffic.json
cat $O/pmc_traffic.json $O/pmc_c2_traffic.json $O/pmc_c2_b4096_f32_traffic.json $O/pmc_f32_traffic.json $O/pmc_u8_traffic.json
