#!/bin/bash
# Diagnostic: rebuild one kernel file with extra flags and print bench.py's per-kernel times.
# usage: [BENCH_ARGS="--bits 8 --prune 0.3"] tools/abl_bench.sh <file.hip> "<flags A>" "<flags B>" ...
cd "$(dirname "$0")/../snnquantprune_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
SRC=$1; shift
for v in "$@"; do
  /opt/rocm/bin/hipcc $F $v -c $SRC -o ${SRC%.hip}.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsnnqp.so api.o quantize.o spikes.o elementwise.o generic_block.o blocks.o conv3x3_u8c2.o conv3x3_bits.o dense_mfma.o fseq_gemm.o || exit 1
  echo "== [$v]"
  (cd ../.. && timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})") || exit 1
done
