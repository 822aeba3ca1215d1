#!/bin/bash
# Diagnostic: A/B two versions of one kernel source on the same box.
# usage: [BENCH_ARGS="--bits 8 --prune 0.3"] tools/ab_src.sh <object name, e.g. conv3x3_bits> <a.hip> <b.hip> ...   (paths inside csrc/)
cd "$(dirname "$0")/../snnquantprune_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
OBJ=$1; shift
for round in 1 2; do
for v in "$@"; do
  [ -f "${v%.hip}.ab.o" ] || /opt/rocm/bin/hipcc $F -c $v -o ${v%.hip}.ab.o 2>/dev/null || exit 1
  cp ${v%.hip}.ab.o $OBJ.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsnnqp.so api.o quantize.o spikes.o elementwise.o generic_block.o blocks.o conv3x3_u8c2.o conv3x3_bits.o dense_mfma.o fseq_gemm.o || exit 1
  echo "== [$v]"
  (cd ../.. && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})") || exit 1
done
done
