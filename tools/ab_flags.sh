#!/bin/bash
# Diagnostic: A/B compile-time flags that reach both fused conv kernels (conv_tile.h) on one box.
# usage: [BENCH_ARGS="..."] tools/ab_flags.sh "<flags A>" "<flags B>" ...
cd "$(dirname "$0")/../snnquantprune_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $F $v -c conv3x3_u8c2.hip -o /tmp/ab_u8c2_$i.o 2>/dev/null &
  /opt/rocm/bin/hipcc $F $v -c conv3x3_bits.hip -o /tmp/ab_bits_$i.o 2>/dev/null &
done
wait
for round in 1 2; do
i=0
for v in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsnnqp.so api.o quantize.o spikes.o elementwise.o generic_block.o blocks.o /tmp/ab_u8c2_$i.o /tmp/ab_bits_$i.o dense_mfma.o fseq_gemm.o || exit 1
  echo "== [$v]"
  (cd ../.. && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})") || exit 1
done
done
