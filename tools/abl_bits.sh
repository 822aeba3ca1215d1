#!/bin/bash
# Diagnostic: rebuild conv3x3_bits.hip with extra flags and time conv1 (C3, B = 1024).
cd "$(dirname "$0")/../snnquantprune_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
for v in "$@"; do
  /opt/rocm/bin/hipcc $F $v -c conv3x3_bits.hip -o conv3x3_bits.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsnnqp.so api.o quantize.o spikes.o elementwise.o generic_block.o blocks.o conv3x3_u8c2.o conv3x3_bits.o dense_mfma.o fseq_gemm.o || exit 1
  echo "== [$v]"
  ONLY_CONV1=1 timeout -k 10 120 python ../../tools/conv_scaling.py 1024 || exit 1
  case "$v" in *SNNQP_F6_TRACE*) timeout -k 10 120 python ../../tools/f6_trace.py | tail -2;; esac
done
