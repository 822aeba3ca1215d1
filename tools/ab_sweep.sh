#!/bin/bash
# tools/ab_sweep.sh OUT "name1 name2" "args1" "args2" ...: every build under every bench argument set
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; names=$2; shift 2
for args in "$@"; do
  echo "== $args"
  BENCH_ARGS="$args" bash tools/ab_bench.sh $out $names
done
