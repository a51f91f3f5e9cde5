#!/bin/bash
# GPU box: A/B of two builds of the library on tools/bench_slice.py (slice mapping only matters); usage: run_slice_ab.sh libA.so libB.so [batches]
cd "$(dirname "$0")/.."
A=$1; B=$2; shift; shift
for rep in 1 2; do
for L in $A $B; do
  echo "== $L"
  DEEPROB_HIP_LIB=$PWD/deeprob-kit_amd/lib/$L timeout 300 python tools/bench_slice.py "$@" 2>/dev/null | grep "^{" | grep "slice\|ring"
done
done
