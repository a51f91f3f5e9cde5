#!/bin/bash
# GPU box: A/B/.. of several builds of the library on tools/bench_slice.py; usage: run_slice_abn.sh "libA.so libB.so ..." [batches]
cd "$(dirname "$0")/.."
LIBS=$1; shift
for rep in 1 2; do
for L in $LIBS; do
  echo "== $L"
  DEEPROB_HIP_LIB=$PWD/deeprob-kit_amd/lib/$L timeout 300 python tools/bench_slice.py "$@" 2>/dev/null | grep "^{" | grep "slice"
done
done
