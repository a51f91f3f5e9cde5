#!/bin/bash
# GPU box: BASELINE config 2 (B = 4096) on several builds of the library; usage: run_config2_ab.sh "libA.so libB.so" [args of bench_config2_modes.py]
cd "$(dirname "$0")/.."
LIBS=$1; shift
for rep in 1 2; do
for L in $LIBS; do
  echo "== $L"
  DEEPROB_HIP_LIB=$PWD/deeprob-kit_amd/lib/$L timeout 300 python tools/bench_config2_modes.py "$@" 2>/dev/null | tail -4
done
done
