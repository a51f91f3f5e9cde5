#!/bin/bash
# Measurement: kernel time of the headline fused kernel under the DPK_ABLATE variants (GPU box).
# usage: tools/ablate.sh [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "" _ab3 _ab4 _ab5; do
  lib=deeprob-kit_amd/lib/libdeeprob_hip$v.so
  [ -f "$lib" ] || continue
  echo -n "variant '${v:-full}': "
  DEEPROB_HIP_LIB=$PWD/$lib python bench.py --steps 100 --warmup 10 --cpu-samples 0 "$@" 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.4f  step_ms %.4f  frac %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
done | tee gpurun_out/ablate.log
