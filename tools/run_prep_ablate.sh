#!/bin/bash
# GPU box: duration of the stand-alone table build of the (8,8) training step with parts of it switched off
# (DPK_PREP_ABLATE: 1 stop after the fingerprint, 2 no fragments, 4 no constants)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for ab in 0 1 2 4 6; do
  rm -rf /tmp/pa; mkdir -p /tmp/pa
  (cd /tmp && DPK_PREP_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o t --output-format csv -- python $OLDPWD/tools/bench_train.py ratspn 512 > /dev/null 2>&1)
  f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
  echo "ablate=$ab: $(grep ratspn_gemm_prep_kernel $f | head -1 | cut -d, -f1-5)"
done
