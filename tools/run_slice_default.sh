#!/bin/bash
# GPU box: the slice mapping's default-mode (parameter tables checked) step against the frozen-model step, with the
# in-launch check and with the stand-alone check launch (DPK_VERIFY_INLINE=0); usage: run_slice_default.sh [batches]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  echo "== frozen";  timeout 300 python tools/bench_slice.py "$@" --only-slice 2>/dev/null | grep "^{"
  echo "== default, in-launch check";  timeout 300 python tools/bench_slice.py "$@" --only-slice --default-mode 2>/dev/null | grep "^{"
  echo "== default, stand-alone check";  DPK_VERIFY_INLINE=0 timeout 300 python tools/bench_slice.py "$@" --only-slice --default-mode 2>/dev/null | grep "^{"
done
