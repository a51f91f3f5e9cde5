#!/bin/bash
# GPU box: parity of the slice mapping, then its kernel times by grid size next to the other two mappings.
# writes gpurun_out/slice/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/slice
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_ratspn_gpu.py -x -q -k "slice or full_size" > $OUT/tests.txt 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.txt
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $OLDPWD/tools/bench_slice.py "$@" > $OUT/stdout.txt 2>$OUT/stderr.txt)
echo "bench rc=$?"; tail -3 $OUT/stderr.txt
grep "^{" $OUT/stdout.txt
python tools/trace_summary.py $OUT ratspn_gemm | tee $OUT/by_grid.txt
find $OUT -name "*kernel_trace.csv" -delete
