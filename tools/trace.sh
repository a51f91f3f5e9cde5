#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (GPU box); writes gpurun_out/trace/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python $OLDPWD/bench.py --steps 100 --warmup 10 --cpu-samples 0 "$@" > $OUT/bench.json 2>$OUT/bench.err)
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $f"; head -12 "$f"
cat $OUT/bench.json | tail -1 | cut -c1-400
