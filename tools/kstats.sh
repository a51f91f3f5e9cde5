#!/bin/bash
# usage: tools/kstats.sh <tag> <command...>   -> per-kernel average durations (rocprofv3 --kernel-trace --stats)
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ks_$tag
rm -rf $OUT; mkdir -p $OUT
R=$PWD
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- "$@" > $OUT/stdout.log 2>$OUT/stderr.log)
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('%-90s %8s %12s %8s' % ('kernel', 'calls', 'avg_us', 'pct'))
for r in rows[:26]:
    print('%-90s %8s %12.2f %8s' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
rm -rf $OUT/trace
