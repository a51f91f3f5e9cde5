#!/bin/bash
# rocprofv3 kernel-trace summary of an arbitrary command (GPU box): tools/trace_cmd.sh <tag> <cmd...>
# writes gpurun_out/trace_<tag>/ and prints the top kernels
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$tag
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- "$@" > $OUT/stdout.txt 2>$OUT/stderr.txt)
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" "$*" <<'PY' | tee $OUT/kernel_stats_summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('rocprofv3 --kernel-trace --stats -- ' + sys.argv[2])
print('%-72s %8s %12s %8s' % ('kernel', 'calls', 'avg_us', 'pct'))
for r in rows[:16]:
    print('%-72s %8s %12.2f %8s' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
rm -f $OUT/*kernel_trace.csv
tail -1 $OUT/stdout.txt | cut -c1-600
