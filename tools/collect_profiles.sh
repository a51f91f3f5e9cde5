#!/bin/bash
# Round profile artefacts (GPU box): rocprofv3 --kernel-trace --stats of the default bench, PMC passes
# for the dominant kernel, and the bench line itself.  Output: gpurun_out/profiles_<tag>/ -- copy what
# should be judged into profiles/.
tag=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles_$tag
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python $OLDPWD/bench.py --cpu-samples 0 --no-secondary > $OUT/trace_bench_line.json 2>/dev/null)
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" > $OUT/kernel_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('rocprofv3 --kernel-trace --stats -- python bench.py --cpu-samples 0   (default steps/warmup)')
print('%-70s %8s %12s %8s' % ('kernel', 'calls', 'avg_us', 'pct'))
for r in rows[:8]:
    print('%-70s %8s %12.2f %8s' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
tools/pmc.sh > $OUT/pmc_summary.txt 2>&1
tools/pmc2.sh >> $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/trace/*kernel_trace.csv   # large; the stats csv is what is summarised
cat $OUT/kernel_stats_summary.txt; tail -1 $OUT/bench_line.json | cut -c 1-1500
