#!/bin/bash
# per-dispatch durations of one kernel grouped by grid shape (GPU box): tools/trace_by_grid.sh <kernel substring> <cmd...>
pat=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_grid
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- "$@" > $OUT/stdout.txt 2>$OUT/stderr.txt) < /dev/null
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no trace"; exit 1; }
python - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
by = collections.defaultdict(list)
for r in rows:
    key = (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
    by[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('grid %-20s n=%5d  median %8.2f us  total %10.1f us' % ('x'.join(k), len(v), v[len(v) // 2], sum(v)))
PY
rm -f $f
