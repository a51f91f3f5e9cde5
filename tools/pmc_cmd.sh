#!/bin/bash
# PMC counters of one kernel for an arbitrary command (GPU box).  Each invocation = one rocprofv3 --pmc pass
# (never combined with tracing flags; FETCH_SIZE and WRITE_SIZE need separate passes; a rejected counter set makes
# rocprofv3 abort and then hang, hence the timeout).  usage: tools/pmc_cmd.sh <tag> <kernel-name-substring> "<counters>" cmd...
tag=$1; kern=$2; counters=$3; shift 3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$tag
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc $counters -d $OUT -o p --output-format csv -- "$@" > $OUT/stdout.txt 2>$OUT/stderr.txt)
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$f" "$kern" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in row['Kernel_Name']:
        continue
    acc[row['Counter_Name']][0] += float(row['Counter_Value']); acc[row['Counter_Name']][1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-28s per-launch %.4g  (n=%d)' % (k, v / n, n))
PY
