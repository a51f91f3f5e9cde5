#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
mkdir -p $OUT
run() {
  name=$1; shift
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python $OLDPWD/bench.py --steps 20 --warmup 3 --cpu-samples 0 --no-kernel-events --no-secondary > /dev/null 2>$OUT/$name.err)
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections, os
f = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if os.environ.get('KERN', 'ratspn_gemm_kernel') not in row['Kernel_Name']:
        continue
    acc[row['Counter_Name']][0] += float(row['Counter_Value']); acc[row['Counter_Name']][1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-28s per-launch %.5g  (n=%d)' % (k, v / n, n))
PY
}
run smem1 SmemLatency
run smem2 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run smem3 SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_DCACHE_INPUT_VALID_READYB
run lat2 VmemLatency
run sq3 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
