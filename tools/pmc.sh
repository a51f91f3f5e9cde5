#!/bin/bash
# PMC passes for the headline kernel (GPU box).  Each counter group is its own rocprofv3 run
# (--pmc only, no tracing flags), output under gpurun_out/pmc/<group>.
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
    print('%-28s per-launch %.4g  (n=%d)' % (k, v / n, n))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
