for mode in "A=1" "DPK_RING_VI=1"; do
env $mode python bench.py --no-secondary --cpu-samples 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config']['ms_per_step_default_mode'], d['roofline']['kernel_ms'])"
done
