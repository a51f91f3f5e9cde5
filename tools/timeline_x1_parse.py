"""Summarise gpurun_out/x1_timeline.txt (tools/timeline_x1.py): per-step spans of MFMA wave 0 / 1 and holder wave 4."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/x1_timeline.txt').read()
lines = [l for l in txt.split('\n') if l.startswith('x1 timeline')]
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for l in lines[3 * layer:3 * layer + 3]:
    w = l.split(':')[0]
    rows = [[int(v) for v in r.split()] for r in re.findall(r'\[([^\]]*)\]', l)]
    print(w)
    hold = 'wave 4' in w
    prev = None
    for i, r in enumerate(rows[:56]):
        end = r[2] if hold else r[4]
        if hold:
            print(i, 'wait+barrier', r[1] - r[0], 'work', r[2] - r[1], 'step', (end - prev) if prev else None)
        else:
            print(i, 'wait', r[1] - r[0], 'barrier', r[2] - r[1], 'issue', (r[3] - r[2]) if r[3] > 0 else 0,
                  'compute', r[4] - (r[3] if r[3] > 0 else r[2]), 'step', (end - prev) if prev else None)
        prev = end
