"""Round-6 profile artefacts (run on the GPU box): kernel-trace statistics and PMC traffic of the benchmarked paths.
Writes gpurun_out/profiles_r06/*; the files to be judged are copied into profiles/ afterwards (see the bottom).

  python tools/collect_r06.py            everything
  python tools/collect_r06.py trace      kernel traces only
  python tools/collect_r06.py pmc        PMC passes only (FETCH_SIZE and WRITE_SIZE in separate runs, no tracing flags)
"""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'profiles_r06')
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR='/tmp')
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
only = sys.argv[2:]      # optional: workload names

WORKLOADS = {   # name -> command (relative to the repo root)
    # (--streams 1: the launches of the two-stream loop overlap and stretch each other's durations; the per-kernel figures
    # are those of launches that have the chip to themselves, as roofline.kernel_ms in the bench line)
    'headline': ['python', 'bench.py', '--cpu-samples', '0', '--no-secondary', '--streams', '1'],
    'headline2s': ['python', 'bench.py', '--cpu-samples', '0', '--no-secondary'],
    'topdown': ['python', 'tools/bench_topdown.py', '65536'],
    'shard32768': ['python', 'bench.py', '--cpu-samples', '0', '--no-secondary', '--streams', '1', '--batch', '32768'],
    'shard16384': ['python', 'bench.py', '--cpu-samples', '0', '--no-secondary', '--streams', '1', '--batch', '16384'],
    'shard8192': ['python', 'bench.py', '--cpu-samples', '0', '--no-secondary', '--streams', '1', '--batch', '8192'],
    'config2': ['python', 'tools/bench_config2_modes.py', '4096'],
    'marginal': ['python', 'tools/bench_small.py', '65536'],
    'wide': ['python', 'tools/bench_wide_small.py', '4096', '65536'],
    'folded': ['python', 'tools/bench_folded_layers.py', '4096', '65536'],
    'train': ['python', 'tools/bench_train.py', 'ratspn', '512'],
    'train_nvp': ['python', 'tools/bench_train.py', 'realnvp', '512'],
    'train_dgc': ['python', 'tools/bench_train.py', 'dgcspn', '512'],
    'train_nvp2d': ['python', 'tools/bench_flows2d_train.py'],
    'config4b': ['python', 'tools/diag_dgc_secondary.py'],
    'config4': ['python', 'tools/bench_dgc.py'],
    'config5': ['python', 'tools/bench_flows.py'],
}


def rocprof(args, name, cmd):
    d = os.path.join(OUT, 'raw_' + name)
    shutil.rmtree(d, ignore_errors=True)
    full = ['rocprofv3'] + args + ['-d', d, '-o', name, '--output-format', 'csv', '--'] + \
           [os.path.join(ROOT, c) if c.endswith('.py') else c for c in cmd]
    r = subprocess.run(full, cwd='/tmp', env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    return d, r


def kernel_trace(name, cmd):
    d, r = rocprof(['--kernel-trace', '--stats'], name, cmd)
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('dpk::', '')
            grid = int(row.get('Grid_Size_X', row.get('Grid_Size', 0))) // max(1, int(row.get('Workgroup_Size_X', row.get('Workgroup_Size', 1))))
            acc[(k[:64], grid, row.get('Grid_Size_Y', '1'))].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    tot = sum(sum(v) for v in acc.values()) or 1.0
    lines = ['rocprofv3 --kernel-trace --stats -- ' + ' '.join(cmd),
             '%-64s %7s %3s %7s %10s %10s %10s %7s' % ('kernel', 'blocks', 'gy', 'calls', 'avg_us', 'min_us', 'med_us', 'pct')]
    for (k, g, gy), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        lines.append('%-64s %7d %3s %7d %10.2f %10.2f %10.2f %6.1f%%' % (k, g, gy, len(v), sum(v) / len(v), v2[0], v2[len(v2) // 2],
                                                                          100 * sum(v) / tot))
    open(os.path.join(OUT, 'r06_%s_kernel_stats.txt' % name), 'w').write('\n'.join(lines[:40]) + '\n')
    shutil.rmtree(d, ignore_errors=True)
    print('\n'.join(lines[:12]))


def pmc(name, cmd, counter):
    d, r = rocprof(['--pmc', counter], name + '_' + counter, cmd)
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('dpk::', '')[:64]
            grid = int(row.get('Grid_Size', 0)) // max(1, int(row.get('Workgroup_Size', 1)))
            a = acc[(k, grid)]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
    shutil.rmtree(d, ignore_errors=True)
    return {k: (v[0] / v[1], v[1], v[0]) for k, v in acc.items()}


if what in ('all', 'trace'):
    for name, cmd in WORKLOADS.items():
        if only and name not in only:
            continue
        kernel_trace(name, cmd)
    if only:
        sys.exit(0)
    r = subprocess.run(['python', os.path.join(ROOT, 'bench.py')], cwd=ROOT, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    open(os.path.join(OUT, 'r06_bench_line.json'), 'wb').write(r.stdout.strip().splitlines()[-1] + b'\n' if r.stdout.strip() else b'')

if what in ('all', 'pmc'):
    report, traffic = [], {}
    prev_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if only and os.path.exists(prev_path):      # (a partial pass keeps the other workloads' entries)
        traffic = json.load(open(prev_path)).get('kernels', {})
    # (kernel-name filter, grid filter or None, key in pmc_traffic.json, launches per step or None = per launch)
    want = {
        'headline': [('ratspn_gemm_slice_kernel', None, 'headline')],
        'config2': [('ratspn_gemm_small_kernel', None, 'config2_2_2'), ('ratspn_gemm_wide_kernel<8, false>', None, 'config2_8_8')],
        'wide': [('ratspn_gemm_wide_ring_kernel', None, 'wide_65536')],
        'marginal': [('ratspn_gemm_marginal_kernel', None, 'marginal_65536')],
        'config5': [('coupling_x1_kernel<true, 4, false', None, 'config5')],
        'config4': [('', None, 'config4')],
        'config4b': [('', None, 'config4b')],
    }
    for name, cmd in WORKLOADS.items():
        if name in ('train', 'folded', 'train_nvp', 'train_dgc', 'train_nvp2d', 'shard32768', 'shard16384', 'shard8192', 'headline2s', 'topdown') or (only and name not in only):
            continue
        fetch, write = pmc(name, cmd, 'FETCH_SIZE'), pmc(name, cmd, 'WRITE_SIZE')
        report.append('== %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- %s' % (name, ' '.join(cmd)))
        report.append('%-64s %7s %7s %14s %14s' % ('kernel', 'blocks', 'calls', 'FETCH_SIZE KB', 'WRITE_SIZE KB'))
        for k in sorted(set(fetch) | set(write)):
            f, w = fetch.get(k, (0, 0, 0)), write.get(k, (0, 0, 0))
            report.append('%-64s %7d %7d %14.1f %14.1f' % (k[0], k[1], max(f[1], w[1]), f[0], w[0]))
        for flt, grid, key in want.get(name, []):
            ks = [k for k in fetch if flt in k[0] and (grid is None or k[1] == grid)]
            if not ks:
                continue
            if key in ('config4', 'config4b'):   # whole step: every kernel of the forward, 13 forwards in tools/bench_dgc.py
                fb = sum(fetch[k][2] for k in fetch) / 13.0
                wb = sum(write[k][2] for k in write) / 13.0
                per = 'step'
            else:
                k = max(ks, key=lambda q: fetch[q][1])
                fb, wb, per = fetch[k][0], write.get(k, (0, 0, 0))[0], 'launch of ' + k[0]
            traffic[key] = {'fetch_size_kb_raw': fb, 'write_size_kb_raw': wb, 'fetch_correction': 2.0,
                            'bytes_per_launch': (2.0 * fb + wb) * 1024.0, 'per': per,
                            'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `%s`, '
                                      'profiles/r06_pmc_summary.txt; FETCH_SIZE doubled (gfx950: 128-B requests '
                                      'tallied at 64 B), WRITE_SIZE as reported' % ' '.join(cmd)}
    open(os.path.join(OUT, 'r06_pmc_summary.txt'), 'w').write('\n'.join(report) + '\n')
    json.dump({'kernels': traffic}, open(os.path.join(OUT, 'pmc_traffic.json'), 'w'), indent=1)
    print('\n'.join(report[:60]))
    print(json.dumps({k: round(v['bytes_per_launch'] / 1e6, 2) for k, v in traffic.items()}))

if what in ('all', 'mfma'):
    # matrix-core occupancy of the coupling kernel (north star: "MFMA utilisation of the coupling GEMM against chip peak"):
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE the kernel's span
    lines = ['rocprofv3 --pmc <counter> (one pass per counter) -- python tools/bench_flows.py; kernel coupling_x1_kernel<true, 4, false, ...>']
    vals = {}
    for counter in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'GRBM_GUI_ACTIVE'):
        try:
            r = pmc('config5', WORKLOADS['config5'], counter)
        except Exception as ex:
            lines.append('%-32s failed: %s' % (counter, ex))
            continue
        ks = [k for k in r if 'coupling_x1_kernel<true, 4, false' in k[0]]
        if ks:
            k = max(ks, key=lambda q: r[q][1])
            vals[counter] = r[k][0]
            lines.append('%-32s per launch %.6g  (n=%d, %s)' % (counter, r[k][0], r[k][1], k[0]))
        else:
            lines.append('%-32s no matching kernel row' % counter)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals and 'GRBM_GUI_ACTIVE' in vals and vals['GRBM_GUI_ACTIVE'] > 0:
        # Both counters arrive summed over the 8 XCDs: GRBM_GUI_ACTIVE per launch is 8 x the kernel's cycles (2.29e6 for a
        # 128 us launch at ~2.2 GHz), the busy cycles are summed over every SIMD.  Per XCD: 128 SIMDs (32 CUs x 4).
        lines.append('matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (128 SIMDs per XCD x GRBM_GUI_ACTIVE summed over the XCDs) = %.4f'
                     % (vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * vals['GRBM_GUI_ACTIVE'])))
        lines.append('(cross-check: the bench line\'s mfma_frac for config 5 -- executed f16 MFMA flops / 2.5 PF over the kernel time -- is 0.19)')
    open(os.path.join(OUT, 'r06_config5_mfma_pmc.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
