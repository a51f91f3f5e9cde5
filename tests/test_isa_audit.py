"""Build-time audit of the x-once coupling kernel's ISA (no GPU needed: hipcc cross-compiles).

coupling_x1_kernel orders its LDS ring with hand-counted ``s_waitcnt vmcnt(n)`` around inline-asm LDS-DMA instructions
the compiler does not count.  That is only sound while the compiler's OWN vector-memory loads in the issuing waves are
waited for conservatively: a register reload from scratch (the fused-base variants spill ~15 registers) followed by a
partial ``vmcnt(k)`` with DMA instructions issued in between would read the register before the reload landed.  The
test compiles the file to assembly and checks, for every instantiation, that each scratch reload is followed by a
``vmcnt(0)`` with no LDS-DMA instruction between the two."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_x_once_coupling_reloads_are_waited_for_in_full(tmp_path):
    src = os.path.join(ROOT, 'deeprob-kit_amd', 'csrc', 'coupling_x3.hip')
    out = str(tmp_path / 'x3.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
                    '-S', '--cuda-device-only', src, '-o', out], check=True, cwd=os.path.dirname(src),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    text = open(out).read()
    names = re.findall(r'^(_ZN3dpk18coupling_x1_kernel\w+):', text, re.M)
    assert len(names) == 32, names          # AFFINE x NU(4) x BASE x PM
    reloads = 0
    for name in names:
        body = re.search(r'^' + name + r':.*?^\s*s_endpgm', text, re.S | re.M).group(0)
        ins = [l.strip() for l in body.split('\n')]
        ins = [l for l in ins if l and not l.startswith(';') and not l.startswith('.')]
        assert sum('global_load_lds_dwordx4' in l for l in ins) > 0, name
        for i, l in enumerate(ins):
            if not l.startswith('scratch_load'):
                continue
            reloads += 1
            j, dmas = i + 1, 0
            while j < len(ins) and not (ins[j].startswith('s_waitcnt') and 'vmcnt' in ins[j]):
                dmas += 'global_load_lds' in ins[j]
                j += 1
            assert j < len(ins) and 'vmcnt(0)' in ins[j] and dmas == 0, (name, i, ins[j] if j < len(ins) else None, dmas)
    print('x-once coupling kernels: {} instantiations, {} scratch reloads, all behind vmcnt(0)'.format(len(names), reloads))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
@pytest.mark.parametrize('source,pattern,expect', [
    ('ratspn_gemm_small.hip', r'ratspn_gemm_small_kernelILi\d+ELi\d+ELi\d+ELi\d+E', 24),
    ('ratspn_gemm_wide.hip', r'ratspn_gemm_wide_kernelILi\d+ELb[01]ELb[01]ELi[48]E', 24),   # (x2: one / two work-groups per block)
    ('ratspn_gemm_slice.hip', r'ratspn_gemm_slice_kernelILi\d+ELi\d+ELi\d+E', 1),
    # (round 6: the DGC-SPN streaming levels, every layout variant -- a rare path's hoisted address math spilled twelve registers)
    ('dgcspn_stream.hip', r'spatial_stream_kernelILi[01]ELi[012]ELb[01]E', 8),
])
def test_small_batch_kernels_use_no_scratch(tmp_path, source, pattern, expect):
    """The 32-sample RAT-SPN kernels run for 9 .. 20 us; a kernel that uses scratch pays for its set-up on every launch
    whether or not the spilling path is taken (round 4: 536 B per lane behind a noinline call cost 3.5 us, a by-value copy
    of the table arguments 264 B).  Every instantiation must report ScratchSize 0."""
    src = os.path.join(ROOT, 'deeprob-kit_amd', 'csrc', source)
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
                        '-c', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', src, '-o', str(tmp_path / 'o.o')],
                       cwd=os.path.dirname(src), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    blocks = re.split(r'remark: [^\n]*Function Name: ', r.stdout)[1:]
    seen = 0
    for blk in blocks:
        name = blk.split()[0]
        if not re.search(pattern, name):
            continue
        seen += 1
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', blk)
        assert m and int(m.group(1)) == 0, (name, m.group(0) if m else None)
    assert seen == expect, seen
