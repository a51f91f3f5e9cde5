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
