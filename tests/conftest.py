import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'deeprob-kit_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a MI355X (HIP device); run with -m gpu on the GPU box')
    # the C-ABI library is a build artefact (git-ignored): build it once if this checkout has none yet
    lib = os.path.join(PKG, 'lib', 'libdeeprob_hip.so')
    if not os.path.isfile(lib):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(PKG, 'csrc'), '-j', str(min(8, os.cpu_count() or 1))], check=True)


def pytest_collection_modifyitems(config, items):
    # GPU tests never run silently on a box without a device
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed():
    np.random.seed(42)
    torch.manual_seed(42)
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture
def golden():
    return load_golden
