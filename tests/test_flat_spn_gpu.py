"""Parity of the HIP flat-SPN evaluator (through the C ABI) with the reference's golden vectors (BASELINE config 1)
and with the oracle on random circuits."""
import os

import numpy as np
import pytest
import torch

from oracle import flat_spn_oracle as forc
from tests.flat_spn_cases import random_circuit, random_inputs
from tests.util import rel_err

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
LL_TOL = 1e-5     # north-star: 1e-5 relative on fp32 log-likelihoods
FLOOR = np.float32(-1e31)


def _load(name):
    from deeprob.spn.structure.io import load_spn_json
    return load_spn_json(os.path.join(GOLD, 'spn_%s.json' % name))


@pytest.mark.parametrize('circuit,vectors', [('binary16', 'binary16'), ('binary16', 'binary16_nan'),
                                             ('mixed4', 'mixed4')])
def test_golden(circuit, vectors):
    """Root and per-node values of the reference's own evaluation; the -1e31 floor is hit exactly where it does."""
    from deeprob.spn.algorithms.inference import log_likelihood
    g = np.load(os.path.join(GOLD, 'spn_%s.npz' % vectors))
    spn = _load(circuit)
    ll = log_likelihood(spn, g['x'])
    assert isinstance(ll, np.ndarray) and ll.dtype == np.float32 and ll.shape == g['ll'].shape
    assert np.array_equal(ll == FLOOR, g['ll'] == FLOOR)
    assert rel_err(ll, g['ll']) <= LL_TOL
    ll2, table = log_likelihood(spn, g['x'], return_results=True)
    assert np.array_equal(ll2, ll)                          # LDS table and global table agree bit for bit
    assert table.shape == g['per_node'].shape
    assert np.array_equal(table == FLOOR, g['per_node'] == FLOOR)
    assert rel_err(table, g['per_node']) <= LL_TOL


def test_config1_normalisation_over_all_assignments():
    """The learned 16-variable circuit is a distribution: its likelihoods over all 2^16 assignments sum to one, and
    marginalising any prefix of the variables keeps that true (size-independent property at the full domain)."""
    from deeprob.spn.algorithms.inference import log_likelihood
    spn = _load('binary16')
    bits = ((np.arange(1 << 16)[:, None] >> np.arange(16)[None, :]) & 1).astype(np.float32)
    ll = log_likelihood(spn, torch.from_numpy(bits).cuda())
    assert isinstance(ll, torch.Tensor) and ll.is_cuda
    assert abs(float(torch.exp(ll.double()).sum()) - 1.0) < 1e-5
    for k in (4, 11):
        part = bits[: 1 << (16 - k)].copy()
        part = np.concatenate([np.full((len(part), k), np.nan, np.float32), part[:, : 16 - k]], axis=1)
        llk = log_likelihood(spn, part)
        assert abs(float(np.exp(llk.astype(np.float64)).sum()) - 1.0) < 1e-5


@pytest.mark.parametrize('n_features,seed,B', [(5, 0, 1), (7, 1, 63), (9, 2, 65), (12, 3, 1000), (17, 4, 4097)])
def test_random_circuits_against_oracle(n_features, seed, B):
    """Mixed leaf families, out-of-support inputs, NaNs, ragged batches; circuits above 256 nodes take the
    workspace route."""
    from deeprob.spn.structure.io import digraph_to_spn
    from deeprob.spn.algorithms.inference import log_likelihood
    d, family = random_circuit(n_features, seed)
    x = random_inputs(family, B, seed + 100)
    x[0, :] = np.nan
    want, want_table = forc.log_likelihood(d, x, return_results=True)
    spn = digraph_to_spn(d)
    got = log_likelihood(spn, x)
    assert abs(got[0]) < 1e-5 and abs(want[0]) < 1e-5      # fully marginalised row (weights sum to 1 within fp32)
    assert np.array_equal(got == FLOOR, want == FLOOR)
    assert rel_err(got, want) <= LL_TOL
    _, table = log_likelihood(spn, x, return_results=True)
    assert rel_err(table, want_table) <= LL_TOL
    if n_features >= 12:
        assert spn.n_nodes > 256 and spn.n_slots <= 64       # thousands of nodes, a few dozen live rows
    spn.n_slots = 0                                          # no on-chip table: the workspace route
    again = log_likelihood(spn, x)
    assert np.array_equal(again, got)


def test_empty_batch_and_errors():
    from deeprob.spn.algorithms.inference import log_likelihood
    spn = _load('mixed4')
    out = log_likelihood(spn, np.zeros((0, 4), np.float32))
    assert out.shape == (0,)
    with pytest.raises(ValueError):
        log_likelihood(spn, np.zeros((3, 2), np.float32))                 # does not cover the scope
    from deeprob.hip import HipError
    with pytest.raises(HipError):
        log_likelihood(spn, torch.zeros(3, 4))                            # host tensor: no CPU fallback
    wider = np.zeros((5, 9), np.float32)                                   # extra columns are ignored
    wider[:, :4] = np.load(os.path.join(GOLD, 'spn_mixed4.npz'))['x'][5:10]
    a = log_likelihood(spn, wider)
    b = log_likelihood(spn, wider[:, :4].copy())
    assert np.array_equal(a, b)
