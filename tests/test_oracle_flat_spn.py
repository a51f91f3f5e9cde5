"""The flat-SPN oracle against the reference's own evaluations of its JSON exports (BASELINE config 1)."""
import os

import numpy as np
import pytest

from oracle import flat_spn_oracle as forc

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = [('binary16', 'binary16'), ('binary16', 'binary16_nan'), ('mixed4', 'mixed4')]


@pytest.mark.parametrize('circuit,vectors', CASES)
def test_oracle_matches_reference(circuit, vectors):
    g = np.load(os.path.join(GOLD, 'spn_%s.npz' % vectors))
    ll, table = forc.log_likelihood(os.path.join(GOLD, 'spn_%s.json' % circuit), g['x'], return_results=True)
    assert ll.dtype == np.float32 and ll.shape == g['ll'].shape
    assert np.array_equal(ll, g['ll'])                       # same scipy calls, same order: bit-exact
    assert np.array_equal(table, g['per_node'])


def test_config1_known_answers():
    """72 nodes, mean LL -9.4747; a fully marginalised row has LL 0."""
    g = np.load(os.path.join(GOLD, 'spn_binary16.npz'))
    assert g['per_node'].shape == (72, 1000)
    assert abs(float(np.mean(g['ll'])) + 9.4747) < 1e-4
    gn = np.load(os.path.join(GOLD, 'spn_binary16_nan.npz'))
    assert gn['ll'][0] == 0.0 and gn['ll'][1] == np.float32(-1e31)


def test_cycle_is_rejected():
    spn = {'nodes': [{'id': 0, 'class': 'Product', 'scope': [0]}, {'id': 1, 'class': 'Product', 'scope': [0]}],
           'links': [{'source': 1, 'target': 0, 'idx': 0}, {'source': 0, 'target': 1, 'idx': 0}]}
    with pytest.raises(ValueError):
        forc.log_likelihood(spn, np.zeros((2, 1), dtype=np.float32))
