"""Host logic of the flat SPN loader (no GPU): layout against the oracle's reading of the same JSON, validity
errors of the reference (check_spn / DAG / labels)."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import flat_spn_oracle as forc
from tests.flat_spn_cases import random_circuit

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _flat(d):
    from deeprob.spn.structure.io import digraph_to_spn
    return digraph_to_spn(d)


@pytest.mark.parametrize('name', ['binary16', 'mixed4'])
def test_layout_matches_the_export(name):
    from deeprob.spn.structure.io import load_spn_json
    path = os.path.join(GOLD, 'spn_%s.json' % name)
    spn = load_spn_json(path)
    spn.check()
    nodes, children = forc.load(path)
    assert spn.n_nodes == len(nodes) and spn.root == 0
    seen = set()
    for i in spn.order:                                   # children before parents, every node once
        assert all(c in seen for c in children[int(i)])
        seen.add(int(i))
    assert seen == set(nodes)
    for i, n in nodes.items():
        if n['class'] in ('Sum', 'Product'):
            got = spn.child_index[spn.arg0[i]:spn.arg0[i] + spn.arg1[i]]
            assert list(got) == children[i]
            if n['class'] == 'Sum':
                w = spn.child_weight[spn.arg0[i]:spn.arg0[i] + spn.arg1[i]]
                assert np.array_equal(w, np.array(n['weights'], np.float32))
        else:
            assert spn.arg0[i] == n['scope'][0]
    with open(path) as f:                                 # file objects load too (reference io.py:73-80)
        assert load_spn_json(f).n_nodes == spn.n_nodes


def test_random_circuits_are_valid():
    for seed in range(3):
        d, _ = random_circuit(9, seed)
        spn = _flat(d)
        spn.check()
        assert spn.n_features == 9 and spn.n_nodes == len(d['nodes'])


@pytest.mark.parametrize('n_features,seed', [(5, 0), (9, 2), (12, 3)])
def test_value_table_rows_are_recycled_safely(n_features, seed):
    """Replay of the evaluator's on-chip table: when a node is evaluated every child's row still holds that child,
    and the root's row survives to the end; far fewer rows than nodes are needed."""
    d, _ = random_circuit(n_features, seed)
    spn = _flat(d)
    owner = {}
    for node in spn.order:
        node = int(node)
        c0, nc = (spn.arg0[node], spn.arg1[node]) if spn.kind[node] <= 1 else (0, 0)
        for j in range(nc):
            assert owner[int(spn.child_slot[c0 + j])] == int(spn.child_index[c0 + j])
        owner[int(spn.node_slot[node])] = node
    assert owner[int(spn.node_slot[spn.root])] == spn.root
    assert spn.n_slots == 1 + max(spn.node_slot) and spn.n_slots <= min(spn.n_nodes, 64)


def test_invalid_structures_raise_like_the_reference():
    d, _ = random_circuit(6, 1)
    bad = copy.deepcopy(d)                                # cycle: the root becomes a child of one of its products
    prod = next(n['id'] for n in bad['nodes'] if n['class'] == 'Product')
    bad['links'].append({'source': 0, 'target': prod, 'idx': 2})
    with pytest.raises(ValueError, match='DAG'):
        _flat(bad)
    bad = copy.deepcopy(d)                                # unknown class
    bad['nodes'][-1]['class'] = 'Isotonic'
    with pytest.raises(ValueError, match='Unknown node'):
        _flat(bad)
    bad = copy.deepcopy(d)                                # labels with a hole
    bad['nodes'][-1]['id'] = 10 ** 6
    with pytest.raises(ValueError):
        _flat(bad)
    bad = copy.deepcopy(d)                                # sum weights that do not sum to one
    next(n for n in bad['nodes'] if n['class'] == 'Sum')['weights'][0] += 0.5
    with pytest.raises(ValueError, match="sum up to 1"):
        _flat(bad)
    bad = copy.deepcopy(d)                                # a product whose children overlap: not decomposable
    p = next(n for n in bad['nodes'] if n['class'] == 'Product')
    kids = [e for e in bad['links'] if e['target'] == p['id']]
    kids[1]['source'] = kids[0]['source']
    with pytest.raises(ValueError, match='decomposable|reachable'):
        _flat(bad).check()
    bad = copy.deepcopy(d)                                # a sum over children of different scopes: not smooth
    s = next(n for n in bad['nodes'] if n['class'] == 'Sum' and len(n['scope']) > 1)
    first = next(e for e in bad['links'] if e['target'] == s['id'])
    leaf = next(n['id'] for n in bad['nodes'] if n['class'] not in ('Sum', 'Product'))
    first['source'] = leaf
    with pytest.raises(ValueError, match='smooth|reachable'):
        _flat(bad).check()


def test_log_likelihood_needs_a_device():
    """No CPU fallback: without a HIP device the evaluation raises instead of computing on the host."""
    import torch
    from deeprob.spn.structure.io import load_spn_json
    from deeprob.spn.algorithms.inference import log_likelihood
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    spn = load_spn_json(os.path.join(GOLD, 'spn_mixed4.json'))
    with pytest.raises(Exception):
        log_likelihood(spn, torch.zeros(3, 4))
    with pytest.raises(TypeError):
        log_likelihood(json.load(open(os.path.join(GOLD, 'spn_mixed4.json'))), np.zeros((3, 4), np.float32))
