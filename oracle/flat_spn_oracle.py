"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's vanilla SPN evaluation.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module;
the shipped package (``deeprob-kit_amd/``) never does.

Restates, for a circuit given as the reference's JSON export (``deeprob/spn/structure/io.py:133-220``:
networkx node-link data, node attributes ``class / scope / weights / params``, edges child -> parent with the
child's position ``idx``), the bottom-up pass of ``deeprob/spn/algorithms/evaluation.py:37-96`` with the node
functions of ``deeprob/spn/algorithms/inference.py:94-103`` (every node's value clamped at -1e31 and stored as
float32) over

* ``Sum.log_likelihood``      ``deeprob/spn/structure/node.py:116-117``  scipy ``logsumexp(x, b=weights)``
* ``Product.log_likelihood``  ``node.py:152-153``                        ``np.sum(x, axis=1)``
* ``Bernoulli / Categorical / Uniform / Gaussian .log_likelihood``  ``deeprob/spn/structure/leaf.py:182-186,
  301-305, 475-479, 553-557``: zero where the input is NaN, scipy.stats log-pmf / log-pdf elsewhere.

The same scipy calls as the reference are used, node by node, on plain arrays (no Node classes).

Pinned: ``tests/test_oracle_flat_spn.py`` checks root and per-node values against ``tests/golden/spn_*.npz``, which
``tools/gen_golden_spn.py`` produced by running the reference's ``log_likelihood`` on its own JSON exports.
"""
import json
import warnings
from typing import Dict, List, Tuple

import numpy as np
import scipy.stats as ss
from scipy.special import logsumexp

FLOOR = -1e31   # inference.py:103


def load(path_or_dict) -> Tuple[Dict[int, dict], Dict[int, List[int]]]:
    """nodes by id and ordered children lists from the JSON export (io.py:178-219)."""
    d = path_or_dict
    if not isinstance(d, dict):
        with open(d, 'r', encoding='utf-8') as f:
            d = json.load(f)
    nodes = {int(n['id']): n for n in d['nodes']}
    edges = d['links'] if 'links' in d else d['edges']
    kids: Dict[int, Dict[int, int]] = {i: {} for i in nodes}
    for e in edges:
        kids[int(e['target'])][int(e['idx'])] = int(e['source'])
    children = {i: [kids[i][k] for k in sorted(kids[i])] for i in nodes}
    return nodes, children


def evaluation_order(children: Dict[int, List[int]], root: int = 0) -> List[int]:
    """children before parents (the reverse of node.py's topological_order from the root)."""
    order, state = [], {}
    stack = [(root, 0)]
    while stack:
        n, k = stack.pop()
        if k == 0:
            if state.get(n) == 2:
                continue
            if state.get(n) == 1:
                raise ValueError("SPN structure is not a directed acyclic graph (DAG)")
            state[n] = 1
        if k < len(children[n]):
            stack.append((n, k + 1))
            c = children[n][k]
            if state.get(c) == 1:
                raise ValueError("SPN structure is not a directed acyclic graph (DAG)")
            if state.get(c) != 2:
                stack.append((c, 0))
        else:
            state[n] = 2
            order.append(n)
    return order


def leaf_log_likelihood(node: dict, col: np.ndarray) -> np.ndarray:
    """leaf.py log_likelihood of the four parametric leaves on one input column."""
    lls = np.zeros(len(col), dtype=np.float32)
    live = ~np.isnan(col)
    p = node['params']
    name = node['class']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if name == 'Bernoulli':
            lls[live] = ss.bernoulli.logpmf(col[live], p['p'])
        elif name == 'Categorical':
            # leaf.py:236-239: categories int64, probabilities float32
            dist = ss.rv_discrete(values=(np.array(p['categories'], np.int64), np.array(p['probabilities'], np.float32)))
            lls[live] = dist.logpmf(col[live].astype(np.int64, copy=False))
        elif name == 'Uniform':
            lls[live] = ss.uniform.logpdf(col[live], p['start'], p['width'])
        elif name == 'Gaussian':
            lls[live] = ss.norm.logpdf(col[live], p['mean'], p['stddev'])
        else:
            raise ValueError("Unknown node of type {}".format(name))
    return lls


def log_likelihood(path_or_dict, x: np.ndarray, return_results: bool = False):
    """inference.py:37-58 on the JSON export: float32 [B] (and the [n_nodes, B] table, row = node id)."""
    nodes, children = load(path_or_dict)
    x = np.asarray(x)
    ls = np.empty((max(nodes) + 1, len(x)), dtype=np.float32)
    for i in evaluation_order(children):
        n = nodes[i]
        if n['class'] == 'Sum':
            w = np.array(n['weights'], dtype=np.float32)                     # node.py:83-84
            stacked = np.stack([ls[c] for c in children[i]], axis=1)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                v = logsumexp(stacked, b=w, axis=1, keepdims=True)
        elif n['class'] == 'Product':
            v = np.sum(np.stack([ls[c] for c in children[i]], axis=1), axis=1, keepdims=True)
        else:
            v = leaf_log_likelihood(n, x[:, n['scope'][0]])[:, None]
        ls[i] = np.squeeze(np.maximum(v, FLOOR), axis=1)                     # inference.py:103
    return (ls[0], ls) if return_results else ls[0]
