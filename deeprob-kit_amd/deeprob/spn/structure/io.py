"""Loading a vanilla (node-graph) SPN from the reference's JSON format (deeprob/spn/structure/io.py:59-102,
133-220) into the flat arrays the HIP evaluator walks.

Only the evaluation side of the node-graph stack is in scope (BASELINE config 1): ``load_spn_json`` returns a
:class:`FlatSpn` -- node kinds, parameters and a child list per node over the file's node ids -- instead of a tree
of ``Node`` objects; ``deeprob.spn.algorithms.inference.log_likelihood`` evaluates it on a HIP device.
"""
import json
import math
import os
from typing import IO, Dict, List, Optional, Union

import numpy as np

KIND = {'Sum': 0, 'Product': 1, 'Bernoulli': 2, 'Categorical': 3, 'Uniform': 4, 'Gaussian': 5}


class FlatSpn:
    """A labelled SPN as arrays over node ids ``0..n_nodes-1`` (root = 0, as ``digraph_to_spn`` returns
    ``nodes[0]``, reference io.py:219): see ``include/deeprob_hip.h`` (dpk_flat_spn_forward) for the layout."""

    def __init__(self, nodes: Dict[int, dict], children: Dict[int, List[int]], root: int = 0):
        ids = sorted(nodes)
        if ids != list(range(len(ids))):
            raise ValueError("SPN is not correctly labeled: node ids must be 0..n-1")
        n = len(ids)
        self.n_nodes, self.root = n, root
        self.scopes = [sorted(int(v) for v in nodes[i]['scope']) for i in ids]
        self.classes = [nodes[i]['class'] for i in ids]
        self.children = [list(children.get(i, [])) for i in ids]
        self.order = np.asarray(self._evaluation_order(), dtype=np.int32)
        self.kind = np.zeros(n, np.int32)
        self.arg0, self.arg1, self.arg2 = (np.zeros(n, np.int32) for _ in range(3))
        self.par0, self.par1 = np.zeros(n, np.float64), np.zeros(n, np.float64)
        child_index, child_weight, cat_value, cat_logp = [], [], [], []
        for i in ids:
            node, name, kids = nodes[i], nodes[i]['class'], self.children[i]
            if name not in KIND:
                raise ValueError("Unknown node of type {}".format(name))
            self.kind[i] = KIND[name]
            if name in ('Sum', 'Product'):
                if not kids:
                    raise ValueError("Inner node {} has no children".format(i))
                self.arg0[i], self.arg1[i] = len(child_index), len(kids)
                if name == 'Sum':
                    weights = np.asarray(node['weights'], dtype=np.float32)      # node.py:83-84
                    if len(weights) != len(kids):
                        raise ValueError("Each child of a sum node must be associated a weight")
                    if not np.isclose(np.sum(weights), 1.0):
                        raise ValueError("Weights don't sum up to 1")
                    child_weight += [float(w) for w in weights]
                else:
                    child_weight += [1.0] * len(kids)
                child_index += kids
                continue
            if kids:
                raise ValueError("Leaf node {} has children".format(i))
            if len(self.scopes[i]) != 1:
                raise ValueError("Leaf node {} must have a single variable in its scope".format(i))
            self.arg0[i] = self.scopes[i][0]
            prm = node.get('params', {})
            if name == 'Bernoulli':            # scipy.stats.bernoulli.logpmf: log p at 1, log1p(-p) at 0
                p = float(prm['p'])
                self.par0[i] = math.log(p) if p > 0.0 else -math.inf
                self.par1[i] = math.log1p(-p) if p < 1.0 else -math.inf
            elif name == 'Categorical':        # rv_discrete over float32 probabilities (leaf.py:236-239)
                cats = np.asarray(prm['categories'], dtype=np.int64)
                probs = np.asarray(prm['probabilities'], dtype=np.float32)
                if len(cats) != len(probs):
                    raise ValueError("Each category must be associated a probability")
                if not np.isclose(np.sum(probs), 1.0):
                    raise ValueError("Probabilities parameter must sum up to 1")
                self.arg1[i], self.arg2[i] = len(cat_value), len(cats)
                cat_value += [int(c) for c in cats]
                with np.errstate(divide='ignore'):
                    cat_logp += [float(v) for v in np.log(probs.astype(np.float64))]
            elif name == 'Uniform':
                self.par0[i], self.par1[i] = float(prm['start']), float(prm['width'])
            else:
                self.par0[i], self.par1[i] = float(prm['mean']), float(prm['stddev'])
        self.child_index = np.asarray(child_index or [0], dtype=np.int32)
        self.child_weight = np.asarray(child_weight or [0.0], dtype=np.float32)
        self.cat_value = np.asarray(cat_value or [0], dtype=np.int32)
        self.cat_logp = np.asarray(cat_logp or [0.0], dtype=np.float32)
        self.n_slots, self.node_slot = self._allocate_slots()
        self.child_slot = self.node_slot[self.child_index].astype(np.int32) if child_index else self.child_index
        self.n_features = 1 + max(max(s) for s in self.scopes)
        self._checked = False
        self._device = {}

    def _evaluation_order(self) -> List[int]:
        """Children before parents, every node reachable from the root once; a cycle is an error (reference:
        topological_order returning None, evaluation.py:79-81)."""
        order, state = [], [0] * self.n_nodes
        stack = [(self.root, 0)]
        while stack:
            node, k = stack.pop()
            if k == 0:
                if state[node] == 2:
                    continue
                state[node] = 1
            if k < len(self.children[node]):
                stack.append((node, k + 1))
                c = self.children[node][k]
                if state[c] == 1:
                    raise ValueError("SPN structure is not a directed acyclic graph (DAG)")
                if state[c] == 0:
                    stack.append((c, 0))
            else:
                state[node] = 2
                order.append(node)
        if len(order) != self.n_nodes:
            raise ValueError("SPN is not correctly labeled: {} nodes are not reachable from the root".format(
                self.n_nodes - len(order)))
        return order

    def _allocate_slots(self):
        """Rows of the evaluator's on-chip value table: a node keeps its row until its last parent (in evaluation
        order) has been computed, then the row returns to the free list."""
        position = {int(n): t for t, n in enumerate(self.order)}
        last_use = [position[i] for i in range(self.n_nodes)]        # a node nobody reads dies where it is born
        for parent in range(self.n_nodes):
            for c in self.children[parent]:
                last_use[c] = max(last_use[c], position[parent])
        last_use[self.root] = self.n_nodes                           # the root's row is read at the end
        expiring = {}
        for i, t in enumerate(last_use):
            expiring.setdefault(t, []).append(i)
        slot = np.zeros(self.n_nodes, np.int32)
        free, used = [], 0
        for t, node in enumerate(self.order):
            node = int(node)
            if free:
                slot[node] = free.pop()
            else:
                slot[node] = used
                used += 1
            # rows whose last reader is this node are free from the next node on (a node's own row is written
            # after its children's rows have been read)
            for dead in expiring.get(t, []):
                free.append(int(slot[dead]))
        return used, slot

    def check(self):
        """Smoothness and decomposability, as ``check_spn`` enforces before every evaluation (reference
        evaluation.py:67, utils/validity.py)."""
        if self._checked:
            return
        for i in range(self.n_nodes):
            kids, scope = self.children[i], self.scopes[i]
            if self.classes[i] == 'Sum':
                if any(self.scopes[c] != scope for c in kids):
                    raise ValueError("SPN is not smooth: children of sum node #{} have different scopes".format(i))
            elif self.classes[i] == 'Product':
                merged = sum((self.scopes[c] for c in kids), [])
                if len(merged) != len(set(merged)) or sorted(merged) != scope:
                    raise ValueError(
                        "SPN is not decomposable: children of product node #{} don't have disjoint scopes".format(i))
        self._checked = True

    def device_arrays(self, device):
        """The flat arrays as tensors on ``device`` (uploaded once per device)."""
        import torch
        key = str(device)
        if key not in self._device:
            names = ('order', 'kind', 'arg0', 'arg1', 'arg2', 'par0', 'par1', 'child_index', 'child_weight',
                     'cat_value', 'cat_logp', 'node_slot', 'child_slot')
            self._device[key] = {k: torch.from_numpy(getattr(self, k)).to(device) for k in names}
        return self._device[key]


def load_spn_json(f: Union[IO, os.PathLike, str], leaves: Optional[list] = None) -> FlatSpn:
    """
    Load SPN from file by using the JSON format (reference io.py:73-102).

    :param f: A file-like object or a filepath of the input JSON file.
    :param leaves: Custom leaf classes are not evaluated by the HIP path.
    :return: The loaded SPN, flattened.
    :raises ValueError: If a node class is unknown or the structure is not a valid labelled DAG.
    """
    if leaves:
        raise NotImplementedError("custom leaf classes are outside the HIP evaluator (Bernoulli, Categorical, "
                                  "Uniform and Gaussian leaves are built in)")
    if isinstance(f, (os.PathLike, str)):
        with open(f, 'r', encoding='utf-8') as file:
            data = json.load(file)
    else:
        data = json.load(f)
    return digraph_to_spn(data)


def digraph_to_spn(data: dict) -> FlatSpn:
    """Node-link data (networkx ``node_link_data``: ``nodes`` with ``id / class / scope / weights / params``, and
    ``links`` -- ``edges`` in newer networkx -- child -> parent with the child's position ``idx``; reference
    io.py:150-175, :178-219) to a :class:`FlatSpn`."""
    nodes = {int(n['id']): n for n in data['nodes']}
    slots: Dict[int, Dict[int, int]] = {i: {} for i in nodes}
    for e in data['links'] if 'links' in data else data['edges']:
        child, parent, idx = int(e['source']), int(e['target']), int(e['idx'])
        if child not in nodes or parent not in nodes:
            raise ValueError("Edge ({}, {}) refers to a missing node".format(child, parent))
        slots[parent][idx] = child
    children = {}
    for i, s in slots.items():
        if sorted(s) != list(range(len(s))):
            raise ValueError("Children positions of node {} are not 0..{}".format(i, len(s) - 1))
        children[i] = [s[k] for k in range(len(s))]
    return FlatSpn(nodes, children)
