"""Golden vectors for the vanilla (node-graph) SPN stack, BASELINE config 1: the reference learns / builds an SPN,
exports it with its own JSON writer (deeprob/spn/structure/io.py:59-70), re-loads the export and evaluates
`log_likelihood` (deeprob/spn/algorithms/inference.py:37-58) on stored inputs.  Outputs (data only):
tests/golden/spn_<name>.json (the reference's export) and tests/golden/spn_<name>.npz (inputs, per-node and root LLs).

    cd tools && PYTHONPATH=/root/reference python3 gen_golden_spn.py
"""
import io
import os
import warnings

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _emit(name, root, x):
    from deeprob.spn.structure.io import save_spn_json, load_spn_json
    from deeprob.spn.algorithms.inference import log_likelihood
    path = os.path.join(OUT, 'spn_%s.json' % name)
    save_spn_json(root, path)
    again = load_spn_json(path)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ll, per_node = log_likelihood(again, x, return_results=True)
    np.savez_compressed(os.path.join(OUT, 'spn_%s.npz' % name), x=x.astype(np.float32),
                        ll=np.asarray(ll, dtype=np.float32), per_node=np.asarray(per_node, dtype=np.float32))
    print(name, 'nodes', per_node.shape[0], 'mean LL', float(np.mean(ll[np.isfinite(ll)])))


def gen_binary16():
    """16 binary variables, 1000 samples from a 4-component Bernoulli mixture."""
    from deeprob.spn.learning.wrappers import learn_estimator
    from deeprob.spn.structure.leaf import Bernoulli
    rs = np.random.RandomState(42)
    z = rs.randint(0, 4, 1000)
    P = rs.rand(4, 16)
    data = (rs.rand(1000, 16) < P[z]).astype(np.float32)
    root = learn_estimator(data, [Bernoulli] * 16, [[0, 1]] * 16, learn_leaf='mle', split_rows='kmeans',
                           split_cols='gvs', min_rows_slice=64, random_state=42)
    _emit('binary16', root, data)
    # the same circuit under marginalisation: 30 % NaN, one fully marginalised row, one value outside {0,1}
    x = data.copy()
    x[rs.rand(*x.shape) < 0.3] = np.nan
    x[0, :] = np.nan
    x[1, 3] = 2.0
    from deeprob.spn.structure.io import load_spn_json
    from deeprob.spn.algorithms.inference import log_likelihood
    again = load_spn_json(os.path.join(OUT, 'spn_binary16.json'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ll, per_node = log_likelihood(again, x, return_results=True)
    np.savez_compressed(os.path.join(OUT, 'spn_binary16_nan.npz'), x=x, ll=np.asarray(ll, dtype=np.float32),
                        per_node=np.asarray(per_node, dtype=np.float32))


def gen_mixed():
    """A hand-built DAG over Gaussian, Categorical and Bernoulli leaves with a shared sub-circuit."""
    from deeprob.spn.structure.leaf import Bernoulli, Categorical, Gaussian
    from deeprob.spn.structure.node import Sum, Product, assign_ids
    rs = np.random.RandomState(7)
    g0 = [Gaussian(0, mean=float(m), stddev=float(s)) for m, s in [(-1.0, 0.5), (0.7, 1.3), (2.5, 0.2)]]
    g1 = [Gaussian(1, mean=float(m), stddev=float(s)) for m, s in [(0.0, 1.0), (4.0, 2.0)]]
    c2 = [Categorical(2, categories=[0, 1, 2, 3], probabilities=list(p)) for p in rs.dirichlet(np.ones(4), 2)]
    b3 = [Bernoulli(3, p=0.2), Bernoulli(3, p=0.85)]
    s0 = Sum(children=g0, weights=[0.2, 0.5, 0.3])
    s1 = Sum(children=g1, weights=[0.6, 0.4])
    s2 = Sum(children=c2, weights=[0.35, 0.65])
    shared = Product(children=[s2, b3[0]])                      # used by both branches below
    p_a = Product(children=[s0, s1, shared])
    p_b = Product(children=[g0[1], g1[0], shared])
    p_c = Product(children=[s0, g1[1], c2[0], b3[1]])
    root = assign_ids(Sum(children=[p_a, p_b, p_c], weights=[0.5, 0.125, 0.375]))
    n = 257
    x = np.stack([rs.randn(n) * 2, rs.randn(n) * 3 + 1, rs.randint(0, 4, n).astype(np.float64),
                  rs.randint(0, 2, n).astype(np.float64)], axis=1).astype(np.float32)
    x[rs.rand(*x.shape) < 0.2] = np.nan
    x[0, :] = np.nan
    x[1, 2] = 7.0            # category outside the support
    x[2, 0] = 1e4            # far tail
    _emit('mixed4', root, x)


if __name__ == '__main__':
    gen_binary16()
    gen_mixed()
