"""Random smooth / decomposable circuits in the reference's JSON node-link layout (test helper)."""
import numpy as np


def random_circuit(n_features: int, seed: int, n_sum: int = 3, kinds=('Gaussian', 'Bernoulli', 'Categorical', 'Uniform')):
    """Recursive random structure: a sum over `n_sum` products, each splitting the scope in two random halves,
    down to single-variable leaves whose family is fixed per variable (so that inputs can be drawn per column).
    Returns (json dict, leaf family per variable)."""
    rs = np.random.RandomState(seed)
    family = [kinds[rs.randint(len(kinds))] for _ in range(n_features)]
    nodes, links = [], []

    def new(attrs):
        attrs['id'] = len(nodes)
        nodes.append(attrs)
        return attrs['id']

    def leaf(v):
        f = family[v]
        if f == 'Gaussian':
            prm = {'mean': float(rs.randn() * 2), 'stddev': float(0.3 + rs.rand() * 2)}
        elif f == 'Bernoulli':
            prm = {'p': float(0.05 + 0.9 * rs.rand())}
        elif f == 'Categorical':
            p = rs.dirichlet(np.ones(5)).astype(np.float32)
            p = (p / p.sum()).astype(np.float32)
            prm = {'categories': [0, 1, 2, 3, 4], 'probabilities': [float(q) for q in p]}
        else:
            prm = {'start': float(rs.randn()), 'width': float(0.5 + 3 * rs.rand())}
        return new({'class': f, 'scope': [v], 'params': prm})

    def build(scope):
        if len(scope) == 1:
            k = rs.randint(1, 3)
            if k == 1:
                return leaf(scope[0])
            me = new({'class': 'Sum', 'scope': list(scope), 'weights': None})
            kids = [leaf(scope[0]) for _ in range(k + 1)]
        else:
            me = new({'class': 'Sum', 'scope': list(scope), 'weights': None})
            kids = []
            for _ in range(n_sum):
                perm = list(rs.permutation(scope))
                cut = rs.randint(1, len(scope))
                left, right = sorted(perm[:cut]), sorted(perm[cut:])
                p = new({'class': 'Product', 'scope': list(scope)})
                for idx, part in enumerate((left, right)):
                    links.append({'source': build(part), 'target': p, 'idx': idx})
                kids.append(p)
        w = rs.dirichlet(np.ones(len(kids))).astype(np.float32)
        w = np.round(w.astype(np.float64), 8)
        w[-1] = np.round(1.0 - w[:-1].sum(), 8)
        nodes[me]['weights'] = [float(v) for v in w]
        for idx, c in enumerate(kids):
            links.append({'source': c, 'target': me, 'idx': idx})
        return me

    root = build(list(range(n_features)))
    assert root == 0
    return {'directed': True, 'multigraph': False, 'graph': {}, 'nodes': nodes, 'links': links}, family


def random_inputs(family, B: int, seed: int, nan_rate: float = 0.2) -> np.ndarray:
    rs = np.random.RandomState(seed)
    cols = []
    for f in family:
        if f == 'Gaussian':
            cols.append(rs.randn(B) * 2.5)
        elif f == 'Bernoulli':
            cols.append(rs.randint(0, 2, B).astype(np.float64))
        elif f == 'Categorical':
            cols.append(rs.randint(0, 6, B).astype(np.float64))      # 5 is outside the support
        else:
            cols.append(rs.randn(B) * 2)
    x = np.stack(cols, axis=1).astype(np.float32)
    x[rs.rand(*x.shape) < nan_rate] = np.nan
    return x
