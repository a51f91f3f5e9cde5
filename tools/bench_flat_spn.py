"""Measurement: the flat-SPN evaluator on BASELINE config 1 (16 binary variables, the reference-learned 72-node
circuit) next to the oracle (= the reference's numpy / scipy pass) on the host.  usage: bench_flat_spn.py [B]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'deeprob-kit_amd'), ROOT]
import numpy as np
import torch
from deeprob.spn.structure.io import load_spn_json
from deeprob.spn.algorithms.inference import log_likelihood
from oracle import flat_spn_oracle as forc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
path = os.path.join(ROOT, 'tests', 'golden', 'spn_binary16.json')
spn = load_spn_json(path)
rs = np.random.RandomState(0)
x = (rs.rand(B, 16) < 0.5).astype(np.float32)
xd = torch.from_numpy(x).cuda()
for _ in range(5):
    ll = log_likelihood(spn, xd)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K):
    ll = log_likelihood(spn, xd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
small = x[:1000]
xs = torch.from_numpy(small).cuda()
for _ in range(5):
    log_likelihood(spn, xs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    log_likelihood(spn, xs)
torch.cuda.synchronize()
dts = (time.perf_counter() - t0) / 200
n_cpu = min(B, 100000)
t0 = time.perf_counter()
want = forc.log_likelihood(path, x[:n_cpu])
dc = time.perf_counter() - t0
err = float(np.max(np.abs(ll[:n_cpu].cpu().numpy() - want) / np.maximum(1, np.abs(want))))
print(json.dumps({'workload': 'vanilla SPN log_likelihood, 16 binary vars, 72 nodes', 'batch': B,
                  'ms_per_batch': dt * 1e3, 'll_per_s': B / dt, 'ms_per_1000_samples_call': dts * 1e3,
                  'oracle_cpu_ll_per_s': n_cpu / dc, 'oracle_sample': n_cpu, 'max_rel_err_vs_oracle': err}))
