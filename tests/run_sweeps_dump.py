"""Helper of tests/test_gpu_update_folds.py (run as a child process so that the engine's environment switches,
which it reads once per process, can differ between runs): a few forced sweeps of one configuration, results
dumped to an .npz.  usage: run_sweeps_dump.py OUT.npz I-J-K N_MODELS SWEEPS DTYPE PLAN LS [UPDATE_METHOD [RANKS]]
(RANKS: comma-separated, replaces ranks 1..20 cycled over N_MODELS; UPDATE_METHOD 1 = update::NNLS on |X|)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, shape, n_models, sweeps, dtype, plan, ls = sys.argv[1:8]
os.environ["CALS_HIP_TREE"] = plan
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402

modes = [int(x) for x in shape.split("-")]
ranks = inputs.ranks_1_to_20(int(n_models))
um = int(sys.argv[8]) if len(sys.argv) > 8 else 0
if len(sys.argv) > 9:
    ranks = [int(v) for v in sys.argv[9].split(",")]
X = inputs.tensor(modes, 3)
if um:
    X = np.abs(X)
e = cc.Engine(modes, sum(ranks), dtype=dtype)
e.set_tensor(X)
e.set_params(cc.default_params(max_iterations=int(sweeps), force_max_iter=1, line_search=int(ls), line_search_interval=2,
                               update_method=um))
models = [cc.Model([np.abs(f) for f in fs] if um else fs, lam) for fs, lam in inputs.model_factors(modes, ranks, 7)]
for m in models:
    e.enqueue(m)
rep = e.run()
arrs = {"tree": np.array([e.tree]), "iter": np.array([rep.iter])}
for k, m in enumerate(models):
    for n, f in enumerate(m.factors):
        arrs["f%d_%d" % (k, n)] = f
    arrs["lam%d" % k] = m.lam
    arrs["err%d" % k] = np.array([m.error, m.fit])
e.close()
np.savez(out, **arrs)
