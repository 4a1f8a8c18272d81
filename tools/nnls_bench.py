"""Cost of the NNLS update at BASELINE config 3's shape (300^3, 256 models of rank 1..20, fp64):
it/s with update::NNLS vs UNCONSTRAINED, and the per-sweep time of the update stage (nnls_kernel +
update_kernel) from the engine's hipEvent statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cp_cals_amd as cc
from cp_cals_amd import inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
models = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sweeps = 10
modes = [n, n, n]
ranks = inputs.ranks_1_to_20(models)
X = np.abs(inputs.low_rank_tensor(modes, 8, seed=1)[0]) + 0.05 * inputs.tensor(modes, 2)
for um in (0, 1):
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 6, force_max_iter=1, update_method=um))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(3)
    e.synchronize()
    t0 = time.time()
    e.sweep(sweeps)
    e.synchronize()
    dt = (time.time() - t0) / sweeps
    e.set_profiling(True)
    e.reset_kernel_stats()
    e.sweep(3)
    e.synchronize()
    ks = e.kernel_stats()
    e.set_profiling(False)
    print("update_method=%d: %.3f ms/sweep (%.1f it/s); update stage %.3f ms/sweep over %d launches"
          % (um, dt * 1e3, 1.0 / dt, ks.update_ms / 3, ks.update_launches), flush=True)
    e.close()
