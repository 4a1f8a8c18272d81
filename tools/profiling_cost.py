"""Cost of the engine's live kernel statistics (hipEvent pairs around every launch): sweeps/s with
profiling off and on, c2 and c3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cp_cals_amd as cc
from cp_cals_amd import inputs

for name, modes, k, ls, sweeps in (("c2", [100] * 3, 64, 0, 300), ("c3", [300] * 3, 256, 1, 40)):
    ranks = inputs.ranks_1_to_20(k)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(inputs.tensor(modes, 0))
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, line_search=ls))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(10)
    e.synchronize()
    for level in (0, 1, 0, 1):
        e.set_profiling(bool(level))
        e.reset_kernel_stats()
        e.synchronize()
        t0 = time.perf_counter()
        e.sweep(sweeps)
        e.synchronize()
        dt = (time.perf_counter() - t0) / sweeps
        print("%s profiling=%d: %.1f us/sweep, %.1f it/s" % (name, level, dt * 1e6, 1 / dt), flush=True)
    e.close()
