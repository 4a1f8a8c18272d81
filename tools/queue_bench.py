"""cp_cals as users run it (driver.cpp:177-180): a queue longer than the buffer, tolerance-driven
eviction, admission and compress every sweep.  Reports sweeps, wall time per sweep of cals_hip_run
against the back-to-back sweep time of the same buffer width, and models/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cp_cals_amd as cc
from cp_cals_amd import inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
buffer = int(sys.argv[3]) if len(sys.argv) > 3 else 2656
modes = [n, n, n]
ranks = inputs.ranks_1_to_20(n_models)
X = inputs.low_rank_tensor(modes, 10, seed=1)[0] + 0.1 * inputs.tensor(modes, 2)
e = cc.Engine(modes, buffer)
e.set_tensor(X)
e.set_params(cc.default_params(max_iterations=40, tol=1e-4, line_search=1, line_search_interval=5))
models = [cc.Model(fs, lam) for fs, lam in inputs.model_factors(modes, ranks, 1)]
for m in models:
    e.enqueue(m)
e.synchronize()
t0 = time.perf_counter()
rep = e.run()
dt = time.perf_counter() - t0
its = np.array([m.iters for m in models])
print("models %d, buffer %d cols: %d sweeps in %.3f s (%.3f ms/sweep), %.1f models/s; model iterations "
      "min/mean/max %d/%.1f/%d; loop_ms %.1f total_ms %.1f" % (
          n_models, buffer, rep.iter, dt, dt / rep.iter * 1e3, n_models / dt, its.min(), its.mean(), its.max(),
          rep.loop_ms, rep.total_ms), flush=True)
# column-sweeps actually computed vs the time: how full was the buffer
print("sum of model iterations x rank = %d column-sweeps; buffer x sweeps = %d (occupancy %.2f)" % (
    int((its * np.array(ranks)).sum()), buffer * rep.iter, float((its * np.array(ranks)).sum()) / (buffer * rep.iter)))
e.close()
