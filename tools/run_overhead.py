"""Diagnostics: cals_hip_run (per-sweep host sync, eviction logic) vs back-to-back cals_hip_sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cp_cals_amd as cc
from cp_cals_amd import inputs
for modes, n_models, iters in (([100, 100, 100], 64, 50), ([40, 30, 20], 64, 50), ([300, 300, 300], 256, 10)):
    ranks = inputs.ranks_1_to_20(n_models)
    X = inputs.tensor(modes, 0)
    base = inputs.model_factors(modes, ranks, 1)
    for ls in (0, 1):
        t0 = time.time()
        e = cc.Engine(modes, sum(ranks)); t_create = time.time() - t0
        t0 = time.time(); e.set_tensor(X); t_set = time.time() - t0
        e.set_params(cc.default_params(max_iterations=iters, force_max_iter=1, line_search=ls))
        for fs, lam in base:
            e.enqueue(cc.Model(fs, lam))
        t0 = time.time(); rep = e.run(); t_run = time.time() - t0
        e.close()
        e = cc.Engine(modes, sum(ranks)); e.set_tensor(X)
        e.set_params(cc.default_params(max_iterations=10**9, force_max_iter=1, line_search=ls))
        for fs, lam in base:
            e.enqueue(cc.Model(fs, lam))
        t0 = time.time(); e.admit(); e.synchronize(); t_admit = time.time() - t0
        t0 = time.time(); e.sweep(iters); e.synchronize(); t_sweep = time.time() - t0
        e.close()
        print("%s models=%d ls=%d: create %.1f ms, set_tensor %.1f ms, admit %.1f ms | run(%d sweeps) %.2f ms (loop %.2f) vs sweep() %.2f ms" % (
            modes, n_models, ls, t_create * 1e3, t_set * 1e3, t_admit * 1e3, rep.iter, t_run * 1e3, rep.loop_ms, t_sweep * 1e3), flush=True)
