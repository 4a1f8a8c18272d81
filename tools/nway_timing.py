"""Sweep time of N > 3 tensors (explicit partial Khatri-Rao + the fused MTTKRP kernel; plan 0 only) against the
2 N prod(I) R flop model.  Usage: python tools/nway_timing.py [sweeps]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for modes, n_models in (([80, 80, 80, 80], 64), ([40, 50, 60, 70], 128), ([24, 24, 24, 24, 24], 64), ([300, 300, 300], 64)):
    ranks = [1 + (k % 20) for k in range(n_models)]
    R = sum(ranks)
    X = np.random.default_rng(0).uniform(-1, 1, size=int(np.prod(modes)))
    e = cc.Engine(modes, R, device=0)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, ranks, seed=1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(2)
    e.synchronize()
    t0 = time.perf_counter()
    e.sweep(sweeps)
    e.synchronize()
    dt = (time.perf_counter() - t0) / sweeps
    flops = 2.0 * len(modes) * float(np.prod(modes)) * R
    print("%-16s %3d models R=%4d: %.3f ms per sweep = %.1f TFLOP/s of MTTKRP work (plan %d)" % (
        "x".join(map(str, modes)), n_models, R, dt * 1e3, flops / dt * 1e-12, e.tree), flush=True)
    e.close()
