"""Diagnostics (CALS_DIAG build): what an NNLS row costs in solves -- rows, calculate_sp solves, Cholesky
factorisations (the rest reuse the wave's cached factor), main-loop and inner-loop passes -- at C3's shape, for
a non-negative low-rank tensor + noise and for an all-positive noise tensor."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["CALS_TTM_TRACE"] = "1"
import ctypes as C
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [300, 300, 300]
ranks = inputs.ranks_1_to_20(256)
for name, X in (("low rank + noise", np.abs(inputs.low_rank_tensor(modes, 8, seed=1)[0]) + 0.05 * inputs.tensor(modes, 2)),
                ("|noise|", np.abs(inputs.tensor(modes, 0)))):
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 6, force_max_iter=1, update_method=1))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model([np.abs(f) for f in fs], lam))
    e.admit()
    buf = (C.c_uint64 * (16 * 2048))()
    prev = np.zeros(6)
    for sweep in range(1, 7):
        e.sweep(1); e.synchronize()
        e._chk(e.lib.cals_hip_debug_ttm_trace(e.h, buf, 16 * 2048))
        t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)[8 * 2048 + 1024:8 * 2048 + 1030]
        d = t - prev; prev = t.copy()
        print("%-17s sweep %d: rows %d, solves per row %.2f, factorisations per row %.2f, main passes %.2f, inner %.2f, rows starting all-passive %.2f" % (
            name, sweep, d[0], d[1] / d[0], d[2] / d[0], d[3] / d[0], d[4] / d[0], d[5] / d[0]), flush=True)
    e.close()
