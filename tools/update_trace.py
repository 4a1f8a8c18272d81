"""Diagnostics (CALS_DIAG build): phase timing of update_kernel for one rank-20 model, mode 0."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_TTM_TRACE"] = "1"
import ctypes as C
import cp_cals_amd as cc
from cp_cals_amd import inputs
for modes, nm in (([100, 100, 100], 64), ([300, 300, 300], 256)):
    ranks = inputs.ranks_1_to_20(nm)
    X = inputs.tensor(modes, 0)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10**9, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(3); e.synchronize()
    buf = (C.c_uint64 * (16 * 2048))()
    e._chk(e.lib.cals_hip_debug_ttm_trace(e.h, buf, 16 * 2048))
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[16 * 2048 - 64:][:10]
    # update_body_lds stamps 0 1 2 3 5 6 7 (no stamp 4: the column scales and lambda are one phase there)
    names = ["hadamard", "cholesky", "rows: solve", "col stats + lambda", "scale pass", "pt + gramian"]
    tt = t[[0, 1, 2, 3, 5, 6, 7]]
    print(modes, "update_kernel phases (cycles @2.39 GHz):", ", ".join("%s %d" % (n, d) for n, d in zip(names, np.diff(tt))), "| total", t[7] - t[0], "| kernel entry -> descriptor %d -> body start %d" % (t[9] - t[8], t[0] - t[9]))
    e.close()
