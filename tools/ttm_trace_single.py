"""Diagnostics (CALS_DIAG build): ttm_kernel stage period with ONE workgroup on the whole GPU vs all
CUs busy -- separates chip-level contention from what the instruction stream itself costs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_TTM_TRACE"] = "1"
os.environ["CALS_HIP_TREE"] = "A"
import ctypes as C
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [160, 24, 300]          # first = mode 0 (10 m-tiles, one M block), s = mode 1, a = mode 2 (19 a-blocks)
for n_models, label in ((6, "1 column block"), (240, "40 column blocks")):
    ranks = [20] * n_models
    X = inputs.tensor(modes, 0)
    e = cc.Engine(modes, sum(ranks))
    assert e.tree == 1
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10**9, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(2); e.synchronize()
    buf = (C.c_uint64 * (16 * 2048))()
    e._chk(e.lib.cals_hip_debug_ttm_trace(e.h, buf, 16 * 2048))
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(8, 2, 2048)[:1, :, :4]
    for grp in (0, 1):
        vm, bar, per, n = (t[:, grp, k] for k in range(4))
        print("%s, CALS_TTM_TEAMS=%s, waves %s: stages %d period %.0f DMA wait %.0f barrier wait %.0f" % (
            label, os.environ.get("CALS_TTM_TEAMS", "-"), "0-3" if grp == 0 else "4-7", n.mean(), (per / n).mean(),
            (vm / n).mean(), (bar / n).mean()))
    e.close()
