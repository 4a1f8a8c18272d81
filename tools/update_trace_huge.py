"""Diagnostics (CALS_DIAG build): phase timing of update_body_huge (mode 0) for ONE model of rank r at C3's shape."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_TTM_TRACE"] = "1"
import ctypes as C
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [300, 300, 300]
X = inputs.tensor(modes, 0)
for r in [int(v) for v in sys.argv[1:]] or [65, 128, 256]:
    e = cc.Engine(modes, r)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10**9, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, [r], 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(3); e.synchronize()
    buf = (C.c_uint64 * (16 * 2048))()
    e._chk(e.lib.cals_hip_debug_ttm_trace(e.h, buf, 16 * 2048))
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[16 * 2048 - 64:][:7]
    names = ["hadamard", "cholesky", "row solves", "col stats", "scale pass", "gramian"]
    print("rank", r, "update_body_huge phases (us @100 MHz memtime?):", ", ".join("%s %d" % (n, d) for n, d in zip(names, np.diff(t))), "| total", t[6] - t[0], flush=True)
    e.close()
