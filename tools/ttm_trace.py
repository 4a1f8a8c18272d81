"""Diagnostics: where the waves of ttm_kernel spend a stage at C3 (library built with CALS_DIAG=1).
Prints, for waves 0 (barrier mid-slab) and 4 (barrier at slab start) of 8 workgroups, the mean stage
period and the mean time at the s_waitcnt vmcnt(0) and at the s_barrier, over steady-state stages of
the first M block (MT = 10: 40 MFMAs per wave per stage, ideal period 2 x 40 x 64 = 5120 cycles)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_TTM_TRACE"] = "1"
os.environ.setdefault("CALS_HIP_TREE", "M")
import ctypes as C
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [300, 300, 300]
ranks = inputs.ranks_1_to_20(256)
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
t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(8, 2, 2048)[:, :, :4]
for grp in (0, 1):
    vm, bar, per, n = (t[:, grp, k] for k in range(4))
    print("waves %s: stages %d  period %.0f  DMA wait (vmcnt) %.0f  barrier wait %.0f   [per workgroup periods: %s]" % (
        "0-3" if grp == 0 else "4-7", n.mean(), (per / n).mean(), (vm / n).mean(), (bar / n).mean(),
        " ".join("%.0f" % v for v in per / n)))
