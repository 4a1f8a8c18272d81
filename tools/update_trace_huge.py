"""Diagnostics (CALS_DIAG build): phase cycles of the rank > 64 update launches (mode 0) for ONE model of rank r at
C3's shape -- huge_potrf_kernel's steps (a) / (b) / (c) and huge_solve_kernel's wait / DMA issue / block part /
triangle / rest (workgroup 0), in shader-clock cycles summed over the kernel's blocks."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_TTM_TRACE"] = "1"
os.environ["CALS_HUGE_NO_SIDE"] = "1"  # the factor launches alone on the main stream, not next to the MTTKRP
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
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)[16 * 2048 - 64:]
    print("rank", r, "| potrf (a) tiles %d, (b) diagonal %d, (c) below + stores %d | solve: wait %d, DMA issue %d, block part %d, "
          "triangle %d, rest %d" % tuple(t[[0, 1, 2, 8, 9, 10, 11, 12]]), flush=True)
    e.close()
