"""What a fresh cp_cals call pays before its first sweep at C3: engine creation (first in the process = HIP runtime
initialisation + code-object load; then again), cals_hip_set_tensor, admission of 256 models."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [300, 300, 300]
X = np.ascontiguousarray(inputs.tensor(modes, 0))
ranks = inputs.ranks_1_to_20(256)
base = inputs.model_factors(modes, ranks, 1)
for rnd in range(3):
    t0 = time.time(); e = cc.Engine(modes, 2656); t1 = time.time()
    e.set_tensor(X); t2 = time.time()
    ms = [cc.Model(fs, lam) for fs, lam in base]
    for m in ms:
        e.enqueue(m)
    n = e.admit(); e.synchronize(); t3 = time.time()
    e.sweep(1); e.synchronize(); t4 = time.time()
    e.sweep(1); e.synchronize(); t5 = time.time()
    print("engine #%d: create %.1f ms, set_tensor %.1f ms, enqueue + admit %d models %.1f ms, first sweep %.1f ms, second %.1f ms" % (
        rnd, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
    e.close()
