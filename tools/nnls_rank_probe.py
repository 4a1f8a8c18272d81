"""Probe: one model of rank r alone at C3's shape under update::NNLS -- time per sweep and the sticky NNLS status
(bit 1 = a row reached the exchange bound).  Usage: python tools/nnls_rank_probe.py 47 48 49"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402

modes = [300, 300, 300]
X = np.abs(inputs.tensor(modes, seed=0))
small = [1 + (k % 20) for k in range(255)]
for r in [int(v) for v in sys.argv[1:]] or [47, 48, 49]:
    fs, lam = inputs.model_factors(modes, small + [r], seed=1)[-1]
    e = cc.Engine(modes, r, device=0)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=12, force_max_iter=1, update_method=1))
    m = cc.Model([np.abs(f) for f in fs], lam.copy())
    e.enqueue(m)
    t0 = time.perf_counter()
    rep = e.run()
    dt = time.perf_counter() - t0
    zeros = sum(int((f == 0.0).sum()) for f in m.factors)
    print("rank %d: %.2f ms per sweep, nnls_status %d, zeros in the factors %d of %d, fit %.6f" % (
        r, dt / 12 * 1e3, rep.nnls_status, zeros, 3 * 300 * r, m.fit), flush=True)
    e.close()
