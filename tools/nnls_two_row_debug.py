"""Debug aid for the two-rows-per-wavefront NNLS kernel: one sweep of one small model, per-row error of every mode's
factor against the oracle (rows 2k / 2k+1 of a wave's pair are lane groups 0 / 1)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cp_cals_amd as cc
from cp_cals_amd import inputs
import oracle as O
from helpers import make_models
modes = [12, 10, 8]
for ranks in ([1], [2], [4], [6], [8], [16]):
    X, _, _ = inputs.low_rank_tensor(modes, 4, seed=31)
    X = np.abs(X) + 0.05 * inputs.tensor(modes, 32)
    base = make_models(inputs, modes, ranks, seed=1)
    e = cc.Engine(modes, sum(ranks)); e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=1, force_max_iter=1, update_method=1))
    gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    for m in gm: e.enqueue(m)
    rep = e.run()
    om = [O.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    ro = O.cp_cals(X, modes, om, O.default_params(max_iterations=1, force_max_iter=1, buffer_size=sum(ranks), update_method=1))
    print("ranks", ranks, "status dev", rep.nnls_status, "oracle", ro.nnls_status)
    for n in range(3):
        err = np.abs(gm[0].factors[n] - om[0].factors[n]).max(axis=1)
        print("  mode", n, "per-row max err:", " ".join("%.1e" % v for v in err))
    if ranks == [2]:
        print("  dev mode0:\n", np.round(gm[0].factors[0], 4).T, "\n  oracle mode0:\n", np.round(om[0].factors[0], 4).T, "\n lam", gm[0].lam, om[0].lam)
    e.close()
