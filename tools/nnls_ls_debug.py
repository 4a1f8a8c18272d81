"""Debug aid: first sweep at which the device and the oracle part ways (NNLS + line search)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cp_cals_amd as cc
import oracle as O
from cp_cals_amd import inputs
from helpers import make_models, rel

modes, ranks = [20, 20, 20], [2, 3, 4, 5, 20, 17]
X, _, _ = inputs.low_rank_tensor(modes, 5, seed=3)
X = np.abs(X) + 0.3 * inputs.tensor(modes, 4)
method = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for iters in range(1, 26):
    kw = dict(max_iterations=iters, force_max_iter=1, line_search=1, line_search_interval=5,
              line_search_method=method, update_method=1)
    base = make_models(inputs, modes, ranks, seed=1)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(**kw))
    gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    for m in gm:
        e.enqueue(m)
    rep = e.run()
    e.close()
    om = [O.Model(fs, lam) for fs, lam, _ in base]
    ro = O.cp_cals(X, modes, om, O.default_params(mttkrp_method=O.MTTKRP, buffer_size=sum(ranks), **kw))
    worst = [max(rel(fa, fb) for fa, fb in zip(a.factors, b.factors)) for a, b in zip(gm, om)]
    print(iters, (rep.ls_performed, rep.ls_failed), (ro.ls_performed, ro.ls_failed),
          ["%.1e" % w for w in worst], ["%.3e" % abs(a.error - b.error) for a, b in zip(gm, om)],
          "min dev %.3e oracle %.3e" % (min(f.min() for m in gm for f in m.factors), min(f.min() for m in om for f in m.factors)),
          "nan", sum(int(np.isnan(f).sum()) for m in gm for f in m.factors), sum(int(np.isnan(f).sum()) for m in om for f in m.factors),
          flush=True)
