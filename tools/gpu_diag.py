"""GPU bring-up diagnostics (not a test): compares the HIP path with the oracle stage by stage and
prints errors and timings instead of asserting.  Run on the GPU box: python tools/gpu_diag.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402
import oracle as O  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def build_models(modes, ranks, seed=1, jk=None):
    mf = inputs.model_factors(modes, ranks, seed)
    gm, om = [], []
    for k, (fs, lam) in enumerate(mf):
        j = None if jk is None else jk[k]
        if j is not None:
            fs = [f.copy() for f in fs]
            fs[j[0]][j[1], :] *= 0.0
        gm.append(cc.Model([f.copy() for f in fs], lam.copy(), jk=j))
        om.append(O.Model([f.copy() for f in fs], lam.copy(), jk=j))
    return gm, om


def check_mttkrp(modes, ranks, seed=0):
    X = inputs.tensor(modes, seed)
    gm, om = build_models(modes, ranks)
    R = sum(ranks)
    e = cc.Engine(modes, R)
    e.set_tensor(X)
    for m in gm:
        e.enqueue(m)
    e.admit()
    facs = [np.asfortranarray(np.hstack([m.factors[n] for m in om])) for n in range(len(modes))]
    out = []
    for n in range(len(modes)):
        G = e.debug_mttkrp(n)
        Go = O.mttkrp(X, modes, facs, n, O.MTTKRP)
        out.append(rel(G, Go))
    xn, jk = e.debug_norms()
    out.append(abs(xn - np.linalg.norm(X)) / np.linalg.norm(X))
    out.append(rel(jk, O.jk_norms(X, modes)))
    e.close()
    print("mttkrp", modes, "R=%d" % R, " ".join("%.2e" % v for v in out), flush=True)


def check_run(modes, ranks, iters, ls=0, jk=None, buffer=None, tol=None, interval=5, tag=""):
    X = inputs.tensor(modes, 3) if tol is None else inputs.low_rank_tensor(modes, 5, seed=9)[0]
    gm, om = build_models(modes, ranks, jk=jk)
    R = sum(ranks) if buffer is None else buffer
    e = cc.Engine(modes, R)
    e.set_tensor(X)
    force = 1 if tol is None else 0
    p = cc.default_params(max_iterations=iters, force_max_iter=force, line_search=ls,
                          line_search_interval=interval, tol=1e-7 if tol is None else tol)
    e.set_params(p)
    for m in gm:
        e.enqueue(m)
    t = time.time()
    rep = e.run()
    tg = time.time() - t
    po = O.default_params(max_iterations=iters, force_max_iter=force, line_search=ls,
                          line_search_interval=interval, buffer_size=R, mttkrp_method=O.MTTKRP,
                          tol=1e-7 if tol is None else tol)
    t = time.time()
    ro = O.cp_cals(X, modes, om, po)
    to = time.time() - t
    worst = 0.0
    werr = 0.0
    wl = 0.0
    bad_iters = 0
    for a, b in zip(gm, om):
        for fa, fb in zip(a.factors, b.factors):
            worst = max(worst, rel(fa, fb))
        wl = max(wl, rel(a.lam, b.lam))
        if np.isfinite(b.error) and b.error < 1e300:
            werr = max(werr, abs(a.error - b.error) / max(abs(b.error), 1e-300))
        bad_iters += int(a.iters != b.iters)
    print("run%s" % tag, modes, "models=%d" % len(ranks), "iters=%d ls=%d" % (iters, ls),
          "sweeps g/o %d/%d" % (rep.iter, ro.iter),
          "factor rel %.2e lam rel %.2e err rel %.2e iters!= %d" % (worst, wl, werr, bad_iters),
          "ls g %d/%d o %d/%d" % (rep.ls_performed, rep.ls_failed, ro.ls_performed, ro.ls_failed),
          "t gpu %.3fs oracle %.3fs" % (tg, to), flush=True)
    e.close()


def bench(modes, n_models, sweeps, ls=0):
    ranks = inputs.ranks_1_to_20(n_models)
    X = inputs.tensor(modes, 0)
    gm, _ = build_models(modes, ranks)
    R = sum(ranks)
    e = cc.Engine(modes, R)
    t = time.time()
    e.set_tensor(X)
    t_set = time.time() - t
    e.set_params(cc.default_params(max_iterations=10 ** 6, force_max_iter=1, line_search=ls))
    for m in gm:
        e.enqueue(m)
    e.admit()
    e.sweep(2)
    e.synchronize()
    e.set_profiling(True)
    t = time.time()
    e.sweep(sweeps)
    e.synchronize()
    dt = time.time() - t
    ks = e.kernel_stats()
    print("bench", modes, "models=%d R=%d ls=%d" % (n_models, R, ls),
          "set_tensor %.2fs" % t_set,
          "%.3f ms/sweep -> %.2f it/s" % (dt / sweeps * 1e3, sweeps / dt),
          "| mttkrp %.3f ms/launch %.2f TF/s | update %.3f ms/launch | other %.3f ms/launch (%d)" % (
              ks.mttkrp_ms / max(ks.mttkrp_launches, 1),
              ks.mttkrp_flops / max(ks.mttkrp_ms, 1e-9) * 1e-9,
              ks.update_ms / max(ks.update_launches, 1),
              ks.other_ms / max(ks.other_launches, 1), ks.other_launches), flush=True)
    e.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["mttkrp", "run", "bench"]
    if "mttkrp" in what:
        check_mttkrp([20, 20, 20], [2, 3, 4, 5])
        check_mttkrp([7, 5, 3], [1, 2, 3])
        check_mttkrp([13, 12, 11], list(range(1, 13)))
        check_mttkrp([100, 37, 41], inputs.ranks_1_to_20(20))
        check_mttkrp([299, 301, 41], inputs.ranks_1_to_20(10))
        check_mttkrp([330, 17, 9], [5, 20, 7])
        check_mttkrp([3, 3, 3, 3], [7, 2])
        check_mttkrp([6, 5, 4, 3], [3, 4, 5])
        check_mttkrp([40, 30, 20], inputs.ranks_1_to_20(40))   # R = 420: several column blocks
    if "run" in what:
        check_run([20, 20, 20], [2, 3, 4, 5], 1)
        check_run([20, 20, 20], [2, 3, 4, 5], 2)
        check_run([20, 20, 20], [2, 3, 4, 5], 10)
        check_run([20, 20, 20], [2, 3, 4, 5], 50)
        check_run([13, 12, 11], list(range(1, 13)) * 3, 30)
        check_run([6, 5, 4, 3], [3, 4, 5], 10)
        check_run([20, 9, 12], [5] * 8, 20, jk=[(0, i) for i in range(8)], tag="-jk")
        check_run([20, 20, 20], [2, 3, 4, 5, 20, 17], 25, ls=1, tag="-ls")
        check_run([13, 12, 11], [r for r in range(1, 13) for _ in range(5)], 200, buffer=30,
                  tol=1e-5, tag="-queue")
        check_run([13, 12, 11], [r for r in range(1, 13) for _ in range(5)], 200, buffer=30,
                  tol=1e-5, ls=1, interval=10, tag="-queue-ls")
    if "bench" in what:
        bench([100, 100, 100], 64, 20)
        bench([300, 300, 300], 256, 10)
        bench([300, 300, 300], 256, 10, ls=1)
