"""Does the engine's cost model (cals_hip_create: plan 0 / A / B / M from the problem size) pick the fastest MTTKRP
plan?  For a set of shapes: sweep rate under every forced plan (CALS_HIP_TREE) and under the default choice.
The device analogue of the reference's lookup tables (src/utils/mttkrp.cpp:19-52, 562-614), checked rather than
tabulated.  Usage: python tools/plan_scan.py [sweeps]"""
import os
import sys
import time

sys.path.insert(0, ".")
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402

SHAPES = [
    ([100, 100, 100], 64, "f64"), ([300, 300, 300], 256, "f64"), ([299, 301, 41], 512, "f32"),
    ([299, 301, 41], 512, "f64"), ([41, 299, 301], 512, "f64"), ([301, 41, 299], 512, "f64"),
    ([500, 60, 50], 256, "f64"), ([50, 60, 500], 256, "f64"), ([1000, 100, 12], 256, "f64"),
    ([64, 64, 64], 256, "f64"), ([200, 200, 200], 128, "f64"), ([405, 136, 19], 512, "f64"),
    ([150, 600, 90], 200, "f64"), ([600, 500, 400], 64, "f64"),
]


def rate(modes, n_models, dtype, plan, sweeps):
    if plan is None:
        os.environ.pop("CALS_HIP_TREE", None)
    else:
        os.environ["CALS_HIP_TREE"] = plan
    ranks = [1 + (k % 20) for k in range(n_models)]
    X = rate.cache.get(tuple(modes))
    if X is None:
        import numpy as np
        X = np.random.default_rng(0).uniform(-1, 1, size=int(np.prod(modes)))
        rate.cache = {tuple(modes): X}
    base = rate.models.get((tuple(modes), n_models))
    if base is None:
        base = inputs.model_factors(modes, ranks, seed=1)
        rate.models = {(tuple(modes), n_models): base}
    e = cc.Engine(modes, sum(ranks), device=0, dtype=dtype)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1))
    for fs, lam in base:
        e.enqueue(cc.Model([f.copy() for f in fs], lam.copy()))
    e.admit()
    e.sweep(4)
    e.synchronize()
    t0 = time.perf_counter()
    e.sweep(sweeps)
    e.synchronize()
    dt = time.perf_counter() - t0
    kind = e.tree
    e.close()
    return sweeps / dt, kind


rate.cache = {}
rate.models = {}


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    names = {0: "0", 1: "A", 2: "B", 3: "M"}
    for modes, n_models, dtype in SHAPES:
        res = {}
        for plan in ("0", "A", "B", "M"):
            res[plan], _ = rate(modes, n_models, dtype, plan, sweeps)
        dflt, kind = rate(modes, n_models, dtype, None, sweeps)
        best = max(res, key=res.get)
        print("%-16s %4d models %s | " % ("x".join(map(str, modes)), n_models, dtype) +
              "  ".join("%s %8.1f" % (p, res[p]) for p in ("0", "A", "B", "M")) +
              " | default = %s %8.1f it/s  (best %s, default at %.1f %% of it)" % (
                  names[kind], dflt, best, 100.0 * dflt / res[best]), flush=True)


if __name__ == "__main__":
    main()
