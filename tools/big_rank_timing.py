"""How slow are the rank > 64 bodies (built to be right, not fast)?  Sweep time of C3's tensor shape with one
model of rank r next to 255 models of ranks 1..20, against the 256 small models alone; both update methods.
Usage: python tools/big_rank_timing.py [sweeps [ranks...]]   (BIG_COUNT=n in the environment: n models of that rank instead of one)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402


def run(modes, ranks, X, update_method, sweeps):
    base = inputs.model_factors(modes, ranks, seed=1)
    e = cc.Engine(modes, sum(ranks), device=0)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, update_method=update_method))
    for fs, lam in base:
        e.enqueue(cc.Model([np.abs(f) for f in fs] if update_method else [f.copy() for f in fs], lam.copy()))
    assert e.admit() == len(ranks)
    e.sweep(2)
    e.synchronize()
    t0 = time.perf_counter()
    e.sweep(sweeps)
    e.synchronize()
    dt = (time.perf_counter() - t0) / sweeps
    e.close()
    return dt * 1e3


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    modes = [300, 300, 300]
    X = np.abs(inputs.tensor(modes, seed=0))
    small = [1 + (k % 20) for k in range(255)]
    for um, name in ((0, "unconstrained"), (1, "NNLS")):
        t0 = run(modes, small + [20], X, um, sweeps)
        print("%-13s 256 models of ranks 1..20: %.2f ms per sweep" % (name, t0), flush=True)
        for r in [int(v) for v in sys.argv[2:]] or (65, 100, 128, 256):
            nb = int(os.environ.get("BIG_COUNT", "1"))
            t = run(modes, small + [r] * nb, X, um, sweeps)
            print("%-13s 255 small + %d rank-%-3d model(s): %.2f ms per sweep (+%.2f ms)" % (name, nb, r, t, t - t0), flush=True)


if __name__ == "__main__":
    main()
