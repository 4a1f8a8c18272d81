"""Is the TTM team size (workgroups that split the s range of one column block; tree_geometry in
cals_hip_engine.cpp) the fastest one?  Sweep rate under CALS_TTM_TEAMS = forced values vs the default rule.
Usage: python tools/team_scan.py [sweeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402

SHAPES = [([100, 100, 100], 64, "f64"), ([200, 200, 200], 128, "f64"), ([64, 64, 64], 256, "f64"),
          ([150, 600, 90], 200, "f64"), ([299, 301, 41], 512, "f32"), ([300, 300, 300], 256, "f64")]


def rate(modes, n_models, dtype, X, base, teams, sweeps):
    # the engine reads CALS_TTM_TEAMS once per process: run every setting in a child process
    import subprocess
    env = dict(os.environ)
    if teams:
        env["CALS_TTM_TEAMS"] = str(teams)
    else:
        env.pop("CALS_TTM_TEAMS", None)
    code = ("import sys,time; sys.path.insert(0,'.'); import numpy as np; import cp_cals_amd as cc; from cp_cals_amd import inputs\n"
            "modes=%r; n=%d; ranks=[1+(k%%20) for k in range(n)]\n"
            "X=np.random.default_rng(0).uniform(-1,1,size=int(np.prod(modes)))\n"
            "base=inputs.model_factors(modes,ranks,seed=1)\n"
            "e=cc.Engine(modes,sum(ranks),device=0,dtype=%r); e.set_tensor(X)\n"
            "e.set_params(cc.default_params(max_iterations=10**9,force_max_iter=1))\n"
            "[e.enqueue(cc.Model([f.copy() for f in fs],lam.copy())) for fs,lam in base]\n"
            "e.admit(); e.sweep(4); e.synchronize(); t0=time.perf_counter(); e.sweep(%d); e.synchronize()\n"
            "print(%d/(time.perf_counter()-t0))\n" % (modes, n_models, dtype, sweeps, sweeps))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    return float(out.stdout.strip().splitlines()[-1])


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for modes, n_models, dtype in SHAPES:
        res = {}
        for t in (0, 4, 6, 8, 10, 12, 16, 20, 25, 33, 50):
            if t and t > max(modes):
                continue
            res[t] = rate(modes, n_models, dtype, None, None, t, sweeps)
        best = max((k for k in res if k), key=lambda k: res[k])
        print("%-14s %4d models %s | default %8.1f | " % ("x".join(map(str, modes)), n_models, dtype, res[0]) +
              "  ".join("T=%d %.1f" % (k, v) for k, v in res.items() if k) +
              " | best T=%d, default at %.1f %% of it" % (best, 100.0 * res[0] / res[best]), flush=True)


if __name__ == "__main__":
    main()
