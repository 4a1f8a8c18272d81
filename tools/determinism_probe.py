"""Is a queue life cycle bitwise reproducible?  Re-runs the first CASES cases of tests/test_gpu_random_shapes.py's
life-cycle generator REPS times each (fresh engine every time; every third repetition an unrelated engine with a
rank-100 model runs in between, so that device memory is recycled differently) and compares every model's factors,
lambdas and iteration counts bit for bit with the first repetition.  Usage: python tools/determinism_probe.py [REPS [CASES]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402
from helpers import make_models  # noqa: E402


def life_cases(n, seed):  # tests/test_gpu_random_shapes.py::_life_cases
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        modes = [int(v) for v in rng.integers(8, 25, size=3)]
        n_models = int(rng.integers(8, 40))
        ranks = [int(v) for v in rng.integers(1, 9, size=n_models)]
        buffer = int(rng.integers(max(ranks), max(max(ranks) + 1, sum(ranks) // 2)))
        out.append((modes, ranks, buffer, ["0", "A", "B", "M"][int(rng.integers(0, 4))],
                    int(rng.integers(0, 2)), float(10.0 ** -rng.integers(3, 6)), int(rng.integers(0, 1 << 30))))
    return out


def run_case(modes, ranks, buffer, plan, ls, tol, seed):
    os.environ["CALS_HIP_TREE"] = plan
    X = inputs.low_rank_tensor(modes, 5, seed=seed % 1000)[0] + 0.05 * inputs.tensor(modes, seed % 977)
    jk = [((0, k % modes[0]) if (k + seed) % 2 == 0 else None) for k in range(len(ranks))] if seed % 3 == 0 else None
    base = make_models(inputs, modes, ranks, seed=1 + seed % 991, jk=jk)
    kw = dict(max_iterations=30, tol=tol, line_search=ls, line_search_interval=3, line_search_method=(seed >> 3) & 1)
    if not ls and (seed >> 4) & 1:
        kw["update_method"] = 1
        X = np.abs(X)
    e = cc.Engine(modes, buffer)
    e.set_tensor(X)
    e.set_params(cc.default_params(**kw))
    gm = [cc.Model([f.copy() for f in fs], lam.copy(), jk=j) for fs, lam, j in base]
    for m in gm:
        e.enqueue(m)
    e.run()
    e.close()
    return [(m.iters, m.lam.tobytes(), [f.tobytes() for f in m.factors]) for m in gm]


def disturb(k):
    modes = [40, 30, 20]
    r = [100, 3, 7][k % 3]
    e = cc.Engine(modes, r + 8)
    e.set_tensor(inputs.tensor(modes, k))
    e.set_params(cc.default_params(max_iterations=3, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, [r, 8], 1 + k):
        e.enqueue(cc.Model(fs, lam))
    e.run()
    e.close()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    bad = 0
    for ci, case in enumerate(life_cases(n_cases, 777)):
        ref = run_case(*case)
        diffs = 0
        for k in range(reps):
            if k % 3 == 2:
                disturb(k)
            out = run_case(*case)
            if out != ref:
                diffs += 1
                which = [i for i, (a, b) in enumerate(zip(out, ref)) if a != b]
                print("case %d rep %d: %d models differ (first %s, iters %s vs %s)" % (
                    ci, k, len(which), which[:4], [out[i][0] for i in which[:4]], [ref[i][0] for i in which[:4]]), flush=True)
        print("case %2d modes %s plan %s ls %d method %d models %d buffer %d: %d of %d repetitions differ" % (
            ci, case[0], case[3], case[4], (case[6] >> 3) & 1, len(case[1]), case[2], diffs, reps), flush=True)
        bad += diffs
    print("TOTAL differing repetitions:", bad)


if __name__ == "__main__":
    main()
