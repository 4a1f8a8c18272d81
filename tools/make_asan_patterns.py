"""Eviction schedules of two GPU test cases, from the CPU oracle, for the engine's host-side sanitizer harness.

tests/asan/test_engine_host_asan replays the EXACT call patterns of the two round-3 anomalies on the fake device
("Next round" 1c): the step API polled every step (17 x 14 x 12, 12 models, buffer 20, tol 1e-4) and the
queue life cycle 23 x 18 x 13 / buffer 26 / ERROR_CHECKING line search / plan M.  The fake device has no numerics;
what it needs from the real run is WHEN each model leaves: its iteration count at eviction, which this script takes
from the oracle (test infrastructure; run on the CPU) and writes to tests/asan/patterns.txt:

    pattern <name> <I> <J> <K> <buffer> <max_iter> <line_search> <ls_method> <ls_interval> <plan> <api>
    <n_models>
    <rank> <iters at eviction>      (one line per model, queue order)

Usage: python tools/make_asan_patterns.py   (rewrites tests/asan/patterns.txt)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402
from helpers import make_models  # noqa: E402


def life_case(seed_wanted):
    sys.argv = sys.argv[:1]
    import importlib
    mod = importlib.import_module("test_gpu_random_shapes")
    for case in mod._life_cases(24, 777):
        if case[-1] == seed_wanted:
            return case
    raise SystemExit("life case not found")


def run(name, modes, ranks, buffer, X, base, api, plan, **kw):
    om = [oracle.Model(fs, lam, jk=j) for fs, lam, j in base]
    oracle.cp_cals(X, modes, om, oracle.default_params(mttkrp_method=oracle.MTTKRP, buffer_size=buffer, **kw))
    head = "pattern %s %d %d %d %d %d %d %d %d %s %s" % (
        name, modes[0], modes[1], modes[2], buffer, kw.get("max_iterations", 200), kw.get("line_search", 0),
        kw.get("line_search_method", 0), kw.get("line_search_interval", 5), plan, api)
    return [head, str(len(ranks))] + ["%d %d" % (r, m.iters) for r, m in zip(ranks, om)]


def main():
    out = []
    # tests/test_gpu_async_eviction.py::test_step_api_results_complete_when_reported
    modes, ranks = [17, 14, 12], [3, 5, 2, 7, 4, 6, 1, 8, 3, 5, 2, 4]
    X = inputs.low_rank_tensor(modes, 4, seed=5)[0] + 0.05 * inputs.tensor(modes, 6)
    base = make_models(inputs, modes, ranks, seed=11)
    out += run("step_api_polled", modes, ranks, 20, X, base, "step", "auto", max_iterations=40, tol=1e-4)
    # tests/test_gpu_random_shapes.py::test_random_queue_life_cycle[...-569869817]
    seed = 569869817
    modes, ranks, buffer, plan, ls, tol, _ = life_case(seed)
    X = inputs.low_rank_tensor(modes, 5, seed=seed % 1000)[0] + 0.05 * inputs.tensor(modes, seed % 977)
    base = make_models(inputs, modes, ranks, seed=1 + seed % 991, jk=None)
    assert seed % 3 != 0 and ls == 1 and (seed >> 3) & 1 == 1 and plan == "M"
    out += run("life_cycle_ec_plan_m", modes, ranks, buffer, X, base, "run", plan, max_iterations=30, tol=tol,
               line_search=ls, line_search_interval=3, line_search_method=1)
    path = os.path.join(ROOT, "tests", "asan", "patterns.txt")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(open(path).read())


if __name__ == "__main__":
    main()
