"""Probe: does splitting a GPU's models over E engines (each with its own HIP stream, driven by its own host
thread) raise the sweep rate?  The latency-bound configs (C2) leave most CUs idle inside every small kernel;
independent model sets can fill them.  Prints ALS it/s (all models advance one sweep per "iteration") for
E = 1, 2, 3, 4 on the workload given.  Usage: python tools/dual_engine_probe.py c2 [sweeps] [stagger_ms ...]
stagger_ms (round 4): with E = 2, engine 1 starts its sweeps that many milliseconds after engine 0 -- a deliberate phase
offset between the two kernel sequences (one engine's TTM against the other's contraction + update), for every value given."""
import sys
import threading
import time

sys.path.insert(0, ".")
import bench  # noqa: E402
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402


def build(modes, ranks, X, base, ls, dtype):
    e = cc.Engine(modes, sum(ranks), device=0, dtype=dtype)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, line_search=ls,
                                   line_search_interval=5, line_search_step=0.0))
    ms = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam in base]
    for m in ms:
        e.enqueue(m)
    assert e.admit() == len(ranks)
    return e, ms


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    modes, k_models, ls = bench.WORKLOADS[wl]
    dtype = bench.WORKLOAD_DTYPE.get(wl, "f64")
    ranks = [1 + (k % 20) for k in range(k_models)]
    X = inputs.tensor(modes, seed=0)
    base = inputs.model_factors(modes, ranks, seed=1)
    staggers = [float(v) for v in sys.argv[3:]]
    for n_eng in (1, 2, 3, 4):
        parts = [list(range(i, k_models, n_eng)) for i in range(n_eng)]
        engs = [build(modes, [ranks[k] for k in p], X, [base[k] for k in p], ls, dtype) for p in parts]
        for e, _ in engs:
            e.sweep(5)
            e.synchronize()

        def run(e):
            e.sweep(sweeps)
            e.synchronize()

        best = 1e30
        for _ in range(3):
            th = [threading.Thread(target=run, args=(e,)) for e, _ in engs]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            best = min(best, time.perf_counter() - t0)
        print("%s engines=%d  plans=%s  %.1f it/s  (%.4f ms per sweep of all models)" % (
            wl, n_eng, ",".join(str(e.tree) for e, _ in engs), sweeps / best, best / sweeps * 1e3), flush=True)
        for off in (staggers if n_eng == 2 else []):
            def run_late(e, delay):
                time.sleep(delay)
                e.sweep(sweeps)
                e.synchronize()
            bests = 1e30
            for _ in range(3):
                th = [threading.Thread(target=run_late, args=(e, k * off * 1e-3)) for k, (e, _) in enumerate(engs)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                bests = min(bests, time.perf_counter() - t0 - off * 1e-3)
            print("%s engines=2  engine 1 starts %.2f ms late: %.1f it/s" % (wl, off, sweeps / bests), flush=True)
        for e, _ in engs:
            e.close()


if __name__ == "__main__":
    main()
