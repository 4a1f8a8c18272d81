"""The TTM kernel alone, on sane operands: N launches of cals_hip_debug_mttkrp_path(mode, FIRST) on an engine whose models
were only admitted (the factors never change, T is written but never consumed), timed by the engine's hipEvent pairs.
For library variants whose TTM writes garbage or nothing to T (CALS_TTM_STRIP bit 1: no T stores) this gives the kernel's
time WITHOUT feeding that garbage back into the next launch -- bench.py cannot: an unstored T turns the factors into NaN and
FP64 MFMA time depends on the data.   Usage: [CALS_HIP_LIB=<variant>] python tools/ttm_alone.py c3|c2|c4 [launches]"""
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
import cp_cals_amd as cc  # noqa: E402
from cp_cals_amd import inputs  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    modes, k_models, _ = bench.WORKLOADS[wl]
    dtype = bench.WORKLOAD_DTYPE.get(wl, "f64")
    ranks = [1 + (k % 20) for k in range(k_models)]
    e = cc.Engine(modes, sum(ranks), device=0, dtype=dtype)
    e.set_tensor(inputs.tensor(modes, seed=0))
    for fs, lam in inputs.model_factors(modes, ranks, seed=1):
        e.enqueue(cc.Model(fs, lam))
    assert e.admit() == k_models
    firsts = {1: [0], 2: [1], 3: [0, 1, 2]}.get(e.tree, [])
    if not firsts:
        raise SystemExit("plan %d has no TTM" % e.tree)
    e.set_profiling(3)
    for mode in firsts:
        e.debug_mttkrp(mode, "first")  # warm-up
        e.reset_kernel_stats()
        for _ in range(n):
            e.debug_mttkrp(mode, "first")
        ks = e.kernel_stats()
        ms = ks.ttm_ms / ks.ttm_launches
        print("%s plan %d  TTM of pair %d alone: %.4f ms per launch over %d launches (%.2f TFLOP/s)" % (
            wl, e.tree, mode, ms, ks.ttm_launches, ks.ttm_flops / ks.ttm_launches / (ms * 1e-3) * 1e-12), flush=True)
    e.close()


if __name__ == "__main__":
    main()
