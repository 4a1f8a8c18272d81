"""The copy-only work folded into the update launch (round 3) must not change a single bit:
  * UpdateArgs::pt -- the update bodies write the packed B-operand tiles Pt of the next TTM themselves instead of
    pack_pt_kernel (CALS_UPDATE_NO_PACK=1 restores the kernel);
  * UpdateArgs::partial -- the bodies sum the split-K partial tiles of their model's columns themselves instead of
    reduce_partials_kernel (CALS_UPDATE_FOLD_MAX_T = largest team width that takes the in-body sum; 0 = never).
Both are restatements of a copy / a fixed-order sum, so whole runs under either setting must agree bit for bit.
Held to the same standard here: the rank > 64 factorisation on the side stream (CALS_HUGE_NO_SIDE=1 keeps it on the
main stream) and update::NNLS with several rows per wavefront (CALS_NNLS_ONE_ROW=1: one row per wave).
The switches are read once per process: every run is a child process (tests/run_sweeps_dump.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, shape, n_models, sweeps, dtype, plan, ls, extra=(), **env):
    out = str(tmp_path / (tag + ".npz"))
    e = dict(os.environ)
    for k in ("CALS_UPDATE_NO_PACK", "CALS_UPDATE_FOLD_MAX_T", "CALS_HUGE_NO_SIDE", "CALS_NNLS_ONE_ROW"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_sweeps_dump.py"), out, shape, str(n_models),
                        str(sweeps), dtype, plan, str(ls)] + [str(v) for v in extra], env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(out))


@pytest.mark.parametrize("shape,n_models,dtype,plan,ls", [
    ("40-30-20", 24, "f64", "M", 0),      # multi-sweep tree, every update packs or not by the schedule
    ("40-30-20", 24, "f64", "M", 1),      # line search between sweeps: Pt is dropped and packed by the kernel again
    ("64-48-40", 40, "f32", "M", 0),      # fp32 storage (rounded sums, fp32 tiles)
    ("29-31-11", 30, "f32", "B", 0),      # config 4's plan on a small shape
    ("40-30-20", 24, "f64", "A", 1),
    ("150-20-17", 45, "f64", "0", 0),     # no tree: three fused MTTKRPs, only the in-body reduction applies
])
def test_folded_copy_work_is_bit_identical(tmp_path, shape, n_models, dtype, plan, ls):
    base = _run(tmp_path, "base", shape, n_models, 5, dtype, plan, ls, CALS_UPDATE_NO_PACK=1, CALS_UPDATE_FOLD_MAX_T=0)
    variants = {
        "pack": dict(CALS_UPDATE_FOLD_MAX_T=0),
        "fold": dict(CALS_UPDATE_NO_PACK=1, CALS_UPDATE_FOLD_MAX_T=1000),
        "both": dict(CALS_UPDATE_FOLD_MAX_T=1000),
        "default": dict(),
    }
    assert int(base["iter"][0]) == 5
    for name, env in variants.items():
        got = _run(tmp_path, name, shape, n_models, 5, dtype, plan, ls, **env)
        assert sorted(got) == sorted(base)
        for k in base:
            assert np.array_equal(got[k], base[k]), "%s differs under %s" % (k, name)


def _same(a, b, what):
    assert sorted(a) == sorted(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), "%s differs: %s" % (k, what)


@pytest.mark.parametrize("plan,ls,dtype", [("M", 0, "f64"), ("M", 1, "f64"), ("0", 0, "f32")])
def test_side_stream_factorisation_is_bit_identical(tmp_path, plan, ls, dtype):
    """Ranks > 64: Hadamard + Cholesky of a mode on the side stream, next to that mode's MTTKRP (round 3), against the
    same launches on the main stream (CALS_HUGE_NO_SIDE=1): same kernels, same inputs -- any difference is a missing
    dependency between the streams."""
    ranks = "70,5,130,12,256,3"
    side = _run(tmp_path, "side", "44-36-28", 6, 4, dtype, plan, ls, extra=(0, ranks))
    main = _run(tmp_path, "main", "44-36-28", 6, 4, dtype, plan, ls, extra=(0, ranks), CALS_HUGE_NO_SIDE=1)
    assert int(side["iter"][0]) == 4
    _same(side, main, "side stream vs main stream")


@pytest.mark.parametrize("ranks", ["1,2,3,5,8,13,16,16,4,7", "17,20,24,9,30,32,18,25,2", "12,16,20,24,28,32,40,48,64,6"])
def test_nnls_rows_per_wavefront_is_bit_identical(tmp_path, ranks):
    """update::NNLS: several rows per wavefront, merged class launch, packed tiles (round 3) against one row per wave
    (CALS_NNLS_ONE_ROW=1): per row the same operations in the same order, so whole runs agree bit for bit."""
    n = len(ranks.split(","))
    multi = _run(tmp_path, "multi", "37-29-23", n, 5, "f64", "M", 0, extra=(1, ranks))
    one = _run(tmp_path, "one", "37-29-23", n, 5, "f64", "M", 0, extra=(1, ranks), CALS_NNLS_ONE_ROW=1)
    assert int(multi["iter"][0]) == 5
    _same(multi, one, "rows per wavefront")
