"""The copy-only work folded into the update launch (round 3) must not change a single bit:
  * UpdateArgs::pt -- the update bodies write the packed B-operand tiles Pt of the next TTM themselves instead of
    pack_pt_kernel (CALS_UPDATE_NO_PACK=1 restores the kernel);
  * UpdateArgs::partial -- the bodies sum the split-K partial tiles of their model's columns themselves instead of
    reduce_partials_kernel (CALS_UPDATE_FOLD_MAX_T = largest team width that takes the in-body sum; 0 = never).
Both are restatements of a copy / a fixed-order sum, so whole runs under either setting must agree bit for bit.
The switches are read once per process: every run is a child process (tests/run_sweeps_dump.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, shape, n_models, sweeps, dtype, plan, ls, **env):
    out = str(tmp_path / (tag + ".npz"))
    e = dict(os.environ)
    for k in ("CALS_UPDATE_NO_PACK", "CALS_UPDATE_FOLD_MAX_T"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_sweeps_dump.py"), out, shape, str(n_models),
                        str(sweeps), dtype, plan, str(ls)], env=e, capture_output=True, text=True, timeout=600)
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
