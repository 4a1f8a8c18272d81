"""CALS_HIP_VERIFY=1: the engine recomputes and compares every operand it keeps across launches (the packed TTM
operand Pt an update launch leaves behind, a T that waits for its second mode -- also over the sweep boundary --, the
other modes' Gramians, the free columns inside the active width) at the launch that would consume it; a difference is
an engine error.  Here: queue life cycles that exercise every one of those hand-overs -- plan M
with both line searches, evictions, compress, admissions -- must run clean with the checks on AND still match the
oracle.  (The whole GPU suite is run once per round with the switch on; this file keeps the mode itself under test.)"""
import os

import numpy as np
import pytest

from helpers import make_models, reconstruct

pytestmark = pytest.mark.gpu


@pytest.fixture()
def verify_on():
    old = {k: os.environ.get(k) for k in ("CALS_HIP_VERIFY", "CALS_HIP_TREE")}
    os.environ["CALS_HIP_VERIFY"] = "1"
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


CASES = [
    # modes, ranks, buffer, plan, line search kwargs, update method
    ([23, 18, 13], [1, 5, 4, 8, 3, 2, 3, 3, 3, 5, 4, 5, 4, 6, 1, 4, 8, 1, 5, 2], 26, "M",
     dict(line_search=1, line_search_interval=3, line_search_method=1), 0),   # the round-3 case: EC line search
    ([23, 18, 13], [1, 5, 4, 8, 3, 2, 3, 3, 3, 5, 4, 5, 4, 6, 1, 4, 8, 1, 5, 2], 26, "M",
     dict(line_search=1, line_search_interval=3, line_search_method=0), 0),   # NEC: stale columns patched
    ([17, 14, 12], [3, 5, 2, 7, 4, 6, 1, 8, 3, 5, 2, 4], 20, "M", dict(), 0),
    ([19, 21, 9], [4, 2, 6, 3, 5, 1, 7, 2], 14, "B", dict(), 1),              # NNLS, non-cubic: Pt layouts differ
    ([16, 16, 16], [20, 3, 12, 7, 16, 9], 40, "A", dict(line_search=1, line_search_interval=2), 0),
]


@pytest.mark.parametrize("modes,ranks,buffer,plan,ls_kw,update", CASES)
def test_queue_life_cycle_clean_under_verify(cc, oracle, inputs, verify_on, modes, ranks, buffer, plan, ls_kw, update):
    os.environ["CALS_HIP_TREE"] = plan
    X = inputs.low_rank_tensor(modes, 5, seed=817)[0] + 0.05 * inputs.tensor(modes, 253)
    if update:
        X = np.abs(X)
    base = make_models(inputs, modes, ranks, seed=97)
    kw = dict(max_iterations=30, tol=1e-5, update_method=update, **ls_kw)
    e = cc.Engine(modes, buffer)
    e.set_tensor(X)
    e.set_params(cc.default_params(**kw))
    gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    for m in gm:
        e.enqueue(m)
    rep = e.run()   # a failed check raises CalsHipError("CALS_HIP_VERIFY: ... differs from its recomputation ...")
    e.close()
    om = [oracle.Model(fs, lam) for fs, lam, _ in base]
    ro = oracle.cp_cals(X, modes, om, oracle.default_params(mttkrp_method=oracle.MTTKRP, buffer_size=buffer, **kw))
    assert (rep.iter, rep.n_ktensors) == (ro.iter, ro.n_ktensors)
    xn = np.linalg.norm(X)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-8 * max(1.0, xn)
