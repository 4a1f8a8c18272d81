"""update::NNLS on the device (nnls_kernel.hip) against the oracle's restatement of
update_factor_non_negative_constrained (src/utils/update.cpp:61-176; the oracle's row solver is itself
pinned against SciPy's Lawson-Hanson nnls in test_oracle_reference_suite.py).

Bar: fp64, relative Frobenius <= TOL_NNLS on factors and lambda after the same number of sweeps.  The
NNLS minimiser of a row is unique, so the device (whose back substitution sums in a different order)
and the oracle agree to rounding times the conditioning of the passive sub-matrix; exact zeros
(active constraints) must coincide up to entries that the algorithm's own tolerance treats as zero.
"""
import numpy as np
import pytest

from helpers import make_models, rel, reconstruct
from test_gpu_parity import _assert_models_match, _run_both

pytestmark = pytest.mark.gpu
TOL_NNLS = 1e-8
TIE = 1e-9  # relative distance of two errors below which an accept / revert test is a tie (rounding of a cancelled sum)
NNLS = 1


def _nonneg_tensor(inputs, modes, rank, seed, noise=0.05):
    X, _, _ = inputs.low_rank_tensor(modes, rank, seed=seed)
    return np.abs(X) + noise * inputs.tensor(modes, seed + 1)


def _check(gm, om, rep, ro, tol=TOL_NNLS, nonneg=True):
    assert rep.nnls_status == 0 and ro.nnls_status == 0
    for m in gm:
        for f in m.factors:
            assert (f >= 0.0).all() or not nonneg
    _assert_models_match(gm, om, ro.X_norm ** 2, tol=tol)
    # the patterns of active constraints: an entry that is exactly zero on one side is at most
    # rounding-sized on the other
    for a, b in zip(gm, om):
        for fa, fb in zip(a.factors, b.factors):
            scale = max(np.abs(fb).max(), 1e-300)
            assert np.abs(fa[fb == 0.0]).max(initial=0.0) <= 1e-9 * scale
            assert np.abs(fb[fa == 0.0]).max(initial=0.0) <= 1e-9 * scale


@pytest.mark.parametrize("modes,ranks,iters", [
    ([20, 20, 20], [2, 3, 4, 5], 30),
    ([18, 17, 16], [5] * 6, 40),                     # tests/als/test_als.cpp:62-103's shape
    ([13, 12, 11], list(range(1, 13)) * 2, 20),
    ([50, 40, 30], None, 10),
    ([6, 5, 4, 3], [3, 4, 5], 10),                   # 4-way
    ([40, 36, 33], [33, 64, 48, 5, 20], 8),          # ranks above 32 (three waves per workgroup)
    ([330, 17, 9], [5, 20, 7], 8),                   # a tall mode: several row chunks per model
])
def test_nnls_forced_iterations_vs_oracle(cc, oracle, inputs, modes, ranks, iters):
    if ranks is None:
        ranks = inputs.ranks_1_to_20(40)
    X = _nonneg_tensor(inputs, modes, 4, seed=31)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, iters, update_method=NNLS)
    assert rep.iter == ro.iter == iters
    _check(gm, om, rep, ro)


@pytest.mark.parametrize("shift", [0.1, 0.5])
def test_nnls_signed_tensor_many_active_constraints(cc, oracle, inputs, shift):
    """A tensor with entries of both signs: most rows end with several constraints active; the warm start
    (previous sweep's passive set) and both exchange loops are exercised.  With shift 0.5 the tensor has
    zero mean and whole columns collapse to zero: lambda = 0 (column left unscaled), a singular H whose
    Cholesky fails inside the warm start (the caught CholFail, update.cpp:117-120)."""
    modes, ranks = [24, 21, 19], [3, 8, 12, 20, 1, 16]
    X = inputs.tensor(modes, 5) - shift
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 15, update_method=NNLS)
    _check(gm, om, rep, ro)
    zeros = sum(int((f == 0.0).sum()) for m in gm for f in m.factors)
    assert zeros > 50


def test_nnls_queue_eviction_compress_vs_oracle(cc, oracle, inputs):
    """Buffer smaller than the queue, tolerance-driven eviction and compress: the active sets move with
    their models' columns."""
    modes = [30, 25, 20]
    ranks = inputs.ranks_1_to_20(30)
    X = _nonneg_tensor(inputs, modes, 6, seed=9)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 40, buffer=64, force_max_iter=0,
                                tol=1e-4, update_method=NNLS)
    assert rep.iter == ro.iter
    assert [m.iters for m in gm] == [m.iters for m in om]
    _check(gm, om, rep, ro)


def test_nnls_jackknife_models_vs_oracle(cc, oracle, inputs):
    modes, comp = [20, 9, 12], 5
    X = _nonneg_tensor(inputs, modes, comp, seed=21)
    jk = [(0, i) for i in range(20)]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, [comp] * 20, X, 15, jk=jk, update_method=NNLS)
    _check(gm, om, rep, ro)
    for i, m in enumerate(gm):
        assert np.all(m.factors[0][i, :] == 0.0)


@pytest.mark.parametrize("method", [0, 1])
def test_nnls_with_line_search_vs_oracle(cc, oracle, inputs, method):
    """Ktensor::copy carries the active sets (src/ktensor.cpp:174): the line search's backup / revert
    restores them with the factors."""
    modes, ranks = [20, 20, 20], [2, 3, 4, 5, 20, 17]
    X = _nonneg_tensor(inputs, modes, 5, seed=3, noise=0.3)
    # Sweep counts chosen before the first tie: from the 17th sweep (10th with error checking) the
    # low-rank models have converged, their extrapolation is a null step, and the accept / revert test
    # compares two errors that are equal up to rounding (either outcome leaves the same factors, but the
    # ls_failed counters and, with error checking, the reference's 0/0 normalisation of an all-zero
    # column of a kept candidate then differ).
    iters = 16 if method == 0 else 9
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, iters, line_search=1,
                                line_search_interval=5, line_search_method=method, update_method=NNLS)
    assert rep.ls_performed == ro.ls_performed and rep.ls_performed > 0
    # one accept / revert decision may fall on the other side: a converged model's extrapolation is a null step and
    # its test compares two errors that are equal up to rounding (the device's reciprocal-multiply / rsq pivots move
    # them by an ulp); either outcome leaves the same factors, which _check verifies
    assert abs(rep.ls_failed - ro.ls_failed) <= 1
    # ... and when one did, it WAS such a tie: some model went through a test whose two errors agree to rounding
    # (cals_hip_debug_ls_margin: the smallest relative distance over the model's tests)
    if rep.ls_failed != ro.ls_failed:
        assert min(m.ls_margin for m in gm) <= TIE, [m.ls_margin for m in gm]
    # the last sweep may end on an extrapolated (unconstrained) state, as in the reference
    _check(gm, om, rep, ro, nonneg=False)


def test_nnls_tree_plans_agree(cc, oracle, inputs):
    """The update method is independent of how the MTTKRP is computed: every dimension-tree plan gives the
    oracle's result."""
    import os
    modes, ranks = [40, 36, 33], [5, 20, 7, 12]
    X = _nonneg_tensor(inputs, modes, 6, seed=12)
    old = os.environ.get("CALS_HIP_TREE")
    try:
        for plan in ("0", "A", "B", "M"):
            os.environ["CALS_HIP_TREE"] = plan
            gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 9, update_method=NNLS)
            _check(gm, om, rep, ro)
    finally:
        if old is None:
            os.environ.pop("CALS_HIP_TREE", None)
        else:
            os.environ["CALS_HIP_TREE"] = old


def test_nnls_f32_storage(cc, oracle, inputs):
    """fp32 storage engine: the NNLS arithmetic is fp64 on fp32 inputs; stated fp32 tolerance as in
    test_gpu_fp32.py."""
    from test_gpu_fp32 import TOL32_FIT, engine32
    modes, ranks = [30, 25, 20], [2, 3, 4, 5, 20, 17]
    X = _nonneg_tensor(inputs, modes, 5, seed=4)
    kw = dict(max_iterations=8, force_max_iter=1, update_method=NNLS)
    e, gm, base = engine32(cc, inputs, modes, ranks, X, params=cc.default_params(**kw))
    rep = e.run()
    e.close()
    om = [oracle.Model(fs, lam) for fs, lam, _ in base]
    ro = oracle.cp_cals(X, modes, om, oracle.default_params(mttkrp_method=oracle.MTTKRP,
                                                           buffer_size=sum(ranks), **kw))
    assert rep.nnls_status == 0
    for a, b in zip(gm, om):
        for f in a.factors:
            assert (f >= 0.0).all()
        assert abs(a.fit - b.fit) < 5e-4
        assert TOL32_FIT > 0


def test_nnls_golden(cc, inputs):
    """The committed vectors of tests/golden/nnls_18x17x16.npz (oracle output, make_golden.py) at 1, 2 and 20
    forced sweeps."""
    import importlib.util
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    case = mg.CASES["nnls_18x17x16"]
    gold = np.load(os.path.join(gdir, "nnls_18x17x16.npz"))
    modes, ranks = case["modes"], case["ranks"]
    X = mg.tensor_of(case)
    for it in case["iters"]:
        base = make_models(inputs, modes, ranks, seed=case["mseed"])
        e = cc.Engine(modes, sum(ranks))
        e.set_tensor(X)
        e.set_params(cc.default_params(max_iterations=it, force_max_iter=1, update_method=NNLS))
        gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
        for m in gm:
            e.enqueue(m)
        rep = e.run()
        e.close()
        assert rep.iter == gold["it%d_sweeps" % it][0] and rep.nnls_status == 0
        for n in range(3):
            assert rel(np.hstack([m.factors[n] for m in gm]), gold["it%d_factor%d" % (it, n)]) < TOL_NNLS
        assert rel(np.concatenate([m.lam for m in gm]), gold["it%d_lambda" % it]) < TOL_NNLS
        assert np.allclose([m.error for m in gm], gold["it%d_error" % it], rtol=1e-9, atol=1e-12)


def test_nnls_slow_error_equals_fast_error(cc, inputs):
    """ComputeCorrectResultConstrained3D's checks on the device result: non-negative factors, finite
    reconstruction error, and the error formula's value equals the reconstructed one."""
    modes, ranks = [18, 17, 16], [5, 5, 5, 7]
    X = _nonneg_tensor(inputs, modes, 5, seed=111)
    base = make_models(inputs, modes, ranks, seed=300)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=100, update_method=NNLS))
    gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    for m in gm:
        e.enqueue(m)
    rep = e.run()
    e.close()
    assert rep.nnls_status == 0
    for m in gm:
        slow = np.linalg.norm(X.ravel(order="F") - reconstruct(m.factors, m.lam, modes))
        assert np.isfinite(slow) and slow < 50
        assert abs(slow - m.error) <= 1e-8 * max(1.0, slow)
        for f in m.factors:
            assert (f >= 0.0).all()
