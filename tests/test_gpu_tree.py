"""Dimension-tree MTTKRP plans (ttm_kernel.hip): modes (0,1) share T = X x_2 C ("A") or modes (1,2)
share T = X x_0 A ("B").  Same oracle, same tolerances as the plain plan (test_gpu_parity.py): the
plan only re-associates the sums, exactly like the reference's own two-step variants
(src/utils/mttkrp.cpp:330-560), which the oracle also restates (TWOSTEP0/1)."""
import os

import numpy as np
import pytest

from helpers import make_models, reconstruct, rel
from test_gpu_parity import TOL_KERNEL, TOL_RUN, _assert_models_match, _run_both, engine_with

pytestmark = pytest.mark.gpu
PLANS = {"0": 0, "A": 1, "B": 2, "M": 3}


@pytest.fixture(params=["M", "A", "B", "0"])
def plan(request):
    old = os.environ.get("CALS_HIP_TREE")
    os.environ["CALS_HIP_TREE"] = request.param
    yield request.param
    if old is None:
        del os.environ["CALS_HIP_TREE"]
    else:
        os.environ["CALS_HIP_TREE"] = old


@pytest.mark.parametrize("modes,ranks", [
    ([20, 20, 20], [2, 3, 4, 5]),            # BASELINE config 1
    ([7, 5, 3], [1, 2, 3]),                  # tiny, ragged
    ([13, 12, 11], list(range(1, 13))),
    ([100, 37, 41], None),
    ([299, 301, 41], None),                  # BASELINE config 4 shape
    ([330, 17, 9], [5, 20, 7]),              # 21 m-tiles: three M blocks of 7 (fp64)
    ([17, 330, 9], [5, 20, 7]),
    ([40, 30, 20], [20] * 21),               # R = 420: four column blocks
    ([161, 23, 50], [32, 1, 20]),            # 11 m-tiles: M blocks of 6 and 5 tiles (k_big = 1)
    ([600, 7, 5], [3, 20]),                  # 38 m-tiles: 4 M blocks; contraction rows > 512 (generic kernel)
    ([6, 1100, 3], [4, 9]),                  # the same for the middle mode (fp32: > 1024 rows)
])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_tree_mttkrp_every_mode_vs_oracle(cc, oracle, inputs, plan, modes, ranks, dtype):
    if ranks is None:
        ranks = inputs.ranks_1_to_20(20)
    X = inputs.tensor(modes, 0)
    base = make_models(inputs, modes, ranks)
    e = cc.Engine(modes, sum(ranks), dtype=dtype)
    assert e.tree == PLANS[plan]
    e.set_tensor(X)
    for fs, lam, _ in base:
        e.enqueue(cc.Model([f.copy() for f in fs], lam.copy()))
    e.admit()
    facs = [np.asfortranarray(np.hstack([fs[n] for fs, _, _ in base])) for n in range(3)]
    tol = TOL_KERNEL if dtype == "f64" else 2e-5
    pairs = {"0": [], "A": [0], "B": [1], "M": [0, 1, 2]}[plan]
    for n in range(3):
        want = oracle.mttkrp(X, modes, facs, n, oracle.MTTKRP)
        assert rel(e.debug_mttkrp(n), want) < tol
        assert rel(e.debug_mttkrp(n, "plain"), want) < tol
        if n in pairs:
            assert rel(e.debug_mttkrp(n, "first"), want) < tol          # TTM kernel, fused G
        else:
            with pytest.raises(cc.CalsHipError):
                e.debug_mttkrp(n, "first")
        if (n + 2) % 3 in pairs:
            assert rel(e.debug_mttkrp(n, "second"), want) < tol         # T of the pair + contraction
    e.close()


def test_auto_plan_prefers_a_tree_for_the_baseline_shapes(cc):
    old = os.environ.pop("CALS_HIP_TREE", None)
    try:
        e = cc.Engine([300, 300, 300], 2656)
        assert e.tree == 3        # cube: every pair costs the same, 3 TTMs per 2 sweeps
        e.close()
        e = cc.Engine([299, 301, 41], 5328, dtype="f32")
        assert e.tree == 2        # contract over the 299-mode; T is J*K*R = 263 MB
        e.close()
        e = cc.Engine([6, 5, 4, 3], 12)
        assert e.tree == 4        # N > 3: the two-group tree (tests/test_gpu_nway.py)
        e.close()
    finally:
        if old is not None:
            os.environ["CALS_HIP_TREE"] = old


@pytest.mark.parametrize("modes,ranks,iters", [
    ([20, 20, 20], [2, 3, 4, 5], 50),
    ([13, 12, 11], list(range(1, 13)) * 3, 30),
    ([50, 40, 30], None, 10),
])
def test_tree_forced_iterations_vs_oracle(cc, oracle, inputs, plan, modes, ranks, iters):
    if ranks is None:
        ranks = inputs.ranks_1_to_20(40)
    X = inputs.tensor(modes, 3)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, iters)
    assert rep.iter == ro.iter == iters
    _assert_models_match(gm, om, ro.X_norm ** 2)


def test_tree_line_search_vs_oracle(cc, oracle, inputs, plan):
    modes, ranks = [20, 20, 20], [2, 3, 4, 5, 20, 17]
    X = inputs.tensor(modes, 3)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 25, line_search=1,
                                line_search_interval=5)
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > 0
    _assert_models_match(gm, om, ro.X_norm ** 2)


@pytest.mark.parametrize("patch_max", ["1.0", "0.5", "0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_plan_m_keeps_a_pending_T_across_line_search(cc, oracle, inputs, patch_max, dtype):
    """T[:, :, c] depends on column c of one factor only.  When a line-search step rewrites some models
    while a T is pending across the sweep boundary, the engine keeps T and recomputes just those columns
    (gather -> fused MTTKRP on the packed columns -> scatter).  CALS_TREE_PATCH_MAX: 1.0 = always patch
    (also when every model extrapolated), 0 = always drop T (round-1 behaviour), 0.5 = the default rule.
    All three must give the oracle's models; with patching on, the fused MTTKRP kernel must actually have
    run (under plan M with no queue it runs for nothing else)."""
    modes = [36, 28, 24]
    ranks = inputs.ranks_1_to_20(23) + [7, 3]      # R = 233: two column blocks, the second partly filled
    X = inputs.low_rank_tensor(modes, 5, seed=13)[0] + 0.3 * inputs.tensor(modes, 9)
    iters = 23
    old = {k: os.environ.get(k) for k in ("CALS_HIP_TREE", "CALS_TREE_PATCH_MAX", "CALS_LS_SCHEDULE_OFF")}
    os.environ["CALS_HIP_TREE"] = "M"
    os.environ["CALS_TREE_PATCH_MAX"] = patch_max
    # (the launch counts below are about patching: the line-search-aware pair schedule, which replaces a TTM by a fused
    # MTTKRP in a sweep predicted to end with a mass extrapolation, has its own test right below)
    os.environ["CALS_LS_SCHEDULE_OFF"] = "1"
    try:
        base = make_models(inputs, modes, ranks, seed=2)
        e = cc.Engine(modes, sum(ranks), dtype=dtype)
        assert e.tree == 3
        e.set_tensor(X)
        e.set_params(cc.default_params(max_iterations=iters, force_max_iter=1, line_search=1, line_search_interval=3))
        gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
        for m in gm:
            e.enqueue(m)
        e.set_profiling(1)
        # stepwise (cals_hip_sweep: the 4-byte read-back path) for the first half, cals_hip_run after
        assert e.admit() == len(ranks)
        e.sweep(11)
        ks_mid = e.kernel_stats()
        rep = e.run()
        ks = e.kernel_stats()
        e.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    om = [oracle.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    ro = oracle.cp_cals(X, modes, om, oracle.default_params(
        max_iterations=iters, force_max_iter=1, buffer_size=sum(ranks), mttkrp_method=oracle.MTTKRP,
        line_search=1, line_search_interval=3))
    assert ro.ls_failed > 0 and ro.ls_performed > ro.ls_failed
    tol = 1e-8 if dtype == "f64" else 2e-3
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < tol
    if patch_max == "0":
        assert ks.mttkrp_launches == 0
    else:
        assert ks_mid.mttkrp_launches > 0 and ks.mttkrp_launches > ks_mid.mttkrp_launches


def test_line_search_aware_pair_schedule(cc, oracle, inputs):
    """Plan M, models admitted together: every model extrapolates in the same sweep, every `interval` sweeps, and the T
    handed over that sweep's boundary is always lost.  After two such events the engine predicts the next one and runs
    the plain fused MTTKRP at that sweep's last mode instead of a TTM nobody would consume: same models as the oracle,
    fewer TTM launches, one fused launch per predicted event -- and exactly the models of the unscheduled run up to the
    association of the sums."""
    # (an ODD interval: with an even one the events fall on sweeps that hand no T over, and there is nothing to save;
    # the data of test_tree_line_search_vs_oracle: far from convergence, so no accept / revert decision is a near-tie
    # that the different association of one MTTKRP's sums could flip)
    modes, ranks, iters, interval = [20, 20, 20], [2, 3, 4, 5, 20, 17], 25, 5
    X = inputs.tensor(modes, 3)
    base = make_models(inputs, modes, ranks, seed=5)
    old = {k: os.environ.get(k) for k in ("CALS_HIP_TREE", "CALS_LS_SCHEDULE_OFF")}
    os.environ["CALS_HIP_TREE"] = "M"
    out = {}
    try:
        for off in (True, False):
            if off:
                os.environ["CALS_LS_SCHEDULE_OFF"] = "1"
            else:
                os.environ.pop("CALS_LS_SCHEDULE_OFF", None)
            e = cc.Engine(modes, sum(ranks))
            e.set_tensor(X)
            e.set_params(cc.default_params(max_iterations=iters, force_max_iter=1, line_search=1,
                                           line_search_interval=interval))
            gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
            for m in gm:
                e.enqueue(m)
            e.set_profiling(1)
            rep = e.run()
            ks = e.kernel_stats()
            e.close()
            out[off] = (gm, ks.ttm_launches, ks.mttkrp_launches, rep.iter)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    (g_off, ttm_off, fused_off, it_off), (g_on, ttm_on, fused_on, it_on) = out[True], out[False]
    # The first two mass extrapolations teach the period, later ones are predicted -- as long as the models stay in
    # phase (a model whose step is reverted falls one sweep behind).  Every correct prediction turns one TTM into one
    # fused launch; a wrong one costs an extra GEMM and clears the pattern.
    # (Fused launches also serve the stale-column patch, with or without the schedule: only the TTM count is telling.)
    assert it_on == it_off
    assert ttm_on < ttm_off and fused_on > 0
    om = [oracle.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    oracle.cp_cals(X, modes, om, oracle.default_params(max_iterations=iters, force_max_iter=1, buffer_size=sum(ranks),
                                                       mttkrp_method=oracle.MTTKRP, line_search=1,
                                                       line_search_interval=interval))
    for a, b, c in zip(g_on, g_off, om):
        for fa, fb, fc in zip(a.factors, b.factors, c.factors):
            assert rel(fa, fb) < 1e-9 and rel(fa, fc) < 1e-8


def test_tree_error_checking_line_search_vs_oracle(cc, oracle, inputs, plan):
    """An accepted ERROR_CHECKING step rewrites a model's factors: a T pending across the sweep
    boundary has to be dropped exactly as for the other method."""
    modes, ranks = [22, 19, 17], [2, 3, 4, 5, 9, 14, 1, 20]
    X = inputs.low_rank_tensor(modes, 6, seed=31)[0] + 0.1 * inputs.tensor(modes, 8)
    # tolerance-driven stop: a converged model's candidates differ from its error by rounding only,
    # and the accept/reject decision would then depend on the association of the sums
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 40, force_max_iter=0, tol=1e-7,
                                line_search=1, line_search_interval=5, line_search_method=1)
    assert rep.iter == ro.iter
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > rep.ls_failed > 0
    _assert_models_match(gm, om, ro.X_norm ** 2)


def test_tree_jackknife_models_vs_oracle(cc, oracle, inputs, plan):
    modes, comp = [20, 9, 12], 5
    X = inputs.low_rank_tensor(modes, comp, seed=21)[0]
    jk = [(0, i) for i in range(20)]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, [comp] * 20, X, 20, jk=jk)
    _assert_models_match(gm, om, ro.X_norm ** 2)
    for i, m in enumerate(gm):
        assert np.all(m.factors[0][i, :] == 0.0)


def test_tree_queue_eviction_compress_vs_oracle(cc, oracle, inputs, plan):
    """60 models through a 30-column buffer with tolerance-driven eviction: the active width R
    changes from sweep to sweep, so T, Pt and the team split are re-derived every sweep."""
    modes = [13, 12, 11]
    ranks = [1 + (k % 10) for k in range(60)]
    X = inputs.low_rank_tensor(modes, 4, seed=5)[0] + 0.1 * inputs.tensor(modes, 7)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 40, buffer=30,
                                force_max_iter=0, tol=1e-6)
    assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        # reference criterion: reconstructed tensors agree (MODEL_DIFF_ACC, test_cals.cpp:7,81-84)
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-9


def test_full_size_c3_tree_identities(cc, inputs, plan):
    """300^3, 256 models (R = 2656): all-ones model => mode sums of X; random columns vs einsum."""
    if plan == "0":
        pytest.skip("plain plan at full size: test_gpu_parity.py")
    modes = [300, 300, 300]
    X = inputs.tensor(modes, 0)
    ranks = inputs.ranks_1_to_20(256)
    base = make_models(inputs, modes, ranks)
    for n in range(3):
        base[0][0][n][:] = 1.0
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    for fs, lam, _ in base:
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    X3 = X.reshape(modes, order="F")
    G = [e.debug_mttkrp(n) for n in range(3)]
    if plan == "M":  # default path under M is "second" for every mode; also check one fused G
        assert rel(e.debug_mttkrp(2, "first")[:, 0], X3.sum(axis=(0, 1))) < 1e-12
    assert rel(G[0][:, 0], X3.sum(axis=(1, 2))) < 1e-12
    assert rel(G[1][:, 0], X3.sum(axis=(0, 2))) < 1e-12
    assert rel(G[2][:, 0], X3.sum(axis=(0, 1))) < 1e-12
    rng = np.random.default_rng(0)
    facs = [np.hstack([fs[n] for fs, _, _ in base]) for n in range(3)]
    for c in rng.integers(1, sum(ranks), size=3):
        assert rel(G[0][:, c], np.einsum("ijk,j,k->i", X3, facs[1][:, c], facs[2][:, c])) < 1e-11
        assert rel(G[1][:, c], np.einsum("ijk,i,k->j", X3, facs[0][:, c], facs[2][:, c])) < 1e-11
        assert rel(G[2][:, c], np.einsum("ijk,i,j->k", X3, facs[0][:, c], facs[1][:, c])) < 1e-11
    e.close()
