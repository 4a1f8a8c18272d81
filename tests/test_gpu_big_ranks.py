"""Models of rank 65..256 (the reference is unbounded, include/ktensor.h; round 1 stopped at 64):
update_body_huge keeps H / L in a global scratch block per model and solves every factor row in place.
Same oracle, same tolerances as the other bodies; mixed with small models in one buffer, queued,
jackknifed, with both line-search methods, under every MTTKRP plan, in fp32 storage, and with the NNLS update
(nnls_huge_kernel: (rank + 63) / 64 active-set words per row, H and the Cholesky factors in global scratch)."""
import os

import numpy as np
import pytest

from helpers import make_models, rel
from test_gpu_parity import TOL_KERNEL, TOL_RUN, _assert_models_match, _run_both, engine_with

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks", [[65], [100], [128], [7, 200, 70, 20], [256]])
def test_single_sweeps_state_vs_oracle(cc, oracle, inputs, ranks):
    modes = [90, 80, 70]
    X = inputs.tensor(modes, 6)
    prm = cc.default_params(max_iterations=100, force_max_iter=1)
    e, gm, base = engine_with(cc, inputs, modes, ranks, X, params=prm)
    e.admit()
    e.sweep(2)
    om = [oracle.Model(fs, lam) for fs, lam, _ in base]
    oracle.cp_cals(X, modes, om, oracle.default_params(max_iterations=2, force_max_iter=1,
                                                       buffer_size=sum(ranks), mttkrp_method=oracle.MTTKRP))
    lam = e.debug_lambda()
    col = 0
    for m, g in zip(om, gm):
        r = m.rank
        for n in range(3):
            F = e.debug_factor(n)[:, col:col + r]
            assert rel(F, m.factors[n]) < 1e-10      # condition of a 256 x 256 Hadamard of Gramians
            Gm = e.debug_gramian(n)[:r, col:col + r]
            assert rel(Gm, m.factors[n].T @ m.factors[n]) < 1e-10
        assert rel(lam[col:col + r], m.lam) < 1e-10
        st, c = e.debug_status(g)
        assert c == col and st.iters == 3
        assert abs(st.approx_error - m.error) <= 1e-9 * max(1.0, m.error)
        col += r
    e.close()


@pytest.mark.parametrize("plan", ["M", "A", "B", "0"])
def test_forced_iterations_line_search_and_plans(cc, oracle, inputs, plan):
    old = os.environ.get("CALS_HIP_TREE")
    os.environ["CALS_HIP_TREE"] = plan
    try:
        modes, ranks = [60, 50, 45], [70, 3, 96, 20, 65]
        X = inputs.low_rank_tensor(modes, 8, seed=3)[0] + 0.2 * inputs.tensor(modes, 4)
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 9, line_search=1, line_search_interval=3)
    finally:
        if old is None:
            os.environ.pop("CALS_HIP_TREE", None)
        else:
            os.environ["CALS_HIP_TREE"] = old
    assert (rep.iter, rep.ls_performed, rep.ls_failed) == (ro.iter, ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > 0
    _assert_models_match(gm, om, ro.X_norm ** 2)


def test_queue_eviction_compress_with_mixed_ranks(cc, oracle, inputs):
    """big and small models through a buffer that holds two big ones at most: admission, tolerance-driven
    eviction and compress move 256-row Gramian columns next to 64-row ones."""
    modes = [40, 35, 30]
    ranks = [80, 5, 66, 12, 90, 3, 70, 20, 100, 1]
    X = inputs.low_rank_tensor(modes, 4, seed=5)[0] + 0.1 * inputs.tensor(modes, 7)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 30, buffer=180, force_max_iter=0, tol=1e-5)
    assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
    _assert_models_match(gm, om, ro.X_norm ** 2, tol=1e-7)


def test_jackknife_and_fp32_storage(cc, oracle, inputs):
    modes, ranks = [50, 44, 40], [72, 9]
    X = inputs.tensor(modes, 2)
    jk = [(0, 3), (0, 17)]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 6, jk=jk)
    _assert_models_match(gm, om, ro.X_norm ** 2)
    for (mode, fiber), m in zip(jk, gm):
        assert not m.factors[mode][fiber, :].any()
    # fp32 storage: the in-place row solves round to float at every step -- stated tolerance 5e-3 after 4 sweeps
    base = make_models(inputs, modes, ranks, seed=1)
    e = cc.Engine(modes, sum(ranks), dtype="f32")
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=4, force_max_iter=1))
    g32 = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    for m in g32:
        e.enqueue(m)
    e.run()
    e.close()
    o64 = [oracle.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    oracle.cp_cals(X, modes, o64, oracle.default_params(max_iterations=4, force_max_iter=1, buffer_size=sum(ranks),
                                                        mttkrp_method=oracle.MTTKRP))
    for a, b in zip(g32, o64):
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < 5e-3


def test_error_checking_line_search(cc, oracle, inputs):
    """ls::ERROR_CHECKING_SERIAL with models above rank 64: the candidate's r x r Hadamard of Gramians
    goes through two global scratch blocks instead of LDS."""
    modes, ranks = [26, 22, 19], [70, 4, 66, 12]
    X = inputs.low_rank_tensor(modes, 6, seed=31)[0] + 0.1 * inputs.tensor(modes, 8)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 30, force_max_iter=0, tol=1e-6,
                                line_search=1, line_search_interval=4, line_search_method=1)
    assert rep.iter == ro.iter
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > 0
    _assert_models_match(gm, om, ro.X_norm ** 2, tol=1e-7)


def _nonneg_tensor(inputs, modes, rank, seed, noise=0.05):
    X, _, _ = inputs.low_rank_tensor(modes, rank, seed=seed)
    return np.abs(X) + noise * inputs.tensor(modes, seed + 1)


def _check_nnls(gm, om, rep, ro, tol=1e-8, nonneg=True):
    assert rep.nnls_status == 0 and ro.nnls_status == 0
    for m in gm:
        for f in m.factors:
            assert (f >= 0.0).all() or not nonneg
    _assert_models_match(gm, om, ro.X_norm ** 2, tol=tol)
    for a, b in zip(gm, om):
        for fa, fb in zip(a.factors, b.factors):
            scale = max(np.abs(fb).max(), 1e-300)
            assert np.abs(fa[fb == 0.0]).max(initial=0.0) <= 1e-9 * scale
            assert np.abs(fb[fa == 0.0]).max(initial=0.0) <= 1e-9 * scale


@pytest.mark.parametrize("modes,ranks,iters", [
    ([30, 26, 22], [65], 6),
    ([30, 26, 22], [70, 4, 129, 20], 5),            # two and three mask words next to one-word models
    ([75, 12, 10], [66, 9], 6),                      # several row chunks per model
    ([24, 20, 18], [256], 3),
])
def test_nnls_forced_iterations_vs_oracle(cc, oracle, inputs, modes, ranks, iters):
    X = _nonneg_tensor(inputs, modes, 5, seed=17)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, iters, update_method=1)
    assert rep.iter == ro.iter == iters
    _check_nnls(gm, om, rep, ro)
    zeros = sum(int((f == 0.0).sum()) for m in gm for f in m.factors)
    assert zeros > 50        # rank >> rank(X): most rows end with many constraints active


def test_nnls_signed_tensor_queue_and_compress(cc, oracle, inputs):
    """entries of both signs (warm start, both exchange loops, caught Cholesky failures) and a buffer smaller
    than the queue: the extra mask words travel with their models' columns through eviction and compress."""
    modes = [22, 19, 17]
    ranks = [80, 5, 66, 12, 90, 3, 70]
    X = inputs.tensor(modes, 5) - 0.1
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 12, buffer=170, force_max_iter=0, tol=1e-4,
                                update_method=1)
    assert rep.iter == ro.iter
    assert [m.iters for m in gm] == [m.iters for m in om]
    _check_nnls(gm, om, rep, ro)


def test_nnls_line_search_and_jackknife(cc, oracle, inputs):
    """Ktensor::copy carries the active sets: the backup / revert moves every mask word of a model."""
    modes, ranks = [20, 18, 16], [68, 4, 100]
    X = _nonneg_tensor(inputs, modes, 5, seed=3, noise=0.3)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 9, line_search=1, line_search_interval=3,
                                update_method=1)
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > 0
    _check_nnls(gm, om, rep, ro, nonneg=False)
    jk = [(0, 2), (0, 11)]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, [70, 6], X, 5, jk=jk, update_method=1)
    _check_nnls(gm, om, rep, ro)
    for (mode, fiber), m in zip(jk, gm):
        assert not m.factors[mode][fiber, :].any()


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        modes = [int(v) for v in rng.integers(12, 34, size=3)]
        n_models = int(rng.integers(3, 9))
        ranks = [int(rng.integers(65, 140)) if rng.integers(0, 2) else int(rng.integers(1, 40)) for _ in range(n_models)]
        ranks[int(rng.integers(0, n_models))] = int(rng.integers(65, 140))
        buffer = int(rng.integers(max(ranks), max(max(ranks) + 1, sum(ranks))))
        out.append((modes, ranks, buffer, int(rng.integers(0, 2)), ["0", "A", "B", "M"][int(rng.integers(0, 4))],
                    int(rng.integers(0, 1 << 30))))
    return out


_N_BIG = int(os.environ.get("CALS_SOAK_BIG", "10"))


@pytest.mark.parametrize("modes,ranks,buffer,nnls,plan,seed", _random_cases(_N_BIG, 4242 + int(os.environ.get("CALS_SOAK_SEED", "0"))))
def test_random_queue_life_cycle_with_big_ranks(cc, oracle, inputs, modes, ranks, buffer, nnls, plan, seed):
    """random mixes of one-, two- and three-word models through a buffer smaller than the queue, both update
    methods, every MTTKRP plan: same admission order, per-model sweep counts and fitted tensors as the oracle."""
    from helpers import reconstruct
    old = os.environ.get("CALS_HIP_TREE")
    os.environ["CALS_HIP_TREE"] = plan
    try:
        X = inputs.low_rank_tensor(modes, 5, seed=seed % 1000)[0] + 0.05 * inputs.tensor(modes, seed % 977)
        kw = dict(tol=1e-4, force_max_iter=0)
        if nnls:
            kw["update_method"] = 1
            X = np.abs(X)
        else:
            kw.update(line_search=seed & 1, line_search_interval=3)
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 12, buffer=buffer, **kw)
        assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
        assert rep.nnls_status == 0 and ro.nnls_status == 0
        for a, b in zip(gm, om):
            assert a.iters == b.iters
            d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
            assert d <= 1e-8 * max(1.0, np.linalg.norm(X))
    finally:
        if old is None:
            os.environ.pop("CALS_HIP_TREE", None)
        else:
            os.environ["CALS_HIP_TREE"] = old


def test_rank_limit_fails_loudly(cc, inputs):
    modes = [30, 20, 10]
    e = cc.Engine(modes, 300)
    (fs, lam), = inputs.model_factors(modes, [257], 1)
    with pytest.raises(cc.CalsHipError):
        e.enqueue(cc.Model(fs, lam))
    e.close()
