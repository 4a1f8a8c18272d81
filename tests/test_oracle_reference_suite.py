"""The reference's own test-suite restated on the CPU oracle (this is what pins the oracle: the
reference ships no golden vectors and cannot be built in this image, see oracle/cals_oracle.h).

  tests/als/test_als.cpp:10-60    ComputeCorrectResult3D      -> test_als_variants_agree_3d
  tests/als/test_als.cpp:62-103   ComputeCorrectResultConstrained3D -> test_als_variants_agree_constrained_3d
  tests/als/test_als.cpp:105-123  ComputeCorrectResult4D      -> test_als_4d
  tests/als/test_als.cpp:125-145  ComputeCorrectError         -> test_fast_error_equals_slow_error
  tests/cals/test_cals.cpp:13-86  SimpleCorrectness           -> test_cals_equals_als
  tests/cals/test_cals.cpp:88-179 LineSearchCorrectness       -> test_cals_equals_als_line_search
  tests/cals/test_cals.cpp:181-297 Jackknifing.LogicCorrectness -> test_jackknife_logic
The NNLS row solver is additionally pinned against SciPy's Lawson-Hanson nnls (an independent
implementation of the same problem, whose solution is unique for an SPD H).
Inputs come from the repo's portable generator, not from std::mt19937 (libstdc++-specific).
"""
import numpy as np
import pytest

from helpers import make_models, reconstruct

MODEL_DIFF_ACC_CALS = 1e-11  # tests/cals/test_cals.cpp:7
VARIANT_ACC = 1e-8           # tests/als/test_als.cpp:58


def test_als_variants_agree_3d(oracle, inputs):
    O = oracle
    modes = [9, 4, 2]
    X, _, _ = inputs.low_rank_tensor(modes, 5, seed=101)
    for p in range(20):
        errs = []
        for method in (O.MTTKRP, O.TWOSTEP0, O.TWOSTEP1, O.AUTO):
            (fs, lam, _), = make_models(inputs, modes, [5], seed=200 + p)
            m = O.Model(fs, lam)
            prm = O.default_params(max_iterations=100, mttkrp_method=method, line_search=0)
            O.cp_als(X, modes, m, prm)
            slow = np.linalg.norm(X - reconstruct(m.factors, m.lam, modes))
            assert np.isfinite(slow) and slow < 50
            errs.append(slow)
        for e in errs:
            assert abs(e - errs[0]) <= VARIANT_ACC


def test_als_variants_agree_constrained_3d(oracle, inputs):
    O = oracle
    modes = [18, 17, 16]
    X, _, _ = inputs.low_rank_tensor(modes, 5, seed=111)
    for p in range(20):
        errs = []
        for method in (O.MTTKRP, O.TWOSTEP0, O.TWOSTEP1, O.AUTO):
            (fs, lam, _), = make_models(inputs, modes, [5], seed=300 + p)
            m = O.Model(fs, lam)
            prm = O.default_params(max_iterations=100, mttkrp_method=method, update_method=O.NNLS)
            rep = O.cp_als(X, modes, m, prm)
            assert rep.nnls_status == 0
            for f in m.factors:
                assert (f >= 0.0).all()
            slow = np.linalg.norm(X - reconstruct(m.factors, m.lam, modes))
            assert np.isfinite(slow) and slow < 50
            errs.append(slow)
        for e in errs:
            assert abs(e - errs[0]) <= VARIANT_ACC


def test_nnls_rows_equal_lawson_hanson(oracle):
    from scipy.optimize import nnls
    rng = np.random.default_rng(7)
    for trial in range(60):
        r, rows = int(rng.integers(1, 21)), int(rng.integers(1, 24))
        A = rng.random((rows + r + 5, r)) - (0.5 if trial % 3 == 0 else 0.0)
        H = A.T @ A
        Y = np.maximum(rng.standard_normal((rows, r)), 0) @ H + 0.1 * rng.standard_normal((rows, r))
        P, act, st = oracle.update_factor_nnls(Y, H)
        assert st == 0
        L = np.linalg.cholesky(H)
        for i in range(rows):
            x, _ = nnls(L.T, np.linalg.solve(L, Y[i]))  # normal equations H x = y
            assert np.abs(x - P[i]).max() <= 1e-10 * max(1.0, np.abs(x).max())
            assert ((P[i] == 0) | (act[i] == 0)).all()
        # warm start from the previous sweep's passive set: same minimiser
        Y2 = Y + 1e-3 * rng.standard_normal(Y.shape)
        P2, _, st2 = oracle.update_factor_nnls(Y2, H, act)
        P3, _, st3 = oracle.update_factor_nnls(Y2, H)
        assert st2 == 0 and st3 == 0
        assert np.abs(P2 - P3).max() <= 1e-9 * max(1.0, np.abs(P3).max())


def test_als_4d(oracle, inputs):
    O = oracle
    modes = [3, 3, 3, 3]
    X, _, _ = inputs.low_rank_tensor(modes, 5, seed=102)
    (fs, lam, _), = make_models(inputs, modes, [7], seed=103)
    m = O.Model(fs, lam)
    O.cp_als(X, modes, m, O.default_params(max_iterations=100))
    slow = np.linalg.norm(X - reconstruct(m.factors, m.lam, modes))
    assert np.isfinite(slow) and slow < 1e-1


def test_fast_error_equals_slow_error(oracle, inputs):
    O = oracle
    modes = [9, 3, 2]
    X, _, _ = inputs.low_rank_tensor(modes, 5, seed=104)
    (fs, lam, _), = make_models(inputs, modes, [5], seed=105)
    m = O.Model(fs, lam)
    O.cp_als(X, modes, m, O.default_params(max_iterations=3))
    slow = np.linalg.norm(X - reconstruct(m.factors, m.lam, modes))
    assert abs(m.error - slow) <= 1e-10


def _cals_vs_als(O, inputs, ls, method, n_copies):
    modes = [13, 12, 11]
    X, _, _ = inputs.low_rank_tensor(modes, 10, seed=106)
    ranks = [r for r in range(1, 13) for _ in range(n_copies)]
    np.random.default_rng(0).shuffle(ranks)
    base = make_models(inputs, modes, ranks, seed=107)
    A = [O.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    B = [O.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    prm = O.default_params(max_iterations=1000, tol=1e-5, buffer_size=30, line_search=ls,
                           line_search_interval=10, line_search_step=0.0,
                           line_search_method=method, mttkrp_method=O.AUTO, threads=4)
    rep = O.cp_cals(X, modes, A, prm)
    assert rep.n_ktensors == len(ranks)
    perf = fail = 0
    for a, b in zip(A, B):
        r = O.cp_als(X, modes, b, prm)
        perf += r.ls_performed
        fail += r.ls_failed
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= MODEL_DIFF_ACC_CALS
        assert a.iters == b.iters
    if ls:
        assert (rep.ls_performed, rep.ls_failed) == (perf, fail)
        assert perf > 0


def test_cals_equals_als(oracle, inputs):
    _cals_vs_als(oracle, inputs, 0, 0, 10)


@pytest.mark.parametrize("method", [0, 1])  # NO_ERROR_CHECKING, ERROR_CHECKING_SERIAL
def test_cals_equals_als_line_search(oracle, inputs, method):
    _cals_vs_als(oracle, inputs, 1, method, 6)


def test_jackknife_logic(oracle, inputs):
    """jk models inside one CALS call on the full X == ALS on physically sub-sampled tensors."""
    O = oracle
    modes = [20, 9, 12]
    comp = 5
    X, _, _ = inputs.low_rank_tensor(modes, comp, seed=108)
    X3 = X.reshape(modes, order="F")
    (fs, lam, _), = make_models(inputs, modes, [comp], seed=109)
    prm = O.default_params(max_iterations=60, tol=1e-4, buffer_size=18, force_max_iter=1, threads=4)
    cals_models, als_models, als_tensors = [], [], []
    for i in range(modes[0]):
        f = [a.copy() for a in fs]
        f[0][i, :] *= 0.0
        cals_models.append(O.Model(f, lam.copy(), jk=(0, i)))
        als_models.append(O.Model([np.asfortranarray(np.delete(fs[0], i, axis=0)), fs[1].copy(),
                                   fs[2].copy()], lam.copy()))
        als_tensors.append(np.ascontiguousarray(np.delete(X3, i, axis=0).ravel(order="F")))
    ref_cals = O.Model([a.copy() for a in fs], lam.copy())
    ref_als = O.Model([a.copy() for a in fs], lam.copy())
    O.cp_cals(X, modes, cals_models + [ref_cals], prm)
    jm = [modes[0] - 1, modes[1], modes[2]]
    for i in range(modes[0]):
        O.cp_als(als_tensors[i], jm, als_models[i], prm)
        a = cals_models[i]
        t1 = reconstruct([np.delete(a.factors[0], i, axis=0), a.factors[1], a.factors[2]], a.lam, jm)
        t2 = reconstruct(als_models[i].factors, als_models[i].lam, jm)
        assert np.linalg.norm(t1 - t2) <= MODEL_DIFF_ACC_CALS
    O.cp_als(X, modes, ref_als, prm)
    assert np.linalg.norm(reconstruct(ref_als.factors, ref_als.lam, modes) -
                          reconstruct(ref_cals.factors, ref_cals.lam, modes)) <= MODEL_DIFF_ACC_CALS
