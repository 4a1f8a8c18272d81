"""CPU tests of the jackknife post-processing: the assignment solver and
utils::jk_permutation_adjustment (reference src/utils/utils.cpp:54-101).

Pins, in this order:
  1. the product's solve_rectangular_linear_sum_assignment == the REFERENCE's own
     extern/rectangular_lsap/rectangular_lsap.cpp (compiled from where it lies into
     oracle/_ref/librectangular_lsap.so by oracle/Makefile; the one piece of the reference that builds
     here) on random square, wide, tall, tied (integer) and constant cost matrices -- identical
     (a, b), not merely equal value -- and == scipy.optimize.linear_sum_assignment;
  2. jk_permutation_adjustment on a hand-made 3-CYCLE, where the reference's call (column-major M
     handed to a row-major solver, then new(:, cur) = old(:, solved[cur])) gives a result that differs
     from "align the replica with the overall model": product == oracle == the reference's call
     emulated with the reference's own compiled solver == the column order worked out by hand.
"""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LSAP = os.path.join(ROOT, "oracle", "_ref", "librectangular_lsap.so")


def _bind(lib):
    f = lib.solve_rectangular_linear_sum_assignment
    f.argtypes = [ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_bool, ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    return f


def _solve(f, cost_rowmajor, maximize):
    nr, nc = cost_rowmajor.shape
    c = np.ascontiguousarray(cost_rowmajor, dtype=np.float64)
    k = min(nr, nc)
    a = np.full(k, -7, dtype=np.int64)
    b = np.full(k, -7, dtype=np.int64)
    rc = f(nr, nc, c.ctypes.data, bool(maximize), a.ctypes.data, b.ctypes.data)
    return rc, a, b


@pytest.fixture(scope="module")
def product():
    return ctypes.CDLL(os.path.join(ROOT, "cp-cals_amd", "libcals.so"))


@pytest.fixture(scope="module")
def ref_lsap():
    if not os.path.exists(REF_LSAP):
        pytest.skip("oracle/_ref/librectangular_lsap.so not built (needs /root/reference at build time)")
    return _bind(ctypes.CDLL(REF_LSAP))


def _cases():
    rng = np.random.default_rng(7)
    out = []
    for _ in range(400):
        nr, nc = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            nc = nr                                   # square, continuous: the reference's call sites
            M = rng.standard_normal((nr, nc))
        elif kind == 1:
            M = rng.standard_normal((nr, nc))         # rectangular
        elif kind == 2:
            M = rng.integers(0, 3, (nr, nc)).astype(float)   # heavy ties
        else:
            M = np.full((nr, nc), float(rng.integers(-2, 3)))  # constant
        out.append((M, bool(rng.integers(0, 2))))
    return out


def test_lsap_equals_reference_build(product, ref_lsap):
    mine = _bind(product)
    for M, mx in _cases():
        rc_r, a_r, b_r = _solve(ref_lsap, M, mx)
        rc_m, a_m, b_m = _solve(mine, M, mx)
        assert rc_m == rc_r == 0
        assert np.array_equal(a_m, a_r) and np.array_equal(b_m, b_r), (M, mx, a_m, b_m, a_r, b_r)


def test_lsap_invalid_and_infeasible_codes(product, ref_lsap):
    mine = _bind(product)
    bad = np.array([[1.0, np.nan], [0.0, 1.0]])
    assert _solve(mine, bad, False)[0] == _solve(ref_lsap, bad, False)[0] == -2
    ninf = np.array([[1.0, -np.inf], [0.0, 1.0]])
    assert _solve(mine, ninf, False)[0] == _solve(ref_lsap, ninf, False)[0] == -2
    inf = np.array([[np.inf, np.inf], [0.0, 1.0]])
    assert _solve(mine, inf, False)[0] == _solve(ref_lsap, inf, False)[0] == -1


def test_lsap_equals_scipy(product):
    from scipy.optimize import linear_sum_assignment
    mine = _bind(product)
    for M, mx in _cases():
        rc, a, b = _solve(mine, M, mx)
        assert rc == 0
        ra, rb = linear_sum_assignment(M, maximize=mx)
        assert M[a, b].sum() == pytest.approx(M[ra, rb].sum(), abs=1e-12)
        assert len(set(a)) == len(a) and len(set(b)) == len(b)


def _three_cycle(seed=3):
    """Overall model with well-separated columns o0, o1, o2 and replicas whose columns are the
    overall model's in the order [o1, o2, o0] (a 3-cycle), slightly perturbed."""
    rng = np.random.default_rng(seed)
    modes = [4, 6, 5]
    r = 3
    overall = []
    for n in range(3):
        q, _ = np.linalg.qr(rng.standard_normal((max(modes[n], r), r)))
        f = q[: modes[n], :r] + 0.01 * rng.standard_normal((modes[n], r))
        overall.append(np.asfortranarray(f / np.linalg.norm(f, axis=0)))
    cyc = [1, 2, 0]
    replicas = []
    for m in range(modes[0]):
        reps = [np.asfortranarray(f[:, cyc] + 1e-3 * rng.standard_normal(f.shape)) for f in overall]
        replicas.append(reps)
    return modes, r, overall, replicas, cyc


def _ptrs(arrs):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def test_jk_permutation_three_cycle_pins_reference_behaviour(product, oracle):
    modes, r, overall, replicas, cyc = _three_cycle()
    md = (ctypes.c_int64 * 3)(*modes)
    # hand-worked expectation (see the module docstring): replica column b holds o_{cyc[b]}; the solver
    # (rows = replica columns) returns solved[b] = cyc[b] = [1, 2, 0]; new(:, cur) = old(:, solved[cur])
    # = o_{cyc[cyc[cur]]} -> overall-column labels [2, 0, 1]: NOT [0, 1, 2].
    expect_labels = [cyc[cyc[c]] for c in range(r)]
    assert expect_labels == [2, 0, 1]

    # (i) product
    mine = [[f.copy(order="F") for f in reps] for reps in replicas]
    flat = [f for reps in mine for f in reps]
    product.cals_jk_permutation_adjustment.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_void_p, ctypes.c_void_p]
    assert product.cals_jk_permutation_adjustment(3, md, r, _ptrs(overall), _ptrs(flat)) == 0
    # (ii) oracle
    orc = oracle.lib()
    orc.or_jk_permutation_adjust.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_void_p]
    theirs = [[f.copy(order="F") for f in reps] for reps in replicas]
    for reps in theirs:
        assert orc.or_jk_permutation_adjust(3, md, r, _ptrs(overall), _ptrs(reps)) == 0
    for m in range(modes[0]):
        for n in range(3):
            assert np.array_equal(mine[m][n], theirs[m][n])
            # every new column cur is the replica's old column solved[cur] = cyc[cur] ...
            assert np.array_equal(mine[m][n], replicas[m][n][:, cyc])
            # ... which carries the overall column expect_labels[cur]
            lab = np.argmax(np.abs(overall[n].T @ mine[m][n]), axis=0)
            if n > 0:  # modes 1, 2 are the ones the matching is built from
                assert lab.tolist() == expect_labels
    # what "line the replica up with the overall model" would have given -- documented as DIFFERENT
    aligned = replicas[0][1][:, np.argsort(cyc)]
    assert np.argmax(np.abs(overall[1].T @ aligned), axis=0).tolist() == [0, 1, 2]
    assert not np.allclose(aligned, mine[0][1])


def test_jk_permutation_reference_call_emulated_with_reference_solver(product, ref_lsap):
    """The reference's call pattern (utils.cpp:69-97) re-enacted in numpy around the reference's OWN
    compiled solver: M built column-major, its buffer passed as is, new(:, cur) = old(:, solved[cur])."""
    modes, r, overall, replicas, cyc = _three_cycle(seed=11)
    md = (ctypes.c_int64 * 3)(*modes)
    mine = [[f.copy(order="F") for f in reps] for reps in replicas]
    flat = [f for reps in mine for f in reps]
    product.cals_jk_permutation_adjustment.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_void_p, ctypes.c_void_p]
    assert product.cals_jk_permutation_adjustment(3, md, r, _ptrs(overall), _ptrs(flat)) == 0
    for m in range(modes[0]):
        M = np.asfortranarray(overall[1].T @ replicas[m][1] + overall[2].T @ replicas[m][2])  # column-major
        buf = M.ravel(order="F").copy()                     # the buffer the reference passes
        a = np.zeros(r, dtype=np.int64)
        b = np.zeros(r, dtype=np.int64)
        assert ref_lsap(r, r, buf.ctypes.data, True, a.ctypes.data, b.ctypes.data) == 0
        for n in range(3):
            assert np.array_equal(mine[m][n], replicas[m][n][:, b])


def test_jk_permutation_involution_is_the_aligned_one(product):
    """For a swap (an involution) the reference's result IS the aligned one -- the common case, and why
    the reference's FunctionCorrectness test cannot see the orientation."""
    rng = np.random.default_rng(5)
    modes, r = [3, 5, 4], 3
    overall = []
    for n in range(3):
        q, _ = np.linalg.qr(rng.standard_normal((max(modes[n], r), r)))
        overall.append(np.asfortranarray(q[: modes[n], :r] + 0.01 * rng.standard_normal((modes[n], r))))
    swap = [1, 0, 2]
    reps = [[np.asfortranarray(f[:, swap]) for f in overall] for _ in range(modes[0])]
    flat = [f for rp in reps for f in rp]
    md = (ctypes.c_int64 * 3)(*modes)
    product.cals_jk_permutation_adjustment.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_void_p, ctypes.c_void_p]
    assert product.cals_jk_permutation_adjustment(3, md, r, _ptrs(overall), _ptrs(flat)) == 0
    for rp in reps:
        for n in range(3):
            assert np.array_equal(rp[n], overall[n])
