"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and (at BASELINE.json's full sizes) through
size-independent properties.  Tolerances (fp64, stated by BASELINE.json north_star): factors and
lambda within 1e-8 relative Frobenius of the reference path after the same number of sweeps; single
kernels are held to 1e-12."""
import os

import numpy as np
import pytest

from helpers import make_models, numpy_mttkrp, reconstruct, rel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_KERNEL = 1e-12
TOL_RUN = 1e-8


def engine_with(cc, inputs, modes, ranks, X, seed=1, jk=None, buffer=None, params=None):
    base = make_models(inputs, modes, ranks, seed=seed, jk=jk)
    e = cc.Engine(modes, sum(ranks) if buffer is None else buffer)
    e.set_tensor(X)
    if params is not None:
        e.set_params(params)
    gm = [cc.Model(fs, lam, jk=j) for fs, lam, j in base]
    for m in gm:
        e.enqueue(m)
    return e, gm, base


def test_native_library_is_the_one_running(cc):
    lib = cc.load_library()
    assert lib.cals_hip_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libcals_hip.so" in f.read()


@pytest.mark.parametrize("modes,ranks", [
    ([20, 20, 20], [2, 3, 4, 5]),            # BASELINE config 1
    ([7, 5, 3], [1, 2, 3]),                  # tiny, ragged
    ([13, 12, 11], list(range(1, 13))),      # reference test shape
    ([100, 37, 41], None),                   # non-cubic, 20 models ranks 1..20
    ([299, 301, 41], None),                  # BASELINE config 4 shape
    ([330, 17, 9], [5, 20, 7]),              # more than 20 m-tiles: two M blocks
    ([40, 30, 20], [20] * 21),               # R = 420: four column blocks
    ([3, 3, 3, 3], [7, 2]),                  # 4-way (reference ComputeCorrectResult4D shape)
    ([6, 5, 4, 3], [3, 4, 5]),
    ([4, 3, 5, 2, 3], [2, 3]),               # 5-way
])
def test_mttkrp_every_mode_vs_oracle(cc, oracle, inputs, modes, ranks):
    if ranks is None:
        ranks = inputs.ranks_1_to_20(20 if modes[0] == 100 else 10)
    X = inputs.tensor(modes, 0)
    e, gm, base = engine_with(cc, inputs, modes, ranks, X)
    e.admit()
    facs = [np.asfortranarray(np.hstack([fs[n] for fs, _, _ in base])) for n in range(len(modes))]
    for n in range(len(modes)):
        G = e.debug_mttkrp(n)
        assert rel(G, oracle.mttkrp(X, modes, facs, n, oracle.MTTKRP)) < TOL_KERNEL
        if len(modes) == 3:  # the reference's other variants agree too (test_als.cpp:10-60)
            assert rel(G, oracle.mttkrp(X, modes, facs, n, oracle.TWOSTEP1)) < TOL_KERNEL
    e.close()


@pytest.mark.parametrize("modes,rank,dtype", [([20, 20, 20], 5, "f64"), ([13, 12, 11], 33, "f64"),
                                              ([40, 30, 20], 300, "f64"), ([6, 5, 4, 3], 4, "f64"),
                                              ([29, 31, 11], 7, "f32")])
def test_kernel_level_mttkrp_of_one_ktensor_vs_oracle(cc, oracle, inputs, modes, rank, dtype):
    """cals_hip_mttkrp = mttkrp::mttkrp(X, u, workspace, mode, params) (src/utils/mttkrp.cpp:562-614) for one
    Ktensor on an idle engine: the callable the reference's MTTKRP micro-benchmark times.  The engine's buffers
    are left as found (a run right after it equals a run on a fresh engine), rank above the per-model cap of
    the update kernels is fine (no update runs), a busy engine refuses."""
    X = inputs.tensor(modes, 2)
    rng = np.random.default_rng(5)
    facs = [np.asfortranarray(rng.uniform(-1, 1, (I, rank))) for I in modes]
    e = cc.Engine(modes, rank, dtype=dtype)
    e.set_tensor(X)
    for n in range(len(modes)):
        G, ms = e.mttkrp(facs, n)
        assert ms > 0.0
        want = oracle.mttkrp(X, modes, facs, n, oracle.MTTKRP)
        assert rel(G, want) < (TOL_KERNEL if dtype == "f64" else 2e-5)
    for n in range(len(modes)):
        assert not e.debug_factor(n).any() if e.active_cols else True
    if rank <= 64 and dtype == "f64":
        ranks = [rank]
        base = make_models(inputs, modes, ranks, seed=3)
        runs = []
        for eng in (e, cc.Engine(modes, rank)):
            if eng is not e:
                eng.set_tensor(X)
            eng.set_params(cc.default_params(max_iterations=4, force_max_iter=1))
            m = cc.Model([f.copy() for f in base[0][0]], base[0][1].copy())
            eng.enqueue(m)
            eng.run()
            runs.append(m)
            if eng is not e:
                eng.close()
        for a, b in zip(runs[0].factors, runs[1].factors):
            assert np.array_equal(a, b)
        m2 = cc.Model([f.copy() for f in base[0][0]], base[0][1].copy())
        e.enqueue(m2)
        e.admit()
        with pytest.raises(cc.CalsHipError):
            e.mttkrp(facs, 0)
    e.close()


def test_mttkrp_golden(cc, inputs):
    gold = np.load(os.path.join(GOLD, "mttkrp.npz"))
    for modes, ranks in (([7, 5, 3], [1, 2, 3]), ([3, 3, 3, 3], [7, 2]), ([40, 30, 20], [20] * 7)):
        X = inputs.tensor(modes, 0)
        e, _, _ = engine_with(cc, inputs, modes, ranks, X)
        e.admit()
        tag = "x".join(str(m) for m in modes)
        for n in range(len(modes)):
            assert rel(e.debug_mttkrp(n), gold["%s_mode%d" % (tag, n)]) < TOL_KERNEL
        e.close()


def test_tensor_norms(cc, oracle, inputs):
    modes = [23, 7, 11]
    X = inputs.tensor(modes, 5)
    e = cc.Engine(modes, 8)
    e.set_tensor(X)
    xn, jk = e.debug_norms()
    assert abs(xn - np.linalg.norm(X)) / np.linalg.norm(X) < 1e-14
    assert rel(jk, oracle.jk_norms(X, modes)) < 1e-14
    e.close()


@pytest.mark.parametrize("ranks", [[1], [2], [5], [20], [32], [1, 20, 7, 32, 13], [33], [48], [64], [3, 40, 64, 17]])
def test_single_sweep_state_vs_oracle(cc, oracle, inputs, ranks):
    """One sweep: factors, lambda, Gramians, error, fit of every model (update kernel unit test
    through the public path; ranks cover every register class of the kernel and, above 32, the
    big-rank body; 64 = the limit)."""
    modes = [40, 36, 33]
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
            assert rel(F, m.factors[n]) < TOL_KERNEL
            Gm = e.debug_gramian(n)[:r, col:col + r]
            assert rel(Gm, m.factors[n].T @ m.factors[n]) < TOL_KERNEL
        assert rel(lam[col:col + r], m.lam) < TOL_KERNEL
        st, c = e.debug_status(g)
        assert c == col and st.iters == 3  # admitted at 1, two sweeps survived
        assert abs(st.approx_error - m.error) <= 1e-10 * max(1.0, m.error)
        assert abs(st.fit - m.fit) <= 1e-12
        col += r
    e.close()


def test_update_golden(cc, inputs):
    """update kernel against the committed unit vectors is covered through whole runs below; here:
    the c1 golden case at 1, 2, 3, 10, 50 forced iterations (BASELINE config 1)."""
    gold = np.load(os.path.join(GOLD, "c1_20cube.npz"))
    modes, ranks = [20, 20, 20], [2, 3, 4, 5]
    X = inputs.tensor(modes, 0)
    for it in (1, 2, 3, 10, 50):
        prm = cc.default_params(max_iterations=it, force_max_iter=1)
        e, gm, _ = engine_with(cc, inputs, modes, ranks, X, params=prm)
        rep = e.run()
        assert rep.iter == gold["it%d_sweeps" % it][0]
        assert [m.iters for m in gm] == list(gold["it%d_iters" % it])
        for n in range(3):
            assert rel(np.hstack([m.factors[n] for m in gm]), gold["it%d_factor%d" % (it, n)]) < TOL_RUN
        assert rel(np.concatenate([m.lam for m in gm]), gold["it%d_lambda" % it]) < TOL_RUN
        assert np.allclose([m.error for m in gm], gold["it%d_error" % it], rtol=1e-9, atol=1e-12)
        e.close()


def _run_both(cc, oracle, inputs, modes, ranks, X, iters, jk=None, buffer=None, **kw):
    force = kw.pop("force_max_iter", 1)
    prm = cc.default_params(max_iterations=iters, force_max_iter=force, **kw)
    e, gm, base = engine_with(cc, inputs, modes, ranks, X, jk=jk, buffer=buffer, params=prm)
    rep = e.run()
    for m in gm:
        m.ls_margin = e.ls_margin(m)  # how close to a tie the model's accept / revert tests were
    e.close()
    om = [oracle.Model(fs, lam, jk=j) for fs, lam, j in base]
    po = oracle.default_params(max_iterations=iters, force_max_iter=force, mttkrp_method=oracle.MTTKRP,
                               buffer_size=sum(ranks) if buffer is None else buffer, **kw)
    ro = oracle.cp_cals(X, modes, om, po)
    return gm, om, rep, ro


def _assert_models_match(gm, om, xnorm2, tol=TOL_RUN):
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < tol
        assert rel(a.lam, b.lam) < tol
        if b.error < 1e300:
            # the error is sqrt(max(||X||^2 + t2 - 2 t3, 0)) (error.cpp:86-87): compare the squares,
            # whose rounding noise is ~1e-16 ||X||^2 (the sqrt of a cancelled difference is not
            # relatively accurate for a converged model)
            assert abs(a.error ** 2 - b.error ** 2) <= 1e-10 * xnorm2
            assert abs(a.fit - b.fit) <= 1e-5


@pytest.mark.parametrize("modes,ranks,iters", [
    ([20, 20, 20], [2, 3, 4, 5], 50),
    ([13, 12, 11], list(range(1, 13)) * 3, 30),
    ([6, 5, 4, 3], [3, 4, 5], 10),
    ([50, 40, 30], None, 10),
    ([40, 36, 33], [33, 64, 48, 5, 20], 10),   # ranks above 32: the big-rank update body
])
def test_forced_iterations_vs_oracle(cc, oracle, inputs, modes, ranks, iters):
    if ranks is None:
        ranks = inputs.ranks_1_to_20(40)
    X = inputs.tensor(modes, 3)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, iters)
    assert rep.iter == ro.iter == iters
    assert (rep.n_ktensors, rep.ktensor_comp_sum) == (ro.n_ktensors, ro.ktensor_comp_sum)
    _assert_models_match(gm, om, ro.X_norm ** 2)


def test_jackknife_models_vs_oracle_and_golden(cc, oracle, inputs):
    modes, comp = [20, 9, 12], 5
    X = inputs.low_rank_tensor(modes, comp, seed=21)[0]
    jk = [(0, i) for i in range(20)]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, [comp] * 20, X, 20, jk=jk)
    _assert_models_match(gm, om, ro.X_norm ** 2)
    for i, m in enumerate(gm):
        assert np.all(m.factors[0][i, :] == 0.0)  # the jk fiber stays exactly zero


def test_line_search_vs_oracle(cc, oracle, inputs):
    modes, ranks = [20, 20, 20], [2, 3, 4, 5, 20, 17]
    X = inputs.tensor(modes, 3)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 25, line_search=1,
                                line_search_interval=5)
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    assert rep.ls_performed > 0
    _assert_models_match(gm, om, ro.X_norm ** 2)
    _assert_margins_match(gm, om)


def _assert_margins_match(gm, om):
    """cals_hip_debug_ls_margin against the oracle's own record (or_model::ls_margin): the smallest relative distance
    between the two errors of a model's accept / revert tests.  The errors are sqrt of a cancelled sum, so the distance
    itself carries their rounding noise: compared absolutely at 1e-9 (the tie threshold the tolerant tests use)."""
    tested = 0
    for a, b in zip(gm, om):
        if b.ls_margin >= 1e300:
            assert a.ls_margin >= 1e300  # the model went through no test on either side
            continue
        tested += 1
        assert 0.0 <= a.ls_margin <= 1.0 and abs(a.ls_margin - b.ls_margin) <= 1e-9, (a.ls_margin, b.ls_margin)
    return tested


def test_no_line_search_no_margin(cc, oracle, inputs):
    modes, ranks = [13, 12, 11], [2, 5]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, inputs.tensor(modes, 3), 6)
    assert all(m.ls_margin == 1e300 for m in gm) and all(m.ls_margin == 1e300 for m in om)


@pytest.mark.parametrize("method,interval,noise", [(1, 5, 0.1), (2, 3, 0.05), (1, 2, 0.3)])
def test_error_checking_line_search_vs_oracle(cc, oracle, inputs, method, interval, noise):
    """ls::ERROR_CHECKING_SERIAL / _PARALLEL (line_search.cpp:86-153): the extrapolation is evaluated
    and kept only if its error is lower.  Same accept / reject decisions, iteration counts and
    factors as the oracle (whose candidate error is the reconstruction's; the device's comes from an
    MTTKRP of the candidate)."""
    modes = [22, 19, 17]
    ranks = [2, 3, 4, 5, 9, 14, 1, 20]
    X = inputs.low_rank_tensor(modes, 6, seed=31)[0] + noise * inputs.tensor(modes, 8)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 40, force_max_iter=0, tol=1e-7,
                                line_search=1, line_search_interval=interval, line_search_method=method)
    assert rep.iter == ro.iter
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    if method == 1:
        assert rep.ls_performed > 0 and 0 < rep.ls_failed < rep.ls_performed  # both outcomes occur
    else:
        assert rep.ls_performed == 0  # the reference never dispatches ERROR_CHECKING_PARALLEL
    _assert_models_match(gm, om, ro.X_norm ** 2)
    assert (_assert_margins_match(gm, om) > 0) == (method == 1)


@pytest.mark.parametrize("ls", [0, 1])
def test_queue_eviction_compress_vs_oracle(cc, oracle, inputs, ls):
    """buffer_size < sum of ranks, tol-based convergence: admission order, eviction sweep, compress
    (restated SimpleCorrectness / LineSearchCorrectness, tests/cals/test_cals.cpp:13-179)."""
    modes = [13, 12, 11]
    ranks = [r for r in range(1, 13) for _ in range(5)]
    X = inputs.low_rank_tensor(modes, 10, seed=9)[0]
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 200, buffer=30, tol=1e-5,
                                force_max_iter=0, line_search=ls, line_search_interval=10)
    assert rep.iter == ro.iter
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        # reference criterion: reconstructed tensors agree (MODEL_DIFF_ACC, test_cals.cpp:7,81-84)
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-9


def test_rejects_bad_input_loudly(cc, inputs):
    e = cc.Engine([8, 8, 8], 16)
    fs = [np.zeros((8, 65), order="F")] * 3
    with pytest.raises(cc.CalsHipError):
        e.enqueue(cc.Model(fs, np.ones(65)))         # rank > CALS_HIP_MAX_RANK / buffer
    with pytest.raises(cc.CalsHipError):
        e.sweep(1)                                     # no tensor yet
    with pytest.raises(cc.CalsHipError):
        e.set_params(cc.default_params(line_search=1, line_search_method=3))
    e.close()


@pytest.mark.parametrize("arg", ["", "ls"])
def test_cpp_api_cp_cals(arg):
    """cals::cp_cals (C++ mirror of the reference boundary) == oracle, restated SimpleCorrectness /
    LineSearchCorrectness; the binary is built by __graft_entry__.build()."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_cals_api")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe] + ([arg] if arg else []), capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cpp_api_jk_cp_cals():
    """cals::jk_cp_cals (generate replicas, one cp_cals call, re-normalise, LSAP column matching) ==
    the oracle's restatement (reference FunctionCorrectness, tests/cals/test_cals.cpp:299-362)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_jk_api")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("extra", [[], ["-p", "f32"]])
def test_cli_driver_runs(extra):
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cp-cals_amd", "examples", "driver")
    import subprocess
    r = subprocess.run([exe, "-t", "30-25-20", "-c", "1:4:3"] + extra, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    # two separate assertions: a crash at process exit (seen once: -11 after all output)
    # must be told apart from a run that did not finish
    assert "Speedup:" in r.stdout, "driver did not finish: rc=%s stderr=%s" % (r.returncode, r.stderr[-400:])
    # the driver installs crash_trace: a fatal signal, also one during process teardown, leaves its
    # backtrace on stderr
    assert r.returncode == 0, "driver finished but exited with rc=%s stderr=%s" % (r.returncode, r.stderr[-4000:])


def test_cli_driver_reads_tensor_file(tmp_path):
    """driver -f FILE: Tensor(file_name) (src/tensor.cpp:35-65: first line = mode sizes separated by
    blanks, then one value per line, mode 0 fastest) feeding cp_cals."""
    import subprocess
    modes = [9, 8, 7]
    rng = np.random.default_rng(3)
    vals = rng.uniform(-1, 1, int(np.prod(modes)))
    path = tmp_path / "tensor.txt"
    path.write_text(" ".join(map(str, modes)) + "\n" + "\n".join(repr(float(v)) for v in vals) + "\n")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cp-cals_amd", "examples", "driver")
    r = subprocess.run([exe, "-f", str(path), "-c", "1:3:2"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert "Tensor read from" in r.stdout and "9-8-7" in r.stdout and "Speedup:" in r.stdout
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "-f", str(tmp_path / "missing.txt")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "cannot open" in r.stderr


# ---- full-size properties (BASELINE configs 2 and 3): no oracle run, exact identities instead ----
def test_full_size_c3_mttkrp_identities(cc, inputs):
    """300^3 fp64, 256 models ranks 1..20 (R = 2656): (1) columns of a model that is all-ones give
    the mode sums of X; (2) linearity: MTTKRP of a doubled factor doubles G exactly."""
    modes = [300, 300, 300]
    X = inputs.tensor(modes, 0)
    ranks = inputs.ranks_1_to_20(256)
    base = make_models(inputs, modes, ranks)
    base[0][0][0][:] = 1.0
    base[0][0][1][:] = 1.0
    base[0][0][2][:] = 1.0
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    models = [cc.Model(fs, lam) for fs, lam, _ in base]
    for m in models:
        e.enqueue(m)
    e.admit()
    X3 = X.reshape(modes, order="F")
    G0 = e.debug_mttkrp(0)
    G1 = e.debug_mttkrp(1)
    G2 = e.debug_mttkrp(2)
    assert rel(G0[:, 0], X3.sum(axis=(1, 2))) < 1e-12
    assert rel(G1[:, 0], X3.sum(axis=(0, 2))) < 1e-12
    assert rel(G2[:, 0], X3.sum(axis=(0, 1))) < 1e-12
    # spot-check 3 random columns against the einsum formulation
    rng = np.random.default_rng(0)
    facs = [np.hstack([fs[n] for fs, _, _ in base]) for n in range(3)]
    for c in rng.integers(1, sum(ranks), size=3):
        want = np.einsum("ijk,j,k->i", X3, facs[1][:, c], facs[2][:, c])
        assert rel(G0[:, c], want) < 1e-11
        want = np.einsum("ijk,i,j->k", X3, facs[0][:, c], facs[1][:, c])
        assert rel(G2[:, c], want) < 1e-11
    e.close()


def test_full_size_c2_fast_error_is_true_error(cc, inputs):
    """100^3, 64 models ranks 1..20, 5 sweeps: the device's fast error equals the brute-force
    ||X - reconstruction|| (reference ComputeCorrectError, test_als.cpp:125-145)."""
    modes = [100, 100, 100]
    X = inputs.tensor(modes, 0)
    ranks = inputs.ranks_1_to_20(64)
    prm = cc.default_params(max_iterations=5, force_max_iter=1)
    e, gm, _ = engine_with(cc, inputs, modes, ranks, X, params=prm)
    e.run()
    e.close()
    for m in gm[::7]:
        slow = np.linalg.norm(X - reconstruct(m.factors, m.lam, modes))
        assert abs(m.error - slow) <= 1e-9 * slow


# ---- edges of the input domain ----
@pytest.mark.parametrize("modes,ranks", [
    ([3, 2, 3, 2, 2, 3, 2, 2], [2, 3]),      # 8-way: CALS_HIP_MAX_MODES
    ([7, 1, 5], [2, 1]),                      # a mode of size 1
    ([1, 1, 9], [1]),                         # a fibre
    ([17, 16, 15], [17 * 2]),                 # rank above every mode size (singular Hadamard of Gramians is possible)
])
def test_edge_shapes_vs_oracle(cc, oracle, inputs, modes, ranks):
    X = inputs.tensor(modes, 9)
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 4)
    assert rep.iter == ro.iter == 4
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        # the rank-deficient cases are compared through the fitted tensor: individual columns of a singular system
        # are not determined
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-7 * max(1.0, np.linalg.norm(X))


def test_empty_queue_and_exact_fit_of_the_buffer(cc, oracle, inputs):
    modes = [9, 8, 7]
    X = inputs.tensor(modes, 2)
    e = cc.Engine(modes, 12)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=5, force_max_iter=1))
    rep = e.run()                                      # nothing queued: returns at once
    assert (rep.iter, rep.n_ktensors) == (0, 0)
    e.close()
    # models whose ranks fill the buffer to the last column, then a queue that needs it twice over
    gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, [5, 4, 3, 12, 6, 6], X, 5, buffer=12)
    assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
    _assert_models_match(gm, om, ro.X_norm ** 2)


def test_a_singular_model_does_not_disturb_its_neighbours(cc, oracle, inputs):
    """Models whose columns are all equal have a singular Hadamard of Gramians: dpotrf reports info > 0 (the
    reference only logs, update.cpp:183-185) or factors a numerically meaningless matrix.  Whatever such a model
    ends up holding, the run terminates and the healthy models in the same buffer -- one per update body: register /
    LDS, blocked in LDS, blocked through L2 -- are exactly what they are without it."""
    modes = [30, 25, 20]
    X = inputs.tensor(modes, 4)
    good_ranks = [5, 12, 40, 70]
    base = make_models(inputs, modes, good_ranks, seed=3)
    bad = []
    for r in (20, 40, 70):
        (fs, lam), = inputs.model_factors(modes, [1], seed=50 + r)
        bad.append(([np.asfortranarray(np.repeat(f, r, axis=1)) for f in fs], np.ones(r)))
    e = cc.Engine(modes, sum(good_ranks) + 130)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=4, force_max_iter=1))
    gm = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
    bm = [cc.Model(fs, lam) for fs, lam in bad]
    for m in (gm[0], bm[0], gm[1], bm[1], gm[2], bm[2], gm[3]):
        e.enqueue(m)
    with np.errstate(all="ignore"):
        rep = e.run()
    e.close()
    assert rep.iter == 4 and rep.n_ktensors == 7
    om = [oracle.Model(fs, lam) for fs, lam, _ in base]
    ro = oracle.cp_cals(X, modes, om, oracle.default_params(max_iterations=4, force_max_iter=1,
                                                            mttkrp_method=oracle.MTTKRP, buffer_size=sum(good_ranks)))
    _assert_models_match(gm, om, ro.X_norm ** 2)
