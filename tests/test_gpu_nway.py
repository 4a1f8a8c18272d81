"""N > 3 modes: the two-group dimension tree (cals_hip_tree() == 4; GroupContractArgs in cals_hip_internal.h)
against the oracle's Khatri-Rao + GEMM MTTKRP (mttkrp.cpp:147-176, 218-328) and against the engine's own plain
path (CALS_HIP_TREE=0: one fused MTTKRP per mode).  Group sizes 2..4 on either side, the Khatri-Rao workspace of
a three- and four-mode outer group, ragged and size-1 modes, fp32 storage, both update methods, the line search,
jackknife models, a queue longer than the buffer."""
import os

import numpy as np
import pytest

from helpers import make_models, reconstruct, rel
from test_gpu_parity import TOL_KERNEL, _assert_models_match, _run_both, engine_with

pytestmark = pytest.mark.gpu

SHAPES = [
    ([6, 5, 4, 3], [3, 4, 5]),                 # 2 + 2
    ([17, 9, 20, 18], [5, 20, 1, 7]),          # rows above one 16-row tile on both sides
    ([40, 33, 6, 50], [20] * 8),               # R = 160: two column blocks; 1320 x 300 merged rows
    ([4, 3, 5, 2, 3], [2, 3]),                 # 5-way: 2 + 3 or 3 + 2
    ([7, 6, 5, 4, 3], [4, 9]),
    ([3, 4, 2, 3, 2, 4], [3, 2]),              # 6-way: 3 + 3
    ([2, 3, 2, 3, 2, 3, 2], [2, 3]),           # 7-way: 3 + 4
    ([3, 2, 3, 2, 2, 3, 2, 2], [2, 3]),        # 8-way: 4 + 4
    ([5, 1, 4, 3], [2, 2]),                    # a mode of size 1 inside a group
]


class plan:
    """CALS_HIP_TREE for the engines created inside the block"""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("CALS_HIP_TREE")
        if self.value is None:
            os.environ.pop("CALS_HIP_TREE", None)
        else:
            os.environ["CALS_HIP_TREE"] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("CALS_HIP_TREE", None)
        else:
            os.environ["CALS_HIP_TREE"] = self.old


@pytest.mark.parametrize("modes,ranks", SHAPES)
def test_mttkrp_through_the_group_tree_vs_oracle_and_plain_path(cc, oracle, inputs, modes, ranks):
    X = inputs.tensor(modes, 3)
    with plan(None):
        e, gm, base = engine_with(cc, inputs, modes, ranks, X)
    assert e.tree == 4
    e.admit()
    facs = [np.asfortranarray(np.hstack([fs[n] for fs, _, _ in base])) for n in range(len(modes))]
    for n in range(len(modes)):
        want = oracle.mttkrp(X, modes, facs, n, oracle.MTTKRP)
        assert rel(e.debug_mttkrp(n), want) < TOL_KERNEL            # the path a sweep takes
        assert rel(e.debug_mttkrp(n, "plain"), want) < TOL_KERNEL   # Khatri-Rao workspace + fused MTTKRP
    e.close()
    with plan("0"):
        e0, _, _ = engine_with(cc, inputs, modes, ranks, X)
    assert e0.tree == 0
    e0.close()


@pytest.mark.parametrize("modes,ranks", SHAPES)
def test_forced_sweeps_vs_oracle(cc, oracle, inputs, modes, ranks):
    X = inputs.tensor(modes, 4)
    with plan(None):
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 6)
    assert rep.iter == ro.iter == 6
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-8 * max(1.0, np.linalg.norm(X))


@pytest.mark.parametrize("kw", [
    dict(line_search=1, line_search_interval=3),
    dict(update_method=1),
])
def test_line_search_and_nnls_on_a_4_way_tensor(cc, oracle, inputs, kw):
    """(ls::ERROR_CHECKING_SERIAL is not compared for N > 3: the reference's error::compute_error rebuilds the
    tensor from factors 0, 1, 2 only, error.cpp:7-30, and reads past its workspace for a 4-way X -- there is no
    reference behaviour to match; the engine evaluates the candidate with the N-way error formula.)"""
    modes, ranks = [9, 8, 7, 6], [2, 5, 3, 8]
    X = inputs.low_rank_tensor(modes, 4, seed=11)[0] + 0.1 * inputs.tensor(modes, 5)
    if kw.get("update_method"):
        X = np.abs(X)
    with plan(None):
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 10, **kw)
    assert (rep.iter, rep.ls_performed, rep.ls_failed) == (ro.iter, ro.ls_performed, ro.ls_failed)
    _assert_models_match(gm, om, ro.X_norm ** 2, tol=1e-7)


def test_error_checking_line_search_is_refused_beyond_3_way(cc, oracle, inputs):
    """ls::ERROR_CHECKING_SERIAL + N > 3: error::compute_error (error.cpp:7-30) rebuilds a 3-way tensor and
    subtracts it from all elements of a 4-way X -- undefined behaviour in the reference.  Engine and oracle both
    refuse the combination instead of each inventing its own meaning (round 2 had two: 11 vs 4 reverts)."""
    modes, ranks = [6, 5, 4, 3], [2, 3]
    X = inputs.tensor(modes, 3)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    with pytest.raises(cc.CalsHipError):
        e.set_params(cc.default_params(line_search=1, line_search_method=1, line_search_interval=2))
    e.set_params(cc.default_params(line_search=1, line_search_method=0, line_search_interval=2))   # fine
    e.close()
    e3 = cc.Engine(modes[:3], sum(ranks))
    e3.set_params(cc.default_params(line_search=1, line_search_method=1, line_search_interval=2))   # 3-way: fine
    e3.close()
    base = inputs.model_factors(modes, ranks, seed=5)
    models = [oracle.Model([f.copy() for f in fs], lam.copy()) for fs, lam in base]
    p = oracle.default_params(max_iterations=4, force_max_iter=1, buffer_size=sum(ranks), line_search=1,
                              line_search_method=1, line_search_interval=2)
    with pytest.raises(Exception):
        oracle.cp_cals(X, modes, models, p)


def test_queue_jackknife_and_fp32_on_a_5_way_tensor(cc, oracle, inputs):
    modes = [6, 5, 4, 5, 3]
    ranks = [3, 1, 4, 2, 5, 2, 3, 1, 4, 2]
    # a rank-3 tensor plus noise (inputs.low_rank_tensor stops at 4 modes)
    (fs, lam), = inputs.model_factors(modes, [3], seed=77)
    X = reconstruct(fs, lam, modes) + 0.05 * inputs.tensor(modes, 8)
    jk = [((0, k % modes[0]) if k % 2 else None) for k in range(len(ranks))]
    with plan(None):
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 25, jk=jk, buffer=9, force_max_iter=0,
                                    tol=1e-4)
    assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-7 * max(1.0, np.linalg.norm(X))
    # fp32 storage: the tree against the plain path of the same engine type, stated tolerance 2e-3 after 4 sweeps
    base = make_models(inputs, modes, ranks[:4], seed=3)
    out = {}
    for tag, env in (("tree", None), ("plain", "0")):
        with plan(env):
            e = cc.Engine(modes, sum(ranks[:4]), dtype="f32")
        e.set_tensor(X)
        e.set_params(cc.default_params(max_iterations=4, force_max_iter=1))
        ms = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
        for m in ms:
            e.enqueue(m)
        e.run()
        e.close()
        out[tag] = ms
    for a, b in zip(out["tree"], out["plain"]):
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < 2e-3


def test_a_bigger_4_way_tensor_tree_equals_plain(cc, inputs):
    """60 x 50 x 40 x 30 (3.6e6 elements), 32 models: three sweeps under the tree and under the plain path agree
    to rounding (no oracle at this size in seconds)."""
    modes = [60, 50, 40, 30]
    ranks = inputs.ranks_1_to_20(32)
    X = inputs.tensor(modes, 1)
    base = make_models(inputs, modes, ranks, seed=9)
    res = {}
    for tag, env in (("tree", None), ("plain", "0")):
        with plan(env):
            e = cc.Engine(modes, sum(ranks))
        e.set_tensor(X)
        e.set_params(cc.default_params(max_iterations=3, force_max_iter=1))
        ms = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]
        for m in ms:
            e.enqueue(m)
        e.run()
        e.close()
        res[tag] = ms
    for a, b in zip(res["tree"], res["plain"]):
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < 1e-9
        assert abs(a.error - b.error) <= 1e-9 * max(1.0, b.error)


def _random_nway_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nm = int(rng.integers(4, 8))
        cap = {4: 14, 5: 8, 6: 6, 7: 4}[nm]
        modes = [int(v) for v in rng.integers(1 if rng.integers(0, 6) == 0 else 2, cap + 1, size=nm)]
        n_models = int(rng.integers(2, 8))
        pr = sorted(modes)
        # keep the Hadamard of the N - 1 Gramians well conditioned (a failed dpotrf leaves nothing to compare):
        # rank well below the product of the three smallest modes
        rmax = max(1, min(24, (pr[0] * pr[1] * pr[2]) // 4))
        ranks = [int(v) for v in rng.integers(1, rmax + 1, size=n_models)]
        buffer = int(rng.integers(max(ranks), max(max(ranks) + 1, sum(ranks))))
        out.append((modes, ranks, buffer, int(rng.integers(0, 3)), int(rng.integers(0, 1 << 30))))
    return out


_N_NWAY = int(os.environ.get("CALS_SOAK_NWAY", "16"))


@pytest.mark.parametrize("modes,ranks,buffer,flavour,seed",
                         _random_nway_cases(_N_NWAY, 31337 + int(os.environ.get("CALS_SOAK_SEED", "0"))))
def test_random_n_way_queue_life_cycles(cc, oracle, inputs, modes, ranks, buffer, flavour, seed):
    """random 4- to 7-way shapes (size-1 modes included), a queue longer than the buffer, tolerance-driven eviction,
    flavour 0 plain / 1 line search / 2 NNLS: the same admission order, sweep counts and fitted tensors as the
    oracle."""
    X = inputs.tensor(modes, seed % 1000)
    kw = dict(tol=1e-3, force_max_iter=0)
    if flavour == 1:
        kw.update(line_search=1, line_search_interval=3)
    elif flavour == 2:
        kw["update_method"] = 1
        X = np.abs(X)
    with plan(None):
        gm, om, rep, ro = _run_both(cc, oracle, inputs, modes, ranks, X, 15, buffer=buffer, **kw)
    assert (rep.iter, rep.n_ktensors, rep.ktensor_comp_sum) == (ro.iter, ro.n_ktensors, ro.ktensor_comp_sum)
    assert (rep.ls_performed, rep.ls_failed) == (ro.ls_performed, ro.ls_failed)
    for a, b in zip(gm, om):
        assert a.iters == b.iters
        d = np.linalg.norm(reconstruct(a.factors, a.lam, modes) - reconstruct(b.factors, b.lam, modes))
        assert d <= 1e-7 * max(1.0, np.linalg.norm(X))
