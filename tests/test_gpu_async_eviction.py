"""The asynchronous eviction copy-out (evicted models are scattered into the callers' storage while the
next sweep runs) through the step API, and engine reuse across runs: results must be complete whenever
the API says so (cals_hip_model_result / cals_hip_synchronize / the end of cals_hip_run)."""
import numpy as np
import pytest

from helpers import make_models, rel

pytestmark = pytest.mark.gpu


def _models(cc, inputs, modes, ranks, seed):
    base = make_models(inputs, modes, ranks, seed=seed)
    return base, [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam, _ in base]


def test_step_api_results_complete_when_reported(cc, oracle, inputs):
    modes, ranks = [17, 14, 12], [3, 5, 2, 7, 4, 6, 1, 8, 3, 5, 2, 4]
    X = inputs.low_rank_tensor(modes, 4, seed=5)[0] + 0.05 * inputs.tensor(modes, 6)
    kw = dict(max_iterations=40, tol=1e-4)
    base, gm = _models(cc, inputs, modes, ranks, 11)
    e = cc.Engine(modes, 20)
    e.set_tensor(X)
    e.set_params(cc.default_params(**kw))
    for m in gm:
        e.enqueue(m)
    om = [oracle.Model(fs, lam) for fs, lam, _ in base]
    oracle.cp_cals(X, modes, om, oracle.default_params(mttkrp_method=oracle.MTTKRP, buffer_size=20, **kw))
    done = set()
    guard = 0
    while e.queue_size or e.models_in_flight:
        e.step()
        guard += 1
        assert guard < 1000
        # poll right after the step: whatever result() reports as evicted must already be final
        for k, m in enumerate(gm):
            if k in done:
                continue
            st = e.result(m)
            if st.evicted:
                done.add(k)
                assert m.iters == om[k].iters
                for fa, fb in zip(m.factors, om[k].factors):
                    assert rel(fa, fb) < 1e-8
                assert rel(m.lam, om[k].lam) < 1e-8
    for k, m in enumerate(gm):
        st = e.result(m)
        assert st.evicted
        for fa, fb in zip(m.factors, om[k].factors):
            assert rel(fa, fb) < 1e-8
    e.close()


def test_engine_reuse_across_runs(cc, oracle, inputs):
    """Two queues through the same engine, one after the other: staging arena, scratch and the pending
    copy-out state carry over."""
    modes = [15, 13, 11]
    X = inputs.low_rank_tensor(modes, 4, seed=8)[0] + 0.05 * inputs.tensor(modes, 9)
    kw = dict(max_iterations=25, tol=1e-4, line_search=1, line_search_interval=4)
    e = cc.Engine(modes, 24)
    e.set_tensor(X)
    e.set_params(cc.default_params(**kw))
    for rnd, ranks in enumerate(([2, 3, 4, 5, 6, 7, 8, 1], [8, 8, 8, 1, 1, 2, 5, 6, 7, 3, 3])):
        base, gm = _models(cc, inputs, modes, ranks, 20 + rnd)
        for m in gm:
            e.enqueue(m)
        rep = e.run()
        om = [oracle.Model(fs, lam) for fs, lam, _ in base]
        ro = oracle.cp_cals(X, modes, om, oracle.default_params(mttkrp_method=oracle.MTTKRP, buffer_size=24, **kw))
        assert rep.iter == ro.iter and rep.n_ktensors == len(ranks)
        for a, b in zip(gm, om):
            assert a.iters == b.iters
            for fa, fb in zip(a.factors, b.factors):
                assert rel(fa, fb) < 1e-8
        assert e.models_in_flight == 0 and e.queue_size == 0
    e.close()


def test_close_with_copy_out_pending(cc, inputs):
    """Destroying an engine right after a step that evicted models must not touch anything any more."""
    modes, ranks = [12, 11, 10], [2, 3, 4]
    X = inputs.tensor(modes, 1)
    base, gm = _models(cc, inputs, modes, ranks, 3)
    e = cc.Engine(modes, sum(ranks))
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=1, force_max_iter=1))
    for m in gm:
        e.enqueue(m)
    _, evicted = e.step()
    assert evicted == len(ranks)
    e.close()
