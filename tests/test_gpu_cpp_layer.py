"""GPU tests of what sits directly above / beside the C ABI for the reference's callers:
the reference-style C++ caller (same includes / symbols / call pattern as the reference's driver, MEX
glue and CSV writers) running against the header set, cals_hip_rebind (the Tensor's device mirror
reused across cp_cals calls, include/tensor.h:56-59) and the per-sweep log behind CalsReport's timer
matrices (include/cals.h:55-63)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_caller_runs(tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "ref_style_caller")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    csv = tmp_path / "report.csv"
    r = subprocess.run([exe, "14-11-9", "1:4:2", str(csv)], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stderr[-2000:]
    lines = csv.read_text().strip().splitlines()
    head = lines[0].split(";")
    assert head[:3] == ["TENSOR_RANK", "TENSOR_MODES", "BUFFER_SIZE"] and "MODE_2_UPDATE" in head
    assert len(lines) >= 3 and all(ln.split(";")[1] == "14-11-9" for ln in lines[1:])
    assert [int(ln.split(";")[8]) for ln in lines[1:]] == list(range(1, len(lines)))   # ITER column


@pytest.mark.parametrize("modes", ["9-8-7-6", "6-5-4-5-3"])
def test_reference_style_caller_on_n_way_tensors(modes):
    """the same caller on 4- and 5-way tensors: the C++ layer over the two-group dimension tree (cp_cals == cp_als
    per model, jackknife replicas in one call, the timer matrices with one MODE_n block per mode)"""
    exe = os.path.join(ROOT, "tests", "cpp", "ref_style_caller")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, modes, "1:3:2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stderr[-2000:]


def test_static_tensor_with_device_mirror_exits_cleanly():
    """The mirror's engine is destroyed during static destruction, after the HIP runtime's exit handlers:
    cals_hip_destroy must then leave HIP alone (drain_devices_at_exit sets the flag)."""
    exe = os.path.join(ROOT, "tests", "cpp", "static_tensor_exit")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-3000:])
    assert "fitted 4 models; mirror alive: 1" in r.stdout
    assert r.returncode == 0, r.stderr[-3000:]


def _models(cc, inputs, modes, ranks, seed):
    return [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam in inputs.model_factors(modes, ranks, seed)]


def test_rebind_reuses_the_tensor_copies(cc, inputs):
    modes = [33, 21, 18]
    X = inputs.tensor(modes, 4)
    prm = cc.default_params(max_iterations=9, force_max_iter=1, line_search=1, line_search_interval=3)
    ranks_a, ranks_b = [5, 3, 8, 2, 7, 6, 1, 4], [4, 6, 2]
    e = cc.Engine(modes, sum(ranks_a))
    e.set_tensor(X)
    e.set_params(prm)
    assert e.capacity == sum(ranks_a)
    ma = _models(cc, inputs, modes, ranks_a, 1)
    for m in ma:
        e.enqueue(m)
    with pytest.raises(cc.CalsHipError) as ei:       # not idle
        e.rebind(10)
    assert ei.value.code == cc.ERR_STATE
    e.run()
    with pytest.raises(cc.CalsHipError) as ei:       # wider than what was allocated
        e.rebind(sum(ranks_a) + 1)
    assert ei.value.code == cc.ERR_FULL
    # second "cp_cals call": other models, smaller buffer (forces queueing), same X copies
    e.rebind(8)
    mb = _models(cc, inputs, modes, ranks_b, 2)
    for m in mb:
        e.enqueue(m)
    rep = e.run()
    assert rep.n_ktensors == 3 and rep.ktensor_comp_sum == 12
    plan = e.tree
    e.close()
    f = cc.Engine(modes, 8)                          # the same call on a fresh engine
    f.set_tensor(X)
    f.set_params(prm)
    assert f.tree == plan
    mf = _models(cc, inputs, modes, ranks_b, 2)
    for m in mf:
        f.enqueue(m)
    rep_f = f.run()
    f.close()
    assert (rep.iter, rep.ls_performed, rep.ls_failed) == (rep_f.iter, rep_f.ls_performed, rep_f.ls_failed)
    for a, b in zip(mb, mf):
        assert a.iters == b.iters
        for fa, fb in zip(a.factors, b.factors):
            assert rel(fa, fb) < 1e-12
        assert rel(a.lam, b.lam) < 1e-12


def test_sweep_log_feeds_the_report_matrices(cc, inputs):
    modes = [40, 30, 20]
    ranks = inputs.ranks_1_to_20(30)
    R = sum(ranks)
    X = inputs.tensor(modes, 0)
    e = cc.Engine(modes, R)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=7, force_max_iter=1, line_search=1, line_search_interval=2))
    for m in _models(cc, inputs, modes, ranks, 1):
        e.enqueue(m)
    e.set_sweep_log(True)
    rep = e.run()
    log = e.sweep_log()
    e.set_sweep_log(False)
    assert rep.iter == 7 and len(log) == 7
    total = float(np.prod(modes))
    for k, rec in enumerate(log):
        assert rec.cols == R and rec.models == 30
        launches = rec.flops / (2.0 * total * R)   # MFMA kernels of the sweep (+ a fraction when stale
        assert 1 - 1e-9 <= launches <= 3 + 1e-9     # columns of a pending T were patched)
        assert rec.iteration_ms > 0 and rec.defrag_ms >= 0 and rec.ls_ms > 0
        dev = sum(rec.mttkrp_ms[n] + rec.update_ms[n] for n in range(3)) + rec.ls_ms
        assert 0 < dev <= rec.iteration_ms * 1.05
        for n in range(3):
            assert rec.update_ms[n] > 0 and rec.mttkrp_ms[n] > 0
            parts = rec.fused_ms[n] + rec.ttm_ms[n] + rec.contract_ms[n] + rec.krp_ms[n]
            assert abs(parts - rec.mttkrp_ms[n]) < 1e-9
    e.close()
