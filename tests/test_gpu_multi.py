"""Work-queue hand-off between ranks (cp-cals_amd/multi_gpu.py) end to end: two processes, one engine
each (both on cuda:0 -- the box has one GPU; on a node each rank takes its own), gloo for the
control plane.  Every model, whichever rank fitted it, equals the oracle's single-model ALS."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import reconstruct

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = [20, 18, 16]
N_MODELS = 36


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ranks():
    return [1 + (k * 5) % 7 for k in range(N_MODELS)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import cp_cals_amd as cc
    from cp_cals_amd import inputs, multi_gpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = inputs.low_rank_tensor(MODES, 4, seed=11)[0] + 0.05 * inputs.tensor(MODES, 3)
    base = inputs.model_factors(MODES, _ranks(), 5)
    eng = cc.Engine(MODES, 16, device=0)
    eng.set_tensor(X)
    eng.set_params(cc.default_params(max_iterations=80, tol=1e-6, line_search=1, line_search_interval=4))

    def make_model(k):
        fs, lam = base[k]
        return cc.Model([f.copy() for f in fs], lam.copy())

    mine, stats = multi_gpu.cp_cals_work_queue(eng, N_MODELS, make_model, claim_models=3)
    eng.close()
    allres = multi_gpu.gather_results(mine)
    q.put((rank, sorted(mine), stats, allres if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_two_ranks_equal_single_model_als(oracle, inputs):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owned = [r[1] for r in res]
    assert sorted(owned[0] + owned[1]) == list(range(N_MODELS))
    assert owned[0] and owned[1]                    # both ranks got work
    allres = res[0][3]
    X = inputs.low_rank_tensor(MODES, 4, seed=11)[0] + 0.05 * inputs.tensor(MODES, 3)
    base = inputs.model_factors(MODES, _ranks(), 5)
    for k in range(N_MODELS):
        fs, lam = base[k]
        m = oracle.Model([f.copy() for f in fs], lam.copy())
        oracle.cp_als(X, MODES, m, oracle.default_params(max_iterations=80, tol=1e-6, line_search=1,
                                                        line_search_interval=4, mttkrp_method=oracle.MTTKRP))
        gf, gl, it, err, fit = allres[k]
        assert it == m.iters
        d = np.linalg.norm(reconstruct(gf, gl, MODES) - reconstruct(m.factors, m.lam, MODES))
        assert d <= 1e-8 * np.linalg.norm(X)


def _rccl_single_rank_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    parts = [torch.zeros_like(t)]
    dist.all_gather(parts, t)
    dist.barrier()
    q.put((dist.get_backend(), dist.get_world_size(), float(t.item()), float(parts[0].item())))
    dist.destroy_process_group()


def test_rccl_collectives_of_the_bench_load_and_run_on_this_box():
    """bench.py's control plane for N > 1 is RCCL (backend "nccl"): barrier, all_reduce(MAX / SUM) and
    all_gather of one float64.  A one-GPU box cannot host two RCCL ranks ("Duplicate GPU detected"), but a
    single-rank group runs the same library calls on the same dtypes -- enough to know the backend is usable
    before the driver's 8-GPU run.  (The N > 1 logic itself: tests/test_sharding_gloo.py, and the two-rank
    gloo rehearsal on this GPU above.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    backend, world, red, gathered = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert (backend, world, red, gathered) == ("nccl", 1, 3.5, 3.5)
