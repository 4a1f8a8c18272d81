"""The N>1 path on CPU: world_size-2 gloo processes run the sharding / aggregation logic that
bench.py uses around the per-GPU engines (no data-path collective exists in this path)."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import cp_cals_amd  # noqa: F401
    from cp_cals_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_models = sharding.weak_scaling_models(5, world) + 1  # 11: uneven on purpose
    mine = sharding.shard_round_robin(n_models, world, rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    sharding.barrier()
    rate, t = sharding.aggregate_rate(local_units=10 * (rank + 1), local_seconds=1.0 + rank)
    # config 5 (strong scaling): the job's 2048 models split m -> rank m mod N; one step = one sweep of
    # the whole job, so job rate = steps / slowest rank (bench.py --workload c5)
    c5 = sharding.shard_round_robin(2048, world, rank)
    srate, st = sharding.strong_rate(job_steps=50, local_seconds=2.0 + 0.5 * rank)
    per_rank = sharding.gather_over_ranks(2.0 + 0.5 * rank)
    q.put((rank, mine, gathered, rate, t, len(c5), c5[:3], srate, st, per_rank))
    dist.destroy_process_group()


def test_round_robin_shards_and_aggregation_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_models = sorted(sum((r[1] for r in res), []))
    assert all_models == list(range(11))            # a partition: nothing lost, nothing doubled
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    for r in res:
        assert r[2] == [res[0][1], res[1][1]]
        assert abs(r[3] - 30.0 / 2.0) < 1e-12          # (10 + 20) units / max(1 s, 2 s)
        assert abs(r[4] - 2.0) < 1e-12
        assert r[5] == 1024 and r[6] == [r[0], r[0] + 2, r[0] + 4]
        assert abs(r[7] - 50 / 2.5) < 1e-12 and abs(r[8] - 2.5) < 1e-12   # steps / max-over-ranks time
        assert r[9] == [2.0, 2.5]


def _queue_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import random
    import time
    import torch.distributed as dist
    import cp_cals_amd  # noqa: F401
    from cp_cals_amd import multi_gpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 1:  # a queue on a private store, on one rank only, must not shift the shared queues' key prefix
        assert multi_gpu.WorkQueue(3, store=dist.HashStore()).claim(5) == [0, 1, 2]
    wq = multi_gpu.WorkQueue(57)
    rnd = random.Random(rank)
    got = []
    while True:
        ks = wq.claim(rnd.randint(1, 5))
        if not ks:
            break
        got += ks
        time.sleep(rnd.random() * 0.002 * (1 + 3 * rank))  # rank 1 is the slow one
    dist.barrier()
    # a second queue with the default name in the same job starts at 0 again (own key prefix per instance)
    wq2 = multi_gpu.WorkQueue(9)
    got2 = []
    while True:
        ks = wq2.claim(2)
        if not ks:
            break
        got2 += ks
    dist.barrier()
    q.put((rank, got, got2))
    dist.destroy_process_group()


def test_work_queue_hands_out_every_index_once_world2():
    """The pull-based hand-off of cp-cals_amd/multi_gpu.py: two ranks claim from one counter at
    different speeds; together they get every index exactly once, the faster rank gets more."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    res = {r: a for r, a, _ in got}
    res2 = {r: b for r, _, b in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0] + res[1]) == list(range(57))
    assert len(res[0]) > len(res[1])
    assert sorted(res2[0] + res2[1]) == list(range(9))


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent (which has made no GPU call) starts the
    torch.distributed launcher as a child; here, without a GPU, BOTH ranks must reach the "needs a GPU" exit of
    bench.py and the parent must return the launcher's non-zero status -- not a usage message.  On a GPU box the
    same path goes on into the process group and the engines."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""        # also on a GPU box: this test exercises the launch, not the engines
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--force-device0", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0
    assert "launch with" not in (r.stdout + r.stderr)
    assert (r.stdout + r.stderr).count("bench.py needs a GPU") >= 1
    assert "nproc-per-node" not in r.stdout          # nothing but rank 0's JSON line ever goes to stdout
