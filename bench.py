#!/usr/bin/env python3
"""bench.py -- the CALS hot path on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or
  started plainly -- WORLD_SIZE unset -- in which case this process, before any GPU call, starts exactly that
  launcher as a child, relays rank 0's line and exits with the child's status)

A step = one ALS sweep (every mode: MTTKRP + batched update; error; line search) of all models in
flight on a GPU.  Workload (BASELINE.json config 3, the one north_star's target is quoted on):
300x300x300 fp64 tensor, 256 concurrent models of rank 1 + (k mod 20) per GPU (R = 2656 columns), line
search on (NO_ERROR_CHECKING, interval 5, step cbrt(iter)); synthetic inputs from
cp-cals_amd/inputs.py.  X, factors and all model state are resident in HBM before the timed region.
N > 1: weak scaling, every rank owns its own 256-model shard (model m -> GPU m mod N), X replicated,
no data-path collective; value = sweeps completed by all ranks / max-over-ranks time.
--workload c5 = BASELINE config 5: 2048 jackknife models IN TOTAL (jk = (mode 0, fiber m mod 300)),
model m -> GPU m mod N, strong scaling (a step = one sweep of the whole job).  The default (c3) run ALSO
times that strong-scaling job right after the weak one and reports it under `strong_c5` (its own K steps,
own barrier-bracketed region, per-rank ms/step): one command per N yields both of north_star's numbers.
Rank 0 prints ONE JSON line: the contract fields, `roofline`, `cpu_baseline`, and beside them
`steady_state` (>= 200 further sweeps), `run_loop` (the cals_hip_step loop users run: status
read-back + eviction decision every sweep) and `dist` (backend, world size seen by the process
group, per-rank ms/step).
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (modes, models per GPU, line search)
    "c3": ([300, 300, 300], 256, 1),   # BASELINE config 3 (default)
    "c2": ([100, 100, 100], 64, 0),    # BASELINE config 2
    "c1": ([20, 20, 20], 4, 0),        # BASELINE config 1 (plumbing)
    "c4": ([299, 301, 41], 512, 0),    # BASELINE config 4: eemdata-shaped, fp32 storage + fp32 MFMA
    "c4f64": ([299, 301, 41], 512, 0),  # config 4's shape and model count, in fp64
    # BASELINE config 5: 2048 jackknife models IN TOTAL sharded round-robin over the N GPUs (strong
    # scaling: a step = one sweep of all 2048 models; value = steps / max time).  N = 1 gives the
    # denominator of north_star's ">= 6x at 8 GPUs".
    "c5": ([300, 300, 300], 2048, 1),
}
STRONG = {"c5"}
JACKKNIFE = {"c5"}     # every model carries jk = (mode 0, fiber m mod I0)  
WORKLOAD_DTYPE = {"c4": "f32"}
PEAK_FP32_MFMA_TFLOPS = 157.3  # dense FP32 matrix peak (v_mfma_f32_16x16x4_f32: 256 flop/cycle/CU x4 SIMD)
PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (datasheet; 256 CU x 4 SIMD x 2.4 GHz
#                               x 2048 flop / 64 cycles).  tools/mfma_f64_peak measures 77.7 on the box.


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(modes, ranks, X, base, jk, ls, threads_all, protocol):
    """The oracle (CPU restatement of the reference algorithm) with the image's MKL runtime for the big
    GEMMs, on the GPU box's host cores -- SURVEY.md section 8(d), protocol of the reference's
    include/experiments/bench_mttkrp_cals.h:49-84: MTTKRP variants {MTTKRP, TWOSTEP0, TWOSTEP1} each
    timed after one warm-up sweep, best of `reps` repetitions of `sweeps` forced sweeps, best variant
    reported, for threads in {all of the cgroup's share, 1}.  protocol "bounded" = 1 repetition of 3 sweeps at all
    threads, the best variant only (1 sweep) at 1 thread; protocol "middle" (default; about two minutes) = best of
    2 x 5 sweeps per variant at all threads to find the best variant, THAT variant then timed as section 8(d) says --
    best of 3 x 10 sweeps -- and 1 x 2 sweeps at 1 thread; protocol "full" = 3 x 10 sweeps for every variant at all
    threads, every variant 1 x 2 sweeps at 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    plan = {"bounded": {"all": (1, 3), "one": (1, 1)}, "middle": {"all": (2, 5), "one": (1, 2)},
            "full": {"all": (3, 10), "one": (1, 2)}}[protocol]
    have_mkl = O.use_mkl(threads_all)
    variants = (("MTTKRP", O.MTTKRP), ("TWOSTEP0", O.TWOSTEP0), ("TWOSTEP1", O.TWOSTEP1))

    def run(method, threads, sweeps):
        models = [O.Model([f.copy() for f in fs], lam.copy(), jk=None if jk is None else jk[k])
                  for k, (fs, lam) in enumerate(base)]
        p = O.default_params(max_iterations=sweeps, force_max_iter=1, buffer_size=sum(ranks),
                             mttkrp_method=method, line_search=ls, line_search_interval=5, threads=threads)
        rep = O.cp_cals(X, modes, models, p)
        return rep.iter / rep.loop_time

    def measure(threads, reps, sweeps, only=None):
        O.use_mkl(threads)
        O.set_threads(threads)
        table = {}
        for name, method in variants:
            if only is not None and name != only:
                continue
            run(method, threads, 1)  # warm-up sweep (thread pools, page faults of the KRP workspace)
            table[name] = max(run(method, threads, sweeps) for _ in range(reps))
        return table

    t_all = measure(threads_all, *plan["all"])
    best = max(t_all, key=t_all.get)
    if protocol == "middle":   # the reported figure by the survey's protocol: best of 3 x 10 sweeps of the best variant
        t_all[best] = max(t_all[best], measure(threads_all, 3, 10, only=best)[best])
    t_one = measure(1, *plan["one"], only=None if protocol == "full" else best)
    best_one = max(t_one, key=t_one.get)
    O.use_own_gemm()
    O.set_threads(1)
    return {
        "value": round(t_all[best], 4), "unit": "ALS it/s", "cores": threads_all, "kind": "port",
        "variant": best, "per_variant": {k: round(v, 4) for k, v in t_all.items()},
        "single_thread": {"value": round(t_one[best_one], 4), "cores": 1, "variant": best_one,
                          "per_variant": {k: round(v, 4) for k, v in t_one.items()}},
        "cpu": cpu_model_name(), "protocol": protocol,
        "sample": "same workload (X, models, line search); per MTTKRP variant 1 warm-up sweep, then best of "
                  "%d x %d forced sweeps at %d threads%s; at 1 thread %s, %d x %d sweeps; GEMMs by %s" % (
                      plan["all"][0], plan["all"][1], threads_all,
                      ", the best variant again as best of 3 x 10 (SURVEY 8d)" if protocol == "middle" else "",
                      "all variants" if protocol == "full" else "the best variant only",
                      plan["one"][0], plan["one"][1],
                      "MKL (libmkl_rt, image runtime)" if have_mkl else "the oracle's own loops"),
    }


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process has made no GPU call (torch is
    not even imported yet), so it starts the launcher the contract names as a CHILD -- one rank per GPU, rendezvous on
    127.0.0.1 at a free port -- with the same arguments, lets the children write to its stdout / stderr (rank 0
    prints the JSON line) and exits with the launcher's status.  Never an exec: a process that may have touched the
    GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_used_gib(torch, dev):
    """GiB of the device in use right now (everything: this process's engines, the runtime, other processes)."""
    free_b, total_b = torch.cuda.mem_get_info(dev)
    return round((total_b - free_b) / 2.0 ** 30, 3), round(total_b / 2.0 ** 30, 1)


def strong_c5_leg(cc, inputs, sharding, torch, X, world, rank, local_rank, steps, warmup, red_dev):
    """BASELINE config 5 next to the weak line: 2048 jackknife models IN TOTAL on C3's X, model m -> GPU m mod N,
    one step = one sweep of the whole job, job rate = K / max-over-ranks time of its own barrier-bracketed region.
    N = 1 gives the denominator of north_star's ">= 6x at 8 GPUs"."""
    modes, total, ls = WORKLOADS["c5"]
    mine = sharding.shard_round_robin(total, world, rank)
    ranks = [1 + (m % 20) for m in mine]
    R = sum(ranks)
    base = inputs.model_factors(modes, ranks, seed=101 + rank)
    jk = [(0, m % modes[0]) for m in mine]
    for (fs, _), (jm, jf) in zip(base, jk):
        fs[jm][jf, :] *= 0.0
    eng = cc.Engine(modes, R, device=local_rank, dtype="f64")
    eng.set_tensor(X)
    eng.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, line_search=ls,
                                     line_search_interval=5, line_search_step=0.0))
    for k, (fs, lam) in enumerate(base):
        eng.enqueue(cc.Model(fs, lam, jk=jk[k]))
    assert eng.admit() == len(mine) and eng.active_cols == R
    eng.set_profiling(3)
    eng.sweep(warmup)
    eng.synchronize()
    eng.reset_kernel_stats()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sweep(steps)
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    value, t_max = sharding.strong_rate(steps, elapsed, device=red_dev)
    per_rank_ms = [round(t / steps * 1e3, 4) for t in sharding.gather_over_ranks(elapsed, device=red_dev)]
    ks = eng.kernel_stats()
    eng.set_profiling(False)
    used_gib, _ = device_used_gib(torch, torch.device("cuda", local_rank))
    eng.close()
    out = {"metric": "ALS iterations/sec of the whole 2048-model job", "value": round(value, 3), "unit": "ALS it/s",
           "device_memory_in_use_GiB": used_gib,
           "scaling": "strong", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(t_max / steps * 1e3, 4), "ms_per_step_per_rank": per_rank_ms,
           "total_models": total, "models_on_rank0": len(mine), "columns_on_rank0": R, "jackknife": True,
           "config": "BASELINE config 5: 300x300x300 fp64, 2048 jackknife models (jk = mode 0, fiber m mod 300), "
                     "model m -> GPU m mod N, line search on"}
    if ks.ttm_launches:
        ms = ks.ttm_ms / ks.ttm_launches
        out["ttm_kernel"] = {"launches": ks.ttm_launches, "avg_launch_ms": round(ms, 4),
                             "achieved_tflops": round(ks.ttm_flops / ks.ttm_launches / (ms * 1e-3) * 1e-12, 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 50 sweeps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="further sweeps timed after the K steps for the `steady_state` field (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-protocol", default="middle", choices=["bounded", "middle", "full"])
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="c3 only: skip the strong-scaling config-5 job (`strong_c5`) timed after the weak one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo only to rehearse the N>1 path")
    ap.add_argument("--force-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --dist-backend gloo)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # dmabuf IPC only on this pool: RCCL's hipIpcGetMemHandle fails without it.  Set here as well as in self_launch, so
    # that a rank started by somebody else's launcher (the driver's torchrun) has it before the runtime initialises.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import cp_cals_amd as cc
    from cp_cals_amd import inputs, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the CALS engine has no CPU fallback")
    if args.force_device0:
        local_rank = 0
    n_visible = torch.cuda.device_count()
    if local_rank >= n_visible and n_visible == 1 and world > 1 and args.dist_backend == "nccl":
        # a launcher that gives every rank its own GPU through *_VISIBLE_DEVICES shows each of them one device: use it.
        # (On a box with ONE GPU for all ranks RCCL refuses the group -- "Duplicate GPU detected" -- right below.)
        local_rank = 0
    if local_rank >= n_visible:
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, n_visible))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg_world = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the requested backend or nothing: a record that says "nccl" must mean RCCL saw `world` ranks
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        pg_world = dist.get_world_size()
        if pg_world != world or dist.get_backend() != args.dist_backend:
            raise SystemExit("process group: backend %s, %d ranks; wanted %s, %d" % (
                dist.get_backend(), pg_world, args.dist_backend, world))
    red_dev = dev if (world == 1 or args.dist_backend == "nccl") else torch.device("cpu")

    modes, k_models, ls = WORKLOADS[args.workload]
    strong = args.workload in STRONG
    if strong:
        total_models = k_models
        mine = sharding.shard_round_robin(total_models, world, rank)   # model m -> GPU m mod N
    else:
        total_models = k_models * world
        mine = [rank + world * k for k in range(k_models)]   # this rank's shard: models m = rank + world*k
    k_models = len(mine)
    ranks = [1 + (m % 20) for m in mine] if strong else [1 + (k % 20) for k in range(k_models)]
    R = sum(ranks)
    X = inputs.tensor(modes, seed=0)        # replicated: every rank generates the same X
    base = inputs.model_factors(modes, ranks, seed=1 + rank)
    jk = None
    if args.workload in JACKKNIFE:
        jk = [(0, m % modes[0]) for m in mine]
        for (fs, _), (jm, jf) in zip(base, jk):
            fs[jm][jf, :] *= 0.0            # Ktensor::fill zeroes the jk fiber (src/ktensor.cpp:21-30)

    dtype = WORKLOAD_DTYPE.get(args.workload, "f64")
    peak = PEAK_FP32_MFMA_TFLOPS if dtype == "f32" else PEAK_FP64_MFMA_TFLOPS
    eng = cc.Engine(modes, R, device=local_rank, dtype=dtype)
    eng.set_tensor(X)
    eng.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, line_search=ls,
                                     line_search_interval=5, line_search_step=0.0))
    models = [cc.Model([f.copy() for f in fs], lam.copy(), jk=None if jk is None else jk[k])
              for k, (fs, lam) in enumerate(base)]
    for m in models:
        eng.enqueue(m)
    assert eng.admit() == k_models and eng.active_cols == R

    # The live kernel statistics cost queue time (an event pair is two more packets and keeps the next launch from
    # overlapping the kernel's tail; tools/profiling_cost.py), so the timed region brackets only the MFMA kernels --
    # the roofline's dominant kernel is one of them -- and the contraction is measured during the warm-up sweeps.
    eng.set_profiling(2)                    # MFMA kernels + the contraction
    eng.reset_kernel_stats()
    eng.sweep(args.warmup)
    eng.synchronize()
    ks_warm = eng.kernel_stats()
    eng.set_profiling(3)                    # hipEvent pairs around the MFMA kernels only
    eng.reset_kernel_stats()

    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sweep(args.steps)                   # EXACTLY K steps
    torch.cuda.synchronize()                # device-wide: covers the engine's own stream
    sharding.barrier()
    elapsed = time.perf_counter() - t0

    if strong:
        value, t_max = sharding.strong_rate(args.steps, elapsed, device=red_dev)
    else:
        value, t_max = sharding.aggregate_rate(args.steps, elapsed, device=red_dev)
    per_rank_ms = [round(t / args.steps * 1e3, 4) for t in sharding.gather_over_ranks(elapsed, device=red_dev)]
    ks = eng.kernel_stats()
    plan = eng.tree
    eng.set_profiling(False)
    used_gib, total_gib = device_used_gib(torch, dev)   # this leg's engine resident (X copies, factors, T, partials)

    # ---- beside the contract: the long-run rate and the loop users run (not part of `value`) ----
    steady = run_loop = None
    if args.steady_steps > 0:
        sharding.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.sweep(args.steady_steps)
        torch.cuda.synchronize()
        sharding.barrier()
        dt = time.perf_counter() - t1
        sv, st_max = (sharding.strong_rate if strong else sharding.aggregate_rate)(args.steady_steps, dt, device=red_dev)
        steady = {"steps": args.steady_steps, "value": round(sv, 3), "unit": "ALS it/s",
                  "ms_per_step": round(st_max / args.steady_steps * 1e3, 4),
                  "after_sweeps": args.warmup + args.steps,
                  "note": "back-to-back sweeps right after the K timed steps, same engine state"}
        n_loop = min(args.steady_steps, 100)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n_loop):
            eng.step()                      # admit (nothing queued) + sweep + status read-back + eviction rule
        eng.synchronize()
        dt2 = time.perf_counter() - t2
        assert eng.models_in_flight == k_models
        rv, rt_max = (sharding.strong_rate if strong else sharding.aggregate_rate)(n_loop, dt2, device=red_dev)
        run_loop = {"steps": n_loop, "value": round(rv, 3), "unit": "ALS it/s",
                    "ms_per_step": round(rt_max / n_loop * 1e3, 4),
                    "note": "cals_hip_step loop = one iteration of cals_hip_run (src/cals.cpp:174-382): per-sweep "
                            "status read-back and eviction decision included (force_max_iter: nothing leaves)"}

    out = None
    if rank == 0:
        # the two MFMA kernels of a sweep: the fused MTTKRP and (dimension-tree plans) the TTM that
        # replaces two of the three MTTKRPs; both do 2*prod(modes)*R algorithmic flops per launch
        kern = {}
        for name, n, ms, fl in (("mttkrp3_kernel (fused MTTKRP)", ks.mttkrp_launches, ks.mttkrp_ms, ks.mttkrp_flops),
                                ("ttm_kernel (TTM shared by two modes + fused G)", ks.ttm_launches, ks.ttm_ms, ks.ttm_flops)):
            if n:
                kern[name] = {"launches": n, "avg_launch_ms": round(ms / n, 4), "total_ms": round(ms, 3),
                              "flops_per_launch": fl / n,
                              "achieved_tflops": round(fl / n / (ms / n * 1e-3) * 1e-12, 3)}
        dom = max(kern, key=lambda k: kern[k]["total_ms"])
        avg_ms = kern[dom]["avg_launch_ms"]
        flops_per_launch = kern[dom]["flops_per_launch"]
        achieved = kern[dom]["achieved_tflops"]
        contract = None
        contract_ms_per_step = 0.0
        if ks_warm.contract_launches and args.warmup > 0:
            contract = {"launches": ks_warm.contract_launches,
                        "avg_launch_ms": round(ks_warm.contract_ms / ks_warm.contract_launches, 4),
                        "achieved_GBps": round(ks_warm.contract_bytes / (ks_warm.contract_ms * 1e-3) * 1e-9, 1),
                        "bound": "hbm", "peak_GBps": 8000,
                        "note": "measured over the %d warm-up sweeps (not bracketed inside the timed region)" % args.warmup}
            contract_ms_per_step = ks_warm.contract_ms / args.warmup
        # HBM bytes per launch of the dominant kernel from the PMC passes (profiles/traffic.json:
        # {workload: {kernel: bytes}}; rocprofv3 cannot run inside this process)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            per_kernel = json.load(open(tpath)).get(args.workload)
            if isinstance(per_kernel, dict):
                traffic = per_kernel.get(dom.split(" ")[0])
        out = {
            "metric": "ALS iterations/sec (all concurrent models)",
            "value": round(value, 3), "unit": "ALS it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": ("%s %s dense tensor, %d concurrent %sCP models per GPU, ranks 1..20 "
                                    "(R=%d columns), line search %s; " % (
                                        "x".join(map(str, modes)), "fp32" if dtype == "f32" else "fp64",
                                        k_models, "jackknife (jk = mode 0, fiber m mod %d) " % modes[0] if jk else "",
                                        R, "on" if ls else "off")) + (
                                    "one step = one ALS sweep of ALL %d models (sharded over %d GPU(s)); "
                                    "value = steps / max time" % (total_models, world) if strong else
                                    "one step = one ALS sweep of a GPU's model shard; value = sweeps by all "
                                    "%d GPU(s) / max time" % world),
                       "name": args.workload, "models_per_gpu": k_models, "total_models": total_models,
                       "jackknife": bool(jk),
                       "sharding": "model m -> GPU m mod N, X replicated, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": "%s, v_mfma_%s" % (
                             dom, "f32_16x16x4_f32" if dtype == "f32" else "f64_16x16x4_f64"),
                         "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": "stored PMC figure: profiles/traffic.json, from separate rocprofv3 --pmc passes "
                                           "of this command (rocprofv3 cannot run inside this process); not measured in this run",
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": round(avg_ms, 4),
                         "launches": kern[dom]["launches"],
                         "plan": {0: "3 fused MTTKRPs per sweep", 1: "dimension tree A (modes 0,1 share X x_2 C)",
                                  2: "dimension tree B (modes 1,2 share X x_0 A)",
                                  3: "multi-sweep dimension tree (every TTM shared by two consecutive "
                                     "updates: 3 TTMs per 2 sweeps)"}[plan],
                         "mfma_kernels": kern, "contract_kernel": contract,
                         "rest_ms_per_step": round(
                             t_max / args.steps * 1e3 - (ks.mttkrp_ms + ks.ttm_ms) / args.steps - contract_ms_per_step, 4)},
            "steady_state": steady,
            "run_loop": run_loop,
            "dist": {"backend": args.dist_backend if world > 1 else None, "world_size": pg_world,
                     "ms_per_step_per_rank": per_rank_ms},
            # rank 0's device with this leg's engine resident; the strong leg (its own engine, created after this one
            # is closed) reports its own figure under strong_c5: the two legs never hold device memory at the same time
            "device_memory": {"in_use_GiB": used_gib, "total_GiB": total_gib},
        }
    eng.close()
    if args.workload == "c3" and not args.no_strong_leg:
        leg = strong_c5_leg(cc, inputs, sharding, torch, X, world, rank, local_rank, args.steps, args.warmup, red_dev)
        if rank == 0:
            out["strong_c5"] = leg
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "c5":
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)
        out["cpu_baseline"] = cpu_baseline(modes, ranks, X, base, jk, ls, threads, args.cpu_protocol)
    elif rank == 0:
        out["cpu_baseline"] = None
        if world == 1 and args.workload == "c5" and not args.no_cpu_baseline:
            out["cpu_baseline_note"] = ("not timed for c5: one oracle sweep of 2048 models is ~8x config 3's "
                                        "(~12 s at 16 threads); the c3 line carries the CPU baseline")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
