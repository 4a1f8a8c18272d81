#!/usr/bin/env python3
"""bench.py -- the CALS hot path on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A step = one ALS sweep (every mode: fused MTTKRP + batched update; error; line search) of all
models in flight on a GPU.  Workload (BASELINE.json config 3, the one north_star's target is quoted
on): 300x300x300 fp64 tensor, 256 concurrent models of rank 1 + (k mod 20) per GPU (R = 2656
columns), line search on (NO_ERROR_CHECKING, interval 5, step cbrt(iter)); synthetic inputs from
cp-cals_amd/inputs.py.  X, factors and all model state are resident in HBM before the timed region.
N > 1: weak scaling, every rank owns its own 256-model shard (model m -> GPU m mod N), X replicated,
no data-path collective; value = sweeps completed by all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
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
    # BASELINE config 5: 2048 models IN TOTAL sharded round-robin over the N GPUs (strong scaling:
    # a step = one sweep of all 2048 models; value = steps / max time).  N = 1 gives the denominator of
    # north_star's ">= 6x at 8 GPUs".
    "c5": ([300, 300, 300], 2048, 1),
}
STRONG = {"c5"}
WORKLOAD_DTYPE = {"c4": "f32"}
PEAK_FP32_MFMA_TFLOPS = 157.3  # dense FP32 matrix peak (v_mfma_f32_16x16x4_f32: 256 flop/cycle/CU x4 SIMD)
PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (datasheet; 256 CU x 4 SIMD x 2.4 GHz
#                               x 2048 flop / 64 cycles).  tools/mfma_f64_peak measures 77.7 on the box.


def local_ranks(k_models):
    return [1 + (k % 20) for k in range(k_models)]


def cpu_baseline(modes, ranks, X, base, ls, threads, sweeps):
    """The oracle (CPU restatement of the reference algorithm, explicit KRP + GEMM / two-step) with
    the image's MKL runtime for the big GEMMs, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    have_mkl = O.use_mkl(threads)
    O.set_threads(threads)
    best = None
    for name, method in (("AUTO(no LUT)", O.AUTO), ("MTTKRP", O.MTTKRP)):
        models = [O.Model([f.copy() for f in fs], lam.copy()) for fs, lam in base]
        p = O.default_params(max_iterations=sweeps, force_max_iter=1, buffer_size=sum(ranks),
                             mttkrp_method=method, line_search=ls, line_search_interval=5,
                             threads=threads)
        rep = O.cp_cals(X, modes, models, p)
        rate = rep.iter / rep.loop_time
        if best is None or rate > best[0]:
            best = (rate, name)
    O.use_own_gemm()
    O.set_threads(1)
    return {
        "value": round(best[0], 4), "unit": "ALS it/s", "cores": threads, "kind": "port",
        "sample": "%d forced sweeps of the same workload per MTTKRP variant {AUTO(no LUT), MTTKRP}, "
                  "best variant (%s) reported; GEMMs by %s, %d threads" % (
                      sweeps, best[1], "MKL (libmkl_rt, image runtime)" if have_mkl else "the oracle's own loops",
                      threads),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-sweeps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo only to rehearse the N>1 path")
    ap.add_argument("--force-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (use with --dist-backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import cp_cals_amd as cc
    from cp_cals_amd import inputs, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the CALS engine has no CPU fallback")
    if args.force_device0:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.dist_backend
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            except Exception as exc:  # control plane only (barrier + max of a scalar): gloo does it too
                print("rank %d: RCCL process group failed (%s); using gloo" % (rank, exc), file=sys.stderr)
                backend = "gloo"
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        args.dist_backend = backend
    red_dev = dev if (world == 1 or args.dist_backend == "nccl") else torch.device("cpu")

    modes, k_models, ls = WORKLOADS[args.workload]
    strong = args.workload in STRONG
    if strong:
        total_models = k_models
        mine = list(range(rank, total_models, world))   # model m -> GPU m mod N
        k_models = len(mine)
        ranks = [1 + (m % 20) for m in mine]
    else:
        total_models = k_models * world
        ranks = local_ranks(k_models)       # this rank's shard: models m = rank + world*k
    R = sum(ranks)
    X = inputs.tensor(modes, seed=0)        # replicated: every rank generates the same X
    base = inputs.model_factors(modes, ranks, seed=1 + rank)

    dtype = WORKLOAD_DTYPE.get(args.workload, "f64")
    peak = PEAK_FP32_MFMA_TFLOPS if dtype == "f32" else PEAK_FP64_MFMA_TFLOPS
    eng = cc.Engine(modes, R, device=local_rank, dtype=dtype)
    eng.set_tensor(X)
    eng.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1, line_search=ls,
                                     line_search_interval=5, line_search_step=0.0))
    models = [cc.Model([f.copy() for f in fs], lam.copy()) for fs, lam in base]
    for m in models:
        eng.enqueue(m)
    assert eng.admit() == k_models and eng.active_cols == R

    eng.sweep(args.warmup)
    eng.synchronize()
    eng.set_profiling(2)                    # hipEvent pairs around the MFMA kernels + the contraction only
    eng.reset_kernel_stats()

    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sweep(args.steps)                   # EXACTLY K steps
    torch.cuda.synchronize()                # device-wide: covers the engine's own stream
    sharding.barrier()
    elapsed = time.perf_counter() - t0

    value, t_max = sharding.aggregate_rate(args.steps, elapsed, device=red_dev)
    if strong:
        value = args.steps / t_max          # one step advances the WHOLE job by one sweep
    ks = eng.kernel_stats()
    plan = eng.tree
    eng.set_profiling(False)

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
        if ks.contract_launches:
            contract = {"launches": ks.contract_launches,
                        "avg_launch_ms": round(ks.contract_ms / ks.contract_launches, 4),
                        "achieved_GBps": round(ks.contract_bytes / (ks.contract_ms * 1e-3) * 1e-9, 1),
                        "bound": "hbm", "peak_GBps": 8000}
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
            "config": {"workload": ("%s %s dense tensor, %d concurrent CP models per GPU, ranks 1..20 "
                                    "(R=%d columns), line search %s; " % (
                                        "x".join(map(str, modes)), "fp32" if dtype == "f32" else "fp64",
                                        k_models, R, "on" if ls else "off")) + (
                                    "one step = one ALS sweep of ALL %d models (sharded over %d GPU(s)); "
                                    "value = steps / max time" % (total_models, world) if strong else
                                    "one step = one ALS sweep of a GPU's model shard; value = sweeps by all "
                                    "%d GPU(s) / max time" % world),
                       "name": args.workload, "models_per_gpu": k_models, "total_models": total_models,
                       "sharding": "model m -> GPU m mod N, X replicated, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": "%s, v_mfma_%s" % (
                             dom, "f32_16x16x4_f32" if dtype == "f32" else "f64_16x16x4_f64"),
                         "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": round(avg_ms, 4),
                         "launches": kern[dom]["launches"],
                         "plan": {0: "3 fused MTTKRPs per sweep", 1: "dimension tree A (modes 0,1 share X x_2 C)",
                                  2: "dimension tree B (modes 1,2 share X x_0 A)",
                                  3: "multi-sweep dimension tree (every TTM shared by two consecutive "
                                     "updates: 3 TTMs per 2 sweeps)"}[plan],
                         "mfma_kernels": kern, "contract_kernel": contract,
                         "rest_ms_per_step": round(
                             t_max / args.steps * 1e3 - (ks.mttkrp_ms + ks.ttm_ms + ks.contract_ms) / args.steps, 4)},
        }
    eng.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not strong:
        threads = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)
        out["cpu_baseline"] = cpu_baseline(modes, ranks, X, base, ls, threads, args.cpu_sweeps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
