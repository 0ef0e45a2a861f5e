#!/usr/bin/env python3
"""Benchmark of the batched intercept-environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one `VecEnv.step` of the hot path over this rank's 65 536 environments (medium
scenario, base physics, fp32; BASELINE.json configs[1]): ONE launch of the fused HIP kernel,
auto-resets included.  Actions come from a pre-generated tape already resident in HBM; launches
are issued back to back from C (`hlx_rollout`), one per step, as a policy-free rollout would.
Environments shard over ranks with no collective in the step (weak scaling: 65 536 envs per GPU).

Prints ONE JSON line on rank 0.  `value` = env-steps/s of the whole job (all ranks, max-over-ranks
time).  `roofline` = algorithmic bytes of one launch / mean launch duration, from HIP events recorded on the
launch stream around the K back-to-back launches of the timed region (launches are dependent and issued
ahead of the GPU, so the train has no gaps: elapsed / K is the per-launch duration rocprofv3 reports).  `cpu_baseline` = the
CPU oracle (scalar C port of the reference's step, oracle/) timed on this host, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
BYTES_PER_ENV_STEP = {"base": 508, "v2dr": 604}     # SURVEY.md 8(d) algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(rc, budget_s=12.0):
    """Scalar C port of the reference step (oracle/), OpenMP over envs on this host's cores."""
    import oracle.oracle as orc

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = 16384
    ov = orc.OracleVec(rc, n)
    rng = np.random.default_rng(0)
    ov.reset(rng.random((n, orc.RESET_SLOTS)))
    acts = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
    sn = rng.standard_normal((n, orc.STEP_SLOTS))
    sn[:, [6, 11, 12, 19]] = rng.random((n, 4))
    rn = rng.random((n, orc.RESET_SLOTS))
    ov.step(acts, sn, rn)   # warm
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        ov.step(acts, sn, rn)
        steps += 1
    dt = time.perf_counter() - t0
    out = {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{n} envs x {steps} steps, medium scenario base physics, oracle/hlx_oracle.c with OpenMP over envs"}
    try:   # the same port on ONE thread (SURVEY.md 8(d) asks for both), ~3 s
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        t0, s1 = time.perf_counter(), 0
        while time.perf_counter() - t0 < 3.0:
            ov.step(acts, sn, rn)
            s1 += 1
        out["single_thread_value"] = n * s1 / (time.perf_counter() - t0)
        gomp.omp_set_num_threads(cores)
    except OSError:
        pass
    return out


def measured_traffic(physics, n):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    None when no measurement exists for this workload."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"{physics}:{n}", {}).get("hbm_bytes_per_launch")
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--physics", default="base", choices=["base", "v2dr"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max(time); gloo + --single-device rehearses the multi-rank "
                         "path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-extra-points", action="store_true",
                    help="skip BASELINE config 3 (v2dr physics, 65 536 envs) and the 4 M-env cache-defeating point of "
                         "SURVEY.md 8(d) that a single-GPU run also times (~10 s)")
    ap.add_argument("--fused", type=int, default=64,
                    help="also time the fused rollout (this many steps per launch, state held on-chip); 0 = skip")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the step is a HIP kernel; there is no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)   # nccl = RCCL; used only for barrier + max(time)

    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.shard import max_over_ranks, shard_range, whole_job_throughput
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    n = args.envs_per_gpu
    rc = resolve_config(scenario_config("medium", args.physics))
    offset, count = shard_range(n * world, world, rank)          # weak scaling: n envs per rank
    assert count == n
    env = HlynrVecEnv(resolved=rc, num_envs=n, device=local_rank, seed=1000, env_id_offset=offset)
    dev = env.device
    variant = env.kernel_variant      # read now: the handle is gone once the extra points have run
    K, W = args.steps, args.warmup
    gen = torch.Generator(device=dev).manual_seed(rank)      # fixed-seed synthetic action tape, U(-1, 1)
    tape_len = max(K, W, 1)
    tape = torch.rand((tape_len, n, 6), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    out_slots = 8

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    env.reset_torch()
    if W:
        env.rollout_torch(tape[:W], out_slots)
    sync_all()
    # HIP events recorded on the launch stream bracket the K back-to-back launches of the timed region
    env.profile(True)
    t0 = time.perf_counter()
    env.rollout_torch(tape[:K], out_slots)
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = env.profile_read()
    env.profile(False)
    red_dev = dev if args.backend == "nccl" else None       # gloo reduces a host scalar
    elapsed = max_over_ranks(elapsed, dist, red_dev)
    kern_us = 1e3 * kern_ms / max(1, launches)
    bytes_per_launch = BYTES_PER_ENV_STEP[args.physics] * n
    achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9 if launches else 0.0

    # SURVEY.md 8(d) caveat: the T-step persistent number beside the one-launch-per-step headline
    fused = None
    if args.fused > 1:
        env.set_rollout_fused(args.fused)
        env.rollout_torch(tape[:min(W, K) or 1], out_slots)
        sync_all()
        env.profile(True)
        t0 = time.perf_counter()
        env.rollout_torch(tape[:K], out_slots)
        sync_all()
        f_elapsed = max_over_ranks(time.perf_counter() - t0, dist, red_dev)
        f_ms, f_steps = env.profile_read()
        env.profile(False)
        env.set_rollout_fused(1)
        fused = {"steps_per_launch": args.fused, "value": whole_job_throughput(n, K, world, f_elapsed), "unit": "env-steps/s",
                 "ms_per_step": 1e3 * f_elapsed / K, "device_us_per_step": 1e3 * f_ms / max(1, f_steps),
                 "note": "same K steps through hlx_rollout with hlx_set_rollout_fused: state stays in registers for "
                         "steps_per_launch steps, bit-identical results; not the headline (a policy in the loop needs one launch per step)"}

    # config 3 and a batch whose state (2.6 GB) defeats the 256 MB Infinity Cache, same clocking method
    extra = None
    if not args.no_extra_points and world == 1:
        extra = []
        env.close()
        for phys, n_x, k_x in (("v2dr", ENVS_PER_GPU, 1000), ("base", 4 * 1024 * 1024, 60)):
            rc_x = resolve_config(scenario_config("medium", phys))
            env = HlynrVecEnv(resolved=rc_x, num_envs=n_x, device=local_rank, seed=1000)
            tape_x = torch.rand((k_x, n_x, 6), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
            env.reset_torch()
            env.rollout_torch(tape_x[:max(1, k_x // 4)], out_slots)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            env.rollout_torch(tape_x, out_slots)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            b = BYTES_PER_ENV_STEP[phys]
            extra.append({"workload": f"medium scenario, {phys} physics, {n_x} envs/GPU", "value": n_x * k_x / dt, "unit": "env-steps/s",
                          "us_per_step": 1e6 * dt / k_x, "algorithmic_bytes_per_env_step": b,
                          "roofline_frac": n_x * k_x * b / dt / 1e9 / HBM_PEAK_GBS})
            env.close()
            del tape_x

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(rc)

    if rank == 0:
        line = {
            "metric": "env-steps/sec whole-node, medium scenario, 64k envs/GPU",
            "value": whole_job_throughput(n, K, world, elapsed), "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"medium scenario, {args.physics} physics, {n} envs/GPU, fp32 "
                                   f"(BASELINE.json configs[{1 if args.physics == 'base' else 2}])",
                       "envs_per_gpu": n, "kernel_variant": variant, "launches_per_step": 1,
                       "sharding": f"{world} x {n} independent envs, no collective in the step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.physics, n),
                         "kernel": "hlx_env_kernel<%s, step>" % variant,
                         "kernel_us": kern_us, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches_timed": launches},
            "cpu_baseline": cpu,
            "fused_rollout": fused,
            "extra_points": extra,
        }
        print(json.dumps(line), flush=True)
    if extra is None:
        env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
