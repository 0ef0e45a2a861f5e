#!/usr/bin/env python3
"""Benchmark of the batched intercept-environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N > 1 without torchrun: the script launches its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one `VecEnv.step` of the hot path over this rank's 65 536 environments (medium scenario, base physics,
fp32; BASELINE.json configs[1]): ONE launch of the fused HIP kernel, auto-resets included.  Actions come from a
pre-generated tape already resident in HBM; launches are issued back to back from C (`hlx_rollout`), one per step, as a
policy-free rollout would.  Environments shard over ranks with no collective in the step (weak scaling).

STEADY STATE.  Right after reset all episodes are in lock-step and every wave takes the same branches; a training run
never sees that phase again.  Before anything is timed the episodes are therefore desynchronised with a fused rollout
of `--desync` steps (default 4096, ~30 ms; independent of --warmup): per-launch time then no longer depends on K.

Prints ONE JSON line on rank 0.  `value` = env-steps/s of the whole job (all ranks, max-over-ranks wall time around the
K timed steps).  `roofline` = algorithmic bytes of one launch / mean launch duration, from HIP events recorded on the
launch stream around the K back-to-back launches (dependent launches issued ahead of the GPU: the train has no gaps).
`hlx_rollout` has no terminal-observation output (as in round 1), so its launches observe every environment once;
`roofline.with_terminal_observations` times the same launches with that output requested.
`selfcheck` = four 64-environment slabs of this rank's batch replayed from reset through every step of the run -- by
the CPU oracle, fed the Philox draws the kernel consumed -- and compared with the timed rollout's own outputs and final
state (non-zero exit status on a mismatch).  `cpu_baseline` = the CPU oracle timed on this host (rank 0, N = 1 only).

Other BASELINE.json workloads, not the headline:  --config 4  hard scenario, policy in the loop + gradient all-reduce;
                                                   --config 5  volley K=3 + HRL controller + resident LSTM state.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
BYTES_PER_ENV_STEP = {"base": 508, "v2dr": 604}     # SURVEY.md 8(d) algorithmic bytes per env-step
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The reference's own Python step cannot travel to the GPU box; its numbers were taken in the build container
# (8 host cores, numpy 2.2.6) and are carried as labelled constants.
REFERENCE_PYTHON = {
    "subproc_16_envs_8_cores": {"value": 4378.0, "unit": "env-steps/s", "cores": 8,
                                "workload": "BASELINE.json configs[0]: easy scenario, 16 envs, one process each, 1000-step rollout"},
    "sequential_1_core": {"value": 2232.0, "unit": "env-steps/s", "cores": 1, "workload": "the same 16 envs stepped in one process"},
    "single_env_1_core": {"value": 3200.0, "unit": "env-steps/s", "cores": 1, "workload": "one medium-scenario env, base physics"},
    "provenance": "measured with the reference's InterceptEnvironment in the build container (BASELINE.md section 2); "
                  "not re-measured on this box: /root/reference does not exist here",
}


# ----------------------------------------------------------------------------------------------------------------
# multi-rank launch: `python bench.py --gpus N` outside torchrun starts its own N ranks
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(argv, n):
    """Start N children (one rank each), BEFORE anything in this process touches the GPU; this parent never does.
    Children are fresh interpreters (no exec from a GPU-initialised process).  Rank 0's stdout is passed through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle port) and self-check
# ----------------------------------------------------------------------------------------------------------------
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
           "sample": f"{n} envs x {steps} steps, medium scenario base physics, oracle/hlx_oracle.c with OpenMP over envs",
           "reference_python": REFERENCE_PYTHON}
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


class SelfCheck:
    """Four 64-environment slabs of the rank's batch, replayed from reset through the same action schedule: on the GPU as
    separate small handles (same seed, `env_id_offset` = the slab's global ids) and on the CPU by the oracle, which is fed
    the Philox draws exported with `hlx_fill_noise`.  At the end the big batch's own outputs and state for those global
    ids must equal the slabs' bit for bit, and the slabs must equal the oracle within BASELINE.json's 1e-5."""

    SLAB = 64

    def __init__(self, rc, n, seed, offset, device_index):
        import oracle.oracle as orc
        from hlynr_intercept_amd.vec_env import HlynrVecEnv
        self.orc = orc
        rng = np.random.default_rng(seed + 17)
        blocks = n // self.SLAB
        picks = [0] + sorted(int(b) for b in rng.choice(np.arange(1, blocks), size=min(3, blocks - 1), replace=False)) if blocks > 1 else [0]
        self.starts = [b * self.SLAB for b in picks]
        self.envs = [HlynrVecEnv(resolved=rc, num_envs=self.SLAB, device=device_index, seed=seed, env_id_offset=offset + s)
                     for s in self.starts]
        self.oras = [orc.OracleVec(rc, self.SLAB) for _ in self.starts]
        self.steps = 0
        self.rew_max = 0.0
        self.dist_max = 0.0
        self.flag_bad = 0
        self.obs_bad = 0
        self.obs_max = 0.0
        self.last = None

    def reset(self):
        for env, ora in zip(self.envs, self.oras):
            _, rn = env.fill_noise(for_reset=True)
            og = env.reset_torch().cpu().numpy()
            oo = ora.reset(rn.cpu().numpy().T.copy())
            self.obs_max = max(self.obs_max, float(np.max(np.abs(og - oo))))

    def step(self, actions):
        """actions: the big batch's [N, 6] action tensor of this step"""
        self.last = []
        for env, ora, s in zip(self.envs, self.oras, self.starts):
            a = actions[s:s + self.SLAB].contiguous()
            sn, rn = env.fill_noise()
            obs, rew, term, trunc, info = env.step_torch(a)
            out = ora.step(a.cpu().numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
            te, tr = term.cpu().numpy(), trunc.cpu().numpy()
            done = (te | tr).astype(bool)
            self.flag_bad += int(((te != out["terminated"]) | (tr != out["truncated"])).sum())
            r = rew.cpu().numpy().astype(np.float64)
            self.rew_max = max(self.rew_max, float(np.max(np.abs(r - out["reward"]) / np.maximum(1.0, np.abs(out["reward"])))))
            d = info["distance"].cpu().numpy().astype(np.float64)
            self.dist_max = max(self.dist_max, float(np.max(np.abs(d - out["distance"]) / np.maximum(1.0, np.abs(out["distance"])))))
            og = np.where(done[:, None], info["terminal_observation"].cpu().numpy(), obs.cpu().numpy())
            oo = np.where(done[:, None], ora.terminal_obs, out["obs"])
            eo = np.max(np.abs(og - oo), axis=1)
            # a Bernoulli detection decided differently at a float32 boundary moves the Kalman track of that env until
            # its episode ends (observation only; state, reward and flags are unaffected): counted, not averaged away
            self.obs_bad += int((eo > 1e-3).sum())
            self.obs_max = max(self.obs_max, float(eo[eo <= 1e-3].max(initial=0.0)))
            self.last.append((obs.clone(), rew.clone(), term.clone(), trunc.clone()))
        self.steps += 1

    def finish(self, big_env, big_last):
        """big_last = (obs, reward, terminated, truncated) of the big batch's LAST step.  Returns the report dict."""
        import torch
        identical = True
        state_max = 0.0
        st_big = np.frombuffer(big_env.get_state(), dtype=np.dtype(type(big_env.get_state()[0])))
        for env, ora, s, last in zip(self.envs, self.oras, self.starts, self.last):
            sl = slice(s, s + self.SLAB)
            for x, y in zip(big_last, last):
                identical &= bool(torch.equal(x[sl], y))
            st = np.frombuffer(env.get_state(), dtype=np.dtype(type(env.get_state()[0])))
            identical &= st_big[sl].tobytes() == st.tobytes()
            so = np.frombuffer(ora.state, dtype=np.dtype(type(ora.state[0])))
            for f in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind"):
                a, b = st[f].astype(np.float64), so[f].astype(np.float64)
                state_max = max(state_max, float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))))
            identical &= bool(np.array_equal(st["steps"], so["steps"]))
        for env in self.envs:
            env.close()
        n = len(self.starts) * self.SLAB
        # Gate on what the step DEFINES exactly: flags, reward, distance, integrated state, and the big batch == its slabs.
        # Observation entries are pure outputs of fast float32 formulas, some ill-conditioned by construction (time to
        # intercept ~ range / closing as closing -> 0): their worst deviation is reported, and gated only at 1e-3.
        ok = identical and self.flag_bad == 0 and self.rew_max <= 1e-5 and self.dist_max <= 1e-5 and state_max <= 2e-5 and \
            self.obs_max <= 1e-3 and self.obs_bad <= max(4, n * self.steps // 20000)
        return {"ok": bool(ok), "envs": n, "global_env_slabs": self.starts, "steps": self.steps, "env_steps": n * self.steps,
                "batch_equals_slabs_bit_for_bit": bool(identical), "reward_max_rel": self.rew_max, "distance_max_rel": self.dist_max,
                "flag_mismatches": self.flag_bad, "obs_max_abs": self.obs_max, "obs_env_steps_with_diverged_detection": self.obs_bad,
                "state_max_rel": state_max, "against": "oracle/hlx_oracle.c fed the exported Philox draws (hlx_fill_noise)"}


def measured_traffic(physics, n):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  STATIC: read from the committed
    file, not collected by this run (counters need rocprofv3 around the process); None when no measurement exists."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f).get(f"{physics}:{n}", {})
            return d.get("hbm_bytes_per_launch"), "static: profiles/hbm_traffic.json (%s)" % d.get("source", "rocprofv3 --pmc, committed")
    except OSError:
        return None, None


def tape_schedule(total, tape_len):
    """Successive (lo, hi) slices of the action tape covering `total` steps (the tape is re-read from its start)."""
    out, done = [], 0
    while done < total:
        k = min(tape_len, total - done)
        out.append((0, k))
        done += k
    return out


# ----------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--desync", type=int, default=4096,
                    help="fused-rollout steps run before anything is timed so that episodes are out of lock-step (steady state)")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--physics", default="base", choices=["base", "v2dr"])
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[]: 2 = headline (configs[1]), 3 = v2dr physics, 4 = policy in the loop + gradient "
                         "all-reduce (configs[3]), 5 = volley + HRL controller + LSTM state (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfcheck", action="store_true")
    ap.add_argument("--no-terminal-obs-point", action="store_true",
                    help="skip the second timing of the same launches with terminal observations requested (profiling runs: keeps "
                         "the per-kernel averages of rocprofv3 about one form of the step only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max(time); gloo + --single-device rehearses the multi-rank "
                         "path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-extra-points", action="store_true",
                    help="skip BASELINE config 3 (v2dr physics, 65 536 envs) and the 4 M-env cache-defeating point of "
                         "SURVEY.md 8(d) that a single-GPU run also times (~10 s)")
    ap.add_argument("--fused", type=int, default=64,
                    help="also time the fused rollout (this many steps per launch, state held on-chip); 0 = skip")
    ap.add_argument("--rollout-steps", type=int, default=128, help="--config 4: n_steps of the rollout (reference: 2048)")
    ap.add_argument("--minibatches", type=int, default=8, help="--config 4: minibatch updates (one gradient all-reduce each)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU work at all: the ranks rendezvous (gloo), run the barrier / max-over-ranks plumbing on a fake time "
                         "and rank 0 prints a line with value null -- checks the self-launch path on a machine without a GPU")
    args = ap.parse_args()
    if args.config == 3:
        args.physics = "v2dr"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))        # before this process imports torch or touches a GPU

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.rehearse_cpu:
        import torch.distributed as dist
        from hlynr_intercept_amd.shard import max_over_ranks, shard_range
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        slowest = max_over_ranks(1.0 + rank, dist if world > 1 else None, None)
        if rank == 0:
            print(json.dumps({"metric": "rehearsal (no GPU work)", "value": None, "n_gpus": world, "slowest_rank_fake_time": slowest,
                              "shards": [shard_range(args.envs_per_gpu * world, world, r) for r in range(world)]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the step is a HIP kernel; there is no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)   # nccl = RCCL

    if args.config in (4, 5):
        import bench_configs
        line = (bench_configs.config4 if args.config == 4 else bench_configs.config5)(args, rank, local_rank, world, dist)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.shard import max_over_ranks, shard_range, whole_job_throughput
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    n = args.envs_per_gpu
    rc = resolve_config(scenario_config("medium", args.physics))
    offset, count = shard_range(n * world, world, rank)          # weak scaling: n envs per rank
    assert count == n
    seed = 1000
    env = HlynrVecEnv(resolved=rc, num_envs=n, device=local_rank, seed=seed, env_id_offset=offset)
    dev = env.device
    variant = env.kernel_variant      # read now: the handle is gone once the extra points have run
    K, W, D = args.steps, args.warmup, max(0, args.desync)
    gen = torch.Generator(device=dev).manual_seed(rank)      # fixed-seed synthetic action tape, U(-1, 1)
    tape_len = max(min(max(K, W), 2048), 1)
    tape = torch.rand((tape_len, n, 6), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    out_slots = 8
    check = None
    if rank == 0 and not args.no_selfcheck:
        check = SelfCheck(rc, n, seed, offset, local_rank)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The launches of a region are issued through the C ABI directly from prepared argument tuples: inside the timed
    # region the host does nothing but call hlx_rollout (one C call per <= tape_len steps), so that a short K is not
    # dominated by Python (the driver may time as few as 20 steps).
    import ctypes as C
    ring = (torch.zeros((out_slots, n, 26), device=dev), torch.zeros((out_slots, n), device=dev),
            torch.zeros((out_slots, n), dtype=torch.uint8, device=dev), torch.zeros((out_slots, n), dtype=torch.uint8, device=dev))
    rollout_c, stream_c = env._lib.hlx_rollout, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def plan(total):
        calls = [(env._h, C.c_void_p(tape[lo:hi].data_ptr()), hi - lo, out_slots, C.c_void_p(ring[0].data_ptr()), C.c_void_p(ring[1].data_ptr()),
                  C.c_void_p(ring[2].data_ptr()), C.c_void_p(ring[3].data_ptr()), stream_c) for lo, hi in tape_schedule(total, tape_len)]
        last = (calls[-1][2] - 1) % out_slots if calls else 0
        return calls, last

    def go(planned):
        for a in planned[0]:
            if rollout_c(*a) != 0:
                raise RuntimeError(env._lib.hlx_last_error().decode())
        return ring, planned[1]

    def run(total, fused=1):
        env.set_rollout_fused(fused)
        ret = go(plan(total))
        env.set_rollout_fused(1)
        return ret

    env.reset_torch()
    run(D, fused=64)                  # desynchronise the episodes (bit-identical to D single-step launches)
    run(W)
    timed = plan(K)
    sync_all()
    # HIP events recorded on the launch stream bracket the K back-to-back launches of the timed region
    env.profile(True)
    t0 = time.perf_counter()
    ring, last_slot = go(timed)
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = env.profile_read()
    env.profile(False)
    red_dev = dev if args.backend == "nccl" else None       # gloo reduces a host scalar
    elapsed = max_over_ranks(elapsed, dist, red_dev)
    kern_us = 1e3 * kern_ms / max(1, launches)
    bytes_per_launch = BYTES_PER_ENV_STEP[args.physics] * n
    achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9 if launches else 0.0
    wall_us = 1e6 * elapsed / max(1, K)

    # the same K steps with terminal observations requested (what hlx_step does for an SB3-style caller): finished
    # environments are then observed twice -- terminal state, then the new episode -- and their waves take a second trip
    # through the observation code
    K2, W2 = (0, 0) if args.no_terminal_obs_point else (min(K, 500), min(W, 64) or 1)
    two_pass_us = None
    if K2:
        env.set_rollout_terminal_obs(True)
        run(W2)
        sync_all()
        env.profile(True)
        ring, last_slot = run(K2)      # (the self-check below compares the outputs of this, the run's very last step)
        sync_all()
        tk_ms, tk_launches = env.profile_read()
        env.profile(False)
        env.set_rollout_terminal_obs(False)
        two_pass_us = 1e3 * tk_ms / max(1, tk_launches)

    selfcheck = None
    if check is not None:     # the slabs and the oracle walk through the same D + W + K steps, then everything is compared
        check.reset()
        for total in (D, W, K, W2, K2):       # ... and the terminal-observation rollouts above
            for lo, hi in tape_schedule(total, tape_len):
                for j in range(lo, hi):
                    check.step(tape[j])
        big_last = tuple(x[last_slot] for x in ring)
        selfcheck = check.finish(env, big_last)

    # SURVEY.md 8(d) caveat: the T-step persistent number beside the one-launch-per-step headline
    fused = None
    if args.fused > 1:
        run(min(W, K) or 1, fused=args.fused)
        sync_all()
        env.profile(True)
        t0 = time.perf_counter()
        run(K, fused=args.fused)
        sync_all()
        f_elapsed = max_over_ranks(time.perf_counter() - t0, dist, red_dev)
        f_ms, f_steps = env.profile_read()
        env.profile(False)
        fused = {"steps_per_launch": args.fused, "value": whole_job_throughput(n, K, world, f_elapsed), "unit": "env-steps/s",
                 "ms_per_step": 1e3 * f_elapsed / K, "device_us_per_step": 1e3 * f_ms / max(1, f_steps),
                 "note": "same K steps through hlx_rollout with hlx_set_rollout_fused: state stays in registers for "
                         "steps_per_launch steps, bit-identical results; not the headline (a policy in the loop needs one launch per step)"}

    # config 3 and a batch whose state (2.6 GB) defeats the 256 MB Infinity Cache, same clocking method
    extra = None
    if not args.no_extra_points and world == 1:
        extra = []
        env.close()
        for phys, n_x, k_x, d_x in (("v2dr", ENVS_PER_GPU, 1000, 4096), ("base", 4 * 1024 * 1024, 60, 0)):
            rc_x = resolve_config(scenario_config("medium", phys))
            env = HlynrVecEnv(resolved=rc_x, num_envs=n_x, device=local_rank, seed=seed)
            tape_x = torch.rand((min(k_x, 256), n_x, 6), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
            env.reset_torch()
            if d_x:
                env.set_rollout_fused(64)
                for _ in range(d_x // tape_x.shape[0]):
                    env.rollout_torch(tape_x, out_slots)
                env.set_rollout_fused(1)
            env.rollout_torch(tape_x[:max(1, k_x // 4)], out_slots)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for lo, hi in tape_schedule(k_x, tape_x.shape[0]):
                env.rollout_torch(tape_x[lo:hi], out_slots)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            b = BYTES_PER_ENV_STEP[phys]
            extra.append({"workload": f"medium scenario, {phys} physics, {n_x} envs/GPU", "value": n_x * k_x / dt, "unit": "env-steps/s",
                          "us_per_step": 1e6 * dt / k_x, "algorithmic_bytes_per_env_step": b, "desync_steps": d_x,
                          "roofline_frac": n_x * k_x * b / dt / 1e9 / HBM_PEAK_GBS})
            env.close()
            del tape_x

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(rc)

    if rank == 0:
        traffic, traffic_src = measured_traffic(args.physics, n)
        line = {
            "metric": "env-steps/sec whole-node, medium scenario, 64k envs/GPU",
            "value": whole_job_throughput(n, K, world, elapsed), "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"medium scenario, {args.physics} physics, {n} envs/GPU, fp32 "
                                   f"(BASELINE.json configs[{1 if args.physics == 'base' else 2}])",
                       "envs_per_gpu": n, "kernel_variant": variant, "launches_per_step": 1,
                       "phase": f"steady state: episodes desynchronised by {D} fused-rollout steps + {W} warmup steps before the timed region",
                       "sharding": f"{world} x {n} independent envs, no collective in the step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "hlx_env_kernel<%s, step>" % variant,
                         "kernel_us": kern_us, "wall_us_per_step": wall_us, "frac_from_wall_clock": bytes_per_launch / (wall_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "launches_timed": launches,
                         "with_terminal_observations": None if two_pass_us is None else {
                             "kernel_us": two_pass_us, "frac": bytes_per_launch / (two_pass_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "note": "same launches with hlx_set_rollout_terminal_obs: finished environments observed twice"}},
            "selfcheck": selfcheck,
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
    if selfcheck is not None and not selfcheck["ok"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
