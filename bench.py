#!/usr/bin/env python3
"""Benchmark of the batched intercept-environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N > 1 without torchrun: the script launches its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one `VecEnv.step` of the hot path over this rank's 65 536 environments (medium scenario, base physics,
fp32; BASELINE.json configs[1]): ONE launch of the fused HIP kernel, auto-resets included, IN THE FORM THE DROP-IN
ISSUES (`HlynrVecEnv.step_torch` -> `hlx_step` with terminal observations, every info plane and the done list: SB3
bootstraps from `infos[i]['terminal_observation']`, the reference's trainers read the info keys).  Actions come from a
pre-generated tape already resident in HBM; launches are issued back to back from C (`hlx_rollout` with
`hlx_set_rollout_outputs`), one per step, as a policy-free rollout would.  Environments shard over ranks with no
collective in the step (weak scaling).

STEADY STATE.  Right after reset all episodes are in lock-step and every wave takes the same branches; a training run
never sees that phase again.  Before anything is timed the episodes are therefore desynchronised with a fused rollout
of `--desync` steps (default 4096, ~30 ms; independent of --warmup): per-launch time then no longer depends on K.

Prints ONE JSON line on rank 0.  `value` = env-steps/s of the whole job (all ranks, max-over-ranks wall time around the
K timed steps, barrier + synchronize on both sides).  `roofline` = algorithmic bytes of one launch / launch duration:
the duration is the MEDIAN of R = max(5, ceil(400 / K)) back-to-back windows of K launches each, bracketed by HIP events
recorded on the launch stream (`window_us` lists them all), behind 2000 launches of the same form issued with no host
synchronisation in between -- a 20-launch window cannot absorb one hiccup, a median of twenty can.  The same
measurement is repeated for the other forms of the same kernel: `roofline.single_pass` (no optional output: finished
environments respawn before ONE observation pass) and `roofline.terminal_obs_only`.
`selfcheck` = four 64-environment slabs of this rank's batch replayed from reset through every step up to the end of
the timed region -- by the CPU oracle, fed the Philox draws the kernel consumed -- and compared with the timed rollout's
own outputs and state (non-zero exit status on a mismatch).  `cpu_baseline` = the CPU oracle timed on this host (rank 0,
N = 1 only).

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
BYTES_PER_ENV_STEP = {"base": 508, "v2dr": 604}     # SURVEY.md 8(d) algorithmic bytes per env-step: state words read once + written
                                                    # once, I/O words once, info{distance, min_distance} included
# ... per FORM of the step (roofline accounting):
#   contract      what HlynrVecEnv.step_torch issues: + info planes fuel 4, fuel_used 4 r + 4 w, flags 1, missiles 1,
#                 interceptor_pos 12, missile_pos 12, steps 4 = + 42 B.  (Terminal observation, episode return / length and
#                 the done list are written for finished environments only, ~0.2 B per env-step: not counted.)
#   single_pass,  no info plane at all: the 8 B of info{distance, min_distance} are not stored, so they are not counted
#   terminal_obs_only
FORM_BYTES_DELTA = {"contract": 42, "single_pass": -8, "terminal_obs_only": -8,
                    "contract_no_done_list": 42}      # (diagnostic: what the compacted done list costs on top of the info planes)
PREROLL = 2000                                      # launches of the same form ahead of every measurement, no host sync behind them
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
    # poll: the first rank that fails takes its siblings down (they would otherwise sit in the rendezvous / a barrier
    # until the process-group timeout); nothing is ever re-exec'd
    rc, live = 0, list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0:
                    rc = abs(code) or 1
        if live and rc == 0:
            time.sleep(0.05)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
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
        self.fuel_used_max = 0.0
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
            # a detection decided differently from the oracle would move the Kalman track of that env until its episode ends:
            # tests/ require that it never happens, and so does this check (counted separately so that the line says which)
            self.obs_bad += int((eo > 1e-3).sum())
            self.obs_max = max(self.obs_max, float(eo.max(initial=0.0)))
            fu = info["fuel_used"].cpu().numpy().astype(np.float64)
            self.fuel_used_max = max(self.fuel_used_max, float(np.max(np.abs(fu - out["fuel_used"]) / np.maximum(1.0, np.abs(out["fuel_used"])))))
            self.last.append((obs.clone(), rew.clone(), term.clone(), trunc.clone()))
        self.steps += 1

    def finish(self, big_state, big_last):
        """big_state = the big batch's exported state, big_last = its (obs, reward, terminated, truncated), both taken right
        behind the last step the slabs have replayed.  Returns the report dict."""
        import torch
        identical = True
        state_max = 0.0
        st_big = np.frombuffer(big_state, dtype=np.dtype(type(big_state[0])))
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
        # Gates: what the step DEFINES exactly (flags, reward, distance, fuel_used, integrated state, big batch == slabs) at
        # BASELINE.json's 1e-5; observation entries (pure outputs of fast float32 formulas) at the tests' bound, 2e-5
        # absolute over ALL rows (observed: <= 4e-6); no environment may leave the oracle's detection history.
        ok = identical and self.flag_bad == 0 and self.rew_max <= 1e-5 and self.dist_max <= 1e-5 and state_max <= 1e-5 and \
            self.fuel_used_max <= 1e-5 and self.obs_max <= 2e-5 and self.obs_bad == 0
        return {"ok": bool(ok), "envs": n, "global_env_slabs": self.starts, "steps": self.steps, "env_steps": n * self.steps,
                "batch_equals_slabs_bit_for_bit": bool(identical), "reward_max_rel": self.rew_max, "distance_max_rel": self.dist_max,
                "fuel_used_max_rel": self.fuel_used_max,
                "flag_mismatches": self.flag_bad, "obs_max_abs": self.obs_max, "obs_env_steps_with_diverged_detection": self.obs_bad,
                "state_max_rel": state_max, "against": "oracle/hlx_oracle.c fed the exported Philox draws (hlx_fill_noise)"}


def measured_traffic(physics, n, form="single_pass"):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  STATIC: read from the committed
    file, not collected by this run (counters need rocprofv3 around the process); None when no measurement exists."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
            d = table.get(f"{physics}:{n}:{form}")      # this workload in this form, or nothing: a figure is never borrowed from another configuration
            if d is None and form == "single_pass":
                d = table.get(f"{physics}:{n}")          # (the entries without a form are round-1/2 measurements of the single-pass launch)
            if d is None:
                return None, None
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
    ap.add_argument("--forms", default="contract,single_pass,terminal_obs_only",
                    help="which forms of the step the event-clocked windows measure (comma separated; the first one is the "
                         "headline's and must be `contract` unless this is a profiling run that wants one form's launches only)")
    ap.add_argument("--preroll", type=int, default=PREROLL,
                    help="launches of the measured form issued ahead of every measurement, with no host synchronisation behind them")
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
    forms = [f for f in args.forms.split(",") if f]
    for f in forms:
        if f not in FORM_BYTES_DELTA:
            raise SystemExit(f"--forms: unknown form {f!r} (known: {sorted(FORM_BYTES_DELTA)})")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))        # before this process imports torch or touches a GPU

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.rehearse_cpu:
        if os.environ.get("HLX_BENCH_TEST_FAIL_RANK") == str(rank):     # tests/test_bench_harness.py: a rank that dies before the rendezvous
            sys.exit(7)
        import torch.distributed as dist
        from hlynr_intercept_amd.shard import gather_over_ranks, max_over_ranks, shard_range
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        slowest = max_over_ranks(1.0 + rank, dist if world > 1 else None, None)
        per_rank = gather_over_ranks(1.0 + rank, dist if world > 1 else None, None)
        if rank == 0:
            print(json.dumps({"metric": "rehearsal (no GPU work)", "value": None, "n_gpus": world, "slowest_rank_fake_time": slowest,
                              "per_rank_fake_time": per_rank, "dist_world_size": dist.get_world_size() if world > 1 else 1,
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
    from hlynr_intercept_amd.shard import gather_over_ranks, max_over_ranks, shard_range, whole_job_throughput
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    n = args.envs_per_gpu
    rc = resolve_config(scenario_config("medium", args.physics))
    offset, count = shard_range(n * world, world, rank)          # weak scaling: n envs per rank
    assert count == n
    seed = 1000
    env = HlynrVecEnv(resolved=rc, num_envs=n, device=local_rank, seed=seed, env_id_offset=offset)
    dev = env.device
    variant = env.kernel_variant      # read now: the handle is gone once the extra points have run
    pool_info = {"fill_interval_steps": env.episode_pool, "load_schedule": env.load_schedule,
                 "note": "next-episode pool (include/hlx.h hlx_set_episode_pool): finished environments copy a prepared episode; one fill launch "
                         "of the step kernel's third mode per fill_interval_steps two-pass step launches -- those launches are INSIDE the "
                         "event-clocked windows and the wall-clock region, i.e. counted in kernel_us and in value"}
    K, W, D, P = args.steps, args.warmup, max(0, args.desync), max(0, args.preroll)
    R = max(5, -(-400 // max(1, K)))  # event-clocked windows of K launches each
    gen = torch.Generator(device=dev).manual_seed(rank)      # fixed-seed synthetic action tape, U(-1, 1)
    tape_len = max(min(max(K, W), 2048), 1)
    tape = torch.rand((tape_len, n, 6), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    out_slots = 8
    check = None
    if rank == 0 and not args.no_selfcheck:
        check = SelfCheck(rc, n, seed, offset, local_rank)
    red_dev = dev if args.backend == "nccl" else None       # gloo reduces a host scalar

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The launches of a region are issued through the C ABI directly from prepared argument tuples: inside a timed
    # region the host does nothing but call hlx_rollout (one C call per <= tape_len steps), so that a short K is not
    # dominated by Python (the driver times as few as 20 steps).
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
        return planned[1]

    def run(total, fused=1):
        env.set_rollout_fused(fused)
        ret = go(plan(total))
        env.set_rollout_fused(1)
        return ret

    def set_form(form):
        if form == "contract":                 # what HlynrVecEnv.step_torch / step_async issue
            env.set_rollout_contract(True, done_list=True)
        elif form == "contract_no_done_list":
            env.set_rollout_contract(True, done_list=False)
        elif form == "terminal_obs_only":
            env.set_rollout_contract(False)
            env.set_rollout_terminal_obs(True)
        else:                                  # single_pass: no optional output
            env.set_rollout_contract(False)

    def windows(form):
        """P launches of `form` with no host synchronisation behind them, then R windows of K launches, each bracketed by
        HIP events on the launch stream (torch events: the launches go to torch's current stream).  Median window."""
        set_form(form)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
        for e in evs:                          # created and recorded once ahead of the measurement (lazy creation is slow)
            e.record()
        pre, win = plan(P), plan(K)
        torch.cuda.synchronize(dev)
        go(pre)
        for r in range(R):
            evs[r].record()
            go(win)
        evs[R].record()
        torch.cuda.synchronize(dev)
        us = sorted(1e3 * evs[r].elapsed_time(evs[r + 1]) / K for r in range(R))
        # `frac` / `achieved` are ALWAYS on SURVEY.md 8(d)'s bytes (508 B base, 604 B v2dr per env-step), whatever the form stores,
        # so that the fraction is comparable across forms and rounds; what this form really moves rides along as *_form_bytes
        b = BYTES_PER_ENV_STEP[args.physics] * n
        bf = (BYTES_PER_ENV_STEP[args.physics] + FORM_BYTES_DELTA[form]) * n
        med = us[len(us) // 2] if len(us) % 2 else 0.5 * (us[len(us) // 2 - 1] + us[len(us) // 2])
        return {"kernel_us": med, "window_us": [round(x, 4) for x in us], "windows": R, "launches_per_window": K, "preroll_launches": P,
                "algorithmic_bytes_per_env_step": b // n, "algorithmic_bytes_per_launch": b,
                "achieved": b / (med * 1e-6) / 1e9, "frac": b / (med * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "form_bytes_per_env_step": bf // n, "frac_form_bytes": bf / (med * 1e-6) / 1e9 / HBM_PEAK_GBS}

    # ---------------------------------------------------------------- the timed region (contract form)
    env.reset_torch()
    run(D, fused=64)                  # desynchronise the episodes (bit-identical to D single-step launches)
    set_form(forms[0])
    run(P)                            # the chip is at its working clocks and the host far ahead of it ...
    run(W)                            # ... then the W untimed warmup steps
    timed = plan(K)
    sync_all()
    # Nothing but the K launches between the two synchronisations: no event is created or recorded in here (round 3: the
    # event pair that used to bracket this region cost the 20-launch window ~50 us of host time -- two lazily created events
    # and their records -- on top of the cold start behind a synchronisation; the kernel clock is taken from the windows below)
    t0 = time.perf_counter()
    last_slot = go(timed)
    sync_all()
    elapsed_local = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_local, dist, red_dev)
    per_rank_ms = [1e3 * x for x in gather_over_ranks(elapsed_local, dist, red_dev)]
    wall_us = 1e6 * elapsed / max(1, K)
    big_last = big_state = None
    if check is not None:             # what the slabs and the oracle will be compared with: taken here, right behind the timed region
        big_last = tuple(x[last_slot].clone() for x in ring)
        big_state = env.get_state()

    # ---------------------------------------------------------------- event-clocked windows, one set per form
    measured = {form: windows(form) for form in forms}
    head = measured[forms[0]]
    set_form("single_pass")

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

    pool_info["auto_resets_computed_in_crowded_waves"] = env.episode_pool_crowded()
    pool_info["auto_resets_computed_inside_step_launches"] = env.episode_pool_misses()     # (all forms of this run; the single-pass form and the fused rollout always compute in place)
    # config 3 and a batch whose state (2.6 GB) defeats the 256 MB Infinity Cache -- each in a FRESH process
    # (tools/extra_point.py: at HBM-bound sizes the step's time depends on the process's whole allocation history)
    extra = None
    if not args.no_extra_points and world == 1:
        extra = []
        env.close()
        torch.cuda.empty_cache()
        for phys, n_x, k_x, d_x in (("v2dr", ENVS_PER_GPU, 1000, 4096), ("base", 4 * 1024 * 1024, 60, 0)):
            cmd = [sys.executable, os.path.join(ROOT, "tools", "extra_point.py"), phys, str(n_x), str(k_x), str(d_x)]
            try:
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
                lines = [ln for ln in res.stdout.splitlines() if ln.startswith("[")]
                extra += json.loads(lines[-1]) if res.returncode == 0 and lines else [{"workload": f"{phys} {n_x}", "error": (res.stderr or res.stdout)[-400:]}]
            except (subprocess.TimeoutExpired, OSError, ValueError) as exc:
                extra.append({"workload": f"{phys} {n_x}", "error": repr(exc)[:400]})

    # ---------------------------------------------------------------- self-check (rank 0; every collective is behind us)
    selfcheck = None
    if check is not None:     # the slabs and the oracle walk through the same D + P + W + K steps, then everything is compared
        check.reset()
        for total in (D, P, W, K):
            for lo, hi in tape_schedule(total, tape_len):
                for j in range(lo, hi):
                    check.step(tape[j])
        selfcheck = check.finish(big_state, big_last)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(rc)

    if rank == 0:
        traffic, traffic_src = measured_traffic(args.physics, n, forms[0])
        roof = {"bound": "hbm", "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "hlx_env_kernel<%s, step>" % variant, "form": forms[0],
                "form_note": "contract = the launch HlynrVecEnv.step_torch issues: terminal observations + every info plane + done list",
                "kernel_us": head["kernel_us"], "kernel_us_clock": f"median of {R} back-to-back windows of {K} launches (HIP events on the "
                                                                    f"launch stream) behind {P} launches of the same form, no host sync in between",
                "window_us": head["window_us"], "algorithmic_bytes_per_env_step": head["algorithmic_bytes_per_env_step"],
                "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
                "bytes_note": "frac = SURVEY.md 8(d) bytes (508 B base / 604 B v2dr per env-step) x envs / kernel_us / peak, for every form; "
                              "frac_form_bytes counts what this form stores on top (contract: + 42 B of info planes)",
                "form_bytes_per_env_step": head["form_bytes_per_env_step"], "frac_form_bytes": head["frac_form_bytes"],
                "timed_region": {"launches": K, "wall_us_per_step": wall_us,
                                 "note": "the K wall-clock-timed launches between two host synchronisations (what `value` is computed from): "
                                         "includes the cold start behind a synchronisation -- first launch into an empty queue, last kernel "
                                         "to host -- which a 20-launch region cannot amortise"},
                "frac_from_wall_clock": head["algorithmic_bytes_per_launch"] / (wall_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        for form in forms[1:]:
            roof[form] = measured[form]
        if "single_pass" in roof and isinstance(roof["single_pass"], dict) and pool_info["fill_interval_steps"] > 0 and pool_info["load_schedule"] == 2:
            roof["single_pass"]["note"] = ("the form without optional outputs; under the lone-wave schedule with the next-episode pool on it runs the "
                                           "two-pass flow too (a finished lane copies its prepared episode either way), so the key names the form, not the flow")
        line = {
            "metric": "env-steps/sec whole-node, medium scenario, 64k envs/GPU",
            "value": whole_job_throughput(n, K, world, elapsed), "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"medium scenario, {args.physics} physics, {n} envs/GPU, fp32 "
                                   f"(BASELINE.json configs[{1 if args.physics == 'base' else 2}])",
                       "envs_per_gpu": n, "kernel_variant": variant, "launches_per_step": 1, "form": forms[0],
                       "episode_pool": pool_info,
                       "phase": f"steady state: episodes desynchronised by {D} fused-rollout steps, then {P} + {W} untimed steps of the timed form",
                       "sharding": f"{world} x {n} independent envs, no collective in the step",
                       "library_build": "safe (hot constants read from memory: the lint refused the product build)" if env.safe_build else "product"},
            "ranks": {"dist_world_size": dist.get_world_size() if dist is not None else 1, "backend": args.backend if dist is not None else None,
                      "per_rank_ms": per_rank_ms},
            "roofline": roof,
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
