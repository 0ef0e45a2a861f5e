"""bench.py plumbing that does not need a GPU: the self-launch of N ranks, the action-tape schedule, and the config-4
harness logic (policy-in-the-loop rollout + ONE flat gradient all-reduce per minibatch) on 2 gloo ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks_when_not_under_torchrun():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["slowest_rank_fake_time"] == 2.0          # max over ranks of (1 + rank)
    assert d["shards"] == [[0, 65536], [65536, 65536]]


def test_tape_schedule_covers_every_step_once():
    sys.path.insert(0, ROOT)
    import bench
    for total, tape in ((0, 7), (5, 7), (7, 7), (4096, 2000), (6296, 2048)):
        sl = bench.tape_schedule(total, tape)
        assert sum(hi - lo for lo, hi in sl) == total and all(0 <= lo < hi <= tape for lo, hi in sl)


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench_configs as bc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)
    policy = bc.flat_policy(torch)
    params = list(policy.parameters())
    n = 32
    g = torch.Generator().manual_seed(rank)             # different data per rank, identical replicas

    def step_fn(a):                                     # stand-in environment: shapes and dtypes of the device pipeline
        obs = torch.randn((n, 104), generator=g)
        return obs, a.sum(dim=1), torch.zeros(n, dtype=torch.uint8), (torch.rand(n, generator=g) < 0.1).to(torch.uint8)

    sections = bc.Sections(torch)
    obs, kept, n_done = bc.rollout_with_policy(step_fn, torch.randn((n, 104), generator=g), policy, 6, sections, torch, keep=3)
    assert len(kept) == 3 and kept[0][0].shape == (n, 104) and kept[0][1].shape == (n, 6)
    o, a, r = kept[0]
    mean, value = policy(o)
    (((mean - a) ** 2).mean() + ((value - r) ** 2).mean() + policy.log_std.sum()).backward()
    local = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    nbytes = bc.all_reduce_flat_grads(params, dist, world)
    merged = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    q.put((rank, nbytes, float((merged - sum(gathered) / world).abs().max()), sorted(sections.totals_us())))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_harness_logic_on_two_gloo_ranks():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, nbytes, err, names in res:
        assert 1.7e6 < nbytes < 1.9e6                    # ~1.8 MB of fp32 gradients in ONE bucket
        assert err < 1e-6                                # every replica ends with the mean of the ranks' gradients
        assert names == ["env+pipeline", "policy"]


def test_a_rank_that_dies_takes_the_job_down_instead_of_hanging_it():
    """launch_ranks waited for its children one after the other; a rank that failed before the rendezvous left the
    others in init_process_group until the timeout.  Now the parent polls, terminates the siblings and returns the code."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HLX_BENCH_TEST_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu"], env=env, capture_output=True,
                         text=True, timeout=240)
    assert out.returncode == 7, (out.returncode, out.stderr[-1500:])
    assert time.time() - t0 < 120


# ---- on a GPU box: every bench mode as a fresh subprocess, so that none of the harnesses can rot unnoticed
def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _finite(x):
    import math
    return isinstance(x, (int, float)) and math.isfinite(x)


@pytest.mark.gpu
def test_bench_headline_line_small():
    d = _bench("--envs-per-gpu", "4096", "--steps", "20", "--warmup", "5", "--desync", "256", "--preroll", "200", "--no-extra-points",
               "--no-cpu-baseline", "--fused", "8")
    r = d["roofline"]
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["config"]["form"] == "contract" and r["form"] == "contract"
    assert d["selfcheck"]["ok"] and d["selfcheck"]["batch_equals_slabs_bit_for_bit"] and d["selfcheck"]["obs_env_steps_with_diverged_detection"] == 0
    # roofline.frac is on SURVEY.md 8(d)'s 508 B for EVERY form (round-3 review); what a form stores on top is reported beside it
    assert len(r["window_us"]) == 20 and r["algorithmic_bytes_per_env_step"] == 508 and r["form_bytes_per_env_step"] == 550
    assert r["single_pass"]["algorithmic_bytes_per_env_step"] == 508 and r["terminal_obs_only"]["form_bytes_per_env_step"] == 500
    assert abs(r["frac"] - 508 * 4096 / (r["kernel_us"] * 1e-6) / 8e12) < 1e-9 and abs(r["frac_form_bytes"] / r["frac"] - 550 / 508) < 1e-9
    for k in (r["kernel_us"], r["single_pass"]["kernel_us"], r["terminal_obs_only"]["kernel_us"], r["frac"], d["value"], d["ms_per_step"]):
        assert _finite(k) and k > 0
    assert d["ranks"]["dist_world_size"] == 1 and len(d["ranks"]["per_rank_ms"]) == 1


@pytest.mark.gpu
def test_bench_config4_policy_in_the_loop_small():
    d = _bench("--config", "4", "--envs-per-gpu", "4096", "--rollout-steps", "8", "--minibatches", "2", "--desync", "64")
    assert d["n_gpus"] == 1 and d["steps"] == 8 and "configs[3]" in d["metric"]
    assert all(_finite(v) and v > 0 for v in d["shares_us_per_step"].values())
    assert all(_finite(v) for v in d["update_us_per_minibatch"].values()) and 1.7e6 < d["gradient_bucket_bytes"] < 1.9e6
    assert d["ranks"]["dist_world_size"] == 1


@pytest.mark.gpu
def test_bench_config5_volley_hrl_lstm_small():
    d = _bench("--config", "5", "--envs-per-gpu", "4096", "--steps", "20")
    assert d["n_gpus"] == 1 and d["steps"] == 20 and "configs[4]" in d["metric"]
    assert all(_finite(v) and v > 0 for v in d["shares_us_per_step"].values())
    assert d["lstm_state_bytes_per_env"] == 4096 and sum(d["options_now"]) == 4096
    # round 4: this repository's share of the controller step is reported apart from the caller's networks, and the measured
    # run has a MIXED option population (every specialist serves some environments)
    assert set(d["shares_us_per_step"]) >= {"controller_us", "grouping_us", "lstm_state_gather_scatter_us", "specialist_forward_us"}
    assert min(d["options_now"]) > 0 and _finite(d["everything_but_the_specialists_forward_us"])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_over_gloo():
    d = _bench("--gpus", "2", "--single-device", "--backend", "gloo", "--envs-per-gpu", "8192", "--steps", "20", "--warmup", "5", "--desync", "256",
               "--preroll", "200", "--no-extra-points", "--no-cpu-baseline", "--fused", "0", "--forms", "contract")
    assert d["n_gpus"] == 2 and d["ranks"]["dist_world_size"] == 2 and d["ranks"]["backend"] == "gloo" and len(d["ranks"]["per_rank_ms"]) == 2
    assert all(_finite(x) and x > 0 for x in d["ranks"]["per_rank_ms"]) and _finite(d["value"]) and d["selfcheck"]["ok"]
