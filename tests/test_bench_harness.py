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
