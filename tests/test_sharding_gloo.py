"""Multi-process sharding logic on CPU (gloo, world_size 2): slabs are contiguous and disjoint, the union
of the shards' trajectories equals one big env set bit for bit (stepping backend here = the CPU oracle,
which is allowed in tests), and the benchmark's max-over-ranks / whole-job arithmetic is right."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _noise(n_total, steps):
    """Draws keyed by GLOBAL env id so that a shard sees the same draws as the big set."""
    import oracle.oracle as orc
    rng = np.random.default_rng(2024)
    rn0 = rng.random((n_total, orc.RESET_SLOTS))
    sn = rng.standard_normal((steps, n_total, orc.STEP_SLOTS))
    sn[:, :, [6, 11, 12, 19]] = rng.random((steps, n_total, 4))
    rn = rng.random((steps, n_total, orc.RESET_SLOTS))
    act = rng.uniform(-1, 1, (steps, n_total, 6)).astype(np.float32)
    return rn0, sn, rn, act


def _run(rc, lo, hi, steps, n_total):
    import oracle.oracle as orc
    rn0, sn, rn, act = _noise(n_total, steps)
    ov = orc.OracleVec(rc, hi - lo)
    obs = [ov.reset(rn0[lo:hi]).copy()]
    rew, done = [], []
    for t in range(steps):
        out = ov.step(act[t, lo:hi], sn[t, lo:hi], rn[t, lo:hi])
        obs.append(out["obs"].copy()); rew.append(out["reward"].copy())
        done.append((out["terminated"] | out["truncated"]).copy())
    return np.stack(obs), np.stack(rew), np.stack(done)


def _worker(rank, world, port, n_total, steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.shard import max_over_ranks, shard_range

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rc = resolve_config(scenario_config("medium", "base", {"max_steps": 25}))
    off, cnt = shard_range(n_total, world, rank)
    obs, rew, done = _run(rc, off, off + cnt, steps, n_total)
    slowest = max_over_ranks(1.0 + rank, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (off, cnt, obs, rew, done))
    dist.barrier()
    if rank == 0:
        q.put((slowest, gathered))
    dist.destroy_process_group()


def test_shard_range_partitions():
    from hlynr_intercept_amd.shard import shard_range, whole_job_throughput
    for total, world in ((524288, 8), (10, 3), (7, 8), (0, 2)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (o1, c1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + c1 == o2
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(524288, 8, 3) == (3 * 65536, 65536)
    assert whole_job_throughput(65536, 1000, 8, 2.0) == 65536 * 1000 * 8 / 2.0
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_two_rank_shards_concatenate_to_the_whole_set():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    n_total, steps, world = 48, 60, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    slowest, gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest == 2.0           # max over ranks of (1 + rank)
    gathered.sort(key=lambda g: g[0])
    assert [g[0] for g in gathered] == [0, 24] and [g[1] for g in gathered] == [24, 24]
    rc = resolve_config(scenario_config("medium", "base", {"max_steps": 25}))
    obs, rew, done = _run(rc, 0, n_total, steps, n_total)
    assert np.array_equal(np.concatenate([g[2] for g in gathered], axis=1), obs)
    assert np.array_equal(np.concatenate([g[3] for g in gathered], axis=1), rew)
    assert np.array_equal(np.concatenate([g[4] for g in gathered], axis=1), done)
    assert done.any(), "the case must exercise auto-reset"
