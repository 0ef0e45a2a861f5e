"""`ShardedHlynrVecEnv` (hlynr_intercept_amd/sharded.py): one vector-env object over several devices, one thread.
On this pool there is one GPU per box, so the facade is exercised with two and three shards on device 0 (own streams, event
fork / join): the union of the shards must equal the unsharded batch bit for bit -- tensor API and SB3 numpy API -- which is
what `train_flat_ppo.py:371` needs to reach 8 x 65 536 environments unchanged.  The argument plumbing runs without a GPU."""
import numpy as np
import pytest


def _cfg(over=None):
    from hlynr_intercept_amd.scenarios import scenario_config
    return scenario_config("medium", "base", dict({"max_steps": 17}, **(over or {})))


def test_shard_layout_and_argument_checks_without_a_gpu():
    import torch
    from hlynr_intercept_amd.shard import shard_range
    from hlynr_intercept_amd.sharded import ShardedHlynrVecEnv, _ShardedInfos
    assert [shard_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    with pytest.raises(ValueError):
        ShardedHlynrVecEnv(_cfg(), num_envs=8, devices=[])
    with pytest.raises(ValueError):
        ShardedHlynrVecEnv(_cfg(), num_envs=1, devices=[0, 0])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):      # the shards are HlynrVecEnvs: HIP kernel or nothing
            ShardedHlynrVecEnv(_cfg(), num_envs=8, devices=[0, 0])

    class Part(list):
        def done_items(self):
            return [(i, d) for i, d in enumerate(self) if d.get("done")]
    infos = _ShardedInfos([Part([{"k": 0}, {"k": 1, "done": True}]), Part([{"k": 2}]), Part([{"k": 3, "done": True}, {"k": 4}])], [0, 2, 3, 5])
    assert len(infos) == 5 and [d["k"] for d in infos] == [0, 1, 2, 3, 4] and infos[3]["k"] == 3 and infos[-1]["k"] == 4
    assert [d["k"] for d in infos[1:4]] == [1, 2, 3] and [i for i, _ in infos.done_items()] == [1, 3]
    with pytest.raises(IndexError):
        infos[5]


@pytest.mark.gpu
@pytest.mark.parametrize("n,shards", [(448, 2), (333, 3)])
def test_two_handles_on_one_device_equal_the_unsharded_batch(n, shards):
    import torch
    from hlynr_intercept_amd.sharded import ShardedHlynrVecEnv
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    whole = HlynrVecEnv(_cfg(), num_envs=n, seed=21)
    sh = ShardedHlynrVecEnv(_cfg(), num_envs=n, devices=[0] * shards, seed=21)
    assert sh.num_envs == n and len(sh.shards) == shards and sum(s.num_envs for s in sh.shards) == n
    assert [s._env_id_offset for s in sh.shards] == sh.offsets[:-1] and all(st is not None for st in sh._streams)
    ref = whole.reset_torch()
    parts = sh.reset_torch()
    assert torch.equal(torch.cat(list(parts)), ref)
    g = torch.Generator(device=whole.device).manual_seed(2)
    done_total = 0
    for t in range(45):
        a = torch.rand((n, 6), generator=g, device=whole.device) * 2 - 1
        if t == 20:
            whole.set_training_step_count(2_000_000); sh.env_method("set_training_step_count", 2_000_000)
        obs, rew, term, trunc, info = whole.step_torch(a)
        outs = sh.step_torch(a)                                       # one [N, 6] tensor: sliced per shard
        assert len(outs) == shards
        for k, (o, r, te, tr, inf) in enumerate(outs):
            lo, hi = sh.offsets[k], sh.offsets[k + 1]
            assert torch.equal(o, obs[lo:hi]) and torch.equal(r, rew[lo:hi]) and torch.equal(te, term[lo:hi]) and torch.equal(tr, trunc[lo:hi])
            for key in ("distance", "fuel_used", "flags", "steps"):
                assert torch.equal(inf[key], info[key][lo:hi]), (t, k, key)
            assert torch.equal(inf["interceptor_pos"], info["interceptor_pos"][:, lo:hi])
            d = (te | tr) != 0
            assert torch.equal(inf["terminal_observation"][d], info["terminal_observation"][lo:hi][d])
        done_total += int(((term | trunc) != 0).sum())
    assert done_total >= 2 * n
    st = whole.get_state()
    for k, s in enumerate(sh.shards):
        assert bytes(s.get_state()) == bytes(st)[sh.offsets[k] * len(bytes(st)) // n: sh.offsets[k + 1] * len(bytes(st)) // n]
    # per-shard action tensors are accepted as they are
    outs = sh.step_torch([torch.zeros((s.num_envs, 6), device=s.device) for s in sh.shards])
    whole.step_torch(torch.zeros((n, 6), device=whole.device))
    assert torch.equal(torch.cat([o[0] for o in outs]), whole.obs)
    whole.close(); sh.close()


@pytest.mark.gpu
def test_sharded_sb3_numpy_api_concatenates_on_the_host():
    from hlynr_intercept_amd.sharded import ShardedHlynrVecEnv
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    n = 300
    whole = HlynrVecEnv(_cfg(), num_envs=n, seed=8)
    sh = ShardedHlynrVecEnv(_cfg(), num_envs=n, devices=[0, 0], seed=8)
    o1, o2 = whole.reset(), sh.reset()
    assert o2.shape == (n, 26) and np.array_equal(o1, o2)
    assert len(sh.reset_infos) == len(whole.reset_infos) == n            # reset()'s info (environment.py:595-601), shard by shard
    for i in (0, 149, 150, 299, -1):
        a, b = whole.reset_infos[i], sh.reset_infos[i]
        assert set(a) == set(b) and a["distance"] == b["distance"] and a["radar_detected"] == b["radar_detected"]
        assert np.array_equal(a["missile_pos"], b["missile_pos"]) and np.array_equal(a["interceptor_pos"], b["interceptor_pos"])
    rng = np.random.default_rng(1)
    seen = 0
    for t in range(40):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        x1, r1, d1, i1 = whole.step(a)
        x2, r2, d2, i2 = sh.step(a)
        assert np.array_equal(x1, x2) and np.array_equal(r1, r2) and np.array_equal(d1, d2) and len(i2) == n
        assert [i for i, _ in i2.done_items()] == [i for i, _ in i1.done_items()] == list(np.nonzero(d1)[0])
        for i, info in i2.done_items():
            ref = i1[i]
            assert np.array_equal(info["terminal_observation"], ref["terminal_observation"]) and info["episode"]["l"] == ref["episode"]["l"]
            assert info["episode"]["r"] == ref["episode"]["r"] and info["TimeLimit.truncated"] == ref["TimeLimit.truncated"]
            seen += 1
        j = int(rng.integers(n))
        assert i2[j]["distance"] == i1[j]["distance"] and np.array_equal(i2[j]["missile_pos"], i1[j]["missile_pos"])
        assert sum(1 for v in i2 if v.get("episode") is not None) == int(d1.sum())      # SB3's per-step scan of all infos
    assert seen >= n
    assert sh.get_attr("observation_generator", indices=[0, n - 1])[1].radar_beam_width > 0
    assert len(sh.env_method("get_current_intercept_radius")) == n and sh.seed(5) == [5] * n
    assert sh.env_is_wrapped(object) == [False] * n
    whole.close(); sh.close()
