"""The drop-in VecEnv facade on a real GPU: SB3 VecEnv contract, lazy infos, curriculum hook,
state export/injection, done-list compaction, size-independent invariants at BASELINE.json's full size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env(n, physics="base", over=None, seed=3, offset=0):
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    return HlynrVecEnv(scenario_config("medium", physics, over), num_envs=n, seed=seed, env_id_offset=offset)


def test_sb3_vecenv_contract_and_auto_reset():
    env = _env(200, over={"max_steps": 30})
    obs = env.reset()
    assert obs.shape == (200, 26) and obs.dtype == np.float32
    assert env.observation_space.shape == (26,) and env.action_space.shape == (6,)
    assert env.kernel_variant == "base"
    rng = np.random.default_rng(0)
    ep_returns = np.zeros(200)
    seen_done = 0
    for t in range(65):
        a = rng.uniform(-1, 1, (200, 6)).astype(np.float32)
        env.step_async(a)
        obs, rew, dones, infos = env.step_wait()
        assert obs.shape == (200, 26) and rew.shape == (200,) and dones.dtype == bool and len(infos) == 200
        ep_returns += rew
        for i, info in infos.done_items():
            assert dones[i]
            assert info["terminal_observation"].shape == (26,)
            assert info["episode"]["l"] == 30 or info["episode"]["l"] < 30
            assert info["episode"]["r"] == pytest.approx(ep_returns[i], rel=1e-4, abs=1e-2)   # Monitor 'r'
            assert info["TimeLimit.truncated"] == (info["episode"]["l"] == 30 and not bool(env.terminated[i].item()))
            ep_returns[i] = 0.0
            seen_done += 1
        if t == 29:
            assert dones.all()                      # max_steps = 30 -> every env truncates together
            st = env.get_state()
            assert all(st[i].steps == 0 for i in range(200))   # ... and was reset in the same launch
        i = int(rng.integers(200))
        assert set(infos[i]) >= {"distance", "intercepted", "fuel_remaining", "TimeLimit.truncated", "min_distance"}
    assert seen_done >= 400
    assert np.all(obs <= 1.0 + 1e-6) and np.all(obs >= -2.0 - 1e-6)
    env.close()


def test_info_keys_the_reference_callers_read():
    """train_hrl_pretrain.py:180-198 / inference.py:535-592 index infos[i] with these keys."""
    env = _env(8)
    env.reset()
    obs, rew, dones, infos = env.step(np.zeros((8, 6), np.float32))
    info = infos[3]
    for key in ("distance", "intercepted", "missile_hit_target", "fuel_remaining", "fuel_used", "clamped", "interceptor_pos",
                "missile_pos", "steps", "radar_detected", "radar_quality", "min_distance", "crossed_threshold",
                "volley_mode", "volley_size", "missiles_intercepted", "missiles_remaining", "missile_min_distances",
                "TimeLimit.truncated"):
        assert key in info, key
    assert info["interceptor_pos"].shape == (3,) and info["missile_pos"].shape == (3,) and info["steps"] == 1
    assert info["missile_min_distances"] == [info["distance"]]           # environment.py:848 outside volley mode
    st = env.get_state()
    assert np.allclose(info["interceptor_pos"], list(st[3].int_pos)) and np.allclose(info["missile_pos"], list(st[3].mis_pos))
    assert abs(float(np.linalg.norm(info["missile_pos"] - info["interceptor_pos"])) - info["distance"]) < 1e-2
    # iteration yields cheap read-only views (what SB3's per-step `for info in infos: info.get("episode")` touches)
    views = list(infos)
    assert len(views) == 8 and views[3].get("episode") is None and "terminal_observation" not in views[3]
    assert views[3]["distance"] == info["distance"] and dict(views[3]).keys() == info.keys()
    env.close()


def test_radar_debug_key_and_episode_files(tmp_path):
    """inference.py:535-548 logs info['radar_debug'] (environment.py:842) next to the two positions; the key is opt-in."""
    import json
    import os

    from hlynr_intercept_amd.episode_log import VecEpisodeRecorder
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    plain = _env(4)
    plain.reset()
    assert "radar_debug" not in plain.step(np.zeros((4, 6), np.float32))[3][0]
    plain.close()
    n = 70   # two workgroups, the second one partial
    env = HlynrVecEnv(scenario_config("medium", "v2", {"max_steps": 25}), num_envs=n, seed=3, radar_debug=True)
    env.reset()
    rec = VecEpisodeRecorder(str(tmp_path), indices=(1, 69), run_name="gpu")
    rng = np.random.default_rng(0)
    reasons = set()
    for t in range(60):
        a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
        rec.on_step(a, rew, dones, infos)
        for i in (0, 1, 64, 69):
            info = infos[i]
            rd = info["radar_debug"]
            assert rd["onboard"]["position"] == info["interceptor_pos"].tolist()
            assert rd["onboard"]["detected"] == info["radar_detected"] and rd["ground"]["detected"] == info["ground_radar_detected"]
            assert abs(rd["onboard"]["range_to_target"] - info["distance"]) <= 1e-3 * max(1.0, info["distance"])
            step_obs = info["terminal_observation"] if dones[i] else obs[i]
            assert rd["fusion"]["datalink_quality"] == float(step_obs[24]) and rd["fusion"]["fusion_confidence"] == float(step_obs[25])
            assert abs(np.linalg.norm(rd["onboard"]["forward_vector"]) - 1.0) < 1e-4
            if info["steps"] < env.rc.onboard_delay:      # the delay line is still filling (core.py:576-582)
                assert rd["onboard"]["detection_reason"] == "sensor_delay_initialization" and not rd["onboard"]["detected"]
            reasons.add(rd["onboard"]["detection_reason"])
            reasons.add(rd["ground"]["detection_reason"])
    rec.close()
    assert {"detected", "sensor_delay_initialization"} <= reasons and "unknown" in reasons
    assert len(rec.results) == 4 and all(r["steps"] == 25 for r in rec.results)     # two episodes per watched env
    d = os.path.join(rec._logs[69].log_dir, "episodes")
    assert sorted(os.listdir(d)) == ["ep_0000.jsonl", "ep_0001.jsonl", "ep_0002.jsonl"]
    with open(os.path.join(d, "ep_0001.jsonl")) as f:
        rows = [json.loads(line) for line in f]
    assert rows[0]["type"] == "header" and rows[-1]["type"] == "footer" and rows[-1]["metrics"]["steps"] == 25
    assert [r["entity_id"] for r in rows[1:-1]] == ["interceptor", "missile", "radar"] * 25
    assert rows[3]["state"]["ground"]["max_range"] == 20000.0
    env.close()


def test_library_loaded_before_torch_still_finds_the_gpu():
    """`__graft_entry__.build()` loads libhlx.so before anything imported torch; `smoke()` may follow in the same process.
    PyTorch-ROCm ships its own HIP runtime: the loader must make sure only one copy serves the process (_lib.load)."""
    import os
    import subprocess
    import sys

    code = ("from hlynr_intercept_amd import _lib; lib = _lib.load(); import numpy as np; "
            "from hlynr_intercept_amd.vec_env import HlynrVecEnv; e = HlynrVecEnv(num_envs=8); e.reset(); "
            "o, r, d, i = e.step(np.zeros((8, 6), np.float32)); assert o.shape == (8, 26); e.close(); print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_closed_environment_fails_loudly():
    """A destroyed handle must not be dereferenced: calls after close() raise instead of reading freed memory."""
    env = _env(8)
    env.reset()
    assert env.kernel_variant
    env.close()
    env.close()      # idempotent
    with pytest.raises(RuntimeError):
        env.kernel_variant
    with pytest.raises(RuntimeError, match="null handle"):
        env.step(np.zeros((8, 6), np.float32))
    with pytest.raises(RuntimeError, match="null"):
        env.curriculum()


def test_curriculum_hook_and_attrs():
    env = _env(8)
    assert env.get_current_intercept_radius() == 100.0
    assert env.get_attr("observation_generator")[0].radar_beam_width == 120.0
    env.env_method("set_training_step_count", 1_000_000)
    assert env.get_current_intercept_radius() == pytest.approx(52.5)
    env.env_method("set_training_step_count", 6_500_000)
    og = env.get_attr("observation_generator")[0]
    assert og.radar_beam_width == pytest.approx(90.0) and og.onboard_detection_reliability == 1.0
    env.reset()
    st = env.get_attr("interceptor_state", indices=[0, 3])
    assert st[0]["position"].shape == (3,) and st[1]["fuel"] == 100.0
    with pytest.raises(AttributeError):
        env.env_method("render")
    env.close()


def test_state_roundtrip_and_done_list():
    import torch
    env = _env(300, physics="v2dr", over={"max_steps": 20})
    env.reset_torch()
    g = torch.Generator().manual_seed(1)
    for t in range(50):
        a = (torch.rand((300, 6), generator=g) * 2 - 1).to(env.device)
        obs, rew, term, trunc, info = env.step_torch(a, want_done_list=True)
        n_done = int(info["n_done"].item())
        expect = torch.nonzero((term | trunc) != 0).flatten().cpu().numpy()
        got = np.sort(info["done_idx"][:n_done].cpu().numpy())
        assert np.array_equal(got, expect)                      # ballot/popcount compaction == nonzero()
    st = env.get_state()
    snap = bytes(st)
    env.set_state(st)
    assert bytes(env.get_state()) == snap                      # export -> inject -> export is the identity
    o1 = env.step_torch(a)[0].clone()
    env.set_state(st)                                           # rewind ... the clock moved, so rings shift,
    env.close()                                                 # but injection itself must not fail
    assert torch.isfinite(o1).all()


def test_invariants_at_full_size():
    """BASELINE.json config 2 size (65 536 envs): properties that do not need the oracle."""
    import torch
    env = _env(65536, over={"max_steps": 150}, seed=11)
    env.reset_torch()
    g = torch.Generator(device=env.device).manual_seed(0)
    fuel_prev = torch.full((65536,), 100.0, device=env.device)
    n_done = 0
    for t in range(320):
        a = torch.rand((65536, 6), generator=g, device=env.device) * 2 - 1
        obs, rew, term, trunc, info = env.step_torch(a)
        done = (term | trunc) != 0
        n_done += int(done.sum())
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert float(obs.max()) <= 1.0 + 1e-6 and float(obs.min()) >= -2.0 - 1e-6
        fuel = info["fuel"]
        assert bool(((fuel <= fuel_prev + 1e-6) | done).all())  # fuel never increases within an episode
        fuel_prev = torch.where(done, torch.full_like(fuel, 100.0), fuel)
        assert bool((info["min_distance"] <= info["distance"] + 1e-3)[~done].all())
    st = env.get_state()
    q = np.array([list(st[i].int_quat) for i in range(0, 65536, 97)])
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-5)         # unit quaternions
    P = np.array([list(st[i].kf_P) for i in range(0, 65536, 97)])
    assert np.all(P[:, 0] > 0) and np.all(P[:, 3] > 0)                     # covariance diagonal stays positive
    steps = np.array([st[i].steps for i in range(0, 65536, 97)])
    assert steps.max() < 150 and n_done >= 2 * 65536                        # every env truncated at least twice
    env.close()


@pytest.mark.parametrize("physics,n,volley", [("base", 1000, 0), ("v2dr", 333, 0), ("v2", 64, 0), ("base", 200, 3)])
def test_fused_rollout_is_bit_identical_to_single_step_launches(physics, n, volley):
    """hlx_set_rollout_fused(k): k steps per launch with the state held in registers must reproduce the
    one-launch-per-step rollout bit for bit (same Philox keys, same arithmetic), including auto-resets
    (max_steps 40 forces several per env), a chunk size that does not divide T, a partial tail block and
    the rotation of the output slots."""
    import torch
    T, slots = 131, 7
    over = {"max_steps": 40}
    if volley:   # generic kernel instantiation; the volley groups travel through the arena every step
        over.update(volley_mode=True, volley_size=volley)
    ref, fus = _env(n, physics, over, seed=5), _env(n, physics, over, seed=5)
    g = torch.Generator(device=ref.device).manual_seed(1)
    tape = torch.rand((T, n, 6), generator=g, device=ref.device) * 2 - 1
    o0, o1 = ref.reset_torch().clone(), fus.reset_torch().clone()
    assert torch.equal(o0, o1)
    fus.set_rollout_fused(32)
    # first half in one call, second half in another: the clock and the ring slots carry across launches
    outs = []
    for env in (ref, fus):
        acc = []
        for lo, hi in ((0, 70), (70, T)):
            o, r, te, tr = env.rollout_torch(tape[lo:hi], slots)
            acc.append([x.clone() for x in (o, r, te, tr)])
        outs.append(acc)
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    sa, sb = ref.get_state(), fus.get_state()
    assert bytes(sa) == bytes(sb)                                        # full logical state, rings included
    assert max(sa[i].steps for i in range(n)) < 40                       # episodes did end (and restart) inside the window
    # and both keep stepping identically through the ordinary step entry point
    a = tape[0]
    ra, rb = ref.step_torch(a), fus.step_torch(a)
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
    ref.close(); fus.close()


def test_maximum_size_batch_steps_and_auto_resets():
    """1 048 576 environments (16x BASELINE.json's per-GPU size, 16 384 workgroups): the arena / ring address
    arithmetic in 64-bit, every lane live, auto-reset and the done list at scale."""
    import torch
    n = 1 << 20
    env = _env(n, over={"max_steps": 3}, seed=2)
    obs = env.reset_torch()
    assert obs.shape == (n, 26) and torch.isfinite(obs).all()
    g = torch.Generator(device=env.device).manual_seed(0)
    for t in range(4):
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        obs, rew, term, trunc, info = env.step_torch(a, want_done_list=True)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        n_done = int(info["n_done"].item())
        assert n_done == int(((term | trunc) != 0).sum())
        if t == 2:
            assert n_done == n                                             # max_steps 3: every env truncates together
            idx = info["done_idx"][:n_done].to(torch.int64)
            assert int(torch.unique(idx).numel()) == n                     # each env listed exactly once
        elif t == 3:
            assert n_done == 0                                             # all restarted, then stepped once
    env.close()


@pytest.mark.parametrize("pattern", ["lu", "luu", "ulu", "fused"])
def test_done_list_survives_interleaved_unlisted_steps(pattern):
    """The done counter of vec step t is armed by launch t - 1 whatever that launch was: a listed step (`want_done_list`)
    may follow any number of unlisted ones -- plain `step_torch`, `rollout_torch`, the fused rollout -- and still starts
    its count from zero (round-1 advisor finding: only consecutive listed steps were covered)."""
    import torch
    n = 300
    env = _env(n, physics="base", over={"max_steps": 7})      # every env truncates every 7th step: dense done lists
    env.reset_torch()
    g = torch.Generator(device=env.device).manual_seed(4)
    tape = torch.rand((5, n, 6), generator=g, device=env.device) * 2 - 1

    def listed():
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        obs, rew, term, trunc, info = env.step_torch(a, want_done_list=True)
        n_done = int(info["n_done"].item())
        expect = torch.nonzero((term | trunc) != 0).flatten().cpu().numpy()
        assert n_done == len(expect), (n_done, len(expect))
        assert np.array_equal(np.sort(info["done_idx"][:n_done].cpu().numpy()), expect)
        return n_done

    total = 0
    for rep in range(30):
        for c in pattern if pattern != "fused" else "l":
            if c == "l":
                total += listed()
            else:
                env.step_torch(torch.rand((n, 6), generator=g, device=env.device) * 2 - 1)
        if pattern == "fused":
            env.set_rollout_fused(1 + rep % 4)
            env.rollout_torch(tape[: 1 + rep % 5], 2)
    assert total > 0
    env.close()


def test_seed_rekeys_the_rng_and_successive_resets_differ():
    """SB3 `VecEnv.seed` / `env_method('seed', s)` (scripts/compare_policies.py:150) re-key Philox: the same seed gives
    the same episodes on one env object, a different seed different ones; and two `reset()`s at one clock value do not
    replay the same spawns (the reference's global generator moves on between resets)."""
    import torch
    n = 128
    env = _env(n, physics="v2dr", over={"max_steps": 30})
    g = torch.Generator(device=env.device).manual_seed(0)
    tape = torch.rand((40, n, 6), generator=g, device=env.device) * 2 - 1

    def episode(seed):
        env.seed(seed)
        t0 = int(env._lib.hlx_vec_steps(env._h))
        o0 = env.reset_torch().clone()
        outs = [env.step_torch(tape[k])[0].clone() for k in range(3)]
        return t0, o0, outs

    _, a0, _ = episode(5)
    o_again = env.reset_torch().clone()
    assert not torch.equal(a0, o_again)                       # second reset at the same clock: new spawns
    # same seed at a later clock: different draws (counter = clock); same (seed, clock) on a fresh object: identical
    other = _env(n, physics="v2dr", over={"max_steps": 30}, seed=5)
    b0 = other.reset_torch().clone()
    fresh = _env(n, physics="v2dr", over={"max_steps": 30}, seed=99)
    fresh.env_method("seed", 5)
    c0 = fresh.reset_torch().clone()
    assert torch.equal(b0, c0)
    for k in range(3):
        assert torch.equal(other.step_torch(tape[k])[0], fresh.step_torch(tape[k])[0])
    fresh.seed(6)
    assert not torch.equal(fresh.reset_torch(), c0)
    env.close(); other.close(); fresh.close()


@pytest.mark.parametrize("physics,over", [("base", {}), ("v2dr", {}), ("config", {}), ("config", {"volley_mode": True, "volley_size": 3}),
                                          ("v2", {"observation_mode": "los_frame"})])
def test_both_load_schedules_give_identical_bits(physics, over):
    """`hlx_set_load_schedule`: the lone-wave (2), the small-batch (1) and the large-batch (0) instantiation of the step
    kernel are the same arithmetic with the Kalman / ring loads -- and, in the first, the pool entry of a finished lane --
    issued at different points; outputs and the full state must agree bit for bit, through auto-resets (episodes shorter
    and longer than the pool's fill interval) and a partial tail block."""
    import torch
    n, T = 777, 150
    o = dict(over, max_steps=60)
    a_env, b_env, c_env = (_env(n, physics, o, seed=21) for _ in range(3))
    a_env.set_load_schedule(1); b_env.set_load_schedule(0); c_env.set_load_schedule(2)
    c_env.set_episode_pool(8)
    assert (a_env.load_schedule, b_env.load_schedule, c_env.load_schedule) == (1, 0, 2)
    assert torch.equal(a_env.reset_torch(), b_env.reset_torch()) and torch.equal(a_env.obs, c_env.reset_torch())
    g = torch.Generator(device=a_env.device).manual_seed(2)
    for t in range(T):
        act = torch.rand((n, 6), generator=g, device=a_env.device) * 2 - 1
        ra, rb, rc = a_env.step_torch(act), b_env.step_torch(act), c_env.step_torch(act)
        for x, y, z in zip(ra[:4], rb[:4], rc[:4]):
            assert torch.equal(x, y) and torch.equal(x, z), t
        assert torch.equal(ra[4]["terminal_observation"], rb[4]["terminal_observation"])
        assert torch.equal(ra[4]["terminal_observation"], rc[4]["terminal_observation"])
    assert bytes(a_env.get_state()) == bytes(b_env.get_state()) == bytes(c_env.get_state())
    a_env.set_load_schedule(-1)
    assert a_env.load_schedule == 2                                         # 777 envs: at most one wave per SIMD
    a_env.close(); b_env.close(); c_env.close()


def _baked_presets():
    from hlynr_intercept_amd.build import BAKED
    return BAKED


@pytest.mark.parametrize("name,scenario,physics,over", _baked_presets(), ids=[b[0] for b in _baked_presets()])
def test_baked_presets_are_selected_and_equal_the_runtime_constant_path(name, scenario, physics, over):
    """The shipped scenario presets run step-kernel instantiations whose configuration constants are compile-time literals
    (hlx_baked_gen.h).  hlx_create must select them for exactly those configurations, and they must reproduce the ordinary
    instantiation (constants fetched at run time; forced with HLX_NO_BAKED) bit for bit -- outputs, terminal observations and
    the full state -- in both load schedules and through the fused rollout.  Any override drops back to the ordinary path."""
    import os
    import torch
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    n, T = 700, 260
    cfg = scenario_config(scenario, physics, over)
    baked = HlynrVecEnv(cfg, num_envs=n, seed=31)
    assert baked.kernel_baked == name
    os.environ["HLX_NO_BAKED"] = "1"
    try:
        plain = HlynrVecEnv(cfg, num_envs=n, seed=31)
    finally:
        del os.environ["HLX_NO_BAKED"]
    assert plain.kernel_baked == "" and plain.kernel_variant == baked.kernel_variant
    other = HlynrVecEnv(scenario_config(scenario, physics, dict(over or {}, max_steps=1999)), num_envs=8, seed=31)
    assert other.kernel_baked == ""
    other.close()
    assert torch.equal(baked.reset_torch(), plain.reset_torch())
    g = torch.Generator(device=baked.device).manual_seed(8)
    # start mid-episode states that reach every branch quickly: a few hundred steps of a fused rollout on both
    tape = torch.rand((64, n, 6), generator=g, device=baked.device) * 2 - 1
    for env in (baked, plain):
        env.set_rollout_fused(32)
        for _ in range(12):
            env.rollout_torch(tape, 2)
        env.set_rollout_fused(1)
    assert bytes(baked.get_state()) == bytes(plain.get_state())
    for t in range(T):
        if t == T // 2:
            baked.set_load_schedule(0); plain.set_load_schedule(0)
        act = torch.rand((n, 6), generator=g, device=baked.device) * 2 - 1
        ra, rb = baked.step_torch(act), plain.step_torch(act)
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y), t
        assert torch.equal(ra[4]["terminal_observation"], rb[4]["terminal_observation"])
    assert bytes(baked.get_state()) == bytes(plain.get_state())
    baked.close(); plain.close()


def test_info_radar_quality_follows_the_onboard_delay_line():
    """environment.py:840 <- core.py:536,584: info['radar_quality'] is the configured quality whenever a delayed onboard
    sample exists (detected or not) and 0.0 only while the delay line is filling -- 3 samples with physics v2 (30 ms),
    immediately without sensor delays (base physics)."""
    n = 8
    env = _env(n, physics="v2", over={"max_steps": 50})
    env.reset()
    seen = []
    for t in range(8):
        _, _, _, infos = env.step(np.zeros((n, 6), np.float32))
        seen.append(infos[0]["radar_quality"])
    assert seen[:2] == [0.0, 0.0] and all(q == env.rc.radar_quality for q in seen[2:]), seen   # step k holds k+1 samples; delay 3
    env.close()
    env = _env(n, physics="base")
    env.reset()
    _, _, _, infos = env.step(np.zeros((n, 6), np.float32))
    assert infos[3]["radar_quality"] == env.rc.radar_quality
    env.close()


def test_single_observation_pass_equals_the_two_pass_form():
    """Without a terminal-observation (or info) consumer the step kernel respawns finished lanes BEFORE its single
    observation pass; with one it observes the terminal state first and the new episode in a second pass.  Same outputs,
    same state, bit for bit -- and the terminal observations a rollout delivers through hlx_set_rollout_terminal_obs are
    the ones hlx_step delivers."""
    import torch
    n, T = 900, 200
    a_env, b_env, c_env = (_env(n, "v2dr", {"max_steps": 23}, seed=8) for _ in range(3))
    g = torch.Generator(device=a_env.device).manual_seed(3)
    tape = torch.rand((T, n, 6), generator=g, device=a_env.device) * 2 - 1
    for e in (a_env, b_env, c_env):
        e.reset_torch()
    o, r, te, tr = (x.clone() for x in a_env.rollout_torch(tape, T))              # single pass
    b_env.set_rollout_terminal_obs(True)
    o2, r2, te2, tr2 = (x.clone() for x in b_env.rollout_torch(tape, T))          # two passes, rollout entry point
    assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(te, te2) and torch.equal(tr, tr2)
    assert bytes(a_env.get_state()) == bytes(b_env.get_state())
    n_done = 0
    for t in range(T):                                                            # two passes, hlx_step
        oc, rc_, tec, trc, info = c_env.step_torch(tape[t])
        assert torch.equal(oc, o[t]) and torch.equal(rc_, r[t]) and torch.equal(tec, te[t]) and torch.equal(trc, tr[t]), t
        n_done += int(((tec | trc) != 0).sum())
    done_last = ((te[T - 1] | tr[T - 1]) != 0)
    assert torch.equal(b_env.terminal_obs[done_last], c_env.terminal_obs[done_last])
    assert n_done > 5 * n and bytes(a_env.get_state()) == bytes(c_env.get_state())
    for e in (a_env, b_env, c_env):
        e.close()


# ----------------------------------------------------------------------------------------------
# round 3: the contract form through hlx_rollout, the caller-owned done counter, reset epochs, reset(options=...)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("physics,volley", [("base", False), ("v2dr", False), ("config", True)])
def test_rollout_contract_form_issues_exactly_the_hlx_step_launch(physics, volley):
    """`hlx_set_rollout_outputs` + `hlx_rollout` (what bench.py times) against `hlx_step` called step by step (what
    `HlynrVecEnv.step_torch` issues): observations, rewards, flags, terminal observations, every info plane, the done
    list, its length in the caller-owned counter and the final state, bit for bit, across several auto-resets."""
    import torch
    over = {"max_steps": 23}
    if volley:
        over.update(volley_mode=True, volley_size=3)
    n, T = 700, 90
    a_env, b_env = _env(n, physics=physics, over=over, seed=31), _env(n, physics=physics, over=over, seed=31)
    a_env.reset_torch(); b_env.reset_torch()
    g = torch.Generator(device=a_env.device).manual_seed(8)
    tape = torch.rand((T, n, 6), generator=g, device=a_env.device) * 2 - 1
    b_env.set_rollout_contract(True, done_list=True)
    planes = [k for k in a_env.info if k not in ("episode_return", "episode_length")]
    seen_done = 0
    for t in range(T):
        obs, rew, term, trunc, info = a_env.step_torch(tape[t], want_done_list=True)
        o, r, te, tr = b_env.rollout_torch(tape[t:t + 1], 1)
        assert torch.equal(obs, o[0]) and torch.equal(rew, r[0]) and torch.equal(term, te[0]) and torch.equal(trunc, tr[0]), t
        done = (term | trunc) != 0
        nd = int(info["n_done"].item())
        assert nd == int(done.sum()) == int(b_env.n_done_pair[int(b_env._lib.hlx_vec_steps(b_env._h)) & 1].item()), t
        seen_done += nd
        assert torch.equal(torch.sort(a_env.done_idx[:nd]).values, torch.sort(b_env.done_idx[:nd]).values)
        for k in planes:
            assert torch.equal(a_env.info[k], b_env.info[k]), (t, k)
        if nd:
            assert torch.equal(a_env.terminal_obs[done], b_env.terminal_obs[done]), t
            for k in ("episode_return", "episode_length"):
                assert torch.equal(a_env.info[k][done], b_env.info[k][done]), (t, k)
    assert seen_done >= 3 * n
    assert bytes(a_env.get_state()) == bytes(b_env.get_state())
    # ... and without the outputs the rollout is the single-pass form again: same observations, rewards, flags, state
    b_env.set_rollout_contract(False)
    obs, rew, term, trunc, info = a_env.step_torch(tape[0])
    o, r, te, tr = b_env.rollout_torch(tape[0:1], 1)
    assert torch.equal(obs, o[0]) and torch.equal(rew, r[0]) and torch.equal(term, te[0]) and torch.equal(trunc, tr[0])
    assert bytes(a_env.get_state()) == bytes(b_env.get_state())
    a_env.close(); b_env.close()


def test_fuel_used_accumulates_like_the_reference():
    """info['fuel_used'] = `total_fuel_used` (environment.py:566, 834, 886): a float32 running sum restarted by every reset;
    checked here against a float32 host accumulation of the same per-step consumption (the oracle comparison is in
    test_gpu_parity: bit for bit in every fixture and free-running case)."""
    import torch
    n = 512
    env = _env(n, physics="v2", over={"max_steps": 17}, seed=6)
    env.reset_torch()
    g = torch.Generator(device=env.device).manual_seed(2)
    for t in range(60):
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        obs, rew, term, trunc, info = env.step_torch(a)
        fu, fuel = info["fuel_used"].cpu().numpy(), info["fuel"].cpu().numpy()
        steps = info["steps"].cpu().numpy()
        assert np.all(fu > 0) and np.all(fu <= 100.0 * 1.01)
        assert np.all(fu[steps == 1] <= 0.04)                        # restarted with the episode
        assert np.all(np.abs(fu - (100.0 - fuel)) <= 2e-3)           # what round 2 reported instead, equal up to rounding
    env.close()


def test_reset_epoch_counts_resets_at_one_clock_value_and_is_shard_independent():
    """The advisor's scenario: a masked reset on ONE shard followed by a full reset must equal the unsharded
    run when every shard sees the same hlx_reset calls; the epoch restarts once a step has advanced the clock; a caller
    that skips shards with an empty mask slice can keep them in step with hlx_set_reset_epoch."""
    import torch
    from hlynr_intercept_amd import _lib as hl
    n, cut = 384, 128
    whole = _env(n, seed=9)
    lo, hi = _env(cut, seed=9, offset=0), _env(n - cut, seed=9, offset=cut)
    mask = torch.zeros(n, dtype=torch.uint8, device=whole.device)
    mask[5:40] = 1                                        # only environments of the FIRST shard
    for e in (whole, lo, hi):
        e.reset_torch()
    assert torch.equal(whole.obs, torch.cat([lo.obs, hi.obs]))
    assert whole._lib.hlx_get_reset_epoch(whole._h) == 1
    whole.reset_torch(mask); lo.reset_torch(mask[:cut]); hi.reset_torch(mask[cut:])      # every shard sees the masked reset
    whole.reset_torch(); lo.reset_torch(); hi.reset_torch()
    assert torch.equal(whole.obs, torch.cat([lo.obs, hi.obs]))
    first = whole.obs.clone()
    # a caller that skipped the second shard (its mask slice is all zero) sets the epoch instead
    whole2, lo2, hi2 = _env(n, seed=9), _env(cut, seed=9, offset=0), _env(n - cut, seed=9, offset=cut)
    for e in (whole2, lo2, hi2):
        e.reset_torch()
    whole2.reset_torch(mask); lo2.reset_torch(mask[:cut])
    hl.check(hi2._lib.hlx_set_reset_epoch(hi2._h, lo2._lib.hlx_get_reset_epoch(lo2._h)))
    whole2.reset_torch(); lo2.reset_torch(); hi2.reset_torch()
    assert torch.equal(whole2.obs, torch.cat([lo2.obs, hi2.obs])) and torch.equal(whole2.obs, first)
    # two resets with no step between them start different episodes; after a step the epoch is 0 again
    assert not torch.equal(whole.reset_torch().clone(), first)
    whole.step_torch(torch.zeros((n, 6), device=whole.device))
    assert whole._lib.hlx_get_reset_epoch(whole._h) == 0
    with pytest.raises(hl.HlxError, match="16 bits"):
        hl.check(whole._lib.hlx_set_reset_epoch(whole._h, 1 << 16))
    for e in (whole, lo, hi, whole2, lo2, hi2):
        e.close()


def test_reset_options_switch_volley_mode_like_the_reference():
    """environment.py:363-366: `reset(options={'volley_mode': ..., 'volley_size': ...})` overrides the configuration from
    that reset on.  Volley mode is a code-generation flag here, so the facade re-creates its handle with the other
    instantiation; the result must be the environment one gets by configuring volley mode up front."""
    import torch
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    n = 256
    env = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", "config")), num_envs=n, seed=4)
    assert env.kernel_variant == "config"
    env.reset()
    ref = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", "config", {"volley_mode": True, "volley_size": 3})), num_envs=n, seed=4)
    o_ref = ref.reset()
    o = env.reset(options={"volley_mode": True, "volley_size": 3})
    assert env.kernel_variant == "config-volley" and env.rc.volley_size == 3
    assert np.array_equal(o, o_ref)
    a = torch.zeros((n, 6), device=env.device)
    for _ in range(5):
        x, y = env.step_torch(a), ref.step_torch(a)
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
        assert torch.equal(x[4]["missile_min_distances"], y[4]["missile_min_distances"])
    infos = env.step(np.zeros((n, 6), np.float32))[3]
    assert infos[0]["volley_mode"] is True and infos[0]["volley_size"] == 3 and len(infos[0]["missile_min_distances"]) == 3
    env.reset()                                            # no options: the override stays in force (the reference's attributes do)
    assert env.kernel_variant == "config-volley"
    env.reset(options={})                                  # options without the keys fall back to the CONFIG's values
    assert env.kernel_variant == "config" and not env.rc.volley_mode
    with pytest.raises(ValueError):
        env.reset(options={"volley_mode": True, "volley_size": 9})
    env.close(); ref.close()


# ----------------------------------------------------------------------------------------------
# round 3: the next-episode pool (hlx.h hlx_set_episode_pool)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("physics,over", [("base", {}), ("v2dr", {}), ("config", {"volley_mode": True, "volley_size": 3}),
                                          ("config", {"observation_mode": "los_frame"})])
def test_next_episode_pool_changes_when_work_is_done_never_a_result(physics, over):
    """Finished environments copy a prepared episode (spawn state, ring samples, first observation) out of the pool, or
    compute it inside the step launch when the pool is off / the entry has not been renewed yet.  Pool off, a fill
    after every step, every eight steps, the default interval (longer than the episodes here), the single-pass form and
    the fused rollout: observations, rewards, flags, terminal observations, info planes and the final state (episode counters
    included) are the same bits, across many auto-resets, a curriculum update, a masked explicit reset and a state
    injection on the way."""
    import torch
    n, T = 1300, 120
    over = dict(over, max_steps=19)
    intervals = [0, 1, 8, -1]
    envs = [_env(n, physics=physics, over=over, seed=77) for _ in intervals]
    for e, iv in zip(envs, intervals):
        e.set_episode_pool(iv)
        assert e.episode_pool == (128 if iv < 0 else iv)
        e.reset_torch()
    g = torch.Generator(device=envs[0].device).manual_seed(12)
    tape = torch.rand((T, n, 6), generator=g, device=envs[0].device) * 2 - 1
    mask = (torch.arange(n, device=envs[0].device) % 5 == 0).to(torch.uint8)
    planes = list(envs[0].info)
    done_total = 0
    for t in range(T):
        if t == 40:
            for e in envs:
                e.set_training_step_count(3_000_000)        # moves the radar curriculum where the scenario has one
        if t == 70:
            outs = [e.reset_torch(mask).clone() for e in envs]
            for o in outs[1:]:
                assert torch.equal(outs[0], o)
        if t == 95:
            for e in envs:
                e.set_state(e.get_state())                   # injection: the pool must not serve what it prepared before
        ref = None
        for e in envs:
            obs, rew, term, trunc, info = e.step_torch(tape[t], want_done_list=True)
            done = (term | trunc) != 0
            cur = [obs.clone(), rew.clone(), term.clone(), trunc.clone(), e.terminal_obs[done].clone()] + \
                  [e.info[k].clone() if k not in ("episode_return", "episode_length") else e.info[k][done].clone() for k in planes]
            if ref is None:
                ref = cur
                done_total += int(done.sum())
            else:
                for k, (a, b) in enumerate(zip(ref, cur)):
                    assert torch.equal(a, b), (t, k, e.episode_pool)
    assert done_total > 5 * n
    s0 = bytes(envs[0].get_state())
    for e in envs[1:]:
        assert bytes(e.get_state()) == s0
    st = envs[0].get_state()
    assert sum(st[i].episode for i in range(n)) > 4 * n
    # the pool was really used: with a fill behind every step an entry is missing only for environments that the masked reset
    # or the injection left without one ... and with an interval longer than the episodes it mostly is
    m_every, m_eight, m_long = (e.episode_pool_misses() for e in envs[1:])
    assert envs[0].episode_pool_misses() == 0                 # (pool off: nothing is counted)
    assert m_every < 0.02 * done_total, (m_every, done_total)
    # (max_steps = 19 here: most episodes of this batch end TOGETHER, by truncation -- waves with more finished environments than
    # a wave copies prepared episodes for (four); the rest of such a wave compute theirs in place and are counted apart)
    crowded = envs[1].episode_pool_crowded()
    assert 0 < crowded and crowded + m_every <= done_total, (crowded, m_every, done_total)
    assert envs[0].episode_pool_crowded() == 0
    assert m_every <= m_eight <= m_long and m_long > 0.3 * done_total, (m_every, m_eight, m_long, done_total)
    # single-pass form and fused rollout against the contract form, pool on
    a, b, c = (_env(n, physics=physics, over=over, seed=5) for _ in range(3))
    for e in (a, b, c):
        e.reset_torch()
    b.set_rollout_fused(8)
    outs_a = [x.clone() for x in a.rollout_torch(tape, T)]
    outs_b = [x.clone() for x in b.rollout_torch(tape, T)]
    for x, y in zip(outs_a, outs_b):
        assert torch.equal(x, y)
    for t in range(T):
        obs, rew, term, trunc, _ = c.step_torch(tape[t])
        assert torch.equal(obs, outs_a[0][t]) and torch.equal(rew, outs_a[1][t]) and torch.equal(term, outs_a[2][t]), t
    assert bytes(a.get_state()) == bytes(b.get_state()) == bytes(c.get_state())
    for e in envs + [a, b, c]:
        e.close()


def test_fuel_used_is_environment_state_in_every_form_of_the_step():
    """`total_fuel_used` (environment.py:204, 566, 886) lives in the arena (hlx_env_state.fuel_used), not in a caller plane
    (round-3 advisor finding): it survives get_state -> set_state mid-episode, and it advances in the forms of the step that
    pass no info plane at all -- fused rollouts and plain rollouts -- so that the next step_torch reports the reference's
    running sum, bit for bit the one an uninterrupted step_torch run reports."""
    import torch
    n, T = 300, 40
    over = {"max_steps": 29}
    ref, a, b, c = (_env(n, physics="v2", over=over, seed=5) for _ in range(4))
    g = torch.Generator(device=ref.device).manual_seed(3)
    tape = torch.rand((T + 1, n, 6), generator=g, device=ref.device) * 2 - 1
    for e in (ref, a, b, c):
        e.reset_torch()
    want = None
    for t in range(T + 1):
        want = [x.clone() for x in ref.step_torch(tape[t])[:4]] + [ref.info[k].clone() for k in ("fuel_used", "fuel", "steps", "flags")]
    want_state = bytes(ref.get_state())
    # (a) checkpoint in the middle of the episodes: export, overwrite the arena with another run's state, restore, go on
    for t in range(T // 2):
        a.step_torch(tape[t])
    snap = a.get_state()
    assert any(s.fuel_used > 0 for s in snap) and all(abs((100.0 - s.fuel) - s.fuel_used) < 2e-3 for s in snap)
    for t in range(7):
        a.step_torch(tape[t])                                   # wander off ...
    a.set_state(snap)                                           # ... and come back: the clock has moved, so only the state is compared
    st = a.get_state()
    assert [s.fuel_used for s in st] == [s.fuel_used for s in snap]
    # (b) T steps as ONE fused rollout (no info plane, state in registers), then a step that reports
    b.set_rollout_fused(16)
    b.rollout_torch(tape[:T].contiguous(), 4)
    b.set_rollout_fused(1)
    # (c) T steps as a plain rollout without the contract outputs (no info plane), then a step that reports
    c.rollout_torch(tape[:T].contiguous(), 4)
    for e in (b, c):
        got = [x.clone() for x in e.step_torch(tape[T])[:4]] + [e.info[k].clone() for k in ("fuel_used", "fuel", "steps", "flags")]
        for x, y in zip(want, got):
            assert torch.equal(x, y)
        assert bytes(e.get_state()) == want_state
    for e in (ref, a, b, c):
        e.close()


def test_packed_info_words_equal_the_separate_planes_and_exclude_them():
    """hlx_info_soa.packed (three 16-byte words per environment) against the nine separate planes of the same C ABI: same
    values, bit for bit, every step across auto-resets, volley mode included; both at once are refused."""
    import ctypes as C
    import torch
    from hlynr_intercept_amd import _lib as hl
    for physics, over in (("base", {"max_steps": 21}), ("config", {"max_steps": 21, "volley_mode": True, "volley_size": 3})):
        n, T = 200, 50
        a, b = _env(n, physics=physics, over=over, seed=11), _env(n, physics=physics, over=over, seed=11)
        dev = a.device
        sep = dict(distance=torch.zeros(n, device=dev), min_distance=torch.zeros(n, device=dev), fuel=torch.zeros(n, device=dev),
                   fuel_used=torch.zeros(n, device=dev), flags=torch.zeros(n, dtype=torch.uint8, device=dev),
                   missiles=torch.zeros(n, dtype=torch.uint8, device=dev), interceptor_pos=torch.zeros((3, n), device=dev),
                   missile_pos=torch.zeros((3, n), device=dev), steps=torch.zeros(n, dtype=torch.int32, device=dev))
        soa = hl.HlxInfoSoa(episode_return=b.info["episode_return"].data_ptr(), episode_length=b.info["episode_length"].data_ptr(),
                            **{k: v.data_ptr() for k, v in sep.items()})
        both = hl.HlxInfoSoa(packed=b.info_packed.data_ptr(), distance=sep["distance"].data_ptr())
        a.reset_torch(); b.reset_torch()
        g = torch.Generator(device=dev).manual_seed(1)
        p = b._step_ptrs
        act0 = torch.zeros((n, 6), device=dev)
        rc = b._lib.hlx_step(b._h, act0.data_ptr(), p[0], p[1], p[2], p[3], p[4], None, None, C.byref(both), b._stream())
        assert rc == -1 and b"packed" in b._lib.hlx_last_error()
        dones = 0
        for t in range(T):
            act = torch.rand((n, 6), generator=g, device=dev) * 2 - 1
            obs, rew, term, trunc, info = a.step_torch(act)
            hl.check(b._lib.hlx_step(b._h, act.data_ptr(), p[0], p[1], p[2], p[3], p[4], None, None, C.byref(soa), b._stream()))
            assert torch.equal(obs, b.obs) and torch.equal(rew, b.reward) and torch.equal(term, b.terminated)
            for k, v in sep.items():
                assert torch.equal(info[k], v), (physics, t, k)
            word = a.info_packed.view(torch.int32)[2, :, 3]
            assert torch.equal((word >> 16) & 1, term.to(torch.int32)) and torch.equal((word >> 17) & 1, trunc.to(torch.int32))
            done = (term | trunc) != 0
            dones += int(done.sum())
            for k in ("episode_return", "episode_length"):
                assert torch.equal(a.info[k][done], b.info[k][done])
        assert dones >= 2 * n
        assert bytes(a.get_state()) == bytes(b.get_state())
        a.close(); b.close()


@pytest.mark.parametrize("ramp", ["beam", "reliability"])
def test_next_episode_pool_under_a_curriculum_that_moves_every_step(ramp):
    """The reference's trainers call set_training_step_count after EVERY step (train_flat_ppo.py:171-177) and the shipped
    radar curriculum ramps the beam width over 3 M steps (config.yaml:85-92).  Round 3 renewed every prepared episode ahead
    of every step of such a ramp (advisor finding).  Now: a beam ramp costs no fill at all -- entries carry the beam test they
    were computed with and are validated against today's threshold, and a new episode looks straight at its missile -- and a
    reliability ramp suspends the pool until the scalars have stood still for 16 steps.  Either way: same bits as no pool."""
    import torch
    n, T = 1300, 150
    over = {"max_steps": 19}
    if ramp == "reliability":
        over.update({"curriculum.radar_curriculum.final_detection_reliability": 0.6,
                     "curriculum.radar_curriculum.reliability_transition_start": 5_000_000,
                     "curriculum.radar_curriculum.reliability_transition_end": 8_000_000,
                     "curriculum.radar_curriculum.final_ground_reliability": 0.7,
                     "curriculum.radar_curriculum.ground_reliability_transition_start": 5_000_000,
                     "curriculum.radar_curriculum.ground_reliability_transition_end": 8_000_000})
    on, off = _env(n, physics="base", over=over, seed=41), _env(n, physics="base", over=over, seed=41)
    off.set_episode_pool(0)
    on.set_episode_pool(8)            # (episodes last 19 steps here: with a fill every 8 steps no environment finishes twice between two fills)
    assert on.load_schedule == 2 and on.episode_pool == 8
    for e in (on, off):
        e.set_training_step_count(4_990_000)
        e.reset_torch()
    g = torch.Generator(device=on.device).manual_seed(6)
    moved = set()
    for t in range(T):
        a = torch.rand((n, 6), generator=g, device=on.device) * 2 - 1
        # ~76 steps inside the ramp 5 M -> 8 M (the scalars move at every step), then they stand still at its end
        step = 4_990_000 + 40_000 * t if t < 100 else 9_000_000
        outs = []
        for e in (on, off):
            e.set_training_step_count(step)
            obs, rew, term, trunc, info = e.step_torch(a, want_done_list=True)
            done = (term | trunc) != 0
            outs.append([obs.clone(), rew.clone(), term.clone(), trunc.clone(), e.terminal_obs[done].clone(), e.info_packed.clone(),
                         e.info["episode_return"][done].clone()])
        for k, (x, y) in enumerate(zip(*outs)):
            assert torch.equal(x, y), (ramp, t, k)
        c = on.curriculum()
        moved.add((round(c["beam_width"], 6), round(c["onboard_reliability"], 6), round(c["ground_reliability"], 6)))
    assert len(moved) > 70                                               # the curriculum really moved at every step of the ramp
    assert bytes(on.get_state()) == bytes(off.get_state())
    st = on.episode_pool_stats()
    resets = T * n // 19
    if ramp == "beam":
        # the initial fill, and partial fills every 128 steps: never a full fill because the beam moved; (nearly) every auto-reset was served
        assert st["full_fills"] == 1 and st["suspended_steps"] == 0 and st["partial_fills"] >= T // 8 - 2, st
        assert st["misses"] <= resets // 20, (st, resets)
    else:
        # suspended while the reliabilities moved (~76 steps + 16 quiet ones), filled once when they had settled
        assert 70 <= st["suspended_steps"] <= 100 and st["full_fills"] == 2, st
    on.close(); off.close()


@pytest.mark.parametrize("physics", ["base", "v2dr"])
def test_reset_returns_the_reference_s_reset_info_and_a_masked_reset_touches_only_its_environments(physics):
    """reset()'s info (environment.py:595-601: missile_pos, interceptor_pos, distance, radar_detected, radar_quality) for every
    environment a reset touches -- `reset_infos` (SB3) and the info planes (hlx_reset_info) -- and nobody else's."""
    import torch
    n = 300
    env = _env(n, physics=physics, over={"max_steps": 50})
    env.reset()
    st = env.get_state()
    assert len(env.reset_infos) == n
    for i in (0, 63, 64, 299):
        ri = env.reset_infos[i]
        assert set(ri) == {"missile_pos", "interceptor_pos", "distance", "radar_detected", "radar_quality"}
        assert np.array_equal(ri["missile_pos"], np.array(st[i].mis_pos[:], np.float32))
        assert np.array_equal(ri["interceptor_pos"], np.array(st[i].int_pos[:], np.float32))
        assert ri["distance"] == float(st[i].prev_distance)
        if physics == "v2dr":      # 30 ms onboard delay line still filling: no delayed sample yet (core.py:576-583)
            assert ri["radar_detected"] is False and ri["radar_quality"] == 0.0
        else:
            assert ri["radar_quality"] == env.rc.radar_quality
    if physics == "base":          # a new episode looks straight at its missile: most are detected at once
        assert sum(env.reset_infos[i]["radar_detected"] for i in range(n)) > n // 2
    g = torch.Generator(device=env.device).manual_seed(1)
    for _ in range(7):
        env.step_torch(torch.rand((n, 6), generator=g, device=env.device) * 2 - 1)
    before = {k: env.info[k].clone() for k in ("distance", "steps", "flags", "fuel_used", "missile_pos")}
    mask = (torch.arange(n, device=env.device) % 3 == 0).to(torch.uint8)
    env.reset_torch(mask)
    m = mask.bool()
    assert torch.equal(env.info["steps"][~m], before["steps"][~m]) and bool((env.info["steps"][m] == 0).all())
    assert torch.equal(env.info["distance"][~m], before["distance"][~m]) and torch.equal(env.info["missile_pos"][:, ~m], before["missile_pos"][:, ~m])
    assert bool((env.info["fuel_used"][m] == 0).all()) and bool((env.info["fuel_used"][~m] > 0).all())
    st = env.get_state()
    for i in (0, 3, 297):
        assert float(env.info["distance"][i]) == float(st[i].prev_distance) and int(st[i].steps) == 0
    env.close()


def test_numpy_path_brings_info_words_over_for_finished_environments_only_and_on_request_for_the_rest():
    """`step_wait()` copies reward / flags / observations, and the info words of FINISHED environments; another environment's info
    is fetched when it is first indexed -- the same values as the device planes -- and refuses to be read for the first time
    once a later step has overwritten them."""
    import torch
    n = 500
    env = _env(n, physics="base", over={"max_steps": 17})
    env.reset()
    rng = np.random.default_rng(2)
    kept = None
    for t in range(40):
        obs, rew, dones, infos = env.step(rng.uniform(-1, 1, (n, 6)).astype(np.float32))
        dist, steps, flags = env.info["distance"].cpu().numpy(), env.info["steps"].cpu().numpy(), env.info["flags"].cpu().numpy()
        for i, info in infos.done_items():                    # compact rows
            assert info["distance"] == float(dist[i]) and info["steps"] == int(steps[i]) and info["intercepted"] == bool(flags[i] & 1)
            assert np.array_equal(info["missile_pos"], env.info["missile_pos"][:, i].cpu().numpy())
        if not dones.all():                                     # (at max_steps the whole batch finishes together)
            live = int(np.nonzero(~dones)[0][0])
            li = infos[live]                                    # fetched now
            assert li["distance"] == float(dist[live]) and li["steps"] == int(steps[live]) and "episode" not in li
            assert np.array_equal(li["interceptor_pos"], env.info["interceptor_pos"][:, live].cpu().numpy())
        if t == 20:
            kept = infos
    assert kept is not None
    done_then = [i for i, _ in kept.done_items()]
    fresh = _env(8)                                             # an `infos` whose live rows were never fetched ...
    fresh.reset()
    a = np.zeros((8, 6), np.float32)
    _, _, _, old = fresh.step(a)
    fresh.step(a)
    with pytest.raises(RuntimeError):
        old[0]                                                  # ... is first read after a later step: refused, not answered with that step's values
    fresh.close()
    if done_then:
        assert kept[done_then[0]]["episode"]["l"] > 0           # finished environments' infos stay valid
    env.close()
