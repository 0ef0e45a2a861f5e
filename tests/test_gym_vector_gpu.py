"""`HlynrGymVectorEnv` (hlynr_intercept_amd/gym_vector.py): the gymnasium.vector.VectorEnv face of the batch
(BASELINE.json north_star; reference surface rl_system/environment.py:15, 192-197, 353, 605).  gymnasium is absent from the
build image: the same checks run once on the plain class and once, in a fresh interpreter, against a stand-in with the
abstract surface of gymnasium 1.x's `VectorEnv` / `AutoresetMode` (pattern of tests/test_sb3_subclass_gpu.py)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECKS = textwrap.dedent('''
    import numpy as np, torch
    from hlynr_intercept_amd.gym_vector import HlynrGymVectorEnv
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    n = 200
    cfg = scenario_config("medium", "base", {"max_steps": 19})
    envs = HlynrGymVectorEnv(cfg, num_envs=n, seed=4)
    plain = HlynrVecEnv(cfg, num_envs=n, seed=4)
    assert envs.num_envs == n and envs.single_observation_space.shape == (26,) and envs.single_action_space.shape == (6,)
    assert envs.observation_space.shape == (n, 26) and envs.action_space.shape == (n, 6)
    assert float(envs.single_observation_space.low[0]) == -2.0 and float(envs.single_action_space.high[0]) == 1.0
    obs, info = envs.reset(seed=123)                                    # (obs, info), keyword-only arguments
    plain.seed(123); ref = plain.reset_torch()
    assert torch.is_tensor(obs) and obs.is_cuda and obs.shape == (n, 26) and torch.equal(obs, ref)
    assert torch.equal(info["distance"], plain.info["distance"]) and info["interceptor_pos"].shape == (n, 3)
    # reset()'s info is the new episode's (environment.py:595-601; hlx_reset_info), not whatever the planes held before
    st = plain.get_state()
    for i in (0, 77, n - 1):
        assert float(info["distance"][i]) == float(st[i].prev_distance) > 0 and int(info["steps"][i]) == 0
        assert info["missile_pos"][i].tolist() == [float(x) for x in st[i].mis_pos] and info["interceptor_pos"][i].tolist() == [float(x) for x in st[i].int_pos]
    assert info["radar_quality"].shape == (n,) and bool((info["radar_quality"] == plain.rc.radar_quality).all())      # base physics: no delay line
    assert info["radar_detected"].dtype == torch.bool and not bool(info["intercepted"].any())
    g = torch.Generator(device=obs.device).manual_seed(0)
    finished = 0
    for t in range(60):
        a = torch.rand((n, 6), generator=g, device=obs.device) * 2 - 1
        out = envs.step(a)
        assert len(out) == 5
        obs, rew, term, trunc, info = out
        o2, r2, te2, tr2, i2 = plain.step_torch(a)
        for x in (obs, rew, term, trunc):
            assert torch.is_tensor(x) and x.is_cuda
        assert term.dtype == torch.bool and trunc.dtype == torch.bool and rew.dtype == torch.float32
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(term, te2 != 0) and torch.equal(trunc, tr2 != 0)
        done = term | trunc
        assert torch.equal(info["_final_obs"], done) and torch.equal(info["_final_observation"], done) and torch.equal(info["_episode"], done)
        if bool(done.any()):                                            # same-step autoreset: the terminal observation rides in info
            assert torch.equal(info["final_obs"][done], plain.terminal_obs[done]) and info["final_obs"] is info["final_observation"]
            assert not torch.equal(info["final_obs"][done], obs[done])
            assert torch.equal(info["episode"]["l"][done], plain.info["episode_length"][done])
            assert torch.equal(info["episode"]["r"][done], plain.info["episode_return"][done])
            assert torch.equal(info["final_info"]["distance"], info["distance"]) and "final_obs" not in info["final_info"]
            assert torch.equal(info["TimeLimit.truncated"], trunc & ~term)
            finished += int(done.sum())
        assert torch.equal(info["fuel_remaining"], plain.info["fuel"]) and torch.equal(info["steps"], plain.info["steps"])
        assert torch.equal(info["intercepted"], (plain.info["flags"] & 1) != 0) and info["intercepted"].dtype == torch.bool
        assert torch.equal(info["missile_pos"], plain.info["missile_pos"].T) and info["missile_pos"].shape == (n, 3)
        assert torch.equal(info["missiles_remaining"], (plain.info["missiles"] >> 4).to(torch.int32))
        assert set(info) >= {"distance", "intercepted", "missile_hit_target", "fuel_remaining", "fuel_used", "clamped", "missile_pos",
                             "interceptor_pos", "steps", "radar_detected", "radar_quality", "volley_mode", "volley_size",
                             "missiles_intercepted", "missiles_remaining", "min_distance", "crossed_threshold", "precision_mode",
                             "proximity_fuze_enabled", "proximity_fuze_triggered", "proximity_kill_radius"}     # environment.py:829-857
    assert finished >= 2 * n
    obs, rew, term, trunc, info = envs.step(np.zeros((n, 6), np.float32))       # array-likes are accepted too
    assert envs.call("get_current_intercept_radius")[0] == plain.get_current_intercept_radius()
    envs.call("set_training_step_count", 1000)
    assert envs.get_attr("training_step_count")[0] == 1000
    envs.close(); envs.close(); plain.close()
    with pytest_raises():
        envs.step(np.zeros((n, 6), np.float32))
''')

RAISES = textwrap.dedent('''
    import contextlib
    @contextlib.contextmanager
    def pytest_raises():
        try:
            yield
        except Exception:
            return
        raise AssertionError("a closed environment stepped")
''')

STANDIN = textwrap.dedent('''
    import enum, sys, types
    gym = types.ModuleType("gymnasium"); vector = types.ModuleType("gymnasium.vector"); spaces = types.ModuleType("gymnasium.spaces")
    import numpy as np

    class Box:                                     # gymnasium.spaces.Box, what the adapter touches
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high = np.full(shape, low, dtype), np.full(shape, high, dtype)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    class AutoresetMode(enum.Enum):                # gymnasium 1.x
        NEXT_STEP = "NextStep"; SAME_STEP = "SameStep"; DISABLED = "Disabled"

    class VectorEnv:                               # gymnasium.vector.VectorEnv 1.x: attribute surface + close() protocol
        metadata = {}; spec = None; render_mode = None; closed = False
        num_envs = None; observation_space = None; action_space = None; single_observation_space = None; single_action_space = None
        def reset(self, *, seed=None, options=None): raise NotImplementedError
        def step(self, actions): raise NotImplementedError
        def close(self, **kwargs):
            if self.closed: return
            self.close_extras(**kwargs); self.closed = True
        def close_extras(self, **kwargs): pass
        @property
        def unwrapped(self): return self

    Box.__module__ = "gymnasium.spaces"
    spaces.Box = Box; vector.VectorEnv = VectorEnv; vector.AutoresetMode = AutoresetMode; gym.vector = vector; gym.spaces = spaces
    sys.modules.update({"gymnasium": gym, "gymnasium.vector": vector, "gymnasium.spaces": spaces})
''')


@pytest.mark.gpu
def test_gym_vector_env_plain_class():
    r = subprocess.run([sys.executable, "-c", RAISES + CHECKS + "\nprint('gym-vector ok')"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gym-vector ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_gym_vector_env_is_a_gymnasium_vector_env_where_gymnasium_is_present():
    tail = textwrap.dedent('''
        from gymnasium.vector import VectorEnv, AutoresetMode
        e = HlynrGymVectorEnv(cfg, num_envs=8)
        assert isinstance(e, VectorEnv) and e.metadata["autoreset_mode"] is AutoresetMode.SAME_STEP and e.unwrapped is e
        assert type(e.single_observation_space).__module__ == "gymnasium.spaces"
        e.close()
        print("gym-vector ok")
    ''')
    r = subprocess.run([sys.executable, "-c", STANDIN + RAISES + CHECKS + tail], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gym-vector ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_lazy_tensor_info_builds_entries_on_first_access_only():
    from hlynr_intercept_amd.gym_vector import LazyTensorInfo
    calls = []
    info = LazyTensorInfo({"a": lambda: calls.append("a") or 1, "b": lambda: calls.append("b") or 2})
    assert set(info) == {"a", "b"} and len(info) == 2 and calls == []
    assert info["a"] == 1 and info["a"] == 1 and calls == ["a"] and "b" in info and info.get("c") is None
    with pytest.raises(KeyError):
        info["c"]
    assert dict(info) == {"a": 1, "b": 2}
