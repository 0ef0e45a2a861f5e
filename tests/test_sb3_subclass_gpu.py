"""Where stable-baselines3 is installed the batched env and the on-device wrappers must BE SB3 `VecEnv`s (PPO wraps
anything else into a DummyVecEnv of one).  SB3 is absent from the build image, so a stand-in with the abstract surface
of SB3 2.x's `VecEnv` (same abstract methods, same constructor bookkeeping) is injected in a fresh interpreter."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import abc, sys, types
    import numpy as np
    sb3 = types.ModuleType("stable_baselines3"); common = types.ModuleType("stable_baselines3.common")
    vec = types.ModuleType("stable_baselines3.common.vec_env")

    class VecEnv(abc.ABC):                       # abstract surface of stable_baselines3.common.vec_env.base_vec_env.VecEnv (2.x)
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
            self.reset_infos = [{} for _ in range(num_envs)]
            self._seeds = [None for _ in range(num_envs)]
            self._options = [{} for _ in range(num_envs)]
            try:
                render_modes = self.get_attr("render_mode")
            except AttributeError:
                render_modes = [None for _ in range(num_envs)]
            self.render_mode = render_modes[0]
        @abc.abstractmethod
        def reset(self): ...
        @abc.abstractmethod
        def step_async(self, actions): ...
        @abc.abstractmethod
        def step_wait(self): ...
        @abc.abstractmethod
        def close(self): ...
        @abc.abstractmethod
        def get_attr(self, attr_name, indices=None): ...
        @abc.abstractmethod
        def set_attr(self, attr_name, value, indices=None): ...
        @abc.abstractmethod
        def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...
        @abc.abstractmethod
        def env_is_wrapped(self, wrapper_class, indices=None): ...
        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()

    vec.VecEnv = VecEnv
    sb3.common = common; common.vec_env = vec
    sys.modules.update({"stable_baselines3": sb3, "stable_baselines3.common": common, "stable_baselines3.common.vec_env": vec})

    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    from hlynr_intercept_amd.wrappers import VecFrameStack, VecNormalize
    env = HlynrVecEnv(scenario_config("medium", "base", {"max_steps": 9}), num_envs=32, seed=1)
    assert isinstance(env, VecEnv) and env.render_mode is None and len(env.reset_infos) == 32
    wrapped = VecNormalize(VecFrameStack(env, n_stack=4), norm_reward=False)
    assert isinstance(wrapped, VecEnv) and wrapped.observation_space.shape == (104,) and wrapped.num_envs == 32
    obs = wrapped.reset()
    seen = 0
    for t in range(20):
        obs, rew, dones, infos = wrapped.step(np.zeros((32, 6), np.float32))     # VecEnv.step -> step_async + step_wait
        assert obs.shape == (32, 104)
        for i in np.nonzero(dones)[0]:                                            # what PPO.collect_rollouts touches
            assert infos[i].get("terminal_observation").shape == (104,)
            assert isinstance(infos[i].get("TimeLimit.truncated", False), bool) and "episode" in infos[i]
            seen += 1
    assert seen >= 32
    wrapped.env_method("set_training_step_count", 10)
    assert wrapped.get_attr("observation_generator")[0].radar_beam_width > 0
    wrapped.close()
    print("sb3-subclass ok")
''')


def test_env_and_wrappers_are_sb3_vecenvs_when_sb3_is_present():
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sb3-subclass ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
