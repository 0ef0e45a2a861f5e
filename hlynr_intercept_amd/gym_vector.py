"""`gymnasium.vector.VectorEnv` face of the batched intercept environment, returning PyTorch-ROCm tensors.

BASELINE.json's north_star asks for "a drop-in Gymnasium VectorEnv that returns PyTorch-ROCm tensors".  The reference's env is a
`gym.Env` (`rl_system/environment.py:15`) with `observation_space = Box(-2, 1, (26,), float32)` (:192-194),
`action_space = Box(-1, 1, (6,), float32)` (:195-197), `reset(seed=None, options=None) -> (obs, info)` (:353) and
`step(action) -> (obs, reward, terminated, truncated, info)` (:605); `gymnasium.vector` batches exactly that surface:

    envs = HlynrGymVectorEnv(config, num_envs=65536)
    obs, info = envs.reset(seed=0)                       # obs: float32 [N, 26] on the GPU
    obs, reward, terminated, truncated, info = envs.step(actions)      # actions: float32 [N, 6] tensor (or anything torch.as_tensor takes)

* every array of the five-tuple is a torch tensor on the environment's GPU (`terminated` / `truncated` are bool tensors, as
  gymnasium's are bool arrays); nothing crosses PCIe.  The tensors are the environment's own output buffers: the next call
  overwrites them (clone what must survive), exactly as `HlynrVecEnv.step_torch` documents;
* auto-reset happens INSIDE the step launch (gymnasium >= 1.0 calls this `AutoresetMode.SAME_STEP`; 0.29's vector envs did the
  same): for an environment that finished, `obs[i]` is the first observation of its next episode, and the last observation
  of the finished one is `info["final_obs"][i]` (0.29 spelling: `info["final_observation"][i]`), rows flagged by the bool mask
  `info["_final_obs"]`.  `info["final_info"]` is the step's own info -- the info planes hold the post-step, pre-reset values
  the reference's `step()` returns (environment.py:829-857);
* `info` follows gymnasium.vector's dict-of-arrays convention: every key of the reference's info dict that callers read
  (environment.py:829-857) maps to a tensor with one entry per environment, built LAZILY on first access (a bool key costs a
  small elementwise kernel; a training loop that reads none of them pays nothing);
  `info["episode"] = {"r", "l"}` with mask `info["_episode"]` are `RecordEpisodeStatistics`-style episode statistics of the
  environments that finished in this step;
* `single_observation_space` / `single_action_space` are the reference's two boxes, `observation_space` / `action_space` their
  batched forms, `num_envs`, `call` / `get_attr` / `set_attr` forward to the batch (gymnasium.vector's method names), `close()`.

Where gymnasium is importable the class IS a `gymnasium.vector.VectorEnv` (isinstance checks of wrappers and trainers pass);
in the build image gymnasium is absent and the same class stands on `object` (tests/test_gym_vector_gpu.py runs both ways,
with a stand-in package in the pattern of tests/test_sb3_subclass_gpu.py).  All compute is `HlynrVecEnv`'s: one fused HIP
kernel launch per step; there is no CPU fallback.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Callable, Dict, Optional

import numpy as np

from . import _lib
from .vec_env import HlynrVecEnv, _box

try:  # optional: absent in the build image
    from gymnasium.vector import VectorEnv as _GymVectorEnv
except Exception:  # pragma: no cover - exercised where gymnasium is missing
    _GymVectorEnv = object

try:
    from gymnasium.vector import AutoresetMode as _AutoresetMode      # gymnasium >= 1.0
    _SAME_STEP: Any = _AutoresetMode.SAME_STEP
except Exception:  # pragma: no cover
    _SAME_STEP = "same_step"


class LazyTensorInfo(Mapping):
    """gymnasium.vector-style `info` (dict of per-environment arrays) whose entries are produced on first access."""

    def __init__(self, makers: Dict[str, Callable[[], Any]]):
        self._makers, self._cache = makers, {}

    def __getitem__(self, key):
        if key not in self._cache:
            self._cache[key] = self._makers[key]()       # KeyError for unknown keys, like a dict
        return self._cache[key]

    def __iter__(self):
        return iter(self._makers)

    def __len__(self):
        return len(self._makers)


class HlynrGymVectorEnv(_GymVectorEnv):
    """N intercept environments as one `gymnasium.vector.VectorEnv`, device tensors in and out."""

    metadata = {"render_modes": [], "autoreset_mode": _SAME_STEP}
    render_mode = None
    spec = None
    closed = False

    def __init__(self, config: Optional[Dict[str, Any]] = None, num_envs: int = 16, device: int = 0, seed: int = 0,
                 env_id_offset: int = 0, resolved=None, venv: Optional[HlynrVecEnv] = None):
        """`venv`: adapt an existing HlynrVecEnv instead of creating one (it is closed with this object)."""
        self.venv = venv if venv is not None else HlynrVecEnv(config, num_envs=num_envs, device=device, seed=seed,
                                                              env_id_offset=env_id_offset, resolved=resolved)
        v = self.venv
        self._torch = v._torch
        self.num_envs = v.num_envs
        self.device = v.device
        self.single_observation_space = v.observation_space                      # environment.py:192-194
        self.single_action_space = v.action_space                                # environment.py:195-197
        self.observation_space = _box(-2.0, 1.0, (self.num_envs, _lib.OBS_DIM))  # gymnasium.vector.utils.batch_space of the two
        self.action_space = _box(-1.0, 1.0, (self.num_envs, _lib.ACT_DIM))
        self._term_b = self._torch.zeros(self.num_envs, dtype=self._torch.bool, device=self.device)
        self._trunc_b = self._torch.zeros(self.num_envs, dtype=self._torch.bool, device=self.device)

    # ------------------------------------------------------------------ gymnasium.vector.VectorEnv
    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        """-> (obs [N, 26] tensor, info).  `seed` re-keys the counter-based generator (environment i draws from
        Philox(seed, global id i, ...): per-environment streams, as gymnasium's `seed + i` convention gives); `options` may carry
        the reference's `{'volley_mode', 'volley_size'}` override (environment.py:363-366)."""
        v = self.venv
        if options is not None:
            v._apply_volley_options(options)
        if seed is not None:
            v.seed(seed)
        import time
        v._t_start = time.time()
        obs = v.reset_torch()
        return obs, LazyTensorInfo(self._info_makers(v.info, fresh=True))

    def step(self, actions):
        """-> (obs, reward, terminated, truncated, info), all tensors on the GPU; see the module docstring for the auto-reset
        and info conventions."""
        v, t = self.venv, self._torch
        if not t.is_tensor(actions):
            actions = t.as_tensor(np.asarray(actions, np.float32))
        obs, rew, term, trunc, info = v.step_torch(actions)
        t.ne(term, 0, out=self._term_b)
        t.ne(trunc, 0, out=self._trunc_b)
        return obs, rew, self._term_b, self._trunc_b, LazyTensorInfo(self._info_makers(info, fresh=False))

    def _info_makers(self, info, fresh):
        v, t = self.venv, self._torch
        flags = info["flags"]
        rc = v.rc

        def bit(b):
            return lambda: (flags & b) != 0

        def const(value, dtype):
            return lambda: t.full((self.num_envs,), value, dtype=dtype, device=self.device)

        mk: Dict[str, Callable[[], Any]] = {
            # environment.py:829-857, one tensor per key
            "distance": lambda: info["distance"], "min_distance": lambda: info["min_distance"],
            "fuel_remaining": lambda: info["fuel"], "fuel_used": lambda: info["fuel_used"], "steps": lambda: info["steps"],
            "interceptor_pos": lambda: info["interceptor_pos"].T, "missile_pos": lambda: info["missile_pos"].T,     # [N, 3]
            "intercepted": bit(1), "missile_hit_target": bit(2), "proximity_fuze_triggered": bit(4), "clamped": bit(8),
            "crossed_threshold": bit(16), "radar_detected": bit(32), "ground_radar_detected": bit(64),
            "radar_quality": lambda: t.where((flags & 128) != 0, float(rc.radar_quality), 0.0),                     # :840
            "missiles_intercepted": lambda: (info["missiles"] & 15).to(t.int32), "missiles_remaining": lambda: (info["missiles"] >> 4).to(t.int32),
            "volley_mode": const(bool(rc.volley_mode), t.bool), "volley_size": const(int(rc.volley_size) if rc.volley_mode else 1, t.int32),
            "precision_mode": const(bool(rc.precision_mode), t.bool), "proximity_fuze_enabled": const(bool(rc.proximity_fuze), t.bool),
            "proximity_kill_radius": const(float(rc.proximity_kill_radius), t.float32),
        }
        if "missile_min_distances" in info:
            mk["missile_min_distances"] = lambda: info["missile_min_distances"][:int(rc.volley_size)].T           # [N, K] (:848)
        if fresh:
            return mk
        done = lambda: self._term_b | self._trunc_b                                                                # noqa: E731
        mk.update({
            "final_obs": lambda: v.terminal_obs, "_final_obs": done,                     # gymnasium >= 1.0, AutoresetMode.SAME_STEP
            "final_observation": lambda: v.terminal_obs, "_final_observation": done,     # gymnasium 0.29
            "final_info": lambda: self._final_info(mk), "_final_info": done,
            "episode": lambda: {"r": info["episode_return"], "l": info["episode_length"]}, "_episode": done,
            "TimeLimit.truncated": lambda: self._trunc_b & ~self._term_b,
        })
        return mk

    @staticmethod
    def _final_info(mk):
        # the step's info planes hold the post-step, PRE-reset values (what the reference's step() returned for the episode
        # that ended): the final info of a finished environment is this step's info
        return LazyTensorInfo({k: f for k, f in mk.items() if not k.startswith(("final_", "_final_"))})

    def close(self, **kwargs):
        if not self.closed:
            self.venv.close()
            self.closed = True

    def close_extras(self, **kwargs):          # gymnasium.vector.VectorEnv.close() calls this hook
        self.venv.close()

    # gymnasium.vector's names for what SB3 calls env_method / get_attr / set_attr
    def call(self, name: str, *args, **kwargs):
        return tuple(self.venv.env_method(name, *args, **kwargs))

    def get_attr(self, name: str):
        return tuple(self.venv.get_attr(name))

    def set_attr(self, name: str, values):
        self.venv.set_attr(name, values[0] if isinstance(values, (list, tuple)) else values)

    # reference env methods callers reach through `envs.call(...)` or directly
    def set_training_step_count(self, step_count: int):
        self.venv.set_training_step_count(step_count)

    def get_current_intercept_radius(self) -> float:
        return self.venv.get_current_intercept_radius()

    @property
    def unwrapped(self):
        return self

    def __repr__(self):
        return f"HlynrGymVectorEnv(num_envs={self.num_envs}, device={self.device}, kernel={self.venv.kernel_variant!r})"
