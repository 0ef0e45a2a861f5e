"""`InterceptEnvironment`: ONE environment with the reference's own class name, constructor and `gym.Env` semantics.

The reference's scripts outside the trainers build the environment directly -- `InterceptEnvironment(config)` in
`rl_system/inference.py:406` (offline inference), `hrl/hierarchical_env.py:175-195` (the base environment its HRL wrappers wrap),
`scripts/evaluate_hrl.py:106`, `scripts/debug_pn_guidance.py:51`, `diagnose_radar.py:21`, `helpers/check_missile_trajectory.py:13`
-- and drive it as a plain `gym.Env` (`rl_system/environment.py:15`):

    obs, info = env.reset(seed=None, options=None)                     # environment.py:353
    obs, reward, terminated, truncated, info = env.step(action)        # environment.py:605   (Python float / bools, a dict)

This class is that face over a batch of one on the GPU (`HlynrVecEnv(num_envs=1)`; the same HIP kernel, no CPU path): same
spaces, same info keys (environment.py:829-857, `radar_debug` included: core.py:650-683), the attributes those scripts read
(`interceptor_state`, `missile_state`, `missile_states`, `target_position`, `steps`, `total_fuel_used`, `training_step_count`,
`observation_generator`, `config`, `dt`, `max_steps`, `volley_mode`, `volley_size`) and the two methods (`set_training_step_count`,
`get_current_intercept_radius`).

Two things differ from the reference, both by construction of the batched step:
  * **No auto-reset is visible.**  The kernel starts the next episode inside the launch in which one ends (VecEnv semantics);
    here `step()` returns the TERMINAL observation of the finished episode, as `gym.Env.step` does, and `reset()` draws the
    episode that follows (one reset-only launch, whose info words hold reset()'s own info: hlx_reset_info).  Between the two,
    `interceptor_state` / `missile_state` hold what the step's info keeps of the FINAL state (positions, fuel); the final velocity
    and orientation are not kept and asking for them raises.  Stepping a finished environment without `reset()` continues the
    episode the kernel started -- the reference's behaviour there is undefined.
  * **Random streams.**  `reset(seed=s)` keys the counter-based generator (Philox: seed, environment id, episode / clock) instead of
    `np.random.seed(s)`: the same seed gives the same episodes run after run, not the reference's particular draws (DESIGN.md
    section 2: parity with the reference's arithmetic is established on injected draws).  Without a seed the key comes from the
    operating system's entropy, as an unseeded `np.random` would.

Throughput is not the point of this face (one launch + one synchronisation per step: about 1e4 steps/s against the reference's
2.4-3.2e3 on one core); it exists so that the scripts above run unchanged on the kernel the batch uses.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import numpy as np

from .vec_env import HlynrVecEnv

try:  # the reference subclasses gym.Env (environment.py:15); gymnasium is optional here
    import gymnasium as _gym

    _GymEnv = _gym.Env
except Exception:  # pragma: no cover - gymnasium is absent in the build container
    _GymEnv = object

class _EndState(dict):
    """`interceptor_state` / `missile_state` after the step that ended an episode: the keys the step's info still holds."""

    def __missing__(self, key):
        raise KeyError(f"{key!r} of the state an episode ended in is not kept: the batched step starts the next episode in the same "
                       f"launch (read it before the terminal step, or use info['interceptor_pos'] / info['missile_pos'])")


class InterceptEnvironment(_GymEnv):
    """One intercept environment stepped on the GPU; `rl_system/environment.py:15-859` from the outside."""

    metadata = {"render_modes": []}
    render_mode = None

    def __init__(self, config: Optional[Dict[str, Any]] = None, *, device: int = 0, seed: Optional[int] = None,
                 radar_debug: bool = True):
        """`config`: the reference's `environment` dict (environment.py:20-190; physics / curriculum blocks merged in as
        `scripts/train_hrl_pretrain.py:335-338` does).  `radar_debug=False` drops `info['radar_debug']` and lets the specialised
        kernel variant serve (the reference always fills the key)."""
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self._venv = HlynrVecEnv(config, num_envs=1, device=device, seed=seed, radar_debug=radar_debug)
        v = self._venv
        self.config = v.config                                                   # environment.py:24 (read by hrl/hierarchical_env.py:193-195)
        self.observation_space, self.action_space = v.observation_space, v.action_space          # environment.py:192-197
        rc = v.rc
        self.dt, self.max_steps = float(rc.dt), int(rc.max_steps)                # environment.py:25-26
        self.target_position = np.asarray(rc.target_pos, np.float32)        # environment.py:39
        self.steps, self.total_fuel_used = 0, 0.0                                # environment.py:203-204
        self._needs_reset = True
        self._final = None                       # what is known of the state an episode ended in (see interceptor_state)
        self._actions = np.zeros((1, 6), np.float32)

    # ------------------------------------------------------------------ gym.Env
    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        """-> (obs float32[26], info) (environment.py:353-603).  `options` may carry `{'volley_mode', 'volley_size'}` (:363-366)."""
        v = self._venv
        if seed is not None:
            v.seed(int(seed))
        obs = v.reset(options=options)[0].copy()
        # reset()'s info, environment.py:595-601 -- the words hlx_reset_info wrote for this environment
        flags = int(v.info["flags"][0].item())
        info = {"missile_pos": v.info["missile_pos"][:, 0].cpu().numpy().copy(),
                "interceptor_pos": v.info["interceptor_pos"][:, 0].cpu().numpy().copy(),
                "distance": float(v.info["distance"][0].item()),
                "radar_detected": bool(flags & 32),
                "radar_quality": float(v.rc.radar_quality) if flags & 128 else 0.0}
        self._needs_reset, self._final = False, None
        self.steps, self.total_fuel_used = 0, 0.0
        return obs, info

    def step(self, action):
        """-> (obs, reward, terminated, truncated, info) (environment.py:605-859)."""
        if self._needs_reset:
            raise RuntimeError("InterceptEnvironment.step() called before reset()")
        v = self._venv
        self._actions[0] = np.asarray(action, np.float32).reshape(6)
        obs, rew, dones, infos = v.step(self._actions)
        info = infos[0]
        # environment.py:814-815: both flags can be raised by the same step; SB3's `dones` hides which
        terminated, truncated = bool(infos._h["terminated"][0]), bool(infos._h["truncated"][0])
        info.pop("TimeLimit.truncated", None)            # (SB3's key; the reference's step() has no such key)
        self.steps, self.total_fuel_used = int(info["steps"]), float(info["fuel_used"])
        out = obs[0].copy()
        if dones[0]:
            # gym.Env hands back the observation of the state the episode ended in; what the kernel wrote to `obs` is the first
            # observation of the episode it has already started
            out = np.asarray(info.pop("terminal_observation"), np.float32).copy()
            info.pop("episode", None)                    # (SB3's Monitor key)
            self._final = (_EndState(position=info["interceptor_pos"].copy(), fuel=float(info["fuel_remaining"])),
                           _EndState(position=info["missile_pos"].copy()))
        return out, float(rew[0]), terminated, truncated, info

    def close(self):
        self._venv.close()

    def render(self):                                    # environment.py has no renderer either
        return None

    # ------------------------------------------------------------------ the reference's own methods and attributes
    def set_training_step_count(self, step_count: int):                          # environment.py:269-351
        self._venv.set_training_step_count(int(step_count))

    def get_current_intercept_radius(self) -> float:                             # environment.py:223-234
        return self._venv.get_current_intercept_radius()

    @property
    def training_step_count(self) -> int:
        return int(self._venv.training_step_count)

    @property
    def observation_generator(self):                                             # train_flat_ppo.py:228-232 reads four knobs off it
        return self._venv.get_attr("observation_generator")[0]

    @property
    def interceptor_state(self) -> Dict[str, Any]:                               # environment.py:201 (debug_pn_guidance.py:77-80)
        """{'position', 'velocity', 'orientation', 'fuel'}.  Between the step that ended an episode and the next reset(): the FINAL
        position and fuel (what scripts/eval_terminal_360.py:91-103 reads for the miss distance); the final velocity and orientation
        are gone -- the kernel has started the next episode in their place -- and asking for them raises instead of answering with
        the next episode's."""
        return self._final[0] if self._final is not None else self._current("interceptor_state")

    @property
    def missile_state(self) -> Dict[str, Any]:                                   # environment.py:200
        return self._final[1] if self._final is not None else self._current("missile_state")

    @property
    def missile_states(self) -> List[Dict[str, Any]]:                            # environment.py:44 (volley mode: one dict per missile)
        st = self._state()
        if not self.volley_mode:
            return []
        k = int(self._venv.rc.volley_size)
        flat_p, flat_v = np.array(st.v_pos[:], np.float32).reshape(-1, 3), np.array(st.v_vel[:], np.float32).reshape(-1, 3)
        return [{"position": flat_p[m].copy(), "velocity": flat_v[m].copy(), "active": bool(st.v_active[m])} for m in range(k)]

    @property
    def volley_mode(self) -> bool:
        return bool(self._venv.rc.volley_mode)

    @property
    def volley_size(self) -> int:
        return int(self._venv.rc.volley_size)

    @property
    def unwrapped(self):
        return self

    def __repr__(self):
        return f"InterceptEnvironment(device={self._venv.device}, kernel={self._venv.kernel_variant!r})"

    # ------------------------------------------------------------------ helpers
    def _state(self):
        return self._venv.get_state()[0]

    def _current(self, name):
        return self._venv.get_attr(name, indices=[0])[0]
