"""On-device `VecFrameStack` and `VecNormalize` for `HlynrVecEnv`.

Drop-in for the two Stable-Baselines3 wrappers the reference's trainers apply
(rl_system/scripts/train_flat_ppo.py:384-399, train_hrl_pretrain.py:367-387):

    from hlynr_intercept_amd.wrappers import VecFrameStack, VecNormalize
    envs = VecFrameStack(envs, n_stack=frame_stack)
    envs = VecNormalize(envs, norm_obs=True, norm_reward=False, clip_obs=10.0, clip_reward=10.0, gamma=gamma)

Same constructor arguments, attributes (`training`, `norm_reward`, `obs_rms`, `ret_rms`, `venv`, ...) and step
semantics as SB3 2.x (restated in oracle/vec_wrappers.py; C ABI and algorithm notes in include/hlx_obs.h).  The work
happens in libhlx.so: the step kernel writes each new observation straight into the pipeline's frame ring, and one
`hlx_obs_push` reduces the batch moments, merges the running statistics and emits the stacked, normalised
[N, n_stack*26] float32 batch.  `VecNormalize(VecFrameStack(env))` collapses into ONE pipeline (stack + normalise in
the same kernels); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import pickle
from typing import Any, Dict, Optional

import numpy as np

from . import _lib
from .vec_env import HlynrVecEnv, _SB3VecEnv, _box


# ---------------------------------------------------------------------------------------------------------------------
# vec_normalize.pkl compatibility (no GPU needed for any of this)
class _Opaque:
    """Stand-in for a class a pickle names but this interpreter cannot import (stable_baselines3 / gymnasium objects
    inside an SB3 VecNormalize pickle): keeps the attributes, nothing else."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:      # (dict, slots-dict) form
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        elif isinstance(state, dict):
            self.__dict__.update(state)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Opaque,), {"__module__": module})


def read_vecnormalize_pickle(path: str) -> Dict[str, Any]:
    """Statistics and settings from a `vec_normalize.pkl`: this module's dict format, or an SB3 `VecNormalize` object
    (attribute names of stable_baselines3/common/vec_env/vec_normalize.py), importable or not."""
    with open(path, "rb") as f:
        obj = _TolerantUnpickler(f).load()
    if isinstance(obj, dict) and obj.get("format") == "hlynr-vecnormalize-v1":
        return obj
    try:
        return {"obs_mean": np.asarray(obj.obs_rms.mean, np.float64), "obs_var": np.asarray(obj.obs_rms.var, np.float64),
                "obs_count": float(obj.obs_rms.count), "ret_mean": float(np.asarray(obj.ret_rms.mean)),
                "ret_var": float(np.asarray(obj.ret_rms.var)), "ret_count": float(obj.ret_rms.count),
                "clip_obs": float(obj.clip_obs), "clip_reward": float(obj.clip_reward), "gamma": float(obj.gamma),
                "epsilon": float(obj.epsilon), "norm_obs": bool(obj.norm_obs), "norm_reward": bool(obj.norm_reward),
                "training": bool(obj.training)}
    except AttributeError as exc:
        raise ValueError(f"{path}: neither this module's format nor an SB3 VecNormalize pickle ({exc})") from exc


def write_vecnormalize_pickle(path: str, state: Dict[str, Any], observation_space, action_space, num_envs: int) -> str:
    """Writes `state` (VecNormalize.state_dict()) as an SB3 `VecNormalize` pickle when stable-baselines3 is importable
    (returns "sb3"), else in this module's dict format (returns "dict")."""
    try:
        from stable_baselines3.common.running_mean_std import RunningMeanStd
        from stable_baselines3.common.vec_env import VecNormalize as _SB3VecNormalize
    except Exception:
        with open(path, "wb") as f:
            pickle.dump(state, f)
        return "dict"

    def rms(mean, var, count, shape):
        r = RunningMeanStd(shape=shape)
        r.mean, r.var, r.count = np.asarray(mean, np.float64).reshape(shape), np.asarray(var, np.float64).reshape(shape), float(count)
        return r

    F = int(np.asarray(state["obs_mean"]).size)
    obj = object.__new__(_SB3VecNormalize)
    # the attribute set of SB3 2.x's VecNormalize.__init__ (+ VecEnvWrapper / VecEnv bookkeeping); __getstate__ drops
    # venv, class_attributes and returns, which therefore only have to exist
    obj.__dict__.update(dict(
        venv=None, class_attributes={}, returns=np.zeros(int(num_envs)), num_envs=int(num_envs),
        observation_space=observation_space, action_space=action_space, render_mode=None,
        # (VecEnv.__init__'s own bookkeeping: a file SB3 2.x wrote carries these too -- tests/golden/sb3/vec_normalize_final.pkl)
        reset_infos=[{} for _ in range(int(num_envs))], _seeds=[None] * int(num_envs), _options=[{} for _ in range(int(num_envs))],
        metadata={"render_modes": []},
        norm_obs_keys=None, obs_rms=rms(state["obs_mean"], state["obs_var"], state["obs_count"], (F,)),
        ret_rms=rms(state["ret_mean"], state["ret_var"], state["ret_count"], ()),
        clip_obs=float(state["clip_obs"]), clip_reward=float(state["clip_reward"]), gamma=float(state["gamma"]),
        epsilon=float(state["epsilon"]), training=bool(state["training"]), norm_obs=bool(state["norm_obs"]),
        norm_reward=bool(state["norm_reward"]), old_obs=np.array([]), old_reward=np.array([])))
    with open(path, "wb") as f:
        pickle.dump(obj, f)
    return "sb3"


class _RmsView:
    """`venv.obs_rms` / `venv.ret_rms`: SB3 RunningMeanStd attributes (`mean`, `var`, `count`), read from the device."""

    def __init__(self, owner, which):
        self._o, self._w = owner, which

    def _get(self):
        mean, var, sc = self._o._get_stats()
        return (mean, var, sc[0]) if self._w == "obs" else (np.float64(sc[1]), np.float64(sc[2]), sc[3])

    mean = property(lambda s: s._get()[0])
    var = property(lambda s: s._get()[1])
    count = property(lambda s: s._get()[2])


class _DeviceObsWrapper(_SB3VecEnv):
    """Shared machinery: one `hlx_obs` pipeline behind one `HlynrVecEnv`.  (An SB3 `VecEnv` where SB3 is installed, so
    that `PPO(..., envs)` takes it as is; the SB3 constructor runs at the end of the concrete classes' __init__.)"""

    def __init__(self, venv, n_stack, norm_obs, norm_reward, training, clip_obs, clip_reward, gamma, epsilon):
        base = venv
        while isinstance(base, _DeviceObsWrapper):
            base = base.venv
        if not isinstance(base, HlynrVecEnv):
            raise TypeError("the on-device wrappers wrap a HlynrVecEnv (or one another)")
        self.venv, self._base = venv, base
        self._torch, self._lib = base._torch, base._lib
        self.num_envs, self.device, self.action_space = base.num_envs, base.device, base.action_space
        self.n_stack = int(n_stack)
        self._cfg = _lib.HlxObsConfig(n_envs=base.num_envs, obs_dim=_lib.OBS_DIM, n_stack=self.n_stack,
                                      device=base.device_index, norm_obs=int(norm_obs), norm_reward=int(norm_reward),
                                      training=int(training), clip_obs=float(clip_obs), clip_reward=float(clip_reward),
                                      gamma=float(gamma), epsilon=float(epsilon))
        self._p = C.c_void_p()
        _lib.check(self._lib.hlx_obs_create(C.byref(self._cfg), C.byref(self._p)))
        self.feature_dim = int(self._lib.hlx_obs_feature_dim(self._p))
        t, n, dev = self._torch, self.num_envs, self.device
        self.stacked = t.zeros((n, self.feature_dim), dtype=t.float32, device=dev)
        self.terminal_stacked = t.zeros((n, self.feature_dim), dtype=t.float32, device=dev)
        self.reward_out = t.zeros(n, dtype=t.float32, device=dev)
        self._orig = None
        self._pending = None
        self._closed = False

    # ------------------------------------------------------------------ plumbing
    def _sb3_init(self):
        if _SB3VecEnv is not object:
            _SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return self._base.env_method(method_name, *args, indices=indices, **kwargs)

    def get_attr(self, attr_name, indices=None):
        return self._base.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self._base.set_attr(attr_name, value, indices)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return self._base.env_is_wrapped(wrapper_class, indices)

    def seed(self, seed=None):
        return self._base.seed(seed)

    def _absorb(self, inner):
        """`VecNormalize(VecFrameStack(env))`: this pipeline does both jobs; the inner wrapper's handle is released."""
        inner._release()

    def _release(self):
        if not self._closed and self._p:
            self._torch.cuda.synchronize(self.device)
            self._lib.hlx_obs_destroy(self._p)
            self._p = C.c_void_p()
            self._closed = True

    def close(self):
        self._release()
        self._base.close()

    def __del__(self):  # pragma: no cover
        try:
            self._release()
        except Exception:
            pass

    def __getattr__(self, name):          # env_method, get_attr, set_training_step_count, curriculum, ... pass through
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._base, name)

    def _mode(self):
        _lib.check(self._lib.hlx_obs_set_mode(self._p, int(self._training), int(self._norm_obs), int(self._norm_reward)))

    def _get_stats(self):
        mean, var = np.zeros(self.feature_dim), np.zeros(self.feature_dim)
        sc = (C.c_double * 4)()
        _lib.check(self._lib.hlx_obs_get_stats(self._p, mean.ctypes.data, var.ctypes.data, C.addressof(sc)))
        return mean, var, [float(x) for x in sc]

    def _set_stats(self, mean, var, scalars):
        mean, var = np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(var, np.float64)
        if mean.shape != (self.feature_dim,) or var.shape != (self.feature_dim,):
            raise ValueError(f"statistics must have shape ({self.feature_dim},)")
        sc = (C.c_double * 4)(*[float(x) for x in scalars])
        _lib.check(self._lib.hlx_obs_set_stats(self._p, mean.ctypes.data, var.ctypes.data, C.addressof(sc)))

    # ------------------------------------------------------------------ device API (torch tensors, no host round trip)
    def reset_torch(self):
        b = self._base
        b.reset_torch(obs_ptr=self._lib.hlx_obs_next_slot(self._p))
        _lib.check(self._lib.hlx_obs_push_reset(self._p, self.stacked.data_ptr(), b._stream()))
        return self.stacked

    def step_torch(self, actions, want_done_list: bool = False):
        """(stacked obs [N, n_stack*26], reward, terminated, truncated, info); info['terminal_observation'] holds the
        stacked (and normalised) terminal observation in the rows of finished environments."""
        b = self._base
        # one FFI crossing per training step: hlx_obs_step issues the step launch (the new frame goes straight into the
        # pipeline's frame ring) and the pipeline's launches behind it
        actions, di, nd = b._step_args(actions, want_done_list)
        rew, term, trunc = b.reward, b.terminated, b.truncated
        b._info_gen += 1
        _lib.check(self._lib.hlx_obs_step(self._p, b._h, actions.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                          b.terminal_obs.data_ptr(), di, nd, C.byref(b._info_soa), self.stacked.data_ptr(),
                                          self.terminal_stacked.data_ptr(), self.reward_out.data_ptr(), b._stream()))
        info = b._step_info(want_done_list)
        info = dict(info)
        info["terminal_observation"] = self.terminal_stacked
        info["original_reward"] = rew
        return self.stacked, self.reward_out, term, trunc, info

    def get_original_obs_torch(self):
        """Un-normalised stacked observations of the last step (VecNormalize.get_original_obs)."""
        if self._orig is None:
            self._orig = self._torch.zeros_like(self.stacked)
        _lib.check(self._lib.hlx_obs_emit(self._p, 0, self._orig.data_ptr(), self._base._stream()))
        return self._orig

    # ------------------------------------------------------------------ SB3 VecEnv API (numpy at the boundary)
    def reset(self):
        import time
        self._base._t_start = time.time()
        obs = self.reset_torch()
        self.reset_infos = self._base.reset_infos = self._base._snapshot_reset_infos()      # reset()'s info (environment.py:595-601)
        return obs.cpu().numpy()

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("step_async() called twice without step_wait()")
        self._pending = self.step_torch(self._base._upload_actions(actions), want_done_list=True)

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError("step_wait() called without step_async()")
        obs, rew, term, trunc, info = self._pending
        self._pending = None
        return self._base._materialise(obs, rew, term, trunc, info, self.terminal_stacked)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


class VecFrameStack(_DeviceObsWrapper):
    """SB3 `VecFrameStack(venv, n_stack)` for the 1-D observation: newest frame last, zero-filled after a reset."""

    def __init__(self, venv, n_stack: int, channels_order: Optional[str] = None):
        if channels_order not in (None, "last"):
            raise ValueError("1-D observations stack along the last axis")
        super().__init__(venv, n_stack, norm_obs=False, norm_reward=False, training=False, clip_obs=10.0, clip_reward=10.0,
                         gamma=0.99, epsilon=1e-8)
        self._training, self._norm_obs, self._norm_reward = False, False, False
        self.observation_space = _box(-2.0, 1.0, (self.feature_dim,))
        self._sb3_init()


class VecNormalize(_DeviceObsWrapper):
    """SB3 `VecNormalize`: running mean/variance normalisation of observations (and optionally rewards)."""

    def __init__(self, venv, training: bool = True, norm_obs: bool = True, norm_reward: bool = True, clip_obs: float = 10.0,
                 clip_reward: float = 10.0, gamma: float = 0.99, epsilon: float = 1e-8, norm_obs_keys: Any = None):
        if norm_obs_keys is not None:
            raise ValueError("norm_obs_keys applies to Dict observation spaces; the intercept observation is a Box")
        n_stack = venv.n_stack if isinstance(venv, VecFrameStack) else 1
        super().__init__(venv, n_stack, norm_obs, norm_reward, training, clip_obs, clip_reward, gamma, epsilon)
        if isinstance(venv, VecFrameStack):
            self._absorb(venv)
        self._training, self._norm_obs, self._norm_reward = bool(training), bool(norm_obs), bool(norm_reward)
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = float(clip_obs), float(clip_reward), float(gamma), float(epsilon)
        self.observation_space = _box(-2.0, 1.0, (self.feature_dim,))   # SB3 keeps the wrapped space
        self.obs_rms, self.ret_rms = _RmsView(self, "obs"), _RmsView(self, "ret")
        self._sb3_init()

    training = property(lambda s: s._training)
    norm_obs = property(lambda s: s._norm_obs)
    norm_reward = property(lambda s: s._norm_reward)

    @training.setter
    def training(self, v):
        self._training = bool(v)
        self._mode()

    @norm_obs.setter
    def norm_obs(self, v):
        self._norm_obs = bool(v)
        self._mode()

    @norm_reward.setter
    def norm_reward(self, v):
        self._norm_reward = bool(v)
        self._mode()

    def normalize_obs(self, obs):
        """clip((obs - mean) / sqrt(var + eps)) with the current statistics; numpy in, float32 numpy out."""
        if not self._norm_obs:
            return obs
        mean, var, _ = self._get_stats()
        return np.clip((np.asarray(obs) - mean) / np.sqrt(var + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def normalize_reward(self, reward):
        if not self._norm_reward:
            return reward
        _, _, sc = self._get_stats()
        return np.clip(np.asarray(reward) / np.sqrt(sc[2] + self.epsilon), -self.clip_reward, self.clip_reward).astype(np.float32)

    def get_original_obs(self):
        return self.get_original_obs_torch().cpu().numpy()

    def get_original_reward(self):
        return self._base.reward.cpu().numpy()

    # ------------------------------------------------------------------ persistence (train_flat_ppo.py:528-531, inference.py:450-477)
    def state_dict(self):
        mean, var, sc = self._get_stats()
        return {"format": "hlynr-vecnormalize-v1", "obs_mean": mean, "obs_var": var, "obs_count": sc[0], "ret_mean": sc[1],
                "ret_var": sc[2], "ret_count": sc[3], "clip_obs": self.clip_obs, "clip_reward": self.clip_reward,
                "gamma": self.gamma, "epsilon": self.epsilon, "norm_obs": self._norm_obs, "norm_reward": self._norm_reward,
                "training": self._training, "n_stack": self.n_stack}

    def save(self, path: str) -> None:
        """`VecNormalize.save` (train_flat_ppo.py:528-531).  Where stable-baselines3 is importable the file is a genuine
        SB3 `VecNormalize` pickle (the reference's `inference.py:450-477` loads it with SB3's own `VecNormalize.load`);
        elsewhere a plain dict with the same numbers.  `load` reads both, with or without SB3 installed."""
        write_vecnormalize_pickle(path, self.state_dict(), self.observation_space, self.action_space, self.num_envs)

    @staticmethod
    def load(path: str, venv) -> "VecNormalize":
        """Counterpart of SB3's `VecNormalize.load(load_path, venv)`: this module's own format or a `vec_normalize.pkl`
        written by SB3 (the reference's trainers) -- the latter also where SB3 itself is not installed."""
        d = read_vecnormalize_pickle(path)
        out = VecNormalize(venv, training=d["training"], norm_obs=d["norm_obs"], norm_reward=d["norm_reward"],
                           clip_obs=d["clip_obs"], clip_reward=d["clip_reward"], gamma=d["gamma"], epsilon=d["epsilon"])
        out._set_stats(np.asarray(d["obs_mean"], np.float64).reshape(-1), np.asarray(d["obs_var"], np.float64).reshape(-1),
                       [d["obs_count"], float(d["ret_mean"]), float(d["ret_var"]), d["ret_count"]])
        return out
