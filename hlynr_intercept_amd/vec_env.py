"""Drop-in vector environment over the HIP step kernel.

Presents the reference's env surface for N environments at once:

* the Stable-Baselines3 ``VecEnv`` contract the reference's trainers consume
  (rl_system/scripts/train_flat_ppo.py:371-399, train_hrl_pretrain.py:349-387): ``num_envs``,
  ``observation_space``, ``action_space``, ``reset()``, ``step_async()/step_wait()`` with auto-reset,
  ``infos[i]['terminal_observation']``, ``infos[i]['TimeLimit.truncated']``, Monitor-style
  ``infos[i]['episode']``, ``env_method``, ``get_attr``, ``set_attr``, ``seed``, ``close``;
* a gymnasium ``VectorEnv``-style ``step_torch()`` that returns PyTorch-ROCm tensors without any
  host round trip (obs, reward, terminated, truncated, info-dict-of-tensors).

All compute happens in ``libhlx.so`` (one fused HIP kernel per step); torch only owns the I/O
buffers and the stream.  There is no CPU fallback: without a GPU / the HIP library construction fails.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Any, Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import _lib
from .episode_log import radar_debug
from .config import ResolvedConfig, resolve_config

try:  # gymnasium is optional (absent in the build image); only `spaces.Box` is needed
    from gymnasium import spaces as _spaces

    def _box(low, high, shape):
        return _spaces.Box(low=low, high=high, shape=shape, dtype=np.float32)
except Exception:  # pragma: no cover - exercised where gymnasium is missing
    class _Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low = np.full(shape, low, dtype)
            self.high = np.full(shape, high, dtype)
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.flat[0]}, {self.high.flat[0]}, {self.shape}, {self.dtype})"

    def _box(low, high, shape):
        return _Box(low, high, shape)


try:  # Stable-Baselines3 is optional (absent in the build image).  When it is there, the batched env IS an SB3 VecEnv:
    # `PPO("MlpPolicy", envs)` wraps anything that is not an instance of its VecEnv class into a DummyVecEnv of one.
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:  # pragma: no cover - exercised where stable-baselines3 is missing
    _SB3VecEnv = object


_NP_DTYPES: Dict[Any, Any] = {}     # torch dtype -> numpy dtype, filled on first use (torch is imported lazily)


class _ObservationGeneratorView:
    """`get_attr('observation_generator')[i]` (train_flat_ppo.py:224-232): curriculum-controlled radar knobs."""

    def __init__(self, venv):
        self._v = venv

    def _cur(self):
        return self._v.curriculum()

    radar_beam_width = property(lambda s: s._cur()["beam_width"])
    onboard_detection_reliability = property(lambda s: s._cur()["onboard_reliability"])
    ground_detection_reliability = property(lambda s: s._cur()["ground_reliability"])
    measurement_noise_level = property(lambda s: s._cur()["noise_level"])


class LazyInfos(Sequence):
    """`infos` of one vec step: behaves like a list of N dicts, builds a dict only when indexed.

    Building 65 536 Python dicts per step would cost more than the step itself."""

    def __init__(self, n, done_rows, host):
        self._n, self._done, self._h = n, done_rows, host

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        h = self._h
        row = self._done.get(i)
        # a finished environment's values came over with the step (compact rows); anybody else's are fetched when first asked for
        s, j = (h["done"], row) if row is not None else (h["full"](), i)
        flags = int(s["flags"][j])
        d: Dict[str, Any] = {
            "distance": float(s["distance"][j]), "min_distance": float(s["min_distance"][j]),
            "fuel_remaining": float(s["fuel"][j]), "intercepted": bool(flags & 1),
            "missile_hit_target": bool(flags & 2), "proximity_fuze_triggered": bool(flags & 4),
            "clamped": bool(flags & 8), "crossed_threshold": bool(flags & 16), "radar_detected": bool(flags & 32),
            "ground_radar_detected": bool(flags & 64),
            "TimeLimit.truncated": bool(h["truncated"][i] and not h["terminated"][i]),
            # environment.py:844-847
            "volley_mode": h["volley"][0], "volley_size": h["volley"][1],
            "missiles_intercepted": int(s["missiles"][j]) & 15, "missiles_remaining": int(s["missiles"][j]) >> 4,
            # environment.py:836-841 (read by train_hrl_pretrain.py:180-198, inference.py:535-560)
            "interceptor_pos": s["interceptor_pos"][:, j].copy(), "missile_pos": s["missile_pos"][:, j].copy(),
            # fuel_used: the reference's `total_fuel_used` (environment.py:886), accumulated step by step in float32 by the kernel
            "steps": int(s["steps"][j]), "fuel_used": float(s["fuel_used"][j]),
            # environment.py:840 <- core.py:536,584: the configured quality whenever a delayed onboard sample exists
            "radar_quality": float(h["radar_quality"]) if flags & 128 else 0.0,
            # environment.py:848: per-missile closest approach in volley mode, [distance] otherwise
            "missile_min_distances": (s["missile_min_distances"][:h["volley"][1], j].tolist() if "missile_min_distances" in s
                                      else [float(s["distance"][j])]),
            # environment.py:852-856: configuration echoed into every info
            "precision_mode": h["constants"][0], "proximity_fuze_enabled": h["constants"][1],
            "proximity_kill_radius": h["constants"][2],
        }
        rd = h.get("radar")
        if rd is not None:   # environment.py:842
            p = s["radar_planes"]
            d["radar_debug"] = radar_debug(rd["rc"], rd["beam_width"], d["interceptor_pos"], d["missile_pos"], p[0:4, j],
                                           float(p[4, j]), int(p[5, j:j + 1].view(np.int32)[0]), flags, float(p[6, j]),
                                           float(p[7, j]))
        if row is not None:
            d["terminal_observation"] = h["terminal_obs"][row]
            d["episode"] = {"r": float(h["ep_return"][row]), "l": int(h["ep_length"][row]),
                            "t": round(time.time() - h["t_start"], 6)}
        return d

    def __iter__(self):
        """Iteration (SB3's `for idx, info in enumerate(infos): info.get("episode")` runs every step over all N
        environments) yields read-only views that materialise nothing until a key is asked for: environments that did not
        finish answer `.get("episode")` / `.get("terminal_observation")` from the done-row table alone."""
        done = self._done
        for i in range(self._n):
            yield _InfoView(self, i, done.get(i))

    def done_items(self):
        """(env index, info dict) for the envs that finished this step - what callbacks iterate over."""
        return [(i, self[i]) for i in sorted(self._done)]


class _ResetInfos(Sequence):
    """SB3's `VecEnv.reset_infos` after a reset: per environment the dict reset() returns in the reference (environment.py:595-601),
    built when indexed from a host snapshot of the words hlx_reset_info wrote."""

    def __init__(self, packed_host, radar_quality):
        self._pk, self._q = packed_host, float(radar_quality)

    def __len__(self):
        return self._pk.shape[1]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        pk = self._pk.numpy()
        flags = int(pk.view(np.uint8)[2, i, 12])
        return {"missile_pos": pk[2, i, 0:3].copy(), "interceptor_pos": pk[1, i, 0:3].copy(), "distance": float(pk[0, i, 0]),
                "radar_detected": bool(flags & 32), "radar_quality": self._q if flags & 128 else 0.0}


class _InfoView(Mapping):
    """One environment's info as a read-only mapping; `dict(view)` or `infos[i]` gives the plain dict."""

    __slots__ = ("_p", "_i", "_row")
    _DONE_KEYS = ("terminal_observation", "episode")

    def __init__(self, parent, i, row):
        self._p, self._i, self._row = parent, i, row

    def get(self, key, default=None):
        if key in self._DONE_KEYS and self._row is None:
            return default                      # the common case, answered without building anything
        return self._p[self._i].get(key, default)

    def __getitem__(self, key):
        if key in self._DONE_KEYS and self._row is None:
            raise KeyError(key)
        return self._p[self._i][key]

    def __contains__(self, key):
        if key in self._DONE_KEYS:
            return self._row is not None
        return key in self._p[self._i]

    def __iter__(self):
        return iter(self._p[self._i])

    def __len__(self):
        return len(self._p[self._i])


class HlynrVecEnv(_SB3VecEnv):
    """N intercept environments on one MI355X, stepped by one fused HIP kernel per call."""

    metadata = {"render_modes": []}

    def __init__(self, config: Optional[Dict[str, Any]] = None, num_envs: int = 16, device: int = 0, seed: int = 0,
                 env_id_offset: int = 0, resolved: Optional[ResolvedConfig] = None, radar_debug: bool = False):
        """radar_debug: also export what `info['radar_debug']` (environment.py:842) is assembled from, and add that key
        to the info dicts (episode_log.radar_debug); off by default - eight more floats stored per env-step, and the generic kernel variant."""
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("HlynrVecEnv needs a ROCm GPU: the step runs in a HIP kernel and has no CPU fallback")
        self._torch = torch
        self.config = config or {}
        self.rc = resolved if resolved is not None else resolve_config(self.config)
        self.num_envs = int(num_envs)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.observation_space = _box(-2.0, 1.0, (_lib.OBS_DIM,))     # environment.py:192-194
        self.action_space = _box(-1.0, 1.0, (_lib.ACT_DIM,))          # environment.py:195-197
        self._lib = _lib.load()
        self._seed, self._env_id_offset, self._radar_debug = int(seed), int(env_id_offset), bool(radar_debug)
        self._h = C.c_void_p()
        self._closed = True
        self._config_volley = (bool(self.rc.volley_mode), int(self.rc.volley_size))     # what `self.config` says (environment.py:42-43)
        self._open()
        self._pending = None
        self._t_start = time.time()
        self._noise = None
        self.training_step_count = 0
        if _SB3VecEnv is not object:   # SB3's own bookkeeping (reset_infos, _seeds, _options, render_mode)
            _SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def _open(self):
        """Create the handle for `self.rc` and the I/O buffers it writes (also used by `reset(options=...)` when the
        volley settings change: volley mode is a code-generation flag, so the other mode is another handle)."""
        torch, radar_debug = self._torch, self._radar_debug
        self._cfg = _lib.make_hlx_config(self.rc)
        if radar_debug:
            self._cfg.flags |= _lib.F_RADAR_DEBUG      # generic kernel variant: the specialised ones carry no debug export
        self._h = C.c_void_p()
        _lib.check(self._lib.hlx_create(C.byref(self._cfg), self.num_envs, self.device_index, self._seed,
                                        self._env_id_offset, C.byref(self._h)))
        self._closed = False
        self._ro_key = None
        n, dev = self.num_envs, self.device
        self.obs = torch.zeros((n, _lib.OBS_DIM), dtype=torch.float32, device=dev)
        # Per-step scalar outputs live in ONE device slab (256-byte aligned typed views), so that the numpy path brings
        # them to the host with a single copy instead of a dozen.
        f32, u8, i32 = torch.float32, torch.uint8, torch.int32
        _NP_DTYPES.update({f32: np.float32, u8: np.uint8, i32: np.int32})
        # hlx_info_soa.packed (include/hlx.h): the nine standard info keys as three 16-byte words per environment -- what the
        # step kernel stores with three coalesced instructions; the keys below are strided VIEWS of it
        spec = [("reward", f32, (n,)), ("terminated", u8, (n,)), ("truncated", u8, (n,)), ("n_done", i32, (2,)),
                ("packed", f32, (3, n, 4))]
        if self.rc.volley_mode:
            spec.append(("missile_min_distances", f32, (_lib.MAX_VOLLEY, n)))
        if radar_debug:
            spec.append(("radar_debug", f32, (8, n)))
        self._slab_layout, off = {}, 0
        for name, dt, shape in spec:
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._slab_layout[name] = (off, nbytes, dt, shape)
            off += (nbytes + 255) // 256 * 256
        self._slab = torch.zeros(off, dtype=u8, device=dev)
        self._slab_head = self._slab_layout["packed"][0]      # bytes in front of the info words: what every step brings to the host
        self._info_gen = 0                                    # counts the launches that rewrite info words (LazyInfos validity)
        v = {name: self._slab[o:o + nb].view(dt).view(shape) for name, (o, nb, dt, shape) in self._slab_layout.items()}
        self.reward, self.terminated, self.truncated = v["reward"], v["terminated"], v["truncated"]
        # the kernel counts finished environments straight into the slab (element `vec-step clock & 1`): no copy per step
        self._n_done2 = v["n_done"]
        # per-step host cost matters once the step itself is ~9 us: the two one-element views of the counter pair and the
        # addresses of the buffers a step always writes are made once, not on every call
        self._n_done_views = (self._n_done2[0:1], self._n_done2[1:2])
        self._n_done_ptrs = (self._n_done_views[0].data_ptr(), self._n_done_views[1].data_ptr())
        self.n_done = self._n_done_views[0]
        _lib.check(self._lib.hlx_set_done_counter(self._h, self._n_done2.data_ptr()))
        self.terminal_obs = torch.zeros((n, _lib.OBS_DIM), dtype=torch.float32, device=dev)
        self.done_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        pk = self.info_packed = v["packed"]
        self.info = dict(episode_return=torch.zeros(n, device=dev), episode_length=torch.zeros(n, dtype=torch.int32, device=dev),
                         **self._unpack_info(pk, pk.view(i32), pk.view(u8)))
        if self.rc.volley_mode:
            self.info["missile_min_distances"] = v["missile_min_distances"]
        if radar_debug:
            self.info["radar_debug"] = v["radar_debug"]
        self._info_soa = _lib.HlxInfoSoa(episode_return=self.info["episode_return"].data_ptr(),
                                         episode_length=self.info["episode_length"].data_ptr(),
                                         missile_min_distances=self.info["missile_min_distances"].data_ptr() if self.rc.volley_mode else None,
                                         radar_debug=self.info["radar_debug"].data_ptr() if radar_debug else None,
                                         packed=pk.data_ptr())
        self._info_ref = C.byref(self._info_soa)
        # reset()'s info (environment.py:595-601) lands in the same words, for the environments a reset touches (hlx_reset_info)
        self._info_reset = _lib.HlxInfoSoa(packed=self.info_packed.data_ptr())
        self._info_reset_ref = C.byref(self._info_reset)
        self._ptr_done_idx = self.done_idx.data_ptr()
        self._step_ptrs = (self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr(), self.truncated.data_ptr(),
                           self.terminal_obs.data_ptr())
        self._actions_dev = torch.zeros((n, _lib.ACT_DIM), dtype=torch.float32, device=dev)
        self._actions_pin = torch.zeros((n, _lib.ACT_DIM), dtype=torch.float32, pin_memory=True)

    @staticmethod
    def _unpack_info(pk, pk_i32, pk_u8):
        """The reference's info keys (environment.py:829-857) as views of hlx_info_soa.packed [3, N, 4] (float32, and the same
        bytes seen as int32 / as uint8 [3, N, 16]); works on torch tensors and numpy arrays alike.  Word 2's last dword:
        byte 0 = the flag bits (hlx.h), byte 1 = missiles intercepted / remaining nibbles."""
        return dict(distance=pk[0, :, 0], min_distance=pk[0, :, 1], fuel=pk[0, :, 2], fuel_used=pk[0, :, 3],
                    interceptor_pos=pk[1, :, 0:3].T, steps=pk_i32[1, :, 3], missile_pos=pk[2, :, 0:3].T,
                    flags=pk_u8[2, :, 12], missiles=pk_u8[2, :, 13])

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if not self._closed and self._h:
            self._torch.cuda.synchronize(self.device)
            self._lib.hlx_destroy(self._h)
            self._h = C.c_void_p()      # later calls fail with "null handle" instead of touching freed memory
            self._closed = True

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def kernel_variant(self) -> str:
        if self._closed:
            raise RuntimeError("kernel_variant: the environment is closed")
        return self._lib.hlx_kernel_variant(self._h).decode()

    @property
    def kernel_baked(self) -> str:
        """Shipped preset whose constants this handle's step launches carry as literals ('' = fetched at run time)."""
        return self._lib.hlx_kernel_baked(self._h).decode()

    # ------------------------------------------------------------------ torch / gymnasium-vector style API
    def reset_torch(self, mask=None, obs_ptr: Optional[int] = None):
        """Reset all envs (or those where `mask` is non-zero); returns the device obs tensor [N, 26].
        `obs_ptr`: raw device address to write the observations to instead (the frame ring of wrappers.py).
        `self.info` of the environments that were reset holds reset()'s info afterwards (environment.py:595-601: positions, spawn
        distance, the first observation's detections; every other key as a new episode has it)."""
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self._torch.uint8).contiguous()
            mptr = mask.data_ptr()
        self._info_gen += 1
        _lib.check(self._lib.hlx_reset_info(self._h, mptr, obs_ptr if obs_ptr is not None else self.obs.data_ptr(),
                                            self._info_reset_ref, self._stream()))
        return self.obs

    def step_torch(self, actions, want_done_list: bool = False, obs_ptr: Optional[int] = None):
        """One vec step on device tensors.  `actions`: float32 [N, 6] on this device.
        `obs_ptr`: raw device address the observations go to instead of `self.obs` (wrappers.py's frame ring).

        Returns (obs, reward, terminated, truncated, info) - all torch tensors living on the GPU; they are
        overwritten by the next call (clone what must survive)."""
        actions, di, nd = self._step_args(actions, want_done_list)
        p = self._step_ptrs
        self._info_gen += 1
        _lib.check(self._lib.hlx_step(self._h, actions.data_ptr(), obs_ptr if obs_ptr is not None else p[0], p[1], p[2], p[3], p[4],
                                      di, nd, self._info_ref, self._stream()))
        return self.obs, self.reward, self.terminated, self.truncated, self._step_info(want_done_list)

    def _step_args(self, actions, want_done_list):
        """Validated action tensor + the done-list pointers of the step about to be issued (also used by wrappers.py)."""
        t = self._torch
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(actions.shape) != (self.num_envs, _lib.ACT_DIM):
            raise ValueError(f"actions must have shape ({self.num_envs}, {_lib.ACT_DIM}), got {tuple(actions.shape)}")
        # the count of the step about to be issued lands in element (clock & 1) of the slab's counter pair
        k = (int(self._lib.hlx_vec_steps(self._h)) + 1) & 1
        self.n_done = self._n_done_views[k]
        if want_done_list:
            return actions, self._ptr_done_idx, self._n_done_ptrs[k]
        return actions, None, None

    def _step_info(self, want_done_list):
        info = dict(self.info)
        info["terminal_observation"] = self.terminal_obs
        if want_done_list:
            info["done_idx"], info["n_done"] = self.done_idx, self.n_done
        return info

    def rollout_torch(self, action_tape, out_slots: int = 1):
        """T steps from a pre-supplied tape [T, N, 6] with one launch per step issued from C
        (open-loop evaluation / benchmark path).  Returns (obs, reward, terminated, truncated) ring buffers."""
        t = self._torch
        T = int(action_tape.shape[0])
        self._info_gen += 1
        assert tuple(action_tape.shape[1:]) == (self.num_envs, _lib.ACT_DIM) and action_tape.dtype == t.float32
        assert action_tape.is_contiguous() and action_tape.device == self.device
        key = ("ro", out_slots)
        if getattr(self, "_ro_key", None) != key:
            n, dev = self.num_envs, self.device
            self._ro = (t.zeros((out_slots, n, _lib.OBS_DIM), device=dev), t.zeros((out_slots, n), device=dev),
                        t.zeros((out_slots, n), dtype=t.uint8, device=dev),
                        t.zeros((out_slots, n), dtype=t.uint8, device=dev))
            self._ro_key = key
        o, r, te, tr = self._ro
        _lib.check(self._lib.hlx_rollout(self._h, action_tape.data_ptr(), T, out_slots, o.data_ptr(), r.data_ptr(),
                                         te.data_ptr(), tr.data_ptr(), self._stream()))
        return o, r, te, tr

    # ------------------------------------------------------------------ SB3 VecEnv API (numpy at the boundary)
    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        """`gym.Env.reset(seed=None, options=None)` for the whole batch (environment.py:353-366); SB3 calls it bare.
        `seed` re-keys the counter-based generator (as `VecEnv.seed`); `options` may carry the reference's per-reset
        override `{'volley_mode': bool, 'volley_size': int}` (environment.py:363-366), which stays in force for later
        resets, as the reference's attributes do."""
        if options is not None:
            self._apply_volley_options(options)
        if seed is not None:
            self.seed(seed)
        self._t_start = time.time()
        obs = self.reset_torch()
        self.reset_infos = self._snapshot_reset_infos()
        return obs.cpu().numpy()                     # (synchronises: the snapshot has landed too)

    def _snapshot_reset_infos(self):
        """Enqueue the host copy of the words hlx_reset_info wrote (valid after the stream's next synchronisation)."""
        pk_h = self._torch.empty(self.info_packed.shape, dtype=self.info_packed.dtype, device="cpu", pin_memory=True)
        pk_h.copy_(self.info_packed, non_blocking=True)
        return _ResetInfos(pk_h, self.rc.radar_quality)

    def _apply_volley_options(self, options):
        # environment.py:364-365: options.get(key, config default) -- a key that is absent falls back to the CONFIG's value
        mode = bool(options.get("volley_mode", self._config_volley[0]))
        size = int(options.get("volley_size", self._config_volley[1]))
        if mode == bool(self.rc.volley_mode) and (not mode or size == int(self.rc.volley_size)):
            return
        if mode and not 1 <= size <= _lib.MAX_VOLLEY:
            raise ValueError(f"volley_size must be 1..{_lib.MAX_VOLLEY} (got {size})")
        import dataclasses
        step_count, was_set = self.training_step_count, getattr(self, "_step_count_set", False)
        self.close()
        self.rc = dataclasses.replace(self.rc, volley_mode=mode, volley_size=size if mode else 1)
        self._open()                      # every environment of the new handle is un-reset; the caller's reset follows
        if was_set:
            self.set_training_step_count(step_count)

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("step_async() called twice without step_wait()")
        self._pending = self.step_torch(self._upload_actions(actions), want_done_list=True)

    def _upload_actions(self, actions):
        """Host action batch -> the device action buffer (pinned staging, stream-ordered before the step kernel)."""
        a = np.asarray(actions)
        if a.shape != (self.num_envs, _lib.ACT_DIM):
            raise ValueError(f"actions must have shape ({self.num_envs}, {_lib.ACT_DIM}), got {a.shape}")
        np.copyto(self._actions_pin.numpy(), a, casting="unsafe")      # one pass: dtype cast + staging
        self._actions_dev.copy_(self._actions_pin, non_blocking=True)
        return self._actions_dev

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError("step_wait() called without step_async()")
        obs, rew, term, trunc, info = self._pending
        self._pending = None
        return self._materialise(obs, rew, term, trunc, info, self.terminal_obs)

    def _materialise(self, obs, rew, term, trunc, info, terminal):
        """Device step results -> the numpy (obs, rewards, dones, infos) tuple SB3 expects (one D2H copy each)."""
        return self._materialise_end(self._materialise_begin(obs, rew, term, trunc, info, terminal))

    def _materialise_begin(self, obs, rew, term, trunc, info, terminal):
        """Enqueue the device-to-host copies of one step's results on the current stream; nothing is waited for (a caller
        with several devices -- sharded.py -- has every shard's copies in flight before it waits for the first)."""
        torch = self._torch

        def d2h(t):   # pinned staging (torch's caching host allocator recycles the blocks), all copies in flight at once
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h

        # the head of the slab (reward, flags, the done counter) in one copy, the observation batch in another; a wrapper's own
        # reward / done tensors (e.g. normalised rewards) are fetched separately.  The info words -- 48 of the 174 bytes per
        # environment and step this path used to bring over -- follow for the finished environments only (compact rows, below);
        # the others' are fetched when somebody indexes such an environment's info (SB3's loops never do: they look at the
        # `episode` / `terminal_observation` / `TimeLimit.truncated` keys of finished ones).
        head_h, obs_t = d2h(self._slab[:self._slab_head]), d2h(obs)
        own = {name: d2h(t) for name, t, mine in (("reward", rew, self.reward), ("terminated", term, self.terminated),
                                                   ("truncated", trunc, self.truncated)) if t.data_ptr() != mine.data_ptr()}
        return head_h, obs_t, own, terminal, torch.cuda.current_stream(self.device), int(self._lib.hlx_vec_steps(self._h)) & 1, self._info_gen

    def _materialise_end(self, ticket):
        head_h, obs_t, own, terminal, stream, parity, gen = ticket
        torch = self._torch
        stream.synchronize()
        head_np = head_h.numpy()

        def plane(name, buf=None, base=0):
            if name in own:
                return own[name].numpy()
            o, nb, dt, shape = self._slab_layout[name]
            return (head_np if buf is None else buf)[o - base:o - base + nb].view(_NP_DTYPES[dt]).reshape(shape)

        def views(pk, extra):
            v = self._unpack_info(pk, pk.view(np.int32), pk.view(np.uint8))
            v.update(extra)
            return v

        obs_h, rew_h = obs_t.numpy(), plane("reward")
        term_h, trunc_h = plane("terminated").astype(bool), plane("truncated").astype(bool)
        dones = term_h | trunc_h
        n_done = int(plane("n_done")[parity])
        host = dict(terminated=term_h, truncated=trunc_h, t_start=self._t_start, radar_quality=self.rc.radar_quality,
                    constants=(bool(self.rc.precision_mode), bool(self.rc.proximity_fuze), float(self.rc.proximity_kill_radius)),
                    volley=(bool(self.rc.volley_mode), int(self.rc.volley_size) if self.rc.volley_mode else 1))
        volley, radar = "missile_min_distances" in self._slab_layout, "radar_debug" in self._slab_layout
        if radar:
            host["radar"] = dict(rc=self.rc, beam_width=self.curriculum()["beam_width"])
        cache = {}

        def full():
            """Every environment's info words of THIS step, fetched once, on first use -- valid until the next step or reset."""
            if "v" not in cache:
                if gen != self._info_gen:
                    raise RuntimeError("the info of an environment that did not finish was first read after a later step() / reset(): "
                                       "index `infos` before stepping again (finished environments' infos stay valid)")
                with torch.cuda.stream(stream):
                    tail = self._slab[self._slab_head:].cpu().numpy()
                extra = {}
                if volley:
                    extra["missile_min_distances"] = plane("missile_min_distances", tail, self._slab_head)
                if radar:
                    extra["radar_planes"] = plane("radar_debug", tail, self._slab_head)
                cache["v"] = views(plane("packed", tail, self._slab_head), extra)
            return cache["v"]

        host["full"] = full
        done_rows: Dict[int, int] = {}
        if n_done:
            with torch.cuda.stream(stream):
                idx = self.done_idx[:n_done].to(torch.int64)
                idx_h = idx.cpu().numpy()
                host["terminal_obs"] = terminal.index_select(0, idx).cpu().numpy()
                host["ep_return"] = self.info["episode_return"].index_select(0, idx).cpu().numpy()
                host["ep_length"] = self.info["episode_length"].index_select(0, idx).cpu().numpy()
                extra = {}
                if volley:
                    extra["missile_min_distances"] = self.info["missile_min_distances"].index_select(1, idx).cpu().numpy()
                if radar:
                    extra["radar_planes"] = self.info["radar_debug"].index_select(1, idx).cpu().numpy()
                host["done"] = views(self.info_packed.index_select(1, idx).cpu().numpy(), extra)
            done_rows = {int(e): r for r, e in enumerate(idx_h)}
        return obs_h, rew_h, dones, LazyInfos(self.num_envs, done_rows, host)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed: Optional[int] = None):
        """SB3 `VecEnv.seed` (`set_random_seed`, scripts/compare_policies.py:150): re-keys the counter-based RNG -- env i
        draws from Philox(seed, env_id_offset + i, vec-step clock) from the next launch on; SB3 callers follow with
        `reset()`.  `None` keeps the current key, as `gym.Env.reset(seed=None)` keeps its generator."""
        if seed is not None:
            self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            _lib.check(self._lib.hlx_set_seed(self._h, self._seed))
        return [seed] * self.num_envs

    def set_episode_pool(self, interval: int = -1):
        """Next-episode pool (hlx.h): `interval` step launches between two fills (-1 = default, 0 = off: every auto-reset is
        computed inside the step launch that needs it).  Results do not depend on it."""
        _lib.check(self._lib.hlx_set_episode_pool(self._h, int(interval)))

    @property
    def episode_pool(self) -> int:
        return int(self._lib.hlx_get_episode_pool(self._h))

    def episode_pool_misses(self) -> int:
        """Auto-resets that were computed inside a step launch, pool on, because their prepared episode was absent or stale
        (diagnostics; synchronises)."""
        out = C.c_int64(0)
        _lib.check(self._lib.hlx_get_episode_pool_misses(self._h, C.byref(out)))
        return int(out.value)

    def episode_pool_crowded(self) -> int:
        """... and the ones computed there although it was ready: a wave copies the prepared episodes of up to four finished
        environments per launch; in a wave with more (a batch-wide max_steps truncation) the rest compute in place (hlx.h)."""
        out = C.c_int64(0)
        _lib.check(self._lib.hlx_get_episode_pool_crowded(self._h, C.byref(out)))
        return int(out.value)

    def episode_pool_stats(self) -> Dict[str, int]:
        """{misses, full_fills, partial_fills, suspended_steps} of the next-episode pool (hlx.h; synchronises)."""
        out = (C.c_int64 * 4)()
        _lib.check(self._lib.hlx_get_episode_pool_stats(self._h, C.byref(out)))
        return dict(misses=int(out[0]), full_fills=int(out[1]), partial_fills=int(out[2]), suspended_steps=int(out[3]))

    @property
    def safe_build(self) -> bool:
        """True if the loaded library is the SAFE build (step-kernel constants read from memory: what build.py falls back to when the
        disassembly lint refuses the product build -- same results, slower; hlx.h hlx_hot_words_from_memory)."""
        return bool(self._lib.hlx_hot_words_from_memory())

    def set_load_schedule(self, mode: int):
        """-1 auto (by batch size), 0 all loads at kernel entry, 1 Kalman / ring loads behind the Philox block, 2 = 1 + the
        lone-wave schedule for at most one wave per SIMD (hlx.h)."""
        _lib.check(self._lib.hlx_set_load_schedule(self._h, int(mode)))

    @property
    def load_schedule(self) -> int:
        return int(self._lib.hlx_get_load_schedule(self._h))

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        if isinstance(indices, int):
            return [indices]
        return indices

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        idx = self._indices(indices)
        if method_name == "set_training_step_count":      # environment.py:269, called every step by the trainer
            self.set_training_step_count(*args, **kwargs)
            return [None] * len(idx)
        if method_name == "get_current_intercept_radius":  # environment.py:223
            return [self.get_current_intercept_radius()] * len(idx)
        if method_name == "seed":                          # scripts/compare_policies.py:150
            self.seed(*args, **kwargs)
            return [None] * len(idx)
        raise AttributeError(f"env_method({method_name!r}) is not part of the batched environment")

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        idx = self._indices(indices)
        if attr_name == "observation_generator":
            return [_ObservationGeneratorView(self)] * len(idx)
        if attr_name in ("interceptor_state", "missile_state"):
            st = self.get_state()
            pre = "int_" if attr_name == "interceptor_state" else "mis_"
            out = []
            for i in idx:
                d = {"position": np.array(st[i].__getattribute__(pre + "pos")[:], np.float32),
                     "velocity": np.array(st[i].__getattribute__(pre + "vel")[:], np.float32)}
                if pre == "int_":
                    d["orientation"] = np.array(st[i].int_quat[:], np.float32)
                    d["fuel"] = float(st[i].fuel)
                out.append(d)
            return out
        if hasattr(self, attr_name):
            return [getattr(self, attr_name)] * len(idx)
        raise AttributeError(attr_name)

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        if attr_name == "training_step_count":
            self.set_training_step_count(value)
        else:
            raise AttributeError(f"set_attr({attr_name!r}) is not supported by the batched environment")

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False] * len(self._indices(indices))

    # ------------------------------------------------------------------ reference env methods
    def set_training_step_count(self, step_count: int):
        """environment.py:269-272 - O(1): the schedules are evaluated host-side, values ride as kernel args."""
        self.training_step_count = int(step_count)
        self._step_count_set = True
        _lib.check(self._lib.hlx_set_global_step(self._h, int(step_count)))

    def curriculum(self) -> Dict[str, float]:
        out = (C.c_double * 5)()
        _lib.check(self._lib.hlx_get_curriculum(self._h, C.byref(out)))
        return dict(intercept_radius=out[0], beam_width=out[1], onboard_reliability=out[2],
                    ground_reliability=out[3], noise_level=out[4])

    def get_current_intercept_radius(self) -> float:
        return self.curriculum()["intercept_radius"]

    # ------------------------------------------------------------------ parity / checkpoint hooks
    def set_noise(self, step_noise=None, reset_noise=None):
        """Parity mode: slot-major float64 device tensors [HLX_STEP_SLOTS, N] / [HLX_RESET_SLOTS, N] replace the Philox draws (None restores)."""
        for x in (step_noise, reset_noise):
            if x is not None and (x.dtype != self._torch.float64 or not x.is_contiguous()):
                raise ValueError("noise tensors must be contiguous float64")
        self._noise = (step_noise, reset_noise)
        _lib.check(self._lib.hlx_set_noise(self._h, step_noise.data_ptr() if step_noise is not None else None,
                                           reset_noise.data_ptr() if reset_noise is not None else None))

    def fill_noise(self, for_reset: bool = False):
        """The Philox draws of the next step (or, `for_reset=True`, of a reset issued now) as slot-major
        tensors ([HLX_STEP_SLOTS, N], [HLX_RESET_SLOTS, N])."""
        t = self._torch
        sn = t.zeros((_lib.STEP_SLOTS, self.num_envs), device=self.device, dtype=t.float64)
        rn = t.zeros((_lib.RESET_SLOTS, self.num_envs), device=self.device, dtype=t.float64)
        _lib.check(self._lib.hlx_fill_noise(self._h, sn.data_ptr(), rn.data_ptr(), 0 if for_reset else 1, self._stream()))
        return sn, rn

    def get_state(self):
        arr = (_lib.HlxEnvState * self.num_envs)()
        self._torch.cuda.synchronize(self.device)
        _lib.check(self._lib.hlx_get_state(self._h, C.addressof(arr)))
        return arr

    def set_state(self, arr):
        self._torch.cuda.synchronize(self.device)
        _lib.check(self._lib.hlx_set_state(self._h, C.addressof(arr)))

    def set_rollout_terminal_obs(self, enable: bool):
        """rollout_torch (one launch per step) also fills `self.terminal_obs` for the environments that finish in a step."""
        _lib.check(self._lib.hlx_set_rollout_terminal_obs(self._h, self.terminal_obs.data_ptr() if enable else None))

    def set_rollout_contract(self, enable: bool, done_list: bool = True):
        """rollout_torch (one launch per step) issues exactly the launches `step_torch` does: terminal observations, every
        info plane (`self.info`) and, with `done_list`, the compacted list of finished environments (`self.done_idx`, its
        length in `self.n_done_pair[clock & 1]`)."""
        if enable:
            _lib.check(self._lib.hlx_set_rollout_outputs(self._h, self.terminal_obs.data_ptr(),
                                                         self.done_idx.data_ptr() if done_list else None, C.byref(self._info_soa)))
        else:
            _lib.check(self._lib.hlx_set_rollout_outputs(self._h, None, None, None))

    @property
    def n_done_pair(self):
        """int32[2] on the device: element (vec-step clock & 1) = number of environments that finished in the latest step."""
        return self._n_done2

    def set_rollout_fused(self, steps_per_launch: int):
        """1 = one launch per step (default); k > 1 = rollout_torch keeps the state on-chip for k steps per launch."""
        _lib.check(self._lib.hlx_set_rollout_fused(self._h, int(steps_per_launch)))

    def profile(self, enable: bool):
        _lib.check(self._lib.hlx_profile(self._h, int(enable)))

    def profile_read(self):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(self._lib.hlx_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
