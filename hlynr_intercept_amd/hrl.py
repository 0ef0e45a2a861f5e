"""On-device HRL controller for batched environments.

The per-environment logic of the reference's hierarchical stack -- `abstract_observation`, the forced-transition rules
with hysteresis and min-dwell, the selector cadence and option bookkeeping of `HierarchicalManager.select_action`
(rl_system/hrl/manager.py:113-203, option_manager.py:62-172, observation_abstraction.py:19-130) -- runs in libhlx.so
(include/hlx_hrl.h) for all N environments at once.  The networks stay the caller's: pass a selector callable for
"model"-style selection, and `select_actions` runs each specialist once per step on the environments whose active
option is that specialist's (the reference runs one environment and one specialist at a time).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import numpy as np

from . import _lib

SEARCH, TRACK, TERMINAL = 0, 1, 2
OPTION_NAMES = ("SEARCH", "TRACK", "TERMINAL")
REASONS = ("continue", "selector", "forced")


class HRLController:
    """Batched `HierarchicalManager`: same constructor switches, same per-environment semantics.

    selector: "rules" (selector_policy.py:162-200, in-kernel), "fixed" (always SEARCH), or a callable
    `abstract[N, 7] float32 tensor -> option indices [N]` (the reference's "model" mode; it is evaluated on the whole
    batch every step and consumed only where the selector is due and no forced transition fired)."""

    def __init__(self, num_envs: int, obs_dim: int = 26, device: int = 0, decision_interval: int = 100,
                 enable_forced_transitions: bool = True, enable_hysteresis: bool = True, enable_min_dwell: bool = True,
                 default_option: int = SEARCH, selector="rules", thresholds: Optional[Dict[str, float]] = None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("HRLController needs a ROCm GPU: the controller runs in a HIP kernel and has no CPU fallback")
        self._torch, self._lib = torch, _lib.load()
        self.num_envs, self.obs_dim = int(num_envs), int(obs_dim)
        self.device = torch.device("cuda", int(device))
        self._selector_fn = selector if callable(selector) else None
        mode = 2 if callable(selector) else {"fixed": 0, "rules": 1}[selector]
        cfg = _lib.HlxHrlConfig(n_envs=self.num_envs, obs_dim=self.obs_dim, device=int(device), decision_interval=int(decision_interval),
                                selector_mode=mode, enable_forced=int(enable_forced_transitions), enable_hysteresis=int(enable_hysteresis),
                                enable_min_dwell=int(enable_min_dwell), default_option=int(default_option))
        self._lib.hlx_hrl_default_thresholds(C.byref(cfg))
        for k, v in (thresholds or {}).items():      # option_definitions.py key names
            field = {"radar_lock_quality_min": "lock_min", "radar_lock_quality_search": "lock_search",
                     "close_range_threshold": "close_range", "terminal_fuel_min": "terminal_fuel_min",
                     "miss_imminent_distance": "miss_imminent", "fuel_critical": "fuel_critical"}.get(k, k)
            setattr(cfg, field, float(v))
        self._cfg = cfg
        self._h = C.c_void_p()
        _lib.check(self._lib.hlx_hrl_create(C.byref(cfg), C.byref(self._h)))
        n, dev = self.num_envs, self.device
        self.abstract = torch.zeros((n, 7), dtype=torch.float32, device=dev)
        self.option = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.info = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._closed = False
        # recurrent specialists (select_actions_recurrent): one live LSTM state per environment, resident on the device
        self._fresh = torch.ones(n, dtype=torch.bool, device=dev)    # controller (re)started: the next predict begins an episode
        self.lstm_state = None                                       # tuple of [layers, N, hidden] tensors, made on first use

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if not self._closed and self._h:
            self._torch.cuda.synchronize(self.device)
            self._lib.hlx_hrl_destroy(self._h)
            self._h = C.c_void_p()
            self._closed = True

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        """`HierarchicalManager.reset()` for all environments (or those where `mask` is non-zero)."""
        m = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self._torch.uint8).contiguous()
            m = mask.data_ptr()
        _lib.check(self._lib.hlx_hrl_reset(self._h, m, self._stream()))
        if mask is None:
            self._fresh.fill_(True)
        else:
            self._fresh |= mask.bool()

    def abstract_observation(self, obs):
        """[N, 26k] observation tensor -> [N, 7] abstract state (no controller state is touched)."""
        obs = self._check_obs(obs)
        _lib.check(self._lib.hlx_hrl_abstract(self._h, obs.data_ptr(), self.abstract.data_ptr(), self._stream()))
        return self.abstract

    def _check_obs(self, obs):
        t = self._torch
        if obs.dtype != t.float32 or not obs.is_contiguous() or obs.device != self.device:
            obs = obs.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(obs.shape) != (self.num_envs, self.obs_dim):
            raise ValueError(f"obs must have shape ({self.num_envs}, {self.obs_dim}), got {tuple(obs.shape)}")
        return obs

    def step(self, obs, terminated=None, truncated=None):
        """One `select_action` decision for every environment.  `terminated` / `truncated`: the flags the PREVIOUS
        env step returned (those environments' controllers restart, as the wrapper's reset() does).
        Returns (option [N] uint8, abstract [N, 7], info [N] uint8 -- see include/hlx_hrl.h for the bit layout)."""
        t = self._torch
        obs = self._check_obs(obs)
        choice = None
        if self._selector_fn is not None:
            a = self.abstract_observation(obs)
            choice = self._selector_fn(a).to(device=self.device, dtype=t.int32).contiguous()
        da = terminated.data_ptr() if terminated is not None else None
        db = truncated.data_ptr() if truncated is not None else None
        _lib.check(self._lib.hlx_hrl_step(self._h, obs.data_ptr(), da, db, choice.data_ptr() if choice is not None else None,
                                          self.abstract.data_ptr(), self.option.data_ptr(), self.info.data_ptr(), self._stream()))
        return self.option, self.abstract, self.info

    def select_actions(self, obs, specialists: Dict[int, Callable], terminated=None, truncated=None, action_dim: int = 6):
        """`HierarchicalManager.select_action` for the batch: decide the options, then run each specialist ONCE on the
        rows whose active option is its own and scatter the actions back.  `specialists[k](obs_rows) -> actions_rows`."""
        t = self._torch
        option, abstract, info = self.step(obs, terminated, truncated)
        actions = t.zeros((self.num_envs, action_dim), dtype=t.float32, device=self.device)
        for k, idx in self._rows_by_option(option, specialists):
            actions.index_copy_(0, idx, specialists[k](obs.index_select(0, idx)).to(t.float32))
        return actions, option, info

    def _rows_by_option(self, option, specialists):
        """(option, row indices ascending) for every option that has a specialist and at least one environment: one stable
        sort and ONE host synchronisation (the three group sizes) instead of a `nonzero` -- and a sync -- per option."""
        t = self._torch
        perm = t.argsort(option, stable=True)
        counts = t.bincount(option.to(t.int64), minlength=3).tolist()
        out, start = [], 0
        for k in range(len(counts)):
            if counts[k] and k in specialists:
                out.append((k, perm[start:start + counts[k]]))
            start += counts[k]
        return out

    def select_actions_recurrent(self, obs, specialists: Dict[int, Callable], terminated=None, truncated=None,
                                 action_dim: int = 6):
        """`select_actions` for recurrent specialists (RecurrentPPO, hrl/specialist_policies.py:95-183).

        `specialists[k](obs_rows, state_rows, episode_start_rows) -> (action_rows, new_state_rows)` with `state_rows` a
        tuple of [layers, rows, hidden] tensors -- the batched form of `model.predict(obs, state=..., episode_start=...)`.

        The reference keeps an LSTM state per specialist and environment, but resets a specialist's state whenever its
        option is switched away from (manager.py:210-215) and all of them when the episode restarts (manager.py:104-107),
        so at any time an environment has exactly ONE live state: that of its active specialist, begun afresh
        (`episode_start=True`, i.e. zeros) on the step an option is entered or the controller restarted.  That state lives
        here, on the device, as `self.lstm_state`; per step each specialist sees the rows of its own environments only."""
        t = self._torch
        option, abstract, info = self.step(obs, terminated, truncated)
        starts = (info & 1).bool() | self._fresh
        for flag in (terminated, truncated):            # controllers restarted by hlx_hrl_step before deciding
            if flag is not None:
                starts |= flag.to(self.device).bool()
        self._fresh.zero_()
        actions = t.zeros((self.num_envs, action_dim), dtype=t.float32, device=self.device)
        for k, idx in self._rows_by_option(option, specialists):
            fn = specialists[k]
            rows = None if self.lstm_state is None else tuple(x.index_select(1, idx) for x in self.lstm_state)
            act, new = fn(obs.index_select(0, idx), rows, starts.index_select(0, idx))
            actions.index_copy_(0, idx, act.to(t.float32))
            if self.lstm_state is None:                 # shapes come from the first specialist that answers
                self.lstm_state = tuple(t.zeros((x.shape[0], self.num_envs, x.shape[2]), dtype=x.dtype, device=self.device)
                                        for x in new)
            for bank, x in zip(self.lstm_state, new):
                bank.index_copy_(1, idx, x)
        return actions, option, info

    @staticmethod
    def decode_info(info_byte: int) -> Dict[str, object]:
        """The reference's 'hrl/*' info keys for one environment (manager.py:176-201)."""
        b = int(info_byte)
        return {"hrl/option_switched": bool(b & 1), "hrl/switch_reason": REASONS[(b >> 1) & 3],
                "hrl/forced_transition": bool(b & 8), "hrl/selector_due": bool(b & 16), "hrl/selector_choice": (b >> 5) & 3}

    def get_state(self) -> np.ndarray:
        """[N, 4] int32: option, steps_in_option, option-manager steps, total_steps."""
        out = np.zeros((self.num_envs, 4), np.int32)
        _lib.check(self._lib.hlx_hrl_get_state(self._h, out.ctypes.data))
        return out

    def set_state(self, state: np.ndarray):
        s = np.ascontiguousarray(state, np.int32)
        assert s.shape == (self.num_envs, 4)
        _lib.check(self._lib.hlx_hrl_set_state(self._h, s.ctypes.data))
