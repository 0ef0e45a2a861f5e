"""On-device HRL controller for batched environments.

The per-environment logic of the reference's hierarchical stack -- `abstract_observation`, the forced-transition rules
with hysteresis and min-dwell, the selector cadence and option bookkeeping of `HierarchicalManager.select_action`
(rl_system/hrl/manager.py:113-203, option_manager.py:62-172, observation_abstraction.py:19-130) -- runs in libhlx.so
(include/hlx_hrl.h) for all N environments at once.  The networks stay the caller's: pass a selector callable for
"model"-style selection, and `select_actions` runs each specialist once per step on the environments whose active
option is that specialist's (the reference runs one environment and one specialist at a time).
"""
from __future__ import annotations

import contextlib
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
        self.lstm_state = None                                       # tuple of [layers, N, hidden] tensors in OPTION-MAJOR row order, made on first use
        # option-major row order (include/hlx_hrl.h hlx_hrl_regroup): order[row] = environment, pos[environment] = row -- torch
        # tensors the library maintains in place (hlx_hrl_bind_order)
        self.order = torch.arange(n, dtype=torch.int32, device=dev)
        self.pos = torch.arange(n, dtype=torch.int32, device=dev)
        _lib.check(self._lib.hlx_hrl_bind_order(self._h, self.order.data_ptr(), self.pos.data_ptr()))
        self._scratch, self._bank_args, self.rows_moved = None, None, 0
        self._rows = None            # row-order buffers of select_actions_recurrent: observations, start flags, actions (+ actions in environment order)
        self.section = None          # optional: callable name -> context manager, entered around each phase of select_actions_recurrent

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
        _lib.check(self._lib.hlx_hrl_reset(self._h, m, self._stream()))     # (the next decision of these environments carries info bit 7)

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

        `specialists[k](obs_rows, state_rows, episode_start_rows) -> (action_rows, new_state_rows | None)` with `state_rows` a
        tuple of [layers, rows, hidden] tensors -- the batched form of `model.predict(obs, state=..., episode_start=...)`.
        `state_rows` are VIEWS of the resident banks: a specialist may update them in place and return `None` for the state
        (no write-back traffic at all), or return new tensors, which are copied into the views.

        The reference keeps an LSTM state per specialist and environment, but resets a specialist's state whenever its
        option is switched away from (manager.py:210-215) and all of them when the episode restarts (manager.py:104-107),
        so at any time an environment has exactly ONE live state: that of its active specialist, begun afresh
        (`episode_start=True`, i.e. zeros) on the step an option is entered or the controller restarted.  That state lives
        here, on the device, as `self.lstm_state` -- in OPTION-MAJOR row order (round 4; include/hlx_hrl.h hlx_hrl_regroup):
        `self.order[row]` is the environment of a row, every option's environments are one contiguous run of rows, and a
        specialist's state is a slice of each bank.  The returned `actions` tensor is reused by the next call.  After the controller has decided, only the rows of environments that
        changed option (and the few they displace) move -- a few hundred KB per step where gathering and scattering
        environment-major banks moved 0.5 GB at 65 536 environments.  One host wait per step reads the three run lengths (the
        slices' bounds); everything else is enqueued without synchronising."""
        t = self._torch
        sec = self.section or (lambda name: contextlib.nullcontext())      # optional per-phase timers (bench_configs.py config5)
        obs = self._check_obs(obs)
        with sec("controller"):
            option, abstract, info = self.step(obs, terminated, truncated)
        with sec("grouping"):
            counts = self._regroup(option)
        with sec("gather / scatter"):
            # the observation rows and the "state starts afresh" flags (option switched | first decision since the controller
            # restarted: info bits 0 and 7) in row order -- one launch (hlx_hrl_rows)
            if self._rows is None or self._rows[2].shape[1] != action_dim:
                n, dev = self.num_envs, self.device
                self._rows = (t.empty((n, self.obs_dim), dtype=t.float32, device=dev), t.zeros(n, dtype=t.uint8, device=dev),
                              t.zeros((n, action_dim), dtype=t.float32, device=dev), t.zeros((n, action_dim), dtype=t.float32, device=dev))
            obs_rows_all, starts_u8, act_rows, actions = self._rows
            _lib.check(self._lib.hlx_hrl_rows(self._h, obs.data_ptr(), info.data_ptr(), obs_rows_all.data_ptr(), starts_u8.data_ptr(), self._stream()))
            starts_rows_all = starts_u8.view(t.bool)
            if any(counts[k] and k not in specialists for k in range(3)):
                act_rows.zero_()                              # an option without a specialist acts with zeros
        off = 0
        for k in range(3):
            c = counts[k]
            if c and k in specialists:
                rows = None if self.lstm_state is None else tuple(x[:, off:off + c] for x in self.lstm_state)
                with sec("specialist forward"):
                    act, new = specialists[k](obs_rows_all[off:off + c], rows, starts_rows_all[off:off + c])
                with sec("gather / scatter"):
                    act_rows[off:off + c] = act.to(t.float32)
                    if new is not None:
                        if self.lstm_state is None:         # shapes come from the first specialist that answers
                            self.lstm_state = tuple(t.zeros((x.shape[0], self.num_envs, x.shape[2]), dtype=x.dtype, device=self.device) for x in new)
                            rows = tuple(x[:, off:off + c] for x in self.lstm_state)
                        for dst, x in zip(rows, new):
                            dst.copy_(x)
            off += c
        with sec("gather / scatter"):      # ... and the actions back into environment order: one launch (hlx_hrl_unrows)
            _lib.check(self._lib.hlx_hrl_unrows(self._h, act_rows.data_ptr(), action_dim, actions.data_ptr(), self._stream()))
        return actions, option, info

    def _regroup(self, option):
        """hlx_hrl_regroup + the one host wait of the step: the three run lengths.  Rows of the resident banks move inside them."""
        t = self._torch
        if self.lstm_state is not None:
            key = tuple(x.data_ptr() for x in self.lstm_state)
            if self._bank_args is None or self._bank_args[0] != key:      # the ctypes arguments are built once per set of banks, not per step
                banks = []
                for x in self.lstm_state:
                    if not x.is_contiguous() or x.device != self.device or x.shape[1] != self.num_envs:
                        raise ValueError("resident state banks must be contiguous [layers, N, hidden] tensors on the controller's device")
                    banks += [x[layer] for layer in range(x.shape[0])]      # [N, hidden] each, contiguous
                nb = len(banks)
                rb = (C.c_int64 * nb)(*[b.shape[1] * b.element_size() for b in banks])
                need = int(self._lib.hlx_hrl_regroup_scratch_bytes(self.num_envs, rb, nb))
                if self._scratch is None or self._scratch.numel() < need:
                    self._scratch = t.empty(need, dtype=t.uint8, device=self.device)
                arr = (C.c_void_p * nb)(*[b.data_ptr() for b in banks])
                self._bank_args = (key, arr, rb, nb, self._scratch.data_ptr(), self._scratch.numel())
            _, arr, rb, nb, sp, sn = self._bank_args
            _lib.check(self._lib.hlx_hrl_regroup(self._h, option.data_ptr(), arr, rb, nb, sp, sn, self._stream()))
        else:
            _lib.check(self._lib.hlx_hrl_regroup(self._h, option.data_ptr(), None, None, 0, None, 0, self._stream()))
        out = (C.c_int32 * 4)()
        _lib.check(self._lib.hlx_hrl_group_counts(self._h, C.byref(out)))
        self.rows_moved = int(out[3])
        return [int(out[0]), int(out[1]), int(out[2])]

    def lstm_state_of(self, env_indices):
        """The resident state rows of the given environments (copies; diagnostics and tests): tuple of [layers, len, hidden]."""
        t = self._torch
        rows = self.pos.to(t.int64).index_select(0, t.as_tensor(env_indices, device=self.device, dtype=t.int64))
        return tuple(x.index_select(1, rows) for x in self.lstm_state)

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
