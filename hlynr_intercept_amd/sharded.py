"""Single-process multi-device facade: one `HlynrVecEnv` shard per device behind ONE vector-env object.

The reference's trainers are single-process (`rl_system/scripts/train_flat_ppo.py:371` builds one vec env and hands it to one
PPO): a user of that script reaches BASELINE config 4's 524 288 environments (8 x 65 536) only if one object spans the
node's GPUs.  `ShardedHlynrVecEnv(config, num_envs=524288, devices=range(8))` is that object:

* shard k owns the contiguous slab `shard_range(num_envs, len(devices), k)` of global environment ids on `devices[k]`, with its
  own handle (`env_id_offset` = the slab's first id) and stream; the counter-based generator is keyed by the GLOBAL id, so the
  union of the shards is bit-identical to one big batch (tests/test_sharded_facade_gpu.py);
* one host thread issues the shards' launches back to back -- every call of the C ABI is asynchronous, so the devices step
  concurrently -- and nothing is ever concatenated on a device: the tensor API takes and returns PER-SHARD tensors, each
  living on its shard's GPU (a data-parallel policy replica consumes them in place);
* the SB3-shaped numpy API (`reset()`, `step_async()` / `step_wait()`, `env_method`, `get_attr`, ...) concatenates on the
  HOST, where SB3 wants its arrays anyway: all shards' device-to-host copies are in flight before the first is waited for.

There is no collective anywhere in here.  A device may appear more than once in `devices` (two shards on
one GPU: how this facade is tested on a one-GPU box); such shards get streams of their own, forked from / joined to the
caller's current stream with events.  UNMEASURED ON > 1 GPU: no round of this build was offered a multi-GPU box
(DESIGN.md section 7).
"""
from __future__ import annotations

import contextlib
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .config import resolve_config
from .shard import shard_range
from .vec_env import HlynrVecEnv, _SB3VecEnv


class _ShardedInfos(Sequence):
    """`infos` of one vec step over all shards: a list of N dicts to the caller, lazily built shard by shard."""

    def __init__(self, parts, offsets):
        self._parts, self._off = parts, offsets          # offsets[k] = global index of shard k's first environment (+ total)

    def __len__(self):
        return self._off[-1]

    def _locate(self, i):
        k = int(np.searchsorted(self._off, i, side="right")) - 1
        return k, i - self._off[k]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        k, j = self._locate(i)
        return self._parts[k][j]

    def __iter__(self):
        for p in self._parts:
            yield from p

    def done_items(self):
        return [(self._off[k] + i, d) for k, p in enumerate(self._parts) for i, d in p.done_items()]


class ShardedHlynrVecEnv(_SB3VecEnv):
    """`num_envs` intercept environments spread over `devices`, one fused HIP kernel launch per device and step."""

    metadata = {"render_modes": []}

    def __init__(self, config: Optional[Dict[str, Any]] = None, num_envs: int = 16, devices: Sequence[int] = (0,), seed: int = 0,
                 env_id_offset: int = 0, resolved=None, radar_debug: bool = False):
        import torch

        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        if num_envs < len(devices):
            raise ValueError(f"{num_envs} environments cannot be spread over {len(devices)} shards")
        self._torch = torch
        self.config = config or {}
        self.rc = resolved if resolved is not None else resolve_config(self.config)
        self.num_envs, self.devices = int(num_envs), devices
        ranges = [shard_range(self.num_envs, len(devices), k) for k in range(len(devices))]
        self.offsets = [off for off, _ in ranges] + [self.num_envs]
        self.shards: List[HlynrVecEnv] = []
        try:
            for (off, cnt), d in zip(ranges, devices):
                self.shards.append(HlynrVecEnv(self.config, num_envs=cnt, device=d, seed=seed, env_id_offset=env_id_offset + off,
                                               resolved=self.rc, radar_debug=radar_debug))
        except Exception:
            for s in self.shards:
                s.close()
            raise
        # a shard that shares its device with another one gets a stream of its own (fork / join with events, no host sync);
        # a shard alone on its device runs on that device's current stream, like a plain HlynrVecEnv
        self._streams = [torch.cuda.Stream(torch.device("cuda", d)) if devices.count(d) > 1 else None for d in devices]
        self.observation_space, self.action_space = self.shards[0].observation_space, self.shards[0].action_space
        self._pending = None
        self.training_step_count = 0
        if _SB3VecEnv is not object:
            _SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    # ------------------------------------------------------------------ plumbing
    @contextlib.contextmanager
    def _on(self, k):
        """Issue shard k's work: on its own stream (forked from the device's current stream, joined back afterwards) when it
        shares the device, else directly on the device's current stream."""
        st = self._streams[k]
        if st is None:
            yield
            return
        cur = self._torch.cuda.current_stream(self.shards[k].device)
        st.wait_stream(cur)
        with self._torch.cuda.stream(st):
            yield
        cur.wait_stream(st)

    def _split_actions(self, actions):
        """One [N, 6] tensor (any device, or numpy) or a sequence of per-shard tensors -> per-shard tensors on their GPUs."""
        t = self._torch
        if isinstance(actions, (list, tuple)):
            if len(actions) != len(self.shards):
                raise ValueError(f"expected {len(self.shards)} per-shard action tensors, got {len(actions)}")
            return list(actions)
        if not t.is_tensor(actions):
            actions = t.as_tensor(np.asarray(actions, np.float32))
        if tuple(actions.shape) != (self.num_envs, 6):
            raise ValueError(f"actions must have shape ({self.num_envs}, 6), got {tuple(actions.shape)}")
        return [actions[self.offsets[k]:self.offsets[k + 1]].to(device=s.device, dtype=t.float32, non_blocking=True)
                for k, s in enumerate(self.shards)]

    # ------------------------------------------------------------------ device API: per-shard tensors, nothing concatenated
    def reset_torch(self, masks=None):
        """Resets every shard (or the environments its mask names); returns the per-shard observation tensors.  EVERY shard
        sees the call, also with an all-zero mask: the reset epoch of the random streams counts calls per handle (hlx.h)."""
        out = []
        for k, s in enumerate(self.shards):
            with self._on(k):
                out.append(s.reset_torch(None if masks is None else masks[k]))
        return tuple(out)

    def step_torch(self, actions, want_done_list: bool = False):
        """One vec step on every device.  `actions`: per-shard tensors (each on its shard's GPU) or one [N, 6] tensor that is
        sliced and moved.  Returns a tuple of per-shard (obs, reward, terminated, truncated, info) -- see HlynrVecEnv.step_torch;
        launches of different devices overlap (every call is asynchronous)."""
        parts = self._split_actions(actions)
        out = []
        for k, (s, a) in enumerate(zip(self.shards, parts)):
            with self._on(k):
                out.append(s.step_torch(a, want_done_list))
        return tuple(out)

    def synchronize(self):
        for s in self.shards:
            self._torch.cuda.synchronize(s.device)

    # ------------------------------------------------------------------ SB3 VecEnv API (numpy at the boundary, concatenated on the host)
    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        if options is not None:
            for s in self.shards:
                s._apply_volley_options(options)
        if seed is not None:
            self.seed(seed)
        obs = self.reset_torch()
        import time
        now = time.time()
        snaps, outs = [], []
        for k, s in enumerate(self.shards):
            s._t_start = now
            with self._on(k):
                snaps.append(s._snapshot_reset_infos())
                outs.append(obs[k].cpu().numpy())              # (synchronises shard k's stream: its snapshot has landed too)
        self.reset_infos = _ShardedInfos(snaps, self.offsets)      # reset()'s info per environment (environment.py:595-601)
        return np.concatenate(outs, axis=0)

    def step_async(self, actions):
        if self._pending is not None:
            raise RuntimeError("step_async() called twice without step_wait()")
        a = np.asarray(actions)
        if a.shape != (self.num_envs, 6):
            raise ValueError(f"actions must have shape ({self.num_envs}, 6), got {a.shape}")
        tickets = []
        for k, s in enumerate(self.shards):
            with self._on(k):       # upload, launch and the device-to-host copies of shard k, all enqueued before anything is waited for
                res = s.step_torch(s._upload_actions(a[self.offsets[k]:self.offsets[k + 1]]), want_done_list=True)
                tickets.append(s._materialise_begin(*res, s.terminal_obs))
        self._pending = tickets

    def step_wait(self):
        if self._pending is None:
            raise RuntimeError("step_wait() called without step_async()")
        tickets, self._pending = self._pending, None
        parts = [s._materialise_end(tk) for s, tk in zip(self.shards, tickets)]
        return (np.concatenate([p[0] for p in parts], axis=0), np.concatenate([p[1] for p in parts]),
                np.concatenate([p[2] for p in parts]), _ShardedInfos([p[3] for p in parts], self.offsets))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        for s in self.shards:
            s.close()

    def seed(self, seed: Optional[int] = None):
        for s in self.shards:
            s.seed(seed)
        return [seed] * self.num_envs

    def _per_shard(self, indices):
        """global indices (None = all) -> [(shard, local indices or None)] in global order"""
        if indices is None:
            return [(s, None) for s in self.shards]
        idx = [indices] if isinstance(indices, int) else list(indices)
        out = []
        for i in idx:
            if not 0 <= i < self.num_envs:
                raise IndexError(i)
            k = int(np.searchsorted(self.offsets, i, side="right")) - 1
            out.append((self.shards[k], [i - self.offsets[k]]))
        return out

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        if method_name == "set_training_step_count":          # environment.py:269, every step (train_flat_ppo.py:175): O(shards)
            self.set_training_step_count(*args, **kwargs)
            return [None] * (self.num_envs if indices is None else len(self._per_shard(indices)))
        out: List[Any] = []
        for s, local in self._per_shard(indices):
            out += s.env_method(method_name, *args, indices=local, **kwargs)
        return out

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        out: List[Any] = []
        for s, local in self._per_shard(indices):
            out += s.get_attr(attr_name, local)
        return out

    def set_attr(self, attr_name: str, value: Any, indices=None) -> None:
        for s in self.shards:
            s.set_attr(attr_name, value)

    def env_is_wrapped(self, wrapper_class, indices=None) -> List[bool]:
        return [False] * (self.num_envs if indices is None else len(self._per_shard(indices)))

    # ------------------------------------------------------------------ reference env methods
    def set_training_step_count(self, step_count: int):
        self.training_step_count = int(step_count)
        for s in self.shards:
            s.set_training_step_count(step_count)

    def curriculum(self) -> Dict[str, float]:
        return self.shards[0].curriculum()

    def get_current_intercept_radius(self) -> float:
        return self.shards[0].get_current_intercept_radius()

    @property
    def kernel_variant(self) -> str:
        return self.shards[0].kernel_variant
