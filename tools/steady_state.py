#!/usr/bin/env python3
"""How the per-launch time of the step kernel depends on the phase of the episodes: 65 536 envs, timed right after
reset (all episodes in lock-step) and after fused-rollout desynchronisation of increasing length.
  python tools/steady_state.py [physics] [n_envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

physics = sys.argv[1] if len(sys.argv) > 1 else "base"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", physics)), num_envs=n, seed=1000)
g = torch.Generator(device=env.device).manual_seed(0)
tape = torch.rand((500, n, 6), generator=g, device=env.device) * 2 - 1
env.reset_torch()


def timed(K):
    env.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, K, 500):
        env.rollout_torch(tape[:min(500, K - k)], 4)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e6
    ms, cnt = env.profile_read()
    env.profile(False)
    return wall, 1e3 * ms / cnt


def desync(steps):
    env.set_rollout_fused(64)
    for k in range(0, steps, 500):
        env.rollout_torch(tape[:min(500, steps - k)], 4)
    env.set_rollout_fused(1)
    torch.cuda.synchronize()


timed(50)
done = 50
print(f"{physics} {n} envs, variant {env.kernel_variant}")
for extra in (0, 1000, 1000, 2000, 4000, 8000):
    if extra:
        desync(extra)
        done += extra
    st = env.get_state()
    import numpy as np
    steps = np.frombuffer(st, dtype=np.dtype(type(st[0])))["steps"]
    w20, e20 = timed(20)
    w2k, e2k = timed(1000)
    done += 1020
    print(f"after {done - 1020:6d} steps: K=20 wall {w20:6.2f} us events {e20:6.2f} us | K=1000 wall {w2k:6.2f} events {e2k:6.2f} | "
          f"episode step mean {steps.mean():7.1f} std {steps.std():7.1f}", flush=True)
env.close()
