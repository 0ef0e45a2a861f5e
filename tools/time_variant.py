#!/usr/bin/env python3
"""Steady-state per-launch time of one kernel variant (HIP events inside hlx_rollout, as bench.py's roofline leg):
  [HLX_LIBRARY=...] python tools/time_variant.py [physics = base | v2 | v2dr | config] [scenario] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

physics = sys.argv[1] if len(sys.argv) > 1 else "base"
scenario = sys.argv[2] if len(sys.argv) > 2 else "medium"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
env = HlynrVecEnv(scenario_config(scenario, physics), num_envs=n, seed=1000)
g = torch.Generator(device=env.device).manual_seed(0)
tape = torch.rand((500, n, 6), generator=g, device=env.device) * 2 - 1
env.reset_torch()
env.set_rollout_fused(64)
for _ in range(8):
    env.rollout_torch(tape, 4)          # 4000 steps: episodes desynchronised
env.set_rollout_fused(1)
env.rollout_torch(tape[:200], 4)
out = []
for rep in range(3):
    env.profile(True)
    for _ in range(4):
        env.rollout_torch(tape, 4)
    torch.cuda.synchronize()
    ms, cnt = env.profile_read()
    env.profile(False)
    out.append(1e3 * ms / cnt)
print(f"{scenario}/{physics} [{env.kernel_variant}{'+baked' if env.kernel_baked else ''}] n={n}: " + " ".join(f"{x:.3f}" for x in out) + " us per launch")
env.close()
