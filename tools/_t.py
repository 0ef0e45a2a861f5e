import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
n = 4 * 1024 * 1024
rc = resolve_config(scenario_config("medium", "base"))
tape = torch.rand((40, n, 6), device="cuda") * 2 - 1
keep = []
for rep in range(8):
    env = HlynrVecEnv(resolved=rc, num_envs=n, seed=1000)
    env.reset_torch()
    env.rollout_torch(tape[:10], 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout_torch(tape, 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"rep {rep}: {1e6*dt:.1f} us/step frac {508*n/dt/8e12:.3f}", flush=True)
    env.close()
    if rep % 2 == 1:
        keep.append(torch.empty(int(300e6 * (rep + 1)), dtype=torch.uint8, device="cuda"))   # perturb the allocator
