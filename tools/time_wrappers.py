"""Cost of the on-device observation pipeline next to the step: python tools/time_wrappers.py [n_envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
from hlynr_intercept_amd.wrappers import VecFrameStack, VecNormalize

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = 300


def timed(stepper, acts):
    """median / min over ten windows of K steps (single windows are at the mercy of one-off stalls: allocation, clock ramp)"""
    for t in range(200):
        stepper(acts[t % len(acts)])
    torch.cuda.synchronize()
    out = []
    for w in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(K):
            stepper(acts[t % len(acts)])
        b.record()
        torch.cuda.synchronize()
        out.append(1e3 * a.elapsed_time(b) / K)
    out.sort()
    return out[len(out) // 2]


env = HlynrVecEnv(scenario_config("medium", "base"), num_envs=n, seed=1)
acts = torch.rand((64, n, 6), device=env.device) * 2 - 1
env.reset_torch()
t_env = timed(lambda a: env.step_torch(a), acts)
env.close()
for label, mk in (("VecFrameStack(4)", lambda e: VecFrameStack(e, 4)),
                  ("VecNormalize(VecFrameStack(4)) training", lambda e: VecNormalize(VecFrameStack(e, 4), norm_reward=False)),
                  ("VecNormalize(VecFrameStack(4)) eval", lambda e: VecNormalize(VecFrameStack(e, 4), norm_reward=False, training=False))):
    w = mk(HlynrVecEnv(scenario_config("medium", "base"), num_envs=n, seed=1))
    w.reset_torch()
    t_w = timed(lambda a: w.step_torch(a), acts)
    print(f"n={n}: step alone {t_env:.2f} us (python-driven), + {label}: {t_w:.2f} us  (pipeline {t_w - t_env:+.2f} us)")
    w.close()
