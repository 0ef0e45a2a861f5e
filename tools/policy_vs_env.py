"""Context for the headline: what one rollout step costs next to the environment step when a policy is in the loop.
The reference's flat-PPO policy (train_flat_ppo.py:405-430: CustomMLP 104 -> 512 -> 512 -> 256 with LayerNorm + ReLU,
then action/value heads) as a plain torch module on the GPU, random weights, batch = 65 536 stacked observations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
from hlynr_intercept_amd.wrappers import VecFrameStack, VecNormalize

n = 65536
dev = torch.device("cuda", 0)


def mlp(dtype):
    layers, d = [], 104
    for h in (512, 512, 256):
        layers += [torch.nn.Linear(d, h), torch.nn.LayerNorm(h), torch.nn.ReLU()]
        d = h
    body = torch.nn.Sequential(*layers).to(dev, dtype)
    return body, torch.nn.Linear(256, 6).to(dev, dtype), torch.nn.Linear(256, 1).to(dev, dtype)


def timed(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


env = VecNormalize(VecFrameStack(HlynrVecEnv(scenario_config("medium", "base"), num_envs=n, seed=1), 4), norm_reward=False)
obs = env.reset_torch()
with torch.no_grad():
    for dtype in (torch.float32, torch.bfloat16):
        body, pi, vf = mlp(dtype)
        x = obs.to(dtype)
        t_pol = timed(lambda: (lambda h: (torch.tanh(pi(h)), vf(h)))(body(x)))
        print(f"policy forward, {str(dtype):15s} batch {n}: {t_pol:8.1f} us")
    acts = torch.rand((n, 6), device=dev) * 2 - 1
    t_env = timed(lambda: env.step_torch(acts), 200)
    print(f"env step + frame stack + normalise (training statistics on):   {t_env:8.1f} us")
    env.training = False
    t_env = timed(lambda: env.step_torch(acts), 200)
    print(f"env step + frame stack + normalise (evaluation mode):           {t_env:8.1f} us")
