import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
for phys in ("base", "v2dr"):
    n = 65536
    env = HlynrVecEnv(scenario_config("medium", phys), num_envs=n, seed=5)
    env.reset_torch()
    g = torch.Generator(device=env.device).manual_seed(0)
    lens = []
    for t in range(3000):
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        obs, rew, term, trunc, info = env.step_torch(a)
        done = (term | trunc) != 0
        if t % 10 == 0:
            lens.append(env.info["episode_length"][done].cpu().numpy())
    l = np.concatenate(lens)
    print(phys, "episodes", l.size, "min", l.min(), "p0.1", np.percentile(l, 0.1), "p1", np.percentile(l, 1), "p5", np.percentile(l, 5), "p50", np.percentile(l, 50), "max", l.max(),
          "frac<64 %.5f <128 %.5f <256 %.5f" % ((l < 64).mean(), (l < 128).mean(), (l < 256).mean()), "misses", env.episode_pool_misses())
    env.close()
