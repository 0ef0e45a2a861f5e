#!/usr/bin/env python3
"""PCIe-inclusive rate of the SB3-shaped numpy path: `env.step(actions: np.ndarray)` copies the action
batch host->device and observation / reward / done / info planes device->host every step, as SB3's VecEnv contract
demands.  The headline (`bench.py`) is the device-resident tensor API; this is the number a caller who insists on
host arrays gets.  Usage: python tools/pcie_rate.py [n_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    import torch
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    env = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", "base")), num_envs=n, seed=1)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (8, n, 6)).astype(np.float32)
    env.reset()
    for t in range(20):
        env.step(acts[t % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        obs, rew, dones, infos = env.step(acts[t % 8])
    dt = time.perf_counter() - t0
    # actions up; observations + the head of the slab (reward, flags, done counter) down; the info words follow for finished
    # environments only (a few rows per step; the rest on request)
    per_step_bytes = acts[0].nbytes + obs.nbytes + env._slab_head
    print(f"numpy path: {n} envs, {1e6 * dt / steps:.1f} us/step, {n * steps / dt:.3e} env-steps/s, "
          f"{per_step_bytes / 1e6:.2f} MB over PCIe per step ({per_step_bytes * steps / dt / 1e9:.1f} GB/s)")
    # tensor API on the same handle, for the ratio
    a_dev = torch.from_numpy(acts).to(env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        env.step_torch(a_dev[t % 8])
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    print(f"tensor path (hlx_step via ctypes, device pointers): {1e6 * dt2 / steps:.1f} us/step, {n * steps / dt2:.3e} env-steps/s")
    env.close()


if __name__ == "__main__":
    main()
