"""Soak: GPU against the oracle, free-running, on many configurations and a few million env-steps each; counts what is
NOT bit-identical (reward, distance, flags, step counters, integrated state at the end) and the worst observation entry.
  python tools/soak_oracle.py [n_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle.oracle as orc
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ONLY = os.environ.get("SOAK_ONLY")
CASES = [
    ("medium", "base", {}), ("medium", "v2dr", {}), ("hard", "config", {}), ("easy", "config", {}), ("medium", "v2", {}),
    ("medium", "config", {"volley_mode": True, "volley_size": 3}),
    ("medium", "v2dr", {"volley_mode": True, "volley_size": 4, "proximity_fuze_enabled": True, "proximity_kill_radius": 80.0}),
    ("medium", "v2", {"observation_mode": "los_frame", "proximity_fuze_enabled": True, "proximity_kill_radius": 60.0}),
    ("medium", "v2", {"observation_mode": "body_frame"}),
    ("medium", "base", {"curriculum.precision_mode": True}),
    ("hard", "v2dr", {"max_steps": 300}),
]
for scenario, physics, over in (CASES if not ONLY else [CASES[int(k)] for k in ONLY.split(",")]):
    rc = resolve_config(scenario_config(scenario, physics, over))
    env = HlynrVecEnv(resolved=rc, num_envs=n, seed=4321)
    ora = orc.OracleVec(rc, n)
    g = torch.Generator(device=env.device).manual_seed(11)
    sn, rn = env.fill_noise(for_reset=True)
    env.reset_torch()
    ora.reset(rn.cpu().numpy().T.copy())
    bad = dict(reward=0, distance=0, flags=0, fuel_used=0)
    worst_obs, n_done, t0 = np.zeros(26), 0, time.time()
    for t in range(T):
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        if t % 3 == 0:
            a[:, 2] = 0.9
        sn, rn = env.fill_noise()
        obs, rew, term, trunc, info = env.step_torch(a)
        out = ora.step(a.cpu().numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
        te, tr = term.cpu().numpy(), trunc.cpu().numpy()
        done = (te | tr).astype(bool)
        n_done += int(done.sum())
        bad["flags"] += int(((te != out["terminated"]) | (tr != out["truncated"]) | ((info["flags"].cpu().numpy() & 1) != out["intercepted"])).sum())
        bad["reward"] += int((rew.cpu().numpy().astype(np.float64) != out["reward"].astype(np.float32).astype(np.float64)).sum())
        bad["distance"] += int((info["distance"].cpu().numpy() != out["distance"]).sum())
        bad["fuel_used"] += int((info["fuel_used"].cpu().numpy() != out["fuel_used"]).sum())
        og = np.where(done[:, None], info["terminal_observation"].cpu().numpy(), obs.cpu().numpy())
        oo = np.where(done[:, None], ora.terminal_obs, out["obs"])
        worst_obs = np.maximum(worst_obs, np.abs(og - oo).max(axis=0))
    st = np.frombuffer(env.get_state(), dtype=np.dtype(type(env.get_state()[0])))
    so = np.frombuffer(ora.state, dtype=np.dtype(type(ora.state[0])))
    state_bad = {f: int((st[f].astype(np.float64) != so[f].astype(np.float64)).any(axis=-1).sum() if st[f].ndim > 1 else (st[f] != so[f]).sum())
                 for f in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind", "steps", "fuel", "kf_x")}
    Po = so["kf_P"].reshape(n, 6, 6)
    state_bad["kf_P"] = int((np.stack([Po[:, 0, 0], Po[:, 0, 3], Po[:, 3, 0], Po[:, 3, 3]], axis=1) != st["kf_P"].astype(np.float64)).any(axis=1).sum())
    dk = np.abs(st["kf_x"] - so["kf_x"])
    print(f"   kf_x max |diff|: position {dk[:, :3].max():.3e} m, velocity {dk[:, 3:].max():.3e} m/s (|v| min {np.abs(so['kf_x'][:, 3:]).max(axis=1).min():.3e})")
    k = int(np.argmax(worst_obs))
    print(f"{scenario}/{physics} {over} [{env.kernel_variant}{'+baked' if env.kernel_baked else ''}]: {n * T} env-steps, {n_done} episodes ended; "
          f"not bit-identical: {bad} state {state_bad}; worst obs entry [{k}] {worst_obs[k]:.2e} ({time.time() - t0:.0f} s)", flush=True)
    env.close()
