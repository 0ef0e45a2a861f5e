"""Diagnostic: GPU vs oracle, seeded random batch, RESYNCED before every step (the GPU arena is overwritten with the
oracle's state), bitwise comparison of the post-step state: which state field differs first, and how often.
  python tools/diag_batch.py <scenario> <physics> [n] [T] [key=value overrides ...]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle.oracle as orc
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
from tests.test_gpu_parity import _oracle_to_gpu_state

scenario, physics = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100
over = {}
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    over[k] = eval(v)
rc = resolve_config(scenario_config(scenario, physics, over))
env = HlynrVecEnv(resolved=rc, num_envs=n, seed=77)
ora = orc.OracleVec(rc, n)
g = torch.Generator().manual_seed(5)
sn, rn = env.fill_noise(for_reset=True)
env.reset_torch()
ora.reset(rn.cpu().numpy().T.copy())
FIELDS = ["int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind", "thrust_actual", "fuel", "prev_distance", "T0", "steps", "kf_x"]
first = Counter()
shown = 0
nbad_r = 0
for t in range(T):
    _oracle_to_gpu_state(ora, env)
    pre = np.frombuffer(ora.state, dtype=np.dtype(orc.OrcState)).copy()
    a = torch.rand((n, 6), generator=g) * 2 - 1
    sn, rn = env.fill_noise()
    obs, rew, term, trunc, info = env.step_torch(a.to(env.device))
    out = ora.step(a.numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
    st = env.get_state()
    sg = np.frombuffer(st, dtype=np.dtype(type(st[0])))
    so = np.frombuffer(ora.state, dtype=np.dtype(orc.OrcState))
    done = (term.cpu().numpy() | trunc.cpu().numpy()).astype(bool)
    r_g, r_o = rew.cpu().numpy().astype(np.float64), out["reward"]
    bad_r = np.abs(r_g - r_o) > 1e-5 * np.maximum(1, np.abs(r_o))
    nbad_r += int(bad_r.sum())
    for f in FIELDS:
        x, y = sg[f], so[f]
        d = (x.astype(np.float64) != y.astype(np.float64))
        if d.ndim > 1:
            d = d.any(axis=1)
        d &= ~done
        if d.any():
            first[f] += int(d.sum())
            if shown < 12:
                i = int(np.argmax(d)); shown += 1
                print(f"t={t} env={i} field={f} gpu={x[i]!r} orc={y[i]!r}  pre int_pos={pre['int_pos'][i]} int_vel={pre['int_vel'][i]} "
                      f"mis_pos={pre['mis_pos'][i]} mis_vel={pre['mis_vel'][i]} wind={pre['wind'][i]} T0={pre['T0'][i]!r} "
                      f"steps={pre['steps'][i]} kf_init={pre['kf_init'][i]} kf64={pre['kf_x_is64'][i]} kf_x_pre={pre['kf_x'][i]!r} g_len={pre['g_len'][i]}")
print("env-steps", n * T, "reward>1e-5:", nbad_r, "fields differing (count of env-steps):", dict(first))
env.close()
