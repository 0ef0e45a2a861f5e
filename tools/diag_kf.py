"""Diagnostic: free-running batch, GPU vs oracle from the same Philox draws; reports the FIRST step at which the Kalman
state (kf_x, float64) of some environment differs in any bit, with that environment's circumstances one step earlier.
  python tools/diag_kf.py [scenario physics n T] [key=value overrides ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle.oracle as orc
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

args = [a for a in sys.argv[1:] if "=" not in a]
over = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        over[k] = {"True": True, "False": False}.get(v, int(v) if v.isdigit() else v)
scenario, physics = (args + ["medium", "base"])[:2] if len(args) >= 2 else ("medium", "base")
n, T = (int(args[2]), int(args[3])) if len(args) >= 4 else (1024, 260)
rc = resolve_config(scenario_config(scenario, physics, over))
env = HlynrVecEnv(resolved=rc, num_envs=n, seed=1234)
ora = orc.OracleVec(rc, n)
g = torch.Generator(device="cpu").manual_seed(7)
sn, rn = env.fill_noise(for_reset=True)
env.reset_torch()
ora.reset(rn.cpu().numpy().T.copy())


def states():
    st = np.frombuffer(env.get_state(), dtype=np.dtype(type(env.get_state()[0])))
    so = np.frombuffer(ora.state, dtype=np.dtype(type(ora.state[0])))
    return st.copy(), so.copy()


prev = states()
shown = 0
for t in range(T):
    a = torch.rand((n, 6), generator=g) * 2 - 1
    if t % 3 == 0:
        a[:, 2] = 0.9
    sn, rn = env.fill_noise()
    obs, rew, term, trunc, info = env.step_torch(a.to(env.device))
    flags = info["flags"].cpu().numpy()
    out = ora.step(a.numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
    st, so = states()
    thr = float(os.environ.get("DIAG_KF_REL", "1e-11"))      # ignore last-bit float64 differences (reciprocal-multiply vs divide)
    bad = np.nonzero((np.abs(st["kf_x"] - so["kf_x"]) > thr * np.maximum(1.0, np.abs(so["kf_x"]))).any(axis=1))[0]
    if len(bad):
        print(f"step {t}: kf_x differs in {len(bad)} of {n} envs")
        for i in bad[:4]:
            pst, pso = prev
            print(f"  env {i}: steps {st['steps'][i]} done {bool(term[i] or trunc[i])} flags {flags[i]:08b}")
            print("    before: gpu kf_init/x64", pst["kf_init"][i], pst["kf_x_is64"][i], " oracle", pso["kf_init"][i], pso["kf_x_is64"][i],
                  " kf_x equal before:", bool((pst["kf_x"][i].view(np.uint64) == pso["kf_x"][i].view(np.uint64)).all()))
            print("    after:  gpu kf_init/x64", st["kf_init"][i], st["kf_x_is64"][i], " oracle", so["kf_init"][i], so["kf_x_is64"][i])
            print("    gpu    kf_x", st["kf_x"][i].tolist())
            print("    oracle kf_x", so["kf_x"][i].tolist())
            print("    diff       ", (st["kf_x"][i] - so["kf_x"][i]).tolist())
            print("    kf_P gpu", st["kf_P"][i].tolist(), " oracle", [so["kf_P"][i][k] for k in (0, 3, 18, 21)])
            print("    g_ring(gpu, newest 2)", st["g_ring"][i][max(0, st["g_len"][i] - 2):st["g_len"][i]].tolist())
        shown += 1
        if shown >= 2:
            break
        # re-synchronise so that the next independent difference can be found
        ora_state = ora.state
        for i in bad:
            for k in range(6):
                ora_state[i].kf_x[k] = float(st["kf_x"][i][k])
    prev = (st, so)
else:
    print(f"no kf_x bit difference in {T} steps x {n} envs")
env.close()
