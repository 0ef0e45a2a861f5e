"""Diagnostic: which observation entries differ between GPU and oracle (free-running random batch), and in what situation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle.oracle as orc
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

if len(sys.argv) > 1 and sys.argv[1] == "base":
    over, phys, n, T = {}, "base", 8192, 2500
else:
    over, phys, n, T = {"observation_mode": "los_frame", "proximity_fuze_enabled": True, "proximity_kill_radius": 60.0, "max_steps": 200}, "v2", 4096, 260
rc = resolve_config(scenario_config("medium", phys, over))
env = HlynrVecEnv(resolved=rc, num_envs=n, seed=1234)
ora = orc.OracleVec(rc, n)
g = torch.Generator(device="cpu").manual_seed(7)
sn, rn = env.fill_noise(for_reset=True)
env.reset_torch(); ora.reset(rn.cpu().numpy().T.copy())
worst = np.zeros(26); shown = 0
for t in range(T):
    a = torch.rand((n, 6), generator=g) * 2 - 1
    if t % 3 == 0: a[:, 2] = 0.9
    sn, rn = env.fill_noise()
    obs, rew, term, trunc, info = env.step_torch(a.to(env.device))
    out = ora.step(a.numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
    done = (term.cpu().numpy() | trunc.cpu().numpy()).astype(bool)
    og = np.where(done[:, None], info["terminal_observation"].cpu().numpy(), obs.cpu().numpy())
    oo = np.where(done[:, None], ora.terminal_obs, out["obs"])
    e = np.abs(og - oo)
    worst = np.maximum(worst, e.max(axis=0))
    bad = np.argwhere(e > 1e-5 * np.maximum(1.0, np.abs(oo)))
    for i, k in bad[:3]:
        if shown < 15:
            shown += 1
            so = np.frombuffer(ora.state, dtype=np.dtype(orc.OrcState))[i]
            print(f"t={t} env={i} obs[{k}] gpu={og[i,k]!r} orc={oo[i,k]!r} | obs gpu {og[i,:17]} | kf_x={so['kf_x']} int_pos={so['int_pos']} int_vel={so['int_vel']} steps={so['steps']} kf64={so['kf_x_is64']}")
            sg = np.frombuffer(env.get_state(), dtype=np.dtype(type(env.get_state()[0])))[i]
            print(f"      gpu state: kf_x={sg['kf_x']} (bits equal: {bool((sg['kf_x'].view(np.uint64) == so['kf_x'].view(np.uint64)).all())}) kf64={sg['kf_x_is64']} kf_init={sg['kf_init']} "
                  f"int_pos equal {bool((sg['int_pos'] == so['int_pos']).all())} int_vel equal {bool((sg['int_vel'] == so['int_vel']).all())} done={bool(done[i])}")
            print(f"      obs orc {oo[i,:17]}")
print("worst per index:", np.array2string(worst, precision=2))
env.close()
