"""How often does a 64-environment wave need each data-dependent block of the observation code?  Steady-state batch
(bench.py's default workload), per-env predicates from the info planes, the observation row and the exported state ->
fraction of environments and fraction of WAVES in which at least one / every lane satisfies it.  A block that no lane of a
wave needs is jumped over with a taken branch (an instruction-buffer refill for the lone wave of a SIMD): blocks needed by
almost no wave belong out of line (RARE()), blocks needed by almost every wave stay in line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
physics = sys.argv[2] if len(sys.argv) > 2 else "base"
env = HlynrVecEnv(scenario_config("medium", physics), num_envs=n, seed=1)
env.reset_torch()
g = torch.Generator(device=env.device).manual_seed(0)
tape = torch.rand((64, n, 6), generator=g, device=env.device) * 2 - 1
env.set_rollout_fused(64)
for _ in range(64):
    env.rollout_torch(tape, 2)
env.set_rollout_fused(1)
acc = {}
def add(name, pred):
    w = pred.reshape(-1, 64)
    a = acc.setdefault(name, [0.0, 0.0, 0.0, 0])
    a[0] += pred.mean(); a[1] += w.any(1).mean(); a[2] += w.all(1).mean(); a[3] += 1
for t in range(40):
    obs, rew, term, trunc, info = env.step_torch(tape[t])
    torch.cuda.synchronize()
    fl = info["flags"].cpu().numpy()
    o = obs.cpu().numpy()
    st = env.get_state()
    steps = np.array([s.steps for s in st]); kfi = np.array([s.kf_init for s in st]) != 0; k64 = np.array([s.kf_x_is64 for s in st]) != 0
    on, gd = (fl & 32) != 0, (fl & 64) != 0
    done = (term | trunc).cpu().numpy() != 0
    add("onboard detected (delayed)", on); add("ground detected (delayed)", gd); add("either detected", on | gd); add("both detected", on & gd)
    add("neither detected", ~on & ~gd)
    add("have_track (obs[0] != -2)", o[:, 0] != -2.0); add("no track", o[:, 0] == -2.0)
    add("ground block valid (obs[17] != -2)", o[:, 17] != -2.0); add("ground block invalid", o[:, 17] == -2.0)
    add("datalink > 0", o[:, 24] > 0); add("datalink == 0", o[:, 24] == 0)
    add("kf_init", kfi); add("kf not init", ~kfi); add("kf_x64", k64); add("kf float32 state", ~k64 & kfi)
    add("steps <= 1", steps <= 1); add("steps <= 5 (ground ring filling)", steps <= 5); add("steps > 1000", steps > 1000)
    add("done this step", done); add("closing > 0 (obs[13] != -1)", o[:, 13] != -1.0)
    add("fuel == 0", o[:, 12] == 0.0)
    # physics regimes (v2 models): altitude bands of the atmosphere / wind profile, Mach bands of the drag model
    buf = np.frombuffer(st, dtype=np.uint8).reshape(n, -1)
    f32 = buf[:, :19 * 4].view(np.float32)          # int_pos 0-2, int_vel 3-5, quat 6-9, fuel 10, thrust 11-13, mis_pos 14-16, mis_vel 17-19
    f32b = buf[:, :20 * 4].view(np.float32)
    iz, mz = f32b[:, 2], f32b[:, 16]
    isp = np.linalg.norm(f32b[:, 3:6], axis=1); msp = np.linalg.norm(f32b[:, 17:20], axis=1)
    add("interceptor z <= 10", iz <= 10); add("interceptor 10 < z <= 1000", (iz > 10) & (iz <= 1000)); add("interceptor z > 1000", iz > 1000)
    add("interceptor z > 11000", iz > 11000); add("missile z > 11000", mz > 11000)
    for nm, sp in (("interceptor", isp), ("missile", msp)):
        m = sp / 340.0
        add(nm + " Mach < 0.8", m < 0.8); add(nm + " 0.8 <= Mach < 1.2", (m >= 0.8) & (m < 1.2)); add(nm + " Mach >= 1.2", m >= 1.2)
        add(nm + " speed < 1e-6", sp < 1e-6)
print(f"{'predicate':42s} {'envs':>8s} {'waves any':>10s} {'waves all':>10s}")
for k, (a, b, c, m) in acc.items():
    print(f"{k:42s} {a / m:8.4f} {b / m:10.4f} {c / m:10.4f}")
env.close()
