"""Where does a wave spend its life?  Runs the stamp-instrumented diagnostic build (libhlx_stamps.so) and
prints the median per-segment share of the wave lifetime.  Shares only: the stamps' fences forbid overlaps the
product kernel has, so this build's absolute run time is never quoted."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HLX_LIBRARY"] = os.path.join(ROOT, "hlynr_intercept_amd", "libhlx_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
physics = sys.argv[2] if len(sys.argv) > 2 else "base"
env = HlynrVecEnv(scenario_config("medium", physics), num_envs=n, seed=1)
env.reset_torch()
lib = env._lib
lib.hlx_debug_read_stamps.restype = C.c_int
lib.hlx_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device=env.device).manual_seed(0)
names = ["entry->loads issued", "Philox block", "wait+unpack+clamp", "interceptor", "missile", "wind+termination",
         "reward", "obs: onboard detection", "obs: ground radar", "obs: datalink+fusion", "obs: fusion+Kalman",
         "obs: 26-D formulas", "obs: loop exit", "state stores", "compaction+tile store"]
if os.environ.get("HLX_STAMP_LEVEL") == "2":
    names = ["entry->loop top", "draw select", "rel+range", "forward_vec", "tom+acosf", "Bernoulli", "->reward stamp(n/a)",
             "onboard ring etc", "obs: ground radar", "obs: datalink+fusion", "obs: fusion+Kalman", "obs: 26-D formulas",
             "obs: loop exit", "state stores", "compaction+tile store"]
acc = []
steady = os.environ.get("HLX_STAMP_STEADY", "1") == "1"
print("kernel variant", env.kernel_variant, "baked", env.kernel_baked, "steady-state" if steady else "lock-step (right after reset)")
tape = torch.rand((64, n, 6), generator=g, device=env.device) * 2 - 1
if steady:      # desynchronise the episodes first (fused rollout), then time single-step launches as bench.py issues them
    env.set_rollout_fused(64)
    for _ in range(64):
        env.rollout_torch(tape, 2)
    env.set_rollout_fused(1)
form = os.environ.get("HLX_STAMP_FORM", "single_pass")     # contract = what step_torch issues: terminal observations + info planes + done list
if form == "contract":
    env.set_rollout_contract(True, done_list=True)
elif form == "terminal_obs_only":
    env.set_rollout_terminal_obs(True)
print("form:", form)
for t in range(60):
    env.rollout_torch(tape[t:t + 1], 2)
    if t >= 20:
        buf = np.zeros(((n + 63) // 64, 16), np.uint64)
        assert lib.hlx_debug_read_stamps(env._h, buf.ctypes.data) == 0
        d = np.diff(buf[:, :16].astype(np.int64), axis=1)
        acc.append(d)
d = np.concatenate(acc)
tot = d.sum(1)
# (s_memtime counts shader-clock cycles of the CU it runs on -- ~2.4 GHz here, so 17 000 ticks ~ 7 us; HLX_STAMP_REALTIME builds read the 100 MHz counter instead)
print(f"n={n} physics={physics}: median wave lifetime {np.median(tot):.0f} s_memtime ticks (shader-clock cycles)")
q = np.percentile(tot, [10, 50, 90, 99, 100])
print("  wave lifetime percentiles p10/p50/p90/p99/max:", " ".join(f"{x:.0f}" for x in q))
slow = tot >= np.percentile(tot, float(os.environ.get("HLX_STAMP_SLOW_PCT", "90")))
print(f"  slowest waves (>= p{os.environ.get('HLX_STAMP_SLOW_PCT', '90')}): median lifetime {np.median(tot[slow]):.0f}; per segment (median ticks, slow waves vs all):")
for k, nm in enumerate(names[:d.shape[1]]):
    print(f"    {nm:24s} {np.median(d[slow, k]):8.0f} vs {np.median(d[:, k]):8.0f}")
# per launch: first start / last end over the waves (absolute stamps), i.e. how long the slowest wave keeps the launch open
for k, nm in enumerate(names[:d.shape[1]]):
    print(f"  {nm:24s} median {np.median(d[:, k]):8.0f} ticks  share {100*np.median(d[:, k]/tot):5.1f}%")
env.close()
