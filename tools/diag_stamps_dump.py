"""Raw per-wave stamps of 40 steady-state launches (libhlx_stamps.so) -> gpurun_out/<dir>/stamps_raw.npz, for offline
analysis of WHICH waves keep a launch open (tools/diag_stamps_analyse.py)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HLX_LIBRARY"] = os.path.join(ROOT, "hlynr_intercept_amd", "libhlx_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = HlynrVecEnv(scenario_config("medium", "base"), num_envs=n, seed=1)
env.reset_torch()
lib = env._lib
lib.hlx_debug_read_stamps.restype = C.c_int
lib.hlx_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device=env.device).manual_seed(0)
tape = torch.rand((64, n, 6), generator=g, device=env.device) * 2 - 1
env.set_rollout_fused(64)
for _ in range(64):
    env.rollout_torch(tape, 2)
env.set_rollout_fused(1)
raw, dones = [], []
for t in range(60):
    o = env.rollout_torch(tape[t:t + 1], 1)
    if t >= 20:
        buf = np.zeros(((n + 63) // 64, 16), np.uint64)
        assert lib.hlx_debug_read_stamps(env._h, buf.ctypes.data) == 0
        raw.append(buf.copy())
        term, trunc = o[2], o[3]
        dones.append(((term[0] | trunc[0]).view(-1, 64).sum(1)).cpu().numpy() if n % 64 == 0 else np.zeros(1))
np.savez_compressed(out, raw=np.stack(raw), dones=np.stack(dones))
print("saved", out)
env.close()
