"""One-off soak: single-step launches (LATE instantiation at this size) against the fused rollout over thousands of steps
at the benchmark size, every output bit compared; plus finiteness / range invariants.  python tools/soak_equal.py [physics] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

physics = sys.argv[1] if len(sys.argv) > 1 else "v2dr"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n, chunk = 65536, 250
a = HlynrVecEnv(scenario_config("medium", physics), num_envs=n, seed=21)
b = HlynrVecEnv(scenario_config("medium", physics), num_envs=n, seed=21)
b.set_rollout_fused(50)
g = torch.Generator(device=a.device).manual_seed(3)
assert torch.equal(a.reset_torch(), b.reset_torch())
dones = 0
for c in range(T // chunk):
    tape = torch.rand((chunk, n, 6), generator=g, device=a.device) * 2 - 1
    tape[:, :, 2] = tape[:, :, 2].abs()          # thrust up: longer, more eventful episodes
    oa = [x.clone() for x in a.rollout_torch(tape, chunk)]
    ob = [x.clone() for x in b.rollout_torch(tape, chunk)]
    for x, y in zip(oa, ob):
        assert torch.equal(x, y), f"chunk {c}: single-step and fused outputs differ"
    assert torch.isfinite(oa[0]).all() and torch.isfinite(oa[1]).all()
    assert float(oa[0].max()) <= 1.0 + 1e-6 and float(oa[0].min()) >= -2.0 - 1e-6
    dones += int(oa[2].sum() + oa[3].sum())
    print(f"chunk {c}: ok, episodes ended so far {dones}", flush=True)
assert bytes(a.get_state()) == bytes(b.get_state())
print("soak ok:", physics, T, "steps x", n, "envs; episodes ended:", dones)
