# quick correctness probe on a minimal v2dr library: the pool on/off forms and fused rollout must agree bit for bit, and the oracle self-check of bench.py
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
n, T = 1300, 150
cfg = scenario_config("medium", "v2dr", {"max_steps": 23})
envs = [HlynrVecEnv(cfg, num_envs=n, seed=9) for _ in range(3)]
envs[0].set_episode_pool(0); envs[1].set_episode_pool(1); envs[2].set_episode_pool(8)
for e in envs: e.reset_torch()
g = torch.Generator(device=envs[0].device).manual_seed(3)
tape = torch.rand((T, n, 6), generator=g, device=envs[0].device) * 2 - 1
for t in range(T):
    ref = None
    for e in envs:
        o, r, te, tr, info = e.step_torch(tape[t], want_done_list=True)
        cur = [o.clone(), r.clone(), te.clone(), tr.clone(), info["flags"].clone(), info["distance"].clone()]
        if ref is None: ref = cur
        else:
            for k, (a, b) in enumerate(zip(ref, cur)):
                assert torch.equal(a, b), (t, k)
s0 = bytes(envs[0].get_state())
assert all(bytes(e.get_state()) == s0 for e in envs[1:])
# state round trip keeps going identically (ring planes re-mapped by the host)
a, b = envs[0], envs[1]
b.set_state(a.get_state())
for t in range(30):
    oa = a.step_torch(tape[t])[0].clone(); ob = b.step_torch(tape[t])[0].clone()
    assert torch.equal(oa, ob), t
# fused rollout against single steps
c = HlynrVecEnv(cfg, num_envs=n, seed=9); d = HlynrVecEnv(cfg, num_envs=n, seed=9)
c.reset_torch(); d.reset_torch(); d.set_rollout_fused(8)
oc = [x.clone() for x in c.rollout_torch(tape[:64], 64)]; od = [x.clone() for x in d.rollout_torch(tape[:64], 64)]
for x, y in zip(oc, od): assert torch.equal(x, y)
assert bytes(c.get_state()) == bytes(d.get_state())
import hashlib
print("state sha256", hashlib.sha256(s0).hexdigest()[:16], hashlib.sha256(bytes(c.get_state())).hexdigest()[:16])
print("v2dr forms agree; misses", envs[1].episode_pool_misses(), "crowded", envs[1].episode_pool_crowded())
