set -e
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base us/step', round(d['ms_per_step']*1000,2), 'fused', round(d['fused_rollout']['ms_per_step']*1000,2))"
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
for over in ({}, {"volley_mode": True, "volley_size": 3}):
    env = HlynrVecEnv(scenario_config("medium", "base", dict(over, observation_mode="world_frame", rotation_invariant=False) if False else over), num_envs=65536, seed=1)
    tape = torch.rand((500, 65536, 6), device=env.device) * 2 - 1
    env.reset_torch(); env.rollout_torch(tape[:100], 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_torch(tape, 4); torch.cuda.synchronize()
    print(env.kernel_variant, over, 'us/step', round((time.perf_counter() - t0) / 500 * 1e6, 2))
    env.close()
PY
