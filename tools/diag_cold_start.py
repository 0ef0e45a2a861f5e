"""Diagnostic: what do the first launches behind a host synchronisation cost?  Host-side time of each hlx_rollout(T=1) call and
the spacing of the kernels on the GPU (one HIP event behind every launch), for 24 launches issued right after a
torch.cuda.synchronize(), in the contract form, steady state.   python tools/diag_cold_start.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv

n, K = 65536, 24
env = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", "base")), num_envs=n, seed=1000)
g = torch.Generator(device=env.device).manual_seed(0)
tape = torch.rand((64, n, 6), generator=g, device=env.device) * 2 - 1
env.reset_torch()
env.set_rollout_fused(64)
for _ in range(64):
    env.rollout_torch(tape, 8)
env.set_rollout_fused(1)
env.set_rollout_contract(True, done_list=True)
for _ in range(30):
    env.rollout_torch(tape, 8)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
for e in evs:
    e.record()
for idle_ms in (0.0, 1.0, 20.0):
    host = []
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    t_prev = time.perf_counter_ns()
    evs[0].record()
    for k in range(K):
        env.rollout_torch(tape[k:k + 1], 8)
        evs[k + 1].record()
        t = time.perf_counter_ns()
        host.append((t - t_prev) / 1e3)
        t_prev = t
    torch.cuda.synchronize()
    gpu = [1e3 * evs[k].elapsed_time(evs[k + 1]) for k in range(K)]
    print(f"idle {idle_ms:5.1f} ms before: host us per call {[round(x, 1) for x in host]}")
    print(f"                      GPU us between events {[round(x, 1) for x in gpu]}", flush=True)
# the benchmark's own case: ONE C call that issues K launches right behind a synchronisation
for K2 in (20, 20, 100, 400):
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns()
    env.rollout_torch(tape[:K2] if K2 <= 64 else torch.cat([tape] * (K2 // 64 + 1))[:K2].contiguous(), 8)
    t1 = time.perf_counter_ns()
    torch.cuda.synchronize()
    t2 = time.perf_counter_ns()
    print(f"one hlx_rollout call of {K2} launches behind a synchronisation: host issue {1e-3 * (t1 - t0):.1f} us ({1e-3 * (t1 - t0) / K2:.2f} per launch), "
          f"until synchronised {1e-3 * (t2 - t0):.1f} us ({1e-3 * (t2 - t0) / K2:.2f} per step)", flush=True)
env.close()
