"""One extra roofline point of bench.py in a FRESH process: create -> desynchronise -> time -> close, nothing else allocated.
(At HBM-bound sizes the step's time depends on the process's whole allocation history -- 527-531 us for the 4 M-env batch in
four fresh processes, 545-560 us behind other work in the same process, profiles/r03_placement_probe_negative.txt --, so
the cache-defeating point of SURVEY.md 8(d) is measured where that history is always the same.)
  python tools/extra_point.py PHYSICS N_ENVS STEPS DESYNC  -> one JSON object per form on stdout"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from bench import BYTES_PER_ENV_STEP, FORM_BYTES_DELTA, HBM_PEAK_GBS, measured_traffic, tape_schedule
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    phys, n, k, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    device = int(os.environ.get("LOCAL_RANK", "0"))
    env = HlynrVecEnv(resolved=resolve_config(scenario_config("medium", phys)), num_envs=n, device=device, seed=1000)
    gen = torch.Generator(device=env.device).manual_seed(0)
    tape = torch.rand((min(k, 64), n, 6), generator=gen, device=env.device, dtype=torch.float32) * 2.0 - 1.0
    env.reset_torch()
    if d:
        env.set_rollout_fused(64)
        for _ in range(d // tape.shape[0]):
            env.rollout_torch(tape, 8)
        env.set_rollout_fused(1)
    out = []
    for form in ("contract", "single_pass"):
        env.set_rollout_contract(form == "contract", done_list=True)
        env.rollout_torch(tape[:max(1, min(k // 4, tape.shape[0]))], 8)
        torch.cuda.synchronize(env.device)
        t0 = time.perf_counter()
        for lo, hi in tape_schedule(k, tape.shape[0]):
            env.rollout_torch(tape[lo:hi], 8)
        torch.cuda.synchronize(env.device)
        dt = time.perf_counter() - t0
        b, bf = BYTES_PER_ENV_STEP[phys], BYTES_PER_ENV_STEP[phys] + FORM_BYTES_DELTA[form]     # SURVEY.md 8(d) bytes; what this form stores
        out.append({"workload": f"medium scenario, {phys} physics, {n} envs/GPU", "form": form, "value": n * k / dt, "unit": "env-steps/s",
                    "us_per_step": 1e6 * dt / k, "algorithmic_bytes_per_env_step": b, "form_bytes_per_env_step": bf, "desync_steps": d,
                    "steps": k, "process": "fresh", "roofline_frac": n * k * b / dt / 1e9 / HBM_PEAK_GBS,
                    "roofline_frac_form_bytes": n * k * bf / dt / 1e9 / HBM_PEAK_GBS,
                    "traffic": measured_traffic(phys, n, form)[0]})      # HBM bytes per launch by counters for THIS workload and form (profiles/hbm_traffic.json), or None
    env.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
