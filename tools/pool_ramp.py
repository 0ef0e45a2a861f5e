"""What a curriculum that moves at every step costs the step launch: `set_training_step_count`
before every step, walking the reference's beam-width ramp (config.yaml:85-92, 5 M -> 8 M global steps) -- and, with
`reliability`, a made-up ramp of the two sensor reliabilities over the same range -- against the same loop at a fixed step count.
  python tools/pool_ramp.py [N_ENVS ...]   -> us per step (HIP events around 1000 hlx_step calls from Python), pool statistics"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n, mode, steps=1000):
    import torch
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    over = {}
    if mode == "reliability":
        rcu = "curriculum.radar_curriculum."
        over = {rcu + "final_detection_reliability": 0.6, rcu + "reliability_transition_start": 5_000_000, rcu + "reliability_transition_end": 8_000_000,
                rcu + "final_ground_reliability": 0.7, rcu + "ground_reliability_transition_start": 5_000_000,
                rcu + "ground_reliability_transition_end": 8_000_000}
    env = HlynrVecEnv(scenario_config("medium", "base", over), num_envs=n, seed=3)
    g = torch.Generator(device=env.device).manual_seed(0)
    tape = torch.rand((64, n, 6), generator=g, device=env.device) * 2 - 1
    env.set_training_step_count(5_000_000)
    env.reset_torch()
    env.set_rollout_fused(64)
    for _ in range(32):
        env.rollout_torch(tape, 2)          # desynchronise the episodes
    env.set_rollout_fused(1)
    out = {}
    for label, moving in (("fixed step count", False), ("moving every step", True), ("fixed again", False)):
        for t in range(200):
            env.step_torch(tape[t % 64])
        s0 = env.episode_pool_stats()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for t in range(steps):
            if moving:
                env.set_training_step_count(5_000_000 + 3000 * t)     # 1000 steps across the whole 3 M-step ramp
            env.step_torch(tape[t % 64])
        ev1.record()
        torch.cuda.synchronize()
        s1 = env.episode_pool_stats()
        out[label] = (1e3 * ev0.elapsed_time(ev1) / steps, {k: s1[k] - s0[k] for k in s1})
        if moving:
            env.set_training_step_count(8_000_000)
    env.close()
    return out


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [4096, 65536]
    for n in sizes:
        for mode in ("beam", "reliability"):
            res = run(n, mode)
            base = res["fixed step count"][0]
            for label, (us, st) in res.items():
                print(f"{n:6d} envs, {mode:11s} ramp, {label:18s}: {us:7.3f} us per step ({us - base:+.3f})  pool: {st}", flush=True)


if __name__ == "__main__":
    main()
