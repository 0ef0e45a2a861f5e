"""Experiment (round 4): what would phase-shifted sub-batches buy?  The cheapest faithful probe:
S independent handles of 65 536 / S environments each, one host thread and one HIP stream per handle, every thread
issuing contract-form rollouts back to back (no cross-stream synchronisation at all: the upper bound of any fork / join
scheme), against ONE handle of 65 536 environments on one stream.  Prints env-steps/s of each arrangement.
  python tools/experiments/split_handles.py [STEPS]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv

    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    total = 65536
    rc = resolve_config(scenario_config("medium", "base"))
    dev = torch.device("cuda", 0)
    for rep in range(2):
        for S in (1, 2, 4, 1):
            n = total // S
            envs = [HlynrVecEnv(resolved=rc, num_envs=n, seed=1000, env_id_offset=s * n) for s in range(S)]
            gen = torch.Generator(device=dev).manual_seed(0)
            tapes = [torch.rand((256, n, 6), generator=gen, device=dev) * 2 - 1 for _ in range(S)]
            streams = [torch.cuda.Stream(dev) for _ in range(S)]
            for e, tp in zip(envs, tapes):
                e.reset_torch()
                e.set_rollout_fused(64)
                for _ in range(16):
                    e.rollout_torch(tp, 8)
                e.set_rollout_fused(1)
                e.set_rollout_contract(True, done_list=True)
                e.rollout_torch(tp, 8)
            torch.cuda.synchronize(dev)
            barrier = threading.Barrier(S + 1)

            def work(e, tp, st):
                with torch.cuda.stream(st):
                    e.rollout_torch(tp, 8)          # buffers of this stream's launches
                    st.synchronize()
                    barrier.wait()
                    for _ in range(K // 256):
                        e.rollout_torch(tp, 8)
                    st.synchronize()
                barrier.wait()

            th = [threading.Thread(target=work, args=(e, tp, st)) for e, tp, st in zip(envs, tapes, streams)]
            for t in th:
                t.start()
            barrier.wait()
            t0 = time.perf_counter()
            barrier.wait()
            dt = time.perf_counter() - t0
            for t in th:
                t.join()
            steps = (K // 256) * 256
            print(f"S={S} handles x {n} envs, {steps} steps each: {1e6 * dt / steps:.3f} us per vec-step of {total} envs, "
                  f"{total * steps / dt:.4g} env-steps/s, sched {envs[0].load_schedule}", flush=True)
            for e in envs:
                e.close()


if __name__ == "__main__":
    main()
