#!/usr/bin/env python3
"""Golden vectors for the HRL controller logic: the reference's own
`HierarchicalManager.select_action` (rl_system/hrl/manager.py:113-203) with its rule-based selector
(hrl/selector_policy.py:162-200), `abstract_observation` / `extract_env_state_for_transitions`
(hrl/observation_abstraction.py:19-130) and `OptionManager.get_forced_transition` (hrl/option_manager.py:62-139)
driven over observation sequences.  Runs ONLY in the build container (needs /root/reference); writes
tests/golden/hrl/*.npz (plain data).

Observation sequences: (a) the 26-D observations recorded in the step fixtures of this directory (real
trajectories incl. episode ends, where the wrapper resets the manager: hrl/wrappers.py:60-62), (b) synthetic sweeps
that walk lock quality / range / fuel across every threshold and hysteresis band.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF = "/root/reference/rl_system"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "hrl")


def _install_gym_shim():
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:
        metadata = {}

    class Wrapper:
        def __init__(self, env=None):
            self.env = env

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    gym.Env, gym.Wrapper, gym.ObservationWrapper, gym.RewardWrapper = Env, Wrapper, Wrapper, Wrapper
    spaces.Box, spaces.Discrete = Box, Discrete
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces


def synthetic_sequence(rng, T):
    """Piecewise-smooth walks of the few entries the controller reads, crossing every threshold repeatedly."""
    obs = np.zeros((T, 26), np.float32)
    lock = 0.1
    dist = 3000.0
    fuel = 1.0
    for t in range(T):
        phase = (t // 60) % 8
        lock += {0: 0.02, 1: 0.0, 2: -0.004, 3: 0.01, 4: -0.03, 5: 0.03, 6: 0.0, 7: -0.01}[phase] + 0.01 * rng.standard_normal()
        lock = float(np.clip(lock, 0.0, 1.0))
        dist += {0: -25.0, 1: -30.0, 2: -12.0, 3: 4.0, 4: 6.0, 5: -6.0, 6: -1.5, 7: 20.0}[phase] + 3.0 * rng.standard_normal()
        dist = float(np.clip(dist, 5.0, 6000.0))
        fuel = max(0.0, fuel - (0.0015 if t > T // 2 else 0.0003))
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        obs[t, 0:3] = (d * dist).astype(np.float32)                 # the wrapper feeds obs-derived "distance" (metres here)
        obs[t, 3:6] = rng.normal(0, 0.2, 3)
        obs[t, 6:12] = rng.normal(0, 0.3, 6)
        obs[t, 12] = fuel
        obs[t, 13] = rng.uniform(-1, 12)                            # time-to-intercept entry, beyond its clip range too
        obs[t, 14] = lock
        obs[t, 15] = rng.normal(0, 400)                             # closing rate, beyond +-500 sometimes
        obs[t, 16] = rng.uniform(-4, 4)                             # off-axis, beyond +-pi sometimes
        obs[t, 17:26] = rng.uniform(-2, 1, 9)
    return obs


def run_case(name, obs_seq, did_reset, decision_interval=100, forced=True, hysteresis=True, min_dwell=True, stack=1):
    from hrl.manager import HierarchicalManager
    from hrl.observation_abstraction import extract_env_state_for_transitions
    from hrl.selector_policy import SelectorPolicy

    mgr = HierarchicalManager(selector=SelectorPolicy(mode="rules"), decision_interval=decision_interval,
                              enable_forced_transitions=forced, enable_hysteresis=hysteresis, enable_min_dwell=min_dwell)
    T = len(obs_seq)
    rec = dict(abstract=np.zeros((T, 7), np.float32), option=np.zeros(T, np.int32), switched=np.zeros(T, np.int32),
               reason=np.zeros(T, np.int32), forced=np.zeros(T, np.int32), choice=np.zeros(T, np.int32),
               steps_in_option=np.zeros(T, np.int32), total_steps=np.zeros(T, np.int32),
               env_distance=np.zeros(T, np.float64))
    reasons = {"continue": 0, "selector": 1, "forced": 2}
    hist = [np.zeros(26, np.float32)] * (stack - 1)
    for t in range(T):
        frame = obs_seq[t]
        full = np.concatenate(hist[-(stack - 1):] + [frame]) if stack > 1 else frame
        env_state = extract_env_state_for_transitions(full, env_info=None)        # hrl/wrappers.py:104
        _, info = mgr.select_action(full, env_state)
        rec["abstract"][t] = np.array(info["hrl/abstract_state"], np.float32)
        rec["option"][t] = info["hrl/option_index"]
        rec["switched"][t] = int(info["hrl/option_switched"])
        rec["reason"][t] = reasons[info["hrl/switch_reason"]]
        rec["forced"][t] = int(info["hrl/forced_transition"])
        rec["choice"][t] = info["hrl/selector_choice"]
        rec["steps_in_option"][t] = info["hrl/steps_in_option"]
        rec["total_steps"][t] = info["hrl/total_steps"]
        rec["env_distance"][t] = env_state["distance"]
        hist.append(frame)
        if did_reset[t]:                                                          # episode ended: wrapper.reset() -> manager.reset()
            mgr.reset()
            hist = [np.zeros(26, np.float32)] * (stack - 1)
    stats = mgr.get_statistics()["transition_stats"]
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), obs=obs_seq.astype(np.float32), did_reset=np.asarray(did_reset, bool),
                        decision_interval=np.int64(decision_interval), forced_enabled=np.int64(forced), hysteresis=np.int64(hysteresis),
                        min_dwell=np.int64(min_dwell), stack=np.int64(stack), **rec)
    print(f"{name:36s} T={T:5d} switches={int(rec['switched'].sum()):4d} forced={int((rec['reason'] == 2).sum()):4d} "
          f"options={np.bincount(rec['option'], minlength=3).tolist()} stats={stats}")


def main():
    _install_gym_shim()
    sys.path.insert(0, REF)
    rng = np.random.default_rng(2024)
    for fx in ("medium_base_pursuit", "eval360_los_fuze_pursuit", "medium_v2_pursuit", "volley3_medium_v2_fuze_pursuit"):
        d = np.load(os.path.join(HERE, fx + ".npz"))
        run_case("hrl_" + fx, d["obs"], d["did_reset"], decision_interval=100)
    d = np.load(os.path.join(HERE, "medium_base_pursuit.npz"))
    run_case("hrl_pursuit_interval10_stack4", d["obs"], d["did_reset"], decision_interval=10, stack=4)
    syn = synthetic_sequence(rng, 1500)
    no_reset = np.zeros(len(syn), bool)
    resets = no_reset.copy()
    resets[[299, 700, 1111]] = True
    run_case("hrl_synthetic_default", syn, resets, decision_interval=100)
    run_case("hrl_synthetic_interval7", syn, no_reset, decision_interval=7)
    run_case("hrl_synthetic_no_hysteresis", syn, resets, decision_interval=50, hysteresis=False)
    run_case("hrl_synthetic_no_min_dwell", syn, resets, decision_interval=50, min_dwell=False)
    run_case("hrl_synthetic_no_forced", syn, resets, decision_interval=20, forced=False)


if __name__ == "__main__":
    main()
