"""TEST INFRASTRUCTURE ONLY: numpy restatement, vectorised over N environments, of the reference's HRL controller
logic --

    abstract_observation / extract_env_state_for_transitions   rl_system/hrl/observation_abstraction.py:19-130
    OptionManager.get_forced_transition / _is_critical_transition   rl_system/hrl/option_manager.py:62-172
    HierarchicalManager.select_action / _switch_option / reset   rl_system/hrl/manager.py:84-221
    SelectorPolicy._rule_based_selection                          rl_system/hrl/selector_policy.py:162-200
    thresholds, hysteresis bands, min-dwell steps                 rl_system/hrl/option_definitions.py:47-86

PARITY PINNED by tests/golden/hrl/*.npz, recorded from the reference's own classes by
tests/golden/make_hrl_golden.py (tests/test_hrl_oracle.py).  Mixed precision follows numpy >= 2 (NEP 50): the
observation is float32; `float(obs[k])` values and the env_state dict are Python floats (float64); np.float32
scalars divided by Python floats stay float32.
"""
from __future__ import annotations

import numpy as np

SEARCH, TRACK, TERMINAL = 0, 1, 2
REASON_CONTINUE, REASON_SELECTOR, REASON_FORCED = 0, 1, 2

DEFAULTS = dict(
    lock_min=0.3, lock_search=0.7, close_range=200.0, terminal_fuel_min=0.1, miss_imminent=400.0, fuel_critical=0.10,
    h_lock_acquire=0.75, h_lock_maintain=0.55, h_terminal_enter=200.0, h_terminal_exit=250.0,
    min_dwell=(50, 50, 30),
)


def _norm3_f32(v):
    """np.linalg.norm of float32 3-vectors = sqrt(x.dot(x)); OpenBLAS sdot as measured in the build container:
    float32-ROUNDED products, summed in float64, rounded once (oracle/hlx_oracle.c dot3), then a float32 sqrt."""
    v = v.astype(np.float32)
    p = (v * v).astype(np.float64)      # products rounded to float32
    s = (p[:, 0] + p[:, 1]) + p[:, 2]
    return np.sqrt(s.astype(np.float32)).astype(np.float32)


def abstract_observation(obs26):
    """[N, 26] float32 -> [N, 7] float32 (observation_abstraction.py:19-78)."""
    o = np.asarray(obs26, np.float32)
    dist = _norm3_f32(o[:, 0:3])
    a = np.zeros((o.shape[0], 7), np.float32)
    a[:, 0] = np.clip(dist / np.float32(5000.0), np.float32(0), np.float32(1))               # float32 / weak float
    a[:, 1] = np.clip(o[:, 15].astype(np.float64) / 500.0, -1, 1)                            # float(obs[15]) / 500.0
    a[:, 2] = o[:, 14]
    a[:, 3] = o[:, 12]
    a[:, 4] = np.clip(o[:, 16].astype(np.float64) / np.pi, -1, 1)
    a[:, 5] = np.clip(o[:, 13].astype(np.float64) / 10.0, 0, 1)
    a[:, 6] = np.clip(o[:, 2] / np.float32(1000.0), np.float32(-1), np.float32(1))           # np.float32 / weak float
    return a, dist


def rule_selector(abstract):
    """selector_policy.py:162-200 on float32 entries (comparisons against Python floats happen in float32)."""
    a = np.asarray(abstract, np.float32)
    dist_m = a[:, 0] * np.float32(5000.0)
    lock, fuel = a[:, 2], a[:, 3]
    out = np.full(a.shape[0], SEARCH, np.int32)
    good = ~(lock < np.float32(0.3))
    term = good & (dist_m < np.float32(100.0)) & (lock > np.float32(0.7)) & (fuel > np.float32(0.1))
    out[good & (lock >= np.float32(0.3))] = TRACK
    out[term] = TERMINAL
    return out


class Controller:
    def __init__(self, n, decision_interval=100, enable_forced=True, enable_hysteresis=True, enable_min_dwell=True,
                 default_option=SEARCH, selector="rules", **thresholds):
        self.n, self.D = int(n), int(decision_interval)
        self.forced, self.hyst, self.dwell = bool(enable_forced), bool(enable_hysteresis), bool(enable_min_dwell)
        self.default, self.selector = int(default_option), selector
        self.th = dict(DEFAULTS, **thresholds)
        self.option = np.full(self.n, self.default, np.int32)
        self.steps_in_option = np.zeros(self.n, np.int32)      # HRLState.steps_in_option
        self.om_steps = np.zeros(self.n, np.int32)             # OptionManager.steps_in_current_option
        self.total_steps = np.zeros(self.n, np.int32)

    def reset(self, mask=None):
        m = np.ones(self.n, bool) if mask is None else np.asarray(mask, bool)
        self.option[m] = self.default
        self.steps_in_option[m] = 0
        self.om_steps[m] = 0
        self.total_steps[m] = 0

    def forced_transition(self, lock, dist, fuel):
        """get_forced_transition for all envs: forced option or -1 (float64 comparisons: env_state holds Python floats)."""
        th, cur = self.th, self.option
        forced = np.full(self.n, -1, np.int32)
        critical = np.zeros(self.n, bool)
        if not self.forced:
            return forced
        lose = th["h_lock_maintain"] if self.hyst else th["lock_min"]
        acquire = th["h_lock_acquire"] if self.hyst else th["lock_search"]
        enter = th["h_terminal_enter"] if self.hyst else th["close_range"]
        leave = th["h_terminal_exit"] if self.hyst else th["miss_imminent"]
        forced[((cur == TRACK) | (cur == TERMINAL)) & (lock < lose)] = SEARCH                      # rule 1
        forced[(cur == SEARCH) & (lock >= acquire)] = TRACK                                          # rule 2
        r3a = (cur == TRACK) & (dist <= enter) & (fuel <= th["fuel_critical"])                      # rule 3a
        r3b = (cur == TRACK) & (dist <= enter) & (fuel >= th["terminal_fuel_min"]) & ~r3a          # rule 3b
        forced[r3a | r3b] = TERMINAL
        critical[r3a] = True
        forced[(cur == TERMINAL) & (dist > leave)] = TRACK                                           # rule 4
        if self.dwell:
            dwell = np.asarray(self.th["min_dwell"], np.int32)[cur]
            is_crit = ((forced == SEARCH) & (cur != SEARCH) & (lock < 0.2)) | (fuel < 0.1)         # _is_critical_transition
            blocked = (forced >= 0) & ~critical & (self.om_steps < dwell) & ~is_crit
            forced[blocked] = -1
        return forced

    def step(self, obs, done_prev=None, selector_choice=None):
        """One `select_action` for every env on the latest 26-D frame of `obs` [N, 26*k].
        `done_prev`: envs whose episode ended on the previous env step (their manager is reset first, as the
        wrapper's reset() does).  Returns dict(abstract, option, switched, reason, forced, choice, due)."""
        if done_prev is not None:
            self.reset(done_prev)
        o = np.asarray(obs, np.float32)[:, -26:]
        abstract, dist32 = abstract_observation(o)
        lock, fuel, dist = o[:, 14].astype(np.float64), o[:, 12].astype(np.float64), dist32.astype(np.float64)
        forced = self.forced_transition(lock, dist, fuel)
        due = ((self.total_steps + 1) % self.D) == 0
        if selector_choice is None:
            choice = rule_selector(abstract) if self.selector == "rules" else np.full(self.n, SEARCH, np.int32)
        else:
            choice = np.clip(np.asarray(selector_choice, np.int32), 0, 2)
        cur = self.option.copy()
        new = cur.copy()
        reason = np.zeros(self.n, np.int32)
        f = forced >= 0
        new[f] = forced[f]
        s = ~f & due
        new[s] = choice[s]
        switched = new != cur
        reason[f] = REASON_FORCED
        reason[s & switched] = REASON_SELECTOR
        self.option = new
        self.steps_in_option[switched] = 0
        self.om_steps[switched] = 0
        self.steps_in_option += 1
        self.total_steps += 1
        self.om_steps += 1
        return dict(abstract=abstract, option=new.copy(), switched=switched, reason=reason, forced=f,
                    choice=np.where(s, choice, new), due=due, steps_in_option=self.steps_in_option.copy(),
                    total_steps=self.total_steps.copy(), env_distance=dist)
