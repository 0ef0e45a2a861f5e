"""The on-device HRL controller (include/hlx_hrl.h, hlynr_intercept_amd/hrl.py) against (1) the golden vectors recorded
from the reference's own manager and (2) the numpy restatement on random batches.  Discrete decisions and the clipped
ratios of the abstract state: bit-exact."""
import glob
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "hrl", "*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gpu_controller_matches_reference_fixture(path):
    import torch
    from hlynr_intercept_amd.hrl import HRLController
    from tests.test_hrl_oracle import stacked
    fx = np.load(path)
    stack = int(fx["stack"])
    obs = stacked(fx["obs"], fx["did_reset"], stack)
    n = 5    # the same sequence in five lanes
    c = HRLController(n, obs_dim=26 * stack, decision_interval=int(fx["decision_interval"]),
                      enable_forced_transitions=bool(fx["forced_enabled"]), enable_hysteresis=bool(fx["hysteresis"]),
                      enable_min_dwell=bool(fx["min_dwell"]))
    dev = c.device
    obs_d = torch.tensor(obs, device=dev)
    done = torch.tensor(fx["did_reset"].astype(np.uint8), device=dev)
    zero = torch.zeros(n, dtype=torch.uint8, device=dev)
    opts, infos, abstracts = [], [], []
    for t in range(len(obs)):
        prev = done[t - 1].expand(n).contiguous() if t > 0 else zero
        o, a, i = c.step(obs_d[t].expand(n, -1).contiguous(), prev, zero)
        opts.append(o.clone()); infos.append(i.clone()); abstracts.append(a.clone())
    opts, infos, abstracts = torch.stack(opts).cpu().numpy(), torch.stack(infos).cpu().numpy(), torch.stack(abstracts).cpu().numpy()
    assert np.all(opts == opts[:, :1]) and np.all(infos == infos[:, :1])
    assert np.array_equal(opts[:, 0], fx["option"].astype(np.uint8))
    assert np.array_equal(infos[:, 0] & 1, fx["switched"].astype(np.uint8))
    assert np.array_equal((infos[:, 0] >> 1) & 3, fx["reason"].astype(np.uint8))
    assert np.array_equal((infos[:, 0] >> 3) & 1, fx["forced"].astype(np.uint8))
    assert np.array_equal((infos[:, 0] >> 5) & 3, fx["choice"].astype(np.uint8))
    assert np.array_equal(abstracts[:, 0], fx["abstract"])                     # bit-exact float32
    st = c.get_state()
    assert st[0, 1] == fx["steps_in_option"][-1] and st[0, 3] == fx["total_steps"][-1]
    c.close()


@pytest.mark.parametrize("mode", ["rules", "external", "no_hysteresis"])
def test_gpu_controller_matches_oracle_on_random_batches(mode):
    import torch
    from hlynr_intercept_amd.hrl import HRLController
    from oracle.hrl_controller import Controller
    n, T = 4099, 400
    rng = np.random.default_rng(5)
    kw = dict(decision_interval=13, enable_hysteresis=mode != "no_hysteresis")
    ext = None
    if mode == "external":     # a stand-in selector network: any function of the abstract state (out-of-range values get clamped)
        ext = lambda a: (a[:, 2] * 4.0 - 0.5).floor().to(torch.int32)  # noqa: E731
    g = HRLController(n, obs_dim=104, selector=ext if ext else "rules", **kw)
    o = Controller(n, selector="rules", **kw)
    lock = rng.random(n).astype(np.float32)
    dist = rng.uniform(50, 600, n).astype(np.float32)
    fuel = np.ones(n, np.float32)
    done = np.zeros(n, bool)
    for t in range(T):
        lock = np.clip(lock + rng.normal(0, 0.05, n), 0, 1).astype(np.float32)
        dist = np.clip(dist + rng.normal(-1, 12, n), 1, 900).astype(np.float32)
        fuel = np.clip(fuel - rng.uniform(0, 0.006, n), 0, 1).astype(np.float32)
        obs = rng.uniform(-2, 1, (n, 104)).astype(np.float32)
        d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        obs[:, 78:81] = (d * dist[:, None]).astype(np.float32)
        obs[:, 78 + 12], obs[:, 78 + 14] = fuel, lock
        obs[:, 78 + 13] = rng.uniform(-2, 14, n); obs[:, 78 + 15] = rng.normal(0, 400, n); obs[:, 78 + 16] = rng.uniform(-4, 4, n)
        obs_d = torch.tensor(obs, device=g.device)
        done_d = torch.tensor(done.astype(np.uint8), device=g.device)
        choice = None
        if ext:
            from oracle.hrl_controller import abstract_observation
            a, _ = abstract_observation(obs[:, -26:])
            choice = np.floor(a[:, 2] * np.float32(4.0) - np.float32(0.5)).astype(np.int32)
        opt, a_d, info = g.step(obs_d, done_d, None)
        r = o.step(obs, done, choice)
        assert np.array_equal(a_d.cpu().numpy(), r["abstract"]), t
        assert np.array_equal(opt.cpu().numpy(), r["option"].astype(np.uint8)), t
        ib = info.cpu().numpy()
        assert np.array_equal(ib & 1, r["switched"].astype(np.uint8)) and np.array_equal((ib >> 1) & 3, r["reason"].astype(np.uint8))
        assert np.array_equal((ib >> 4) & 1, r["due"].astype(np.uint8))
        done = rng.random(n) < 0.01
        fuel[done] = 1.0
    st = g.get_state()
    assert np.array_equal(st[:, 0], o.option) and np.array_equal(st[:, 1], o.steps_in_option)
    assert np.array_equal(st[:, 2], o.om_steps) and np.array_equal(st[:, 3], o.total_steps)
    assert len(np.unique(st[:, 0])) == 3                                        # all three options in use at the end
    g.close()


def test_select_actions_groups_specialists_by_option():
    import torch
    from hlynr_intercept_amd.hrl import HRLController, SEARCH, TRACK, TERMINAL
    from hlynr_intercept_amd.scenarios import scenario_config
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    n = 512
    env = HlynrVecEnv(scenario_config("medium", "base", {"max_steps": 60}), num_envs=n, seed=4)
    ctl = HRLController(n, decision_interval=10)
    calls = {SEARCH: 0, TRACK: 0, TERMINAL: 0}

    def make(k, thrust):
        def f(rows):
            calls[k] += rows.shape[0]
            a = torch.zeros((rows.shape[0], 6), device=rows.device)
            a[:, 2] = thrust
            return a
        return f

    spec = {SEARCH: make(SEARCH, 0.1), TRACK: make(TRACK, 0.5), TERMINAL: make(TERMINAL, 0.9)}
    obs = env.reset_torch()
    term = trunc = None
    for t in range(80):
        actions, option, info = ctl.select_actions(obs, spec, term, trunc)
        want = torch.tensor([0.1, 0.5, 0.9], device=obs.device)[option.long()]
        assert torch.equal(actions[:, 2], want)                                  # every env got ITS specialist's action
        obs, rew, term, trunc, _ = env.step_torch(actions)
    assert sum(calls.values()) == 80 * n and calls[TERMINAL] > 0                 # each env served by exactly one specialist per step
    d = HRLController.decode_info(int(info[0]))
    assert set(d) == {"hrl/option_switched", "hrl/switch_reason", "hrl/forced_transition", "hrl/selector_due", "hrl/selector_choice"}
    env.close(); ctl.close()


def test_recurrent_specialists_keep_one_live_lstm_state_per_environment():
    """manager.py:104-107 (reset: every specialist's LSTM state dropped) and :210-215 (switch: the old option's state
    dropped) leave each environment with exactly one live state, started afresh when an option is entered.  Emulated here
    per environment in Python (options from the pinned oracle controller), against the batched on-device state bank."""
    import torch
    from hlynr_intercept_amd.hrl import HRLController
    from oracle.hrl_controller import Controller
    n, T, H = 777, 300, 5
    rng = np.random.default_rng(11)
    kw = dict(decision_interval=9)
    g = HRLController(n, obs_dim=26, **kw)
    o = Controller(n, selector="rules", **kw)
    gains = {0: 0.25, 1: 0.5, 2: 0.75}          # three different "networks": h' = gain * h + mean(obs); c' = c + 1

    def make(k):
        def f(rows, state, starts):
            L = 2
            if state is None:
                state = (torch.zeros((L, rows.shape[0], H), device=rows.device), torch.zeros((L, rows.shape[0], H), device=rows.device))
            h, c = state
            keep = (~starts).to(h.dtype)[None, :, None]                      # episode_start -> zeros (RecurrentPPO semantics)
            h2 = gains[k] * h * keep + rows.mean(1)[None, :, None]
            c2 = c * keep + 1.0
            act = torch.zeros((rows.shape[0], 6), device=rows.device)
            act[:, 0] = h2[1, :, 0]; act[:, 1] = c2[0, :, 0]; act[:, 2] = float(k)
            return act, (h2, c2)
        return f

    spec = {k: make(k) for k in gains}
    h_ref = np.zeros(n, np.float64); c_ref = np.zeros(n, np.float64)
    live = np.zeros(n, bool)                                                    # a state exists (not None)
    lock = rng.random(n).astype(np.float32); dist = rng.uniform(50, 600, n).astype(np.float32)
    done = np.zeros(n, bool)
    for t in range(T):
        lock = np.clip(lock + rng.normal(0, 0.08, n), 0, 1).astype(np.float32)
        dist = np.clip(dist + rng.normal(-1, 15, n), 1, 900).astype(np.float32)
        obs = rng.uniform(-1, 1, (n, 26)).astype(np.float32)
        d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        obs[:, 0:3] = (d * dist[:, None]).astype(np.float32); obs[:, 12] = 0.8; obs[:, 14] = lock
        r = o.step(obs, done, None)
        # reference semantics, one environment at a time
        live[done] = False                                                      # manager.reset(): all states None
        live[r["switched"]] = False                                             # the specialist entered has no state yet
        gain = np.array([gains[int(k)] for k in r["option"]])
        m = obs.astype(np.float32).mean(1, dtype=np.float32).astype(np.float64)
        h_ref = np.where(live, gain * h_ref, 0.0) + m
        c_ref = np.where(live, c_ref, 0.0) + 1.0
        live[:] = True
        acts, opt, info = g.select_actions_recurrent(torch.tensor(obs, device=g.device), spec,
                                                     torch.tensor(done.astype(np.uint8), device=g.device), None)
        a = acts.cpu().numpy()
        assert np.array_equal(opt.cpu().numpy(), r["option"].astype(np.uint8)), t
        assert np.array_equal(a[:, 2], r["option"].astype(np.float32)), t
        assert np.allclose(a[:, 0], h_ref, rtol=1e-5, atol=1e-5), (t, np.abs(a[:, 0] - h_ref).max())
        assert np.array_equal(a[:, 1], c_ref.astype(np.float32)), t              # exact: counts steps since the state began
        done = rng.random(n) < 0.02
    assert g.lstm_state[0].shape == (2, n, H) and c_ref.max() > 20 and (c_ref == 1).any()
    g.reset()                                                                    # HierarchicalManager.reset(): everything starts afresh
    acts, _, _ = g.select_actions_recurrent(torch.tensor(obs, device=g.device), spec, None, None)
    assert torch.all(acts[:, 1] == 1.0)
    g.close()


def test_option_major_regroup_moves_only_the_rows_that_must_move():
    """include/hlx_hrl.h hlx_hrl_regroup: after the options have moved, every option's environments are again one contiguous run of
    rows, `order` / `pos` are inverse permutations, every environment's state rows have followed it -- 1 KiB rows (16-byte copy
    units) and 20-byte rows (4-byte units) alike -- and only O(switches) rows were touched."""
    import torch
    from hlynr_intercept_amd.hrl import HRLController
    n, T = 5003, 40
    ctl = HRLController(n, obs_dim=26)
    dev = ctl.device
    ids = torch.arange(n, device=dev, dtype=torch.float32)
    wide = ids.view(1, n, 1).repeat(1, 1, 256).contiguous() + torch.arange(256, device=dev).view(1, 1, 256) * 1e-3      # [1, N, 256]: 1 KiB rows
    narrow = (ids.view(1, n, 1).repeat(2, 1, 5) * 2.0).contiguous() + torch.tensor([[[0.0]], [[0.5]]], device=dev)   # [2, N, 5]: 20-byte rows, two layers
    ctl.lstm_state = (wide, narrow)
    g = torch.Generator(device=dev).manual_seed(3)
    option = torch.zeros(n, dtype=torch.uint8, device=dev)
    for t in range(T):
        frac = 0.5 if t in (0, 17) else 0.02                           # two big reshuffles, otherwise 2 % of the environments switch
        change = torch.rand(n, generator=g, device=dev) < frac
        new = torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.uint8)
        changed = int((change & (new != option)).sum())
        option = torch.where(change, new, option).contiguous()
        counts = ctl._regroup(option)
        assert counts == torch.bincount(option.to(torch.int64), minlength=3).tolist(), t
        order, pos = ctl.order.to(torch.int64), ctl.pos.to(torch.int64)
        assert torch.equal(torch.sort(order).values, torch.arange(n, device=dev)) and torch.equal(pos[order], torch.arange(n, device=dev)), t
        runs = option[order].to(torch.int64)
        assert bool((runs[1:] >= runs[:-1]).all()), t                    # option-major: three contiguous runs
        e = order.to(torch.float32)
        assert torch.equal(ctl.lstm_state[0][0, :, 0], e) and torch.equal(ctl.lstm_state[0][0, :, 255], e + (torch.arange(256, device=dev) * 1e-3)[255]), t
        assert torch.equal(ctl.lstm_state[1][0, :, 4], e * 2.0) and torch.equal(ctl.lstm_state[1][1, :, 0], e * 2.0 + 0.5), t
        assert ctl.rows_moved <= 4 * changed + 4, (t, ctl.rows_moved, changed)
        if t == 5:
            assert ctl.rows_moved > 0
    same = ctl._regroup(option)                                            # nothing changed: nothing moves
    assert ctl.rows_moved == 0 and same == counts
    st = ctl.lstm_state_of([0, 17, n - 1])
    assert st[0].shape == (1, 3, 256) and st[0][0, :, 0].tolist() == [0.0, 17.0, float(n - 1)]
    ctl.close()
