"""GPU parity tests: the HIP step (through the C ABI in libhlx.so) against
  (1) the golden fixtures captured from the reference, and
  (2) the CPU oracle on seeded random batches, free-running over hundreds of steps.

Tolerance (BASELINE.json north_star): 1e-5 relative fp32 on next-state, observation, reward; flags
identical.  Observations are O(1) quantities normalised by the reference itself, so their bound is
absolute; state / reward / distance bounds are relative to max(1, |x|).
"""
import ctypes as C

import numpy as np
import pytest

from tests.golden_util import compare_radar_debug, fixture_names, load_fixture, load_radar_fixture, radar_fixture_names

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # the bar (BASELINE.json north_star: 1e-5 relative fp32), for state, reward, distance AND observation:
OBS_ATOL = 1e-5      # an observation entry x is compared as |x - ref| <= 1e-5 * max(1, |ref|)   (|obs| <= 2)


def _obs_err(a, b):
    """[n, 26] observation error in units of the bar: |a - b| / max(1, |b|)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


_WORST = {}


def _report(test, **worst):
    """Worst observed value per test (shown with `pytest -rP` / `-s`; also kept for the summary test at the end)."""
    _WORST[test] = worst
    print("worst observed:", test, {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in worst.items()})
# Per-step reward = 0.8 x (prev_distance - distance): the difference of two ~km float32 numbers, so ONE ulp
# of a position (2.4e-4 m at 4 km) is a 1e-4 relative change of the reward.  The kernel therefore restates the
# integrator operation by operation (hlx_device.h).  Where the reference's arithmetic is all +,-,*,/,sqrt
# (base physics) that makes state, distance and reward come out bit-identical and the bar applies to every
# step of a free-running trajectory.
# With ISA atmosphere / boundary-layer wind the reference calls the host libm's float32 pow(); the kernel restates
# glibc's algorithm (hlx_device.h pow_ref, pinned by tests/test_ref_math.py), so those configurations meet the same bar.


def _check_reward_errors(errs, rc, what, resynced=False):
    errs = np.asarray(errs, np.float64)
    if errs.size == 0:
        return
    assert errs.max() <= RTOL, (what, errs.max(), int((errs > RTOL).sum()), errs.size)


def _obs_tolerance(rc, ora, done):
    """Per-entry tolerance [n, 26]: the bar, everywhere.  (Round 2 widened three `los_frame` entries -- the LOS rates obs[2:4] and
    the lead-angle cosine obs[5] -- to 5e-4 where the filtered range is under 100 m or the velocity estimate under 2 m/s: the
    fast float32 formulas lose digits there.  The kernel now replays the reference's own arithmetic in that regime, in the
    reference's dtype (hlx_device.h los_exact32 / los_exact64), and the Kalman covariance follows OpenBLAS's sgemm order, so
    the exception is gone.)"""
    return np.full((ora.n, 26), OBS_ATOL)


def _oracle_to_gpu_state(ora, env):
    """Copy the oracle's full per-env state (incl. Kalman filter and delay rings) into the GPU arena."""
    st = env.get_state()
    for i in range(env.num_envs):
        o, g = ora.state[i], st[i]
        for f in ("int_pos", "int_vel", "int_quat", "thrust_actual", "mis_pos", "mis_vel", "wind", "kf_x"):
            src, dst = getattr(o, f), getattr(g, f)
            for k in range(len(dst)):
                dst[k] = src[k]
        g.fuel, g.prev_distance, g.min_distance, g.last_distance = o.fuel, o.prev_distance, o.min_distance, o.last_distance
        g.fuel_used = o.total_fuel_used                    # environment.py:886 (float32-valued in the oracle's double)
        g.steps, g.worsening, g.crossed, g.kf_init, g.kf_x_is64 = o.steps, o.worsening, o.crossed, o.kf_init, o.kf_x_is64
        g.kf_P[0], g.kf_P[1], g.kf_P[2], g.kf_P[3] = o.kf_P[0], o.kf_P[3], o.kf_P[18], o.kf_P[21]
        g.on_delay, g.on_len = o.on_delay, o.on_len
        for k in range(o.on_len):
            g.on_ring[k][0], g.on_ring[k][1], g.on_ring[k][2], g.on_ring[k][3] = o.on_ring[k][0], o.on_ring[k][1], o.on_ring[k][2], float(o.on_det[k])
        g.g_len = o.g_len
        for k in range(o.g_len):
            r = o.g_ring[k]
            g.g_ring[k][0], g.g_ring[k][1], g.g_ring[k][2], g.g_ring[k][3] = r[0], r[1], r[2], r[6]
            g.g_ring[k][4], g.g_ring[k][5], g.g_ring[k][6], g.g_ring[k][7] = r[3], r[4], r[5], float(o.g_pos_is64[k])
        # the oracle holds the reference's Python floats; the kernel holds the float32 roundings each use applies
        g.T0, g.base_cd, g.transonic_peak_m1, g.cd_super = o.T0, o.base_cd, o.transonic_peak - 1.0, o.base_cd * ora.rc.supersonic_multiplier
        for k in range(4):   # volley: every missile, its activity and minimum distance, the priority index
            for j in range(3):
                g.v_pos[k][j], g.v_vel[k][j] = o.v_pos[k][j], o.v_vel[k][j]
            g.v_active[k], g.v_min[k] = o.v_active[k], o.v_min[k]
        g.prio, g.n_intercepted = o.prio, o.n_intercepted
    env.set_state(st)


def _torch():
    import torch
    return torch


def _make_env(rc, n, global_step=None, seed=0, offset=0, radar_debug=False):
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    env = HlynrVecEnv(resolved=rc, num_envs=n, seed=seed, env_id_offset=offset, radar_debug=radar_debug)
    if global_step is not None:
        env.set_training_step_count(global_step)
    return env


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


# ----------------------------------------------------------------------------------------------
# (1) golden fixtures replayed on the GPU
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", fixture_names())
def test_gpu_matches_reference_fixture(name):
    _replay_fixture(name, None)


@pytest.mark.parametrize("name", radar_fixture_names())
def test_gpu_radar_debug_matches_reference(name):
    """Same replay with HLX_F_RADAR_DEBUG (generic kernel variant + debug planes): everything above still holds and
    info['radar_debug'] of every step equals the reference's recorded dict (tests/golden/radar)."""
    _replay_fixture(name, load_radar_fixture(name))


_VARIANTS_REPLAYED = {}


def test_every_kernel_variant_replays_a_reference_fixture():
    """Runs after the fixture replays above: each of the eight kernel variants hlx_create can select (hlx_host.inc) must
    have been exercised by at least one trajectory recorded from the reference."""
    if not _VARIANTS_REPLAYED:
        pytest.skip("fixture replays were not part of this run")
    want = {"base", "v2", "v2dr", "config", "config-easy", "config-volley", "generic", "generic-volley"}
    if sum(len(v) for v in _VARIANTS_REPLAYED.values()) < len(fixture_names()):
        pytest.skip("only a subset of the fixture replays ran")
    assert want <= set(_VARIANTS_REPLAYED), want - set(_VARIANTS_REPLAYED)


def test_radar_planes_need_the_flag():
    from hlynr_intercept_amd import _lib as hl
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config
    torch = _torch()
    env = _make_env(resolve_config(scenario_config("medium", "base")), 4)
    env.reset_torch()
    env.info["radar_debug"] = torch.zeros((8, 4), device=env.device)
    env._info_soa.radar_debug = env.info["radar_debug"].data_ptr()
    with pytest.raises(hl.HlxError, match="HLX_F_RADAR_DEBUG"):
        env.step_torch(torch.zeros((4, 6), device=env.device))
    env.close()


def _replay_fixture(name, radar):
    torch = _torch()
    from hlynr_intercept_amd.config import resolve_config

    fx = load_fixture(name)
    rc = resolve_config(fx["config"])
    n = 3   # same inputs in three lanes: also checks lane independence
    env = _make_env(rc, n, fx["global_step_or_none"], radar_debug=radar is not None)
    if radar is None:
        _VARIANTS_REPLAYED.setdefault(env.kernel_variant, []).append(name)
    dev = env.device
    T = len(fx["action"])
    radar_bad = []
    from hlynr_intercept_amd import _lib as hl
    S, R = hl.STEP_SLOTS, hl.RESET_SLOTS

    def slots(a, width):   # fixture slot arrays (20/32 wide, or 32/48 for volley cases) -> the ABI's width, NaN = unused
        a = np.nan_to_num(np.asarray(a, np.float64), nan=0.5)
        out = np.full(a.shape[:-1] + (width,), 0.5)
        out[..., :a.shape[-1]] = a
        return out

    sn_all = torch.tensor(slots(fx["step_noise"], S), dtype=torch.float64, device=dev)       # [T, S]
    sn_all = sn_all[:, :, None].expand(T, S, n).contiguous()
    rn0 = torch.tensor(slots(fx["reset_noise0"], R), dtype=torch.float64, device=dev)[:, None].expand(R, n).contiguous()
    if "reset_noise" in fx:
        rn_all = torch.tensor(slots(fx["reset_noise"], R), dtype=torch.float64, device=dev)
        rn_all = rn_all[:, :, None].expand(rn_all.shape[0], R, n).contiguous()
    actions = torch.tensor(fx["action"], dtype=torch.float32, device=dev)[:, None, :].expand(T, n, 6).contiguous()

    env.set_noise(sn_all[0], rn0)
    obs0 = env.reset_torch().cpu().numpy()
    assert np.max(_obs_err(obs0, fx["reset_obs0"][None])) <= OBS_ATOL
    st = env.get_state()
    # reset()'s info (environment.py:595-601) in the words hlx_reset_info wrote: the spawn the state holds, nothing used, no flag
    # but the first observation's detections -- and no onboard detection while a delay line is still filling (core.py:576-583)
    for i in range(n):
        assert np.array_equal(env.info["interceptor_pos"][:, i].cpu().numpy(), np.array(st[i].int_pos[:], np.float32))
        assert np.array_equal(env.info["missile_pos"][:, i].cpu().numpy(), np.array(st[i].mis_pos[:], np.float32))
        assert float(env.info["distance"][i]) == float(st[i].prev_distance) == float(env.info["min_distance"][i])
        assert float(env.info["fuel"][i]) == float(st[i].fuel) and float(env.info["fuel_used"][i]) == 0.0 and int(env.info["steps"][i]) == 0
        fl = int(env.info["flags"][i])
        assert fl & 0x1F == 0 and bool(fl & 128) == (int(st[i].on_delay) == 0), (fl, int(st[i].on_delay))
        if int(st[i].on_delay) > 0:
            assert not fl & 32
        K = int(rc.volley_size) if rc.volley_mode else 1
        assert int(env.info["missiles"][i]) == K << 4
    # forced edge cases: overwrite the kinematic state the generator tweaked after its first reset
    for i in range(n):
        for fld in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel"):
            arr = getattr(st[i], fld)
            for k, x in enumerate(fx["init_" + fld]):
                arr[k] = float(x)
        st[i].fuel = float(fx["init_fuel"]); st[i].steps = int(fx["init_steps"])
        st[i].prev_distance = float(fx["init_prev_distance"]); st[i].min_distance = float(fx["init_min_distance"])
        st[i].last_distance = float(fx["init_last_distance"]); st[i].worsening = int(fx["init_worsening"])
        st[i].crossed = int(fx["init_crossed"])
        if "init_v_pos" in fx:   # volley fixtures: every missile (the tweaks move them individually)
            for k in range(fx["init_v_pos"].shape[0]):
                for j in range(3):
                    st[i].v_pos[k][j] = float(fx["init_v_pos"][k][j]); st[i].v_vel[k][j] = float(fx["init_v_vel"][k][j])
                st[i].v_active[k] = int(fx["init_v_active"][k]); st[i].v_min[k] = float(fx["init_v_min"][k])
            st[i].prio = int(fx["init_prio"])
    env.set_state(st)

    k_reset = 0
    st_at = {int(t): j for j, t in enumerate(fx["st_index"])}
    worst = dict(obs=0.0, distance=0.0, reset_obs=0.0)
    rew_errs = []
    for t in range(T):
        rn = rn_all[k_reset] if fx["did_reset"][t] else rn0
        env.set_noise(sn_all[t], rn)
        obs, rew, term, trunc, info = env.step_torch(actions[t])
        obs_h, rew_h = obs.cpu().numpy(), rew.cpu().numpy()
        term_h, trunc_h = term.cpu().numpy(), trunc.cpu().numpy()
        flags = info["flags"].cpu().numpy()
        assert np.all(term_h == int(fx["terminated"][t])), (t, term_h, fx["terminated"][t])
        assert np.all(trunc_h == int(fx["truncated"][t])), (t, trunc_h)
        assert np.all((flags & 1) == int(fx["intercepted"][t])), (t, flags)
        assert np.all(((flags >> 1) & 1) == int(fx["hit_target"][t])), (t, flags)
        if "missiles_intercepted" in fx:   # volley: info['missiles_intercepted'] / ['missiles_remaining']
            m = info["missiles"].cpu().numpy()
            assert np.all((m & 15) == int(fx["missiles_intercepted"][t])) and np.all((m >> 4) == int(fx["missiles_remaining"][t])), (t, m)
            K = int(fx["volley_size"])      # info['missile_min_distances'] (environment.py:848)
            md = info["missile_min_distances"].cpu().numpy()[:K, 0]
            assert np.max(_rel(md, fx["missile_min_distances"][t])) <= 2 * RTOL, (t, md, fx["missile_min_distances"][t])
        step_obs = info["terminal_observation"].cpu().numpy() if fx["did_reset"][t] else obs_h
        worst["obs"] = max(worst["obs"], float(np.max(_obs_err(step_obs, fx["obs"][t][None]))))
        if "fuel_used" in info:    # info['fuel_used'] = total_fuel_used (environment.py:834, 886): the float32 running sum, bit for bit
            fu = info["fuel_used"].cpu().numpy()
            assert np.all(fu == np.float32(fx["fuel_used"][t])), (t, fu, fx["fuel_used"][t])
        rew_errs.append(float(np.max(_rel(rew_h, fx["reward"][t]))))
        worst["distance"] = max(worst["distance"], float(np.max(_rel(info["distance"].cpu().numpy(), fx["distance"][t]))))
        assert np.all(obs_h == obs_h[0:1]) and np.all(rew_h == rew_h[0]), "lanes with identical inputs diverged"
        if radar is not None:   # environment.py:842 <- core.py:650-683, rebuilt from the kernel's side outputs
            from hlynr_intercept_amd.episode_log import radar_debug
            p = info["radar_debug"].cpu().numpy()
            mine = radar_debug(rc, env.curriculum()["beam_width"], info["interceptor_pos"].cpu().numpy()[:, 0],
                               info["missile_pos"].cpu().numpy()[:, 0], p[0:4, 0], float(p[4, 0]),
                               int(p[5, 0:1].view(np.int32)[0]), int(flags[0]), float(p[6, 0]), float(p[7, 0]))
            radar_bad += compare_radar_debug(mine, radar, t)
        if t in st_at:   # info['interceptor_pos' | 'missile_pos' | 'steps'] = the post-step state of the (possibly finished) episode
            j = st_at[t]
            assert np.max(_rel(info["interceptor_pos"].cpu().numpy()[:, 0], fx["st_int_pos"][j])) <= 2 * RTOL, t
            assert np.max(_rel(info["missile_pos"].cpu().numpy()[:, 0], fx["st_mis_pos"][j])) <= 2 * RTOL, t
            assert int(info["steps"][0]) == int(fx["st_steps"][j]), t
            # info['crossed_threshold'] (environment.py:851) as the step leaves it -- also on the step that ends the episode
            # (until round 3 a finished environment reported the flag of its NEXT episode, i.e. always False)
            assert np.all(((flags >> 4) & 1) == int(fx["st_crossed"][j])), (t, flags, int(fx["st_crossed"][j]))
            assert np.max(_rel(info["min_distance"].cpu().numpy()[0], fx["st_min_distance"][j])) <= RTOL, t
        if fx["did_reset"][t]:
            worst["reset_obs"] = max(worst["reset_obs"], float(np.max(_obs_err(obs_h, fx["reset_obs"][k_reset][None]))))
            k_reset += 1
    _report(f"fixture {name}" + (" (radar)" if radar is not None else ""), reward=max(rew_errs, default=0.0), **worst)
    assert worst["obs"] <= OBS_ATOL, worst
    assert worst["reset_obs"] <= OBS_ATOL, worst
    assert not radar_bad, (len(radar_bad), radar_bad[:5])
    _check_reward_errors(rew_errs, rc, name)
    assert worst["distance"] <= RTOL, worst
    # final state vs the recorded reference state
    st = env.get_state()[0]
    last = -1
    for fld, key in (("int_pos", "st_int_pos"), ("int_vel", "st_int_vel"), ("int_quat", "st_int_quat"),
                     ("mis_pos", "st_mis_pos"), ("mis_vel", "st_mis_vel"), ("wind", "st_wind")):
        ref = fx[key][last] if not fx["did_reset"][T - 1] else fx["rst_" + fld][-1]
        assert np.max(_rel(np.array(getattr(st, fld)[:]), ref)) <= 2 * RTOL, (fld, np.array(getattr(st, fld)[:]), ref)
    if "st_v_pos" in fx:
        src = "st_" if not fx["did_reset"][T - 1] else "rst_"
        K = fx[src + "v_pos"].shape[1]
        assert np.max(_rel(np.array([list(st.v_pos[k]) for k in range(K)]), fx[src + "v_pos"][-1])) <= 2 * RTOL
        assert np.max(_rel(np.array([list(st.v_vel[k]) for k in range(K)]), fx[src + "v_vel"][-1])) <= 2 * RTOL
        assert np.max(_rel(np.array([st.v_min[k] for k in range(K)]), fx[src + "v_min"][-1])) <= 2 * RTOL
        assert [int(st.v_active[k]) for k in range(K)] == [int(x) for x in fx[src + "v_active"][-1]]
        assert st.prio == int(fx[src + "prio"][-1])
    env.close()


# ----------------------------------------------------------------------------------------------
# (2) GPU vs oracle, seeded random batch, free-running, Philox draws exported to the oracle
# ----------------------------------------------------------------------------------------------
# (scenario, physics preset, overrides, kernel variant hlx_create must select): every shipped variant is covered, and the
# test ids carry the variant name
CASES = [
    ("medium", "base", {}, "base"),
    ("medium", "v2dr", {}, "v2dr"),
    ("hard", "v2", {"max_steps": 150}, "v2"),
    ("easy", "config", {"observation_mode": "body_frame", "max_steps": 120}, "generic"),
    ("medium", "v2", {"observation_mode": "los_frame", "proximity_fuze_enabled": True, "proximity_kill_radius": 60.0,
                      "max_steps": 200}, "generic"),
    ("medium", "base", {"curriculum.precision_mode": True, "max_steps": 100}, "generic"),
    # volley mode: K missiles per episode (environment.py:236-267, 631-692, 724-748)
    ("medium", "base", {"volley_mode": True, "volley_size": 3, "max_steps": 200}, "generic-volley"),
    ("medium", "v2dr", {"volley_mode": True, "volley_size": 4, "proximity_fuze_enabled": True, "proximity_kill_radius": 80.0,
                        "max_steps": 150}, "generic-volley"),
    # the reference's own config.yaml physics (ISA atmosphere only): what train_flat_ppo.py / inference.py run out of the box
    ("medium", "config", {}, "config"),
    ("hard", "config", {"max_steps": 180}, "config"),
    ("easy", "config", {"max_steps": 150}, "config-easy"),
    ("medium", "config", {"volley_mode": True, "volley_size": 3, "max_steps": 200}, "config-volley"),
]
CASE_IDS = [f"{v}-{s}-{p}-{i}" for i, (s, p, o, v) in enumerate(CASES)]


@pytest.mark.parametrize("late", [2, 1, 0], ids=["lone-wave", "late-loads", "entry-loads"])
@pytest.mark.parametrize("scenario,physics,over,variant", CASES, ids=CASE_IDS)
def test_gpu_matches_oracle_free_running(scenario, physics, over, variant, late):
    torch = _torch()
    import oracle.oracle as orc
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    rc = resolve_config(scenario_config(scenario, physics, over))
    n, T = 1024, 260
    env = _make_env(rc, n, seed=1234)
    assert env.kernel_variant == variant, (env.kernel_variant, variant)
    env.set_load_schedule(late)      # both load schedules of the production kernel (the large-batch one is the 4 M-env roofline point)
    ora = orc.OracleVec(rc, n)
    g = torch.Generator(device="cpu").manual_seed(7)
    sn, rn = env.fill_noise(for_reset=True)
    # reset both from the same Philox draws
    obs_g = env.reset_torch().cpu().numpy()
    obs_o = ora.reset(rn.cpu().numpy().T.copy())
    assert np.max(_obs_err(obs_g, obs_o)) <= OBS_ATOL
    alive = np.ones(n, bool)     # envs whose discrete history still agrees
    n_done_total = 0
    worst = dict(obs=0.0, distance=0.0)
    rew_errs = []
    for t in range(T):
        a = (torch.rand((n, 6), generator=g) * 2 - 1)
        if t % 3 == 0:   # mix in pursuit-like thrust so that intercepts happen
            a[:, 2] = 0.9
        sn, rn = env.fill_noise()
        obs, rew, term, trunc, info = env.step_torch(a.to(env.device))
        out = ora.step(a.numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
        term_h, trunc_h = term.cpu().numpy(), trunc.cpu().numpy()
        same = (term_h == out["terminated"]) & (trunc_h == out["truncated"]) & \
               ((info["flags"].cpu().numpy() & 1) == out["intercepted"])
        alive &= same
        done = (term_h | trunc_h).astype(bool)
        n_done_total += int(done.sum())
        obs_h = obs.cpu().numpy()
        term_obs = info["terminal_observation"].cpu().numpy()
        step_obs_g = np.where(done[:, None], term_obs, obs_h)
        step_obs_o = np.where(done[:, None], ora.terminal_obs, out["obs"])
        eo_all = _obs_err(step_obs_g, step_obs_o) * (OBS_ATOL / _obs_tolerance(rc, ora, done))
        eo = np.max(eo_all, axis=1)
        if eo.max() > worst.get("obs_max_any", 0.0):
            i_w, k_w = np.unravel_index(np.argmax(eo_all), eo_all.shape)
            worst["obs_max_any"] = float(eo.max())
            worst["obs_worst_at"] = f"t={t} env={i_w} entry={k_w} gpu={step_obs_g[i_w, k_w]!r} oracle={step_obs_o[i_w, k_w]!r} done={bool(done[i_w])}"
        er = _rel(rew.cpu().numpy(), out["reward"])
        ed = _rel(info["distance"].cpu().numpy(), out["distance"])
        # an env whose observation jumps (a detection decided differently at a float32 boundary) is retired
        alive &= eo <= 50 * OBS_ATOL
        worst["obs"] = max(worst["obs"], float(eo[alive].max(initial=0.0)))
        rew_errs.append(er[alive])
        worst["distance"] = max(worst["distance"], float(ed[alive].max(initial=0.0)))
        # reset observations of finished envs
        if done.any():
            sel = done & alive
            if sel.any():
                assert np.max(_obs_err(obs_h[sel], out["obs"][sel])) <= OBS_ATOL
        fu_bad = info["fuel_used"].cpu().numpy() != out["fuel_used"]
        assert not fu_bad[alive].any(), (t, int(fu_bad.sum()))       # info['fuel_used']: bit for bit
    # detection decisions are taken with the reference's own arithmetic near their thresholds (hlx_kernels.hip): no environment
    # may leave the oracle's discrete history -- round 1 tolerated 0.5 % of them
    assert alive.all(), f"{(~alive).sum()} of {n} envs diverged in their discrete history"
    assert worst["obs"] <= OBS_ATOL, worst
    assert worst["distance"] <= RTOL, worst
    _check_reward_errors(np.concatenate(rew_errs), rc, (scenario, physics))
    # full state comparison at the end (alive envs)
    st = env.get_state()
    for name in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind"):
        ref = ora.field(name)
        mine = np.array([list(getattr(st[i], name)) for i in range(n)])
        assert np.max(_rel(mine[alive], ref[alive])) <= 2 * RTOL, name
    # the Kalman filter: observed bit-identical covariance and <= 3e-5 m of position estimate (tools/soak_oracle.py); the bounds
    # are 1e-6 relative, a tenth of the bar (round 2 allowed 5e-5 / 1e-3: a hundredfold regression would have passed)
    kx = np.array([list(st[i].kf_x) for i in range(n)])
    kx_err = float(np.max(_rel(kx[alive][:, :3], ora.field("kf_x")[alive][:, :3])))
    P = np.array([list(st[i].kf_P) for i in range(n)])
    Po = ora.field("kf_P").reshape(n, 6, 6)
    Pref = np.stack([Po[:, 0, 0], Po[:, 0, 3], Po[:, 3, 0], Po[:, 3, 3]], axis=1)
    kp_err = float(np.max(np.abs(P[alive] - Pref[alive]) / np.maximum(1e-2, np.abs(Pref[alive]))))
    _report(f"free-running {variant} {scenario}/{physics} late={late}", reward=float(np.concatenate(rew_errs).max(initial=0.0)),
            kf_x_pos=kx_err, kf_P=kp_err, **worst)
    assert kx_err <= 1e-6, kx_err
    assert kp_err <= 1e-6, kp_err
    steps_g = np.array([st[i].steps for i in range(n)])
    assert np.array_equal(steps_g[alive], ora.field("steps")[alive])
    if rc.volley_mode:
        K = rc.volley_size
        vp = np.array([[list(st[i].v_pos[k]) for k in range(K)] for i in range(n)])
        assert np.max(_rel(vp[alive], ora.field("v_pos")[alive][:, :K])) <= 2 * RTOL
        va = np.array([[int(st[i].v_active[k]) for k in range(K)] for i in range(n)])
        assert np.array_equal(va[alive], ora.field("v_active")[alive][:, :K])
        assert np.array_equal(np.array([st[i].prio for i in range(n)])[alive], ora.field("prio")[alive])
    assert n_done_total > 0 or rc.max_steps > T, "case never exercised auto-reset"
    env.close()


_RESYNC = [1, 2, 3, 7, 8, 11]
@pytest.mark.parametrize("scenario,physics,over,variant", [CASES[i] for i in _RESYNC], ids=[CASE_IDS[i] for i in _RESYNC])
def test_gpu_matches_oracle_from_identical_state(scenario, physics, over, variant):
    """Single-step parity: before every step the GPU arena is overwritten with the oracle's state, so
    differences cannot accumulate.  Exercises set_state/get_state with rings and Kalman state as well."""
    torch = _torch()
    import oracle.oracle as orc
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    rc = resolve_config(scenario_config(scenario, physics, over))
    n, T = 256, 120
    env = _make_env(rc, n, seed=77)
    assert env.kernel_variant == variant
    ora = orc.OracleVec(rc, n)
    g = torch.Generator(device="cpu").manual_seed(5)
    sn, rn = env.fill_noise(for_reset=True)
    env.reset_torch()
    ora.reset(rn.cpu().numpy().T.copy())
    rew_errs, worst_obs, worst_dist, flag_mismatch = [], 0.0, 0.0, 0
    for t in range(T):
        _oracle_to_gpu_state(ora, env)
        a = (torch.rand((n, 6), generator=g) * 2 - 1)
        sn, rn = env.fill_noise()
        obs, rew, term, trunc, info = env.step_torch(a.to(env.device))
        out = ora.step(a.numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
        term_h, trunc_h = term.cpu().numpy(), trunc.cpu().numpy()
        done = (term_h | trunc_h).astype(bool)
        same = (term_h == out["terminated"]) & (trunc_h == out["truncated"])
        flag_mismatch += int((~same).sum())
        step_obs_g = np.where(done[:, None], info["terminal_observation"].cpu().numpy(), obs.cpu().numpy())
        step_obs_o = np.where(done[:, None], ora.terminal_obs, out["obs"])
        eo = np.max(_obs_err(step_obs_g, step_obs_o) * (OBS_ATOL / _obs_tolerance(rc, ora, done)), axis=1)
        ok = same & (eo <= 50 * OBS_ATOL)          # a Bernoulli detection decided at a float32 boundary is retired
        flag_mismatch += int((same & ~ok).sum())
        worst_obs = max(worst_obs, float(eo[ok].max(initial=0.0)))
        worst_dist = max(worst_dist, float(_rel(info["distance"].cpu().numpy(), out["distance"])[ok].max(initial=0.0)))
        rew_errs.append(_rel(rew.cpu().numpy(), out["reward"])[ok])
    _report(f"resynced {variant} {scenario}/{physics}", obs=worst_obs, distance=worst_dist, reward=float(np.concatenate(rew_errs).max(initial=0.0)))
    assert flag_mismatch == 0, flag_mismatch
    assert worst_obs <= OBS_ATOL and worst_dist <= RTOL, (worst_obs, worst_dist)
    _check_reward_errors(np.concatenate(rew_errs), rc, (scenario, physics), resynced=True)
    env.close()


@pytest.mark.parametrize("physics,variant", [("base", "base"), ("v2dr", "v2dr")])
def test_full_size_batch_matches_oracle(physics, variant):
    """BASELINE.json configs 2 and 3 at their full size -- 65 536 environments on one GPU -- against the oracle, free-running
    for 50 steps from the same Philox draws (max_steps 40: every environment is auto-reset inside the window), with the load
    schedule hlx_create picks for this batch size."""
    torch = _torch()
    import oracle.oracle as orc
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    rc = resolve_config(scenario_config("medium", physics, {"max_steps": 40}))
    n, T = 65536, 50
    env = _make_env(rc, n, seed=2024)
    assert env.kernel_variant == variant
    ora = orc.OracleVec(rc, n)
    g = torch.Generator(device=env.device).manual_seed(17)
    sn, rn = env.fill_noise(for_reset=True)
    obs_g = env.reset_torch().cpu().numpy()
    obs_o = ora.reset(rn.cpu().numpy().T.copy())
    assert np.max(_obs_err(obs_g, obs_o)) <= OBS_ATOL
    alive = np.ones(n, bool)
    worst = dict(obs=0.0, distance=0.0, reset_obs=0.0)
    rew_errs, n_done = [], 0
    for t in range(T):
        a = torch.rand((n, 6), generator=g, device=env.device) * 2 - 1
        sn, rn = env.fill_noise()
        obs, rew, term, trunc, info = env.step_torch(a)
        out = ora.step(a.cpu().numpy(), sn.cpu().numpy().T.copy(), rn.cpu().numpy().T.copy())
        term_h, trunc_h = term.cpu().numpy(), trunc.cpu().numpy()
        alive &= (term_h == out["terminated"]) & (trunc_h == out["truncated"]) & ((info["flags"].cpu().numpy() & 1) == out["intercepted"])
        done = (term_h | trunc_h).astype(bool)
        n_done += int(done.sum())
        obs_h = obs.cpu().numpy()
        step_obs_g = np.where(done[:, None], info["terminal_observation"].cpu().numpy(), obs_h)
        step_obs_o = np.where(done[:, None], ora.terminal_obs, out["obs"])
        eo = np.max(_obs_err(step_obs_g, step_obs_o) * (OBS_ATOL / _obs_tolerance(rc, ora, done)), axis=1)
        alive &= eo <= 50 * OBS_ATOL             # a Bernoulli detection decided at a float32 boundary retires the env
        worst["obs"] = max(worst["obs"], float(eo[alive].max(initial=0.0)))
        rew_errs.append(_rel(rew.cpu().numpy(), out["reward"])[alive])
        worst["distance"] = max(worst["distance"], float(_rel(info["distance"].cpu().numpy(), out["distance"])[alive].max(initial=0.0)))
        sel = done & alive
        if sel.any():
            worst["reset_obs"] = max(worst["reset_obs"], float(np.max(_obs_err(obs_h[sel], out["obs"][sel]))))
    _report(f"full size {physics}", reward=float(np.concatenate(rew_errs).max(initial=0.0)), **worst)
    assert n_done >= n, "every environment must have restarted inside the window"
    assert alive.all(), f"{(~alive).sum()} of {n} envs diverged in their discrete history"
    assert worst["obs"] <= OBS_ATOL and worst["reset_obs"] <= OBS_ATOL, worst
    assert worst["distance"] <= RTOL, worst
    _check_reward_errors(np.concatenate(rew_errs), rc, ("full size", physics))
    st = np.frombuffer(env.get_state(), dtype=np.dtype(type(env.get_state()[0])))
    so = np.frombuffer(ora.state, dtype=np.dtype(type(ora.state[0])))
    for name in ("int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind"):
        assert np.max(_rel(st[name][alive], so[name][alive])) <= 2 * RTOL, name
    assert np.array_equal(st["steps"][alive], so["steps"][alive])
    env.close()


@pytest.mark.parametrize("volley", [False, True])
def test_philox_path_equals_noise_buffer_path(volley):
    """The in-kernel Philox draws and the same draws fed through the noise buffers give identical bits."""
    torch = _torch()
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    rc = resolve_config(scenario_config("medium", "v2dr", {"max_steps": 40, "volley_mode": volley, "volley_size": 3}))
    n = 512
    a_env, b_env = _make_env(rc, n, seed=99), _make_env(rc, n, seed=99)
    sn, rn = b_env.fill_noise(for_reset=True)
    b_env.set_noise(sn, rn)
    oa, ob = a_env.reset_torch().clone(), b_env.reset_torch().clone()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cpu").manual_seed(3)
    for t in range(100):
        act = (torch.rand((n, 6), generator=g) * 2 - 1).to(a_env.device)
        sn, rn = b_env.fill_noise()
        b_env.set_noise(sn, rn)
        ra = a_env.step_torch(act)
        rb = b_env.step_torch(act)
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y), t
    a_env.close(); b_env.close()


def test_sharding_is_concatenation():
    """Env i's trajectory depends on (seed, global env id), not on which shard / GPU holds it."""
    torch = _torch()
    from hlynr_intercept_amd.config import resolve_config
    from hlynr_intercept_amd.scenarios import scenario_config

    rc = resolve_config(scenario_config("medium", "base", {"max_steps": 50}))
    n = 640
    whole = _make_env(rc, n, seed=5)
    lo, hi = _make_env(rc, 256, seed=5, offset=0), _make_env(rc, n - 256, seed=5, offset=256)
    o = whole.reset_torch().clone()
    assert torch.equal(o, torch.cat([lo.reset_torch(), hi.reset_torch()]))
    g = torch.Generator(device="cpu").manual_seed(11)
    for t in range(120):
        act = (torch.rand((n, 6), generator=g) * 2 - 1).to(whole.device)
        rw = whole.step_torch(act)
        rl, rh = lo.step_torch(act[:256].contiguous()), hi.step_torch(act[256:].contiguous())
        for k in range(4):
            assert torch.equal(rw[k], torch.cat([rl[k], rh[k]])), (t, k)
    whole.close(); lo.close(); hi.close()
