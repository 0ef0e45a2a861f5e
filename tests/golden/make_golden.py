#!/usr/bin/env python3
"""Golden-vector generator for the intercept-environment hot path.

Runs ONLY in the build container (needs /root/reference); its outputs
(tests/golden/*.npz) are plain data and are what travels to the GPU box.

What it does
------------
* registers an in-memory `gymnasium` stand-in (only `Env` and `spaces.Box` are touched by
  the reference: rl_system/environment.py:6-7,15,192-197,355), imports the reference's
  `InterceptEnvironment` read-only (no bytecode is written),
* wraps the three random streams of the path with recorders so every variate the reference
  consumes is captured as a *unit* draw (U(0,1), N(0,1), Exp(1)) in a fixed slot layout
  (SURVEY.md §8 a21); the wrappers return bit-identical values to the unwrapped calls
  (checked by `_selfcheck_wrappers`),
* drives a list of cases (scenario x physics x observation mode x reward mode + forced edge
  cases), with VecEnv-style auto-reset, and dumps per step: action, noise slots, post-step
  state (incl. Kalman state/covariance and delay-ring bookkeeping), obs[26], reward, flags.

Slot layout (kept in sync with include/hlx.h HLX_SLOT_*):
  step : 0-2 evasion N | 3-5 wind N | 6 gust U | 7-9 gust dir N | 10 gust mag Exp
         11 onboard U | 12 ground U | 13-15 ground pos N | 16-18 ground vel N | 19 datalink U
  reset: 0-2 missile pos U (box) or radius/azimuth/elevation U (spherical) | 3 missile speed U
         4-6 interceptor pos U | 7-9 interceptor vel U (box) or 7 speed U (toward_missile)
         10 onboard U | 11 ground U | 12-14 ground pos N | 15-17 ground vel N | 18 datalink U
         19-31 domain-randomisation N x13
Unused slots hold NaN.
"""
import sys
import types
import json
import copy
import os

sys.dont_write_bytecode = True
import numpy as np
import yaml

REF = "/root/reference/rl_system"
OUT = os.path.dirname(os.path.abspath(__file__))

N_STEP_SLOTS = 20
N_RESET_SLOTS = 32
# volley fixtures (K <= 4 missiles) use the extended layout: missile k >= 1 draws its evasion normals into step slots
# 20+3(k-1) .. and its spawn uniforms (position x3, speed) into reset slots 32+4(k-1) ..
V_STEP_SLOTS = 32
V_RESET_SLOTS = 48


# --------------------------------------------------------------------------------------
# gymnasium stand-in (in memory only)
# --------------------------------------------------------------------------------------
def _install_gym_shim():
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:
        metadata = {}

        def reset(self, seed=None, options=None):
            return None

    class Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    gym.Env = Env
    spaces.Box = Box
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces


# --------------------------------------------------------------------------------------
# RNG recorders
# --------------------------------------------------------------------------------------
class Tape:
    """Collects the unit draws of the current step / reset into slot arrays."""

    def __init__(self):
        self.mode = "reset"
        self.clear()

    volley = False
    cur_missile = 0

    def clear(self):
        self.step = np.full(V_STEP_SLOTS if self.volley else N_STEP_SLOTS, np.nan)
        self.reset = np.full(V_RESET_SLOTS if self.volley else N_RESET_SLOTS, np.nan)
        self._reset_uniform_calls = 0
        self._ground_normal_calls = 0
        self._wind_normal_calls = 0
        self._dr_calls = 0

    def put(self, where, start, vals):
        arr = self.step if where == "step" else self.reset
        vals = np.atleast_1d(np.asarray(vals, dtype=np.float64))
        assert np.all(np.isnan(arr[start:start + len(vals)])), (where, start, "slot written twice")
        arr[start:start + len(vals)] = vals


TAPE = Tape()
_orig_default_rng = np.random.default_rng
_orig_uniform = np.random.uniform
_orig_randn = np.random.randn


def _caller(depth=2):
    return sys._getframe(depth).f_code.co_name


class RecGen:
    """Proxy for numpy.random.Generator: draws unit variates from the real generator,
    records them and applies loc/scale exactly as numpy's C code does (loc + scale * z)."""

    def __init__(self, gen):
        self._g = gen

    def random(self):
        u = self._g.random()
        who = _caller()
        if who == "compute_radar_detection":
            TAPE.put(*(("step", 11) if TAPE.mode == "step" else ("reset", 10)), u)
        elif who == "_compute_ground_radar_detection":
            TAPE.put(*(("step", 12) if TAPE.mode == "step" else ("reset", 11)), u)
        elif who == "_compute_datalink_quality":
            TAPE.put(*(("step", 19) if TAPE.mode == "step" else ("reset", 18)), u)
        elif who == "get_wind_vector":
            TAPE.put("step", 6, u)
        elif who == "randomize_for_episode":
            pass  # the 'timestamp' draw; has no effect on the path
        else:
            raise RuntimeError("unexpected random() caller " + who)
        return u

    def normal(self, loc=0.0, scale=1.0, size=None):
        z = self._g.standard_normal(size)
        who = _caller()
        if who == "_compute_ground_radar_detection":
            k = TAPE._ground_normal_calls
            TAPE._ground_normal_calls += 1
            base = (13 if TAPE.mode == "step" else 12) + 3 * k
            TAPE.put(TAPE.mode, base, z)
        elif who == "get_wind_vector":
            k = TAPE._wind_normal_calls
            TAPE._wind_normal_calls += 1
            TAPE.put("step", 3 if k == 0 else 7, z)
        elif who in ("randomize_for_episode", "_randomize_multiplier"):
            TAPE.put("reset", 19 + TAPE._dr_calls, z)
            TAPE._dr_calls += 1
        else:
            raise RuntimeError("unexpected normal() caller " + who)
        return loc + scale * z

    def exponential(self, scale=1.0):
        e = self._g.standard_exponential()
        assert _caller() == "get_wind_vector"
        TAPE.put("step", 10, e)
        return scale * e


def _rec_default_rng(seed=None):
    return RecGen(_orig_default_rng(seed))


def _rec_uniform(low=0.0, high=1.0, size=None):
    who = _caller()
    assert who == "reset", who
    low_a, high_a = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
    shape = np.broadcast(low_a, high_a).shape
    u = np.random.random_sample(shape if shape else None)
    k = TAPE._reset_uniform_calls
    TAPE._reset_uniform_calls += 1
    TAPE._uniform_log.append((k, np.atleast_1d(u).copy()))
    val = low_a + (high_a - low_a) * u
    # numpy returns a Python float for the scalar call (weak in NEP-50 promotion): keep that
    return float(val) if shape == () else val


def _rec_randn(*shape):
    z = np.random.standard_normal(shape)
    who = _caller()
    if who == "_update_missile_state":
        k = TAPE.cur_missile
        TAPE.put("step", 0 if k == 0 else 20 + 3 * (k - 1), z)
    elif who == "_update_wind":
        TAPE.put("step", 3, z)
    else:
        raise RuntimeError("unexpected randn caller " + who)
    return z


def _selfcheck_wrappers():
    st = np.random.get_state()
    a = _orig_uniform([1, 2, 3], [4, 6, 9])
    b = _orig_uniform(3.0, 7.5)
    c = _orig_randn(3)
    np.random.set_state(st)
    TAPE._uniform_log = []
    frame_reset = lambda: None  # noqa: E731

    def reset():  # name matters: the wrappers assert on their caller's name
        return _rec_uniform([1, 2, 3], [4, 6, 9]), _rec_uniform(3.0, 7.5)

    def _update_missile_state():
        return _rec_randn(3)

    a2, b2 = reset()
    c2 = _update_missile_state()
    assert np.array_equal(a, a2) and b == b2 and np.array_equal(c, c2), "global-stream wrappers not bit-identical"
    g1, g2 = _orig_default_rng(7), RecGen(_orig_default_rng(7))
    x1 = g1.normal(1.0, 0.3)

    def randomize_for_episode():
        return g2.normal(1.0, 0.3)

    TAPE.clear()
    assert x1 == randomize_for_episode(), "Generator.normal wrapper not bit-identical"
    x1 = g1.normal(0, 2.5, 3)

    def get_wind_vector():
        return g2.normal(0, 2.5, 3), g2.exponential(5.0)

    TAPE.clear()
    TAPE.mode = "step"
    y, e = get_wind_vector()
    assert np.array_equal(x1, y) and e == g1.exponential(5.0)
    TAPE.clear()


# --------------------------------------------------------------------------------------
# state capture
# --------------------------------------------------------------------------------------
def ring_dump(buf, width, getter):
    """Logical (oldest -> newest) contents of a SensorDelayBuffer."""
    if buf is None:
        return dict(delay=0, count=0, data=np.zeros((0, width)), det=np.zeros(0, np.int32))
    data = np.zeros((len(buf.measurement_buffer), width))
    for i, m in enumerate(buf.measurement_buffer):
        data[i] = getter(m)
    return dict(delay=buf.delay_samples, count=buf.samples_received, data=data,
                det=np.array(list(buf.detection_buffer), dtype=np.int32))


def capture_state(env):
    og = env.observation_generator
    kf = og.kalman_filter
    ist, mst = env.interceptor_state, env.missile_state
    s = dict(
        int_pos=np.array(ist["position"], np.float64), int_vel=np.array(ist["velocity"], np.float64),
        int_quat=np.array(ist["orientation"], np.float64), fuel=np.float64(ist["fuel"]),
        mis_pos=np.array(mst["position"], np.float64), mis_vel=np.array(mst["velocity"], np.float64),
        wind=np.array(env.current_wind, np.float64),
        thrust_actual=np.array(getattr(env, "interceptor_thrust_actual", np.zeros(3)), np.float64),
        steps=np.int64(env.steps),
        prev_distance=np.float64(env._prev_distance), min_distance=np.float64(env._episode_min_distance),
        last_distance=np.float64(env._last_distance), worsening=np.int64(env._distance_worsening_count),
        crossed=np.int64(bool(env._crossed_threshold)),
        kf_init=np.int64(bool(kf.initialized)), kf_x=np.array(kf.state, np.float64),
        kf_x_is64=np.int64(kf.state.dtype == np.float64), kf_P=np.array(kf.P, np.float64),
        total_fuel_used=np.float64(env.total_fuel_used),
    )
    if getattr(env, "volley_mode", False):
        ms = env.missile_states
        s.update(v_pos=np.array([m["position"] for m in ms], np.float64), v_vel=np.array([m["velocity"] for m in ms], np.float64),
                 v_active=np.array([bool(m["active"]) for m in ms], np.int64),
                 v_min=np.array(env.missile_min_distances, np.float64),
                 prio=np.int64([i for i, m in enumerate(ms) if m is env.missile_state][0]))
    on = ring_dump(og.sensor_delay_buffer, 3, lambda m: m["rel_pos"])
    gr = ring_dump(og.ground_sensor_delay_buffer, 7,
                   lambda m: np.concatenate([m["rel_pos"], m["rel_vel"], [m["quality"]]]))
    s.update(on_delay=np.int64(on["delay"]), on_count=np.int64(on["count"]), on_ring=on["data"], on_det=on["det"],
             g_delay=np.int64(gr["delay"]), g_count=np.int64(gr["count"]), g_ring=gr["data"])
    if env.atmospheric_model is not None:
        s["T0"] = np.float64(env.atmospheric_model.constants.SEA_LEVEL_TEMPERATURE)
    else:
        s["T0"] = np.float64(288.15)
    if env.mach_drag_model is not None:
        s["base_cd"] = np.float64(env.mach_drag_model.base_cd)
        s["transonic_peak"] = np.float64(env.mach_drag_model.transonic_peak_multiplier)
    else:
        s["base_cd"], s["transonic_peak"] = np.float64(0.3), np.float64(3.0)
    return s


def stack_states(states):
    keys = states[0].keys()
    out = {}
    for k in keys:
        vals = [s[k] for s in states]
        if k in ("on_ring", "g_ring", "on_det"):
            # ragged during ring warm-up: pad to the max depth with NaN / -1
            depth = max(v.shape[0] for v in vals)
            width = vals[0].shape[1] if vals[0].ndim == 2 else None
            if width is None:
                pad = np.full((len(vals), depth), -1, np.int32)
                for i, v in enumerate(vals):
                    pad[i, :len(v)] = v
            else:
                pad = np.full((len(vals), depth, width), np.nan)
                for i, v in enumerate(vals):
                    pad[i, :v.shape[0]] = v
            out[k] = pad
        else:
            out[k] = np.stack(vals)
    return out


# --------------------------------------------------------------------------------------
# case running
# --------------------------------------------------------------------------------------
def load_yaml(rel):
    with open(os.path.join(REF, rel)) as f:
        return yaml.safe_load(f)


def scenario_config(name, physics="config", overrides=None, base="config.yaml"):
    """Env config the way the specialist trainer / offline inference build it:
    config.yaml `environment` shallow-updated by the scenario's `environment`
    (rl_system/inference.py:383-390), `curriculum` and `physics_enhancements` merged in
    (rl_system/scripts/train_hrl_pretrain.py:335-338)."""
    cfg = load_yaml(base)
    env_cfg = copy.deepcopy(cfg["environment"])
    if name is not None:
        env_cfg.update(copy.deepcopy(load_yaml(f"configs/scenarios/{name}.yaml")["environment"]))
    if physics == "config":
        env_cfg["curriculum"] = copy.deepcopy(cfg.get("curriculum", {}))
        env_cfg["physics_enhancements"] = copy.deepcopy(cfg.get("physics_enhancements", {}))
    elif physics == "base":
        env_cfg["curriculum"] = copy.deepcopy(cfg.get("curriculum", {}))
        env_cfg["physics_enhancements"] = {"enabled": False}
    elif physics == "v2":
        # constructor defaults = everything on (what train_flat_ppo.py:369 effectively runs)
        env_cfg["curriculum"] = copy.deepcopy(cfg.get("curriculum", {}))
        env_cfg["physics_enhancements"] = {"enabled": True}
    elif physics == "v2dr":
        env_cfg["curriculum"] = copy.deepcopy(cfg.get("curriculum", {}))
        pe = copy.deepcopy(cfg.get("physics_enhancements", {}))
        for k in ("atmospheric_model", "sensor_delays", "mach_effects", "thrust_dynamics", "enhanced_wind",
                  "domain_randomization"):
            pe[k]["enabled"] = True
        env_cfg["physics_enhancements"] = pe
    elif physics == "flat":
        pass  # train_flat_ppo.py:369 - only `environment`, env falls back to ctor defaults
    else:
        raise ValueError(physics)
    for path, val in (overrides or {}).items():
        d = env_cfg
        keys = path.split(".")
        for k in keys[:-1]:
            d = d.setdefault(k, {})
        d[keys[-1]] = val
    return env_cfg


def pursuit_action(env, rng, gain=1.0, jitter=0.2):
    """Crude pursuit controller so that intercepts actually occur in some cases."""
    rel = env.missile_state["position"] - env.interceptor_state["position"]
    closing = env.missile_state["velocity"] - env.interceptor_state["velocity"]
    tgo = np.clip(np.linalg.norm(rel) / (np.linalg.norm(closing) + 1e-3), 0.05, 5.0)
    aim = rel + closing * tgo
    a = np.zeros(6, np.float32)
    d = aim / (np.linalg.norm(aim) + 1e-6)
    d[2] += 0.25  # fight gravity
    if env.observation_mode == "los_frame":
        a[0:3] = [gain, 0.0, 0.3]
    else:
        a[0:3] = np.clip(gain * d, -1, 1)
    a[3:6] = 0.02 * rng.standard_normal(3)
    a[0:3] += jitter * rng.standard_normal(3)
    return np.clip(a, -1, 1).astype(np.float32)


RADAR_ONLY = False     # `make_golden.py radar`: re-run a case and keep only info['radar_debug'] per step (radar/<name>.npz)


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def run_case(name, env_cfg, n_steps, seed, policy="random", tweak=None, global_step=0, reseed_each_reset=False,
             state_every=1):
    from environment import InterceptEnvironment

    env = InterceptEnvironment(copy.deepcopy(env_cfg))
    if env.physics_randomizer is not None:
        # the randomiser's generator is entropy-seeded (physics_randomizer.py:128); its own seed() hook makes the
        # domain-randomised fixtures reproducible (`--check`)
        env.physics_randomizer.seed(seed + 4242)
    if global_step:
        env.set_training_step_count(global_step)
    arng = _orig_default_rng(seed + 77)
    TAPE.volley = bool(env.volley_mode)
    K = int(env.volley_size) if env.volley_mode else 1
    if env.volley_mode:
        # tell the evasion recorder which missile a draw belongs to (only active missiles draw: environment.py:632-636)
        inner = env._update_missile_state

        def tagged(missile_state):
            TAPE.cur_missile = [i for i, m in enumerate(env.missile_states) if m is missile_state][0]
            try:
                return inner(missile_state)
            finally:
                TAPE.cur_missile = 0

        def _update_missile_state(missile_state):   # name matters: the recorder asserts on its caller's name
            return tagged(missile_state)

        env._update_missile_state = _update_missile_state

    def do_reset(s):
        TAPE.clear()
        TAPE.mode = "reset"
        TAPE._uniform_log = []
        obs, info = env.reset(seed=s)
        # assign the global-stream uniform draws to slots
        calls = TAPE._uniform_log
        spherical = env.missile_spawn_range.get("position_mode", "box") == "spherical"
        slots = TAPE.reset
        idx = 0
        for k in range(K):                       # per missile: position (1 vector or 3 scalar calls), then speed
            base = 0 if k == 0 else 32 + 4 * (k - 1)
            if spherical:
                for j in range(3):
                    slots[base + j] = calls[idx][1][0]
                    idx += 1
            else:
                slots[base:base + 3] = calls[idx][1]
                idx += 1
            slots[base + 3] = calls[idx][1][0]
            idx += 1
        slots[4:7] = calls[idx][1]
        idx += 1
        last = calls[idx][1]
        if len(last) == 3:
            slots[7:10] = last
        else:
            slots[7] = last[0]
        assert idx == len(calls) - 1
        return obs.copy(), slots.copy()

    obs0, reset_noise0 = do_reset(seed)
    if tweak is not None:
        tweak(env)
    init_state = capture_state(env)
    rec = dict(action=[], step_noise=[], obs=[], reward=[], terminated=[], truncated=[], distance=[],
               intercepted=[], hit_target=[], info_min_distance=[], fuel_used=[], state=[],
               missiles_intercepted=[], missiles_remaining=[], missile_min_distances=[],
               did_reset=[], reset_noise=[], reset_obs=[], reset_state=[], radius=[], st_index=[])
    for t in range(n_steps):
        if policy == "random":
            a = arng.uniform(-1, 1, 6).astype(np.float32)
        elif policy == "pursuit":
            a = pursuit_action(env, arng)
        elif policy == "coast":
            a = np.zeros(6, np.float32)
            a[3:6] = arng.uniform(-1, 1, 3)
        elif policy == "hover":
            a = np.zeros(6, np.float32)
            a[2] = 0.49
            a += 0.02 * arng.standard_normal(6).astype(np.float32)
        elif callable(policy):
            a = policy(env, arng, t)
        else:
            raise ValueError(policy)
        TAPE.clear()
        TAPE.mode = "step"
        obs, r, term, trunc, info = env.step(a)
        rec["action"].append(a)
        rec["step_noise"].append(TAPE.step.copy())
        rec["obs"].append(obs.copy())
        rec["reward"].append(np.float64(r))
        rec["terminated"].append(bool(term))
        rec["truncated"].append(bool(trunc))
        rec["distance"].append(np.float64(info["distance"]))
        rec["intercepted"].append(bool(info["intercepted"]))
        rec["hit_target"].append(bool(info["missile_hit_target"]))
        rec["info_min_distance"].append(np.float64(info["min_distance"]))
        rec["fuel_used"].append(np.float64(info["fuel_used"]))
        rec["radius"].append(np.float64(env.get_current_intercept_radius()))
        if RADAR_ONLY:
            rec.setdefault("radar", []).append(_flatten(info["radar_debug"]))
        if env.volley_mode:
            rec["missiles_intercepted"].append(np.int64(info["missiles_intercepted"]))
            rec["missiles_remaining"].append(np.int64(info["missiles_remaining"]))
            rec["missile_min_distances"].append(np.array(info["missile_min_distances"], np.float64))
        if t % state_every == 0 or term or trunc or t == n_steps - 1:
            # full post-step state (long cases keep every `state_every`-th one to stay small)
            rec["state"].append(capture_state(env))
            rec["st_index"].append(t)
        if term or trunc:
            ro, rn = do_reset(seed + 1000 + t if reseed_each_reset else None)
            rec["did_reset"].append(True)
            rec["reset_noise"].append(rn)
            rec["reset_obs"].append(ro)
            rec["reset_state"].append(capture_state(env))
        else:
            rec["did_reset"].append(False)
    if RADAR_ONLY:
        # same seeds, same draws: the trajectory must be the one already stored in <name>.npz
        main = np.load(os.path.join(OUT, name + ".npz"))
        assert np.array_equal(main["obs"], np.array(rec["obs"])), name + ": re-run differs from the stored fixture"
        cols = {k: np.array([r[k] for r in rec["radar"]]) for k in rec["radar"][0]}
        os.makedirs(os.path.join(OUT, "radar"), exist_ok=True)
        np.savez_compressed(os.path.join(OUT, "radar", name + ".npz"), **cols)
        reasons = {k: sorted(set(cols[k].tolist())) for k in cols if k.endswith("detection_reason")}
        print(f"{name:34s} steps={n_steps:5d} radar_debug columns={len(cols)} reasons={reasons}")
        return
    og = env.observation_generator
    gr = og.ground_radar
    effective = dict(
        dt=env.dt, max_steps=env.max_steps, max_range=env.max_range, max_velocity=env.max_velocity,
        target_pos=[float(x) for x in env.target_position],
        atmosphere=env.atmospheric_model is not None, mach_drag=env.mach_drag_model is not None,
        enhanced_wind=env.enhanced_wind_model is not None, thrust_lag=bool(env.thrust_dynamics_enabled),
        domain_randomization=env.physics_randomizer is not None and bool(env.physics_randomizer.enabled),
        validation=bool(env.physics_validation_enabled), evasion=bool(env.config.get("missile_evasion", False)),
        base_wind=[float(x) for x in env.base_wind], wind_variability=env.wind_variability,
        use_curriculum=bool(env.use_curriculum), initial_radius=env.initial_intercept_radius,
        final_radius=env.final_intercept_radius, curriculum_steps=env.curriculum_steps,
        precision_mode=bool(env.precision_mode), proximity_fuze=bool(env.proximity_fuze_enabled),
        proximity_kill_radius=env.proximity_kill_radius,
        radar_curriculum_active=bool(env.use_radar_curriculum and env.radar_curriculum_config),
        radar_quality=env.radar_quality, radar_range=og.radar_range,
        radar_beam_width_now=og.radar_beam_width, onboard_reliability_now=og.onboard_detection_reliability,
        ground_reliability_now=og.ground_detection_reliability, intercept_radius_now=env.get_current_intercept_radius(),
        onboard_delay=(og.sensor_delay_buffer.delay_samples if og.sensor_delay_buffer else 0) if not (
            env.physics_randomizer is not None and env.physics_randomizer.enabled) else -1,
        ground_enabled=gr is not None,
        ground_delay=og.ground_sensor_delay_buffer.delay_samples if og.ground_sensor_delay_buffer else 0,
        obs_mode=og.observation_mode,
        volley_size=(int(env.volley_size) if env.volley_mode else 0),
    )
    if gr is not None:
        effective.update(ground_pos=[float(x) for x in gr.position], ground_max_range=gr.max_range,
                         ground_min_elev=float(gr.min_elevation_angle), ground_max_elev=float(gr.max_elevation_angle),
                         ground_range_accuracy=gr.range_accuracy, ground_velocity_accuracy=gr.velocity_accuracy,
                         ground_base_quality=gr.base_quality, max_datalink_range=og.max_datalink_range,
                         datalink_packet_loss=og.datalink_packet_loss)
    if env.mach_drag_model is not None and not effective["domain_randomization"]:
        m = env.mach_drag_model
        effective.update(subsonic_mach=m.subsonic_mach, supersonic_mach=m.supersonic_mach,
                         transonic_peak_multiplier=m.transonic_peak_multiplier, supersonic_multiplier=m.supersonic_multiplier)
    if env.enhanced_wind_model is not None:
        w = env.enhanced_wind_model
        effective.update(boundary_layer_height=w.boundary_layer_height, turbulence_intensity=w.turbulence_intensity,
                         gust_scale=w.gust_scale)
    if env.thrust_dynamics_enabled:
        effective["thrust_tau"] = env.thrust_time_constant
    out = {"config_json": np.array(json.dumps(env_cfg)), "effective_json": np.array(json.dumps(effective)),
           "numpy_version": np.array(np.__version__),
           "seed": np.int64(seed), "global_step": np.int64(global_step),
           "reset_noise0": reset_noise0, "reset_obs0": obs0.astype(np.float32)}
    for k, v in init_state.items():
        out["init_" + k] = v
    for k in ("action", "step_noise", "obs", "reward", "terminated", "truncated", "distance", "intercepted",
              "hit_target", "info_min_distance", "fuel_used", "did_reset", "radius", "st_index"):
        out[k] = np.array(rec[k])
    if env.volley_mode:
        out["volley_size"] = np.int64(K)
        for k in ("missiles_intercepted", "missiles_remaining", "missile_min_distances"):
            out[k] = np.array(rec[k])
    for k, v in stack_states(rec["state"]).items():
        out["st_" + k] = v
    if rec["reset_state"]:
        out["reset_noise"] = np.array(rec["reset_noise"])
        out["reset_obs"] = np.array(rec["reset_obs"], np.float32)
        for k, v in stack_states(rec["reset_state"]).items():
            out["rst_" + k] = v
    # shrink: float32 where the reference value is float32 anyway
    for k in list(out.keys()):
        v = out[k]
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and k.split("_", 1)[-1] in (
                "int_pos", "int_vel", "int_quat", "mis_pos", "mis_vel", "wind", "thrust_actual", "kf_P", "v_pos", "v_vel"):
            if np.array_equal(v.astype(np.float32).astype(np.float64), v, equal_nan=True):
                out[k] = v.astype(np.float32)  # (simple-wind `wind` is float64 in the reference: kept as is)
    n_ep = int(np.sum(out["did_reset"]))
    print(f"{name:34s} steps={n_steps:5d} episodes_ended={n_ep:3d} intercepts={int(np.sum(out['intercepted'])):3d} "
          f"reward_sum={float(np.sum(out['reward'])):12.2f}")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


# ---- forced edge cases: tweak the reference env's state right after reset ------------
def tw_fuel_low(env):
    env.interceptor_state["fuel"] = np.float32(0.05)


def tw_missile_low_near(env):
    env.missile_state["position"][:] = np.array([120.0, 90.0, 6.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-40.0, -30.0, -60.0], np.float32)


def tw_missile_low_far(env):
    env.missile_state["position"][:] = np.array([900.0, 800.0, 5.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-40.0, -30.0, -60.0], np.float32)


def tw_interceptor_dive(env):
    env.interceptor_state["position"][:] = np.array([10.0, 20.0, 3.0], np.float32)
    env.interceptor_state["velocity"][:] = np.array([30.0, 10.0, -40.0], np.float32)


def tw_close(env):
    env.interceptor_state["position"][:] = env.missile_state["position"] - np.array([150.0, 120.0, 100.0], np.float32)
    env.interceptor_state["velocity"][:] = np.array([200.0, 160.0, 120.0], np.float32)
    d = np.linalg.norm(env.missile_state["position"] - env.interceptor_state["position"])
    env._prev_distance = np.float32(d)
    env._last_distance = env._prev_distance
    env._episode_min_distance = env._prev_distance


def tw_fast(env):
    # interceptor/missile speeds spanning Mach 0.8 / 1.2
    env.interceptor_state["velocity"][:] = np.array([150.0, 150.0, 160.0], np.float32)
    env.missile_state["velocity"] *= np.float32(2.2)


def tw_look_away(env):
    # flip the radar away from the target: onboard loses the beam, ground radar may remain
    env.interceptor_state["orientation"][:] = np.array([0.0, 1.0, 0.0, 0.0], np.float32)


def tw_flyaway(env):
    # missile far away and receding: after step 1000 the "distance worsening" counter climbs every step and
    # the smart early termination (environment.py:795-811) fires at step 1501
    env.missile_state["position"] += np.array([1500.0, 1500.0, 900.0], np.float32)
    env.missile_state["velocity"][:] = np.array([150.0, 150.0, 30.0], np.float32)
    d = np.linalg.norm(env.missile_state["position"] - env.interceptor_state["position"])
    env._prev_distance = np.float32(d)
    env._last_distance = env._prev_distance
    env._episode_min_distance = env._prev_distance


def tw_blind(env):
    # missile below ground-radar horizon and outside onboard range -> KF never initialised
    env.missile_state["position"][:] = np.array([5200.0, 5100.0, 180.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-150.0, -150.0, -2.0], np.float32)
    d = np.linalg.norm(env.missile_state["position"] - env.interceptor_state["position"])
    env._prev_distance = np.float32(d)
    env._last_distance = env._prev_distance
    env._episode_min_distance = env._prev_distance


def _sync_distances(env):
    d = np.linalg.norm(env.missile_state["position"] - env.interceptor_state["position"])
    env._prev_distance = np.float32(d)
    env._last_distance = env._prev_distance
    env._episode_min_distance = env._prev_distance


def tw_alt_11km(env):
    # both vehicles cross the 11 km tropopause (physics_models.py:70-74, 95-108): interceptor climbing, missile descending
    env.interceptor_state["position"][:] = np.array([10.0, 20.0, 10985.0], np.float32)
    env.interceptor_state["velocity"][:] = np.array([30.0, 20.0, 180.0], np.float32)
    env.missile_state["position"][:] = np.array([1500.0, 1400.0, 11020.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-120.0, -110.0, -160.0], np.float32)
    _sync_distances(env)


def tw_alt_20km(env):
    # ... and the 20 km boundary of the isothermal layer (physics_models.py:72-78, 102-113)
    env.interceptor_state["position"][:] = np.array([10.0, 20.0, 19985.0], np.float32)
    env.interceptor_state["velocity"][:] = np.array([30.0, 20.0, 180.0], np.float32)
    env.missile_state["position"][:] = np.array([1500.0, 1400.0, 20025.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-120.0, -110.0, -160.0], np.float32)
    _sync_distances(env)


def tw_ground_out_of_range(env):
    # missile just outside the ground radar's 20 km (core.py:396-397 'out_of_range'), closing fast: comes into range mid-case
    env.missile_state["position"][:] = np.array([14200.0, 14100.0, 1500.0], np.float32)
    env.missile_state["velocity"][:] = np.array([-520.0, -510.0, -20.0], np.float32)
    _sync_distances(env)


def tw_ground_above_coverage(env):
    # missile almost overhead of the station at (0, 0, 100): elevation > 85 deg (core.py:405-406 'above_coverage'), drifting out
    env.missile_state["position"][:] = np.array([90.0, 60.0, 3000.0], np.float32)
    env.missile_state["velocity"][:] = np.array([160.0, 120.0, -40.0], np.float32)
    _sync_distances(env)


def spin_then_random(env, rng, t):
    a = rng.uniform(-1, 1, 6).astype(np.float32)
    if (t // 40) % 2 == 0:
        a[3:6] = np.array([1.0, -1.0, 0.7], np.float32)  # tumble: beam sweeps on/off the target
    return a


def _npz_equal(a, b):
    fa, fb = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
    if sorted(fa.files) != sorted(fb.files):
        return "key sets differ: " + str(sorted(set(fa.files) ^ set(fb.files)))
    for k in fa.files:
        x, y = fa[k], fb[k]
        if x.dtype != y.dtype or x.shape != y.shape:
            return f"{k}: {x.dtype}{x.shape} vs {y.dtype}{y.shape}"
        same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        if not same:
            return f"{k}: values differ"
    return None


def check():
    """`make_golden.py --check`: regenerate every fixture from the reference into a scratch directory and compare it,
    array by array and bit for bit, with the committed file -- fixture drift (or a reference / numpy change) shows up as
    a named difference instead of going unnoticed."""
    global OUT
    import tempfile
    committed = OUT
    with tempfile.TemporaryDirectory() as tmp:
        OUT = tmp
        generate_all()
        radar_cases(scenario_config)
        bad, n = [], 0
        for sub in ("", "radar"):
            have = sorted(f for f in os.listdir(os.path.join(committed, sub)) if f.endswith(".npz"))
            made = sorted(f for f in os.listdir(os.path.join(tmp, sub)) if f.endswith(".npz")) if os.path.isdir(os.path.join(tmp, sub)) else []
            for f in sorted(set(have) | set(made)):
                n += 1
                if f not in have or f not in made:
                    bad.append((os.path.join(sub, f), "only " + ("committed" if f in have else "regenerated")))
                    continue
                why = _npz_equal(os.path.join(committed, sub, f), os.path.join(tmp, sub, f))
                if why:
                    bad.append((os.path.join(sub, f), why))
        OUT = committed
    for f, why in bad:
        print("DIFFERS", f, why)
    print(f"check: {n - len(bad)} of {n} fixtures regenerate bit-identically")
    return 1 if bad else 0


def main():
    _install_gym_shim()
    sys.path.insert(0, REF)
    np.random.default_rng = _rec_default_rng
    np.random.uniform = _rec_uniform
    np.random.randn = _rec_randn
    _selfcheck_wrappers()
    S = scenario_config
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(check())
    if len(sys.argv) > 1 and sys.argv[1] == "radar":
        radar_cases(S)       # adds / refreshes radar/<name>.npz for a few of the cases below
        return
    if len(sys.argv) > 1 and sys.argv[1] == "volley":
        volley_cases(S)      # adds / refreshes the volley fixtures only
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        round2_cases(S)      # adds / refreshes the fixtures added in round 2 only
        return
    for f in os.listdir(OUT):
        if f.endswith(".npz"):
            os.remove(os.path.join(OUT, f))
    generate_all()


def round2_cases(S):
    """Round 2: the reference's own config.yaml physics (ISA atmosphere only) on the medium / hard scenarios and in volley
    mode -- what the `config` / `config-volley` kernel variants run --, the ISA layers above 11 km / 20 km, and the two
    ground-radar reasons no earlier trajectory reached."""
    V = lambda k, extra=None: dict({"volley_mode": True, "volley_size": k}, **(extra or {}))  # noqa: E731
    run_case("medium_config_random", S("medium", "config"), 400, 1200)
    run_case("hard_config_random", S("hard", "config"), 300, 1201)
    run_case("medium_config_pursuit", S("medium", "config"), 2000, 1202, policy="pursuit", state_every=25)
    run_case("medium_config_short_eps", S("medium", "config", {"max_steps": 60}), 400, 1203)
    run_case("volley3_medium_config_pursuit", S("medium", "config", V(3)), 2500, 1204, policy="pursuit", state_every=25)
    run_case("volley3_medium_config_short_eps", S("medium", "config", V(3, {"max_steps": 50})), 300, 1205)
    run_case("edge_isa_11km_v2", S("medium", "v2"), 120, 1210, tweak=tw_alt_11km)
    run_case("edge_isa_20km_v2", S("medium", "v2"), 120, 1211, tweak=tw_alt_20km)
    run_case("edge_isa_11km_config", S("medium", "config"), 120, 1212, tweak=tw_alt_11km)
    run_case("edge_isa_20km_config", S("medium", "config"), 120, 1213, tweak=tw_alt_20km)
    run_case("edge_ground_out_of_range", S("medium", "base"), 60, 1214, tweak=tw_ground_out_of_range, policy="coast")
    run_case("edge_ground_above_coverage", S("medium", "base"), 60, 1215, tweak=tw_ground_above_coverage, policy="coast")


def generate_all():
    S = scenario_config

    # --- scenario x physics, random actions -------------------------------------------
    run_case("easy_config_random", S("easy", "config"), 300, 1000)
    run_case("medium_base_random", S("medium", "base"), 400, 1001)
    run_case("medium_v2_random", S("medium", "v2"), 400, 1002)
    run_case("hard_base_random", S("hard", "base"), 250, 1003)
    run_case("hard_v2_random", S("hard", "v2"), 250, 1004)
    run_case("medium_flat_defaults", S("medium", "flat"), 250, 1005)
    # short episodes -> many auto-resets (truncation), incl. domain randomisation
    run_case("medium_base_short_eps", S("medium", "base", {"max_steps": 60}), 400, 1010)
    run_case("medium_v2dr_short_eps", S("medium", "v2dr", {"max_steps": 50}), 420, 1011)
    run_case("medium_v2dr_seeded_resets", S("medium", "v2dr", {"max_steps": 40}), 200, 1012, reseed_each_reset=True)
    # --- pursuit controller: real intercepts, terminal rewards ------------------------
    run_case("easy_config_pursuit", S("easy", "config"), 1500, 1020, policy="pursuit", state_every=25)
    run_case("medium_base_pursuit", S("medium", "base"), 2000, 1021, policy="pursuit", state_every=25)
    run_case("medium_v2_pursuit", S("medium", "v2"), 2000, 1022, policy="pursuit", state_every=25)
    run_case("medium_base_pursuit_late_curric", S("medium", "base"), 2000, 1023, policy="pursuit",
             global_step=1500000, state_every=25)
    # --- observation modes -------------------------------------------------------------
    run_case("medium_v2_body_random", S("medium", "v2", {"observation_mode": "body_frame"}), 250, 1030,
             policy=spin_then_random)
    run_case("medium_base_rotinv_random", S("medium", "base", {"rotation_invariant": True}), 200, 1031)
    run_case("medium_v2_los_random", S("medium", "v2", {"observation_mode": "los_frame"}), 250, 1032,
             policy=spin_then_random)
    ev = load_yaml("configs/eval_360_los.yaml")
    ec = copy.deepcopy(ev["environment"])
    ec["curriculum"] = copy.deepcopy(ev["curriculum"])
    ec["physics_enhancements"] = copy.deepcopy(ev["physics_enhancements"])
    run_case("eval360_los_fuze_pursuit", ec, 1200, 1033, policy="pursuit", state_every=10)
    ec2 = copy.deepcopy(ec)
    ec2["proximity_fuze_enabled"] = False
    ec2["curriculum"]["precision_mode"] = True
    run_case("eval360_los_precision_pursuit", ec2, 1200, 1034, policy="pursuit", state_every=10)
    # --- reward / termination modes ----------------------------------------------------
    run_case("medium_base_precision_pursuit", S("medium", "base", {"curriculum.precision_mode": True}), 2000, 1040,
             policy="pursuit", state_every=25)
    run_case("medium_v2_fuze_pursuit", S("medium", "v2", {"proximity_fuze_enabled": True,
                                                         "proximity_kill_radius": 30.0}), 2000, 1041,
             policy="pursuit", state_every=25)
    run_case("medium_base_nocurric", S("medium", "base", {"curriculum.enabled": False}), 150, 1042)
    # --- forced edge cases ---------------------------------------------------------------
    run_case("edge_fuel_out", S("medium", "v2"), 60, 1050, tweak=tw_fuel_low)
    run_case("edge_fuel_out_base", S("medium", "base"), 60, 1051, tweak=tw_fuel_low)
    run_case("edge_ground_hit_near", S("medium", "base"), 40, 1052, tweak=tw_missile_low_near)
    run_case("edge_ground_hit_far", S("medium", "v2"), 40, 1053, tweak=tw_missile_low_far)
    run_case("edge_crash", S("medium", "base"), 40, 1054, tweak=tw_interceptor_dive)
    run_case("edge_close_intercept", S("medium", "base"), 80, 1055, tweak=tw_close, policy="coast")
    run_case("edge_close_precision", S("medium", "v2", {"curriculum.precision_mode": True}), 120, 1056,
             tweak=tw_close, policy="coast")
    run_case("edge_mach_sweep", S("hard", "v2"), 200, 1057, tweak=tw_fast, policy="pursuit")
    run_case("edge_look_away", S("medium", "v2"), 120, 1058, tweak=tw_look_away, policy="coast")
    run_case("edge_early_termination", S("medium", "base", {"max_steps": 4000}), 1560, 1059, tweak=tw_flyaway,
             policy="hover", state_every=50)
    run_case("edge_blind_kf_uninit", S("medium", "v2"), 80, 1060, tweak=tw_blind, policy="coast")
    run_case("edge_radar_curriculum_mid", S("medium", "base"), 120, 1061, global_step=6500000,
             policy=spin_then_random)
    run_case("edge_no_ground_radar", S("medium", "base", {"ground_radar": {"enabled": False}}), 100, 1062)
    volley_cases(S)
    round2_cases(S)


def radar_cases(S):
    """info['radar_debug'] (environment.py:842, core.py:650-683) of cases that already have a fixture: every
    detection_reason of both radars, the delay lines, the radar curriculum, volley priority switching."""
    global RADAR_ONLY
    RADAR_ONLY = True
    V = lambda k, extra=None: dict({"volley_mode": True, "volley_size": k}, **(extra or {}))  # noqa: E731
    run_case("medium_base_random", S("medium", "base"), 400, 1001)
    run_case("medium_v2_random", S("medium", "v2"), 400, 1002)
    run_case("medium_base_short_eps", S("medium", "base", {"max_steps": 60}), 400, 1010)
    run_case("edge_ground_hit_near", S("medium", "base"), 40, 1052, tweak=tw_missile_low_near)
    run_case("edge_ground_hit_far", S("medium", "v2"), 40, 1053, tweak=tw_missile_low_far)
    run_case("edge_look_away", S("medium", "v2"), 120, 1058, tweak=tw_look_away, policy="coast")
    run_case("edge_blind_kf_uninit", S("medium", "v2"), 80, 1060, tweak=tw_blind, policy="coast")
    run_case("edge_radar_curriculum_mid", S("medium", "base"), 120, 1061, global_step=6500000,
             policy=spin_then_random)
    run_case("edge_no_ground_radar", S("medium", "base", {"ground_radar": {"enabled": False}}), 100, 1062)
    run_case("volley3_medium_base_random", S("medium", "base", V(3)), 400, 1100)
    run_case("edge_ground_out_of_range", S("medium", "base"), 60, 1214, tweak=tw_ground_out_of_range, policy="coast")
    run_case("edge_ground_above_coverage", S("medium", "base"), 60, 1215, tweak=tw_ground_above_coverage, policy="coast")
    RADAR_ONLY = False


def tw_volley_ground(env):
    # missile 1 about to hit the ground next to the target (mission failure), missile 2 far from it (harmless)
    ms = env.missile_states
    ms[1]["position"][:] = np.array([130.0, 80.0, 5.0], np.float32)
    ms[1]["velocity"][:] = np.array([-40.0, -30.0, -70.0], np.float32)
    ms[2]["position"][:] = np.array([1400.0, 900.0, 9.0], np.float32)
    ms[2]["velocity"][:] = np.array([-40.0, -30.0, -50.0], np.float32)


def tw_volley_close(env):
    # interceptor right behind the volley: several intercepts within a few steps
    ms = env.missile_states
    c = ms[0]["position"].copy()
    for k, m in enumerate(ms[1:], 1):
        m["position"][:] = c + np.array([25.0 * k, -20.0 * k, 15.0 * k], np.float32)
        m["velocity"][:] = ms[0]["velocity"]
    env.interceptor_state["position"][:] = c - np.array([150.0, 120.0, 100.0], np.float32)
    env.interceptor_state["velocity"][:] = np.array([200.0, 160.0, 120.0], np.float32)
    ip = env.interceptor_state["position"]
    env.missile_min_distances[:] = [np.linalg.norm(m["position"] - ip) for m in ms]
    d = np.linalg.norm(ms[0]["position"] - ip)
    env._prev_distance = np.float32(d)
    env._last_distance = env._prev_distance
    env._episode_min_distance = env._prev_distance


def volley_cases(S):
    """Volley mode (environment.py:236-267, 386-439, 470-487, 631-692, 724-748): K missiles per environment."""
    V = lambda k, extra=None: dict({"volley_mode": True, "volley_size": k}, **(extra or {}))  # noqa: E731
    run_case("volley3_medium_base_random", S("medium", "base", V(3)), 400, 1100)
    run_case("volley3_medium_base_pursuit", S("medium", "base", V(3)), 2500, 1101, policy="pursuit", state_every=25)
    run_case("volley3_medium_v2_fuze_pursuit", S("medium", "v2", V(3, {"proximity_fuze_enabled": True,
                                                                      "proximity_kill_radius": 30.0})), 2500, 1102,
             policy="pursuit", state_every=25)
    run_case("volley2_medium_v2_los_pursuit", S("medium", "v2", V(2, {"observation_mode": "los_frame"})), 1500, 1103,
             policy="pursuit", state_every=25)
    run_case("volley4_medium_v2dr_short_eps", S("medium", "v2dr", V(4, {"max_steps": 50})), 300, 1104)
    run_case("volley3_edge_ground_hits", S("medium", "base", V(3)), 60, 1105, tweak=tw_volley_ground)
    run_case("volley3_edge_close_intercepts", S("medium", "base", V(3)), 120, 1106, tweak=tw_volley_close, policy="coast")
    ev = load_yaml("configs/eval_360_los.yaml")
    ec = copy.deepcopy(ev["environment"])
    ec["curriculum"] = copy.deepcopy(ev["curriculum"])
    ec["physics_enhancements"] = copy.deepcopy(ev["physics_enhancements"])
    ec.update(V(3))
    run_case("volley3_eval360_spherical_pursuit", ec, 1500, 1107, policy="pursuit", state_every=25)


if __name__ == "__main__":
    main()
