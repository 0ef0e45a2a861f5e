"""`InterceptEnvironment` (hlynr_intercept_amd/single_env.py): ONE environment with the reference's class name and `gym.Env`
semantics (rl_system/environment.py:15, 353, 605) over the batch kernel -- what inference.py:406, hrl/hierarchical_env.py:175 and
the debug scripts construct directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# environment.py:829-857, every key of the reference's step() info
STEP_KEYS = {"distance", "intercepted", "missile_hit_target", "fuel_remaining", "fuel_used", "clamped", "missile_pos", "interceptor_pos",
             "steps", "radar_detected", "radar_quality", "radar_debug", "volley_mode", "volley_size", "missiles_intercepted",
             "missiles_remaining", "missile_min_distances", "min_distance", "crossed_threshold", "precision_mode",
             "proximity_fuze_enabled", "proximity_fuze_triggered", "proximity_kill_radius"}
RESET_KEYS = {"missile_pos", "interceptor_pos", "distance", "radar_detected", "radar_quality"}      # environment.py:595-601


def _cfg(physics="base", **over):
    from hlynr_intercept_amd.scenarios import scenario_config
    return scenario_config("medium", physics, dict(over))


def test_single_environment_speaks_gym_env_with_the_reference_s_keys_and_types():
    from hlynr_intercept_amd import InterceptEnvironment
    env = InterceptEnvironment(_cfg(max_steps=40), seed=3)
    with pytest.raises(RuntimeError):
        env.step(np.zeros(6, np.float32))
    obs, info = env.reset()
    assert obs.shape == (26,) and obs.dtype == np.float32 and env.observation_space.shape == (26,) and env.action_space.shape == (6,)
    assert set(info) == RESET_KEYS
    st = env.interceptor_state, env.missile_state
    assert np.array_equal(info["interceptor_pos"], st[0]["position"]) and np.array_equal(info["missile_pos"], st[1]["position"])
    assert info["distance"] == pytest.approx(float(np.linalg.norm(st[1]["position"].astype(np.float64) - st[0]["position"])), rel=1e-6)
    assert isinstance(info["radar_detected"], bool) and info["radar_quality"] == env._venv.rc.radar_quality      # base physics: no delay line
    assert set(st[0]) == {"position", "velocity", "orientation", "fuel"} and set(st[1]) == {"position", "velocity"}
    assert env.steps == 0 and env.total_fuel_used == 0.0 and env.dt == 0.01 and env.max_steps == 40
    assert env.get_current_intercept_radius() > 0 and env.target_position.shape == (3,)
    rng = np.random.default_rng(0)
    done, t = False, 0
    while not done:
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, 6))
        t += 1
        assert obs.shape == (26,) and obs.dtype == np.float32 and type(r) is float and type(term) is bool and type(trunc) is bool
        assert set(info) - STEP_KEYS == {"ground_radar_detected"} and not STEP_KEYS - set(info), set(info) ^ STEP_KEYS     # (one key more than the reference)
        assert info["steps"] == env.steps == t and info["fuel_used"] == env.total_fuel_used > 0
        done = term or trunc
    assert t <= 40 and (trunc or term)
    # between the terminal step and reset(): the state the episode ENDED in, as far as it is kept -- never the next episode's
    assert np.array_equal(env.interceptor_state["position"], info["interceptor_pos"]) and env.interceptor_state["fuel"] == info["fuel_remaining"]
    assert np.array_equal(env.missile_state["position"], info["missile_pos"])
    with pytest.raises(KeyError):
        env.interceptor_state["velocity"]
    obs2, info2 = env.reset()
    assert set(info2) == RESET_KEYS and env.steps == 0 and not np.array_equal(obs2, obs)
    env.set_training_step_count(1234)
    assert env.training_step_count == 1234
    g = env.observation_generator
    assert all(hasattr(g, k) for k in ("radar_beam_width", "onboard_detection_reliability", "ground_detection_reliability", "measurement_noise_level"))
    env.close()


def test_single_environment_is_the_batch_kernel_and_its_terminal_observation():
    """The first episode of `InterceptEnvironment(seed=s)` is environment 0 of `HlynrVecEnv(seed=s)`: same observations, rewards,
    flags and info values bit for bit; the observation returned by the step that ends it is the batch's `terminal_observation`."""
    from hlynr_intercept_amd import HlynrVecEnv, InterceptEnvironment
    cfg = _cfg("v2dr", max_steps=60)
    one = InterceptEnvironment(cfg, seed=11)
    vec = HlynrVecEnv(cfg, num_envs=1, seed=11, radar_debug=True)
    o1, i1 = one.reset()
    ov = vec.reset()
    assert np.array_equal(o1, ov[0])
    assert i1["radar_detected"] is False and i1["radar_quality"] == 0.0        # 30 ms onboard delay: the line is still filling (core.py:576-583)
    rng = np.random.default_rng(5)
    for t in range(60):
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        o1, r1, te, tr, inf1 = one.step(a)
        ov, rv, dv, infv = vec.step(a[None])
        iv = infv[0]
        assert r1 == float(rv[0]) and (te or tr) == bool(dv[0])
        for k in ("distance", "min_distance", "fuel_remaining", "fuel_used", "steps", "intercepted", "radar_detected", "crossed_threshold"):
            assert inf1[k] == iv[k], (t, k)
        assert np.array_equal(inf1["missile_pos"], iv["missile_pos"]) and inf1["radar_debug"] == iv["radar_debug"]
        if dv[0]:
            assert np.array_equal(o1, iv["terminal_observation"]) and (tr and not te) == iv["TimeLimit.truncated"]
            break
        assert np.array_equal(o1, ov[0])
    else:
        raise AssertionError("the episode did not end within max_steps")
    one.close(); vec.close()


def test_single_environment_volley_option_and_seed_reproducibility():
    from hlynr_intercept_amd import InterceptEnvironment
    a = InterceptEnvironment(_cfg(max_steps=30), seed=1)
    b = InterceptEnvironment(_cfg(max_steps=30), seed=2)
    oa, _ = a.reset(seed=77)
    ob, _ = b.reset(seed=77)
    assert np.array_equal(oa, ob)                       # reset(seed=) re-keys the generator: same seed, same episode
    oc, _ = b.reset()
    assert not np.array_equal(oc, ob)                   # ... and the next reset draws another one
    obs, info = a.reset(options={"volley_mode": True, "volley_size": 3})        # environment.py:363-366
    assert a.volley_mode and a.volley_size == 3 and len(a.missile_states) == 3 and all(m["active"] for m in a.missile_states)
    obs, r, te, tr, info = a.step(np.zeros(6))
    assert info["volley_mode"] is True and info["volley_size"] == 3 and len(info["missile_min_distances"]) == 3
    assert info["missiles_remaining"] == 3 and info["missiles_intercepted"] == 0
    a.close(); b.close()
