"""The parameter resolver (hlynr_intercept_amd/config.py) against the EFFECTIVE values the reference's
constructor arrived at, recorded per fixture by tests/golden/make_golden.py (`effective_json`), and the
built-in scenario presets against the configs the fixtures were generated from."""
import json
import math

import pytest

from hlynr_intercept_amd.config import ConfigError, resolve_config
from hlynr_intercept_amd.scenarios import scenario_config
from tests.golden_util import fixture_names, load_fixture

_OBS = {"world_frame": 0, "body_frame": 1, "los_frame": 2}


@pytest.mark.parametrize("name", fixture_names())
def test_resolver_matches_reference_constructor(name):
    fx = load_fixture(name)
    eff = json.loads(str(fx["effective_json"]))
    rc = resolve_config(fx["config"])
    for k in ("dt", "max_steps", "max_range", "max_velocity", "atmosphere", "mach_drag", "enhanced_wind", "thrust_lag",
              "domain_randomization", "validation", "evasion", "wind_variability", "use_curriculum", "initial_radius",
              "final_radius", "curriculum_steps", "precision_mode", "proximity_fuze", "proximity_kill_radius",
              "radar_quality", "radar_range", "ground_enabled", "ground_delay"):
        assert getattr(rc, k) == eff[k], (k, getattr(rc, k), eff[k])
    assert rc.target_pos == eff["target_pos"] and rc.base_wind == pytest.approx(eff["base_wind"], rel=1e-7)
    assert rc.radar_curriculum.active == eff["radar_curriculum_active"]
    assert rc.obs_mode == _OBS[eff["obs_mode"]]
    if eff["onboard_delay"] >= 0:
        assert rc.onboard_delay == eff["onboard_delay"]
    if eff["ground_enabled"]:
        for k in ("ground_pos", "ground_max_range", "ground_range_accuracy", "ground_velocity_accuracy",
                  "ground_base_quality", "max_datalink_range", "datalink_packet_loss"):
            assert getattr(rc, k) == eff[k], k
        assert rc.ground_min_elev == eff["ground_min_elev"] and rc.ground_max_elev == eff["ground_max_elev"]
    for k in ("subsonic_mach", "supersonic_mach", "transonic_peak_multiplier", "supersonic_multiplier",
              "boundary_layer_height", "turbulence_intensity", "gust_scale", "thrust_tau"):
        if k in eff:
            assert getattr(rc, k) == eff[k], k
    # curriculum position at the end of the recorded run
    gs = fx["global_step_or_none"]
    assert rc.intercept_radius(0 if gs is None else gs) == pytest.approx(eff["intercept_radius_now"], rel=1e-12)
    sched = rc.radar_schedule(gs)
    assert sched["beam_width"] == pytest.approx(eff["radar_beam_width_now"], rel=1e-12)
    assert sched["onboard_reliability"] == pytest.approx(eff["onboard_reliability_now"], rel=1e-12)
    assert sched["ground_reliability"] == pytest.approx(eff["ground_reliability_now"], rel=1e-12)


@pytest.mark.parametrize("scenario,physics,fixture", [
    ("easy", "config", "easy_config_random"), ("medium", "base", "medium_base_random"),
    ("medium", "v2", "medium_v2_random"), ("hard", "base", "hard_base_random"), ("hard", "v2", "hard_v2_random"),
    ("medium", "v2dr", "medium_v2dr_seeded_resets"),
])
def test_builtin_scenarios_equal_the_reference_yaml(scenario, physics, fixture):
    """The presets used by bench.py / smoke() resolve to exactly what the reference's YAML resolves to."""
    over = {"max_steps": 40} if fixture == "medium_v2dr_seeded_resets" else None
    mine = resolve_config(scenario_config(scenario, physics, over)).to_dict()
    ref = resolve_config(load_fixture(fixture)["config"]).to_dict()
    assert mine == ref


def test_dead_yaml_keys_stay_dead():
    """Scenario-level radar_range / radar_quality / radar_beam_width are NOT read by the reference
    (environment.py:136-138,153,171-172 read config['radar'][...]); effective values are the defaults."""
    rc = resolve_config(scenario_config("hard", "base"))
    assert (rc.radar_range, rc.radar_quality, rc.radar_beam_width) == (5000.0, 1.0, 60.0)
    rc2 = resolve_config(dict(scenario_config("hard", "base"), radar={"radar_range": 3500.0, "radar_quality": 0.75}))
    assert (rc2.radar_range, rc2.radar_quality) == (3500.0, 0.75)


def test_delay_samples_and_flags():
    rc = resolve_config(scenario_config("medium", "base"))
    assert rc.ground_delay == 5 and rc.onboard_delay == 0 and not rc.atmosphere
    rc = resolve_config(scenario_config("hard", "v2"))
    assert rc.ground_delay == 6 and rc.onboard_delay == 3 and rc.atmosphere and rc.mach_drag and rc.enhanced_wind
    rc = resolve_config({})      # train_flat_ppo.py:369: constructor defaults
    assert rc.initial_radius == 200.0 and rc.final_radius == 20.0 and rc.curriculum_steps == 5000000
    assert rc.ground_enabled is False and rc.ground_delay == 0      # empty ground_radar dict -> no station (core.py:296-320)
    assert rc.ground_min_elev == math.radians(5.0)


def test_curriculum_schedules():
    rc = resolve_config(scenario_config("medium", "base"))
    assert rc.intercept_radius(0) == 100.0 and rc.intercept_radius(2_000_000) == 5.0
    assert rc.intercept_radius(1_000_000) == pytest.approx(52.5)
    assert rc.radar_schedule(None)["beam_width"] == 120.0
    assert rc.radar_schedule(6_500_000)["beam_width"] == pytest.approx(90.0)
    assert rc.radar_schedule(9_000_000)["beam_width"] == 60.0


def test_volley_mode_is_carried_and_oversized_volleys_fail_loudly():
    rc = resolve_config({"volley_mode": True, "volley_size": 3})          # environment.py:42-43
    assert rc.volley_mode and rc.volley_size == 3
    assert not resolve_config({}).volley_mode
    with pytest.raises(ConfigError):                                       # the arena holds at most 4 missiles per env
        resolve_config({"volley_mode": True, "volley_size": 5})
