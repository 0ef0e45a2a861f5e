"""Built-in scenario presets (parameter data only), in the reference's own config schema.

The reference builds an env config as `config.yaml:environment` shallow-updated by
`configs/scenarios/<name>.yaml:environment` (rl_system/inference.py:383-390) with the top-level
`curriculum` / `physics_enhancements` blocks merged in (rl_system/scripts/train_hrl_pretrain.py:335-338).
These presets hold the values that procedure yields for the three shipped scenarios, so the bench
and the GPU tests can run where /root/reference does not exist.  `tests/test_config.py` checks them
against the configs recorded in the golden fixtures.  A user's own YAML goes through `load_config`.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional

_GROUND = {"enabled": True, "position": [0, 0, 100], "max_range": 20000.0, "min_elevation_angle": 5.0,
           "max_elevation_angle": 85.0, "range_accuracy": 10.0, "velocity_accuracy": 2.0, "base_quality": 0.95,
           "weather_sensitivity": 0.2, "max_datalink_range": 50000.0, "datalink_packet_loss": 0.05,
           "ground_sensor_delay_ms": 50.0}

_COMMON = {"dt": 0.01, "max_steps": 2000, "max_range": 10000.0, "max_velocity": 1000.0, "target_position": [0, 0, 0]}

# NOTE: radar_range / radar_quality / radar_beam_width at this level are dead keys in the reference
# (it reads config['radar'][...], environment.py:136-138,153,171-172); they are kept so the dict is
# the same one the reference would see.
SCENARIOS: Dict[str, Dict[str, Any]] = {
    "easy": dict(_COMMON,
                 interceptor_spawn={"position": [[0, 0, 0], [30, 30, 10]], "velocity": [[30, 30, 50], [50, 50, 90]]},
                 missile_spawn={"position": [[1000, 1000, 1200], [2000, 2000, 2000]],
                                "velocity": [[-60, -60, -30], [-90, -90, -50]]},
                 wind={"velocity": [2.0, 0.0, 0.0], "variability": 0.05},
                 radar_range=6000.0, radar_noise=0.02, radar_quality=1.0, radar_beam_width=90.0,
                 ground_radar=dict(_GROUND, range_accuracy=8.0, velocity_accuracy=1.5, base_quality=0.98,
                                   weather_sensitivity=0.1, datalink_packet_loss=0.02, ground_sensor_delay_ms=40.0),
                 missile_evasion=False),
    "medium": dict(_COMMON,
                   interceptor_spawn={"position": [[0, 0, 0], [50, 50, 10]], "velocity": [[20, 20, 40], [40, 40, 80]]},
                   missile_spawn={"position": [[1800, 1800, 1800], [3200, 3200, 3200]],
                                  "velocity": [[-110, -110, -55], [-160, -160, -75]]},
                   wind={"velocity": [8.0, 3.0, 0.0], "variability": 0.15},
                   radar_range=4500.0, radar_noise=0.08, radar_quality=0.9, radar_beam_width=60.0,
                   ground_radar=dict(_GROUND), missile_evasion=True),
    "hard": dict(_COMMON,
                 interceptor_spawn={"position": [[0, 0, 0], [70, 70, 15]], "velocity": [[10, 10, 30], [35, 35, 70]]},
                 missile_spawn={"position": [[2500, 2500, 2500], [4000, 4000, 4000]],
                                "velocity": [[-140, -140, -75], [-200, -200, -100]]},
                 wind={"velocity": [15.0, 8.0, -2.0], "variability": 0.25},
                 radar_range=3500.0, radar_noise=0.15, radar_quality=0.75, radar_beam_width=45.0,
                 ground_radar=dict(_GROUND, max_range=18000.0, range_accuracy=15.0, velocity_accuracy=3.0,
                                   base_quality=0.88, weather_sensitivity=0.3, max_datalink_range=45000.0,
                                   datalink_packet_loss=0.10, ground_sensor_delay_ms=60.0),
                 missile_evasion=True),
}

# config.yaml `curriculum` block (radius 100 m -> 5 m over 2 M steps; radar curriculum present)
CURRICULUM = {
    "enabled": True, "initial_radius": 100.0, "final_radius": 5.0, "curriculum_steps": 2000000,
    "radar_curriculum": {
        "enabled": True, "initial_beam_width": 120.0, "final_beam_width": 60.0,
        "beam_width_transition_start": 5000000, "beam_width_transition_end": 8000000,
        "initial_detection_reliability": 1.0, "final_detection_reliability": 1.0,
        "reliability_transition_start": 15000000, "reliability_transition_end": 20000000,
        "initial_ground_reliability": 1.0, "final_ground_reliability": 1.0,
        "ground_reliability_transition_start": 15000000, "ground_reliability_transition_end": 20000000,
        "initial_noise_level": 0.0, "final_noise_level": 0.0,
        "noise_transition_start": 15000000, "noise_transition_end": 20000000,
    },
}

# config.yaml `domain_randomization` variations
_DR = {"enabled": True, "drag_coefficient_variation": 0.2, "air_density_variation": 0.1,
       "sensor_delay_variation": 0.5, "thrust_response_variation": 0.3, "wind_variation": 0.3}

PHYSICS = {
    # BASELINE.json config 2: `physics_enhancements.enabled = false`
    "base": {"enabled": False},
    # config.yaml as shipped: ISA atmosphere only
    "config": {"enabled": True, "atmospheric_model": {"enabled": True}, "sensor_delays": {"enabled": False},
               "mach_effects": {"enabled": False}, "thrust_dynamics": {"enabled": False},
               "domain_randomization": {"enabled": False}, "enhanced_wind": {"enabled": False}},
    # constructor defaults = everything on except domain randomisation (what train_flat_ppo.py:369 runs)
    "v2": {"enabled": True},
    # BASELINE.json config 3: everything on + domain randomisation
    "v2dr": {"enabled": True, "atmospheric_model": {"enabled": True}, "sensor_delays": {"enabled": True, "radar_delay_ms": 30.0},
             "mach_effects": {"enabled": True}, "thrust_dynamics": {"enabled": True}, "enhanced_wind": {"enabled": True},
             "domain_randomization": _DR},
}


def scenario_config(name: str = "medium", physics: str = "base", overrides: Optional[Dict[str, Any]] = None
                    ) -> Dict[str, Any]:
    """Env-config dict for a shipped scenario; `overrides` uses dotted paths ('curriculum.enabled')."""
    if name not in SCENARIOS:
        raise KeyError(f"unknown scenario {name!r}; have {sorted(SCENARIOS)}")
    if physics not in PHYSICS:
        raise KeyError(f"unknown physics preset {physics!r}; have {sorted(PHYSICS)}")
    cfg = copy.deepcopy(SCENARIOS[name])
    cfg["curriculum"] = copy.deepcopy(CURRICULUM)
    cfg["physics_enhancements"] = copy.deepcopy(PHYSICS[physics])
    for path, val in (overrides or {}).items():
        d = cfg
        keys = path.split(".")
        for k in keys[:-1]:
            d = d.setdefault(k, {})
        d[keys[-1]] = val
    return cfg


def load_config(config_yaml: str, scenario_yaml: Optional[str] = None, merge_top_level: bool = True
                ) -> Dict[str, Any]:
    """Build an env config from the reference's YAML files the way its trainers do
    (inference.py:383-390 + train_hrl_pretrain.py:335-338; `merge_top_level=False` = train_flat_ppo.py:369)."""
    import yaml

    with open(config_yaml) as f:
        cfg = yaml.safe_load(f)
    env_cfg = copy.deepcopy(cfg.get("environment", {}))
    if scenario_yaml:
        with open(scenario_yaml) as f:
            env_cfg.update(copy.deepcopy(yaml.safe_load(f).get("environment", {})))
    if merge_top_level:
        if "curriculum" in cfg:
            env_cfg["curriculum"] = copy.deepcopy(cfg["curriculum"])
        if "physics_enhancements" in cfg:
            env_cfg["physics_enhancements"] = copy.deepcopy(cfg["physics_enhancements"])
    return env_cfg
