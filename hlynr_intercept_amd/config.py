"""Config resolver: reference env-config dict -> flat parameter set for the batched step.

The reference reads its YAML with `dict.get(..., default)` everywhere and never validates it, so a
number of scenario keys are dead and several sections fall back to constructor defaults
(SURVEY.md §3.1).  This module reproduces the *effective* values, citing the line each default
comes from, so that the HIP step (and the test oracle) run the physics the reference actually runs.

All citations are to /root/reference/rl_system/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Any, Dict, List, Optional

OBS_WORLD, OBS_BODY, OBS_LOS = 0, 1, 2
MAX_VOLLEY = 4   # include/hlx.h HLX_MAX_VOLLEY
_OBS_MODES = {"world_frame": OBS_WORLD, "body_frame": OBS_BODY, "los_frame": OBS_LOS}


class ConfigError(ValueError):
    """Raised for configurations the batched step does not implement (never silently ignored)."""


def _norm3(v) -> float:
    # np.linalg.norm of a python list -> float64 sqrt of the sum of squares (environment.py:380-381,449-450)
    return math.sqrt(float(v[0]) * float(v[0]) + float(v[1]) * float(v[1]) + float(v[2]) * float(v[2]))


@dataclass
class RadarCurriculum:
    """environment.py:274-351 — staggered piecewise-linear radar difficulty vs global training step."""
    active: bool = False
    initial_beam_width: float = 120.0
    final_beam_width: float = 60.0
    beam_width_transition_start: float = 3000000
    beam_width_transition_end: float = 5000000
    initial_detection_reliability: float = 1.0
    final_detection_reliability: float = 0.75
    reliability_transition_start: float = 4500000
    reliability_transition_end: float = 6000000
    initial_ground_reliability: float = 1.0
    final_ground_reliability: float = 0.85
    ground_reliability_transition_start: float = 4500000
    ground_reliability_transition_end: float = 6000000
    initial_noise_level: float = 0.0
    final_noise_level: float = 0.05
    noise_transition_start: float = 6000000
    noise_transition_end: float = 7000000


@dataclass
class ResolvedConfig:
    # --- timing / normalisation (environment.py:25-28)
    dt: float = 0.01
    max_steps: int = 1000
    max_range: float = 10000.0
    max_velocity: float = 1000.0
    # --- geometry (environment.py:31-39)
    target_pos: List[float] = field(default_factory=lambda: [900.0, 900.0, 5.0])
    mis_spawn_spherical: bool = False
    mis_pos_lo: List[float] = field(default_factory=lambda: [-500.0, -500.0, 200.0])
    mis_pos_hi: List[float] = field(default_factory=lambda: [500.0, 500.0, 500.0])
    mis_radius: List[float] = field(default_factory=lambda: [800.0, 1500.0])
    mis_azimuth_deg: List[float] = field(default_factory=lambda: [0.0, 360.0])
    mis_elevation_deg: List[float] = field(default_factory=lambda: [10.0, 60.0])
    mis_speed: List[float] = field(default_factory=lambda: [0.0, 0.0])
    int_pos_lo: List[float] = field(default_factory=lambda: [400.0, 400.0, 50.0])
    int_pos_hi: List[float] = field(default_factory=lambda: [600.0, 600.0, 200.0])
    int_vel_lo: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    int_vel_hi: List[float] = field(default_factory=lambda: [50.0, 50.0, 20.0])
    int_vel_toward_missile: bool = False
    int_speed: List[float] = field(default_factory=lambda: [0.0, 0.0])
    # --- physics switches (environment.py:52-106)
    atmosphere: bool = True
    mach_drag: bool = True
    enhanced_wind: bool = True
    thrust_lag: bool = True
    domain_randomization: bool = False
    validation: bool = True
    evasion: bool = False
    subsonic_mach: float = 0.8
    supersonic_mach: float = 1.2
    transonic_peak_multiplier: float = 3.0
    supersonic_multiplier: float = 2.5
    base_wind: List[float] = field(default_factory=lambda: [5.0, 0.0, 0.0])
    wind_variability: float = 0.1
    boundary_layer_height: float = 1000.0
    turbulence_intensity: float = 0.1
    gust_scale: float = 5.0
    thrust_tau: float = 0.1
    # domain randomisation: only the variations that reach the path (physics_randomizer.py:243-297)
    dr_variations: List[float] = field(default_factory=lambda: [0.1, 0.05 * 20.0, 0.2, 0.15, 0.5, 0.3, 0.1, 0.3, 0.2,
                                                                 0.3, 0.4, 0.1, 0.1])
    # --- curriculum / termination (environment.py:111-129)
    use_curriculum: bool = True
    initial_radius: float = 200.0
    final_radius: float = 20.0
    curriculum_steps: float = 5000000
    precision_mode: bool = False
    proximity_fuze: bool = False
    proximity_kill_radius: float = 20.0
    radar_curriculum: RadarCurriculum = field(default_factory=RadarCurriculum)
    # --- sensors (environment.py:136-179, core.py:258-335)
    radar_quality: float = 1.0
    radar_range: float = 5000.0
    radar_beam_width: float = 60.0
    onboard_delay: int = 0            # samples; 0 = no delay ring (core.py:292-293)
    ground_enabled: bool = True       # a GroundRadarStation object exists (core.py:296-320)
    ground_enabled_flag: bool = True  # `ground_radar_enabled` itself (core.py:296): True with an empty section too
    ground_pos: List[float] = field(default_factory=lambda: [0.0, 0.0, 100.0])
    ground_max_range: float = 20000.0
    ground_min_elev: float = math.radians(5.0)
    ground_max_elev: float = math.radians(85.0)
    ground_range_accuracy: float = 10.0
    ground_velocity_accuracy: float = 2.0
    ground_base_quality: float = 0.95
    max_datalink_range: float = 50000.0
    datalink_packet_loss: float = 0.05
    ground_delay: int = 5             # samples; 0 = no delay ring
    weather_factor: float = 1.0       # environment.py:149 (never varied)
    obs_mode: int = OBS_WORLD
    volley_mode: bool = False
    volley_size: int = 1

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    # ---- curriculum scalars, evaluated host-side once per vec step (O(1)) ----------
    def intercept_radius(self, global_step: int) -> float:
        """environment.py:223-234."""
        if not self.use_curriculum:
            return self.final_radius
        progress = min(1.0, global_step / self.curriculum_steps)
        return self.initial_radius * (1.0 - progress) + self.final_radius * progress

    def radar_schedule(self, global_step: Optional[int]) -> Dict[str, float]:
        """Beam width / reliabilities in force at `global_step`.

        `None` = the constructor state, i.e. before any `set_training_step_count` call
        (environment.py:153-185); an int = after `_update_radar_curriculum` (environment.py:274-351).
        """
        rc = self.radar_curriculum
        if not rc.active:
            return dict(beam_width=self.radar_beam_width, onboard_reliability=1.0, ground_reliability=1.0,
                        noise_level=0.05)
        if global_step is None:
            return dict(beam_width=rc.initial_beam_width, onboard_reliability=rc.initial_detection_reliability,
                        ground_reliability=rc.initial_ground_reliability, noise_level=rc.initial_noise_level)

        def lerp(a, b, s0, s1):
            if global_step < s0:
                return a
            if global_step >= s1:
                return b
            p = (global_step - s0) / (s1 - s0)
            return a * (1.0 - p) + b * p

        return dict(
            beam_width=lerp(rc.initial_beam_width, rc.final_beam_width, rc.beam_width_transition_start,
                            rc.beam_width_transition_end),
            onboard_reliability=lerp(rc.initial_detection_reliability, rc.final_detection_reliability,
                                     rc.reliability_transition_start, rc.reliability_transition_end),
            ground_reliability=lerp(rc.initial_ground_reliability, rc.final_ground_reliability,
                                    rc.ground_reliability_transition_start, rc.ground_reliability_transition_end),
            noise_level=lerp(rc.initial_noise_level, rc.final_noise_level, rc.noise_transition_start,
                             rc.noise_transition_end),
        )


def _delay_samples(ms: float, dt: float) -> int:
    # core.py:292,316 : int(ms / (dt*1000)) if ms > 0 else 0 ; SensorDelayBuffer clamps to >= 1 (core.py:164)
    if ms > 0:
        n = int(ms / (dt * 1000.0))
        return max(1, n) if n > 0 else 0
    return 0


def resolve_config(config: Optional[Dict[str, Any]] = None) -> ResolvedConfig:
    """Mirror of `InterceptEnvironment.__init__` (environment.py:20-221) + `Radar26DObservation.__init__`
    (core.py:258-335): same keys, same defaults, same dead keys."""
    config = config or {}
    rc = ResolvedConfig()
    rc.dt = config.get("dt", 0.01)
    rc.max_steps = int(config.get("max_steps", 1000))
    rc.max_range = config.get("max_range", 10000.0)
    rc.max_velocity = config.get("max_velocity", 1000.0)

    ms = config.get("missile_spawn", {"position": [[-500, -500, 200], [500, 500, 500]],
                                      "velocity": [[50, 50, -20], [150, 150, -50]]})
    isp = config.get("interceptor_spawn", {"position": [[400, 400, 50], [600, 600, 200]],
                                           "velocity": [[0, 0, 0], [50, 50, 20]]})
    rc.target_pos = [float(x) for x in config.get("target_position", [900, 900, 5])]

    # volley mode (environment.py:42-43): K missiles per episode, spawned from the same ranges
    rc.volley_mode = bool(config.get("volley_mode", False))
    rc.volley_size = int(config.get("volley_size", 1))
    if rc.volley_mode and not 1 <= rc.volley_size <= MAX_VOLLEY:
        raise ConfigError(f"volley_size must be 1..{MAX_VOLLEY} (got {rc.volley_size}): the state arena holds "
                          f"at most {MAX_VOLLEY} missiles per environment")

    # missile spawn (environment.py:376-415)
    rc.mis_pos_lo = [float(x) for x in ms["position"][0]]
    rc.mis_pos_hi = [float(x) for x in ms["position"][1]]
    vlo, vhi = ms["velocity"]
    rc.mis_spawn_spherical = ms.get("position_mode", "box") == "spherical"
    rc.mis_radius = [float(ms.get("radius_min", 800.0)), float(ms.get("radius_max", 1500.0))]
    rc.mis_azimuth_deg = [float(x) for x in ms.get("azimuth_range", [0, 360])]
    rc.mis_elevation_deg = [float(x) for x in ms.get("elevation_range", [10, 60])]
    rc.mis_speed = [float(ms.get("speed_min", _norm3(vlo))), float(ms.get("speed_max", _norm3(vhi)))]
    # interceptor spawn (environment.py:442-467)
    rc.int_pos_lo = [float(x) for x in isp["position"][0]]
    rc.int_pos_hi = [float(x) for x in isp["position"][1]]
    rc.int_vel_lo = [float(x) for x in isp["velocity"][0]]
    rc.int_vel_hi = [float(x) for x in isp["velocity"][1]]
    rc.int_vel_toward_missile = isp.get("velocity_mode", "box") == "toward_missile"
    rc.int_speed = [float(isp.get("speed_min", _norm3(rc.int_vel_lo))),
                    float(isp.get("speed_max", _norm3(rc.int_vel_hi)))]

    # physics (environment.py:52-106); `gravity`, `drag_coefficient`, `air_density` keys are dead (:47-49)
    pc = config.get("physics_enhancements", {})
    enabled = pc.get("enabled", True)
    rc.atmosphere = bool(enabled and pc.get("atmospheric_model", {}).get("enabled", True))
    rc.mach_drag = bool(enabled and pc.get("mach_effects", {}).get("enabled", True))
    mc = pc.get("mach_effects", {})
    rc.subsonic_mach = mc.get("subsonic_mach", 0.8)
    rc.supersonic_mach = mc.get("supersonic_mach", 1.2)
    rc.transonic_peak_multiplier = mc.get("transonic_peak_multiplier", 3.0)
    rc.supersonic_multiplier = mc.get("supersonic_multiplier", 2.5)
    wc = config.get("wind", {})
    rc.base_wind = [float(x) for x in wc.get("velocity", [5.0, 0.0, 0.0])]
    rc.wind_variability = wc.get("variability", 0.1)
    rc.enhanced_wind = bool(enabled and pc.get("enhanced_wind", {}).get("enabled", True))
    ew = pc.get("enhanced_wind", {})
    rc.boundary_layer_height = ew.get("boundary_layer_height", 1000.0)
    rc.turbulence_intensity = ew.get("turbulence_intensity", 0.1)
    rc.gust_scale = ew.get("max_gust_speed", 5.0)
    rc.thrust_lag = bool(enabled and pc.get("thrust_dynamics", {}).get("enabled", True))
    rc.thrust_tau = pc.get("thrust_dynamics", {}).get("response_time_constant", 0.1)
    drc = pc.get("domain_randomization", {})
    rc.domain_randomization = bool(enabled and drc.get("enabled", False))
    if rc.domain_randomization:
        # physics_randomizer.py:19-41 defaults, :108-117 overrides; order = draw order (:165-214)
        p = dict(air_density=0.1, temperature=0.05, drag=0.2, mach=0.15, sensor_delay=0.5, radar_noise=0.3,
                 radar_quality=0.1, thrust=0.3, fuel=0.2, wind=0.3, turbulence=0.4, mass=0.1)
        if "drag_coefficient_variation" in drc:
            p["drag"] = drc["drag_coefficient_variation"]
        if "air_density_variation" in drc:
            p["air_density"] = drc["air_density_variation"]
        if "sensor_delay_variation" in drc:
            p["sensor_delay"] = drc["sensor_delay_variation"]
        if "thrust_response_variation" in drc:
            p["thrust"] = drc["thrust_response_variation"]
        if "wind_variation" in drc:
            p["wind"] = drc["wind_variation"]
        rc.dr_variations = [p["air_density"], p["temperature"] * 20.0, p["drag"], p["mach"], p["sensor_delay"],
                            p["radar_noise"], p["radar_quality"], p["thrust"], p["fuel"], p["wind"],
                            p["turbulence"], p["mass"], p["mass"]]
    rc.validation = bool(pc.get("performance", {}).get("enable_physics_validation", True))
    rc.evasion = bool(config.get("missile_evasion", False))

    # curriculum (environment.py:111-133)
    cc = config.get("curriculum", {})
    rc.use_curriculum = bool(cc.get("enabled", True))
    rc.initial_radius = cc.get("initial_radius", 200.0)
    rc.final_radius = cc.get("final_radius", 20.0)
    rc.curriculum_steps = cc.get("curriculum_steps", 5000000)
    rc.precision_mode = bool(cc.get("precision_mode", False))
    rc.proximity_fuze = bool(config.get("proximity_fuze_enabled", False))
    rc.proximity_kill_radius = config.get("proximity_kill_radius", 20.0)
    rcc = cc.get("radar_curriculum", {})
    use_rc = rcc.get("enabled", True)
    cur = RadarCurriculum(active=bool(use_rc and rcc))   # environment.py:154,182,285: dict must be non-empty
    if cur.active:
        for k in asdict(cur):
            if k != "active" and k in rcc:
                setattr(cur, k, rcc[k])
    rc.radar_curriculum = cur

    # radar: read from config['radar'], NOT from the scenario's flat keys (environment.py:136-138,153,171-172)
    rd = config.get("radar", {})
    rc.radar_quality = rd.get("radar_quality", 1.0)
    rc.radar_range = rd.get("radar_range", 5000.0)
    rc.radar_beam_width = rd.get("radar_beam_width", 60.0)

    sensor_delay_ms = 0.0
    if enabled and pc.get("sensor_delays", {}).get("enabled", True):
        sensor_delay_ms = pc.get("sensor_delays", {}).get("radar_delay_ms", 30.0)
    rc.onboard_delay = _delay_samples(sensor_delay_ms, rc.dt)

    gr = config.get("ground_radar", {})
    ground_enabled_flag = gr.get("enabled", True) if gr else True
    rc.ground_enabled = bool(ground_enabled_flag and gr)   # core.py:296-320: empty dict -> no station object
    rc.ground_enabled_flag = bool(ground_enabled_flag)
    if rc.ground_enabled:
        rc.ground_pos = [float(x) for x in gr.get("position", [0, 0, 100])]
        rc.ground_max_range = gr.get("max_range", 20000.0)
        rc.ground_min_elev = math.radians(gr.get("min_elevation_angle", 5.0))
        rc.ground_max_elev = math.radians(gr.get("max_elevation_angle", 85.0))
        rc.ground_range_accuracy = gr.get("range_accuracy", 10.0)
        rc.ground_velocity_accuracy = gr.get("velocity_accuracy", 2.0)
        rc.ground_base_quality = gr.get("base_quality", 0.95)
        rc.max_datalink_range = gr.get("max_datalink_range", 50000.0)
        rc.datalink_packet_loss = gr.get("datalink_packet_loss", 0.05)
        rc.ground_delay = _delay_samples(gr.get("ground_sensor_delay_ms", 50.0), rc.dt)
    else:
        rc.ground_delay = 0

    mode = config.get("observation_mode", "world_frame")
    if mode not in _OBS_MODES:
        mode = "world_frame"   # core.py:870-882: anything else falls through to the world-frame branch
    if config.get("rotation_invariant", False) and mode == "world_frame":
        mode = "body_frame"    # environment.py:164-166
    rc.obs_mode = _OBS_MODES[mode]
    return rc
