"""ctypes binding of include/hlx.h (the C ABI of libhlx.so).

There is no fallback: if the HIP library is missing or fails to load, importing callers get a
RuntimeError.  Nothing in this package routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

f32, i32, u32, i64, u64, f64, u8p = C.c_float, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_double, C.c_void_p

OBS_DIM, ACT_DIM, STEP_SLOTS, RESET_SLOTS, RING_CAP, MAX_STEPS, MAX_VOLLEY = 26, 6, 32, 48, 11, 8191, 4
ABI_VERSION = 4      # include/hlx.h HLX_ABI_VERSION: the revision these ctypes mirrors were written for

# hlx_flags
F_ATMOSPHERE, F_MACH_DRAG, F_ENH_WIND, F_THRUST_LAG, F_DOMAIN_RAND, F_VALIDATION, F_EVASION, F_PRECISION = (
    1 << 0, 1 << 1, 1 << 2, 1 << 3, 1 << 4, 1 << 5, 1 << 6, 1 << 7)
F_PROX_FUZE, F_GROUND, F_SPHERICAL, F_TOWARD_MISSILE, F_OBS_BODY, F_OBS_LOS, F_USE_CURRICULUM, F_RADAR_CURRICULUM = (
    1 << 8, 1 << 9, 1 << 10, 1 << 11, 1 << 12, 1 << 13, 1 << 14, 1 << 15)
F_VOLLEY = 1 << 16
F_RADAR_DEBUG = 1 << 17


class HlxConfig(C.Structure):
    _fields_ = [
        ("flags", u32), ("max_steps", i32), ("onboard_delay", i32), ("ground_delay", i32),
        ("dt", f64), ("max_range", f64), ("max_velocity", f64),
        ("target_pos", f64 * 3), ("mis_pos_lo", f64 * 3), ("mis_pos_hi", f64 * 3),
        ("mis_radius", f64 * 2), ("mis_azimuth_deg", f64 * 2), ("mis_elevation_deg", f64 * 2), ("mis_speed", f64 * 2),
        ("int_pos_lo", f64 * 3), ("int_pos_hi", f64 * 3), ("int_vel_lo", f64 * 3), ("int_vel_hi", f64 * 3),
        ("int_speed", f64 * 2),
        ("subsonic_mach", f64), ("supersonic_mach", f64), ("transonic_peak_multiplier", f64),
        ("supersonic_multiplier", f64),
        ("base_wind", f64 * 3), ("wind_variability", f64), ("boundary_layer_height", f64),
        ("turbulence_intensity", f64), ("gust_scale", f64), ("thrust_tau", f64),
        ("dr_variations", f64 * 13), ("proximity_kill_radius", f64),
        ("radar_quality", f64), ("radar_range", f64), ("radar_beam_width", f64),
        ("ground_pos", f64 * 3), ("ground_max_range", f64), ("ground_min_elev", f64), ("ground_max_elev", f64),
        ("ground_range_accuracy", f64), ("ground_velocity_accuracy", f64), ("ground_base_quality", f64),
        ("max_datalink_range", f64), ("datalink_packet_loss", f64), ("weather_factor", f64),
        ("initial_radius", f64), ("final_radius", f64), ("curriculum_steps", f64),
        ("rc_beam", f64 * 4), ("rc_onboard", f64 * 4), ("rc_ground", f64 * 4), ("rc_noise", f64 * 4),
        ("volley_size", i32), ("pad1", i32),
    ]


class HlxInfoSoa(C.Structure):
    _fields_ = [("distance", C.c_void_p), ("min_distance", C.c_void_p), ("fuel", C.c_void_p), ("flags", C.c_void_p),
                ("episode_return", C.c_void_p), ("episode_length", C.c_void_p), ("missiles", C.c_void_p),
                ("interceptor_pos", C.c_void_p), ("missile_pos", C.c_void_p), ("steps", C.c_void_p),
                ("missile_min_distances", C.c_void_p), ("radar_debug", C.c_void_p), ("fuel_used", C.c_void_p),
                ("packed", C.c_void_p)]


class HlxEnvState(C.Structure):
    _fields_ = [
        ("int_pos", f32 * 3), ("int_vel", f32 * 3), ("int_quat", f32 * 4), ("fuel", f32),
        ("thrust_actual", f32 * 3), ("mis_pos", f32 * 3), ("mis_vel", f32 * 3),
        ("prev_distance", f32), ("min_distance", f32), ("last_distance", f32),
        ("steps", i32), ("worsening", i32), ("crossed", i32), ("kf_init", i32), ("kf_x_is64", i32), ("episode", i32),
        ("wind", f64 * 3), ("kf_x", f64 * 6), ("kf_P", f32 * 4),
        ("on_delay", i32), ("on_len", i32), ("on_ring", (f32 * 4) * RING_CAP),
        ("g_len", i32), ("g_ring", (f64 * 8) * RING_CAP),
        ("T0", f64), ("base_cd", f32), ("transonic_peak_m1", f32), ("cd_super", f32), ("ep_return", f32),
        ("v_pos", (f32 * 3) * MAX_VOLLEY), ("v_vel", (f32 * 3) * MAX_VOLLEY), ("v_min", f32 * MAX_VOLLEY),
        ("v_active", i32 * MAX_VOLLEY), ("prio", i32), ("n_intercepted", i32),
        ("fuel_used", f32), ("pad2", i32),
    ]


class HlxObsConfig(C.Structure):   # include/hlx_obs.h hlx_obs_config
    _fields_ = [("n_envs", i32), ("obs_dim", i32), ("n_stack", i32), ("device", i32), ("norm_obs", i32),
                ("norm_reward", i32), ("training", i32), ("pad0", i32), ("clip_obs", f64), ("clip_reward", f64),
                ("gamma", f64), ("epsilon", f64)]


class HlxHrlConfig(C.Structure):   # include/hlx_hrl.h hlx_hrl_config
    _fields_ = [("n_envs", i32), ("obs_dim", i32), ("device", i32), ("decision_interval", i32), ("selector_mode", i32),
                ("enable_forced", i32), ("enable_hysteresis", i32), ("enable_min_dwell", i32), ("default_option", i32),
                ("min_dwell", i32 * 3), ("lock_min", f64), ("lock_search", f64), ("close_range", f64),
                ("terminal_fuel_min", f64), ("miss_imminent", f64), ("fuel_critical", f64), ("h_lock_acquire", f64),
                ("h_lock_maintain", f64), ("h_terminal_enter", f64), ("h_terminal_exit", f64)]


# every symbol include/hlx.h, include/hlx_obs.h and include/hlx_hrl.h declare: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "hlx_create": (C.c_int, [C.POINTER(HlxConfig), i32, i32, u64, i64, C.POINTER(_P)]),
    "hlx_destroy": (C.c_int, [_P]),
    "hlx_reset": (C.c_int, [_P, _P, _P, _P]),
    "hlx_reset_info": (C.c_int, [_P, _P, _P, _P, _P]),
    "hlx_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(HlxInfoSoa), _P]),
    "hlx_rollout": (C.c_int, [_P, _P, i32, i32, _P, _P, _P, _P, _P]),
    "hlx_set_global_step": (C.c_int, [_P, i64]),
    "hlx_get_curriculum": (C.c_int, [_P, C.POINTER(f64 * 5)]),
    "hlx_set_noise": (C.c_int, [_P, _P, _P]),
    "hlx_fill_noise": (C.c_int, [_P, _P, _P, i32, _P]),
    "hlx_get_state": (C.c_int, [_P, _P]),
    "hlx_set_state": (C.c_int, [_P, _P]),
    "hlx_set_rollout_fused": (C.c_int, [_P, i32]),
    "hlx_set_rollout_terminal_obs": (C.c_int, [_P, _P]),
    "hlx_set_rollout_outputs": (C.c_int, [_P, _P, _P, C.POINTER(HlxInfoSoa)]),
    "hlx_set_done_counter": (C.c_int, [_P, _P]),
    "hlx_set_reset_epoch": (C.c_int, [_P, u32]),
    "hlx_get_reset_epoch": (u32, [_P]),
    "hlx_set_seed": (C.c_int, [_P, u64]),
    "hlx_selftest_math": (C.c_int, [i32, _P, f32, _P, i64, _P]),
    "hlx_set_episode_pool": (C.c_int, [_P, i32]),
    "hlx_get_episode_pool": (i32, [_P]),
    "hlx_get_episode_pool_misses": (C.c_int, [_P, C.POINTER(i64)]),
    "hlx_get_episode_pool_crowded": (C.c_int, [_P, C.POINTER(i64)]),
    "hlx_get_episode_pool_stats": (C.c_int, [_P, C.POINTER(i64 * 4)]),
    "hlx_set_load_schedule": (C.c_int, [_P, i32]),
    "hlx_get_load_schedule": (i32, [_P]),
    "hlx_profile": (C.c_int, [_P, i32]),
    "hlx_profile_read": (C.c_int, [_P, C.POINTER(f64), C.POINTER(i64)]),
    "hlx_num_envs": (i32, [_P]),
    "hlx_vec_steps": (i64, [_P]),
    "hlx_kernel_variant": (C.c_char_p, [_P]),
    "hlx_kernel_baked": (C.c_char_p, [_P]),
    "hlx_sizeof_config": (i32, []),
    "hlx_sizeof_env_state": (i32, []),
    "hlx_sizeof_info_soa": (i32, []),
    "hlx_abi_version": (i32, []),
    "hlx_hot_words_from_memory": (i32, []),
    "hlx_last_error": (C.c_char_p, []),
    "hlx_version": (C.c_char_p, []),
    # include/hlx_obs.h
    "hlx_obs_create": (C.c_int, [C.POINTER(HlxObsConfig), C.POINTER(_P)]),
    "hlx_obs_destroy": (C.c_int, [_P]),
    "hlx_obs_next_slot": (_P, [_P]),
    "hlx_obs_push_reset": (C.c_int, [_P, _P, _P]),
    "hlx_obs_push": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hlx_obs_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(HlxInfoSoa), _P, _P, _P, _P]),
    "hlx_obs_emit": (C.c_int, [_P, i32, _P, _P]),
    "hlx_obs_set_mode": (C.c_int, [_P, i32, i32, i32]),
    "hlx_obs_get_stats": (C.c_int, [_P, _P, _P, _P]),
    "hlx_obs_set_stats": (C.c_int, [_P, _P, _P, _P]),
    "hlx_obs_feature_dim": (i32, [_P]),
    # include/hlx_hrl.h
    "hlx_hrl_default_thresholds": (None, [C.POINTER(HlxHrlConfig)]),
    "hlx_hrl_create": (C.c_int, [C.POINTER(HlxHrlConfig), C.POINTER(_P)]),
    "hlx_hrl_destroy": (C.c_int, [_P]),
    "hlx_hrl_reset": (C.c_int, [_P, _P, _P]),
    "hlx_hrl_abstract": (C.c_int, [_P, _P, _P, _P]),
    "hlx_hrl_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hlx_hrl_get_state": (C.c_int, [_P, _P]),
    "hlx_hrl_set_state": (C.c_int, [_P, _P]),
    "hlx_hrl_regroup": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(i64), i32, _P, i64, _P]),
    "hlx_hrl_regroup_scratch_bytes": (i64, [i32, C.POINTER(i64), i32]),
    "hlx_hrl_rows": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "hlx_hrl_unrows": (C.c_int, [_P, _P, i32, _P, _P]),
    "hlx_hrl_group_counts": (C.c_int, [_P, C.POINTER(i32 * 4)]),
    "hlx_hrl_order": (_P, [_P]),
    "hlx_hrl_pos": (_P, [_P]),
    "hlx_hrl_bind_order": (C.c_int, [_P, _P, _P]),
}

_lib = None
_load_error = None


class HlxError(RuntimeError):
    pass


def load(build_if_missing: bool = True):
    """Load libhlx.so (building it in-tree with hipcc if absent).  Fails loudly; never falls back."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:      # one failed (re)build per process is enough: every later caller gets the same error at once
        raise _load_error
    try:
        return _load(build_if_missing)
    except Exception as exc:
        _load_error = exc
        raise


def _load(build_if_missing: bool):
    global _lib
    path = _build.LIB
    if build_if_missing and not os.environ.get("HLX_LIBRARY"):
        # Built in-tree by `__graft_entry__.build()` / `python -m hlynr_intercept_amd.build`.  A missing library, or one built
        # from other sources than those in csrc/ (content hash), is (re)built here -- under a file lock, so that processes
        # started together queue up instead of compiling into the same file (hlynr_intercept_amd/build.py).
        try:
            _build.build()
        except Exception as exc:  # pragma: no cover
            if not os.path.exists(path):
                raise RuntimeError(f"libhlx.so is missing and could not be built: {exc}") from exc
            # An existing library whose sources have moved on: refuse it where a rebuild was POSSIBLE and failed (a compile
            # error must not be papered over with stale code); use it, loudly, where this installation cannot rebuild at all
            # (no hipcc, or a read-only package directory: an installed copy next to sources it was not built from).
            import shutil
            import warnings
            can_rebuild = (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")) and os.access(os.path.dirname(path), os.W_OK)
            if can_rebuild:
                raise RuntimeError(f"libhlx.so is out of date with csrc/ and could not be rebuilt: {exc}") from exc
            warnings.warn(f"libhlx.so differs from the sources in csrc/ and cannot be rebuilt here ({exc}); using the existing library",
                          RuntimeWarning, stacklevel=3)
    if not os.path.exists(path):
        raise RuntimeError(f"HIP library not found at {path}: build it with `python -m hlynr_intercept_amd.build`")
    # One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64; libhlx.so names the
    # same SONAMEs, and whichever copy is mapped first serves every later user.  If libhlx.so came first (the system ROCm's
    # copy) and torch afterwards, torch would run half on its own libraries and half on the system's and report "no
    # ROCm-capable device".  So torch, when it is installed, is always mapped first; without torch the system copy is used.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - plain C-ABI use
        pass
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise RuntimeError(f"cannot load {path} (is the ROCm runtime, libamdhip64, present?): {exc}") from exc
    # ABI guard first -- also on the cannot-rebuild path above, and for an HLX_LIBRARY override: a library from another
    # revision of include/hlx.h (a struct that grew, an entry point that changed meaning) is refused, not half-used
    try:
        abi = lib.hlx_abi_version
    except AttributeError as exc:
        raise RuntimeError(f"{path} predates hlx_abi_version(): it was built from another revision of include/hlx.h; rebuild it "
                           "(python -m hlynr_intercept_amd.build)") from exc
    abi.restype, abi.argtypes = i32, []
    if abi() != ABI_VERSION:
        raise RuntimeError(f"{path} implements ABI revision {abi()} of include/hlx.h, this binding revision {ABI_VERSION}: rebuild it "
                           "(python -m hlynr_intercept_amd.build)")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError = symbol missing = broken build
        fn.restype, fn.argtypes = res, args
    for what, c_size, py_size in (("hlx_config", lib.hlx_sizeof_config(), C.sizeof(HlxConfig)),
                                  ("hlx_env_state", lib.hlx_sizeof_env_state(), C.sizeof(HlxEnvState)),
                                  ("hlx_info_soa", lib.hlx_sizeof_info_soa(), C.sizeof(HlxInfoSoa))):
        if c_size != py_size:
            raise RuntimeError(f"{what} layout mismatch: C {c_size} vs ctypes {py_size} bytes")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise HlxError(f"hlx error {rc}: {load().hlx_last_error().decode()}")


def make_hlx_config(rc) -> HlxConfig:
    """ResolvedConfig -> hlx_config (float32 parameters, float64 curriculum schedules)."""
    c = HlxConfig()
    flags = 0
    for cond, bit in ((rc.atmosphere, F_ATMOSPHERE), (rc.mach_drag, F_MACH_DRAG), (rc.enhanced_wind, F_ENH_WIND),
                      (rc.thrust_lag, F_THRUST_LAG), (rc.domain_randomization, F_DOMAIN_RAND),
                      (rc.validation, F_VALIDATION), (rc.evasion, F_EVASION), (rc.precision_mode, F_PRECISION),
                      (rc.proximity_fuze, F_PROX_FUZE), (rc.ground_enabled, F_GROUND),
                      (rc.mis_spawn_spherical, F_SPHERICAL), (rc.int_vel_toward_missile, F_TOWARD_MISSILE),
                      (rc.obs_mode == 1, F_OBS_BODY), (rc.obs_mode == 2, F_OBS_LOS),
                      (rc.use_curriculum, F_USE_CURRICULUM), (rc.radar_curriculum.active, F_RADAR_CURRICULUM),
                      (rc.volley_mode, F_VOLLEY)):
        if cond:
            flags |= bit
    c.flags = flags
    c.volley_size = int(rc.volley_size) if rc.volley_mode else 1
    simple = ("max_steps", "dt", "max_range", "max_velocity", "subsonic_mach", "supersonic_mach",
              "transonic_peak_multiplier", "supersonic_multiplier", "wind_variability", "boundary_layer_height",
              "turbulence_intensity", "gust_scale", "thrust_tau", "proximity_kill_radius", "radar_quality",
              "radar_range", "radar_beam_width", "onboard_delay", "ground_max_range", "ground_min_elev",
              "ground_max_elev", "ground_range_accuracy", "ground_velocity_accuracy", "ground_base_quality",
              "max_datalink_range", "datalink_packet_loss", "weather_factor", "ground_delay", "initial_radius",
              "final_radius", "curriculum_steps")
    for k in simple:
        setattr(c, k, getattr(rc, k))
    for k in ("target_pos", "mis_pos_lo", "mis_pos_hi", "mis_radius", "mis_azimuth_deg", "mis_elevation_deg",
              "mis_speed", "int_pos_lo", "int_pos_hi", "int_vel_lo", "int_vel_hi", "int_speed", "base_wind",
              "dr_variations", "ground_pos"):
        arr = getattr(c, k)
        for j, x in enumerate(getattr(rc, k)):
            arr[j] = float(x)
    cur = rc.radar_curriculum
    for name, vals in (("rc_beam", (cur.initial_beam_width, cur.final_beam_width, cur.beam_width_transition_start,
                                    cur.beam_width_transition_end)),
                       ("rc_onboard", (cur.initial_detection_reliability, cur.final_detection_reliability,
                                       cur.reliability_transition_start, cur.reliability_transition_end)),
                       ("rc_ground", (cur.initial_ground_reliability, cur.final_ground_reliability,
                                      cur.ground_reliability_transition_start, cur.ground_reliability_transition_end)),
                       ("rc_noise", (cur.initial_noise_level, cur.final_noise_level, cur.noise_transition_start,
                                     cur.noise_transition_end))):
        arr = getattr(c, name)
        for j, x in enumerate(vals):
            arr[j] = float(x)
    return c
