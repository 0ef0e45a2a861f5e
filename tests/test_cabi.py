"""The C ABI library loads (ROCm runtime present, no GPU needed) and exports every symbol that
include/hlx.h and include/hlx_obs.h declare; argument validation works without touching a device."""
import ctypes as C
import os
import re

import pytest

from hlynr_intercept_amd import _lib
from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.scenarios import scenario_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    out = set()
    for header in ("hlx.h", "hlx_obs.h", "hlx_hrl.h"):
        txt = open(os.path.join(ROOT, "include", header)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(hlx_[a-z_0-9]+)\s*\(", txt))
    return sorted(out)


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert declared == sorted(_lib.SYMBOLS), (set(declared) ^ set(_lib.SYMBOLS))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.hlx_version().decode().startswith("hlx")
    assert lib.hlx_sizeof_config() == C.sizeof(_lib.HlxConfig)
    assert lib.hlx_sizeof_env_state() == C.sizeof(_lib.HlxEnvState)


def test_argument_validation_without_a_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.hlx_create(None, 16, 0, 0, 0, C.byref(h)) == -1 and b"null" in lib.hlx_last_error()
    cfg = _lib.make_hlx_config(resolve_config(scenario_config("medium", "base")))
    assert lib.hlx_create(C.byref(cfg), 0, 0, 0, 0, C.byref(h)) == -1 and b"n_envs" in lib.hlx_last_error()
    cfg.max_steps = 100000
    assert lib.hlx_create(C.byref(cfg), 16, 0, 0, 0, C.byref(h)) == -1 and b"max_steps" in lib.hlx_last_error()
    assert lib.hlx_step(None, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.hlx_destroy(None) == 0
    # observation pipeline (include/hlx_obs.h)
    p = C.c_void_p()
    oc = _lib.HlxObsConfig(n_envs=16, obs_dim=26, n_stack=99, device=0, norm_obs=1, norm_reward=0, training=1,
                           clip_obs=10.0, clip_reward=10.0, gamma=0.99, epsilon=1e-8)
    assert lib.hlx_obs_create(C.byref(oc), C.byref(p)) == -1 and b"n_stack" in lib.hlx_last_error()
    assert lib.hlx_obs_create(None, C.byref(p)) == -1
    assert lib.hlx_obs_push(None, None, None, None, None, None, None, None, None) == -1
    assert lib.hlx_obs_destroy(None) == 0 and lib.hlx_obs_feature_dim(None) == 0
    assert C.sizeof(_lib.HlxObsConfig) == 64
    # HRL controller (include/hlx_hrl.h)
    hc = _lib.HlxHrlConfig(n_envs=8, obs_dim=27, device=0, decision_interval=100, selector_mode=1)
    lib.hlx_hrl_default_thresholds(C.byref(hc))
    assert hc.h_lock_acquire == 0.75 and list(hc.min_dwell) == [50, 50, 30] and hc.close_range == 200.0
    assert lib.hlx_hrl_create(C.byref(hc), C.byref(p)) == -1 and b"obs_dim" in lib.hlx_last_error()
    assert lib.hlx_hrl_step(None, None, None, None, None, None, None, None, None) == -1 and lib.hlx_hrl_destroy(None) == 0
    assert lib.hlx_hrl_regroup(None, None, None, None, 0, None, 0, None) == -1 and lib.hlx_hrl_rows(None, None, None, None, None, None) == -1
    rb = (C.c_int64 * 2)(1024, 20)           # staging area of hlx_hrl_regroup: every bank's N rows, each region rounded up to 16 bytes
    assert lib.hlx_hrl_regroup_scratch_bytes(100, rb, 2) == 102400 + 2000 and lib.hlx_hrl_regroup_scratch_bytes(3, rb, 2) == 3072 + 64


def test_config_struct_carries_the_flags():
    rc = resolve_config(scenario_config("medium", "v2dr", {"observation_mode": "los_frame", "proximity_fuze_enabled": True}))
    cfg = _lib.make_hlx_config(rc)
    for bit in (_lib.F_ATMOSPHERE, _lib.F_MACH_DRAG, _lib.F_ENH_WIND, _lib.F_THRUST_LAG, _lib.F_DOMAIN_RAND,
                _lib.F_VALIDATION, _lib.F_EVASION, _lib.F_GROUND, _lib.F_OBS_LOS, _lib.F_PROX_FUZE,
                _lib.F_USE_CURRICULUM, _lib.F_RADAR_CURRICULUM):
        assert cfg.flags & bit
    assert not cfg.flags & (_lib.F_OBS_BODY | _lib.F_PRECISION | _lib.F_SPHERICAL)
    assert cfg.dt == 0.01 and cfg.onboard_delay == 3 and cfg.ground_delay == 5
    assert list(cfg.rc_beam) == [120.0, 60.0, 5000000.0, 8000000.0]


def test_vec_env_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hlynr_intercept_amd.vec_env import HlynrVecEnv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HlynrVecEnv(scenario_config("medium", "base"), num_envs=4)


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under hlynr_intercept_amd/ may import, load or link it, and the C ABI
    library must not contain its entry points (no CPU fallback hiding in the product)."""
    pkg = os.path.join(ROOT, "hlynr_intercept_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)
                assert "orc_step" not in txt and "orc_reset" not in txt, os.path.join(dirpath, f)
    lib = _lib.load()
    for sym in ("orc_step", "orc_reset", "orc_step_batch", "hlx_step_cpu", "hlx_reset_cpu"):
        assert not hasattr(lib, sym), sym
