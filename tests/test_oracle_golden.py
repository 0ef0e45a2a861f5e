"""The CPU oracle (oracle/hlx_oracle.c) replayed against every golden fixture captured from the
reference (tests/golden/make_golden.py).  This is what pins the oracle: free-running replay (the
oracle's state is never re-synchronised to the reference's) over up to 2000 steps per case,
including auto-resets, all observation modes, reward modes and forced edge cases.

Tolerances: rewards, distances, flags, info['fuel_used'], the integrated physics state and the Kalman
covariance come out bit-identical (the filter's float32 matrix products follow OpenBLAS's sgemm order:
one fused multiply-add chain per element, oracle/hlx_oracle.c kf_predict); what remains is a last-bit
difference of the spawn quaternion's float64 arccos / sin / cos chain now and then and of SVML float32
transcendentals in the observation, hence the small non-zero bounds below.  All are well inside the
1e-5 relative bar of BASELINE.json.
"""
import pytest

from tests.golden_util import fixture_names, load_fixture, replay_oracle

NAMES = fixture_names()


def test_fixture_inventory():
    assert len(NAMES) >= 54
    for must in ("medium_base_random", "medium_v2_random", "medium_v2dr_short_eps", "eval360_los_fuze_pursuit",
                 "medium_base_precision_pursuit", "edge_fuel_out", "edge_crash", "edge_early_termination",
                 "edge_blind_kf_uninit", "edge_mach_sweep", "medium_v2_body_random", "medium_v2_los_random",
                 # round 2: config.yaml physics (the `config` / `config-volley` kernel variants), ISA layers above 11 / 20 km,
                 # the ground-radar reasons 'out_of_range' / 'above_coverage'
                 "medium_config_random", "hard_config_random", "medium_config_pursuit", "volley3_medium_config_pursuit",
                 "edge_isa_11km_v2", "edge_isa_20km_v2", "edge_isa_11km_config", "edge_isa_20km_config",
                 "edge_ground_out_of_range", "edge_ground_above_coverage"):
        assert must in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    fx = load_fixture(name)
    assert str(fx["numpy_version"]).startswith("2."), "fixtures must come from a NEP-50 numpy"
    r = replay_oracle(fx)
    assert r["flag_mismatch"] == [], r["flag_mismatch"][:3]
    assert r["int_mismatch"] == [], r["int_mismatch"][:3]
    assert r["structure_violations"] == 0       # KF covariance keeps its 3 x (2x2) block structure
    assert r["max_obs"] <= 2e-6, r["max_obs"]            # absolute, obs are O(1); observed 1.7e-6
    assert r["reset_obs"] <= 1e-6, r["reset_obs"]
    assert r["max_reward"] <= 1e-6, r["max_reward"]      # relative to max(1,|r|)
    assert r["max_distance"] <= 1e-6, r["max_distance"]
    assert r["fuel_used_bits_differ"] == 0               # info['fuel_used']: the float32 running sum, bit for bit
    for k, v in r["state"].items():
        # the filter's covariance is bit-identical to the reference's since its sgemm accumulation order (one FMA chain per
        # element) is restated (round 3; 2e-5 before), and its state follows to a float32 ulp
        tol = {"kf_x": 1e-7, "kf_P": 0.0}.get(k, 1e-6)
        assert v <= tol, (k, v)
