"""The numpy restatement of the HRL controller (oracle/hrl_controller.py) replayed against the golden vectors recorded
from the reference's own HierarchicalManager / OptionManager / rule selector (tests/golden/make_hrl_golden.py).
Everything is discrete or a clipped ratio: the bar is bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle.hrl_controller import Controller

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "hrl", "*.npz")))


def stacked(obs, did_reset, stack):
    """What the wrapper hands to the manager: the frame-stacked observation (zero history at episode start)."""
    if stack == 1:
        return obs
    T = len(obs)
    out = np.zeros((T, 26 * stack), np.float32)
    hist = [np.zeros(26, np.float32)] * (stack - 1)
    for t in range(T):
        out[t] = np.concatenate(hist[-(stack - 1):] + [obs[t]])
        hist.append(obs[t])
        if did_reset[t]:
            hist = [np.zeros(26, np.float32)] * (stack - 1)
    return out


def test_fixture_inventory():
    assert len(FIXTURES) >= 10


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_controller_matches_reference(path):
    fx = np.load(path)
    obs = stacked(fx["obs"], fx["did_reset"], int(fx["stack"]))
    c = Controller(1, decision_interval=int(fx["decision_interval"]), enable_forced=bool(fx["forced_enabled"]),
                   enable_hysteresis=bool(fx["hysteresis"]), enable_min_dwell=bool(fx["min_dwell"]))
    done_prev = np.zeros(1, bool)
    for t in range(len(obs)):
        r = c.step(obs[t:t + 1], done_prev)
        assert np.array_equal(r["abstract"][0], fx["abstract"][t]), (t, r["abstract"][0], fx["abstract"][t])
        got = (int(r["option"][0]), int(r["switched"][0]), int(r["reason"][0]), int(r["forced"][0]), int(r["choice"][0]),
               int(r["steps_in_option"][0]), int(r["total_steps"][0]))
        want = (int(fx["option"][t]), int(fx["switched"][t]), int(fx["reason"][t]), int(fx["forced"][t]), int(fx["choice"][t]),
                int(fx["steps_in_option"][t]), int(fx["total_steps"][t]))
        assert got == want, (t, got, want)
        assert r["env_distance"][0] == fx["env_distance"][t]
        done_prev = fx["did_reset"][t:t + 1]
