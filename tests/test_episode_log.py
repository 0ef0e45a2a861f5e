"""Host-side episode sinks: the `info['radar_debug']` assembly against the reference's recorded
dicts, and the episode-file writer against the record layout of the reference's UnifiedLogger (logger.py:148-288).

No GPU here: `radar_debug()` is fed the reference's own recorded post-step states (tests/golden/<name>.npz, st_*)
plus the two detection facts it cannot derive (taken from the recorded dict itself), which pins the geometry, the
ground-radar reason ladder and the formatting; the kernel-side bits are covered by tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

from hlynr_intercept_amd.config import resolve_config
from hlynr_intercept_amd.episode_log import ONBOARD_REASONS, EpisodeLog, VecEpisodeRecorder, radar_debug
from tests.golden_util import compare_radar_debug, load_fixture, load_radar_fixture, radar_fixture_names

CODES = {v: k for k, v in ONBOARD_REASONS.items()}


def test_radar_fixtures_present():
    assert len(radar_fixture_names()) >= 8


@pytest.mark.parametrize("name", radar_fixture_names())
def test_radar_debug_assembly_matches_reference(name):
    fx, cols = load_fixture(name), load_radar_fixture(name)
    rc = resolve_config(fx["config"])
    sched = rc.radar_schedule(fx["global_step_or_none"])
    st_at = {int(t): j for j, t in enumerate(fx["st_index"])}
    bad, checked = [], 0
    for t in range(len(fx["action"])):
        if t not in st_at:
            continue
        j = st_at[t]
        bits = CODES.get(str(cols["onboard.detection_reason"][t]), 0)
        if str(cols["ground.detection_reason"][t]) in ("detected", "unknown"):
            bits |= 8
        flags = (32 if cols["onboard.detected"][t] else 0) | (64 if cols["ground.detected"][t] else 0)
        mine = radar_debug(rc, sched["beam_width"], fx["st_int_pos"][j], fx["st_mis_pos"][j], fx["st_int_quat"][j],
                           float(cols["ground.quality"][t]), bits, flags, float(cols["fusion.datalink_quality"][t]),
                           float(cols["fusion.fusion_confidence"][t]))
        bad += compare_radar_debug(mine, cols, t)
        checked += 1
    assert checked >= 40 and not bad, (checked, len(bad), bad[:5])


def test_radar_debug_is_json_serialisable_and_complete():
    rc = resolve_config({})
    d = radar_debug(rc, 120.0, [0, 0, 100], [500, 400, 900], [1, 0, 0, 0], 0.5, 3, 64, 0.9, 0.3)
    json.dumps(d)
    assert set(d) == {"onboard", "ground", "fusion"}
    assert len(d["onboard"]) == 11 and len(d["ground"]) == 10 and len(d["fusion"]) == 4    # core.py:650-683
    assert d["onboard"]["detection_reason"] == "poor_signal" and d["onboard"]["quality"] == 0.0


def _read(path):
    with open(path) as f:
        return [json.loads(line) for line in f]


def test_episode_log_record_layout(tmp_path):
    log = EpisodeLog(str(tmp_path), run_name="unit")
    log.begin_episode("ep_0000", {"seed": np.int64(3)})
    for k in range(150):   # crosses the 100-entry flush threshold of logger.py:203
        log.log_state("interceptor", {"position": np.arange(3, dtype=np.float32) + k, "fuel": np.float32(99.5)})
    log.log_event("intercept", "interceptor", "missile", {"distance": 1.5})
    log.end_episode("intercepted", {"total_reward": np.float32(12.5), "steps": 150})
    files = os.listdir(os.path.join(log.log_dir, "episodes"))
    assert files == ["ep_0000.jsonl"]
    rows = _read(os.path.join(log.log_dir, "episodes", files[0]))
    assert rows[0]["type"] == "header" and set(rows[0]) == {"type", "episode_id", "start_time", "metadata"}
    assert rows[0]["metadata"] == {"seed": 3}
    states = [r for r in rows if r["type"] == "state"]
    assert len(states) == 150 and set(states[0]) == {"type", "timestamp", "entity_id", "state"}
    assert states[7]["state"]["position"] == [7.0, 8.0, 9.0] and 0 <= states[0]["timestamp"] < 5.0   # relative time
    ev = [r for r in rows if r["type"] == "event"]
    assert len(ev) == 1 and set(ev[0]) == {"type", "timestamp", "event_type", "source", "target", "data"}
    foot = rows[-1]
    assert foot["type"] == "footer" and set(foot) == {"type", "episode_id", "end_time", "duration", "outcome", "metrics"}
    assert foot["outcome"] == "intercepted" and foot["metrics"]["total_reward"] == 12.5
    m = _read(log.metrics_file)
    assert len(m) == 1 and m[0]["episode"] == "ep_0000" and m[0]["steps"] == 150 and "timestamp" in m[0]
    log.log_state("interceptor", {"x": 1})      # no current episode: ignored (logger.py:178)
    log.end_episode("failed", {})
    assert len(_read(log.metrics_file)) == 1


class _FakeInfos:
    def __init__(self, rows):
        self.rows = rows

    def __getitem__(self, i):
        return self.rows[i]


def test_vec_recorder_writes_one_file_per_episode(tmp_path):
    rec = VecEpisodeRecorder(str(tmp_path), indices=(0, 2), run_name="t")
    n = 3
    for t in range(7):
        infos = _FakeInfos([dict(interceptor_pos=np.array([t, 0, 100.0], np.float32), missile_pos=np.array([50.0, 0, 100.0], np.float32),
                                 fuel_remaining=90.0 - t, distance=50.0 - t, fuel_used=10.0 + t, intercepted=(i == 2),
                                 radar_debug={"onboard": {"detected": True}}) for i in range(n)])
        dones = np.array([False, False, t in (2, 5)])
        rec.on_step(np.zeros((n, 6), np.float32), np.full(n, 0.5, np.float32), dones, infos)
    rec.close()
    assert [r["env_index"] for r in rec.results] == [2, 2]
    assert [r["steps"] for r in rec.results] == [3, 3] and all(r["outcome"] == "intercepted" for r in rec.results)
    assert rec.results[0]["min_distance"] == 48.0
    d2 = os.path.join(rec._logs[2].log_dir, "episodes")
    assert sorted(os.listdir(d2)) == ["ep_0000.jsonl", "ep_0001.jsonl", "ep_0002.jsonl"]
    rows = _read(os.path.join(d2, "ep_0000.jsonl"))
    assert [r.get("entity_id") for r in rows[1:4]] == ["interceptor", "missile", "radar"]      # inference.py:535-548
    assert rows[1]["state"]["action"] == [0.0] * 6 and rows[-1]["type"] == "footer"
    assert set(rows[-1]["metrics"]) == {"total_reward", "steps", "final_distance", "fuel_used", "volley_mode",
                                        "missiles_intercepted", "volley_size"}                  # inference.py:606-614
    rows0 = _read(os.path.join(rec._logs[0].log_dir, "episodes", "ep_0000.jsonl"))
    assert len([r for r in rows0 if r["type"] == "state"]) == 21 and rows0[-1]["type"] == "state"   # still running


def test_episode_files_equal_the_ones_the_references_logger_writes(tmp_path):
    """tests/golden/logs/*.jsonl were written by the reference's own `UnifiedLogger` (logger.py:148-288) from the scripted
    session of tests/log_script.py under a deterministic clock (tests/golden/make_log_golden.py).  `EpisodeLog`, fed the
    same session, must produce the same files BYTE FOR BYTE: record order, key order, flush boundaries, relative
    timestamps, numpy -> JSON conversion, NaN / inf -> null."""
    import hlynr_intercept_amd.episode_log as el
    from tests.log_script import FakeTime, session

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "logs")
    log = EpisodeLog(str(tmp_path), run_name="golden")
    real_time = el.time
    el.time = FakeTime()
    try:
        for method, kw in session():
            if method == "log_metrics":
                log.log_metrics(kw["metrics"])
            else:
                getattr(log, method)(**kw)
    finally:
        el.time = real_time
    mine = {"episodes_" + f: os.path.join(log.log_dir, "episodes", f) for f in os.listdir(os.path.join(log.log_dir, "episodes"))}
    mine["metrics.jsonl"] = log.metrics_file
    assert sorted(mine) == sorted(os.listdir(golden))
    for name, path in mine.items():
        with open(path) as a, open(os.path.join(golden, name)) as b:
            la, lb = a.read().splitlines(), b.read().splitlines()
        assert len(la) == len(lb), name
        for k, (x, y) in enumerate(zip(la, lb)):
            assert x == y, (name, k, x[:200], y[:200])
