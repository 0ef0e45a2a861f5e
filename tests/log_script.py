"""One scripted logging session (two episodes), played once into the reference's `UnifiedLogger`
(tests/golden/make_log_golden.py, build container only) and once into `EpisodeLog` (tests/test_episode_log.py): the
calls inference.py:489-616 makes -- header, three entities per step, an event, footer, metrics -- with the value kinds
that reach the JSON layer: numpy arrays and scalars, bools, nested dicts (`info['radar_debug']`), None, NaN / inf."""
import numpy as np


class FakeTime:
    """Stands in for the `time` MODULE inside the logger under test: a clock that advances 0.25 s per reading."""

    def __init__(self):
        self.k = 0

    def time(self):
        self.k += 1
        return 1000.0 + 0.25 * self.k


def session():
    """Yields (method, kwargs).  `log_event` always carries an explicit timestamp (the reference's EpisodeEvent has one)."""
    rng = np.random.default_rng(7)
    for ep in range(2):
        yield "begin_episode", dict(episode_id=None if ep == 0 else "ep_custom", metadata=None if ep == 0 else
                                    {"seed": np.int64(3), "scenario": "medium", "volley_mode": np.bool_(True),
                                     "spawn": np.array([1.5, 2.5, 3.5], np.float32)})
        for t in range(120 if ep == 0 else 7):     # 360 state records in episode 0: crosses the 100-record flush three times
            pos = rng.standard_normal(3).astype(np.float32)
            yield "log_state", dict(entity_id="interceptor", state={"position": pos.tolist(), "fuel": np.float32(99.5 - t),
                                                                     "action": rng.uniform(-1, 1, 6).astype(np.float32)},
                                    timestamp=None)
            yield "log_state", dict(entity_id="missile", state={"position": (pos * 2).tolist()}, timestamp=None if t % 2 else 2000.0 + t)
            yield "log_state", dict(entity_id="radar", timestamp=None, state={
                "onboard": {"detected": bool(t % 3), "detection_reason": "detected" if t % 3 else "outside_beam",
                            "range_to_target": float(np.float32(1234.5 + t)), "beam_angle_deg": np.float64(12.25),
                            "forward_vector": [0.0, 0.0, 1.0]},
                "ground": {"enabled": True, "detected": np.bool_(t % 2 == 0), "quality": float("nan") if t == 5 else 0.5},
                "fusion": {"fusion_confidence": np.float32(0.75), "both_detected": False, "datalink": float("inf") if t == 6 else 1.0}})
        yield "log_event", dict(event_type="interception", source="interceptor", target="missile" if ep == 0 else None,
                                data={"distance": np.float32(3.5), "step": np.int32(119)} if ep == 0 else None,
                                timestamp=1900.0 + ep)
        yield "end_episode", dict(outcome="intercepted" if ep == 0 else "failed",
                                  metrics={"total_reward": np.float64(4123.5), "steps": 120 if ep == 0 else 7,
                                           "final_distance": np.float32(3.5), "fuel_used": 41.0, "volley_mode": False,
                                           "missiles_intercepted": None, "volley_size": None})
    yield "log_state", dict(entity_id="interceptor", state={"x": 1}, timestamp=None)     # no episode open: ignored
    yield "log_metrics", dict(metrics={"success_rate": np.float32(0.5), "episodes": np.int64(2)})
