"""Episode / metrics sinks for the vectorised environment.

Two things the reference's offline evaluation (rl_system/inference.py:489-616) expects from the step path:

* `info['radar_debug']` (environment.py:842, assembled in core.py:650-683): `radar_debug()` rebuilds that dict on
  the HOST for one environment from what the step kernel exports (hlx_info_soa: positions, and the `radar_debug`
  planes - quaternion, delayed ground quality, the delayed onboard detection reason and the current ground
  detection bit).  Pure formatting + geometry on a handful of floats, restated from the formulas at the lines cited
  below; nothing here is on the hot path and nothing runs unless an info dict is actually built.
* the episode files of `UnifiedLogger` (rl_system/logger.py:148-273): `EpisodeLog` writes the same JSONL records
  (header / state / event / footer + metrics.jsonl), `VecEpisodeRecorder` drives one per watched environment of a
  `HlynrVecEnv` with exactly the entities inference.py:535-548 logs per step and the footer of inference.py:606-614.
"""
from __future__ import annotations

import json
import os
import time
from datetime import datetime
from typing import Any, Dict, Iterable, Optional

import numpy as np

ONBOARD_REASONS = {1: "out_of_range", 2: "outside_beam", 3: "poor_signal", 4: "sensor_delay_initialization"}
_f32 = np.float32


def _forward_vector(q):
    """core.py:1143-1152 on a float32 quaternion (w, x, y, z)."""
    w, x, y, z = (_f32(v) for v in q)
    f = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], dtype=np.float32)
    return f / (np.linalg.norm(f) + 1e-6)


def radar_debug(rc, beam_width: float, int_pos, mis_pos, quat, ground_quality: float, bits: int, flags: int,
                datalink: float, fusion: float) -> Dict[str, Any]:
    """The reference's `info['radar_debug']` for one environment.

    rc: ResolvedConfig; beam_width: the curriculum's current `radar_beam_width` (degrees);
    int_pos / mis_pos / quat: float32 post-step state of the interceptor and the (priority) missile;
    ground_quality, bits: planes 4 and 5 of hlx_info_soa.radar_debug; flags: hlx_info_soa.flags (bit5 delayed onboard
    detection, bit6 ground detection as reported); datalink / fusion: observation entries 24 and 25 of that step."""
    int_pos = np.asarray(int_pos, np.float32)
    mis_pos = np.asarray(mis_pos, np.float32)
    on_det, g_det = bool(flags & 32), bool(flags & 64)
    # ---- onboard (core.py:531-566, 650-663): geometry is the CURRENT one, detection / reason are the delayed sample's
    rel = mis_pos - int_pos
    rng = np.linalg.norm(rel)
    fwd = _forward_vector(quat)
    beam_angle = np.arccos(np.clip(np.dot(fwd, rel / (rng + 1e-6)), -1, 1))
    half_beam = np.radians(beam_width / 2.0)
    code = int(bits) & 7
    onboard = {
        "position": int_pos.tolist(),
        "forward_vector": fwd.tolist(),
        "beam_width_deg": float(beam_width),
        "beam_angle_to_target_deg": float(np.degrees(beam_angle)),
        "half_beam_width_deg": float(np.degrees(half_beam)),
        "in_beam": bool(beam_angle <= half_beam),
        "range_to_target": float(rng),
        "max_range": float(rc.radar_range),
        "detected": on_det,
        "detection_reason": ONBOARD_REASONS.get(code, "detected" if on_det else "unknown"),
        "quality": float(rc.radar_quality) if on_det else 0.0,
    }
    # ---- ground (core.py:368-438, 664-675): reason / range / elevation belong to THIS step's detection attempt
    g_range = g_elev = 0.0
    if not (rc.ground_enabled_flag and rc.ground_enabled):
        reason = "ground_radar_disabled"
    else:
        g2m = mis_pos - np.asarray(rc.ground_pos, np.float32)
        r = np.linalg.norm(g2m)
        if r > rc.ground_max_range:
            reason = "out_of_range"
        else:
            elevation = 0.0
            reason = None
            if r > 1e-6:
                elevation = np.arcsin(np.clip(g2m[2] / r, -1.0, 1.0))
                if elevation < rc.ground_min_elev:
                    reason = "below_horizon"
                elif elevation > rc.ground_max_elev:
                    reason = "above_coverage"
            if reason is None and mis_pos[2] < 50.0:
                reason = "terrain_masking"
            if reason is None and not (int(bits) & 8):
                reason = "weak_return"
            if reason is None:
                reason = "detected" if g_det else "unknown"   # detected now, delay line still filling (core.py:618-622)
            g_range, g_elev = float(r), float(np.degrees(elevation))
    station = rc.ground_enabled
    ground = {
        "position": [float(x) for x in np.asarray(rc.ground_pos, np.float32)] if station else [0, 0, 0],
        "enabled": bool(rc.ground_enabled_flag),
        "max_range": float(rc.ground_max_range) if station else 0.0,
        "min_elevation_deg": float(np.degrees(rc.ground_min_elev)) if station else 0.0,
        "max_elevation_deg": float(np.degrees(rc.ground_max_elev)) if station else 0.0,
        "range_to_target": g_range,
        "elevation_deg": g_elev,
        "detected": g_det,
        "detection_reason": reason,
        "quality": float(ground_quality),
    }
    return {
        "onboard": onboard,
        "ground": ground,
        "fusion": {"datalink_quality": float(datalink), "fusion_confidence": float(fusion),
                   "both_detected": on_det and g_det, "any_detected": on_det or g_det},
    }


def _plain(obj):
    """JSON-ready copy, value for value what logger.py:19-57 produces: containers recursively, numpy arrays as lists,
    numpy scalars as their Python value, a Python float NaN / inf as null (numpy-scalar NaNs stay NaN there too)."""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, float) and (obj != obj or obj in (float("inf"), float("-inf"))):
        return None
    return obj


class EpisodeLog:
    """Writer of the reference's run directory layout (logger.py:94-123): `<log_dir>/<run_id>/episodes/<id>.jsonl`
    with header / state / event / footer records (logger.py:148-262) and `metrics.jsonl` (logger.py:275-288)."""

    def __init__(self, log_dir: str = "logs", run_name: Optional[str] = None):
        stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        self.run_id = f"{run_name}_{stamp}" if run_name else f"run_{stamp}"
        self.log_dir = os.path.join(log_dir, self.run_id)
        os.makedirs(os.path.join(self.log_dir, "episodes"), exist_ok=True)
        self.metrics_file = os.path.join(self.log_dir, "metrics.jsonl")
        self.episode_count = 0
        self.current_episode: Optional[str] = None
        self.episode_file: Optional[str] = None
        self.episode_start_time: Optional[float] = None
        self._buffer = []

    def begin_episode(self, episode_id: Optional[str] = None, metadata: Optional[Dict] = None):
        self.episode_count += 1
        episode_id = episode_id or f"ep_{self.episode_count:06d}"
        self.episode_start_time = time.time()
        self.episode_file = os.path.join(self.log_dir, "episodes", f"{episode_id}.jsonl")
        with open(self.episode_file, "w") as f:
            f.write(json.dumps(_plain({"type": "header", "episode_id": episode_id,
                                       "start_time": self.episode_start_time, "metadata": metadata or {}})) + "\n")
        self.current_episode = episode_id
        self._buffer = []

    def log_state(self, entity_id: str, state: Dict[str, Any], timestamp: Optional[float] = None):
        if not self.current_episode:
            return
        t = (timestamp or time.time()) - self.episode_start_time
        self._buffer.append({"type": "state", "timestamp": t, "entity_id": entity_id, "state": state})
        if len(self._buffer) >= 100:
            self._flush()

    def log_event(self, event_type: str, source: str, target: Optional[str] = None, data: Optional[Dict] = None,
                  timestamp: Optional[float] = None):
        if not self.current_episode:
            return
        t = (timestamp or time.time()) - self.episode_start_time
        self._buffer.append({"type": "event", "timestamp": t, "event_type": event_type, "source": source,
                             "target": target, "data": data})

    def end_episode(self, outcome: str, metrics: Dict[str, Any]):
        if not self.current_episode:
            return
        self._flush()
        end = time.time()
        duration = end - self.episode_start_time
        with open(self.episode_file, "a") as f:
            f.write(json.dumps(_plain({"type": "footer", "episode_id": self.current_episode, "end_time": end,
                                       "duration": duration, "outcome": outcome, "metrics": metrics})) + "\n")
        self.log_metrics(dict({"episode": self.current_episode, "outcome": outcome, "duration": duration}, **metrics))
        self.current_episode = self.episode_file = self.episode_start_time = None

    def log_metrics(self, metrics: Dict[str, Any]):
        with open(self.metrics_file, "a") as f:
            f.write(json.dumps(_plain(dict({"timestamp": time.time()}, **metrics))) + "\n")

    def _flush(self):
        if self._buffer and self.episode_file:
            with open(self.episode_file, "a") as f:
                for entry in self._buffer:
                    f.write(json.dumps(_plain(entry)) + "\n")
        self._buffer = []


class VecEpisodeRecorder:
    """Episode files for a few watched environments of a vectorised run.

    Call `on_step(actions, rewards, dones, infos)` after every `venv.step(actions)`; for each watched environment it
    logs the three entities of inference.py:535-548 and, when the episode ends, the footer of inference.py:555-614
    and starts the next episode file.  Everything else in the batch is left untouched (no per-env Python work)."""

    def __init__(self, log_dir: str = "logs", indices: Iterable[int] = (0,), run_name: Optional[str] = None,
                 volley_mode: bool = False):
        self.indices = [int(i) for i in indices]
        self.volley_mode = bool(volley_mode)
        self._logs = {i: EpisodeLog(log_dir, f"{run_name or 'vec'}_env{i:06d}") for i in self.indices}
        self._acc = {}
        self.results = []
        for i in self.indices:
            self._begin(i)

    def _begin(self, i):
        log = self._logs[i]
        log.begin_episode(f"ep_{log.episode_count:04d}", {"env_index": i})
        self._acc[i] = dict(total_reward=0.0, steps=0, min_distance=float("inf"))

    def on_step(self, actions, rewards, dones, infos):
        actions = np.asarray(actions)
        for i in self.indices:
            info, log, acc = infos[i], self._logs[i], self._acc[i]
            acc["total_reward"] += float(rewards[i])
            acc["steps"] += 1
            log.log_state("interceptor", {"position": np.asarray(info["interceptor_pos"]).tolist(),
                                          "fuel": info.get("fuel_remaining", 0), "action": actions[i].tolist()})
            log.log_state("missile", {"position": np.asarray(info["missile_pos"]).tolist()})
            if info.get("radar_debug") is not None:
                log.log_state("radar", info["radar_debug"])
            d = float(np.linalg.norm(np.asarray(info["interceptor_pos"]) - np.asarray(info["missile_pos"])))
            acc["min_distance"] = min(acc["min_distance"], d)
            if not dones[i]:
                continue
            if self.volley_mode:   # inference.py:569-583
                got, size = info.get("missiles_intercepted", 0), info.get("volley_size", 1)
                outcome = "all_intercepted" if got == size else ("partial_interception" if got > 0 else "failed")
            else:
                outcome = "intercepted" if info.get("intercepted", False) else "failed"
            metrics = {"total_reward": acc["total_reward"], "steps": acc["steps"], "final_distance": info["distance"],
                       "fuel_used": info.get("fuel_used", 0), "volley_mode": self.volley_mode,
                       "missiles_intercepted": info.get("missiles_intercepted", 0) if self.volley_mode else None,
                       "volley_size": info.get("volley_size", 1) if self.volley_mode else None}
            self.results.append(dict(metrics, env_index=i, outcome=outcome, min_distance=acc["min_distance"],
                                     episode_id=log.current_episode))
            log.end_episode(outcome, metrics)
            self._begin(i)

    def close(self):
        for i in self.indices:
            log = self._logs[i]
            if log.current_episode:
                log._flush()
