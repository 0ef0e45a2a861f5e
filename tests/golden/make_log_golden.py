#!/usr/bin/env python3
"""Writes tests/golden/logs/*.jsonl with the reference's own `UnifiedLogger` (rl_system/logger.py:91-288; stdlib + numpy)
from the scripted session of tests/log_script.py, under a deterministic clock.  Build container only (needs
/root/reference); the outputs are plain data.  `--check` regenerates into a scratch directory and compares."""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/rl_system")

from tests.log_script import FakeTime, session  # noqa: E402


def generate(out_dir):
    import logger as ref_logger

    with tempfile.TemporaryDirectory() as tmp:
        log = ref_logger.UnifiedLogger(log_dir=tmp, run_name="golden")
        ref_logger.time = FakeTime()      # only the logger module's own clock readings (logging keeps the real clock)
        for method, kw in session():
            if method == "log_event":
                log.log_event(ref_logger.EpisodeEvent(timestamp=kw["timestamp"], event_type=kw["event_type"], source=kw["source"],
                                                      target=kw["target"], data=kw["data"]))
            elif method == "log_metrics":
                log.log_metrics(kw["metrics"])
            else:
                getattr(log, method)(**kw)
        os.makedirs(out_dir, exist_ok=True)
        for name in sorted(os.listdir(log.log_dir / "episodes")):
            shutil.copy(log.log_dir / "episodes" / name, os.path.join(out_dir, "episodes_" + name))
        shutil.copy(log.metrics_file, os.path.join(out_dir, "metrics.jsonl"))


if __name__ == "__main__":
    target = os.path.join(HERE, "logs")
    if "--check" in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            generate(tmp)
            bad = [f for f in sorted(set(os.listdir(tmp)) | set(os.listdir(target)))
                   if not (os.path.exists(os.path.join(tmp, f)) and os.path.exists(os.path.join(target, f))
                           and open(os.path.join(tmp, f)).read() == open(os.path.join(target, f)).read())]
        print("check:", "ok" if not bad else bad)
        sys.exit(1 if bad else 0)
    generate(target)
    print(sorted(os.listdir(target)))
