"""Fixture drift guard (build container only): the committed golden fixtures must be exactly what the generators produce
from the reference today.  Skipped where /root/reference does not exist (the GPU box): fixtures are data there."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/rl_system"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")


@needs_ref
def test_step_and_radar_fixtures_regenerate_bit_for_bit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "66 of 66 fixtures regenerate bit-identically" in out.stdout


@needs_ref
def test_episode_log_fixtures_regenerate():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_log_golden.py"), "--check"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "check: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
