"""The SAFE build (-DHLX_HOT_FROM_MEMORY=1): what hlynr_intercept_amd/build.py falls back to when hotcheck.py refuses the product
build because the compiler spilled a hot-word register.  Same source, constants read from the parameter block in memory instead of
across lanes: it must give the product build's bits."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import hashlib, json, sys, torch
from hlynr_intercept_amd.scenarios import scenario_config
from hlynr_intercept_amd.vec_env import HlynrVecEnv
out = {}
import os
CASES = (("base", {"max_steps": 40}), ("v2dr", {"max_steps": 40}), ("config", {"max_steps": 40, "volley_mode": True, "volley_size": 3}))
for physics, over in CASES[:int(os.environ["HLX_SAFE_PROBE_CASES"])]:
    env = HlynrVecEnv(scenario_config("medium", physics, over), num_envs=700, seed=21)
    out["safe"] = env.safe_build
    h = hashlib.sha256()
    h.update(env.reset_torch().cpu().numpy().tobytes())
    g = torch.Generator(device=env.device).manual_seed(2)
    for t in range(90):
        o, r, te, tr, info = env.step_torch(torch.rand((700, 6), generator=g, device=env.device) * 2 - 1, want_done_list=True)
        for x in (o, r, te, tr, info["flags"], info["distance"], env.terminal_obs[(te | tr) != 0]):
            h.update(x.cpu().numpy().tobytes())
    h.update(bytes(env.get_state()))
    out[physics] = h.hexdigest()
    env.close()
print(json.dumps(out))
"""


def _run(library, kind):
    env = dict(os.environ, PYTHONPATH=ROOT, HLX_SAFE_PROBE_CASES="3" if kind == "full" else "1")
    if library:
        env["HLX_LIBRARY"] = library
    else:
        env.pop("HLX_LIBRARY", None)
    res = subprocess.run([sys.executable, "-c", _PROBE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_safe_build_gives_the_product_build_s_bits():
    """Against the full safe library when the build made one (__graft_entry__.build() does: three kernel variants compared), else
    against a base-variant-only one compiled here in seconds."""
    from hlynr_intercept_amd import build
    kind = build.safe_lib_variants()
    if not kind:
        build.build_safe(minimal=True)
        kind = build.safe_lib_variants()
    product, safe = _run(None, kind), _run(build.SAFE_LIB, kind)
    assert product.pop("safe") is False and safe.pop("safe") is True
    assert product == safe and len(product) == (3 if kind == "full" else 1), (product, safe)


def test_build_falls_back_to_the_safe_variant_when_the_lint_refuses(monkeypatch, tmp_path):
    """build.py's control flow, no compiler involved: a HotcheckViolation on the product build is answered with a second compilation
    carrying -DHLX_HOT_FROM_MEMORY=1, which is not linted for hot words, and leaves the marker file."""
    from hlynr_intercept_amd import build, hotcheck
    lib = str(tmp_path / "libhlx.so")
    calls = []

    def fake_compile(cmd, cwd=None):
        calls.append(cmd)
        open(cmd[cmd.index("-o") + 1], "w").write("x")

    def fake_verify(path=None):
        raise hotcheck.HotcheckViolation("hotcheck: the register allocator spilled a hot-constant register (test)")

    monkeypatch.setattr(build, "LIB", lib)
    monkeypatch.setattr(build, "generate_baked", lambda verbose=False: "")
    monkeypatch.setattr(build.subprocess, "check_call", fake_compile)
    monkeypatch.setattr(hotcheck, "verify", fake_verify)
    monkeypatch.delenv("HLX_SAFE_BUILD", raising=False)
    monkeypatch.setenv("HLX_SINGLE_TU", "1")          # (one compiler call per library: the parallel path is test_parallel_build... below)
    assert build.build(force=True) == lib
    assert len(calls) == 2 and "-DHLX_HOT_FROM_MEMORY=1" not in calls[0] and "-DHLX_HOT_FROM_MEMORY=1" in calls[1]
    assert os.path.exists(lib) and os.path.exists(lib + ".safe") and os.path.exists(lib + ".rejected") and os.path.exists(lib + ".srchash")
    # ... and a clean product build afterwards removes the marker
    monkeypatch.setattr(hotcheck, "verify", lambda path=None: 1)
    calls.clear()
    assert build.build(force=True) == lib and len(calls) == 1 and not os.path.exists(lib + ".safe")


def test_parallel_build_splits_the_instantiations_and_falls_back_to_one_translation_unit(monkeypatch, tmp_path):
    """build.py's parallel path, no compiler involved: one host unit + PARTS part units compiled with the TU macros and linked; a
    failure there is answered with the single translation unit; the list of instantiations is a pure function of a listing."""
    from hlynr_intercept_amd import build, hotcheck
    lib = str(tmp_path / "libhlx.so")
    calls = []

    def fake(cmd, cwd=None):
        calls.append(cmd)
        open(cmd[cmd.index("-o") + 1], "w").write("x")

    monkeypatch.setattr(build, "LIB", lib)
    monkeypatch.setattr(build, "generate_baked", lambda verbose=False: "")
    monkeypatch.setattr(build.subprocess, "check_call", fake)
    monkeypatch.setattr(hotcheck, "verify", lambda path=None: 1)
    monkeypatch.delenv("HLX_SINGLE_TU", raising=False)
    monkeypatch.delenv("HLX_SAFE_BUILD", raising=False)
    assert build._listed() >= 80                                   # the committed hint file names the shipped instantiations
    assert build.build(force=True) == lib
    units = [c for c in calls if "-c" in c]
    assert len(units) == build.PARTS + 1 and sum("-DHLX_TU_HOST" in c for c in units) == 1
    assert sorted(d for c in units for d in c if d.startswith("-DHLX_TU_PART=")) == sorted(f"-DHLX_TU_PART={k}" for k in range(build.PARTS))
    assert len(calls) == build.PARTS + 2 and "-shared" in calls[-1] and all("-shared" not in c for c in units)
    # a compiler that fails on the split units: one translation unit, as in rounds 1-4
    calls.clear()

    def flaky(cmd, cwd=None):
        calls.append(cmd)
        if any(d.startswith("-DHLX_TU_PART=") for d in cmd):
            raise build.subprocess.CalledProcessError(1, cmd)
        open(cmd[cmd.index("-o") + 1], "w").write("x")

    monkeypatch.setattr(build.subprocess, "check_call", flaky)
    assert build.build(force=True) == lib and "-shared" in calls[-1] and not any(d.startswith("-DHLX_TU") for d in calls[-1])
    text = "<_ZN12_GLOBAL__N_114hlx_env_kernelILj608ELi0ELb0ELb0ELi2ELi1EEEvP>:\n<_ZN12_GLOBAL__N_114hlx_env_kernelILj639ELi1ELb1ELb0ELi0ELi0EEEvP>:"
    assert build.instantiations(text) == [(608, 0, 0, 0, 2, 1), (639, 1, 1, 0, 0, 0)]
