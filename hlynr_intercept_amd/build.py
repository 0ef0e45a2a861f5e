"""Builds the HIP shared library in-tree: hlynr_intercept_amd/libhlx.so (gfx950 only)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.environ.get("HLX_LIBRARY") or os.path.join(_HERE, "libhlx.so")   # HLX_LIBRARY: diagnostic builds
SOURCES = ["hlx_kernels.hip"]
DEPS = ["hlx_kernels.hip", "hlx_host.inc", "hlx_obs.inc", "hlx_device.h", "hlx_kargs.h", os.path.join("..", "..", "include", "hlx.h"),
        os.path.join("..", "..", "include", "hlx_obs.h"), "hlx_hrl.inc", os.path.join("..", "..", "include", "hlx_hrl.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (ROCm toolchain required)")


def needs_build() -> bool:
    if os.environ.get("HLX_LIBRARY"):
        return False
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > mt for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    tmp = LIB + ".unverified"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value",
           "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    # the constants of the step kernel are fetched across lanes (v_readlane): a register-allocator spill of those two
    # VGPRs would silently corrupt them, so the code object is inspected before the library is put in place
    from . import hotcheck
    try:
        hotcheck.verify(tmp)
    except Exception:
        os.replace(tmp, LIB + ".rejected")
        raise
    os.replace(tmp, LIB)
    return LIB


def build_stamps(level: int = 1) -> str:
    """Diagnostic build with s_memtime stamps (tools/diag_stamps.py); never used by the product path."""
    out = os.path.join(_HERE, "libhlx_stamps.so")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value", f"-DHLX_STAMPS={level}", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES],
                          cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
