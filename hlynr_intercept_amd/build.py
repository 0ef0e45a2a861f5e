"""Builds the HIP shared library in-tree: hlynr_intercept_amd/libhlx.so (gfx950 only)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libhlx.so")
SOURCES = ["hlx_kernels.hip"]
DEPS = ["hlx_kernels.hip", "hlx_host.inc", "hlx_device.h", "hlx_kargs.h", os.path.join("..", "..", "include", "hlx.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (ROCm toolchain required)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > mt for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
