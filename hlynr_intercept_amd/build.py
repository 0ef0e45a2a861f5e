"""Builds the HIP shared library in-tree: hlynr_intercept_amd/libhlx.so (gfx950 only)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.environ.get("HLX_LIBRARY") or os.path.join(_HERE, "libhlx.so")   # HLX_LIBRARY: diagnostic builds
SOURCES = ["hlx_kernels.hip"]
DEPS = ["hlx_kernels.hip", "hlx_inst_gen.h", "hlx_host.inc", "hlx_obs.inc", "hlx_device.h", "hlx_kargs.h", "hlx_kcfg.h", "hlx_bake_gen.cpp", os.path.join("..", "..", "include", "hlx.h"),
        os.path.join("..", "..", "include", "hlx_obs.h"), "hlx_hrl.inc", os.path.join("..", "..", "include", "hlx_hrl.h")]


SAFE_FLAGS = ["-DHLX_HOT_FROM_MEMORY=1"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (ROCm toolchain required)")


# Scenario presets whose constants are baked into dedicated step-kernel instantiations (literals instead of cross-lane
# fetches): (name, scenario, physics, overrides).  hlx_create uses a baked instantiation only when the configuration it
# is given derives exactly these constants; anything else runs the ordinary variants.
BAKED = [
    ("medium/base", "medium", "base", None),                      # BASELINE.json configs[1] (the headline)
    ("medium/v2dr", "medium", "v2dr", None),                      # configs[2]
    ("hard/config", "hard", "config", None),                      # configs[3]
    ("medium/config/volley3", "medium", "config", {"volley_mode": True, "volley_size": 3}),   # configs[4]
    ("medium/config", "medium", "config", None),                  # config.yaml as shipped
    ("medium/v2", "medium", "v2", None),                          # constructor defaults (train_flat_ppo.py)
    ("easy/config", "easy", "config", None),                      # configs[0]
]


def generate_baked(verbose: bool = False) -> str:
    """(Re)write csrc/hlx_baked_gen.h from the presets above with the g++-compiled generator (same build_kcfg as the
    library).  The file is committed; it is rewritten only when its content changes."""
    import ctypes as C
    import tempfile

    from . import _lib
    from .config import resolve_config
    from .scenarios import scenario_config

    out = os.path.join(CSRC, "hlx_baked_gen.h")
    gxx = shutil.which("g++") or shutil.which("c++")
    if not gxx:
        if os.path.exists(out):
            return out          # no host compiler: keep the committed tables (hlx_create verifies them anyway)
        raise RuntimeError("g++ not found and csrc/hlx_baked_gen.h is missing")
    with tempfile.TemporaryDirectory() as tmp:
        exe, rec = os.path.join(tmp, "bake_gen"), os.path.join(tmp, "cfgs.bin")
        subprocess.check_call([gxx, "-O1", "-std=c++17", "-o", exe, os.path.join(CSRC, "hlx_bake_gen.cpp"), "-lm"])
        with open(rec, "wb") as f:
            for name, scen, phys, over in BAKED:
                cfg = _lib.make_hlx_config(resolve_config(scenario_config(scen, phys, over)))
                f.write(name.encode().ljust(64, b"\0"))
                f.write(bytes(cfg))
        text = subprocess.check_output([exe, rec]).decode()
    old = open(out).read() if os.path.exists(out) else None
    if text != old:
        with open(out, "w") as f:
            f.write(text)
        if verbose:
            print("wrote", out)
    return out


def _src_hash() -> str:
    """sha256 over the contents of every source the library is built from (mtimes do not survive a copy to another box)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS + ["hlx_baked_gen.h"]):
        path = os.path.join(CSRC, d)
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def needs_build() -> bool:
    """True when the library is missing or was built from different sources (content hash in libhlx.so.srchash)."""
    if os.environ.get("HLX_LIBRARY"):
        return False
    if not os.path.exists(LIB):
        return True
    try:
        with open(LIB + ".srchash") as f:
            return f.read().strip() != _src_hash()
    except OSError:
        mt = os.path.getmtime(LIB)          # a library without its sidecar: fall back to modification times
        return any(os.path.getmtime(os.path.join(CSRC, d)) > mt for d in DEPS)


INST_LIST = os.path.join(CSRC, "hlx_inst_gen.h")
PARTS = 8      # translation units the step kernel's instantiations are spread over (one hipcc each, run at once)
_INST_RE = __import__("re").compile(r"hlx_env_kernelILj(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)ELi(\d+)EE")


def instantiations(text: str):
    """The (SPEC, MODE, NOISE, PERSIST, LATE, BAKE) tuples of the step-kernel instantiations a disassembly listing names."""
    return sorted({tuple(int(x) for x in m.groups()) for m in _INST_RE.finditer(text)})


def write_inst_list(insts, verbose: bool = False) -> bool:
    """csrc/hlx_inst_gen.h: which instantiation is compiled in which part.  Heavy ones first, dealt round robin: a generic (run-time
    flags) or fused-rollout kernel takes several times as long to compile as a baked single-step one.  Committed, like
    hlx_baked_gen.h; rewritten only when its content changes.  Returns True if it was rewritten."""
    def weight(t):
        spec, mode, noise, persist, late, bake = t
        return -((4 if spec & 0x80000000 else 2 if not bake else 1) * (3 if persist else 2 if mode == 0 else 1) + (1 if noise else 0))
    order = sorted(insts, key=lambda t: (weight(t), t))
    lines = ["// generated by hlynr_intercept_amd/build.py from the instantiations of the last library it built: which translation-unit part",
             "// compiles which instantiation of the step kernel (hlx_kernels.hip, \"Launchers\").  A hint about WHERE to compile, never about what",
             "// exists: an instantiation that is not listed is compiled with the host code, as in a single translation unit."]
    for k in range(PARTS):
        lines.append(f"#if !defined(HLX_TU_PART) || HLX_TU_PART == {k}")
        for spec, mode, noise, persist, late, bake in order[k::PARTS]:
            lines.append(f"HLX_INST({spec}u, {mode}, {'true' if noise else 'false'}, {'true' if persist else 'false'}, {late}, {bake})")
        lines.append("#endif")
    text = "\n".join(lines) + "\n"
    old = open(INST_LIST).read() if os.path.exists(INST_LIST) else None
    if text == old:
        return False
    with open(INST_LIST, "w") as f:
        f.write(text)
    if verbose:
        print("wrote", INST_LIST, f"({len(order)} instantiations in {PARTS} parts)")
    return True


def _listed() -> int:
    try:
        return open(INST_LIST).read().count("HLX_INST(")
    except OSError:
        return 0


def _compile_parallel(flags, extra, out, verbose):
    """The library from PARTS + 1 translation units compiled at once (hlx_kernels.hip, "Launchers"), linked into `out`."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    src = os.path.join(CSRC, SOURCES[0])
    cflags = [f for f in flags if f != "-shared"]
    with tempfile.TemporaryDirectory(dir=_HERE) as tmp:
        jobs = [(os.path.join(tmp, "host.o"), ["-DHLX_TU_HOST"])] + [(os.path.join(tmp, f"part{k}.o"), [f"-DHLX_TU_PART={k}"]) for k in range(PARTS)]

        def one(job):
            obj, defs = job
            cmd = [_hipcc()] + cflags + extra + defs + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
            return obj

        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2)))) as pool:
            objs = list(pool.map(one, jobs))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)


def build(force: bool = False, verbose: bool = False) -> str:
    """Build (if needed) under an exclusive file lock: processes started together (pytest-xdist, several ranks without
    LOCAL_RANK, notebooks) queue up instead of compiling into the same file; the compiler writes to a name unique to this
    process and the result is renamed into place only after the hot-word check has passed."""
    import fcntl
    if not force and not needs_build():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():      # somebody else built it while this process waited
                return LIB
            generate_baked(verbose)
            tmp = f"{LIB}.{os.getpid()}.unverified"
            flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                     "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value"]

            def compile_(extra):
                # several translation units at once when csrc/hlx_inst_gen.h says how to split the instantiations (HLX_SINGLE_TU=1:
                # never); any failure on that path falls back to the single translation unit of rounds 1-4
                if _listed() and not os.environ.get("HLX_SINGLE_TU"):
                    try:
                        _compile_parallel(flags, extra, tmp, verbose)
                        return
                    except Exception as why:
                        print(f"hlynr_intercept_amd.build: parallel build failed ({why}); compiling one translation unit", flush=True)
                cmd = [_hipcc()] + flags + extra + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd, cwd=CSRC)

            safe = bool(os.environ.get("HLX_SAFE_BUILD"))       # force the safe build (tests; a toolchain known to need it)
            compile_(SAFE_FLAGS if safe else [])
            # the constants of the step kernel are fetched across lanes (v_readlane): a register-allocator spill of those two
            # VGPRs would silently corrupt them, so the code object is inspected before the library is put in place
            from . import hotcheck
            if not safe:
                try:
                    hotcheck.verify(tmp)
                except hotcheck.HotcheckToolsMissing:
                    os.replace(tmp, LIB + ".unchecked")
                    raise
                except hotcheck.HotcheckViolation as why:
                    # This compiler does to a hot-word register what the cross-lane scheme cannot survive.  Not a reason to have no
                    # library: the SAFE build reads the same constants from memory -- same results, slower (hlx_kernels.hip,
                    # HLX_HOT_FROM_MEMORY) -- and says so (hlx_hot_words_from_memory(), bench.py's line, the marker file).
                    os.replace(tmp, LIB + ".rejected")
                    print(f"hlynr_intercept_amd.build: {why}\n  -> building the SAFE variant (-DHLX_HOT_FROM_MEMORY=1) instead", flush=True)
                    safe = True
                    compile_(SAFE_FLAGS)
                except Exception:
                    os.replace(tmp, LIB + ".rejected")
                    raise
            marker = LIB + ".safe"
            if safe:
                with open(marker, "w") as f:
                    f.write("built with -DHLX_HOT_FROM_MEMORY=1 (hot constants read from memory; see build.py)\n")
            elif os.path.exists(marker):
                os.remove(marker)
            os.replace(tmp, LIB)
            try:      # where the NEXT build compiles what: from what this one contains (a hint file: never a reason to fail)
                write_inst_list(instantiations(hotcheck.listing(LIB)), verbose)
            except Exception:
                pass
            with open(LIB + ".srchash", "w") as f:
                f.write(_src_hash() + "\n")
            return LIB
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


SAFE_LIB = os.path.join(_HERE, "libhlx_safe.so")


def build_safe(force: bool = False, minimal: bool = False) -> str:
    """The SAFE variant beside the product library (hlynr_intercept_amd/libhlx_safe.so, never loaded by default): what
    tests/test_safe_build_gpu.py compares with the product build bit for bit.  `minimal`: the base kernel variant only (seconds
    instead of minutes).  Returns the path; up to date when its sidecar holds the current source hash (+ the variant)."""
    want = _src_hash() + (" minimal" if minimal else " full")
    try:
        with open(SAFE_LIB + ".srchash") as f:
            have = f.read().strip()
    except OSError:
        have = ""
    if not force and os.path.exists(SAFE_LIB) and (have == want or (minimal and have == _src_hash() + " full")):
        return SAFE_LIB
    generate_baked(False)
    tmp = f"{SAFE_LIB}.{os.getpid()}.tmp"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16",
             "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value"]
    done = False
    if not minimal and _listed() and not os.environ.get("HLX_SINGLE_TU"):
        try:
            _compile_parallel(flags, SAFE_FLAGS, tmp, False)
            done = True
        except Exception as why:
            print(f"hlynr_intercept_amd.build: parallel build of the safe variant failed ({why}); compiling one translation unit", flush=True)
    if not done:
        subprocess.check_call([_hipcc()] + flags + SAFE_FLAGS + (["-DHLX_AB_MINIMAL"] if minimal else []) + ["-o", tmp] +
                              [os.path.join(CSRC, s) for s in SOURCES], cwd=CSRC)
    os.replace(tmp, SAFE_LIB)
    with open(SAFE_LIB + ".srchash", "w") as f:
        f.write(want + "\n")
    return SAFE_LIB


def safe_lib_variants() -> str:
    """'full', 'minimal' or '' for the libhlx_safe.so that is there."""
    try:
        with open(SAFE_LIB + ".srchash") as f:
            h, kind = f.read().split()
        return kind if h == _src_hash() and os.path.exists(SAFE_LIB) else ""
    except (OSError, ValueError):
        return ""


def build_stamps(level: int = 1) -> str:
    """Diagnostic build with s_memtime stamps (tools/diag_stamps.py); never used by the product path."""
    out = os.path.join(_HERE, "libhlx_stamps.so")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-mllvm", "-amdgpu-kernarg-preload-count=16", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-unused-value", f"-DHLX_STAMPS={level}", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES],
                          cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
