"""Build-time guard for the hot-constant scheme of the step kernel.

Every constant of the 512-byte parameter block lives in ONE lane of two VGPRs per wave ("hot words") and is fetched
with v_readlane at its use site.  That is only correct while those two VGPRs hold all 64 lanes: a register-allocator
spill reload (scratch_load) or live-range-split copy (v_mov) of a hot word inside divergent control flow restores only
the ACTIVE lanes, and a later v_readlane of an inactive lane would return garbage.  (Seen once: the parity-only NOISE
instantiation of the generic variant, at 256 VGPRs with 440 B of scratch, lost `on_rel` that way.)

This script disassembles the gfx950 code object inside the BUILT library (what ships) and checks, for every
hlx_env_kernel instantiation, each VGPR that is a v_readlane source and never a v_writelane destination (those are the
compiler's own SGPR-spill registers, which it saves in whole-wave mode): between the vector load that fills it and its
every v_readlane is classified by what its source register holds AT THAT POINT -- the last instruction that wrote it:
the parameter-block load itself (`global_load_dword`: the kernel's only single-dword vector loads) = a hot word with all
64 lanes valid; an ordinary computation = a computed value read across lanes (since round 2: the wave-cooperative respawn
draws, evaluated in wave-uniform control flow); a plain copy whose chain leads back to a hot word (`v_mov_b32 vX, vHOT`:
a live-range split, possibly copied back later), a scratch reload or an AGPR read-back = the hazard, made under whatever
EXEC mask was current.  Registers are recycled (a baked instantiation may never read one of its hot words and reuse the
register for something else), so the classification is per read, not per register.
Round 3 (advisor finding: a hot word rematerialised through v_cndmask / a DPP move / v_or / v_perm passed as "computed"):
  * a hot word is identified by its LOAD, not by the opcode alone: `global_load_dword vX, vLANE, s[P:P+1]` (offset 0 or 256)
    from the SGPR pair the kernel's first such load uses -- the parameter-block pointer; other single-dword loads (the
    info['fuel_used'] accumulator) are ordinary values;
  * while a register holds a hot word, v_readlane_b32 is the ONLY instruction that may read it: any other reader (a copy,
    a select, a DPP / permute move, a spill store, an AGPR write) is reported, whatever it feeds -- a lane-local use of a
    hot word has no meaning in this kernel, so every such instruction is the beginning of a rematerialisation;
  * "computed" cross-lane reads are legitimate only for the wave-cooperative respawn draws (RS_ITEMS x 4 = 44) and the
    compaction bookkeeping: at most COMPUTED_READLANE_LIMIT per instantiation (measured: 19-70; the respawn block exists in
    both observation trips).
Usage: python -m hlynr_intercept_amd.hotcheck [libhlx.so | listing.s]; exit code 1 on a spill reload.
hlynr_intercept_amd/build.py runs the same check after every build (`verify`)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def _llvm_bin():
    """Directory of llvm-objcopy / clang-offload-bundler / llvm-objdump: next to the hipcc that built the library
    (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), so a versioned ROCm install found through PATH works; HLX_LLVM_BIN overrides."""
    import shutil
    cands = [os.environ.get("HLX_LLVM_BIN")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if c and all(os.path.exists(os.path.join(c, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
            return c
    raise HotcheckToolsMissing("hotcheck tools not found (llvm-objcopy, clang-offload-bundler, llvm-objdump): looked in "
                               + ", ".join(c for c in cands if c) + "; set HLX_LLVM_BIN")


class HotcheckViolation(RuntimeError):
    """The code object reloads, splits or re-uses a hot-word register: its cross-lane reads would be wrong.  build.py answers with
    the SAFE build (constants read from memory, -DHLX_HOT_FROM_MEMORY=1)."""


class HotcheckToolsMissing(RuntimeError):
    """The disassembler tool chain is missing: the check could not RUN (distinct from a hot-word violation)."""


LLVM_BIN = None


def listing(path=None):
    """Assembly text of the kernels: a `hipcc -S` listing as is, or the disassembly of the library's gfx950 code object."""
    path = path or os.path.join(HERE, "libhlx.so")
    if path.endswith(".s"):
        return open(path).read()
    global LLVM_BIN
    LLVM_BIN = LLVM_BIN or _llvm_bin()
    return "\n".join(subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                                    capture_output=True, text=True).stdout for co in code_objects(path))


def code_objects(path):
    """The gfx950 code objects of a library, as files.  A library linked from several translation units (build.py compiles the
    step kernel's instantiations in parallel) carries one offload bundle per unit, back to back in `.hip_fatbin`."""
    global LLVM_BIN
    LLVM_BIN = LLVM_BIN or _llvm_bin()
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path], check=True)
    blob, magic, out = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__", []
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] or [0]
    for k, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = os.path.join(tmp, f"fat{k}.bin"), os.path.join(tmp, f"gfx950_{k}.co")
        with open(part, "wb") as f:
            f.write(blob[lo:hi])
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + part, "--output=" + co, "--unbundle"], check=True)
        out.append(co)
    return out


def regs(operand):
    """'v12' -> [12]; 'v[4:7]' -> [4, 5, 6, 7]; source modifiers ('-v3', '|v3|') are looked through."""
    operand = operand.strip("-|")
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    return list(range(int(m.group(1)), int(m.group(2)) + 1)) if m else []


COMPUTED_READLANE_LIMIT = 104    # cross-lane reads of computed values per instantiation: the respawn block's draw collection (RS_ITEMS x 4 =
                                 # 44) exists twice since the second observation trip is a specialised copy, + bookkeeping; measured 19-70
_STORES = ("global_store", "buffer_store", "scratch_store", "flat_store", "ds_write", "global_atomic", "buffer_atomic", "flat_atomic", "ds_add")


def check(text):
    """-> (kernels seen, [(kernel, reg, {opcode: count})] violations, [] (kept for the old three-value signature))."""
    lines = [l.split("//")[0].rstrip() for l in text.split("\n")]
    label = re.compile(r"^(?:[0-9a-f]+ <)?(_ZN\S*hlx_env_kernel[^>:\s]*)>?:")
    any_label = re.compile(r"^(?:[0-9a-f]+ <\S+>:|\S+:\s*(;.*)?$)")
    starts = [i for i, l in enumerate(lines) if label.match(l)]
    fail, info = [], []
    for i in starts:
        name = label.match(lines[i]).group(1)
        end = next((j for j in range(i + 1, len(lines))
                    if lines[j].strip().startswith(".end_amdhsa_kernel") or (any_label.match(lines[j]) and "hlx_env_kernel" in lines[j])
                    or re.match(r"^[0-9a-f]+ <_Z", lines[j])), len(lines))
        readlane_src, writelane_dst = set(), set()
        reads = collections.defaultdict(list)           # reg -> [positions of v_readlane]
        writes = collections.defaultdict(list)          # reg -> [(position, opcode, source operands)]
        other_readers = []                              # (position, opcode, [source VGPRs]) of everything that is not a v_readlane
        hot_base = None                                 # SGPR pair of the parameter-block pointer: the first `global_load_dword v, v, s[a:b]`
        for pos, t in enumerate(x.strip() for x in lines[i + 1:end]):
            if not t or t.startswith((".", ";")):
                continue
            parts = t.replace(",", " ").split()
            op = parts[0]
            if op == "global_load_dword" and len(parts) > 3 and re.fullmatch(r"s\[\d+:\d+\]", parts[3]):
                if hot_base is None:
                    hot_base = parts[3]
                if parts[3] == hot_base and (len(parts) == 4 or parts[4] in ("offset:256",)):
                    op = "HOT_LOAD"
            if op != "v_readlane_b32" and len(parts) > 1 and (op.startswith(("v_", "ds_", "scratch_", "global_", "buffer_", "flat_"))):
                src_ops = parts[1:] if op.startswith(_STORES) or op.startswith("v_cmp") else parts[2:]
                if op.startswith("v_pk_") and op.endswith("f32"):
                    # packed float32: a 64-bit source names a register PAIR, of which op_sel / op_sel_hi pick one element for
                    # the low and one for the high result (`v[218:219] ... op_sel_hi:[0,1,1]` reads v218 twice and never v219)
                    sel = {k: [int(x) for x in m.group(1).split(",")] for k in ("op_sel", "op_sel_hi")
                           for m in [re.search(k + r":\[([01,]+)\]", t)] if m}
                    srcs = []
                    for si, o in enumerate(x for x in src_ops if not x.startswith(("op_sel", "neg_", "clamp"))):
                        rs = regs(o)
                        if len(rs) == 2:
                            lo = sel.get("op_sel", [0, 0, 0])[si] if si < 3 else 0
                            hi = sel.get("op_sel_hi", [1, 1, 1])[si] if si < 3 else 1
                            rs = sorted({rs[lo], rs[hi]})
                        srcs += rs
                else:
                    srcs = [r for o in src_ops for r in regs(o)]
                if srcs:
                    other_readers.append((pos, parts[0], srcs))
            if op == "v_readlane_b32":
                for r in regs(parts[2]):
                    readlane_src.add(r)
                    reads[r].append(pos)
            if op == "v_writelane_b32":
                writelane_dst.update(regs(parts[1]))
            if op.startswith(("v_", "global_load", "buffer_load", "scratch_load", "ds_read", "ds_bpermute", "flat_load", "HOT_LOAD")) and \
                    not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and len(parts) > 1:
                for r in regs(parts[1]):
                    writes[r].append((pos, op, parts[2:]))
        def last_write(r, pos):
            prev = [w for w in writes.get(r, ()) if w[0] < pos]
            return prev[-1] if prev else None

        def origin(r, pos, depth=0):
            """what the value in register r at position pos IS: 'hot' (filled by the parameter-block load), 'copy-of-hot',
            'reload' (scratch / AGPR spill space), 'nothing', or 'computed'"""
            w = last_write(r, pos)
            if w is None:
                return "nothing"
            wpos, op, src = w
            if op == "HOT_LOAD":
                return "hot"
            if op.startswith("scratch_load") or op.startswith("v_accvgpr_read"):
                return "reload"
            if op.startswith("v_mov_b32") and depth < 8:
                for o in src[:1]:
                    for x in regs(o):
                        inner = origin(x, wpos, depth + 1)
                        if inner in ("hot", "copy-of-hot"):
                            return "copy-of-hot"
                        if inner == "reload":
                            return "reload"
            return "computed"

        # Every v_readlane is classified by what its source register holds AT THAT POINT (registers are recycled: the
        # register of a hot word that a baked instantiation never reads may later carry a computed value):
        #   hot        -- still the value the parameter-block load put there: all 64 lanes valid, fine;
        #   computed   -- an ordinary value (the wave-cooperative respawn draws, evaluated in wave-uniform control flow);
        #   copy / reload of a hot word -- made under whatever EXEC mask was current: the hazard this check exists for.
        computed_reads = 0
        for r in sorted(readlane_src - writelane_dst):
            kinds = collections.Counter(origin(r, rp) for rp in reads[r])
            computed_reads += kinds.get("computed", 0)
            bad = {k: v for k, v in kinds.items() if k in ("copy-of-hot", "reload", "nothing")}
            if bad:
                fail.append((name, f"v{r}", {"v_readlane of a " + k: v for k, v in bad.items()}))
            else:
                info.append((name, f"v{r}", dict(kinds)))
        # a register that holds a hot word may be read by v_readlane_b32 and by nothing else
        leaks = collections.Counter()
        for pos, opname, srcs in other_readers:
            for r in srcs:
                if r not in writelane_dst and origin(r, pos) == "hot":
                    leaks[(f"v{r}", opname)] += 1
        for (reg, opname), cnt in sorted(leaks.items()):
            fail.append((name, reg, {"hot word read by " + opname + " (only v_readlane_b32 may)": cnt}))
        if computed_reads > COMPUTED_READLANE_LIMIT:
            fail.append((name, "*", {f"v_readlane of computed values: {computed_reads} > {COMPUTED_READLANE_LIMIT}": computed_reads}))
    return len(starts), fail, info


def tail_load_positions(text):
    """{kernel: instructions from its start to the scalar load of the kernarg tail (the four output pointers, byte 56 on)}.
    At entry that load's ~0.45 us hide behind the state loads and the Philox block; sunk to its first use behind the Philox
    block they are exposed in full ("The kernarg tail").  None = no such load found."""
    out, cur, n = {}, None, 0
    for line in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_ZN\S*hlx_env_kernel\S*?)>?:", line)
        if m:
            cur, n = m.group(1), 0
            out[cur] = None
            continue
        t = line.strip()
        if cur is None or not t:
            continue
        if t.startswith((".end_amdhsa_kernel", ".section")) or re.match(r"^[0-9a-f]+ <", line):
            cur = None
            continue
        if line.startswith(("\t", " ")) and not t.startswith((".", ";", "//", "s_nop")):   # (s_nop: the 256-byte preload header's padding)
            n += 1
            if out[cur] is None and t.startswith("s_load_dword") and re.search(r",\s*0x38\b", t.split("//")[0]):
                out[cur] = n
    return out


TAIL_LOAD_LIMIT = 64     # instructions; every instantiation has it within its first ten today


def verify(path=None):
    """Raise if the library at `path` (default: the in-tree libhlx.so) reloads a hot word from scratch, or issues the
    kernarg-tail load late in a single-step kernel."""
    text = listing(path)
    late = {k: v for k, v in tail_load_positions(text).items() if v is None or v > TAIL_LOAD_LIMIT}
    if late and not os.environ.get("HLX_SKIP_TAIL_CHECK"):
        raise RuntimeError("hotcheck: the scalar load of the output pointers is not issued at kernel entry (its latency would be "
                           "exposed behind the Philox block; HLX_SKIP_TAIL_CHECK=1 to build anyway): "
                           + "; ".join(f"{k[28:70]}... at {v}" for k, v in late.items()))
    n, fail, _ = check(text)
    if n == 0:
        raise RuntimeError("hotcheck: no hlx_env_kernel instantiation found in the code object")
    if fail:
        raise HotcheckViolation("hotcheck: the register allocator spilled, split or re-used a hot-constant register "
                           "(v_readlane would read stale lanes): " + "; ".join(f"{k[:60]}... {r} {ops}" for k, r, ops in fail))
    return n


if __name__ == "__main__":
    n, fail, info = check(listing(sys.argv[1] if len(sys.argv) > 1 else None))
    print(f"{n} hlx_env_kernel instantiations checked: {len(fail)} hot-word violations")
    for name, r, ops in fail:
        print(f"  VIOLATION: {name[:72]}... {r}: {ops}")
    sys.exit(1 if fail or n == 0 else 0)
