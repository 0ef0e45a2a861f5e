"""Dynamic instruction count of a step kernel's COMMON path, read off the disassembly of the built library: walk from the kernel's
entry, follow unconditional branches, fall through every conditional one (the kernel is laid out that way: rare bodies sit behind
the hot path's s_endpgm -- RARE(), DESIGN.md section 5 -- and the few short forward skips inside it guard blocks nearly every
wave executes), stop at s_endpgm.  Prints the instruction count by class, the bytes of code the path touches and how many 64-byte
instruction-cache lines that is; `--list` prints the path.
  python tools/isa_hotpath.py [libhlx.so] [substring of the kernel symbol, default the headline: base, lone-wave schedule, baked]"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hlynr_intercept_amd import hotcheck      # noqa: E402


def disassemble(path):
    """{symbol: [(address, text)]} of every hlx_env_kernel instantiation (llvm-objdump with addresses)."""
    import subprocess
    txt = "\n".join(subprocess.run([os.path.join(hotcheck.LLVM_BIN or hotcheck._llvm_bin(), "llvm-objdump"), "-d", co], check=True,
                                   capture_output=True, text=True).stdout for co in hotcheck.code_objects(path))
    out, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(2), []) if "hlx_env_kernel" in m.group(2) else None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)", line)
        if m:
            cur.append((int(m.group(2), 16), m.group(1).strip(), 4 * len(m.group(3).split())))
    return out


def classify(op):
    if op.startswith("v_") and "f64" in op:
        return "valu_f64"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop")):
        return op.split("_")[1]
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_load", "buffer_load", "s_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "global_atomic", "buffer_atomic")):
        return "vmem_store"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def walk(ins):
    by_addr = {a: k for k, (a, _, _) in enumerate(ins)}
    k, path, taken = 0, [], 0
    seen = set()
    while k < len(ins) and k not in seen:
        seen.add(k)
        a, t, nb = ins[k]
        path.append(k)
        op = t.split()[0]
        if op == "s_endpgm":
            break
        if op == "s_branch":                      # unconditional: follow (objdump prints the target as `<symbol+0xOFF>` or a number of dwords)
            m = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", t)
            if m:
                tgt = ins[0][0] + int(m.group(1), 16)
            else:
                off = int(t.split()[1])
                tgt = a + 4 + 4 * (off if off < 32768 else off - 65536)
            taken += 1
            k = by_addr[tgt]
            continue
        k += 1
    return path, taken


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(hotcheck.HERE), "hlynr_intercept_amd", "libhlx.so")
    key = args[1] if len(args) > 1 else "ILj608ELi0ELb0ELb0ELi2ELi1E"
    ks = disassemble(lib)
    for name, ins in ks.items():
        if key not in name:
            continue
        path, taken = walk(ins)
        cls = collections.Counter(classify(ins[k][1].split()[0]) for k in path)
        nbytes = sum(ins[k][2] for k in path)
        lines = len({ins[k][0] // 64 for k in path})
        total_bytes = ins[-1][0] + ins[-1][2] - ins[0][0]
        print(f"{name[28:60]}: common path {len(path)} instructions ({taken} taken branches), {nbytes} bytes in {lines} 64-byte lines; "
              f"whole kernel {len(ins)} instructions, {total_bytes} bytes")
        print("   ", dict(sorted(cls.items(), key=lambda kv: -kv[1])))
        if "--list" in sys.argv:
            for k in path:
                print(f"      {ins[k][0]:x}  {ins[k][1]}")


if __name__ == "__main__":
    main()
