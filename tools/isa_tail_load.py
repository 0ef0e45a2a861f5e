"""Where does each hlx_env_kernel instantiation issue the scalar load of its kernarg tail (the four output pointers, byte 56 on)?
At entry its latency (~0.45 us) hides behind the state loads and the Philox block; sunk to its first use behind the Philox block
it is exposed in full.  python tools/isa_tail_load.py listing.s  ->  one line per kernel that has it late."""
import sys
txt = open(sys.argv[1]).read().split("\n")
cur, n, info = None, 0, {}
for l in txt:
    if l.startswith("_ZN") and "hlx_env_kernel" in l and ":" in l:
        cur = l.split("EEvP15")[0][28:]; n = 0; info[cur] = [None, None]
        continue
    if cur is None:
        continue
    t = l.strip()
    if t.startswith(".end_amdhsa_kernel") or l.startswith("\t.section"):
        cur = None
        continue
    if l.startswith("\t") and t and not t.startswith((".", ";")):
        n += 1
        if t.startswith("s_load_dword") and ", 0x38" in t and info[cur][0] is None:
            info[cur][0] = n
        if t.startswith("global_load") and info[cur][1] is None:
            info[cur][1] = n
late = {k: v for k, v in info.items() if v[0] is None or v[0] > 200}
print(len(info), "kernels;", len(late), "issue the kernarg tail load more than 200 instructions in:")
for k, v in late.items():
    print("  ", k, "tail load at", v[0], "first state load at", v[1])
