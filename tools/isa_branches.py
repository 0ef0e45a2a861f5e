"""Branches of one kernel in a -DHLX_MARKS -S listing (see tools/isa_sections.py for the compile line): position, section
(the last HLXMARK passed), opcode, target and the distance jumped in instructions.  A lone wave per SIMD pays an instruction
buffer refill for every TAKEN branch, so a forward skip over a block that is rarely executed (taken almost always) costs
more than the block placed out of line behind a branch that is almost never taken (RARE()).
  python tools/isa_branches.py /tmp/marks.s 'ILj608ELi0ELb0ELb0ELb1ELi1' [first_section last_section]"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and ':' in l)
end = next(i for i in range(start + 1, len(txt)) if txt[i].strip().startswith('.end_amdhsa_kernel') or txt[i].startswith('\t.section'))
L = txt[start:end]
pos, n, idx = {}, 0, []
for l in L:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        pos[m.group(1)] = n
    if l.startswith('\t') and t and not t.startswith(('.', ';')):
        n += 1
    idx.append(n)
sec = -1
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (-1, 99)
fwd_short = 0
for i, l in enumerate(L):
    m = re.search(r'; HLXMARK (\d+)', l)
    if m:
        sec = int(m.group(1))
    t = l.strip()
    if t.startswith(('s_cbranch', 's_branch')) and lo <= sec <= hi:
        op, tgt = t.split()[:2]
        d = pos.get(tgt)
        dist = (d - idx[i]) if d is not None else None
        if dist is not None and 0 < dist < 200 and op != 's_branch':
            fwd_short += 1
        print(f"{idx[i]:5d}  sec {sec:2d}  {op:18s} {tgt:12s} {dist}")
print("instructions", n, "| short forward conditional skips (candidates):", fwd_short)
