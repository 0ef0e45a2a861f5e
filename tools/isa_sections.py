"""Static instruction counts per kernel section: compile with -DHLX_MARKS -S and split the listing of one kernel at the
`; HLXMARK k` comments the STAMP points leave behind.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=max-ilp -DHLX_MARKS -S \
        --cuda-device-only -o /tmp/marks.s hlynr_intercept_amd/csrc/hlx_kernels.hip
  python tools/isa_sections.py /tmp/marks.s 'ILj608ELi0ELb0ELb0ELb1'"""
import collections, re, sys
txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and ':' in l)
end = next(i for i in range(start + 1, len(txt)) if txt[i].strip().startswith('.end_amdhsa_kernel') or txt[i].startswith('\t.section'))
sec, counts, classes = 'entry', collections.OrderedDict(), collections.defaultdict(collections.Counter)
for l in txt[start:end]:
    m = re.search(r'; HLXMARK (\d+)', l)
    if m:
        sec = 'after mark ' + m.group(1)
        continue
    t = l.strip()
    if not l.startswith('\t') or t.startswith(('.', ';')) or not t:
        continue
    op = t.split()[0]
    counts[sec] = counts.get(sec, 0) + 1
    cls = 'f64' if ('f64' in op and op.startswith('v_')) else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'mem'
    classes[sec][cls] += 1
names = {0: 'entry', 1: 'Philox block', 2: 'unpack+clamp', 3: 'interceptor', 4: 'missile', 5: 'wind+termination', 6: 'reward', 7: 'scalar outputs + onboard detection',
         8: 'ground radar', 9: 'datalink+fusion conf', 10: 'fusion+Kalman', 11: '26-D formulas', 12: 'loop exit', 13: 'state stores', 14: 'compaction', 15: 'tail'}
tot = sum(counts.values())
for k, v in counts.items():
    n = names.get(int(k.split()[-1]), '') if k != 'entry' else 'prologue (loads issued)'
    print(f"{k:14s} {n:36s} {v:5d}  {100 * v / tot:5.1f}%  {dict(classes[k])}")
print('total', tot)
