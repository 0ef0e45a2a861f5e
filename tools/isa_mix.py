"""Static instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s <substring of symbol>"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and ':' in l)
end = next(i for i in range(start + 1, len(txt)) if txt[i].startswith('.Lfunc_end') or txt[i].strip().startswith('.end_amdhsa_kernel') or txt[i].startswith('\t.section'))
lines = [l.strip() for l in txt[start:end] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
ops = collections.Counter(l.split()[0] for l in lines)
cat = collections.Counter()
for op, c in ops.items():
    if op.startswith('v_') and 'f64' in op: cat['valu_f64'] += c
    elif op.startswith('v_'): cat['valu_other'] += c
    elif op.startswith('s_'): cat['salu'] += c
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cat['vmem'] += c
    elif op.startswith('ds_'): cat['lds'] += c
    else: cat['other'] += c
print('total', len(lines), dict(cat))
print(ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40))
