"""static instruction mix of a kernel in a hipcc -save-temps .s file (diagnostic)
usage: tools_isa_mix.py file.s kernel_substring"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
lines = s.split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % key, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ops = collections.Counter()
for line in lines[start + 1:end]:
    line = line.strip()
    if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
        continue
    ops[line.split()[0]] += 1
groups = collections.Counter()
for op, c in ops.items():
    if op.startswith('v_pk'): groups['v_pk'] += c
    elif re.match(r"v_(fma|mul|add|sub|mac|fmac|mad)_f", op): groups['v_fp'] += c
    elif op.startswith('v_'): groups['v_other'] += c
    elif op.startswith('ds_'): groups['ds'] += c
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): groups['vmem'] += c
    elif op.startswith('s_waitcnt'): groups['waitcnt'] += c
    elif op.startswith('s_'): groups['salu'] += c
    else: groups['other'] += c
print("total", sum(ops.values()), dict(groups))
print(ops.most_common(45))
