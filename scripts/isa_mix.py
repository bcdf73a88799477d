"""Instruction mix of the kernels in a device assembly file (hipcc -S --cuda-device-only)."""
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in s.split('.globl')[1:]:
    name = blk.split()[0]
    if pats and not any(p in name for p in pats):
        continue
    if 's_endpgm' not in blk:
        continue
    body = blk[:blk.index('s_endpgm')]
    ins = []
    for l in body.splitlines():
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    print(name, len(ins))
    for k, v in c.most_common(45):
        print('    %-28s %d' % (k, v))
