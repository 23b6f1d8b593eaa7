"""Instruction mix of a kernel between its s_barrier instructions, from the -save-temps .s of build.py --keep-temps.
Usage: python tools/asm_segments.py <file.s> <mangled-name-substring>"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = re.compile(r"^(\S*%s\S*):[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(sys.argv[2]), re.S | re.M)
for m in pat.finditer(s):
    print(m.group(1))
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    segs = [[]]
    for l in lines:
        segs[-1].append(l)
        if l.startswith('s_barrier'):
            segs.append([])
    for i, sg in enumerate(segs):
        c = Counter()
        for l in sg:
            op = l.split()[0]
            if op.startswith('v_mfma'): c['mfma'] += 1
            elif op.startswith('v_pk'): c['v_pk'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('ds_'): c[op] += 1
            elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
            elif op.startswith('s_nop'): c['nop'] += 1
            elif op.startswith('s_'): c['salu'] += 1
            elif op.startswith(('global', 'buffer', 'flat', 'scratch')): c['vmem'] += 1
            else: c[op] += 1
        print(i, len(sg), dict(c))
