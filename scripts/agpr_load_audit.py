"""Audit of a -save-temps .s file for the asm global loads that target AGPRs (f4_load_agpr): hipcc does not count them, so
between such a load and the `s_waitcnt vmcnt(0)` that the kernel issues by hand nothing may read or move its destination
registers.   python scripts/agpr_load_audit.py <file.s> <kernel name substring>"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
start = re.search(r'^(_ZN3lwm\d+%s[0-9A-Za-z_]*):' % name, s, re.M)
body = s[start.end():s.find('.Lfunc_end', start.end())]
pending, bad, loads = set(), [], 0
for line in body.split('\n'):
    t = line.strip()
    if not t or t.startswith((';', '.')) or t.endswith(':'):
        continue
    regs = set()
    for lo, hi in re.findall(r'\ba\[(\d+):(\d+)\]', t):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(r) for r in re.findall(r'\ba(\d+)\b', t))
    if t.startswith('global_load_dwordx4 a['):
        pending |= regs
        loads += 1
        continue
    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        pending.clear()
        continue
    if pending & regs:
        bad.append(t)
print(start.group(1), 'AGPR loads', loads, 'instructions touching a pending destination:', len(bad))
for b in bad[:10]:
    print('   ', b)
sys.exit(1 if bad else 0)
