#!/usr/bin/env python3
"""Build-time check for k_hop4b's hand-waited loads (ld_sv_async, BCG_HOP4B_PREO): between their issue and the hand-written
`s_waitcnt vmcnt(0)` that retires them, no instruction may read or copy their destination registers (hipcc believes the
values are there from the start).  usage: tools/check_async_regs.py <device asm from hipcc -S --cuda-device-only>"""
import re
import sys

txt = open(sys.argv[1]).read()
ok = True
for m in re.finditer(r'^(_ZN3bcg\S*k_hop4b[^:\s]*):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    lines = m.group(2).split('\n')
    idx = [i for i, l in enumerate(lines) if 'global_load_dwordx4' in l and ' lds' not in l and 'ASMSTART' in lines[i - 1]]
    if not idx:
        continue
    # the loop body: the last contiguous group of six
    idx = idx[-6:]
    regs = set()
    for i in idx:
        r = re.search(r'v\[(\d+):(\d+)\]', lines[i])
        regs |= set(range(int(r.group(1)), int(r.group(2)) + 1))
    j = idx[-1] + 1
    while j < len(lines) and not ('s_waitcnt vmcnt(0)' in lines[j] and 'ASMSTART' in lines[j - 1]):
        j += 1
    bad = []
    for k in range(idx[0] + 1, j):
        if k in idx:
            continue
        l = lines[k].split(';')[0]
        used = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', l):
            used |= set(range(int(a), int(b) + 1))
        used |= {int(r) for r in re.findall(r'\bv(\d+)\b', l)}
        if used & regs:
            bad.append((k, lines[k].strip()))
    print(m.group(1)[:60], 'async regs', min(regs), '-', max(regs), 'instructions to the wait:', j - idx[-1], 'touched:', len(bad))
    for b in bad[:8]:
        print('   ', b)
    ok = ok and not bad
sys.exit(0 if ok else 1)
