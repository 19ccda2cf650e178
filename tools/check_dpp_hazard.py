#!/usr/bin/env python3
"""Build-time check for the hand-written v_fmac_f64_dpp instructions of k_hop4b (fmac_bcast, inline asm).

gfx9 requires two wait states between a VALU instruction that writes a VGPR and a DPP instruction that reads that VGPR
through the DPP operand (src0), and five after a VALU write of EXEC.  hipcc inserts such waits for its own DPP instructions;
it does not look inside inline asm.  The broadcast operands here are written by ds_read_b128 (no hazard, the compiler's
s_waitcnt orders them) -- unless the register allocator puts a copy (v_mov, v_accvgpr_read) of one right in front of its use.
This script reads the device assembly and fails if, within the two instructions in front of any v_fmac_f64_dpp, a VALU
instruction writes a register of its src0, or within five a v_cmpx / v_writelane... writes EXEC.

usage: tools/check_dpp_hazard.py <device asm from `hipcc -S --cuda-device-only`>   exit status 1 on a violation
"""
import re
import sys


def regs(tok):
    m = re.match(r'-?\|?v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'-?\|?v(\d+)\b', tok)
    return {int(m.group(1))} if m else set()


def main(path):
    lines = [l.split(';')[0].strip() for l in open(path)]
    code = [(i, l) for i, l in enumerate(lines) if l and not l.startswith('.') and not l.endswith(':')]
    n_dpp, bad = 0, []
    for k, (i, l) in enumerate(code):
        if not l.startswith('v_fmac_f64_dpp'):
            continue
        n_dpp += 1
        ops = [o.strip() for o in l.split(None, 1)[1].split(',')]
        src0 = regs(ops[1])
        for back in range(1, 6):
            if k - back < 0:
                break
            j, p = code[k - back]
            if p.startswith('s_nop'):
                break  # (any s_nop the compiler placed here is more than this check asks for)
            if back <= 2 and p.startswith('v_') and not p.startswith('v_fmac_f64_dpp'):
                dst = regs(p.split(None, 1)[1].split(',')[0].strip()) if ' ' in p else set()
                if dst & src0:
                    bad.append((j + 1, p, i + 1, l))
            if p.startswith('v_cmpx') or re.match(r'v_\w+\s+exec', p):
                bad.append((j + 1, p, i + 1, l))
    print(f'{n_dpp} v_fmac_f64_dpp instructions, {len(bad)} with a VALU write of their broadcast operand (or of EXEC) too close in front')
    for b in bad[:10]:
        print('   line %d: %s   ->   line %d: %s' % b)
    if n_dpp == 0:
        print('no v_fmac_f64_dpp found (not a BCAST build?)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
