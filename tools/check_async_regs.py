#!/usr/bin/env python3
"""Build-time check for k_hop4b's hand-waited global loads (ld_sv_async in the PIPE schedule).

hipcc does not see these loads (inline asm): it believes their destination registers hold the values from the moment of
issue.  Correctness therefore needs that between a group's ISSUE and the hand-written wait that RETIRES it no instruction
on the path reads, copies or overwrites those registers.  The kernel brackets every group with asm comments
    ; ASYNC_ISSUE <tag>  ...loads...  ; ASYNC_ISSUED <tag>      and marks the point behind the retiring wait     ; ASYNC_RETIRE <tag>
and this script walks the device assembly of every k_hop4b instantiation from ISSUED along the control flow (unconditional
branches are followed, conditional ones fork) to the RETIRE marker or to a hand-written end-of-step wait (vmcnt(3) or
less): no instruction on any such path may name a destination register of the group, and every path must end in one.

Paths that leave a step without a retire ("lost") FAIL the check, with one allow-list: the walker is not path-sensitive, and
the groups n1 / n2 (the next step's outer rows) are issued under `more` while the loop's exit and the last step's waits sit
under `!more`, so for each of them exactly two statically infeasible paths exist -- one that comes round to the next issue
through the `!more` side of the step's tail, one that leaves the loop behind the issue and runs to s_endpgm.  Those two
(at most one of each kind, groups n1 and n2 only) are reported and tolerated; any further lost path, a lost path of another
group, or a jump table fails the build (tests/test_build_checks.py feeds a synthetic loop-without-retire listing).

usage: tools/check_async_regs.py <device asm from `hipcc -S --cuda-device-only`>   exit status 1 on a violation
"""
import re
import sys


def regs_of(text):
    """registers an instruction names: architectural VGPR n as n, accumulation register n as 1000 + n"""
    used = set()
    for kind, base in (('v', 0), ('a', 1000)):
        for a, b in re.findall(r'\b%s\[(\d+):(\d+)\]' % kind, text):
            used |= set(range(base + int(a), base + int(b) + 1))
        used |= {base + int(r) for r in re.findall(r'\b%s(\d+)\b' % kind, text)}
    return used


def main(path):
    txt = open(path).read()
    ok = True
    seen = 0
    for m in re.finditer(r'^(_ZN3bcg\S*k_hop(?:4b|5)[^:\s]*):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        lines = m.group(2).split('\n')
        label = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\w+:', l)}
        starts = [(i, l.split('ASYNC_ISSUE ')[1].split()[0]) for i, l in enumerate(lines) if 'ASYNC_ISSUE ' in l]
        for i, tag in starts:
            seen += 1
            j = i + 1
            regs = set()
            while j < len(lines) and f'ASYNC_ISSUED {tag}' not in lines[j]:
                d = re.search(r'global_load_dwordx4\s+([va])\[(\d+):(\d+)\]', lines[j])
                if d:
                    base = 1000 if d.group(1) == 'a' else 0
                    regs |= set(range(base + int(d.group(2)), base + int(d.group(3)) + 1))
                j += 1
            todo, done, bad, reached, lost = [j + 1], set(), [], 0, 0
            why = []
            walked = 0
            while todo:
                k = todo.pop()
                while True:
                    if k in done:
                        break
                    done.add(k)
                    if k >= len(lines):
                        lost += 1
                        why.append('ran off the end')
                        break
                    if f'ASYNC_RETIRE {tag}' in lines[k]:
                        reached += 1
                        break
                    # the end-of-step wait on the path that issued nothing (`more` false: the walker cannot know that the
                    # issue and this path exclude each other) retires everything but the stores just the same
                    w = re.search(r's_waitcnt\s+vmcnt\((\d+)\)', lines[k])
                    if w and int(w.group(1)) <= 3 and k > 0 and 'ASMSTART' in lines[k - 1]:
                        reached += 1
                        break
                    if f'ASYNC_ISSUE {tag}' in lines[k]:  # came round the loop without a retire
                        lost += 1
                        why.append(f'reached the next issue at line {k}')
                        break
                    l = lines[k].split(';')[0].strip()
                    walked += 1
                    if l and not l.startswith('.') and not l.endswith(':') and regs_of(l) & regs:
                        bad.append((k, lines[k].strip()))
                    b = re.match(r's_branch\s+(\S+)', l)
                    if b:
                        k = label[b.group(1)]
                        continue
                    c = re.match(r's_cbranch_\w+\s+(\S+)', l)
                    if c:
                        todo.append(label[c.group(1)])
                    if re.match(r's_setpc|s_swappc', l):  # a jump table: cannot follow
                        lost += 1
                        why.append(f'jump table at line {k}')
                        break
                    k += 1
            # statically infeasible paths (the issue sits under `more`, some waits under `!more`) may wander past the loop: they are
            # walked and checked like the others; what must hold is that NO walked instruction touches the registers
            kinds = {'next issue': sum('reached the next issue' in w for w in why), 'off the end': sum('ran off the end' in w for w in why),
                     'jump table': sum('jump table' in w for w in why)}
            allowed = 1 if tag in ('n1', 'n2') else 0
            lost_ok = kinds['jump table'] == 0 and kinds['next issue'] <= allowed and kinds['off the end'] <= allowed
            good = reached > 0 and not bad and regs and lost_ok
            print(m.group(1)[28:62], f'group {tag}: {len(regs)} registers ({"AGPRs" if min(regs) >= 1000 else "VGPRs"} from {min(regs) % 1000}), {walked} instructions walked, paths to the retire:',
                  reached, 'lost:', lost, 'touched in between:', len(bad))
            for b_ in bad[:6]:
                print('     ', b_)
            for w_ in why:
                print('      (path left the step without a retire marker:', w_ + (')' if lost_ok else ') -- NOT on the allow-list: FAIL'))
            ok = ok and bool(good)
    if seen == 0:
        print('no hand-waited load groups found (not a PIPE build?)')
        return 1
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
