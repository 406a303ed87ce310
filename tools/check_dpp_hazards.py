#!/usr/bin/env python3
"""gfx9 DPP hazard check on compiler output (hipcc -S --cuda-device-only): a VALU instruction that writes a VGPR must be followed by 2 wait states before an
instruction reads that VGPR as its DPP source (src0 of a *_dpp instruction).  The compiler's hazard recogniser inserts them for its own DPP instructions but
does not look inside inline asm -- k_attn_bwd_f.hip's row_newbcast multiply-adds are inline asm.  Prints every violation; exit code 1 if there is one.
usage: python tools/check_dpp_hazards.py file.s [...]"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'^v(\d+)$', tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    bad = 0
    ndpp = 0
    for path in sys.argv[1:]:
        window = []   # (wait states this instruction provides to later ones, set of VGPRs written by VALU)
        for ln, line in enumerate(open(path), 1):
            l = line.split(';')[0].strip()
            if not l or l.startswith('.') or l.endswith(':') or l.startswith(';'):
                if l.endswith(':'):
                    window = []   # label: a join -- be conservative only across straight-line code (the broadcast sources are loop-invariant registers)
                continue
            parts = l.split(None, 1)
            op = parts[0]
            ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
            if '_dpp' in op and len(ops) >= 2:
                ndpp += 1
                src = regs(ops[1].split()[0])
                dist = 0
                for ws, wr in reversed(window):
                    if dist >= 2:
                        break
                    if wr & src:
                        print('%s:%d: DPP source %s written %d wait state(s) earlier: %s' % (path, ln, ops[1].split()[0], dist, l))
                        bad += 1
                        break
                    dist += ws
            if '_dpp' in op:   # second rule: a VALU write of EXEC (v_cmpx*, or a VALU with exec as destination) needs 5 wait states before a DPP instruction
                dist = 0
                for ws, wr in reversed(window):
                    if dist >= 5:
                        break
                    if -1 in wr:
                        print('%s:%d: DPP instruction %d wait state(s) behind a VALU write of EXEC: %s' % (path, ln, dist, l))
                        bad += 1
                        break
                    dist += ws
            if op == 's_nop':
                window.append((int(ops[0], 0) + 1, set()))
            elif op.startswith('v_cmpx') or (op.startswith('v_') and ops and ops[0] == 'exec'):
                window.append((1, {-1}))
            elif op.startswith('v_') and not op.startswith('v_cmp') and not op.startswith('v_readlane') and not op.startswith('v_readfirstlane'):
                window.append((1, regs(ops[0]) if ops else set()))
            else:
                window.append((1, set()))
            window = window[-8:]
    print('%d DPP instructions checked, %d hazards' % (ndpp, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
