#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for LDS reads that are waited for one at a time: `ds_read` ... `s_waitcnt lgkmcnt(0)` with no other LDS read in
between, many times in a row.  k_upfuse's bicubic taps were compiled that way (every 16-byte vector into ONE register quad: read, wait, four multiply-adds, read, ...;
128 dependent LDS round trips per thread).  Prints, per kernel, the LDS reads, how many of them are waited for alone, and the longest run of such reads.
usage: python tools/scan_serial_lds.py /tmp/asm/*.s"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n


for path in sys.argv[1:]:
    lines = open(path).read().splitlines()
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^(_Z\w+|k_\w+):\s*(;.*)?$', l)] if m]
    for idx, (st, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        reads = alone = run = best = 0
        pending = 0   # LDS reads since the last full wait
        for l in lines[st:end]:
            t = l.strip()
            if t.startswith('ds_read') or t.startswith('ds_load'):
                reads += 1
                pending += 1
            elif t.startswith('s_waitcnt') and 'lgkmcnt(0)' in t:
                if pending == 1:
                    alone += 1
                    run += 1
                    best = max(best, run)
                elif pending > 1:
                    run = 0
                pending = 0
            elif t.startswith('s_endpgm'):
                break
        if reads >= 32 and alone >= 16:
            print('%-16s reads %5d  waited for alone %5d (%2d %%)  longest run %4d   %s' % (path.split('/')[-1], reads, alone, 100 * alone // reads, best, demangle(name)[:90]))
