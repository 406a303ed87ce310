#!/usr/bin/env python3
"""Step budget from a tools/trace_step.sh listing: launches and time per group of kernels (the table of DESIGN.md section 4).
usage: python tools/step_budget.py gpurun_out/trace_<tag>/step.txt [...]"""
import re
import sys

GROUPS = [
    ('FFN forward', ('k_ffn_xr', 'k_ffn_xs', 'k_ffn_x32', 'k_ffn1_x64', 'k_ffn2_x64', 'k_ffn_scales', 'k_split_w', 'k_ffn_fwd')),
    ('FFN backward', ('k_ffn1_bwd', 'k_ffn_dw_bwd')),
    ('local mixer forward', ('k_attn_m<', 'k_attn<', 'k_pos_transpose')),
    ('local mixer backward', ('k_attn_bwd', 'k_proj_o2_bwd')),
    ('FFT mixer', ('k_fft',)),
    ('pixel kernels (embed / down / up-fuse / tail, both directions)', ('k_embed', 'k_down', 'k_upfuse', 'k_tail', 'k_upadj')),
    ('data steps', ('k_dstep', 'k_resample')),
    ('weight-gradient / reduce / prep / loss / Adam / fills', ('',)),
]


def main():
    for path in sys.argv[1:]:
        rows = []
        for line in open(path):
            m = re.match(r'\s*([0-9.]+) us\s+gap\s+([0-9.-]+)\s+(.*)', line)
            if m:
                rows.append((float(m.group(1)), m.group(3)))
        tot = sum(t for t, _ in rows)
        print('%s: %d launches, %.3f ms' % (path, len(rows), tot / 1e3))
        acc = {g: [0, 0.0] for g, _ in GROUPS}
        for t, name in rows:
            for g, pats in GROUPS:
                if any(p in name for p in pats):
                    acc[g][0] += 1
                    acc[g][1] += t
                    break
        for g, _ in GROUPS:
            n, t = acc[g]
            print('  %-66s %4d launches  %7.3f ms  %4.1f %%' % (g, n, t / 1e3, 100 * t / tot))


if __name__ == '__main__':
    main()
