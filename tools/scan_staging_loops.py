#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for loops that keep a global load, a full vector-memory wait and a store in ONE loop body: a staging
loop the compiler did not turn into "request everything, then store" is one dependent HBM round trip per trip (lesson 4; k_upfuse had
one for four rounds).  usage: python tools/scan_staging_loops.py /tmp/asm/*.s"""
import re
import sys


def scan(path):
    func = None
    lines = open(path).read().splitlines()
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            func = m.group(1)
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:   # backward branch = loop
            body = lines[labels[m.group(1)]:i]
            loads = [b for b in body if re.search(r'\b(global_load|buffer_load|flat_load)', b)]
            waits0 = [b for b in body if re.search(r's_waitcnt.*vmcnt\(0\)', b)]
            if loads and waits0 and len(body) < 400:
                print('%s: %s loop %s: %d lines, %d loads, %d vmcnt(0) waits' % (path.split('/')[-1], func, m.group(1), len(body), len(loads), len(waits0)))


for p in sys.argv[1:]:
    scan(p)
