#!/usr/bin/env python3
"""Prints VGPR / spill / scratch use of every kernel in a --save-temps gfx950 .s file:
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude --save-temps -c <file.hip>; kernel_resources.py <file>-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re, sys
t = open(sys.argv[1]).read()
names = re.findall(r'\.name:\s+(\S+)', t)
for blk in t.split('  - .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
    n = re.search(r'\.name:\s+(\S+)', blk)
    print((n.group(1) if n else '?')[:70], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
