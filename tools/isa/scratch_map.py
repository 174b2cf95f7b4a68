#!/usr/bin/env python3
"""Map the scratch_load / scratch_store instructions of an ISA dump (tools/isa/probe.sh ... -gline-tables-only) to source lines."""
import collections
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
files, cur, cnt = {}, None, collections.Counter()
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if re.match(r'\s+scratch_(load|store)', l):
        cnt[(cur, 'load' if 'load' in l else 'store')] += 1
agg = collections.Counter()
for (loc, kind), n in cnt.items():
    agg[loc] += n
for loc, n in agg.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    print(loc, n, "loads", cnt[(loc, 'load')], "stores", cnt[(loc, 'store')])
