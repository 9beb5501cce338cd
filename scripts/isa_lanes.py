#!/usr/bin/env python3
"""Per kernel of a gfx950 listing: MFMA count, v_readlane / v_writelane (SGPR spills to VGPR lanes) and how many of them sit between the
first and the last MFMA (the K loop).  usage: python scripts/isa_lanes.py csrc/build/<file>.s [name filter]"""
import re, sys
k = None
stats = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        k = m.group(1)
        stats[k] = dict(rd=0, wr=0, mf=0, first=None, last=None, n=0, rl=[])
    if k is None:
        continue
    st = stats[k]
    st['n'] += 1
    if 'v_mfma' in line:
        st['mf'] += 1
        st['first'] = st['first'] or st['n']
        st['last'] = st['n']
    if 'v_readlane_b32' in line or 'v_writelane_b32' in line:
        st['rd' if 'read' in line else 'wr'] += 1
        st['rl'].append(st['n'])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, st in stats.items():
    if st['mf'] == 0 or flt not in k:
        continue
    inside = sum(1 for n in st['rl'] if st['first'] < n < st['last'])
    print(k[:110], 'mfma', st['mf'], 'readlane', st['rd'], 'writelane', st['wr'], 'inside mfma span', inside)
