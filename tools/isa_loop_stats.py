#!/usr/bin/env python3
"""Instruction-class census of the main loop of one kernel in a hipcc -S listing (largest backward-branch span).
usage: isa_loop_stats.py listing.s mangled_prefix"""
import re, sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith(pref) and l.rstrip().endswith(':') is False and ':' in l)
end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
lines = [l.strip() for l in L[start:end]]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.match(r'^s_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
b = max(loops, key=lambda s: s[1] - s[0])
def cls(op):
    for p, c in (('v_mfma', 'mfma'), ('v_pk', 'valu_pk'), ('v_readlane', 'readlane'), ('v_writelane', 'writelane'), ('v_', 'valu'), ('s_waitcnt', 'wait'), ('s_nop', 'nop'), ('s_', 'salu'),
                 ('ds_', 'lds'), ('buffer_', 'vmem'), ('global_', 'vmem')):
        if op.startswith(p): return c
    return 'other'
c = Counter(); ops = Counter()
for l in lines[b[0]:b[1] + 1]:
    if not l or l[0] in ';.': continue
    op = l.split()[0]; ops[op] += 1; c[cls(op)] += 1
print('loop', b, dict(c))
if len(sys.argv) > 3:
    for k, v in ops.most_common(40): print(v, k)
for l in L[end:end + 120]:
    if re.search(r'NumSgprs|NumVgprs|Occupancy|ScratchSize', l): print(l.strip())
