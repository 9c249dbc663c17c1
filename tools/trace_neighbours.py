#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> for a kernel name pattern: how many launches, their grid sizes, and which kernels run right
before / after them (to find the host call that issues them).   tools/trace_neighbours.py <dir> <pattern>"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
rows = []
for p in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:70]


before, after, grids = collections.Counter(), collections.Counter(), collections.Counter()
idx = [i for i, n in enumerate(names) if pat in n]
for i in idx:
    before[short(names[i - 1]) if i else '-'] += 1
    after[short(names[i + 1]) if i + 1 < len(names) else '-'] += 1
    r = rows[i]
    grids[(r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')))] += 1
print(f'{len(idx)} launches of *{pat}* among {len(rows)} kernels')
print('grid/workgroup:', grids.most_common(8))
print('preceded by:')
for k, v in before.most_common(15):
    print(f'  {v:6d} {k}')
print('followed by:')
for k, v in after.most_common(15):
    print(f'  {v:6d} {k}')
# position profile: index of each matching launch modulo the distance between two adam kernels (one step)
ad = [i for i, n in enumerate(names) if 'adam_kernel' in n]
if len(ad) >= 3:
    print('kernels between successive adam launches:', [b - a for a, b in zip(ad, ad[1:])][:12])
    per = collections.Counter()
    for i in idx:
        k = sum(1 for a in ad if a < i)
        per[k] += 1
    print('matching launches per adam-delimited interval:', sorted(per.items()))
