#!/usr/bin/env python3
"""Where does a kernel run?  For every dispatch of kernels matching PATTERN in a rocprofv3 *_kernel_trace.csv, count the
(previous kernel, next kernel, grid size) contexts.   tools/trace_context.py DIR PATTERN [top]"""
import collections, csv, glob, re, sys
d, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: re.sub(r'\(anonymous namespace\)::|^void ', '', n).split('(')[0][:70]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if re.search(pat, r['Kernel_Name']):
        p = short(rows[i - 1]['Kernel_Name']) if i else '-'
        n = short(rows[i + 1]['Kernel_Name']) if i + 1 < len(rows) else '-'
        ctx[(p, n, r.get('Grid_Size_X', r.get('Grid_Size', '?')))] += 1
for (p, n, g), c in ctx.most_common(top):
    print(f'{c:6d}  grid {g:>10}  after {p}  before {n}')
