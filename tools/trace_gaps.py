#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv of bench.py -> how much of a replayed step the GPU is idle between kernels.

    tools/trace_gaps.py <dir with *_kernel_trace.csv>

Steps are delimited by adam_kernel dispatches (two per step: query encoder, the rest); only the last steps (graph replays) count.
Prints per step: wall (first kernel start -> last kernel end), busy (union of kernel intervals), idle = wall - busy, the number of
kernels, and the histogram of the gaps between consecutive busy intervals."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
ends = adam[1::2]                                            # the second Adam launch closes a step
steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
for a, b in steps[-3:]:
    ks = rows[a:b]
    wall = max(k[1] for k in ks) - ks[0][0]
    busy, gaps, cur_s, cur_e = 0, [], ks[0][0], ks[0][1]
    for s, e, _ in ks[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(k[1] - k[0] for k in ks)
    hist = {}
    for g in gaps:
        key = '<1us' if g < 1000 else '<2us' if g < 2000 else '<4us' if g < 4000 else '<8us' if g < 8000 else '>=8us'
        hist[key] = hist.get(key, 0) + 1
    print(f'kernels {len(ks)}  wall {wall / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms  idle {(wall - busy) / 1e6:.2f} ms  '
          f'sum of kernel times {ksum / 1e6:.2f} ms  gaps {len(gaps)}: {hist}')
