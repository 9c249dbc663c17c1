#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: per-step time of every kernel (steps = number of adam_kernel calls)."""
import csv, glob, re, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
one = [int(r['Calls']) for r in rows if 'slab_reduce_multi_kernel' in r['Name']]          # one launch per backward pass
n = one[0] if one else [int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']][0]
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / n
groups = {}
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'^void ', '', name)
    key = name.split('<')[0].split('(')[0]
    groups[key] = groups.get(key, 0) + float(r['TotalDurationNs']) / 1e6 / n
print(f'steps {n}  kernel time per step {tot:.2f} ms')
print('  '.join(f'{k}={v:.2f}' for k, v in sorted(groups.items(), key=lambda kv: -kv[1])[:14]))
for r in rows[:top]:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'^void ', '', name)[:84]
    print(f"{float(r['TotalDurationNs'])/1e6/n:8.2f} ms/step  calls/step {int(r['Calls'])/n:7.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}")
