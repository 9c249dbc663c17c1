#!/usr/bin/env python3
"""rocprofv3 --pmc pass of SQ counters -> per hot kernel: where its wave cycles go.

    tools/pmc_sq_summary.py <dir> [min_ms]

SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY
(MI355X_MICROARCH.md, PMC slots); the ACTIVE_INST_* sub-buckets say which pipe the issued instructions kept busy."""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_summary import short  # noqa: E402

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.Counter())
cnt = collections.Counter()
for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            cnt[k] += 1
names = sorted({c for v in acc.values() for c in v})
print('counters:', names)
rows = sorted(acc.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])
print(f'{"kernel":52s} {"launches":>8s} {"wave_cyc/launch":>15s}  ' + '  '.join(f'{n.replace("SQ_", "")[:16]:>16s}' for n in names if n != 'SQ_WAVE_CYCLES'))
for k, v in rows[:40]:
    wc = v['SQ_WAVE_CYCLES'] or 1.0
    print(f'{k:52s} {cnt[k]:8d} {wc / max(cnt[k], 1):15.3e}  ' + '  '.join(f'{v[n] / wc:16.3f}' for n in names if n != 'SQ_WAVE_CYCLES'))
