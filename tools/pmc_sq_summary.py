#!/usr/bin/env python3
"""rocprofv3 --pmc pass of SQ counters -> per hot kernel: where its wave cycles go, and its MFMA utilisation against chip peak.

    tools/pmc_sq_summary.py <dir> [shader_clock_GHz]

Wave-cycle buckets (quad-cycle units, MI355X_MICROARCH.md "rocprofv3 PMC slots"; each as a FRACTION of SQ_WAVE_CYCLES):
    SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY,
    ACTIVE_INST_* = which pipe the issued instructions kept busy.
MFMA utilisation (column mfma_util): SQ_VALU_MFMA_BUSY_CYCLES counts SHADER CYCLES in which a SIMD's matrix pipe is busy, summed over
the chip's SIMDs (MI355X_MICROARCH.md, per-instruction cycle constants: "= 32 x N_mfma for 32x32x16 bf16") -- it is NOT in the
quad-cycle unit of SQ_WAVE_CYCLES, so dividing one by the other (round 2's table did) is meaningless.  The utilisation is
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x shader clock x 1024 SIMDs)
with the dispatch's own Start / End timestamps from the same pass and the chip's MAXIMUM shader clock (2.4 GHz unless given): a
fraction of the dense MFMA peak the roofline prices against (the clock a profiled pass really holds is lower, 1.9 - 2.0 GHz, so the
fraction of the cycles the pipe was ACTUALLY offered is about 1.2x the figure printed)."""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_summary import short  # noqa: E402

d = sys.argv[1]
clock = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 2.4e9
SIMDS = 256 * 4
acc = collections.defaultdict(lambda: collections.Counter())
cnt = collections.Counter()
dur = collections.Counter()
seen = set()
for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (path, r['Dispatch_Id'])
        if key not in seen:                                   # one duration per dispatch (every counter row repeats the timestamps)
            seen.add(key)
            cnt[k] += 1
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
names = sorted({c for v in acc.values() for c in v} - {'SQ_WAVE_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES'})
print(f'shader clock used for mfma_util: {clock / 1e9:.2f} GHz x {SIMDS} SIMDs; wave-cycle buckets are fractions of SQ_WAVE_CYCLES')
rows = sorted(acc.items(), key=lambda kv: -dur[kv[0]])
print(f'{"kernel":52s} {"launches":>8s} {"us/launch":>10s} {"mfma_util":>10s}  ' + '  '.join(f'{n.replace("SQ_", "")[:16]:>16s}' for n in names))
for k, v in rows[:45]:
    wc = v['SQ_WAVE_CYCLES'] or 1.0
    util = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur[k] * clock * SIMDS) if dur[k] > 0 else 0.0
    print(f'{k:52s} {cnt[k]:8d} {dur[k] / max(cnt[k], 1) * 1e6:10.1f} {util:10.4f}  ' + '  '.join(f'{v[n] / wc:16.3f}' for n in names))
