#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py -> HBM bytes per launch of the hot kernels.

    tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [commit] [date]

FETCH_SIZE / WRITE_SIZE are reported in units of 1024 bytes by rocprofv3's derived counters; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so it is doubled (MI355X_MICROARCH.md, section HBM).
A pass that produced no rows for a counter is an ERROR (exit 2): a crashed profiling run must not turn into zeros."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not files:
        sys.exit(f'pmc_summary: no *counter_collection.csv under {d!r} -- the {counter} pass did not complete')
    acc = collections.defaultdict(lambda: [0.0, 0])
    rows = 0
    for path in files:
        for r in csv.DictReader(open(path)):
            if r.get('Counter_Name') != counter:
                continue
            rows += 1
            acc[r['Kernel_Name']][0] += float(r['Counter_Value'])
            acc[r['Kernel_Name']][1] += 1
    if rows == 0:
        sys.exit(f'pmc_summary: {len(files)} file(s) under {d!r} hold no {counter} rows -- wrong directory or a failed pass')
    return acc


HOT = re.compile(r'\b(gemm_\w*kernel|attn2?_(?:fwd|bwd)_kernel|dwconv_\w+_kernel|ln_\w+_kernel|leff_\w+_kernel|conv3x3_kernel'
                 r'|slab_reduce_multi_kernel|adam_kernel|ema_kernel)(<.*>)?\(')


def short(name):
    """rocprofv3's demangled kernel name -> the spelling fw_gemm_last_kernel() and bench.py use: template arguments kept, spaces
    dropped, "unsigned short" -> bf16; None for kernels outside the hot list."""
    m = HOT.search(name.replace('(anonymous namespace)::', ''))
    if not m:
        return None
    return (m.group(1) + (m.group(2) or '')).replace('unsigned short', 'bf16').replace(' ', '')


def main():
    if len(sys.argv) < 4:
        sys.exit(__doc__)
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
    for name, (v, n) in fetch.items():
        k = short(name)
        if k:
            agg[k][0] += v; agg[k][2] += n
    for name, (v, n) in write.items():
        k = short(name)
        if k:
            agg[k][1] += v; agg[k][3] += n
    kernels = {}
    for k, (fv, wv, nf, nw) in agg.items():
        if nf == 0 or nw == 0:
            print(f'pmc_summary: {k}: seen in only one pass (fetch launches {nf}, write launches {nw}) -- skipped', file=sys.stderr)
            continue
        kernels[k] = {'launches_fetch_pass': nf, 'launches_write_pass': nw, 'fetch_kib_raw_per_launch': fv / nf,
                      'write_kib_per_launch': wv / nw, 'hbm_bytes_per_launch': int((2.0 * fv / nf + wv / nw) * 1024)}
    doc = {'commit': sys.argv[4] if len(sys.argv) > 4 else None, 'collected': sys.argv[5] if len(sys.argv) > 5 else None,
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of bench.py; bytes = 2 * FETCH_SIZE * 1024 '
                     '(gfx950 correction) + WRITE_SIZE * 1024, per launch', 'kernels': kernels}
    json.dump(doc, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_fetch_pass']):
        print(f"{k:60s} launches {v['launches_fetch_pass']:6d}  read {2.048e-3 * v['fetch_kib_raw_per_launch']:9.2f} MB  "
              f"write {1.024e-3 * v['write_kib_per_launch']:9.2f} MB per launch")


if __name__ == '__main__':
    main()
