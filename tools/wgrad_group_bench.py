#!/usr/bin/env python3
"""Where the grouped weight-gradient launch of the B = 16 step spends its time.

One eager training step records the queued products (n, k, m, bias?) of the backward pass; they are then re-created on fresh
operands and timed (a) as the step launches them (one group per tile form) and (b) CLASS BY CLASS (all products of one (n, k, m)
shape as their own group), so that the classes far from their HBM / MFMA roofline show.

    python tools/wgrad_group_bench.py [--chunk 4096] [--classes]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
import bench  # noqa: E402
from fwair import engine as E, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--classes', action='store_true')
ap.add_argument('--chunk', type=int, default=0)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
from net.model import AirNet  # noqa: E402

torch.manual_seed(1234)
net = AirNet(bench.make_opt(16, 'bf16', 128, 'Uformer', None)).to(dev).train()
eng = E.TrainEngine(net, lr=2e-4, contrast_loss_weight=0.6, use_graph=False)
clean, xq, xk = bench.synth_batch(16, 128, 25, 1234, dev)
eng.step_eager(xq, xk, clean)
rec = []
orig = ops._launch_group


def spy(work, tile=128):
    rec.append((tile, [(n, k, m, db is not None, g.stride(0), x.stride(0)) for (g, x, n, k, m, dw, db) in work]))
    return orig(work, tile)


ops._launch_group = spy
eng.step_eager(xq, xk, clean)
torch.cuda.synchronize()
ops._launch_group = orig
del eng, net
torch.cuda.empty_cache()
if args.chunk:
    ops._GROUP_CHUNK = args.chunk


def materialise(probs):
    work = []
    for n, k, m, has_b, ldg, ldx in probs:
        g = (torch.randn(m, ldg, device=dev) * 0.5).to(torch.bfloat16)[:, :n]
        x = (torch.randn(m, ldx, device=dev) * 0.5).to(torch.bfloat16)[:, :k]
        work.append((g, x, n, k, m, torch.zeros(n, k, device=dev), torch.zeros(n, device=dev) if has_b else None))
    return work


_fire = ops._group_fire
_ev = []


def _timed_fire(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _fire(*a)
    e1.record()
    _ev.append((e0, e1))


ops._group_fire = _timed_fire


def timed(work, tile, reps=3):
    """mean GPU time of the launch itself (events around fw_wgrad_group; the host-side list building stays outside)"""
    for i in range(reps + 1):
        if i == 1:
            _ev.clear()
        orig(work, tile)
        ops._pending.clear()                                 # the slab fold is not part of this measurement
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in _ev) * 1e-3 / reps


def price(probs):
    fl = sum(2.0 * n * k * m for n, k, m, *_ in probs)
    by = sum((n + k) * m * 2 + n * k * 4 for n, k, m, *_ in probs)
    return fl, by


print(f'chunk = {ops._GROUP_CHUNK}')
for tile, probs in rec:
    fl, by = price(probs)
    t = timed(materialise(probs), tile)
    print(f'tile {tile}: {len(probs)} products  {t * 1e3:.3f} ms   {fl / t / 1e12:.1f} TFLOP/s  {by / t / 1e9:.0f} GB/s   '
          f'roofline {max(fl / 2.5e15, by / 8e12) * 1e3:.3f} ms')
    if not args.classes:
        continue
    cls = collections.Counter((n, k, m, b, lg, lx) for n, k, m, b, lg, lx in probs)
    rows = []
    for key, cnt in cls.items():
        ps = [key] * cnt
        fl, by = price(ps)
        t = timed(materialise(ps), tile)
        rows.append((t, key, cnt, fl, by))
    print(f'  {"n":>5s} {"k":>5s} {"m":>7s} bias  ldg  ldx  count     ms   TFLOP/s    GB/s  roofline_ms')
    for t, (n, k, m, b, lg, lx), cnt, fl, by in sorted(rows, reverse=True):
        print(f'  {n:5d} {k:5d} {m:7d} {int(b):4d} {lg:4d} {lx:4d} {cnt:6d} {t * 1e3:6.3f} {fl / t / 1e12:9.1f} {by / t / 1e9:7.0f} '
              f'{max(fl / 2.5e15, by / 8e12) * 1e3:10.3f}')
    print(f'  sum of classes: {sum(r[0] for r in rows) * 1e3:.3f} ms')
