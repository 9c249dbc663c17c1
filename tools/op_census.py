#!/usr/bin/env python3
"""Census of the PyTorch-side (non-fwair) device ops of one eager training step: which aten ops run, how often, and from
which line of the host code -- used to hunt fills / copies / adds that should be fused into the HIP kernels."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
import bench  # noqa: E402
from fwair import engine as E  # noqa: E402
from net.model import AirNet  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = AirNet(bench.make_opt(16, 'bf16')).to(dev).train()
eng = E.TrainEngine(net, use_graph=False)
clean, xq, xk = bench.synth_batch(16, 128, 25, 1, dev)
for _ in range(2):
    eng.step_eager(xq, xk, clean)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    eng.step_eager(xq, xk, clean)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::stack', 'aten::mul', 'aten::floor',
                   'aten::div', 'aten::clone', 'aten::contiguous', 'aten::bernoulli_', 'aten::uniform_', 'aten::rand', 'aten::sum'):
        site = 'autograd-engine'
        for fr in ev.stack:
            if 'fwair/' in fr or 'net/model' in fr:
                site = fr.split('frequency-wised_all-in-one_image_restoration_model_amd/')[-1]
                break
        shapes = str(ev.input_shapes)[:60]
        cnt[(ev.name, site, shapes)] += 1
for (name, site, shapes), n in cnt.most_common(400):
    print(f'{n:5d}  {name:18s} {site:60s} {shapes}')
