#!/usr/bin/env python3
"""Rate of the device input pipeline: fwair.augment.DeviceBatcher.batch at B = 16, 128x128 crops from 64 resident
256x256 uint8 images, noise sigma = 25 synthesised in-kernel.  Prints one JSON line (profiles/r03_input_pipeline.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import augment as A  # noqa: E402

dev = 'cuda'
rng = np.random.default_rng(4)
imgs = [torch.from_numpy(rng.integers(0, 256, (3, 256, 256), dtype=np.uint8)).to(dev) for _ in range(64)]
out = {}
for name, tasks in (('denoise sigma=25', [25] * 64),
                    ('all-in-one (denoise 15/25/50, derain, dehaze)', (['denoising_15', 'denoising_25', 'denoising_50', 'deraining', 'dehazing'] * 13)[:64])):
    bt = A.DeviceBatcher(imgs, tasks, 128, generator=torch.Generator().manual_seed(5))
    idx = [list(range(k, k + 16)) for k in range(0, 64, 16)]
    g = torch.Generator(device=dev).manual_seed(1)
    for i in range(8):
        bt.batch(idx[i % 4], generator=g)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        bt.batch(idx[i % 4], generator=g)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    out[name] = {'images_per_s': round(16 * n / t_all, 1), 'us_per_batch_of_16': round(t_all / n * 1e6, 1), 'host_issue_us_per_batch': round(t_issue / n * 1e6, 1)}
print(json.dumps({'metric': 'device input pipeline, training samples/s (two degraded + two clean 128x128 crops each)', 'batch': 16, 'results': out,
                  'note': 'one torch.randint + one fw_train_batch launch per batch; no host synchronisation inside (host issue time < total)'}))
