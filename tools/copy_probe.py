#!/usr/bin/env python3
"""Which host call sites of one eager training step still go through torch's own operators (each one a device-to-device copy --
rocprofv3 shows those as __amd_rocclr_copyBuffer -- or an elementwise kernel of torch) instead of the HIP library?
Runs the bench configuration once under a TorchDispatchMode and tallies every aten operator that touches a device tensor by the
innermost frame of this repository that led to it.

    python tools/copy_probe.py [n_rows]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
import torch                                  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode   # noqa: E402
import bench                                  # noqa: E402

VIEWS = ('view', 'reshape', 'as_strided', 'slice', 'select', 'transpose', 'permute', 'expand', 'unsqueeze', 'squeeze', 't.default',
         'detach', 'alias', 'unbind', 'split', 'narrow', 'unflatten', 'flatten', '_unsafe_view', 'empty', 'sym_', 'is_', 'size',
         'stride', 'storage_offset', 'numel', 'dim', 'lift_fresh', '_local_scalar_dense', 'set_', 'resize_', 'record_stream')


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        short = name.replace('aten.', '')
        if any(short.startswith(v) for v in VIEWS):
            return out
        flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        if torch.is_tensor(out):
            flat.append(out)
        if not any(t.is_cuda for t in flat):
            return out
        own = [f for f in traceback.extract_stack() if ('fwair' in f.filename or '/net/' in f.filename) and 'copy_probe' not in f.filename]
        site = f'{os.path.basename(own[-1].filename)}:{own[-1].lineno} {own[-1].name}' if own else '(outside the package)'
        self.sites[(short, site)] += 1
        return out


def main():
    nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    from fwair import engine as E
    from net.model import AirNet
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    net = AirNet(bench.make_opt(16, 'bf16', 128)).to(dev).train()
    eng = E.TrainEngine(net, lr=2e-4, contrast_loss_weight=0.6, use_graph=False)
    clean, xq, xk = bench.synth_batch(16, 128, 25, 1234, dev)
    for _ in range(2):
        eng.step(xq, xk, clean)
    torch.cuda.synchronize()
    tally = Tally()
    with tally:
        eng.step(xq, xk, clean)
        torch.cuda.synchronize()
    print(f'{"calls":>6s}  aten operator / innermost package frame')
    for (name, site), n in tally.sites.most_common(nrows):
        print(f'{n:6d}  {name:28s} {site}')
    kinds = collections.Counter()
    for (name, _), n in tally.sites.items():
        kinds[name] += n
    print('totals by operator:', dict(kinds.most_common(20)))
    print('all torch operators on device tensors in one step:', sum(kinds.values()))


if __name__ == '__main__':
    main()
