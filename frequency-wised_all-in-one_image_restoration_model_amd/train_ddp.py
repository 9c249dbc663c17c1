#!/usr/bin/env python3
"""Throughput training driver beside the reference's train.py.

Same schedule as train.py: `--epochs_encoder` epochs of encoder-only contrastive training (train.py:82-86), then the
full loss l1 + w * contrast (:87-92); the learning-rate rule of :142-149; the log lines of :98-117 (train.log,
options.log); the final checkpoint `ckpt/epoch_<E>.pth` = net.state_dict() (:120-129).  Beyond the reference: one
process per GPU under torchrun (batch-sharded replicas, RCCL gradient all-reduce), the fused HIP-graph engine,
per-epoch checkpoints and resume (`--save_every`, `--resume`; optimizer moments travel in `epoch_<E>.opt.pth`).

    python train_ddp.py --de_type denoising_15 denoising_25 denoising_50 --degradation_embedding_method all_3_bands \\
        --contrast_loss_weight 0.6 --compute_dtype bf16 [--per_gpu_batch 16] [--synthetic_steps 100]
    python -m torch.distributed.run --nproc-per-node 8 train_ddp.py ...

Data: the reference's `utils.dataset_utils.TrainDataset` when that package is importable (put the reference root on
PYTHONPATH behind this directory) -- each rank draws its own shard; `--synthetic_steps N` trains on N synthetic batches
per epoch instead (no dataset on disk needed).
"""
import argparse
import os
import sys
import time

_own = argparse.ArgumentParser(add_help=False)
_own.add_argument('--per_gpu_batch', type=int, default=0, help='samples per GPU and step (default: len(de_type), the reference batch)')
_own.add_argument('--synthetic_steps', type=int, default=0, help='train on this many synthetic batches per epoch')
_own.add_argument('--resume', type=str, default='', help='checkpoint epoch_<E>.pth to continue from (its .opt.pth beside it)')
_own.add_argument('--save_every', type=int, default=0, help='also checkpoint every this many epochs')
_own.add_argument('--no_graph', action='store_true', help='eager launches instead of HIP-graph replay')
_own.add_argument('--no_eval', action='store_true', help='skip the per-epoch evaluation / results.log of train.py:131-139')
_ARGS, _rest = _own.parse_known_args()
sys.argv = [sys.argv[0]] + _rest                       # option.py parses sys.argv at import (reference option.py:3)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL across processes needs it on this driver
import torch                                            # noqa: E402

from fwair import engine as E                           # noqa: E402
from fwair.synthetic import synth_batch, synth_task_batch          # noqa: E402
from net.model import AirNet                            # noqa: E402
from option import options as opt                       # noqa: E402


def lr_for_next_epoch(epoch):
    """train.py:142-149, applied at the end of `epoch`."""
    if epoch <= opt.epochs_encoder:
        return opt.lr * (0.1 ** (epoch // 60))
    return 0.0001 * (0.5 ** ((epoch - opt.epochs_encoder) // 125))


def sigma_of(task):
    return int(task.split('_')[1]) if task.startswith('denoising_') and task.split('_')[1].isdigit() else 25


_LOADER = []                  # (dataset, sampler, loader): built once -- a DataLoader with `num_workers` processes per epoch is a fork storm


def batches(epoch, rank, world, B, dev):
    if _ARGS.synthetic_steps > 0:
        sig = [sigma_of(t) for t in opt.de_type] or [25]
        mixed = any(not t.startswith('denoising_') or t.endswith('_0') for t in opt.de_type)
        for i in range(_ARGS.synthetic_steps):
            seed = 1234 + rank + 1000 * (epoch * _ARGS.synthetic_steps + i)
            if mixed:                                    # BASELINE configs[2]: the task changes from sample to sample (dataset_utils.py:99)
                clean, d1, d2 = synth_task_batch(B, opt.patch_size, list(opt.de_type), seed, dev)
            else:
                clean, d1, d2 = synth_batch(B, opt.patch_size, sig[i % len(sig)], seed, dev)
            yield d1, d2, clean
        return
    if not _LOADER:
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        from utils.dataset_utils import TrainDataset    # the reference's dataset (needs its root on PYTHONPATH and the data on disk)
        ds = TrainDataset(opt)
        sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
        _LOADER.append((ds, sampler, DataLoader(ds, batch_size=B, sampler=sampler, pin_memory=True, drop_last=True,
                                                num_workers=opt.num_workers, persistent_workers=opt.num_workers > 0)))
    _, sampler, loader = _LOADER[0]
    sampler.set_epoch(epoch)                             # a new shuffle per epoch, same worker pool
    for (_, d1, d2, c1, _) in loader:
        yield d1.to(dev, non_blocking=True), d2.to(dev, non_blocking=True), c1.to(dev, non_blocking=True)


def evaluate_tasks(net, epoch, dev, results):
    """train.py:131-139: after every phase-2 epoch, `<E> Epochs Results:` and one `task: PSNR/SSIM` line per test task in results.log.
    The tiles are restored by fwair.evaluate.tiled_restore (test.py:48-71 on the device, averaging the RESTORED tiles).  With
    `--synthetic_steps` the test images are synthetic too (4 images of 1.5 x patch_size per task); otherwise the reference's
    TestDataset is read.  PSNR and SSIM (utils/val_utils.py:50-66) are computed on the device (fwair.evaluate: fw_ssim7 restates
    skimage's structural_similarity defaults)."""
    from fwair import augment as A
    from fwair.evaluate import psnr, ssim, tiled_restore
    results.write('%s Epochs Results:\n' % str(epoch + 1))
    net.eval()
    for task in opt.test_de_type:
        vals, svals = [], []
        if _ARGS.synthetic_steps > 0:
            S = opt.patch_size * 3 // 2
            clean, _, _ = synth_batch(4, S, 0, 4321, dev)
            g = torch.Generator(device='cpu'); g.manual_seed(4321)
            cu8 = (clean * 255.0).round().to(torch.uint8)
            for i in range(clean.shape[0]):
                try:
                    deg = A.degrade(cu8[i], task, g).float().div_(255.0)     # 'denoising_bsd68_25' -> sigma 25; deraining; dehazing
                except ValueError:
                    break                                                    # a task without a synthetic stand-in (deblurring)
                rest = tiled_restore(net, deg[None], opt.crop_test_imgs_size)
                vals.append(psnr(rest, clean[i:i + 1])); svals.append(ssim(rest, clean[i:i + 1]))
        else:
            from torch.utils.data import DataLoader
            from utils.dataset_utils import TestDataset
            for (_, inp, cl) in DataLoader(TestDataset(opt, task), batch_size=1, shuffle=False, num_workers=0):
                rest = tiled_restore(net, inp.to(dev), opt.crop_test_imgs_size)
                vals.append(psnr(rest, cl.to(dev))); svals.append(ssim(rest, cl.to(dev)))
        result = 'PSNR/SSIM: %.2f/%.4f' % (sum(vals) / len(vals) if vals else float('nan'), sum(svals) / len(svals) if svals else float('nan'))
        results.write(task + ': ' + ' ' * (25 - len(task)) + result + '\n')
    results.flush()
    net.train()


def _epoch_rendezvous(rank, world, epoch):
    """End-of-epoch meeting point that a long evaluation on rank 0 cannot time out: a counter in the process group's TCP store (no
    RCCL collective is pending while rank 0 runs its eval forwards), then one ordinary barrier once everybody has arrived."""
    store = torch.distributed.distributed_c10d._get_default_store()
    key = f'fw_epoch_{epoch}'
    store.add(key, 1)
    while int(store.add(key, 0)) < world:
        time.sleep(0.05)
    torch.distributed.barrier()


def main():
    rank, local, world = E.init_distributed()
    dev = torch.device('cuda', local if world > 1 else opt.cuda)
    torch.cuda.set_device(dev)
    B = _ARGS.per_gpu_batch or opt.batch_size
    opt.batch_size = B                                   # MoCo queue K = 3 * per-replica batch (net/model.py:35)
    w = opt.contrast_loss_weight if opt.contrast_loss_weight is not None else opt.default_contrast_loss_weight
    if rank == 0:
        os.makedirs(opt.output_path, exist_ok=True)
        os.makedirs(opt.ckpt_path, exist_ok=True)
        with open(opt.output_path + 'options.log', 'w') as f:           # train.py:39-46
            f.write(f"|{'=' * 151}|\n")
            for key, value in opt.__dict__.items():
                f.write(f"|{str(key):>50s}|{str(value):<100s}|\n")
            f.write(f"|{'=' * 151}|\n")
    net = AirNet(opt).to(dev).train()
    freq = None
    if opt.num_frequency_bands_l1 != -1:                     # train.py:69-70: frequency L1 on the (re, im) spectra of the bands
        from net.utils.frequency_decompose import FrequencyDecompose
        freq = (FrequencyDecompose('frequency_decompose', 1. / opt.num_frequency_bands_l1, opt.patch_size, opt.patch_size, inverse=False),
                opt.frequency_l1_loss_weight)
    eng = E.TrainEngine(net, lr=opt.lr, contrast_loss_weight=w, use_graph=not _ARGS.no_graph, freq_l1=freq,
                        grad_wire_dtype=torch.bfloat16 if opt.grad_allreduce_dtype == 'bf16' else torch.float32)
    start = 0
    if _ARGS.resume:
        net.load_state_dict(torch.load(_ARGS.resume, map_location=dev, weights_only=True))
        st = torch.load(_ARGS.resume[:-4] + '.opt.pth', map_location='cpu', weights_only=True)
        eng.load_optimizer_state(st)
        eng.resync()
        start = int(st['epoch']) + 1
        eng.set_lr(lr_for_next_epoch(start - 1))
    log = open(opt.output_path + 'train.log', 'a' if _ARGS.resume else 'w') if rank == 0 else None
    results = open(opt.output_path + 'results.log', 'a' if _ARGS.resume else 'w') if rank == 0 else None      # train.py:36-37

    def save(epoch):
        if rank != 0:
            return
        torch.save(net.state_dict(), opt.ckpt_path + 'epoch_' + str(epoch + 1) + '.pth')                  # train.py:126
        st = eng.optimizer_state()
        st['epoch'] = torch.tensor(epoch)
        torch.save(st, opt.ckpt_path + 'epoch_' + str(epoch + 1) + '.opt.pth')

    for epoch in range(start, opt.epochs):
        t0, n, out = time.time(), 0, None
        for d1, d2, clean in batches(epoch, rank, world, B, dev):
            out = eng.step_phase1(d1, d2) if epoch < opt.epochs_encoder else eng.step(d1, d2, clean)
            n += 1
        if out is not None and rank == 0:
            v = [float(x) for x in out.flatten()]
            if epoch < opt.epochs_encoder:                                # train.py:98-106
                line = 'Epoch (%d)  Loss: contrast_loss:%0.4f\n' % (epoch, v[0])
            else:                                                         # train.py:107-117
                line = 'Epoch (%d)  Loss: l1_loss:%0.4f contrast_loss:%0.4f\n' % (epoch, v[1], v[2])
            dt = time.time() - t0
            print(line.rstrip('\n') + f'   [{n * B * world / max(dt, 1e-9):.1f} images/s]', flush=True)
            log.write(line)
            log.flush()
        if epoch + 1 == opt.epochs or (_ARGS.save_every and (epoch + 1) % _ARGS.save_every == 0):
            save(epoch)
        if epoch >= opt.epochs_encoder and rank == 0 and not _ARGS.no_eval:     # train.py:131-139 (rank 0 only: an eval forward has no collective)
            evaluate_tasks(net, epoch, dev, results)
        if world > 1:
            # the other ranks wait here while rank 0 evaluates: a host-side (gloo-free) wait with no collective timeout -- rank 0
            # publishes the epoch through the store, the others poll it
            _epoch_rendezvous(rank, world, epoch)
        eng.set_lr(lr_for_next_epoch(epoch))
    if log:
        log.close()
        results.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
