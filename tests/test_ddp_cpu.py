"""World-size-2 check of the data-parallel path on CPU (gloo): flat parameter storage, in-place gradient
accumulation into the flat buffer, bucketed all-reduce with loss pre-scaling == single-process mean gradient."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd')


def _toy():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 33), torch.nn.GELU(), torch.nn.Linear(33, 5))


def _worker(rank, world, port, wire, out):
    sys.path.insert(0, PKG)
    from fwair.engine import GradAllReducer, attach_flat_grads, flatten_parameters
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net = _toy()
    params = list(net.parameters())
    flat = flatten_parameters(params, torch.device('cpu'))
    assert all(p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + flat.numel() * 4 for p in params)
    g = torch.zeros_like(flat)
    attach_flat_grads(params, g)
    torch.manual_seed(100)
    x, y = torch.randn(8, 12), torch.randn(8, 5)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    red = GradAllReducer(g, bucket_elems=200, wire_dtype=wire)          # several buckets
    for it in range(2):                                                 # second pass checks zero + re-accumulate
        g.zero_()
        loss = torch.nn.functional.l1_loss(net(xs), ys) / world         # pre-scaled by 1/world
        loss.backward()
        assert params[0].grad.data_ptr() == g.data_ptr()                # autograd accumulated in place
        if it == 0:
            red()
        else:                                                           # the two-stage form of the engine: tail range first, head range after
            cut = 136
            wa = red.launch(cut, g.numel())
            wb = red.launch(0, cut)
            GradAllReducer.finish(wa); GradAllReducer.finish(wb)
    if rank == 0:
        torch.save(torch.cat([p.grad.reshape(-1) for p in params]), out)      # per-parameter views (the flat buffer is padded)
    dist.destroy_process_group()


def _run(wire, tmp_path, port):
    out = str(tmp_path / f'g_{port}.pt')
    mp.spawn(_worker, args=(2, port, wire, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    net = _toy()
    torch.manual_seed(100)
    x, y = torch.randn(8, 12), torch.randn(8, 5)
    # mean over ranks of per-shard mean losses == the reference's per-replica semantics averaged by all-reduce
    loss = sum(torch.nn.functional.l1_loss(net(x[r * 4:(r + 1) * 4]), y[r * 4:(r + 1) * 4]) for r in range(2)) / 2
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    return got, ref


def test_allreduce_fp32(tmp_path):
    got, ref = _run(torch.float32, tmp_path, 29611)
    assert torch.allclose(got, ref, atol=1e-6)


def test_allreduce_bf16_wire(tmp_path):
    got, ref = _run(torch.bfloat16, tmp_path, 29612)
    assert torch.allclose(got, ref, atol=2e-2 * ref.abs().max().item())
