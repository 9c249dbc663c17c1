"""Two data-parallel ranks of the fused engine for real, rehearsed on ONE GPU: both ranks are pinned to device 0
(FW_DIST_DEVICE) and the gradient exchange runs over gloo (FW_DIST_BACKEND; RCCL refuses two ranks on one device).  Everything
else is the multi-GPU path of bench.py: torchrun rendezvous on 127.0.0.1, replicas broadcast from rank 0, the three-graph capture
(forward + decoder backward | encoder backward | Adam) with the bucketed all-reduce of the decoder's slice launched under the
encoder's backward, the ranks' agreement on the capture fallback, max-over-ranks timing and the one JSON line of rank 0.
tests/test_ddp_cpu.py covers the reducer arithmetic and the sharding logic with gloo on the CPU."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_three_graph_step_on_one_gpu():
    env = dict(os.environ, FW_DIST_BACKEND='gloo', FW_DIST_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
           '--no-cpu-baseline', '--no-profile']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 4
    assert d['config']['hip_graph'] is True, 'the three-graph capture fell back to eager launches'
    assert 'capture failed' not in r.stderr and 'all-reduce will follow the backward pass' not in r.stderr
    assert all(math.isfinite(v) for v in d['loss'].values()) and 0.0 < d['loss']['l1'] < 1.0
    assert d['value'] > 0


def test_plain_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun around it (the way the driver invokes `--gpus 1`): bench.py must start the two
    ranks itself, before any HIP call, and rank 0 must report n_gpus = 2."""
    env = dict(os.environ, FW_DIST_BACKEND='gloo', FW_DIST_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # the data of BASELINE configs[3]: the five tasks (denoise 15 / 25 / 50, derain, dehaze) cycled over a per-rank batch of 5
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '5', '--tasks', 'allinone',
           '--no-cpu-baseline', '--no-profile']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 10
    assert 'configs[2]' in d['config']['workload'] and 'all-in-one' in d['config']['workload']
    assert d['config']['hip_graph'] is True
    assert all(math.isfinite(v) for v in d['loss'].values())
