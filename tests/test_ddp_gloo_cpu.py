"""world_size-2 gloo test of the data-parallel exchange (wav2sleep_amd/ddp.py) on CPU tensors: ranged all-reduce of the
flat gradient with the 1/world factor folded into the loss gradient == DDP mean; packed metric reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wav2sleep_amd.ddp import FlatGradReducer, reduce_metrics


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    n = 1003
    local = torch.randn(n, generator=g)            # this rank's d(mean loss)/d(theta)
    red = FlatGradReducer((local * (1.0 / world)).clone())
    assert red.grad_scale == 1.0 / world
    # ranges in "backward completion order": tail first, then two encoders
    for lo, hi in [(600, n), (0, 250), (250, 600)]:
        red.reduce_range(lo, hi)
    red.wait()
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    ok_grad = torch.allclose(red.flat, want, atol=1e-6)
    cm = torch.tensor([[3 + rank, 1], [0, 5]])
    loss_count = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
    gmean, rmean, cms = reduce_metrics(loss_count, cm)
    ok_m = abs(float(gmean) - (1 * 10 + 2 * 20) / 30) < 1e-9 and abs(float(rmean) - 1.5) < 1e-9 and cms.tolist() == [[7, 2], [0, 10]]
    q.put((rank, bool(ok_grad), bool(ok_m)))
    dist.destroy_process_group()


def test_flat_grad_reducer_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g and m for _, g, m in res), res


def test_single_process_is_identity():
    flat = torch.arange(8.0)
    red = FlatGradReducer(flat.clone())
    red.reduce_range(0, 8); red.wait()
    assert torch.equal(red.flat, flat) and red.grad_scale == 1.0
    g, r, cm = reduce_metrics(torch.tensor([2.0, 4.0]), torch.eye(2, dtype=torch.long))
    assert float(g) == 2.0 and float(r) == 2.0 and cm.tolist() == [[1, 0], [0, 1]]
