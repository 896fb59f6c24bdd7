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


def test_reduce_ranges_partition_the_flat_buffer():
    """Every element of the flat gradient is all-reduced exactly once: the '_tail' range plus one range per encoder are disjoint,
    contiguous and cover the buffer -- for the headline model and for the optional parameters (signal embedding, register tokens,
    output norms, shared encoders, no residual branch)."""
    import wav2sleep_amd as W
    from wav2sleep_amd.ddp import flat_layout, reduce_ranges
    cases = [
        (dict(ABD='ABD', THX='THX', ECG='ECG', PPG='PPG'), {}, {}),
        (dict(ABD='RESP', THX='RESP', ECG='ECG', PPG='PPG'), dict(embed_signals=True, output_norm=True), dict(register_tokens=2)),
        ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, dict(use_residual=False), {}),
        (dict(ECG='UNI'), dict(causal=True), {}),
    ]
    for sm, enc_kw, mix_kw in cases:
        model = W.Wav2Sleep(W.SignalEncoders(sm, 128, 'gelu', norm='instance', chunk_causal=False, **enc_kw),
                            W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8, **mix_kw),
                            W.SequenceCNN(128, dropout=0.1, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4)
        named = list(model.named_parameters())
        layout, total = flat_layout([p.shape for _, p in named])
        assert all(o % 4 == 0 for o, _, _ in layout) and total % 4 == 0
        ranges = reduce_ranges(layout, [n for n, _ in named])
        assert set(ranges) == {'_tail'} | set(sm.values())
        cover = torch.zeros(total, dtype=torch.int32)
        for lo, hi in ranges.values():
            cover[lo:hi] += 1
        assert bool((cover == 1).all()), (sm, int((cover == 0).sum()), int((cover > 1).sum()))
        # the tail (everything outside the encoders) sits behind the encoders: it is final first and reduced first
        assert ranges['_tail'][1] == total and all(hi <= ranges['_tail'][0] for k, (lo, hi) in ranges.items() if k != '_tail')
