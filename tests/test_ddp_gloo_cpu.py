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


# ---- world 4, real gradients: "per-rank mean loss, then mean over ranks" (trainer/main.py:162-163 + Lightning DDP's gradient mean) --------------
SM5 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG', 'EOG-L': 'EOG-L'}


def _rank_batch(cfg, rank):
    """Rank r's batch: 2 recordings x 3 epochs; the ranks hold DIFFERENT numbers of valid labels (rank 0 nearly all, rank 3 two), and on rank 2
    the THX rows of every sample are `-inf` (BASELINE configs[4]: a rank whose draw of SignalMasker left one modality fully masked)."""
    from oracle import wav2sleep_oracle as O
    x, y = O.make_inputs(cfg, 2, 3, seed=900 + rank, missing={'THX': [0, 1]} if rank == 2 else ({'ECG': [1]} if rank == 1 else None), frac_unlabelled=0.0)
    keep = [6, 4, 3, 2][rank]
    y.view(-1)[keep:] = -1.0
    return x, y


def _flatten(grads, names, layout, total):
    flat = torch.zeros(total, dtype=torch.float32)
    for (o, n, _), name in zip(layout, names):
        flat[o:o + n] = grads[name].reshape(-1)
    return flat


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import wav2sleep_oracle as O
    from wav2sleep_amd.ddp import flat_layout, reduce_ranges
    cfg = O.ModelConfig(signal_map=SM5, num_classes=4)
    sd = O.make_state_dict(cfg, seed=5)
    names = list(sd)
    layout, total = flat_layout([sd[n].shape for n in names])
    ranges = reduce_ranges(layout, names)
    x, y = _rank_batch(cfg, rank)
    loss, logits, grads = O.loss_and_grads(sd, cfg, x, y)
    # what FusedTrainStep does: 1/world folded into dLoss (w2s_ce_fwd_bwd gscale) => this rank's flat buffer holds grad / world ...
    red = FlatGradReducer(_flatten(grads, names, layout, total) * (1.0 / world))
    # ... the trunk's range when the encoder backward starts, then ONE range over all encoders (FusedTrainStep._enc_range) -- absent or fully
    # masked encoders take part with their zeros: the layout is static
    enc = [r for k, r in ranges.items() if k != '_tail']
    red.reduce_range(*ranges['_tail'])
    red.reduce_range(min(lo for lo, _ in enc), max(hi for _, hi in enc))
    red.wait()
    thx_zero = bool(all(float(grads[n].abs().max()) == 0.0 for n in names if n.startswith('signal_encoders.encoders.THX.')))
    pred, true = logits.argmax(-1).reshape(-1), y.reshape(-1)
    cm = O.confusion_matrix(pred[true >= 0], true[true >= 0].long(), 4)
    count = float((y >= 0).sum())
    gmean, rmean, cms = reduce_metrics(torch.tensor([loss, count]), torch.as_tensor(cm).long())
    q.put((rank, red.flat.numpy().copy(), thx_zero,   # (numpy: pickled by value -- a torch tensor would travel as a file descriptor of a process that has exited)
           float(gmean), float(rmean), cms.tolist()))
    dist.destroy_process_group()


def test_world4_unequal_label_counts_is_mean_of_per_rank_means():
    """Four gloo ranks with 6 / 4 / 3 / 2 valid labels, oracle gradients of the five-modality model (configs[4]); rank 2's THX modality is
    fully masked.  The reduced flat gradient must be mean_r grad(mean loss of rank r) -- what Lightning DDP computes -- and NOT the
    gradient of the mean over all 15 labels; every rank ends with the same buffer; the masked encoder's zeros ride in the static layout;
    the packed metric all-reduce gives both loss conventions and the summed confusion matrix."""
    from oracle import wav2sleep_oracle as O
    from wav2sleep_amd.ddp import flat_layout
    world = 4
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = O.ModelConfig(signal_map=SM5, num_classes=4)
    sd = O.make_state_dict(cfg, seed=5)
    names = list(sd)
    layout, total = flat_layout([sd[n].shape for n in names])
    per_rank, losses, counts, cm_sum = [], [], [], torch.zeros(4, 4, dtype=torch.long)
    xs, ys = [], []
    for r in range(world):
        x, y = _rank_batch(cfg, r)
        loss, logits, grads = O.loss_and_grads(sd, cfg, x, y)
        per_rank.append(_flatten(grads, names, layout, total).double())
        losses.append(loss); counts.append(float((y >= 0).sum()))
        pred, true = logits.argmax(-1).reshape(-1), y.reshape(-1)
        cm_sum += torch.as_tensor(O.confusion_matrix(pred[true >= 0], true[true >= 0].long(), 4)).long()
        xs.append(x); ys.append(y)
    want = sum(per_rank) / world                                                   # DDP: mean over ranks of the per-rank-mean-loss gradient
    pooled = sum(c * g for c, g in zip(counts, per_rank)) / sum(counts)            # gradient of the mean over ALL valid labels: not what DDP does
    assert float((want - pooled).norm() / want.norm()) > 1e-2                      # (the two conventions really differ on this batch)
    for rank, flat, thx_zero, gmean, rmean, cms in res:
        flat = torch.from_numpy(flat)
        assert float((flat.double() - want).norm() / want.norm()) < 1e-5, rank   # (fp32 autograd at another thread count: ~1e-6; the pooled convention is > 1e-2 away)
        assert torch.equal(flat, torch.from_numpy(res[0][1]))                                        # every rank holds the same reduced buffer
        assert thx_zero == (rank == 2)                                             # only rank 2's THX encoder saw no gradient
        assert abs(gmean - sum(l * c for l, c in zip(losses, counts)) / sum(counts)) < 1e-6
        assert abs(rmean - sum(losses) / world) < 1e-6                             # trainer/main.py:165 self.log(..., sync_dist=True)
        assert cms == cm_sum.tolist()
    assert counts == [6.0, 4.0, 3.0, 2.0]
