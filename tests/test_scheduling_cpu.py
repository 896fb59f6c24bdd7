"""Host logic of round 2 that needs no GPU: the engine's round-robin enqueue (`Engine._interleave`), the trainer's two all-reduce ranges, and
bench.py's bookkeeping helpers (profile-name -> timer-key mapping, kernel families, step traffic from the committed profiles)."""
import contextlib
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wav2sleep_amd import engine as E  # noqa: E402
from wav2sleep_amd.ddp import flat_layout, reduce_ranges  # noqa: E402


class _FakeEngine:
    """Just enough of Engine for _interleave: the two reduction queues."""
    _interleave = E.Engine._interleave

    def __init__(self):
        self._rjobs, self._cjobs = [], []


def _run_interleave(monkeypatch, interleave, with_trunk=True):
    monkeypatch.setattr(torch.cuda, 'stream', lambda st: contextlib.nullcontext())   # no device here: stream contexts are no-ops
    eng = _FakeEngine()
    log = []

    def enc(name, blocks):
        for b in range(blocks):
            eng._rjobs.append((name, b))          # a weight-gradient slab queued by this block
            log.append((name, b))
            yield
        assert [j[0] for j in eng._rjobs] == [name] * blocks, 'a task must only ever see its own queue'
        eng._rjobs = []                           # = _flush_reduce on its stream

    def trunk():
        assert eng._cjobs == ['ln-colsum'], 'the trunk task inherits what the data-gradient chain queued'
        for k in range(2):
            log.append(('trunk', k))
            yield
        eng._cjobs = []

    eng._cjobs = ['ln-colsum']
    tasks = {'ECG': ('s-ecg', [enc('ECG', 3)]), 'ABD': ('s-abd', [enc('ABD', 2), enc('ABD2', 1)])}   # ABD2: a signal sharing ABD's encoder
    eng._interleave(tasks, trunk=('main', trunk()) if with_trunk else None)
    assert eng._rjobs == [] and eng._cjobs == ([] if with_trunk else ['ln-colsum'])
    return log


def test_interleave_round_robin(monkeypatch):
    log = _run_interleave(monkeypatch, True)
    # one block per task and turn, in task order; a second signal of the same encoder runs after the first one on the same stream
    assert log == [('ECG', 0), ('ABD', 0), ('trunk', 0), ('ECG', 1), ('ABD', 1), ('trunk', 1), ('ECG', 2), ('ABD2', 0)]
    assert _run_interleave(monkeypatch, True, with_trunk=False) == [('ECG', 0), ('ABD', 0), ('ECG', 1), ('ABD', 1), ('ECG', 2), ('ABD2', 0)]


def test_encoder_ranges_are_one_contiguous_slice_next_to_the_trunk():
    """FusedTrainStep all-reduces [trunk] early and [all encoders] at the end: the encoder ranges must form one slice that does not
    interleave with the trunk's, for every model variant (incl. shared encoders and the signal embedding)."""
    import wav2sleep_amd as W
    for sm, kw in (({'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}, {}), ({'ABD': 'RESP', 'THX': 'RESP', 'ECG': 'ECG'}, {}),
                   ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, dict(embed_signals=True)), ({'ECG': 'UNI'}, dict(output_norm=True))):
        model = W.Wav2Sleep(W.SignalEncoders(sm, 128, 'gelu', norm='instance', causal=False, chunk_causal=False, **kw),
                            W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                            W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)
        names = [n for n, _ in model.named_parameters()]
        layout, total = flat_layout([tuple(p.shape) for _, p in model.named_parameters()])
        rng = reduce_ranges(layout, names)
        enc = [r for k, r in rng.items() if k != '_tail']
        lo, hi = min(a for a, _ in enc), max(b for _, b in enc)
        tlo, thi = rng['_tail']
        assert hi <= tlo or thi <= lo, (sm, (lo, hi), (tlo, thi))
        assert sum(b - a for a, b in enc) == hi - lo, 'encoder ranges leave a hole or overlap'
        assert (hi - lo) + (thi - tlo) == total


def test_bench_bookkeeping_helpers():
    import bench
    assert bench.pmc_key('conv_wide_kernel<8, 8, 4, 1, 3, 1, 4, 2, 0, 0>') == 'conv_wide_kernel<8, 8, 1, 3, 1>'
    assert bench.pmc_key('conv_wide_np_kernel<8, 8, 4, 1, 3, 1, 4, 2, 0, 0, 0>') == 'conv_wide_kernel<8, 8, 1, 3, 1>'   # (forward entry point, round 4)
    assert bench.pmc_key('wgrad_wide_kernel<4, 4, 2, 5, 3, 2, 4, 2, 2, 2>') == 'wgrad_wide_kernel<4, 4, 2, 5, 3>'
    assert bench.pmc_key('bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0>') == 'bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0>'
    # round 4: the persistent statistics producers carry a trailing FIN template argument that the timer keys do not
    assert bench.pmc_key('bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0, 0, 0>') == 'bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0, 0>'
    assert bench.pmc_key('bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0, 0>') == 'bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0, 0>'
    assert bench.pmc_key('conv_fwd_bf_kernel<1, 1, 4, 1, 3, 1>') == 'conv_fwd_bf_kernel<1, 1, 4, 1, 3>'
    assert bench.pmc_key('conv_wide_kernel<8, 8, 4, 1, 3, 1, 4, 2, 0, 0, 0>') == 'conv_wide_kernel<8, 8, 1, 3, 1>'
    assert bench.pmc_key('bwd_wide_kernel<4, 4, 1, 4, 4, 2, 2, 2, 0, 0, 1>') == bench.pmc_key('bwd_wide_kernel<4, 4, 1, 4, 4, 2, 2, 2, 0, 0>')
    fams = {bench.family_of(k) for k in ('bwd_fused_bf_kernel<1, 1, 4, 1, 0, 0>', 'conv_fwd_bf_kernel<1, 1, 4, 1, 3>', 'conv_wide_kernel<4, 4, 1, 3, 1>',
                                         'wgrad_wide_kernel<4, 4, 1, 4, 3>', 'wgrad_bf_kernel<8, 1, 8, 8, 1, -1, -1>', 'conv_cl_kernel<8, 4, 1, 1, 0, 4, 0, 3, 1>')}
    assert len(fams) == 5 and 'other' not in fams
    t = bench.step_traffic(os.path.join(ROOT, 'profiles'))   # committed PMC bytes/launch x committed launch counts
    assert t is not None and 100e9 < t < 200e9


def test_wave_layout_covers_the_batch_once_and_sizes_the_loss_partials():
    """FusedTrainStep(waves=n): contiguous, non-empty, disjoint sample ranges covering the batch; per-wave loss-partial block counts."""
    from wav2sleep_amd.trainer import wave_layout
    for B in (1, 2, 3, 5, 16, 33):
        for nw in range(1, B + 1):
            bounds, nblk = wave_layout(B, 960, nw)
            assert bounds[0][0] == 0 and bounds[-1][1] == B and len(bounds) == nw
            assert all(a1 == b0 for (_, a1), (b0, _) in zip(bounds, bounds[1:]))
            sizes = [b1 - b0 for b0, b1 in bounds]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
            assert nblk == [(n * 960 + 255) // 256 for n in sizes]
    bounds, nblk = wave_layout(5, 3, 3)
    assert bounds == [(0, 1), (1, 3), (3, 5)] and nblk == [1, 1, 1]
    import pytest
    with pytest.raises(ValueError):
        wave_layout(2, 10, 3)
