"""CPU-only tests of the host side that mirrors the reference's interface (no kernels run)."""
import os

import numpy as np
import pytest
import torch

import wav2sleep_amd as W
from oracle import wav2sleep_oracle as O
from tests.golden_util import load
from wav2sleep_amd import api
from wav2sleep_amd.engine import EngineSpec
from wav2sleep_amd.lib import W2SError

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def build(signal_map=SM4, nc=4, dropout=0.1):
    return W.Wav2Sleep(W.SignalEncoders(signal_map, 128, 'gelu', norm='instance', chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), nc)


@pytest.mark.parametrize('signal_map,nc,count', [(SM4, 4, 2948740), ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 3345797), ({'ECG': 'UNI'}, 4, 2233412)])
def test_state_dict_abi_matches_reference_schema(signal_map, nc, count):
    """Key names / shapes / parameter counts of SURVEY.md 8b (probed on the reference)."""
    m = build(signal_map, nc)
    sd = m.state_dict()
    want = O.param_shapes(O.ModelConfig(signal_map=signal_map, num_classes=nc))
    assert set(sd) == set(want)
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in m.parameters()) == count
    m.load_state_dict(O.make_state_dict(O.ModelConfig(signal_map=signal_map, num_classes=nc), seed=1), strict=True)
    assert m.valid_signals == list(signal_map)
    assert m.signal_encoders.causal is False and m.num_classes == nc and m.feature_dim == 128


def test_no_cpu_fallback():
    m = build()
    with pytest.raises(W2SError):
        m({'ECG': torch.zeros(1, 1024)})


def test_error_conventions():
    with pytest.raises(ValueError):
        W.SignalEncoders({'XYZ': 'XYZ'}, 128, 'gelu')
    with pytest.raises(ValueError):
        W.SignalEncoders({'ECG': 'ECG'}, 128, 'tanh')
    with pytest.raises(ValueError):
        EngineSpec(signal_map={'BAD': 'BAD'})
    with pytest.raises(ValueError):
        EngineSpec(signal_map={'ECG': 'ECG'}, feature_dim=64, mixer_nhead=4)
    spec = build({'EOG-L': 'E', 'EOG-R': 'E'}, 5).spec()
    assert spec.channels('E') == [16, 16, 32, 32, 64, 64, 128, 128, 128, 128]
    assert build().spec().channels('ABD') == [16, 16, 32, 32, 64, 64]


def test_scheduler_and_stats_match_reference_goldens():
    g = load('misc')
    for k, v in zip(g['lr_steps'], g['lr_values']):
        assert W.exp_warmup_lr(int(k)) == pytest.approx(float(v), rel=1e-12)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sched = W.ExpWarmUpScheduler(opt, lr_max=1e-3, warmup_steps=2000, tau=10000)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step(); sched.step()
    np.testing.assert_allclose(lrs, g['lr_values'][:5], rtol=1e-12)
    assert W.cohens_kappa(g['cm'], 4) == pytest.approx(float(g['kappa']), rel=1e-12)
    assert W.confusion_accuracy(g['cm']) == pytest.approx(float(g['acc']), rel=1e-12)


def test_masker_invariants_and_errors():
    torch.manual_seed(0)
    B = 256
    x = {s: torch.randn(B, 8) for s in SM4}
    x['ECG'][:32] = float('-inf'); x['PPG'][32:64] = float('-inf'); x['ABD'][64:96] = float('-inf')
    avail = torch.stack([~torch.isinf(v[:, 0]) for v in x.values()], -1)
    masker = W.SignalMasker({'ABD': 0.7, 'THX': 0.7, 'ECG': 0.5, 'PPG': 0.1}, backups=['ECG', 'PPG'])
    keeps = []
    for _ in range(40):
        xm = masker({k: v.clone() for k, v in x.items()})
        keep = torch.stack([~torch.isinf(v[:, 0]) for v in xm.values()], -1)
        assert not (keep & ~avail).any() and keep.any(-1).all()
        keeps.append(keep.float())
    rate = torch.stack(keeps).mean((0, 1))  # keep rates ~ (1-p) * availability (+ backups), as the reference's own draws
    ref = torch.tensor(load('misc')['masker_keep']).float().mean((0, 1))
    assert torch.allclose(rate, ref, atol=0.08), (rate, ref)
    bad = {s: torch.full((2, 8), float('-inf')) for s in SM4}
    with pytest.raises(ValueError):
        masker(bad)
    x2 = {k: v[:4].clone() for k, v in x.items()}
    before = {k: v.clone() for k, v in x2.items()}
    W.invert_signals(x2)
    for k in x2:
        fin = torch.isfinite(before[k])
        ratio = (x2[k][fin] / before[k][fin])
        assert torch.all((ratio - 1).abs().lt(1e-6) | (ratio + 1).abs().lt(1e-6))


def test_target_instantiation_of_reference_config(tmp_path):
    cfg = {'_target_': 'wav2sleep.models.wav2sleep.Wav2Sleep', 'num_classes': 4,
           'signal_encoders': {'_target_': 'wav2sleep.models.wav2sleep.SignalEncoders', 'signal_map': {'ECG': 'ECG', 'THX': 'THX'},
                               'feature_dim': 128, 'activation': 'gelu', 'norm': 'instance', 'causal': False, 'chunk_causal': False,
                               'initial_channels': 16, 'max_channels': 128, 'output_norm': False, 'use_residual': True},
           'epoch_mixer': {'_target_': 'wav2sleep.models.wav2sleep.MultiModalAttentionEmbedder', 'feature_dim': 128, 'dropout': 0.1,
                           'activation': 'gelu', 'layers': 2, 'dim_ff': 512, 'nhead': 8},
           'sequence_mixer': {'_target_': 'wav2sleep.models.wav2sleep.SequenceCNN', 'feature_dim': 128, 'dropout': 0.1, 'activation': 'gelu',
                              'norm': 'layer', 'causal': False, 'num_layers': 2, 'kernel_size': 7, 'num_dilations': 6}}
    m = api.instantiate(cfg)
    assert isinstance(m, W.Wav2Sleep) and m.valid_signals == ['ECG', 'THX']
    import yaml
    (tmp_path / 'config.yaml').write_text(yaml.safe_dump(cfg))
    torch.save({'model.' + k: v for k, v in m.state_dict().items()}, tmp_path / 'state_dict.pth')
    m2 = W.load_model(str(tmp_path), device='cpu')
    assert not m2.training and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with pytest.raises(FileNotFoundError):
        W.load_model(str(tmp_path / 'nope'), device='cpu')
    with pytest.raises(ValueError):
        api.instantiate({'_target_': 'os.system'})


def test_ema_callback_and_checkpoint_api_surface():
    """trainer/callbacks.py:33-34 argument check, hook names and state-dict schema (no device work on CPU)."""
    import inspect
    import wav2sleep_amd as W
    with pytest.raises(ValueError):
        W.EMACallback(decay=1.5)
    cb = W.EMACallback(decay=0.999, start_step=10, device='cpu')
    for hook in ('setup', 'on_train_batch_end', 'on_validation_epoch_start', 'on_validation_epoch_end', 'on_test_epoch_start',
                 'on_test_epoch_end', 'on_train_end', 'state_dict', 'load_state_dict'):
        assert callable(getattr(cb, hook))
    assert cb.state_dict() == {'ema_state_dict': None, 'step_count': 0}
    cb.on_train_batch_end(None, None, None, None, 0)          # before setup(): counts the step, touches nothing
    assert cb.state_dict()['step_count'] == 1 and not cb._should_update()
    assert list(inspect.signature(W.EMACallback.__init__).parameters) == ['self', 'decay', 'start_step', 'device']


# ---- parquet in -> .preds.csv out (data/dataset.py:132-183, api.py:193-222); host logic only ------------
def _write_recording(path, epochs, cols=('ECG', 'THX'), labels=True, datetime_index=False, seed=0):
    """A model-ready parquet file as preprocessing.py leaves it: one row per timestamp of the fastest signal, slower
    signals and the 30-s stage labels NaN elsewhere."""
    import pandas as pd
    from wav2sleep_amd.settings import COLS_TO_SAMPLES_PER_EPOCH as SPE
    rng = np.random.default_rng(seed)
    frames = []
    for c in cols:
        n = epochs * SPE[c]
        t = np.arange(n) * (30.0 / SPE[c])
        frames.append(pd.Series(rng.standard_normal(n).astype(np.float32) * 3 + 1, index=t, name=c))
    if labels:
        st = rng.integers(0, 5, epochs).astype(np.float64)
        st[1] = np.nan                                       # an unscored epoch: dropna() would shorten the column ...
        frames.append(pd.Series(st, index=np.arange(epochs) * 30.0 + 1e-3, name='Stage').fillna(7.0))   # ... so mark it out-of-map
    df = pd.concat(frames, axis=1).sort_index()
    df.index.name = 'Timestamp'
    if datetime_index:
        df.index = pd.Timestamp('2024-01-01 22:00:00') + pd.to_timedelta(df.index, unit='s')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    df.to_parquet(path)
    return df


def test_parquet_dataset_matches_reference_contract(tmp_path):
    import os as _os
    fp = str(tmp_path / 'site' / 'rec1.parquet')
    df = _write_recording(fp, epochs=5)
    ds = W.ParquetDataset([fp], columns=['ECG', 'PPG', 'THX'], num_classes=4, require_labels=True, max_length_hours=None)
    x, y = ds[0]
    assert list(x) == ['ECG', 'THX', 'PPG'] and x['ECG'].shape == (5 * 1024,) and x['THX'].shape == (5 * 256,)
    assert torch.isinf(x['PPG']).all() and x['PPG'].shape == (5 * 1024,)                       # absent column: -inf (dataset.py:170-173)
    ecg = torch.from_numpy(df['ECG'].dropna().values).float()
    torch.testing.assert_close(x['ECG'], O.zscore_normalize(ecg))                               # host z-score == oracle restatement
    stages = df['Stage'].dropna().values
    want = torch.tensor([{0: 0, 1: 1, 2: 1, 3: 2, 4: 3}.get(int(s), -1) for s in stages]).float()
    assert torch.equal(y, want) and y[1] == -1
    raw = W.ParquetDataset([fp], columns=['ECG', 'THX'], normalize_on_device=True)[0][0]
    assert torch.equal(raw['ECG'], ecg)                                                         # device mode hands over raw samples
    short = W.ParquetDataset([fp], columns=['ECG'], max_length_hours=0)                         # truncation to max_length_epochs
    assert short[0][0]['ECG'].numel() == 0 and short[0][1].numel() == 0
    with pytest.raises(ValueError):
        W.ParquetDataset([fp], columns=['EEG'])
    with pytest.raises(ValueError):
        W.ParquetDataset([fp], columns=['PPG'])[0]                                              # no relevant column in the file
    with pytest.raises(ValueError):
        W.load_dataset(str(tmp_path / 'empty'), ['ECG'])
    causal = W.ParquetDataset([fp], columns=['ECG'], require_labels=False, causal=True)[0][0]['ECG']   # online EMA normalisation (dataset.py:165-167)
    want, _ = O.causal_rolling_normalize(ecg.numpy(), 1024 / 30.0, tau_seconds=900.0, baseline_tau_seconds=120.0, min_sigma=0.1)
    torch.testing.assert_close(causal, torch.from_numpy(want).float(), rtol=1e-5, atol=1e-6)


def test_save_predictions_tree_timestamps_and_overwrite(tmp_path):
    import pandas as pd
    src, out = tmp_path / 'pq', tmp_path / 'out'
    _write_recording(str(src / 'a' / 'r1.parquet'), epochs=4, seed=1)
    _write_recording(str(src / 'b' / 'c' / 'r2.parquet'), epochs=4, labels=False, datetime_index=True, seed=2)
    ds = W.load_dataset(str(src), ['ECG', 'THX'], num_classes=4, max_length_hours=10)
    assert sorted(ds.files) == sorted([str(src / 'a' / 'r1.parquet'), str(src / 'b' / 'c' / 'r2.parquet')]) and ds.normalize_on_device
    preds = torch.tensor([[0, 1, 2, 3], [3, 2, 1, 0]])
    labels = torch.tensor([[0., 1., -1., 2.], [1., 1., 1., 1.]])
    W.save_predictions(preds, str(src), str(out), ds, labels=labels)
    i1 = ds.files.index(str(src / 'a' / 'r1.parquet'))
    t1 = pd.read_csv(out / 'a' / 'r1.preds.csv')
    assert list(t1.columns) == ['Timestamp', 'Pred', 'Stage'] and list(t1['Timestamp']) == [30.0, 60.0, 90.0, 120.0]   # api.py:212
    assert list(t1['Pred']) == preds[i1].tolist() and list(t1['Stage']) == labels[i1].tolist()
    t2 = pd.read_csv(out / 'b' / 'c' / 'r2.preds.csv', index_col=0, parse_dates=True)
    assert t2.index[0] == pd.Timestamp('2024-01-01 22:00:30') and t2.index[-1] == pd.Timestamp('2024-01-01 22:02:00')
    W.save_predictions(preds + 1, str(src), str(out), ds, labels=None)                           # existing files are kept ...
    assert list(pd.read_csv(out / 'a' / 'r1.preds.csv')['Pred']) == preds[i1].tolist()
    W.save_predictions((preds + 1) % 4, str(src), str(out), ds, labels=None, overwrite=True)     # ... unless overwrite
    t1 = pd.read_csv(out / 'a' / 'r1.preds.csv')
    assert list(t1.columns) == ['Timestamp', 'Pred'] and list(t1['Pred']) == ((preds[i1] + 1) % 4).tolist()
    with pytest.raises(ValueError):
        W.predict_on_folder(str(src), str(out))                                                  # neither model nor model_folder


def test_exported_scheduler_and_lr_state_is_what_torch_holds_after_k_steps():
    """after k optimiser steps torch's param group and the scheduler hold lr(k + 1) (the scheduler has already stepped), and the
    scheduler state dict has the key set of `ExpWarmUpScheduler.state_dict()`."""
    from types import SimpleNamespace
    from wav2sleep_amd.checkpoint import _scheduler_state
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-4)
    sched = W.ExpWarmUpScheduler(opt, lr_max=1e-3, warmup_steps=2000, tau=10000.0)
    for _ in range(7):
        p.grad = torch.ones(3); opt.step(); sched.step()
    step = SimpleNamespace(lr_max=1e-3, warmup_steps=2000, tau=10000.0, step_count=7, lr_at=lambda k: W.exp_warmup_lr(k, 1e-3, 2000, 10000.0))
    got, want = _scheduler_state(step), sched.state_dict()
    assert set(got) == set(want)
    assert got['last_epoch'] == want['last_epoch'] == 7 and got['_step_count'] == want['_step_count']
    assert got['_last_lr'] == pytest.approx(want['_last_lr']) and got['_last_lr'][0] == pytest.approx(opt.param_groups[0]['lr'])
    assert got['_last_lr'][0] == pytest.approx(1e-3 * 8 / 2000)


def test_generic_tape_chunks_and_gradient_accumulation():
    """Host logic of the generic path's backward (wav2sleep_amd/generic.py), no kernels: how a contraction is cut into power-of-two blocks, and
    how the tape sums gradients where a tensor fans out -- a fresh buffer is added into in place by a convolution's data gradient (`acc`),
    a buffer that two inputs share (the fan-out of an add) or a view of one never is."""
    from wav2sleep_amd.generic import GenericForward, _chunks
    assert _chunks(16) == [(0, 16)] and _chunks(128) == [(0, 128)] and _chunks(256) == [(0, 128), (128, 128)]
    assert _chunks(48) == [(0, 32), (32, 16)] and _chunks(1024)[-1] == (896, 128) and _chunks(208) == [(0, 128), (128, 64), (192, 16)]
    with pytest.raises(NotImplementedError):
        _chunks(24)
    gf = GenericForward(grad=True)
    gf._add = lambda a, b: a + b            # (the kernel behind it needs a GPU)
    x = torch.zeros(4)
    calls = []

    def conv_like(scale):
        def bw(g, acc=None):
            calls.append(('acc' if acc is not None else 'new', scale))
            if acc is not None:
                acc += scale * g
                return (acc,)
            return (scale * g,)
        return bw
    # y1 = f1(x), y2 = f2(x) (both "convolutions" of the same input), s = y1 + y2, out = view(s)
    y1, y2, s = torch.zeros(4), torch.zeros(4), torch.zeros(4)
    root = torch.zeros(1)
    gx = []
    gf._rec(x, (root,), lambda g: (gx.append(g.clone()) or None,))   # (x is itself an output: its accumulated gradient arrives here)
    gf._rec(y1, (x,), conv_like(2.0), acc_ok=True)
    gf._rec(y2, (x,), conv_like(3.0), acc_ok=True)
    gf._rec(s, (y1, y2), lambda g: (g, g), share='all')
    out = s.view(2, 2)
    gf._rec(out, (s,), lambda g: (g.reshape(4),), share='view')
    g_out = torch.arange(4.0).view(2, 2)
    gf.backward(out, g_out)
    assert torch.equal(gx[0], 5.0 * torch.arange(4.0))
    # the second convolution met a FRESH gradient of x (the first one's output) and added into it; the first one had nothing to add into
    assert calls == [('new', 3.0), ('acc', 2.0)]
    assert torch.equal(g_out, torch.arange(4.0).view(2, 2))   # the caller's gradient (shared by the add's fan-out through a view) was never written
