"""CPU pins added in round 2: restatements that were "parity unpinned" are now held against vectors produced by the reference itself
(tests/golden/make_goldens_r2.py): the default initialisation under seed 42, the oracle on near-tied default-init logits,
`ParquetDataset` / `_zscore_normalize` on committed parquet files."""
import os

import numpy as np
import pytest
import torch

import wav2sleep_amd as W
from oracle import wav2sleep_oracle as O
from tests.golden_util import GOLDEN_DIR, load

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def default_init_model(dropout=0.1):
    """The build's parameter containers under scripts/config/main.yaml:35's seed, constructed as scripts/config/model/wav2sleep.yaml does."""
    torch.manual_seed(42)
    return W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)


def test_default_init_equals_the_reference_init_under_seed_42():
    """183 tensors, same names, same order, same values as the REFERENCE model built under torch.manual_seed(42) (checksums stored by
    make_goldens_r2.py from the reference's own modules): the containers draw from the RNG exactly as the reference's modules do."""
    g = load('default_init')
    sd = default_init_model().state_dict()
    assert list(sd.keys()) == [str(n) for n in g['names']]
    got_abs = np.array([float(v.double().abs().sum()) for v in sd.values()])
    got_sum = np.array([float(v.double().sum()) for v in sd.values()])
    got_first = np.array([float(v.flatten()[0]) for v in sd.values()])
    np.testing.assert_array_equal(got_abs, g['abs_sums'])
    np.testing.assert_array_equal(got_sum, g['sums'])
    np.testing.assert_array_equal(got_first, g['first'])


@pytest.mark.parametrize('tag,B,S,seed,missing', [('a', 2, 16, 4242, None), ('b', 3, 8, 4243, {'ABD': [0], 'ECG': [1], 'PPG': [2]})])
def test_oracle_on_default_init_matches_reference_logits(tag, B, S, seed, missing):
    """Near-tied logits (top-2 margin down to 3e-3): the oracle reproduces the reference's logits to fp32 round-off and its arg-max exactly."""
    g = load('default_init')
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = {k: v.detach().clone() for k, v in default_init_model().state_dict().items()}
    x, _ = O.make_inputs(cfg, B, S, seed=seed, missing=missing)
    lg = O.forward(sd, cfg, x)
    np.testing.assert_allclose(lg.numpy(), g[f'logits_{tag}'], rtol=1e-5, atol=2e-6)
    assert np.array_equal(lg.argmax(-1).numpy(), g[f'pred_{tag}'])
    assert float(g[f'margin_{tag}'].min()) < 0.02   # the case is what it claims to be


def test_zscore_normalize_matches_reference():
    """data/dataset.py:76-87 run by the reference on finite / constant / -inf / empty / NaN-carrying inputs."""
    g = load('dataset')
    names = sorted({k.split('.', 2)[2] for k in g.files if k.startswith('zs.in.')})
    assert set(names) == {'a', 'const', 'inf', 'empty', 'tiny', 'one_nan'}
    sig = {k: torch.from_numpy(g[f'zs.in.{k}']) for k in names}
    got = W.ParquetDataset._zscore_normalize({k: v.clone() for k, v in sig.items()})
    for k in names:
        np.testing.assert_array_equal(got[k].numpy(), g[f'zs.out.{k}'], err_msg=k)
        if sig[k].numel() and torch.isfinite(sig[k]).all():
            np.testing.assert_allclose(O.zscore_normalize(sig[k]).numpy(), g[f'zs.out.{k}'], rtol=1e-6, atol=1e-7, err_msg=k)


@pytest.mark.parametrize('name,labels', [('full', True), ('no_ppg', True), ('flat_thx', False)])
@pytest.mark.parametrize('nc', [4, 5])
@pytest.mark.parametrize('mlh', [None, 0])
def test_parquet_dataset_matches_reference(name, labels, nc, mlh):
    """`ParquetDataset.__getitem__` (data/dataset.py:132-183) of the reference on the committed parquet files: key order, z-scored
    signals, -inf padding of absent columns, label mapping (4 / 5 classes, out-of-map stage -> -1), truncation."""
    pytest.importorskip('pyarrow')
    g = load('dataset')
    fp = os.path.join(GOLDEN_DIR, 'dataset', name + '.parquet')
    ds = W.ParquetDataset([fp], columns=['ABD', 'THX', 'ECG', 'PPG'], num_classes=nc, require_labels=labels, max_length_hours=mlh)
    x, y = ds[0]
    tag = f'{name}.nc{nc}.mlh{mlh}'
    assert list(x.keys()) == [str(k) for k in g[f'ds.{tag}.keys']]
    for k, v in x.items():
        np.testing.assert_array_equal(v.numpy(), g[f'ds.{tag}.x.{k}'], err_msg=k)
        assert v.dtype == torch.float32
    np.testing.assert_array_equal(y.numpy(), g[f'ds.{tag}.y'])
    assert y.dtype == torch.float32


# ---- generic-path variants: the parameter containers match the reference's modules (keys, order, default initialisation under a seed) ----
def build_variant(name):
    from tests.golden_util import VARIANTS, perturb_state, variant_index
    v = VARIANTS[name]
    torch.manual_seed(4000 + variant_index(name))
    model = W.Wav2Sleep(W.SignalEncoders(**v['enc']), W.MultiModalAttentionEmbedder(**v['mix']), W.SequenceCNN(**v['seq']), num_classes=v['nc'])
    model.load_state_dict(perturb_state(model.state_dict(), seed=77), strict=True)
    return model


def build_ppgnet():
    from tests.golden_util import perturb_state
    torch.manual_seed(4100)
    ppg = W.SleepPPGNet(n_classes=4, feature_dim=128, dropout=0.2, activation='leaky', norm='batch')
    ppg.load_state_dict(perturb_state(ppg.state_dict(), seed=78), strict=True)
    return ppg


@pytest.mark.parametrize('name', ['causality', 'causality_train', 'leaky_auto_rms', 'silu_group', 'relu_nonorm', 'ppgnet'])
def test_variant_state_dicts_equal_the_reference_modules(name):
    """Same keys in the same order and the same values as the REFERENCE modules built under the same seed (BatchNorm / GroupNorm /
    RMS / layer / no norm, every activation, feature_dim 16..64, SleepPPGNet): checksums stored by make_goldens_r2.py."""
    g = load('variants')
    model = build_ppgnet() if name == 'ppgnet' else build_variant(name)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[f'{name}.keys']]
    np.testing.assert_array_equal(np.array([float(t.double().abs().sum()) for t in sd.values()]), g[f'{name}.checksum'])
    if name != 'ppgnet':
        assert not model.fused_ok()          # these run on the generic path
    with pytest.raises(W.lib.W2SError if hasattr(W, 'lib') else Exception):
        model(torch.zeros(1, 1228800)) if name == 'ppgnet' else model({s: torch.zeros(1, 1024) for s in model.valid_signals})   # no CPU path


def test_unknown_norm_and_activation_raise_like_the_reference():
    with pytest.raises(ValueError):
        W.ConvLayer1D(16, 16, norm='spectral')
    with pytest.raises(ValueError):
        W.ConvLayer1D(16, 16, activation='tanh')
    with pytest.raises(ValueError):
        W.wav2sleep.ConvGroupNorm(20, num_groups=8)


def test_save_predictions_writes_the_reference_csv_text(tmp_path):
    """`save_predictions` against the REFERENCE function itself (api.py:193-221, run by make_goldens_r2.py with empty stand-ins for
    the hydra / omegaconf / pyedflib imports of api.py): same tree, same CSV bytes, with and without labels, relative and
    absolute (DatetimeIndex) time stamps."""
    import shutil
    g = load('save_predictions')
    ddir = os.path.join(GOLDEN_DIR, 'dataset')
    src = tmp_path / 'in'
    (src / 'sub').mkdir(parents=True)
    shutil.copy(os.path.join(ddir, 'full.parquet'), src / 'a.parquet')
    shutil.copy(os.path.join(ddir, 'no_ppg.parquet'), src / 'sub' / 'b.parquet')
    shutil.copy(os.path.join(ddir, 'abs_time.parquet'), src / 'sub' / 'c_abs.parquet')
    files = sorted(str(src / f) for f in ('a.parquet', 'sub/b.parquet', 'sub/c_abs.parquet'))
    assert [os.path.relpath(f, src) for f in files] == [str(f) for f in g['files']]
    ds = W.ParquetDataset(files, columns=['ABD', 'THX', 'ECG', 'PPG'], num_classes=4, require_labels=False)
    preds = [torch.from_numpy(g[f'pred{i}']) for i in range(3)]
    labels = [torch.from_numpy(g[f'label{i}']) for i in range(3)]
    for tag, lab in (('with_labels', labels), ('no_labels', None)):
        dst = tmp_path / ('out_' + tag)
        W.save_predictions(preds, str(src), str(dst), ds, labels=lab)
        for fp in files:
            rel = os.path.relpath(fp, src)
            got = open(dst / (os.path.splitext(rel)[0] + '.preds.csv')).read()
            assert got == str(g[f'{tag}.{rel}']), (tag, rel, got)
    lds = W.load_dataset(str(src), ['ECG', 'THX'], num_classes=5, max_length_hours=None, normalize_on_device=False)
    assert [os.path.relpath(f, src) for f in lds.files] == [str(f) for f in g['load_dataset.files']]
    assert list(lds.columns) == [str(c) for c in g['load_dataset.columns']]
    x0, y0 = lds[0]
    assert list(x0.keys()) == [str(k) for k in g['load_dataset.keys0']] and np.array_equal(y0.numpy(), g['load_dataset.y0'], equal_nan=True)


def test_predict_cli_has_the_reference_flags():
    """scripts/predict.py against the parser of the REFERENCE script (captured by make_goldens_r2.py `cli`): same options, destinations,
    types, actions, required flags and defaults -- except --model-folder, whose reference default is a Hugging Face Hub URI (no network
    here: the option must be given)."""
    import argparse
    import importlib.util
    import json
    ref = json.load(open(os.path.join(GOLDEN_DIR, 'predict_cli.json')))
    spec = importlib.util.spec_from_file_location('w2s_predict_cli_flags', os.path.join(os.path.dirname(GOLDEN_DIR), '..', 'scripts', 'predict.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ap = argparse.ArgumentParser()
    for flag, kw in mod.FLAGS:
        ap.add_argument(flag, **kw)
    mine = {a.option_strings[0]: a for a in ap._actions if a.option_strings and a.dest != 'help'}
    assert list(mine) == [o['flags'][0] for o in ref]
    for o in ref:
        a = mine[o['flags'][0]]
        assert a.dest == o['dest'] and bool(a.required) == o['required'] and type(a).__name__ == o['action'] and a.nargs == o['nargs'], o
        assert (getattr(a.type, '__name__', None) if a.type else None) == o['type'], o
        if o['dest'] != 'model_folder':
            assert a.default == o['default'], o
    assert ref[[o['dest'] for o in ref].index('model_folder')]['default'].startswith('hf://') and mine['--model-folder'].default is None
