"""Pin the CPU oracle (oracle/wav2sleep_oracle.py) against golden vectors produced by the real reference.

CPU-only (`-m "not gpu"`).  fp32 tolerances: forward 2e-5 abs / 1e-4 rel on O(1) activations; grads and
two-step parameters 1e-4 rel (different but equivalent op orders in autograd).
"""
import numpy as np
import pytest
import torch

from oracle import wav2sleep_oracle as O
from tests.golden_util import CASES, assert_summary_close, case_config, checksum, load


def _setup(name):
    signal_map, nc, B, S, missing, wseed, iseed = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=wseed)
    x, y = O.make_inputs(cfg, B, S, seed=iseed, missing=missing)
    return cfg, sd, x, y, (B, S, missing, iseed)


@pytest.mark.parametrize('name', list(CASES))
def test_generators_reproduce(name):
    g = load(name)
    cfg, sd, x, y, _ = _setup(name)
    assert checksum(sd) == pytest.approx(float(g['weights_checksum']), rel=1e-12)
    assert checksum(x) + float(y.sum()) == pytest.approx(float(g['inputs_checksum']), rel=1e-12)


@pytest.mark.parametrize('name', list(CASES))
def test_forward_matches_reference(name):
    g = load(name)
    cfg, sd, x, y, _ = _setup(name)
    taps = {}
    with torch.no_grad():
        logits = O.forward(sd, cfg, x, taps)
    np.testing.assert_allclose(logits.numpy(), g['logits'], rtol=1e-4, atol=2e-5)
    assert np.array_equal(logits.argmax(-1).numpy(), g['pred'])
    np.testing.assert_allclose(taps['mixer'].numpy(), g['mixer'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(taps['seq'].numpy(), g['seq'], rtol=1e-4, atol=2e-5)
    for s in cfg.signal_map:
        got, want = taps[f'z.{s}'].numpy(), g[f'z.{s}']
        assert np.array_equal(np.isinf(got), np.isinf(want))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('name', ['c1_ecg_only', 'c2_four_mod', 'c4_eog_pair', 'c6_causal', 'c7_chunk_causal'])
def test_block_taps_match_reference(name):
    g = load(name)
    cfg, sd, x, y, _ = _setup(name)
    seen = {}
    for sig in x:  # dict order == call order in the reference
        enc = cfg.signal_map[sig]
        taps = {}
        xb = torch.where(torch.isinf(x[sig]), 0.0, x[sig])
        with torch.no_grad():
            O.signal_encoder(sd, cfg, enc, sig, xb, taps)
        j = seen.get(enc, 0)
        seen[enc] = j + 1
        for k, v in taps.items():
            assert_summary_close(v, g[f'tap.{k}.{j}'], rtol=1e-4, atol=2e-5, what=f'{k}.{j}')


@pytest.mark.parametrize('name', list(CASES))
def test_train_steps_match_reference(name):
    g = load(name)
    cfg, sd, x, y, (B, S, missing, iseed) = _setup(name)
    state = {}
    for step in range(2):
        xs, ys = O.make_inputs(cfg, B, S, seed=iseed + 1000 * step, missing=missing)
        if step == 0:
            loss0, _, grads = O.loss_and_grads(sd, cfg, xs, ys)
            for k, gr in grads.items():
                assert_summary_close(gr, g[f'grad0.{k}'], rtol=2e-4, atol=2e-5, what=f'grad0.{k}')
        loss, _, gn, lr = O.train_step(sd, cfg, xs, ys, state)
        assert loss == pytest.approx(float(g[f'loss{step}']), rel=1e-5)
        assert gn == pytest.approx(float(g[f'gnorm{step}']), rel=1e-4)
        assert lr == pytest.approx(float(g[f'lr{step}']), rel=1e-12)
    for k, p in sd.items():
        assert_summary_close(p, g[f'param2.{k}'], rtol=1e-5, atol=2e-7, what=f'param2.{k}')


def test_masked_sample_equals_subset_run():
    """Reference semantics (SURVEY 7 'ragged modality sets'): a sample whose modality is -inf behaves as
    if that modality were not passed at all; the other samples are untouched."""
    cfg, sd, x, y, _ = _setup('c2_four_mod')
    with torch.no_grad():
        full = O.forward(sd, cfg, x)
        sub = O.forward(sd, cfg, {k: v[2:3] for k, v in x.items() if k != 'PPG'})  # sample 2 lacks PPG
    np.testing.assert_allclose(full[2:3].numpy(), sub.numpy(), rtol=1e-4, atol=1e-5)


def test_scheduler_kappa_confusion_masker():
    g = load('misc')
    for k, v in zip(g['lr_steps'], g['lr_values']):
        assert O.exp_warmup_lr(int(k)) == pytest.approx(float(v), rel=1e-12)
    assert O.cohens_kappa(g['cm'], 4) == pytest.approx(float(g['kappa']), rel=1e-12)
    assert O.confusion_accuracy(g['cm']) == pytest.approx(float(g['acc']), rel=1e-12)
    assert O.cohens_kappa(g['cm5'], 5) == pytest.approx(float(g['kappa5']), rel=1e-12)
    # confusion matrix: hand-computed example (parity unpinned by the reference, see oracle header)
    pred = torch.tensor([0, 1, 1, 2, 3, 3, 0])
    true = torch.tensor([0, 1, 2, 2, -1, 3, 1])
    cm = O.confusion_matrix(pred, true, 4)
    want = torch.tensor([[1, 0, 0, 0], [1, 1, 0, 0], [0, 1, 1, 0], [0, 0, 0, 1]])
    assert torch.equal(cm, want)
    # masker invariants observed on the reference's own draws: never un-mask an unavailable channel,
    # always keep >= 1 channel, a lone kept channel that was not drawn must be an available backup.
    avail, keep = g['masker_avail'], g['masker_keep']
    assert not (keep & ~avail[None]).any()
    assert keep.any(-1).all()
    # explicit-draw restatement: all-dropped sample falls back to the chosen backup
    x = {s: torch.randn(3, 4) for s in ('ABD', 'THX', 'ECG', 'PPG')}
    x['ECG'][1] = float('-inf')
    draws = {s: torch.tensor([False, False, True]) for s in x}
    out = O.apply_masker(x, draws, backup_pick=torch.tensor([0, 1, 0]), backups=['ECG', 'PPG'])
    kept = torch.stack([~torch.isinf(out[s][:, 0]) for s in x], -1)
    assert kept.tolist() == [[False, False, True, False], [False, False, False, True], [True, True, True, True]]
    with pytest.raises(ValueError):
        O.apply_masker(x, draws, backup_pick=torch.tensor([0, 0, 0]), backups=['ECG', 'PPG'])


@pytest.mark.parametrize('name', ['ecg_drift', 'abd_spikes', 'eog_default', 'flat_then_active'])
def test_causal_normalisation_matches_reference(name):
    """oracle.causal_rolling_normalize against vectors produced by the reference's data/normalization.py (make_goldens.py:run_causal_norm)."""
    g = load('causal_norm')
    kw = {k.split('.kw.')[1]: float(g[k]) for k in g.files if k.startswith(name + '.kw.')}
    y, m = O.causal_rolling_normalize(g[name + '.x'], sampling_freq=int(g[name + '.spe']) / 30.0, **kw)
    assert np.array_equal(m, g[name + '.mask'])
    np.testing.assert_allclose(y, g[name + '.y'].astype(np.float64), rtol=1e-5, atol=1e-6)


def test_confusion_matrix_against_scikit_learn():
    """`confusion_matrix` is "parity unpinned" (the reference uses torchmetrics' MulticlassConfusionMatrix(ignore_index=-1), trainer/main.py:
    49-59, and torchmetrics is not installable here).  Short of that pin: the same well-defined function from an independent third party --
    scikit-learn -- on random labels with ignored entries, 4 and 5 classes, rows = true, columns = predicted."""
    import numpy as np
    sk = pytest.importorskip('sklearn.metrics')
    rng = np.random.default_rng(7)
    for nc in (4, 5):
        for n in (1, 37, 5000):
            true = rng.integers(-1, nc, size=n)
            pred = rng.integers(0, nc, size=n)
            got = O.confusion_matrix(torch.from_numpy(pred), torch.from_numpy(true), nc).numpy()
            keep = true != -1
            want = sk.confusion_matrix(true[keep], pred[keep], labels=list(range(nc))) if keep.any() else np.zeros((nc, nc), dtype=np.int64)
            assert np.array_equal(got, want), (nc, n)
            assert got.sum() == int(keep.sum())
