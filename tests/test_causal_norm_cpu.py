"""The native causal (EMA) normalisation `wav2sleep_amd.inputs.causal_rolling_normalize` -- host code in the library, no GPU --
against vectors from the reference's data/normalization.py (tests/golden/causal_norm.npz), the oracle, and the properties the
reference's own tests/data/test_normalization.py checks: edge cases, type / dtype / device preservation, causality, determinism,
realistic sizes.  Tolerance as in that file: rtol 1e-5."""
import numpy as np
import pytest
import torch

from oracle import wav2sleep_oracle as O
from tests.golden_util import load
from wav2sleep_amd.inputs import causal_rolling_normalize


@pytest.mark.parametrize('name', ['ecg_drift', 'abd_spikes', 'eog_default', 'flat_then_active'])
def test_matches_vectors_from_the_reference_module(name):
    g = load('causal_norm')
    kw = {k.split('.kw.')[1]: float(g[k]) for k in g.files if k.startswith(name + '.kw.')}
    x = g[name + '.x']
    y, m = causal_rolling_normalize(x, sampling_freq=int(g[name + '.spe']) / 30.0, return_outlier_mask=True, **kw)
    assert y.dtype == x.dtype and m.dtype == bool
    np.testing.assert_allclose(y, g[name + '.y'], rtol=1e-5, atol=1e-6)
    assert np.array_equal(m, g[name + '.mask'])


@pytest.mark.parametrize('tau,base,thr', [(900.0, None, 4.0), (300.0, 60.0, 4.0), (60.0, 120.0, 2.5)])
def test_matches_the_oracle_loop(tau, base, thr):
    rng = np.random.default_rng(123)
    x = rng.standard_normal(5000).astype(np.float32)
    x[rng.integers(0, 5000, 25)] += 30.0
    y, m = causal_rolling_normalize(x, sampling_freq=34.0, tau_seconds=tau, baseline_tau_seconds=base, outlier_threshold_sigma=thr,
                                    return_outlier_mask=True)
    yo, mo = O.causal_rolling_normalize(x, 34.0, tau_seconds=tau, baseline_tau_seconds=base, outlier_threshold_sigma=thr)
    np.testing.assert_allclose(y, yo, rtol=1e-5, atol=1e-6)
    assert np.array_equal(m, mo) and m.sum() >= 20


def test_edge_cases():
    assert len(causal_rolling_normalize(np.array([], dtype=np.float32), sampling_freq=34.0)) == 0
    one = causal_rolling_normalize(np.array([1.0], dtype=np.float32), sampling_freq=34.0)
    assert len(one) == 1 and np.isfinite(one[0])
    assert np.all(np.isfinite(causal_rolling_normalize(np.ones(1000, dtype=np.float32) * 5.0, sampling_freq=34.0)))
    short = causal_rolling_normalize(np.random.default_rng(0).standard_normal(10).astype(np.float32), sampling_freq=34.0)
    assert len(short) == 10 and np.all(np.isfinite(short))
    with pytest.raises(ValueError):
        causal_rolling_normalize(np.zeros((4, 4), dtype=np.float32), sampling_freq=34.0)


def test_type_dtype_and_device_are_preserved():
    assert isinstance(causal_rolling_normalize(np.random.randn(1000).astype(np.float32), sampling_freq=34.0), np.ndarray)
    for dtype in (torch.float32, torch.float64):
        s = torch.randn(1000, dtype=dtype)
        r = causal_rolling_normalize(s, sampling_freq=34.0)
        assert isinstance(r, torch.Tensor) and r.dtype == dtype and r.device == s.device
    r, m = causal_rolling_normalize(torch.randn(1000), sampling_freq=34.0, return_outlier_mask=True)
    assert isinstance(m, torch.Tensor) and m.dtype == torch.bool
    e, m = causal_rolling_normalize(torch.zeros(0), sampling_freq=34.0, return_outlier_mask=True)
    assert e.numel() == 0 and m.dtype == torch.bool


def test_no_information_from_the_future():
    rng = np.random.default_rng(789)
    prefix = rng.standard_normal(8000).astype(np.float32)
    a = np.concatenate([prefix, rng.standard_normal(2000).astype(np.float32)])
    b = np.concatenate([prefix, rng.standard_normal(2000).astype(np.float32) * 10])
    ra = causal_rolling_normalize(a, sampling_freq=34.0, baseline_tau_seconds=120.0)
    rb = causal_rolling_normalize(b, sampling_freq=34.0, baseline_tau_seconds=120.0)
    assert np.array_equal(ra[:8000], rb[:8000])


def test_deterministic():
    x = np.random.default_rng(101).standard_normal(2000).astype(np.float32)
    runs = [causal_rolling_normalize(x.copy(), sampling_freq=34.0) for _ in range(4)]
    assert all(np.array_equal(runs[0], r) for r in runs[1:])


@pytest.mark.parametrize('spe', [1024, 256, 4096])
def test_one_hour_of_each_modality(spe):
    x = np.random.default_rng(42).standard_normal(spe * 120).astype(np.float32)
    r = causal_rolling_normalize(x, sampling_freq=spe / 30.0, baseline_tau_seconds=120.0)
    assert r.shape == x.shape and np.all(np.isfinite(r))
    assert abs(np.mean(r)) < 1.0 and 0.1 < np.std(r) < 10.0


def test_dataset_applies_it_per_signal_and_passes_missing_ones_through():
    from wav2sleep_amd.data import ParquetDataset
    sig = {'ECG': torch.randn(1024 * 4), 'ABD': torch.full((256 * 4,), float('-inf')), 'THX': torch.randn(256 * 4) * 3 + 2}
    out = ParquetDataset._causal_normalize(sig)
    assert torch.equal(out['ABD'], sig['ABD'])
    want, _ = O.causal_rolling_normalize(sig['THX'].numpy(), 256 / 30.0, tau_seconds=900.0, baseline_tau_seconds=120.0, min_sigma=0.1)
    np.testing.assert_allclose(out['THX'].numpy(), want, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        ParquetDataset([], ['ECG'], causal=True, normalize_on_device=True)
