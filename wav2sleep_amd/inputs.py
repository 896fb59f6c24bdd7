"""Device-side input pipeline: the step just before the hot path.

The reference normalises every recording on the CPU inside the DataLoader worker (`ParquetDataset._zscore_normalize`,
data/dataset.py:76-87), pads missing columns with -inf (:170-173), maps 5-class AASM labels (:174-182), and then on the
device flips polarity and masks modalities with torch indexing (trainer/main.py:131-138,342-353; masker.py:10-51).  Here
the three per-sample passes are single kernels over the raw [B, T] tensors (`csrc/input_pipe.hip`); the random draws stay
tiny torch ops on [B, C] tensors so the sampling rule is the reference's own.
"""

from __future__ import annotations

import torch

from . import lib
from .settings import COLS_TO_SAMPLES_PER_EPOCH


def zscore_normalize(signals: dict[str, torch.Tensor], eps: float = 1e-6) -> dict[str, torch.Tensor]:
    """Per-recording z-score of every [B, T] signal (each row is one recording); rows holding non-finite values
    (the -inf "missing modality" rows) pass through unchanged, like dataset.py:80-82."""
    out = {}
    for k, x in signals.items():
        x = x.contiguous().float()
        B, T = x.shape
        nblk = max(1, min(256, T // 4096))
        part = torch.empty(B * nblk * 3, device=x.device, dtype=torch.float64)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            lib.zscore(x, y, B, T, part, nblk, eps)
        out[k] = y
    return out


def pad_missing(signals: dict[str, torch.Tensor], columns: list[str], epochs: int, batch: int, device) -> dict[str, torch.Tensor]:
    """Missing columns become full-length -inf rows (dataset.py:170-173)."""
    out = dict(signals)
    for c in columns:
        if c not in out:
            out[c] = torch.full((batch, epochs * COLS_TO_SAMPLES_PER_EPOCH[c]), float('-inf'), device=device)
    return {c: out[c] for c in columns}


def map_labels(stages: torch.Tensor, num_classes: int) -> torch.Tensor:
    """AASM stages {0..4, NaN} -> float labels in {0..num_classes-1, -1} (settings.py:52-56, dataset.py:174-182)."""
    src = stages.contiguous().float()
    dst = torch.empty_like(src)
    with torch.cuda.device(src.device):
        lib.map_labels(src, dst, src.numel(), num_classes)
    return dst


def augment_(signals: dict[str, torch.Tensor], flip_polarity: bool = True, masker=None) -> dict[str, torch.Tensor]:
    """invert_signals + SignalMasker, in place, one pass per signal (trainer/main.py:131-138).  The random draws are made in the
    reference's order -- the polarity of every signal first (`invert_signals`, one `randint` per signal in dict order, main.py:342-353),
    then the masker's Bernoulli / categorical draws (masker.py:10-46) -- so a run seeded like the reference consumes the device RNG
    stream identically; only the application is fused (w2s_augment: sign and -inf rows in one pass)."""
    names = list(signals.keys())
    first = signals[names[0]]
    B, dev = first.shape[0], first.device
    signs = [(2 * torch.randint(0, 2, (B, 1), dtype=torch.float, device=dev) - 1).reshape(B) if flip_polarity else None for _ in names]
    keep_BC = None
    if masker is not None:
        _, keep_BC = masker.draw(signals)   # availability is read from the -inf rows, which a sign flip leaves -inf or +inf alike
    for j, name in enumerate(names):
        x = signals[name]
        assert x.is_contiguous() and x.dtype == torch.float32
        keep = keep_BC[:, j].to(torch.uint8).contiguous() if keep_BC is not None else None
        with torch.cuda.device(dev):
            lib.augment(x, B, x.shape[1], signs[j], keep)
    return signals


def causal_rolling_normalize(signal, sampling_freq: float, tau_seconds: float = 900.0, eps: float = 1e-6, outlier_threshold_sigma: float = 4.0,
                             return_outlier_mask: bool = False, baseline_tau_seconds: float | None = None, min_sigma: float = 0.1):
    """Causal EMA z-score of ONE recording (reference data/normalization.py:106-230, same arguments): running mean (time constant
    `baseline_tau_seconds`) and running variance (`tau_seconds`) with residuals clipped at `outlier_threshold_sigma` sigma, sigma floored
    at `min_sigma`.  numpy array or torch tensor in, the same type / dtype / device out.  The scan is sequential per recording, so it
    runs on the host inside the dataset workers, as the reference's numba loop does: `w2s_causal_normalize_host` in the native library."""
    import ctypes as C

    import numpy as np
    is_torch = isinstance(signal, torch.Tensor)
    arr = signal.detach().cpu().numpy() if is_torch else np.asarray(signal)
    n = int(arr.shape[0]) if arr.ndim else 0
    if arr.ndim != 1:
        raise ValueError(f'expected a 1-D signal, got shape {arr.shape}')
    if n == 0:
        mask = torch.zeros(0, dtype=torch.bool, device=signal.device) if is_torch else np.zeros(0, dtype=bool)
        return (signal, mask) if return_outlier_mask else signal
    x64 = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.empty(n, dtype=np.float64)
    flags = np.empty(n, dtype=np.uint8)
    fn = lib.load().w2s_causal_normalize_host
    fn.argtypes = [C.c_void_p, C.c_long] + [C.c_double] * 6 + [C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    rc = fn(x64.ctypes.data, n, float(sampling_freq), float(tau_seconds), float(eps), float(outlier_threshold_sigma),
            float(baseline_tau_seconds) if baseline_tau_seconds is not None else -1.0, float(min_sigma), out.ctypes.data, flags.ctypes.data)
    if rc != 0:
        raise lib.W2SError(f'w2s_causal_normalize_host failed with code {rc}')
    if is_torch:
        res = torch.from_numpy(out).to(device=signal.device, dtype=signal.dtype)
        return (res, torch.from_numpy(flags.astype(bool)).to(signal.device)) if return_outlier_mask else res
    res = out.astype(arr.dtype) if np.issubdtype(arr.dtype, np.floating) else out
    return (res, flags.astype(bool)) if return_outlier_mask else res
