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
    lib.map_labels(src, dst, src.numel(), num_classes)
    return dst


def augment_(signals: dict[str, torch.Tensor], flip_polarity: bool = True, masker=None) -> dict[str, torch.Tensor]:
    """invert_signals + SignalMasker, in place, one pass per signal (trainer/main.py:131-138)."""
    names = list(signals.keys())
    first = signals[names[0]]
    B, dev = first.shape[0], first.device
    keep_BC = None
    if masker is not None:
        _, keep_BC = masker.draw(signals)
    for j, name in enumerate(names):
        x = signals[name]
        assert x.is_contiguous() and x.dtype == torch.float32
        sign = (2 * torch.randint(0, 2, (B,), dtype=torch.float, device=dev) - 1) if flip_polarity else None
        keep = keep_BC[:, j].to(torch.uint8).contiguous() if keep_BC is not None else None
        lib.augment(x, B, x.shape[1], sign, keep)
    return signals
