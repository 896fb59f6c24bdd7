"""Agreement statistics on a [C, C] confusion matrix (rows = true stage, columns = predicted stage), host side.

Same quantities as the reference's `stats.py:9-30` (accuracy; Cohen's kappa with the unweighted disagreement matrix 1 - I), written
in closed form: with N = total count, p_o = trace / N the observed agreement and p_e = sum_i rowsum_i * colsum_i / N^2 the chance
agreement, kappa = 1 - (1 - p_o) / (1 - p_e) = (p_o - p_e) / (1 - p_e).
"""
import numpy as np


def _counts(cmat, n_classes=None) -> np.ndarray:
    m = np.asarray(cmat, dtype=np.float64)
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise ValueError(f'confusion matrix must be square, got {m.shape}')
    if n_classes is not None and m.shape[0] != n_classes:
        raise ValueError(f'confusion matrix is {m.shape[0]} x {m.shape[0]} but n_classes={n_classes}')
    return m


def confusion_accuracy(cmat) -> float:
    m = _counts(cmat)
    return float(np.trace(m) / m.sum())


def cohens_kappa(cmat, n_classes: int = 4) -> float:
    m = _counts(cmat, n_classes)
    total = m.sum()
    observed = np.trace(m) / total
    chance = float(m.sum(axis=1) @ m.sum(axis=0)) / (total * total)
    return float((observed - chance) / (1.0 - chance))
