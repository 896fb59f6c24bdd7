"""Evaluation statistics (src/wav2sleep/stats.py:9-30), host-side on the [C, C] confusion matrix."""
import numpy as np


def confusion_accuracy(cmat) -> float:
    cmat = np.asarray(cmat)
    return float(np.trace(cmat) / np.sum(cmat))


def cohens_kappa(cmat, n_classes: int = 4) -> float:
    cmat = np.asarray(cmat).astype(float)
    sum0 = np.sum(cmat, axis=0)
    sum1 = np.sum(cmat, axis=1)
    expected = np.outer(sum0, sum1) / np.sum(sum0)
    w_mat = np.ones((n_classes, n_classes)) - np.eye(n_classes)
    k = np.sum(w_mat * cmat) / np.sum(w_mat * expected)
    return float(1 - k)
