"""Parquet in -> `.preds.csv` out: the callers either side of the inference forward.

Mirrors `ParquetDataset` (reference data/dataset.py:23-183; global z-score or, `causal=True`, the online EMA normalisation), `load_dataset`,
`save_predictions` and `predict_on_folder` (api.py:141-160,193-301).  File discovery, truncation, the length
consistency checks, label mapping, `-inf` padding of absent columns, output tree and CSV schema (index `Timestamp`
= 30 s * k + 30, columns `Pred` [, `Stage`]) follow the reference line by line in behaviour.

MI355X-first difference: with `normalize_on_device=True` (what `load_dataset` builds) the dataset hands over RAW
samples and the per-recording z-score runs as one kernel per signal over the whole [B, T] batch after the transfer
(`inputs.zscore_normalize` -> `w2s_zscore`), instead of per file in a DataLoader worker; `-inf` rows pass through
exactly like dataset.py:80-82.  `normalize_on_device=False` reproduces the reference's host-side order of operations.

EDF / CSV preprocessing (`prepare`, pyedflib + resampling) is ingestion, out of scope: `preprocess=True` accepts only
folders that already hold model-ready parquet files and raises otherwise.
"""
from __future__ import annotations

import logging
import os
from glob import glob
from pathlib import Path
from typing import Iterable, Optional, Tuple

import numpy as np
import torch

from .settings import COLS_TO_SAMPLES_PER_EPOCH, INTEGER_LABEL_MAPS, LABEL, PRED, TIMESTAMP

logger = logging.getLogger(__name__)

__all__ = ['ParquetDataset', 'load_dataset', 'save_predictions', 'predict_on_folder', 'try_read_parquet']


def try_read_parquet(fp: str, columns: list[str] | None = None, max_retries: int = 3):
    """`pd.read_parquet` that survives a flaky filesystem: up to `max_retries` further attempts, every failure logged, then a
    ValueError naming the file -- the contract of the reference's helper of the same name (data/dataset.py:188-198), as a loop."""
    import pandas as pd
    for _ in range(max_retries + 1):
        try:
            return pd.read_parquet(fp, columns=columns)
        except Exception as e:  # noqa: BLE001 (anything the reader raises counts as a failed attempt, as in the reference)
            logger.error(f'Failed to read parquet {fp=} - {e}')
    raise ValueError(f'Failed to read parquet {fp=}')


class ParquetDataset(torch.utils.data.Dataset):
    """data/dataset.py:23-186 (`causal=True`: online EMA normalisation per recording instead of the global z-score, :89-130,165-167)."""

    def __init__(self, parquet_fps: list[str], columns: list[str], num_classes: int = 4, require_labels: bool = True,
                 max_length_hours: int | None = None, causal: bool = False, normalize_on_device: bool = False):
        if causal and normalize_on_device:
            raise ValueError('causal normalisation is a sequential scan per recording and runs in the dataset workers: normalize_on_device=False')
        self.files = parquet_fps
        self.columns = columns
        for col in self.columns:
            if col not in COLS_TO_SAMPLES_PER_EPOCH:
                raise ValueError(f'Column {col} unrecognised.')
        self.map = INTEGER_LABEL_MAPS[num_classes]
        self.require_labels = require_labels
        self.max_length_epochs = 1_000_000 if max_length_hours is None else max_length_hours * 60 * 2
        self.causal = causal
        self.normalize_on_device = normalize_on_device

    @staticmethod
    def _zscore_normalize(signals: dict) -> dict:
        """Host z-score, dataset.py:76-87 (used when normalize_on_device=False)."""
        out = {}
        eps = 1e-6
        for k, x in signals.items():
            if x.numel() == 0 or not torch.isfinite(x).all():
                out[k] = x
                continue
            mu = torch.mean(x)
            std = torch.std(x)
            std = std if std > eps else torch.tensor(eps, dtype=x.dtype)
            out[k] = (x - mu) / std
        return out

    @staticmethod
    def _causal_normalize(signals: dict) -> dict:
        """Online EMA z-score, dataset.py:89-130: per signal with its own sampling rate; missing / non-finite signals pass through."""
        from .inputs import causal_rolling_normalize
        from .settings import (CAUSAL_NORM_BASELINE_TAU_SECONDS, CAUSAL_NORM_MIN_SIGMA, CAUSAL_NORM_TAU_SECONDS, NORM_OUTLIER_THRESHOLD)
        out = {}
        for k, x in signals.items():
            if x.numel() == 0 or not torch.isfinite(x).all() or k not in COLS_TO_SAMPLES_PER_EPOCH:
                out[k] = x
                continue
            out[k] = causal_rolling_normalize(x, sampling_freq=COLS_TO_SAMPLES_PER_EPOCH[k] / 30.0, tau_seconds=CAUSAL_NORM_TAU_SECONDS,
                                              outlier_threshold_sigma=NORM_OUTLIER_THRESHOLD, baseline_tau_seconds=CAUSAL_NORM_BASELINE_TAU_SECONDS,
                                              min_sigma=CAUSAL_NORM_MIN_SIGMA)
        return out

    def _signals(self, df, fp):
        """Present columns truncated to whole epochs (<= max length); the recording length every column must agree on."""
        present, n_epochs = {}, None
        for col in self.columns:
            if col not in df.columns:
                continue
            spe = COLS_TO_SAMPLES_PER_EPOCH[col]
            x_T = torch.from_numpy(df[col].dropna().values).float()
            if torch.isinf(x_T).any():
                raise ValueError(f'{fp=} has inf. values for {col=}')
            epochs = x_T.shape[0] // spe
            if n_epochs is not None and n_epochs != epochs:
                raise ValueError(f'prev_inferred_recording_length_epochs={n_epochs} != inferred_recording_length_epochs={epochs} for {fp=}')
            n_epochs = epochs
            present[col] = x_T[: spe * min(epochs, self.max_length_epochs)]
        if n_epochs is None:
            raise ValueError(f'No relevant columns found in {fp=}. {self.columns=}')
        return present, n_epochs

    def _labels(self, df, n_epochs, fp):
        if not (self.require_labels or LABEL in df.columns):
            return torch.full((n_epochs,), -1).float()[: self.max_length_epochs]
        stages = df[LABEL].dropna().map(self.map)
        y = torch.from_numpy(stages.fillna(-1).values.T).float()
        if y.shape[0] != n_epochs:
            raise ValueError(f'labels.shape={tuple(y.shape)} != inferred_recording_length_epochs={n_epochs} for {fp=}')
        return y[: self.max_length_epochs]

    def __getitem__(self, idx):
        fp = self.files[idx]
        df = try_read_parquet(fp)
        present, n_epochs = self._signals(df, fp)
        if self.causal:
            present = self._causal_normalize(present)
        elif not self.normalize_on_device:
            present = self._zscore_normalize(present)
        kept = min(n_epochs, self.max_length_epochs)
        out = {}
        for col in self.columns:  # absent columns: full-length -inf vectors (dataset.py:170-173); dict order = columns present first
            if col in present:
                out[col] = present[col]
        for col in self.columns:
            if col not in present:
                out[col] = torch.full((COLS_TO_SAMPLES_PER_EPOCH[col] * kept,), float('-inf')).float()
        return out, self._labels(df, n_epochs, fp)

    def __len__(self) -> int:
        return len(self.files)


def _get_parquet_files(folder: str) -> list[str]:
    return glob(os.path.join(folder, '**/*.parquet'), recursive=True)


def load_dataset(parquet_folder: str, signals: Iterable[str], num_classes: int = 4, max_length_hours: Optional[int] = None,
                 normalize_on_device: bool = True) -> ParquetDataset:
    """api.py:141-160."""
    signals = list(signals)
    input_fps = _get_parquet_files(parquet_folder)
    if len(input_fps) == 0:
        raise ValueError(f'No parquet files found in {parquet_folder}.')
    return ParquetDataset(parquet_fps=input_fps, num_classes=num_classes, columns=signals, require_labels=False,
                          max_length_hours=max_length_hours, normalize_on_device=normalize_on_device)


def _prediction_frame(pred_S, labels_S, first_index, datetime_index: bool):
    """One recording's output table: row k is the epoch ending 30*(k+1) s after the start (api.py:210-217)."""
    import pandas as pd
    n = int(len(pred_S))
    pred_S = pred_S.numpy() if isinstance(pred_S, torch.Tensor) else np.asarray(pred_S)
    if isinstance(labels_S, torch.Tensor):
        labels_S = labels_S.numpy()
    stamps = pd.Index(30.0 * (np.arange(n, dtype=np.float64) + 1.0), name=TIMESTAMP)
    if datetime_index:
        stamps = first_index + pd.to_timedelta(stamps, unit='s')
    table = pd.DataFrame({PRED: pred_S[:n]}, index=stamps)
    if labels_S is not None:
        table[LABEL] = labels_S[:n]
    return table


def save_predictions(predictions: torch.Tensor, parquet_folder: str, output_folder: str, dataset: ParquetDataset,
                     labels: Optional[torch.Tensor] = None, overwrite: bool = False, max_length_hours: Optional[int] = None) -> None:
    """api.py:193-222: `<output_folder>/<path below parquet_folder>/<name>.preds.csv` per recording; existing files are
    skipped with a warning unless `overwrite`; a DatetimeIndex input yields absolute timestamps."""
    import pandas as pd
    for i, fp in enumerate(dataset.files):
        target = Path(output_folder) / Path(fp).relative_to(parquet_folder).with_suffix('.preds.csv')
        if target.exists() and not overwrite:
            logger.warning(f'File {target} exists. Skipping.')
            continue
        source = pd.read_parquet(fp)
        source = source[list(set(dataset.columns) & set(source.columns))]
        table = _prediction_frame(predictions[i], None if labels is None else labels[i], source.index[0],
                                  isinstance(source.index, pd.DatetimeIndex))
        target.parent.mkdir(parents=True, exist_ok=True)
        table.to_csv(str(target))


def predict_on_folder(input_folder: str, output_folder: str, *, model=None, model_folder: Optional[str] = None,
                      signals: Optional[Iterable[str]] = None, device: str = 'auto', batch_size: int = 4, num_workers: int = 4,
                      preprocess: bool = True, max_length_hours: int = 10, overwrite: bool = False, compile: bool = False,
                      return_tensors: bool = False) -> None | Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """api.py:225-301."""
    from .api import _resolve_device, load_model, predict
    device = _resolve_device(device)
    if model is None:
        if model_folder is None:
            raise ValueError('Either `model` or `model_folder` must be provided.')
        model = load_model(model_folder, device=device, compile=compile)
    else:
        model = model.to(device)
        model.eval()
    if signals is None:
        if not hasattr(model, 'valid_signals'):
            raise AttributeError('Model does not expose `valid_signals`. Please pass `signals` explicitly.')
        signals = list(model.valid_signals)
    else:
        signals = list(signals)
        if hasattr(model, 'valid_signals'):
            valid = set(model.valid_signals)
            if not set(signals).issubset(valid):
                raise ValueError(f'Invalid signal subset: {signals}. Valid signals are: {sorted(valid)}')
    if preprocess:
        raw = [f for ext in ('edf', 'csv') for f in glob(os.path.join(input_folder, f'**/*.{ext}'), recursive=True)]
        if raw:
            raise NotImplementedError('EDF/CSV preprocessing (api.prepare: pyedflib + resampling) is ingestion, outside the MI355X hot '
                                      'path; run the reference `prepare` once and pass the parquet folder with preprocess=False')
    parquet_folder = input_folder
    ds = load_dataset(parquet_folder=parquet_folder, signals=signals, num_classes=model.num_classes, max_length_hours=max_length_hours)
    preds, labels = predict(model=model, dataset=ds, device=device, batch_size=batch_size, num_workers=num_workers)
    save_predictions(predictions=preds, parquet_folder=parquet_folder, output_folder=output_folder, dataset=ds, labels=labels,
                     overwrite=overwrite, max_length_hours=max_length_hours)
    return (preds, labels) if return_tensors else None
