"""Public inference API mirrored from src/wav2sleep/api.py:53-99,163-190: `load_model`, `predict`.

`load_model(folder)` reads the reference's deployment artefact unchanged -- `config.yaml` (the fully resolved model
config with Hydra `_target_` keys, written by log.py:64-83) + `state_dict.pth` -- without hydra/omegaconf: a
minimal `_target_` instantiator maps `wav2sleep.models.wav2sleep.*` onto the classes of this package.
HF-Hub download (`hf://...`) needs network and is out of scope.
"""

from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import yaml

from . import ppgnet as _ppg
from . import wav2sleep as _w

_TARGETS = {
    'wav2sleep.models.wav2sleep.Wav2Sleep': _w.Wav2Sleep,
    'wav2sleep.models.wav2sleep.SignalEncoders': _w.SignalEncoders,
    'wav2sleep.models.wav2sleep.MultiModalAttentionEmbedder': _w.MultiModalAttentionEmbedder,
    'wav2sleep.models.wav2sleep.SequenceCNN': _w.SequenceCNN,
    'wav2sleep.models.ppgnet.SleepPPGNet': _ppg.SleepPPGNet,   # scripts/config/model/ppgnet.yaml
}


def instantiate(cfg):
    """Recursive `_target_` instantiation (the subset of hydra.utils.instantiate that api.py:91 relies on)."""
    if isinstance(cfg, dict):
        kwargs = {k: instantiate(v) for k, v in cfg.items() if k != '_target_'}
        if '_target_' in cfg:
            t = cfg['_target_']
            if t not in _TARGETS:
                raise ValueError(f'unknown _target_ {t}')
            return _TARGETS[t](**kwargs)
        return kwargs
    if isinstance(cfg, list):
        return [instantiate(v) for v in cfg]
    return cfg


def _resolve_device(device: str) -> str:
    if device == 'auto':
        device = 'cuda' if torch.cuda.is_available() else 'cpu'
    return device


def load_model(folder: str, device: str = 'auto', compile: bool = False, revision: str | None = None, cache_dir: str | None = None):
    """api.py:53-99.  `compile=True` calls `model.compile()` as the reference does (api.py:96-97); on these modules that records the
    request and changes nothing (the forward already is hand-written gfx950 code)."""
    if str(folder).startswith('hf://'):
        raise NotImplementedError('Hugging Face Hub download needs network; pass a local folder with config.yaml + state_dict.pth')
    device = _resolve_device(device)
    config_fp = os.path.join(folder, 'config.yaml')
    if not os.path.exists(config_fp):
        raise FileNotFoundError(f'No config file found at {config_fp}. Has the model been downloaded?')
    with open(config_fp, 'r') as f:
        model_cfg = yaml.safe_load(f)
    model = instantiate(model_cfg)
    ckpt_path = os.path.join(folder, 'state_dict.pth')
    if not os.path.exists(ckpt_path):
        raise FileNotFoundError(f'No state dict found at {ckpt_path}. Has the model been downloaded?')
    sd = torch.load(ckpt_path, weights_only=True, map_location='cpu')
    if all(k.startswith('model.') for k in sd):  # Lightning checkpoint prefix (log.py:57-58)
        sd = {k[len('model.'):]: v for k, v in sd.items()}
    model.load_state_dict(sd)
    model.eval()
    if compile:
        model.compile()
    return model.to(device)


@torch.inference_mode()
def predict(model, dataset, device: str = 'auto', batch_size: int = 4, num_workers: int = 0) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """api.py:163-190: dataset yields (dict signal -> [T], labels [S]); returns (preds [N, S] cpu, labels | None)."""
    device = _resolve_device(device)
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, pin_memory=device.startswith('cuda'),
                                         shuffle=False)
    preds, labels = [], []
    for x, yb in loader:
        x = {k: v.to(device) for k, v in x.items()}
        if getattr(dataset, 'normalize_on_device', False):  # raw samples: per-recording z-score as one kernel per signal
            from .inputs import zscore_normalize
            x = zscore_normalize(x)
        preds.append(model(x).argmax(dim=-1))
        labels.append(yb)
    preds = torch.cat(preds, dim=0).cpu()
    labels = torch.cat(labels, dim=0).cpu()
    if (labels == -1).all():
        labels = None
    return preds, labels
