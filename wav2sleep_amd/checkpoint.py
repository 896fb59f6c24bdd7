"""Checkpoint / EMA interop with the reference training stack.

Mirrors, on the flat device buffers of the HIP engine:
  * `EMACallback` (reference trainer/callbacks.py:12-128): same constructor, hook names, update rule
    `ema = decay*ema + (1-decay)*param`, swap-in for validation/test, adoption at train end, `state_dict` schema
    (`{'ema_state_dict': {'model.<name>': tensor}, 'step_count': int}`) -- ONE kernel per update / swap for all tensors.
  * Lightning `.ckpt` files as the reference writes and reads them (trainer/main.py:299-334, log.py:50-60): a dict with
    `state_dict` (keys prefixed `model.`), `optimizer_states` (torch.optim.AdamW layout), `lr_schedulers`, `global_step`,
    `epoch`, `callbacks`, `gradient_clip_val`, `gradient_clip_algorithm`, `rng_state`, `cuda_rng_state_all`.
  * the exported model folder `config.yaml` + `state_dict.pth` (log.py:75-83) that `load_model` reads.
"""
from __future__ import annotations

import logging
import os
from typing import Any

import torch

from . import lib

logger = logging.getLogger(__name__)

__all__ = ['EMACallback', 'lightning_checkpoint', 'save_lightning_checkpoint', 'load_lightning_checkpoint', 'save_model']


def _module_and_model(pl_module):
    """Accept a SleepModule (has .model and .trainer) or a bare Wav2Sleep."""
    model = getattr(pl_module, 'model', pl_module)
    return pl_module, model


class EMACallback:
    """Exponential moving average of the model weights (reference trainer/callbacks.py:12-128).

    decay in [0, 1] (ValueError otherwise, callbacks.py:33-34); updates start once `start_step` batches were seen.
    The average lives in one flat fp32 device buffer aligned with the model's flat parameter buffer; `device='cpu'`
    (a memory-saving option of the reference) is accepted and keeps a host copy that is refreshed lazily for
    `state_dict()` only -- the arithmetic always runs on the GPU.
    """

    def __init__(self, decay: float = 0.9999, start_step: int = 0, device: str | None = None):
        if not 0.0 <= decay <= 1.0:
            raise ValueError(f'decay must be in [0, 1], got {decay}')
        self.decay = decay
        self.start_step = start_step
        self.device = device
        self._ema_flat: torch.Tensor | None = None
        self._model = None
        self._swapped = False
        self._step_count = 0

    # -- Lightning hook names -----------------------------------------------------------------------
    def setup(self, trainer=None, pl_module=None, stage: str = 'fit') -> None:
        _, model = _module_and_model(pl_module)
        if stage == 'fit' and self._ema_flat is None:
            model._ensure_flat()
            self._model = model
            self._ema_flat = model._flat.clone()

    def _should_update(self) -> bool:
        return self._step_count >= self.start_step

    def on_train_batch_end(self, trainer=None, pl_module=None, outputs: Any = None, batch: Any = None, batch_idx: int = 0) -> None:
        self._step_count += 1
        if self._ema_flat is None or not self._should_update():
            return
        _, model = _module_and_model(pl_module)
        model._ensure_flat()
        lib.ema_update(self._ema_flat, model._flat, self._ema_flat.numel(), self.decay)

    def _swap(self, pl_module) -> None:
        _, model = _module_and_model(pl_module)
        model._ensure_flat()
        lib.swap(model._flat, self._ema_flat, self._ema_flat.numel())
        model.mark_params_dirty()

    def _swap_to_ema(self, pl_module) -> None:
        if self._ema_flat is None or self._swapped:
            return
        self._swap(pl_module)
        self._swapped = True

    def _swap_to_original(self, pl_module) -> None:
        if self._ema_flat is None or not self._swapped:
            return
        self._swap(pl_module)
        self._swapped = False

    def on_validation_epoch_start(self, trainer=None, pl_module=None) -> None:
        self._swap_to_ema(pl_module)

    def on_validation_epoch_end(self, trainer=None, pl_module=None) -> None:
        self._swap_to_original(pl_module)

    def on_test_epoch_start(self, trainer=None, pl_module=None) -> None:
        self._swap_to_ema(pl_module)

    def on_test_epoch_end(self, trainer=None, pl_module=None) -> None:
        self._swap_to_original(pl_module)

    def on_train_end(self, trainer=None, pl_module=None) -> None:
        if self._ema_flat is None:
            return
        _, model = _module_and_model(pl_module)
        model._ensure_flat()
        model._flat.copy_(self._ema_flat)
        model.mark_params_dirty()

    # -- checkpoint state (same schema as the reference) ---------------------------------------------
    def state_dict(self) -> dict[str, Any]:
        sd = None
        if self._ema_flat is not None:
            dev = self.device if self.device else self._ema_flat.device
            sd = {}
            for (o, n, shape), (name, _) in zip(self._model._layout, self._model.named_parameters()):
                sd['model.' + name] = self._ema_flat[o:o + n].view(shape).clone().to(dev)
        return {'ema_state_dict': sd, 'step_count': self._step_count}

    def load_state_dict(self, state_dict: dict[str, Any], pl_module=None) -> None:
        self._step_count = state_dict.get('step_count', 0)
        sd = state_dict.get('ema_state_dict')
        if sd is None:
            self._ema_flat = None
            return
        if pl_module is not None:
            _, self._model = _module_and_model(pl_module)
        if self._model is None:
            raise RuntimeError('EMACallback.load_state_dict needs the module (call setup() first or pass pl_module=...)')
        self._model._ensure_flat()
        flat = torch.zeros_like(self._model._flat)
        for (o, n, shape), (name, _) in zip(self._model._layout, self._model.named_parameters()):
            key = 'model.' + name if ('model.' + name) in sd else name
            flat[o:o + n].view(shape).copy_(sd[key].to(flat.device, torch.float32))
        self._ema_flat = flat


# ------------------------------------------------------------------------------------------------------------------
# Lightning checkpoint dictionaries
# ------------------------------------------------------------------------------------------------------------------
def _optimizer_state(step: 'FusedTrainStep') -> dict:
    """torch.optim.AdamW.state_dict() layout: per-parameter {step, exp_avg, exp_avg_sq} in parameter order."""
    model = step.model
    state = {}
    k = torch.tensor(float(step.step_count))
    for i, ((o, n, shape), _) in enumerate(zip(model._layout, model.named_parameters())):
        if step.step_count > 0:
            state[i] = {'step': k.clone(), 'exp_avg': step.m[o:o + n].view(shape).clone(), 'exp_avg_sq': step.v[o:o + n].view(shape).clone()}
    # after k optimiser steps the scheduler has stepped k times too: the group's lr is the one step k + 1 will use
    group = dict(lr=step.lr_at(step.step_count + 1), betas=tuple(step.betas), eps=step.eps, weight_decay=step.wd, amsgrad=False,
                 maximize=False, foreach=None, capturable=False, differentiable=False, fused=None, initial_lr=step.lr_max,
                 params=list(range(len(model._layout))))
    return {'state': state, 'param_groups': [group]}


def _scheduler_state(step: 'FusedTrainStep') -> dict:
    """`ExpWarmUpScheduler.state_dict()` (trainer/scheduler.py:7-32: every attribute but the optimiser) after `step_count` optimiser steps,
    produced by the scheduler class itself on a stand-in optimiser so that the key set is torch's own."""
    from .trainer import ExpWarmUpScheduler
    import warnings
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=step.lr_max)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sched = ExpWarmUpScheduler(opt, lr_max=step.lr_max, warmup_steps=step.warmup_steps, tau=step.tau)
    k = step.step_count
    sd = sched.state_dict()
    sd.update(last_epoch=k, _step_count=k + 1, _last_lr=[step.lr_at(k + 1)])
    return sd


def lightning_checkpoint(module, epoch: int = 0, callbacks: dict | None = None) -> dict:
    """The dictionary Lightning would write for `SleepLightningModule` after `module.trainer.step_count` steps."""
    step = module.trainer
    model = module.model
    model._ensure_flat()
    ckpt = {
        'epoch': epoch,
        'global_step': step.step_count,
        'pytorch-lightning_version': 'wav2sleep_amd',
        'state_dict': {'model.' + k: v.detach().clone() for k, v in model.state_dict().items()},
        'optimizer_states': [_optimizer_state(step)],
        'lr_schedulers': [_scheduler_state(step)],
        'callbacks': dict(callbacks or {}),
        # SleepLightningModule.on_save_checkpoint (trainer/main.py:299-308)
        'gradient_clip_val': step.max_norm,
        'gradient_clip_algorithm': 'norm',
        'rng_state': torch.get_rng_state(),
        'w2s_seed_state': (model._seed_base, model._seed_ctr),
    }
    if torch.cuda.is_available():
        ckpt['cuda_rng_state_all'] = torch.cuda.get_rng_state_all()
    return ckpt


def save_lightning_checkpoint(path: str, module, epoch: int = 0, callbacks: dict | None = None) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(lightning_checkpoint(module, epoch=epoch, callbacks=callbacks), path)
    return path


def load_lightning_checkpoint(checkpoint, module, strict: bool = True, restore_rng: bool = True) -> dict:
    """Resume `module` (SleepModule) from a Lightning `.ckpt` path or dict written by the reference or by
    `save_lightning_checkpoint`: weights, AdamW moments, step / scheduler position, RNG, and the gradient-clipping
    consistency warning of SleepLightningModule.on_load_checkpoint (trainer/main.py:310-334).

    With a process group (world > 1) this is a COLLECTIVE call, as Lightning's resume is: EVERY rank calls it with the same checkpoint;
    it ends with `FusedTrainStep.sync_parameters()`, which broadcasts rank 0's weights, moments and scalar optimiser state, so the ranks
    leave identical whatever each of them read.  Do not call `sync_parameters()` again on a subset of ranks afterwards."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location='cpu', weights_only=False)
    step = module.trainer
    model = module.model
    sd = checkpoint['state_dict']
    sd = {(k[len('model.'):] if k.startswith('model.') else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=strict)
    model._ensure_flat()
    model.mark_params_dirty()

    ckpt_clip = checkpoint.get('gradient_clip_val', None)
    if ckpt_clip != step.max_norm:
        logger.warning('\n%s\nGRADIENT CLIPPING MISMATCH DETECTED!\n  Checkpoint trained with: gradient_clip_val=%s\n'
                       '  Current config has:      gradient_clip_val=%s\n%s\n', '=' * 70, ckpt_clip, step.max_norm, '=' * 70)

    opt = (checkpoint.get('optimizer_states') or [None])[0]
    step.m.zero_()
    step.v.zero_()
    k = int(checkpoint.get('global_step', 0))
    if opt is not None:
        for i, (o, n, shape) in enumerate(model._layout):
            st = opt['state'].get(i)
            if st is None:
                continue
            step.m[o:o + n].view(shape).copy_(st['exp_avg'].to(step.m.device, torch.float32))
            step.v[o:o + n].view(shape).copy_(st['exp_avg_sq'].to(step.v.device, torch.float32))
            k = int(st['step']) if 'step' in st else k
        g = opt['param_groups'][0]
        step.betas, step.eps, step.wd = tuple(g['betas']), g['eps'], g['weight_decay']
        step.lr_max = g.get('initial_lr', step.lr_max)
    step.step_count = k
    step.micro = 0   # a load in the middle of an accumulation window starts a fresh window (the flat gradient holds another run's partial sum)
    if 'w2s_seed_state' in checkpoint:
        model._seed_base, model._seed_ctr = checkpoint['w2s_seed_state']
    if restore_rng:
        if 'rng_state' in checkpoint:
            torch.set_rng_state(checkpoint['rng_state'])
        if 'cuda_rng_state_all' in checkpoint and torch.cuda.is_available():
            try:
                torch.cuda.set_rng_state_all(checkpoint['cuda_rng_state_all'])
            except (RuntimeError, IndexError):  # different device count than the run that saved it
                pass
    step.sync_parameters()   # collective (every rank is in this function): rank 0's weights, moments and scalar state everywhere; a no-op at world size 1
    return checkpoint


def save_model(folder: str, model, config: dict | None = None) -> str:
    """`config.yaml` + `state_dict.pth` as log.py:75-83 logs them (the layout `load_model` and the reference's
    `api.load_model` read).  `config` defaults to the `_target_` tree reconstructed from the model's own attributes."""
    import yaml
    os.makedirs(folder, exist_ok=True)
    if config is None:
        config = model.config_dict()
    with open(os.path.join(folder, 'config.yaml'), 'w') as f:
        yaml.safe_dump(config, f, sort_keys=False)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(folder, 'state_dict.pth'))
    return folder
