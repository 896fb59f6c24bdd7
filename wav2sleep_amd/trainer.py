"""Callers of the hot path, mirrored from the reference's trainer package (src/wav2sleep/trainer/*.py).

* `exp_warmup_lr` / `ExpWarmUpScheduler`  -- trainer/scheduler.py:7-32
* `invert_signals`, `SignalMasker`         -- trainer/main.py:342-353, trainer/masker.py:6-51 (device-side augmentations)
* `FusedTrainStep`                         -- what Lightning's loop does around `SleepLightningModule._step`
  (trainer/main.py:140-183 + clip 1.0 + AdamW + scheduler, scripts/config/training/main.yaml) as ONE launch sequence:
  forward -> masked CE + confusion matrix -> backward -> [bucketed RCCL all-reduce] -> global-norm clip + AdamW,
  all on flat buffers, no host synchronisation inside the step.
* `SleepModule` (alias `SleepLightningModule`) -- same constructor / `_step` semantics without the Lightning dependency.
"""

from __future__ import annotations

import math
import os
from collections import defaultdict

import torch
from torch.distributions.one_hot_categorical import OneHotCategorical
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LRScheduler

from . import lib
from .ddp import FlatGradReducer, reduce_metrics, reduce_ranges
from .wav2sleep import Wav2Sleep


def exp_warmup_lr(step: int, lr_max: float = 1e-3, warmup_steps: int = 2000, tau: float = 10000.0) -> float:
    """LR used by optimiser step `step` (1-based): linear warm-up then exponential decay (scheduler.py:23-32)."""
    if step <= warmup_steps:
        return lr_max * (step / warmup_steps)
    return lr_max * math.exp(-(step - warmup_steps) / tau)


class ExpWarmUpScheduler(LRScheduler):
    """trainer/scheduler.py:7-32 (for callers that drive a torch optimiser themselves)."""

    def __init__(self, optimizer: Optimizer, lr_max: float, warmup_steps: int, tau: float) -> None:
        self.lr_max = lr_max
        self.warmup_steps = warmup_steps
        self.tau = tau
        self.num_param_groups = len(optimizer.param_groups)
        super().__init__(optimizer, last_epoch=-1)

    def get_lr(self):
        return [exp_warmup_lr(self.last_epoch + 1, self.lr_max, self.warmup_steps, self.tau)] * self.num_param_groups


def invert_signals(signals: dict[str, torch.Tensor]):
    """Random polarity flip per (sample, signal), in place (trainer/main.py:342-353)."""
    for name, x_BT in signals.items():
        B = x_BT.shape[0]
        flip = 2 * torch.randint(0, 2, (B, 1), dtype=torch.float, device=x_BT.device) - 1
        signals[name] *= flip
    return signals


class SignalMasker:
    """Stochastic modality dropout with backup channel; writes -inf rows (trainer/masker.py:6-51)."""

    def __init__(self, dropouts: dict[str, float], backups: list[str] | None = None):
        self.channel_dropouts = dropouts
        self.backup_channels = backups

    def draw(self, signals):
        """-> (names, keep mask [B, C] bool) following the reference's sampling rule."""
        probs, onehot, unavailable = [], [], []
        x_BT = None
        for name, x_BT in signals.items():
            z_B = torch.isinf(x_BT[:, 0])
            p = self.channel_dropouts.get(name, 0.0)
            if p < 0.0 or p > 1:
                raise ValueError(f'channel_dropout={p} is not a valid probability.')
            probs.append(p)
            if self.backup_channels is not None:
                onehot.append(~z_B if name in self.backup_channels else torch.zeros_like(z_B))
            else:
                onehot.append(~z_B * (1 - p))
            unavailable.append(z_B)
        z_BC = torch.stack(unavailable, dim=-1)
        if z_BC.all(dim=-1).any():
            raise ValueError('Found batch element with all signals unavailable.')
        B = z_BC.size(0)
        p_BC = torch.tensor(probs, dtype=torch.float32, device=x_BT.device)[None, :].repeat(B, 1)
        if (p_BC == 1).all(dim=-1).any():
            raise ValueError('Dropout probability equal to 1 for all channels.')
        p_min = torch.stack(onehot, dim=-1).to(x_BT.device).float()
        if (p_min == 0).all(dim=-1).any():
            raise ValueError('No backup channels for stochastic sampling were available')
        min_m = OneHotCategorical(p_min).sample().bool()
        m_BC = (1 - p_BC).bernoulli().bool()
        all_zero = torch.logical_or(z_BC, ~m_BC).all(dim=-1)
        m_BC[all_zero] = min_m[all_zero]
        if torch.logical_or(z_BC, ~m_BC).all(dim=-1).any():
            raise ValueError('Masking will result in no available channels for a batch element.')
        return list(signals.keys()), m_BC

    def __call__(self, signals):
        names, m_BC = self.draw(signals)
        for name, m_B in zip(names, m_BC.T):
            signals[name][~m_B] = float('-inf')
        return signals


def wave_layout(B: int, S: int, nw: int):
    """Sample ranges [(b0, b1), ...] of the `nw` waves of a batch of B recordings (sizes differ by at most one, none empty for nw <= B) and
    the number of 256-row loss partial blocks each wave owns (the waves' partials lie one after the other: w2s_ce_wave / w2s_ce_final)."""
    if not 1 <= nw <= B:
        raise ValueError(f'{nw} waves for a batch of {B}')
    bounds = [(B * i // nw, B * (i + 1) // nw) for i in range(nw)]
    return bounds, [((b1 - b0) * S + 255) // 256 for b0, b1 in bounds]


class FusedTrainStep:
    """One optimiser step of the reference's recipe on the HIP engine (fp32, deterministic reductions).

    loss = CrossEntropy(mean over labels != -1) per rank; DDP = mean of per-rank gradients;
    clip_grad_norm_(max_norm) on the reduced gradient; AdamW(lr_k, wd); lr_k from ExpWarmUp at step k.

    `accumulate=k` = Lightning's `accumulate_grad_batches=k`, how the reference reaches its effective batch 16 when one
    micro-batch does not fit (scripts/train.py:59-76): every call of `step()` is one micro-batch whose loss is scaled by 1/k,
    gradients add up in the flat buffer, and the all-reduce (Lightning's `no_sync` on the first k-1), clip, AdamW and
    scheduler step run on the k-th call only.  `step_count` counts optimiser steps, as `trainer.global_step` does.
    """

    def __init__(self, model: Wav2Sleep, lr: float = 1e-3, weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 1.0, warmup_steps: int = 2000, tau: float = 10000.0, process_group=None, scheduler: bool = True,
                 accumulate: int = 1, waves: int | None = None):
        self.model = model
        # sample waves of the pipelined step (engine.train_waves); 1 = the whole batch in one pass
        self.waves = max(1, int(os.environ.get('W2S_WAVES', '1')) if waves is None else int(waves))
        model._ensure_flat()
        self.eng = model._engine
        flat = model._flat
        self.n = flat.numel()
        dev = flat.device
        self.device = dev
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.lr_max, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.warmup_steps, self.tau, self.use_sched = warmup_steps, tau, scheduler
        if accumulate < 1:
            raise ValueError(f'accumulate must be >= 1, got {accumulate}')
        self.accumulate = int(accumulate)
        self.micro = 0       # micro-batches seen since the last optimiser step
        self.step_count = 0  # optimiser steps
        self.nparts = 256
        self.sumsq = torch.empty(self.nparts, device=dev, dtype=torch.float32)
        self.hyper = torch.zeros(8, device=dev, dtype=torch.float32)
        self.normcoef = torch.zeros(2, device=dev, dtype=torch.float32)
        self.loss_out = torch.zeros(2, device=dev, dtype=torch.float32)
        nc = model.num_classes
        self.cmat = torch.zeros(nc, nc, device=dev, dtype=torch.int64)
        self.reducer = FlatGradReducer(model._flat_grad, group=process_group)
        self._range = reduce_ranges(model._layout, [n for n, _ in model.named_parameters()])
        # Two collectives per step: the trunk's range ('_tail': set-fusion transformer + SequenceCNN + classifier, final when the
        # encoder backward STARTS: its all-reduce hides behind ~20 ms of encoder backward) and ONE for all encoders after the encoder
        # streams have joined.  The encoders run side by side and finish together, so per-encoder all-reduces had nothing left to hide
        # behind, and each collective costs ~0.2 ms of stream hand-overs even at world size 1 (measured: 5 collectives +1.2 ms/step).
        enc = [r for k, r in self._range.items() if k != '_tail']
        self._enc_range = (min(lo for lo, _ in enc), max(hi for _, hi in enc)) if enc else (0, 0)
        tlo, thi = self._range.get('_tail', (0, 0))
        if not (self._enc_range[1] <= tlo or thi <= self._enc_range[0]):
            raise RuntimeError('flat gradient layout: the encoder ranges and the trunk range interleave')
        self.sync_parameters()

    def sync_parameters(self):
        """Rank 0's weights, AdamW moments AND scalar optimiser state (step / schedule position, lr_max, betas, eps, weight decay, the
        dropout seed state) to every rank: Lightning DDP broadcasts the module at construction.  COLLECTIVE: every rank of the group
        calls it the same number of times (four broadcasts).  `load_lightning_checkpoint` ends with one call of it, so a resume is
        "every rank calls the loader" -- do not call it again on a subset of ranks.  A no-op at world size 1."""
        if self.reducer.world > 1:
            import torch.distributed as dist
            src = dist.get_global_rank(self.reducer.group, 0) if self.reducer.group is not None else 0
            for t in (self.model._flat, self.m, self.v):
                dist.broadcast(t, src=src, group=self.reducer.group)
            model = self.model
            sc = torch.tensor([self.step_count, self.micro, self.lr_max, self.betas[0], self.betas[1], self.eps, self.wd,
                               model._seed_base, model._seed_ctr], dtype=torch.float64, device=model._flat.device)
            dist.broadcast(sc, src=src, group=self.reducer.group)
            sc = sc.tolist()
            self.step_count, self.micro = int(sc[0]), int(sc[1])
            self.lr_max, self.betas, self.eps, self.wd = sc[2], (sc[3], sc[4]), sc[5], sc[6]
            model._seed_base, model._seed_ctr = int(sc[7]), int(sc[8])
            model.mark_params_dirty()

    def lr_at(self, step: int) -> float:
        return exp_warmup_lr(step, self.lr_max, self.warmup_steps, self.tau) if self.use_sched else self.lr_max

    def apply_optimizer(self) -> float:
        """Global-norm clip + AdamW on the flat gradient buffer as it stands (3 launches); advances `step_count`; returns the lr used."""
        model = self.model
        self.step_count += 1
        k = self.step_count
        b1, b2 = self.betas
        # A fresh pinned staging tensor per step: the host runs several steps ahead of the GPU, so re-using one buffer would let step
        # k+1's values overwrite step k's before its asynchronous copy has run (the pinned allocator recycles a block only after the
        # copies that read it have completed).
        h = torch.tensor([self.lr_at(k), self.wd, b1, b2, self.eps, 1 - b1 ** k, 1 - b2 ** k, self.max_norm if self.max_norm else 0.0],
                         dtype=torch.float32).pin_memory()
        with torch.cuda.device(self.device):
            self.hyper.copy_(h, non_blocking=True)
            lib.sumsq_partial(model._flat_grad, self.n, self.sumsq, self.nparts)
            lib.clip_coef(self.sumsq, self.nparts, self.hyper, self.normcoef)
            lib.adamw(model._flat, model._flat_grad, self.m, self.v, self.n, self.hyper, self.normcoef)
        model.mark_params_dirty()
        return float(h[0])

    def step(self, x: dict[str, torch.Tensor], y: torch.Tensor) -> dict:
        """x: dict signal -> [B, T] (device, fp32, -inf rows = missing modality); y: [B, S] float labels, -1 = ignore.
        With accumulate = k > 1 this is one micro-batch; the returned dict has `stepped` = whether the optimiser ran."""
        model, eng = self.model, self.eng
        model._ensure_flat()
        first = self.micro == 0
        self.micro += 1
        last = self.micro == self.accumulate
        B = next(iter(x.values())).shape[0] if len(x) else 0
        nw = min(self.waves, B)
        reduce = last and (self.reducer.world > 1 or self.reducer.force)
        with torch.cuda.device(self.device):
            if nw > 1:
                logits = self._step_waves(x, y, nw, first, self._on_ready if reduce else None)
            else:
                eng.step_seed = model._next_seed()
                logits = eng.forward(x, train=True, save=True, pack_key=model.param_version())
                B, S, nc = logits.shape
                rows = B * S
                yv = y.reshape(rows)
                if yv.dtype != torch.float32:
                    yv = yv.float()
                part = torch.empty((rows + 255) // 256, 2, device=logits.device, dtype=torch.float32)
                glogits = torch.empty(rows, nc, device=logits.device, dtype=torch.float32)
                lib.zero_(self.cmat)
                lib.ce_fwd_bwd(logits, yv.contiguous(), rows, nc, part, self.loss_out, glogits, self.cmat, self.reducer.grad_scale / self.accumulate)
                eng.backward(glogits, accumulate=not first, hook=self._on_ready if reduce else None)
            if reduce:
                self.reducer.reduce_range(*self._enc_range)   # the encoder streams have joined the current stream
                self.reducer.wait()
        out = dict(loss=self.loss_out[0], count=self.loss_out[1], cmat=self.cmat, logits=logits, stepped=last)
        if last:
            self.micro = 0
            out['lr'] = self.apply_optimizer()
            out['grad_norm'] = self.normcoef[0]
        return out

    def _step_waves(self, x, y, nw, first, hook):
        """The same micro-batch as `nw` sample waves in a software pipeline (engine.train_waves): the trunk of one wave runs beside the
        encoders of its neighbours.  The loss is the mean over the WHOLE batch's counted labels: their number is taken from the labels
        before anything else runs, every wave's gradient is scaled by it, and the loss value is reduced once over all waves' partials."""
        model, eng = self.model, self.eng
        dev = self.device
        nc = model.num_classes
        B, S = y.shape[0], y.numel() // max(1, y.shape[0])
        rows = B * S
        yv = y.reshape(rows)
        if yv.dtype != torch.float32:
            yv = yv.float()
        yv = yv.contiguous()
        bounds, nblk = wave_layout(B, S, nw)
        part = torch.empty(sum(nblk), 2, device=dev, dtype=torch.float32)
        count = torch.empty(1, device=dev, dtype=torch.float32)
        logits = torch.empty(B, S, nc, device=dev, dtype=torch.float32)
        lib.zero_(self.cmat)
        lib.ce_count(yv, rows, nc, count)
        scale = self.reducer.grad_scale / self.accumulate

        def ce(lg, b0, b1):
            w = [b[0] for b in bounds].index(b0)
            r = (b1 - b0) * S
            g = torch.empty(r, nc, device=dev, dtype=torch.float32)
            lib.ce_wave(lg, yv[b0 * S:], r, nc, part[sum(nblk[:w]):], count, g, self.cmat, scale)
            return g
        eng.train_waves(x, ce, bounds, [model._next_seed() for _ in bounds], logits, pack_key=model.param_version(), accumulate=not first,
                        hook=hook)
        lib.ce_final(part, sum(nblk), self.loss_out)
        return logits

    def _on_ready(self, stage: str):
        if stage == '_tail':
            self.reducer.reduce_range(*self._range.get('_tail', (0, 0)))

    def metrics(self):
        """(global mean loss, rank-mean loss as the reference logs it, summed confusion matrix) -- one packed all-reduce."""
        return reduce_metrics(self.loss_out, self.cmat, self.reducer.group)


class GenericTrainStep:
    """FusedTrainStep's recipe -- CE (mean over labels != -1), DDP mean of per-rank gradients, global-norm clip, AdamW, ExpWarmUp, gradient
    accumulation -- for the models that run on the generic path: `SleepPPGNet` (trainer/main.py:72,109-113 feeds it the one signal of the
    batch) and `Wav2Sleep` configurations outside the production family.  Forward and backward are generic.py's walker and tape (HIP
    kernels), the loss / clip / AdamW kernels are the fused step's own, on one flat parameter and one flat gradient buffer."""

    def __init__(self, model: nn.Module, lr: float = 1e-3, weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 1.0, warmup_steps: int = 2000, tau: float = 10000.0, process_group=None, scheduler: bool = True,
                 accumulate: int = 1, num_classes: int | None = None):
        from .ddp import flat_layout
        self.model = model
        named = list(model.named_parameters())
        dev = named[0][1].device
        if dev.type != 'cuda':
            raise lib.W2SError('wav2sleep_amd runs on MI355X only: move the model to a cuda device (there is no CPU fallback)')
        self.device = dev
        layout, off = flat_layout([p.shape for _, p in named])
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.views = {}
        with torch.no_grad():
            for (o, n, shape), (name, p) in zip(layout, named):   # parameters become views of ONE buffer: one clip reduction, one AdamW launch
                v = self.flat[o:o + n].view(shape)
                v.copy_(p.detach().float())
                p.data = v
                self.views[p] = self.flat_grad[o:o + n].view(shape)
        self.n = off
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.lr_max, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.warmup_steps, self.tau, self.use_sched = warmup_steps, tau, scheduler
        if accumulate < 1:
            raise ValueError(f'accumulate must be >= 1, got {accumulate}')
        self.accumulate, self.micro, self.step_count = int(accumulate), 0, 0
        self.nparts = 256
        self.sumsq = torch.empty(self.nparts, device=dev, dtype=torch.float32)
        self.hyper = torch.zeros(8, device=dev, dtype=torch.float32)
        self.normcoef = torch.zeros(2, device=dev, dtype=torch.float32)
        self.loss_out = torch.zeros(2, device=dev, dtype=torch.float32)
        nc = num_classes if num_classes is not None else model.classifier.out_features
        self.num_classes = nc
        self.cmat = torch.zeros(nc, nc, device=dev, dtype=torch.int64)
        self.reducer = FlatGradReducer(self.flat_grad, group=process_group)
        # what checkpoint.py / EMACallback read on a model (the fused models carry these themselves): the storage is settled HERE
        model._flat, model._flat_grad, model._layout = self.flat, self.flat_grad, layout
        model._ensure_flat = lambda: None
        if not hasattr(model, 'mark_params_dirty'):
            model.mark_params_dirty = lambda: None
        if not hasattr(model, '_seed_base'):
            model._seed_base, model._seed_ctr = 0, 0
        self.sync_parameters()

    lr_at = FusedTrainStep.lr_at

    def sync_parameters(self):
        """Rank 0's weights, moments and scalar optimiser state to every rank (collective; FusedTrainStep.sync_parameters's contract)."""
        if self.reducer.world > 1:
            import torch.distributed as dist
            src = dist.get_global_rank(self.reducer.group, 0) if self.reducer.group is not None else 0
            for t in (self.flat, self.m, self.v):
                dist.broadcast(t, src=src, group=self.reducer.group)
            sc = torch.tensor([self.step_count, self.micro, self.lr_max, self.betas[0], self.betas[1], self.eps, self.wd, self.model._seed_ctr],
                              dtype=torch.float64, device=self.device)
            dist.broadcast(sc, src=src, group=self.reducer.group)
            sc = sc.tolist()
            self.step_count, self.micro = int(sc[0]), int(sc[1])
            self.lr_max, self.betas, self.eps, self.wd = sc[2], (sc[3], sc[4]), sc[5], sc[6]
            self.model._seed_ctr = int(sc[7])

    def _run(self, gf, x):
        if isinstance(self.model, Wav2Sleep):
            return gf.wav2sleep(self.model, x)
        if isinstance(x, dict):   # trainer/main.py:109-113
            if len(x) != 1:
                raise ValueError(f'{x.keys()=} but expected unimodal input!')
            x = x[list(x.keys())[0]]
        return gf.ppgnet(self.model, x)

    def step(self, x, y: torch.Tensor) -> dict:
        from .generic import GenericForward
        first = self.micro == 0
        self.micro += 1
        last = self.micro == self.accumulate
        self.model._seed_ctr += 1
        with torch.cuda.device(self.device), torch.no_grad():
            gf = GenericForward(training=True, seed=self.model._seed_base * 1000003 + self.model._seed_ctr, grad=True)
            logits = self._run(gf, x)
            B, S, nc = logits.shape
            rows = B * S
            yv = y.reshape(rows).float().contiguous()
            part = torch.empty((rows + 255) // 256, 2, device=self.device, dtype=torch.float32)
            glogits = torch.empty(rows, nc, device=self.device, dtype=torch.float32)
            lib.zero_(self.cmat)
            lib.ce_fwd_bwd(logits, yv, rows, nc, part, self.loss_out, glogits, self.cmat, self.reducer.grad_scale / self.accumulate)
            pg = gf.backward(logits, glogits.view(B, S, nc))
            # the tape's gradients into the flat buffer (plumbing: one multi-tensor copy / add instead of a launch per parameter)
            dst, src = [self.views[p] for p in pg], list(pg.values())
            if first:
                if len(pg) < len(self.views):
                    lib.zero_(self.flat_grad)
                torch._foreach_copy_(dst, src)
            else:
                torch._foreach_add_(dst, src)
            if last and (self.reducer.world > 1 or self.reducer.force):
                self.reducer.reduce_range(0, self.n)
                self.reducer.wait()
        out = dict(loss=self.loss_out[0], count=self.loss_out[1], cmat=self.cmat, logits=logits, stepped=last)
        if last:
            self.micro = 0
            self.step_count += 1
            k = self.step_count
            b1, b2 = self.betas
            h = torch.tensor([self.lr_at(k), self.wd, b1, b2, self.eps, 1 - b1 ** k, 1 - b2 ** k, self.max_norm if self.max_norm else 0.0],
                             dtype=torch.float32).pin_memory()
            with torch.cuda.device(self.device):
                self.hyper.copy_(h, non_blocking=True)
                lib.sumsq_partial(self.flat_grad, self.n, self.sumsq, self.nparts)
                lib.clip_coef(self.sumsq, self.nparts, self.hyper, self.normcoef)
                lib.adamw(self.flat, self.flat_grad, self.m, self.v, self.n, self.hyper, self.normcoef)
            out['lr'] = float(h[0])
            out['grad_norm'] = self.normcoef[0]
        return out

    def metrics(self):
        return reduce_metrics(self.loss_out, self.cmat, self.reducer.group)


def _loss_and_counts(logits: torch.Tensor, labels: torch.Tensor, num_classes: int):
    """(loss_out [mean CE over labels != -1, count], confusion matrix [C, C] int64: rows true, cols pred) of one batch, on the device."""
    rows = logits.numel() // num_classes
    dev = logits.device
    part = torch.empty((rows + 255) // 256, 2, device=dev, dtype=torch.float32)
    out = torch.zeros(2, device=dev, dtype=torch.float32)
    cm = torch.zeros(num_classes, num_classes, device=dev, dtype=torch.int64)
    with torch.cuda.device(dev):
        lib.ce_fwd_bwd(logits.reshape(rows, num_classes).contiguous().float(), labels.reshape(rows).float().contiguous(), rows, num_classes, part, out,
                       None, cm, 1.0)
    return out, cm


def confusion_matrix_from_logits(logits: torch.Tensor, labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """argmax + MulticlassConfusionMatrix(ignore_index=-1) on the device (trainer/main.py:49-59): rows true, cols pred."""
    return _loss_and_counts(logits, labels, num_classes)[1]


class SleepModule:
    """Lightning-free mirror of SleepLightningModule (trainer/main.py:62-240): same constructor keywords, `_step`
    semantics (loss, confusion matrices per mode / signal subset / dataset, cross-rank sums), augmentations on device.
    `optimizer` / `scheduler` partials are accepted for signature compatibility; the fused AdamW step is used instead.
    """

    def __init__(self, model: Wav2Sleep, criterion=None, optimizer=None, aux_metrics=None, scheduler=None, debug_level=2,
                 on_step: bool = False, on_epoch: bool = True, num_classes: int = 4, masker: SignalMasker | None = None,
                 flip_polarity: bool = True, causal: bool = False, lr: float = 1e-3, weight_decay: float = 1e-4, max_norm: float = 1.0,
                 process_group=None, accumulate_grad_batches: int = 1):
        self.model = model
        self.num_classes = num_classes
        self.masker = masker if isinstance(model, Wav2Sleep) else None
        self.flip_polarity = flip_polarity
        self.causal = causal
        self.unified = isinstance(model, Wav2Sleep) and len(model.signal_encoders) > 1   # trainer/main.py:106
        self.aux_outputs = {mode: defaultdict(lambda: defaultdict(lambda: 0)) for mode in ('train', 'val', 'test')}
        fused = isinstance(model, Wav2Sleep) and model.fused_ok()
        kw = dict(lr=lr, weight_decay=weight_decay, max_norm=max_norm, process_group=process_group, accumulate=accumulate_grad_batches)
        # Trainer(accumulate_grad_batches=...), scripts/train.py:59-76; SleepPPGNet and the non-production configurations: the generic path
        self.trainer = FusedTrainStep(model, **kw) if fused else GenericTrainStep(model, num_classes=num_classes, **kw)

    def forward(self, x):
        """trainer/main.py:108-114: SleepPPGNet takes the batch's one signal as a tensor."""
        if not isinstance(self.model, Wav2Sleep):
            if len(x) != 1:
                raise ValueError(f'{x.keys()=} but expected unimodal input!')
            x = x[list(x.keys())[0]]
        return self.model(x)

    def _forward_subsets(self, x, subsets) -> dict:
        if isinstance(self.trainer, FusedTrainStep):
            return self.model.forward_subsets(x, subsets)   # every encoder once for all subsets
        return {(tuple(sub) if sub is not None else None): self.forward(x if sub is None else {s: x[s] for s in sub}) for sub in subsets}

    def on_after_batch_transfer(self, batch, training: bool = True):
        x, y = batch
        if training:  # one fused pass per signal (csrc/input_pipe.hip) instead of torch indexing; same sampling rule
            from .inputs import augment_
            augment_(x, flip_polarity=self.flip_polarity, masker=self.masker if self.unified else None)
        return x, y

    def training_step(self, batch, ds_name: str = 'all'):
        x, y = batch
        self.model.train()
        out = self.trainer.step(x, y)
        _, _, cm = self.trainer.metrics()
        self.aux_outputs['train'][None][ds_name] += cm
        return out['loss']

    @torch.no_grad()
    def eval_step(self, batch, mode: str = 'val', signals: tuple | None = None, ds_name: str = 'all'):
        x, y = batch
        if signals is not None:
            x = {s: x[s] for s in signals}
        self.model.eval()
        logits = self.forward(x)
        out, cm = _loss_and_counts(logits, y, self.num_classes)
        _, _, cm = reduce_metrics(out, cm, self.trainer.reducer.group)
        prefix = '_'.join(signals) if signals is not None else (None if self.unified else '_'.join(x.keys()))
        self.aux_outputs[mode][prefix][ds_name] += cm
        return out[0]

    @torch.no_grad()
    def validation_step(self, batch, ds_name: str = 'all', mode: str = 'val', combined: bool = False):
        """trainer/main.py:188-224: all-signal evaluation plus the ECG / ECG+THX / PPG / PPG+THX subset evaluations the
        reference runs per dataset -- here from ONE pass over the encoders (Wav2Sleep.forward_subsets)."""
        x, y = batch
        self.model.eval()
        subsets = [None]
        valid = self.model.valid_signals if self.unified else []
        if self.unified and not combined:
            if 'ECG' in x and 'ECG' in valid:
                subsets.append(('ECG',))
                if 'THX' in x and 'THX' in valid and (mode == 'test' or ds_name in ('shhs', 'mesa')):
                    subsets.append(('ECG', 'THX'))
            if 'PPG' in x and 'PPG' in valid and ds_name in ('mesa', 'cfs', 'ccshs', 'chat'):
                subsets.append(('PPG',))
                if 'THX' in x and 'THX' in valid and ds_name in ('mesa',):
                    subsets.append(('PPG', 'THX'))
        logits = self._forward_subsets(x, subsets)
        losses = {}
        for sub, lg in logits.items():
            out, cm = _loss_and_counts(lg, y, self.num_classes)
            _, _, cm = reduce_metrics(out, cm, self.trainer.reducer.group)
            prefix = '_'.join(sub) if sub is not None else (None if self.unified else '_'.join(x.keys()))
            self.aux_outputs[mode][prefix][ds_name] += cm
            losses[sub] = out[0]
        return losses

    @torch.no_grad()
    def predict_step(self, batch):
        """trainer/main.py:226-240: predictions from ECG, ECG+THX and all modalities."""
        x, y = batch
        self.model.eval()
        out = {'labels': y}
        if 'ECG' in x:
            out['preds_ECG'] = self.model({'ECG': x['ECG']}).argmax(dim=-1)
        if 'ECG' in x and 'THX' in x:
            out['preds_ECG_THX'] = self.model({'ECG': x['ECG'], 'THX': x['THX']}).argmax(dim=-1)
        out['preds'] = self.model(x).argmax(dim=-1)
        return out


SleepLightningModule = SleepModule
