"""CPU oracle for the wav2sleep hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain restatement, on stock PyTorch CPU fp32 ops, of the arithmetic the reference
runs for its train step / inference forward.  It is *only* allowed to be imported by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, and there only as the checker /
the reported CPU baseline -- never by ``wav2sleep_amd`` (the product path has no CPU fallback).

Pinning: every function here is checked against golden vectors produced by importing the real
reference modules in the build container (``tests/golden/make_goldens.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).  The one exception is
``confusion_matrix`` (the reference delegates to torchmetrics, which is not installed):
PARITY UNPINNED for that function only; it is cross-checked with a hand-computed example.

Each function cites the reference file:line it follows (paths relative to /root/reference).
Everything is functional: weights come in as a ``state_dict``-style ``dict[str, Tensor]`` with
the reference's key names, so the same dict drives the reference, this oracle and the HIP path.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

# src/wav2sleep/settings.py:13-26
SAMPLES_PER_EPOCH = {'ABD': 256, 'THX': 256, 'ECG': 1024, 'PPG': 1024, 'EOG-L': 4096, 'EOG-R': 4096}
# src/wav2sleep/settings.py:52-56
INTEGER_LABEL_MAPS = {4: {0: 0, 1: 1, 2: 1, 3: 2, 4: 3}, 5: {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}}


@dataclass
class ModelConfig:
    """Resolved hyper-parameters (scripts/config/model/wav2sleep.yaml + config/main.yaml)."""

    signal_map: dict = field(default_factory=lambda: {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'})
    feature_dim: int = 128
    num_classes: int = 4
    initial_channels: int = 16
    max_channels: int = 128
    mixer_layers: int = 2
    mixer_nhead: int = 8
    mixer_dim_ff: int = 512
    seq_blocks: int = 2
    seq_dilations: int = 6
    seq_kernel: int = 7
    instance_eps: float = 1e-2  # models/wav2sleep.py:213-215
    causal: bool = False  # scripts/config/main.yaml:22 `causal` (with model yaml `chunk_causal: False`): causal-padded convolutions
    embed_signals: bool = False  # SignalEncoders(embed_signals=...): nn.Embedding row per signal added to its encoder output
    output_norm: bool = False  # SignalEncoder(output_norm=True): nn.LayerNorm(feature_dim) on the encoder output (wav2sleep.py:232-233,266)
    use_residual: bool = True  # ConvBlock1D(use_residual=...): the 1x1/stride-2 branch and its `downsample` weight (blocks.py:49-55,67-68)
    register_tokens: int = 0  # MultiModalAttentionEmbedder(register_tokens=R): R learnable tokens next to the CLS token
    chunk_causal: bool = False  # SignalEncoders(chunk_causal=...): with causal=True the encoders see one 30-s epoch at a time instead
    layer_eps: float = 1e-5  # nn.LayerNorm default / models/utils.py:12

    def encoder_channels(self, signal: str) -> list[int]:
        """models/wav2sleep.py:198-201."""
        spe = SAMPLES_PER_EPOCH[signal]
        nb = int(math.log2(spe)) - 2
        return [min(self.initial_channels * 2 ** (i // 2), self.max_channels) for i in range(nb)]


# ----------------------------------------------------------------------------------------------
# Weights: deterministic, reference-independent generator (so fixtures need not store weights)
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: ModelConfig) -> dict[str, tuple]:
    """State-dict key/shape schema of reference `Wav2Sleep` (probed)."""
    shapes: dict[str, tuple] = {}
    Fd = cfg.feature_dim
    done = set()
    for sig, enc in cfg.signal_map.items():
        if enc in done:
            continue
        done.add(enc)
        cin = 1
        chans = cfg.encoder_channels(sig)
        for i, c in enumerate(chans):
            p = f'signal_encoders.encoders.{enc}.cnn.{i}.'
            shapes[p + 'conv1.conv.weight'] = (c, cin, 3)
            shapes[p + 'conv2.conv.weight'] = (c, c, 3)
            shapes[p + 'conv3.conv.weight'] = (c, c, 3)
            if cfg.use_residual:
                shapes[p + 'downsample.weight'] = (c, cin, 1)
            cin = c
        shapes[f'signal_encoders.encoders.{enc}.linear.weight'] = (Fd, 4 * chans[-1])
        shapes[f'signal_encoders.encoders.{enc}.linear.bias'] = (Fd,)
        if cfg.output_norm:
            shapes[f'signal_encoders.encoders.{enc}.output_norm.weight'] = (Fd,)
            shapes[f'signal_encoders.encoders.{enc}.output_norm.bias'] = (Fd,)
    for l in range(cfg.mixer_layers):
        p = f'epoch_mixer.transformer_encoder.layers.{l}.'
        shapes[p + 'self_attn.in_proj_weight'] = (3 * Fd, Fd)
        shapes[p + 'self_attn.in_proj_bias'] = (3 * Fd,)
        shapes[p + 'self_attn.out_proj.weight'] = (Fd, Fd)
        shapes[p + 'self_attn.out_proj.bias'] = (Fd,)
        shapes[p + 'linear1.weight'] = (cfg.mixer_dim_ff, Fd)
        shapes[p + 'linear1.bias'] = (cfg.mixer_dim_ff,)
        shapes[p + 'linear2.weight'] = (Fd, cfg.mixer_dim_ff)
        shapes[p + 'linear2.bias'] = (Fd,)
        shapes[p + 'norm1.weight'] = (Fd,)
        shapes[p + 'norm1.bias'] = (Fd,)
        shapes[p + 'norm2.weight'] = (Fd,)
        shapes[p + 'norm2.bias'] = (Fd,)
    shapes['epoch_mixer.register_tokens'] = (1, 1, Fd, cfg.register_tokens + 1)
    if cfg.embed_signals:
        shapes['signal_encoders.embedder.weight'] = (len(cfg.signal_map), Fd)  # models/wav2sleep.py:130-131
    for b in range(cfg.seq_blocks):
        for j in range(cfg.seq_dilations):
            p = f'sequence_mixer.dilated_convs.{b}.conv_layers.{j}.'
            shapes[p + 'conv.weight'] = (Fd, Fd, cfg.seq_kernel)
            shapes[p + 'norm.weight'] = (1, Fd, 1)
            shapes[p + 'norm.bias'] = (1, Fd, 1)
    shapes['classifier.weight'] = (cfg.num_classes, Fd)
    shapes['classifier.bias'] = (cfg.num_classes,)
    return shapes


def make_state_dict(cfg: ModelConfig, seed: int = 0, scale: float = 1.0) -> dict[str, Tensor]:
    """Deterministic synthetic weights (CPU generator; identical on every box with this torch).

    Kaiming-uniform-like fan-in scaling so activations stay O(1) through the depth; norm
    weights near 1, biases small.  Not the reference's init -- goldens load THIS dict into the
    reference modules, so the reference is evaluated on exactly these weights.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in sorted(param_shapes(cfg).items()):
        if k.endswith('norm.weight') or k.endswith('norm1.weight') or k.endswith('norm2.weight'):  # incl. output_norm.weight
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith('bias'):
            t = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith('register_tokens') or k.endswith('embedder.weight'):
            t = torch.randn(shp, generator=g)
        else:
            fan_in = int(np.prod(shp[1:]))
            bound = scale * math.sqrt(3.0 / fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * bound
        sd[k] = t.float().contiguous()
    return sd


def make_inputs(cfg: ModelConfig, batch: int, epochs: int, seed: int = 1234, missing: dict | None = None,
                frac_unlabelled: float = 0.1):
    """Synthetic overnight batch: z-scored-like N(0,1) signals, labels with -1 holes.

    missing: {signal: [batch indices]} rows replaced by -inf (data/dataset.py:170-173, masker.py:49-50).
    """
    g = torch.Generator().manual_seed(seed)
    x = {}
    for sig in cfg.signal_map:
        x[sig] = torch.randn(batch, epochs * SAMPLES_PER_EPOCH[sig], generator=g)
    y = torch.randint(0, cfg.num_classes, (batch, epochs), generator=g).float()
    holes = torch.rand(batch, epochs, generator=g) < frac_unlabelled
    y[holes] = -1.0
    if missing:
        for sig, rows in missing.items():
            x[sig][rows] = float('-inf')
    return x, y


# ----------------------------------------------------------------------------------------------
# Forward pieces
# ----------------------------------------------------------------------------------------------
def gelu(x: Tensor) -> Tensor:
    """nn.GELU(approximate='none') -- models/utils.py:61-74."""
    return F.gelu(x)


def instance_norm(x_BCL: Tensor, eps: float) -> Tensor:
    """nn.InstanceNorm1d(affine=False, track_running_stats=False) -- models/utils.py:89-92."""
    return F.instance_norm(x_BCL, eps=eps)


def causal_conv1d(x_BCL: Tensor, w: Tensor, stride: int = 1, dilation: int = 1) -> Tensor:
    """The causal branch of ConvLayer1D (models/blocks.py:150-152,178-182): symmetric padding (k-1)*dil, then the last
    max(pad-(stride-1), 0) outputs are dropped, i.e. out[j] = sum_k w[k] x[j*stride - (k_size-1-k)*dil] with zeros before the start."""
    pad = (w.size(-1) - 1) * dilation
    y = F.conv1d(x_BCL, w, None, stride=stride, padding=pad, dilation=dilation)
    trim = max(pad - (stride - 1), 0)
    return y[:, :, :-trim] if trim > 0 else y


def conv_layer_in(x_BCL: Tensor, w: Tensor, stride: int, eps: float, causal: bool = False) -> Tensor:
    """ConvLayer1D (k=3, pad=1 or causal, no bias) -> InstanceNorm -> GELU.  models/blocks.py:173-186."""
    y = causal_conv1d(x_BCL, w, stride) if causal else F.conv1d(x_BCL, w, None, stride=stride, padding=1)
    return gelu(instance_norm(y, eps))


def conv_block(sd: dict, p: str, x_BCL: Tensor, eps: float, taps: dict | None = None, causal: bool = False) -> Tensor:
    """ConvBlock1D.forward -- models/blocks.py:57-71 (the 1x1/stride-2 residual conv is the same in causal mode)."""
    h1 = conv_layer_in(x_BCL, sd[p + 'conv1.conv.weight'], 1, eps, causal)
    h2 = conv_layer_in(h1, sd[p + 'conv2.conv.weight'], 1, eps, causal)
    h3 = conv_layer_in(h2, sd[p + 'conv3.conv.weight'], 2, eps, causal)
    out = gelu(h3 + F.conv1d(x_BCL, sd[p + 'downsample.weight'], None, stride=2)) if (p + 'downsample.weight') in sd else gelu(h3)
    if taps is not None:
        taps[p + 'out'] = out
    return out


def signal_encoder(sd: dict, cfg: ModelConfig, enc: str, sig: str, x_BT: Tensor, taps: dict | None = None) -> Tensor:
    """SignalEncoder.forward -- models/wav2sleep.py:235-267: the whole-sequence path (:256-261), which is also what `causal=True` with
    `chunk_causal=False` (scripts/config/model/wav2sleep.yaml:10-11) runs, with causal convolutions inside the blocks (:204,220)."""
    spe = SAMPLES_PER_EPOCH[sig]
    if x_BT.size(-1) % spe:
        raise ValueError(f'Input length {x_BT.size(-1)} must be divisible by samples_per_epoch={spe}.')
    B = x_BT.size(0)
    S = x_BT.size(-1) // spe
    nb = len(cfg.encoder_channels(sig))
    if cfg.causal and cfg.chunk_causal:
        # quasi-causal: every epoch is encoded on its own with the ordinary (symmetric) padding -- models/wav2sleep.py:204,248-255
        y = x_BT.reshape(B * S, 1, spe)
        for i in range(nb):
            y = conv_block(sd, f'signal_encoders.encoders.{enc}.cnn.{i}.', y, cfg.instance_eps, taps, False)
        y = y.transpose(-1, -2).reshape(B, S, y.size(1) * 4)
    else:
        y = x_BT.unsqueeze(1)
        for i in range(nb):
            y = conv_block(sd, f'signal_encoders.encoders.{enc}.cnn.{i}.', y, cfg.instance_eps, taps, cfg.causal)
        epoch_dim = y.size(1) * 4
        y = y.transpose(-1, -2).reshape(B, -1, epoch_dim)
    y = gelu(F.linear(y, sd[f'signal_encoders.encoders.{enc}.linear.weight'], sd[f'signal_encoders.encoders.{enc}.linear.bias']))
    if cfg.output_norm:  # wav2sleep.py:266
        y = F.layer_norm(y, (y.size(-1),), sd[f'signal_encoders.encoders.{enc}.output_norm.weight'],
                         sd[f'signal_encoders.encoders.{enc}.output_norm.bias'], cfg.layer_eps)
    return y


def signal_encoders(sd: dict, cfg: ModelConfig, x: dict[str, Tensor], taps: dict | None = None) -> dict[str, Tensor]:
    """SignalEncoders.forward -- models/wav2sleep.py:146-161 (embed_signals=False)."""
    z = {}
    for sig, x_BT in x.items():
        if sig not in cfg.signal_map:
            raise ValueError(f'Unknown signal {sig}')
        mask_B = torch.isinf(x_BT[:, 0])
        x_BT = torch.where(torch.isinf(x_BT), 0.0, x_BT)
        z_BSF = signal_encoder(sd, cfg, cfg.signal_map[sig], sig, x_BT, taps)
        z[sig] = torch.where(mask_B[:, None, None], float('-inf'), z_BSF)
        if cfg.embed_signals:  # wav2sleep.py:155-159: index = position of the signal among the sorted signal_map keys (:129)
            z[sig] = z[sig] + sd['signal_encoders.embedder.weight'][sorted(cfg.signal_map).index(sig)][None, None, :]
    return z


def encoder_layer(sd: dict, p: str, x_NDF: Tensor, pad_ND: Tensor, nhead: int, eps: float) -> Tensor:
    """nn.TransformerEncoderLayer(norm_first=True, batch_first=True, GELU), dropout off.

    torch semantics restated (models/wav2sleep.py:286-299 instantiates it): in_proj rows
    [0:F]=q, [F:2F]=k, [2F:3F]=v; softmax(q k^T / sqrt(hd) + (-inf on padded keys)); SURVEY App. A.
    """
    N, D, Fd = x_NDF.shape
    hd = Fd // nhead
    h = F.layer_norm(x_NDF, (Fd,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps)
    qkv = F.linear(h, sd[p + 'self_attn.in_proj_weight'], sd[p + 'self_attn.in_proj_bias'])
    q, k, v = qkv.split(Fd, dim=-1)
    q = q.view(N, D, nhead, hd).transpose(1, 2)
    k = k.view(N, D, nhead, hd).transpose(1, 2)
    v = v.view(N, D, nhead, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(pad_ND[:, None, None, :], float('-inf'))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(N, D, Fd)
    x = x_NDF + F.linear(o, sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'])
    h = F.layer_norm(x, (Fd,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps)
    ff = F.linear(gelu(F.linear(h, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])),
                  sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
    return x + ff


def epoch_mixer(sd: dict, cfg: ModelConfig, z: dict[str, Tensor]) -> Tensor:
    """MultiModalAttentionEmbedder.forward -- models/wav2sleep.py:301-346."""
    signals = sorted(z.keys())
    if len(signals) == 0:
        raise ValueError('No signals provided to MultiModalAttentionEmbedder.')
    zs, ms = [], []
    for s in signals:
        z_BSF = z[s]
        m_B = torch.isinf(z_BSF).any(dim=2).any(dim=1)
        zs.append(torch.where(m_B[:, None, None], 0.0, z_BSF))
        ms.append(m_B)
    z_BSFC = torch.stack(zs, dim=-1)
    m_BC = torch.stack(ms, dim=-1)
    B, S, Fd, C = z_BSFC.shape
    if Fd != cfg.feature_dim:
        raise ValueError(f'Feature dimension {Fd} does not match {cfg.feature_dim}.')
    cls = sd['epoch_mixer.register_tokens']
    z_BSFD = torch.cat([cls.repeat(B, S, 1, 1), z_BSFC], dim=-1)
    R1 = cls.size(-1)  # CLS + register tokens, always attendable (wav2sleep.py:334-337)
    D = C + R1
    m_BD = torch.cat([torch.zeros(B, R1, dtype=torch.bool), m_BC], dim=-1)
    x = z_BSFD.flatten(0, 1).permute(0, 2, 1)  # [N, D, F]
    pad = m_BD[:, None, :].repeat(1, S, 1).flatten(0, 1)  # [N, D]
    for l in range(cfg.mixer_layers):
        x = encoder_layer(sd, f'epoch_mixer.transformer_encoder.layers.{l}.', x, pad, cfg.mixer_nhead, cfg.layer_eps)
    return x[:, 0, :].reshape(B, S, Fd)


def conv_layer_norm(x_BCT: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """ConvLayerNorm.forward -- models/utils.py:17-23."""
    mu = x_BCT.mean(1, keepdim=True)
    var = (x_BCT - mu).pow(2).mean(1, keepdim=True)
    return w * ((x_BCT - mu) / torch.sqrt(var + eps)) + b


def sequence_cnn(sd: dict, cfg: ModelConfig, z_BSF: Tensor) -> Tensor:
    """SequenceCNN.forward + DilatedConvBlock.forward -- wav2sleep.py:379-390, blocks.py:115-126."""
    x = z_BSF.transpose(-1, -2)
    k = cfg.seq_kernel
    for b in range(cfg.seq_blocks):
        h = x
        for j in range(cfg.seq_dilations):
            d = 2 ** j
            p = f'sequence_mixer.dilated_convs.{b}.conv_layers.{j}.'
            pad = (k + (k - 1) * (d - 1)) // 2
            h = causal_conv1d(h, sd[p + 'conv.weight'], 1, d) if cfg.causal else F.conv1d(h, sd[p + 'conv.weight'], None, padding=pad, dilation=d)
            h = gelu(conv_layer_norm(h, sd[p + 'norm.weight'], sd[p + 'norm.bias'], cfg.layer_eps))
        x = gelu(h + x)
    return x.transpose(-1, -2)


def forward(sd: dict, cfg: ModelConfig, x: dict[str, Tensor], taps: dict | None = None) -> Tensor:
    """Wav2Sleep.forward (eval / dropout-free) -- models/wav2sleep.py:48-67.  -> logits [B,S,nc]."""
    z = signal_encoders(sd, cfg, x, taps)
    if taps is not None:
        for s, v in z.items():
            taps[f'z.{s}'] = v
    m = epoch_mixer(sd, cfg, z)
    if taps is not None:
        taps['mixer'] = m
    q = sequence_cnn(sd, cfg, m)
    if taps is not None:
        taps['seq'] = q
    return F.linear(q, sd['classifier.weight'], sd['classifier.bias'])


def predict(sd: dict, cfg: ModelConfig, x: dict[str, Tensor]) -> Tensor:
    """Wav2Sleep.predict -- models/wav2sleep.py:69-80."""
    return forward(sd, cfg, x).argmax(dim=2)


# ----------------------------------------------------------------------------------------------
# Loss, metrics, optimiser (callers of the path: trainer/main.py, stats.py, scheduler.py)
# ----------------------------------------------------------------------------------------------
def cross_entropy(logits_BSC: Tensor, y_BS: Tensor) -> Tensor:
    """reshape_for_loss + CrossEntropyLoss(mean, ignore_index=-1) -- trainer/main.py:116-119,162-163."""
    nc = logits_BSC.size(-1)
    return F.cross_entropy(logits_BSC.reshape(-1, nc), y_BS.reshape(-1).long(), ignore_index=-1)


def confusion_matrix(pred_N: Tensor, true_N: Tensor, num_classes: int) -> Tensor:
    """torchmetrics MulticlassConfusionMatrix(ignore_index=-1): rows=true, cols=pred.

    trainer/main.py:49-59,88.  PARITY UNPINNED (torchmetrics absent); trivial integer op.
    """
    true_N = true_N.reshape(-1).long()
    pred_N = pred_N.reshape(-1).long()
    keep = true_N != -1
    idx = true_N[keep] * num_classes + pred_N[keep]
    return torch.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def confusion_accuracy(cmat) -> float:
    """stats.py:9-11."""
    cmat = np.asarray(cmat)
    return float(np.trace(cmat) / np.sum(cmat))


def cohens_kappa(cmat, n_classes: int = 4) -> float:
    """stats.py:14-30: 1 - (disagreement mass observed) / (disagreement mass of the chance table built from the marginals)."""
    m = np.asarray(cmat, dtype=np.float64)
    chance = np.outer(m.sum(axis=0), m.sum(axis=1)) / m.sum()
    off_diag = 1.0 - np.eye(n_classes)   # every disagreement costs 1
    return float(1.0 - (off_diag * m).sum() / (off_diag * chance).sum())


def exp_warmup_lr(step: int, lr_max: float = 1e-3, warmup_steps: int = 2000, tau: float = 10000.0) -> float:
    """ExpWarmUpScheduler.get_lr at optimiser step `step` (1-based) -- trainer/scheduler.py:23-32."""
    if step <= warmup_steps:
        return lr_max * (step / warmup_steps)
    return lr_max * math.exp(-(step - warmup_steps) / tau)


def clip_grad_norm(grads: dict[str, Tensor], max_norm: float = 1.0) -> float:
    """torch.nn.utils.clip_grad_norm_(L2) in place; returns total norm.  training/main.yaml:21-22."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(total)


def adamw_step(params: dict[str, Tensor], grads: dict[str, Tensor], state: dict, lr: float,
               wd: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.AdamW single step, restated (optimizer/adamw.yaml; trainer/main.py:273-275)."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault('m.' + k, torch.zeros_like(p))
        v = state.setdefault('v.' + k, torch.zeros_like(p))
        p.mul_(1 - lr * wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def loss_and_grads(sd: dict, cfg: ModelConfig, x: dict, y: Tensor):
    """Forward + CE + autograd backward on the restatement.  -> (loss, logits, grads dict)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = forward(params, cfg, x)
    loss = cross_entropy(logits, y)
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
    return float(loss.detach()), logits.detach(), grads


def train_step(sd: dict, cfg: ModelConfig, x: dict, y: Tensor, opt_state: dict, max_norm: float = 1.0,
               lr_max: float = 1e-3, wd: float = 1e-4, lr: float | None = None):
    """One optimiser step as the Lightning loop runs it: fwd, CE, bwd, clip, AdamW, lr(k).
    `lr` given: no scheduler (`scheduler: null`), that constant learning rate.

    Mutates `sd` and `opt_state` in place; returns (loss, logits, grad_norm, lr).
    """
    loss, logits, grads = loss_and_grads(sd, cfg, x, y)
    gn = clip_grad_norm(grads, max_norm)
    lr = exp_warmup_lr(opt_state.get('step', 0) + 1, lr_max) if lr is None else lr
    adamw_step(sd, grads, opt_state, lr, wd)
    return loss, logits, gn, lr


def zscore_normalize(x_T: Tensor, eps: float = 1e-6) -> Tensor:
    """ParquetDataset._zscore_normalize for one recording -- data/dataset.py:76-87.
    Pinned by tests/golden/dataset.npz (the reference's own function run by tests/golden/make_goldens_r2.py)."""
    if x_T.numel() == 0 or not torch.isfinite(x_T).all():
        return x_T
    mu = torch.mean(x_T)
    std = torch.std(x_T)
    std = std if std > eps else torch.tensor(eps, dtype=x_T.dtype)
    return (x_T - mu) / std


def map_labels(stages: Tensor, num_classes: int) -> Tensor:
    """df[LABEL].map(INTEGER_LABEL_MAPS[nc]).fillna(-1) -- data/dataset.py:174-182, settings.py:52-56."""
    m = INTEGER_LABEL_MAPS[num_classes]
    out = torch.full_like(stages, -1.0)
    for k, v in m.items():
        out[stages == k] = float(v)
    return out


# ----------------------------------------------------------------------------------------------
# Augmentations (explicit-draw restatements; the reference samples with torch RNG on device)
# ----------------------------------------------------------------------------------------------
def apply_polarity(x: dict[str, Tensor], sign_BC: dict[str, Tensor]) -> dict[str, Tensor]:
    """invert_signals with the +-1 draws given explicitly -- trainer/main.py:342-353."""
    return {k: v * sign_BC[k][:, None] for k, v in x.items()}


def apply_masker(x: dict[str, Tensor], keep_draw: dict[str, Tensor], backup_pick: Tensor, backups: list[str]):
    """SignalMasker.__call__ with the Bernoulli keep draws and the backup choice given explicitly.

    trainer/masker.py:10-51.  keep_draw[s][b] = True keeps; if a sample ends with no channel,
    the channel named backups[backup_pick[b]] (must be available) is kept instead.
    """
    sigs = list(x.keys())
    z_BC = torch.stack([torch.isinf(x[s][:, 0]) for s in sigs], dim=-1)
    if z_BC.all(dim=-1).any():
        raise ValueError('Found batch element with all signals unavailable.')
    m_BC = torch.stack([keep_draw[s] for s in sigs], dim=-1).clone()
    all_zero = torch.logical_or(z_BC, ~m_BC).all(dim=-1)
    for b in torch.nonzero(all_zero).flatten().tolist():
        name = backups[int(backup_pick[b])]
        j = sigs.index(name)
        if z_BC[b, j]:
            raise ValueError('No backup channels for stochastic sampling were available')
        m_BC[b] = False
        m_BC[b, j] = True
    out = {}
    for j, s in enumerate(sigs):
        v = x[s].clone()
        v[~m_BC[:, j]] = float('-inf')
        out[s] = v
    return out


def causal_rolling_normalize(signal, sampling_freq: float, tau_seconds: float = 900.0, eps: float = 1e-6, outlier_threshold_sigma: float = 4.0,
                             baseline_tau_seconds: float | None = None, min_sigma: float = 0.1):
    """data/normalization.py:18-80 (the EMA loop) and :106-230 (alphas, warm-up estimates, final division), as a plain Python loop over
    a numpy array: -> (normalised fp64 array, outlier mask).  Slow by construction (checker for small inputs)."""
    x = np.asarray(signal)
    n = len(x)
    if n == 0:
        return x.astype(np.float64), np.zeros(0, dtype=bool)
    tau_mu = tau_seconds if baseline_tau_seconds is None else baseline_tau_seconds
    dt = 1.0 / sampling_freq
    a_mu, a_var = dt / tau_mu, dt / tau_seconds
    floor2 = min_sigma * min_sigma
    warm = max(1, min(int(min(tau_mu, tau_seconds) * sampling_freq), n // 10))   # :190-195
    mu = np.empty(n)
    var = np.empty(n)
    hit = np.zeros(n, dtype=bool)
    mu[0] = float(np.mean(x[:warm]))
    var[0] = max(max(float(np.var(x[:warm])), floor2), floor2, eps)              # :199 and :54
    xs = x.astype(np.float64)
    for t in range(1, n):
        mu[t] = a_mu * xs[t] + (1.0 - a_mu) * mu[t - 1]
        r = xs[t] - mu[t]
        lim = outlier_threshold_sigma * math.sqrt(max(var[t - 1], floor2))
        if abs(r) > lim:
            hit[t] = True
            r = lim if r > lim else -lim
        var[t] = a_var * r * r + (1.0 - a_var) * var[t - 1]
    return (x - mu) / np.sqrt(np.maximum(var, floor2)), hit
