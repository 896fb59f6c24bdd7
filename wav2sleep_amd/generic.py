"""Generic (untuned) inference path: every module variant the reference can be configured into besides its shipped production model.

`models/utils.py:26-96` offers batch / layer / rms / group / instance / no normalisation and ReLU / LeakyReLU / GELU / SiLU / linear
activations; `MultiModalAttentionEmbedder` and `SequenceCNN` take any feature size and head count (the reference's own
tests/model/test_causality.py builds feature_dim 16, ReLU, BatchNorm, 4 layers x 4 heads); `models/ppgnet.py` is a ninth-of-a-kind CNN
from the same blocks.  The production model (GELU, instance / layer norm, 128 features, 16-wide heads) runs on the fused kernels
(engine.py); everything else runs here: the reference's call graph, layer by layer, on channels-last device tensors, every convolution and
GEMM through `w2s_conv_forward` (no prologue; contractions wider than 128 channels as accumulating launches) and the norms / activations /
attention core through the three kernels of csrc/generic.hip.  Forward only: these variants have no backward kernels, the result carries
no autograd graph.  Host-side torch is used for plumbing only: folding per-(sample, channel) statistics and affine parameters into
(scale, shift) vectors -- [B, C] scalars --, zero-padding the one-channel input, slicing weights.

Reference call sites mirrored: ConvLayer1D.forward (blocks.py:173-186), ConvBlock1D.forward (:57-71), DilatedConvBlock.forward (:115-126),
SignalEncoder.forward (wav2sleep.py:235-267), SignalEncoders.forward (:146-161), MultiModalAttentionEmbedder.forward (:301-346),
SequenceCNN.forward (:379-390), Wav2Sleep.forward (:48-67), SleepPPGNet.forward (ppgnet.py:50-80).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import lib
from .settings import COLS_TO_SAMPLES_PER_EPOCH


def _cdiv(a, b):
    return (a + b - 1) // b


def _act_code(name: str) -> int:
    if name not in lib.ACT:
        raise ValueError(f'{name=} is unsupported.')
    return lib.ACT[name]


def _pow2_chunk(c: int) -> int:
    """Contraction width of one launch: the kernels take a power-of-two channel count in [16, 128]."""
    if c > 128:
        if c % 128:
            raise NotImplementedError(f'{c} input channels: widths above 128 must be multiples of 128')
        return 128
    if c < 16 or c & (c - 1):
        raise NotImplementedError(f'{c} input channels: the generic kernels take powers of two in [16, 128] (or multiples of 128)')
    return c


def act_(x: torch.Tensor, name: str, slope: float = 0.01) -> torch.Tensor:
    """In-place activation of a contiguous [..., C] tensor."""
    code = _act_code(name)
    if code:
        Cc = x.shape[-1]
        lib.affine_act(x, Cc, None, None, 0, x, Cc, 1, x.numel() // Cc, Cc, code, slope)
    return x


class GenericForward:
    """Stateless walker over the parameter containers of wav2sleep.py (same attribute names as the reference modules)."""

    def __init__(self, training: bool = False, seed: int = 0):
        self.training = training
        self.seed = seed
        self._site = 0
        lib.load()

    # ------------------------------------------------------------------ convolution + normalisation + activation
    def _conv(self, x, w, bias, L_out, *, stride, pad, dil, want_stats=None, eps=1e-5):
        """x [B, L_in, Cin] -> y [B, L_out, Cout]; left padding `pad` (zeros), taps j read x[t*stride + j*dil - pad].
        want_stats: None | 0 (mean, rstd) | 1 (E[y], E[y^2]) -> [B, Cout, 2]."""
        B, L_in, cin = x.shape
        cout, _, k = w.shape
        if cout % 16:
            raise NotImplementedError(f'{cout} output channels: the generic kernels produce multiples of 16')
        dev = x.device
        if cin == 1:   # zero-pad the one-channel input to the narrowest tile the matrix path takes
            x16 = torch.zeros(B, L_in, 16, device=dev, dtype=torch.float32)
            x16[..., 0] = x[..., 0]
            w16 = torch.zeros(cout, 16, k, device=dev, dtype=torch.float32)
            w16[:, 0] = w[:, 0]
            x, w, cin = x16, w16, 16
        ck = _pow2_chunk(cin)
        nchunk = cin // ck
        if want_stats is not None and nchunk > 1:
            raise NotImplementedError('statistics-based norms on layers wider than 128 input channels')
        y = torch.empty(B, L_out, cout, device=dev, dtype=torch.float32)
        if k == 7 and stride == 1:
            mode = lib.MODE_DILATED
        elif dil == 1 and (k, stride) in ((3, 1), (3, 2), (1, 1), (1, 2)):
            mode = lib.MODE_CONTIG
        elif k == stride and k in (3, 4) and dil == 1:
            mode = lib.MODE_DILATED
        else:
            raise NotImplementedError(f'kernel_size={k}, stride={stride}, dilation={dil}: no generic kernel')
        stats = None
        for q in range(nchunk):
            wq = w[:, q * ck:(q + 1) * ck, :].permute(0, 2, 1).contiguous()   # [cout][k][ck]
            xq = x if nchunk == 1 else x[..., q * ck:]                          # pointer offset; row stride stays the full width
            a = lib.conv_args(x=xq, w=wq, y=y, B=B, L_in=L_in, L_out=L_out, cin=ck, cout=cout, taps=k, stride=stride, pad=pad, dil=dil, mode=mode,
                              ldx=cin, epi=lib.EPI_BIAS if (bias is not None and q == 0) else (lib.EPI_STATS if want_stats is not None else lib.EPI_PLAIN),
                              bias=bias if q == 0 else None, accumulate=q > 0)
            if want_stats is not None:
                nt = _cdiv(L_out, lib.conv_tile_of(a))
                part = torch.empty(B, nt, 2, cout, device=dev, dtype=torch.float32)
                lib.set_part(a, part)
                lib.conv_forward(a)
                stats = torch.empty(B, cout, 2, device=dev, dtype=torch.float32)
                lib.stats_finalize(part, B, nt, cout, L_out, eps, want_stats, stats)
            else:
                lib.conv_forward(a)
        return y, stats

    def _norm_act(self, layer, y, stats, act_name):
        """norm -> activation of one ConvLayer1D output y [B, L, C], in place (blocks.py:183-185)."""
        B, L, Cc = y.shape
        rows = B * L
        act = _act_code(act_name)
        norm = layer.norm
        kind = layer.norm_name
        if kind is None or kind == 'weight':
            lib.affine_act(y, Cc, None, None, 0, y, Cc, L, rows, Cc, act, 0.01)
        elif kind == 'instance':   # stats = (mean, rstd) per (b, c)
            scale = stats[..., 1].contiguous()
            shift = (-stats[..., 0] * stats[..., 1]).contiguous()
            lib.affine_act(y, Cc, scale, shift, Cc, y, Cc, L, rows, Cc, act, 0.01)
        elif kind == 'batch':
            if self.training and norm.training:   # batch statistics over (B, L) + running-statistics update (nn.BatchNorm1d)
                m = stats[..., 0].mean(0)
                var = (stats[..., 1].mean(0) - m * m).clamp_min(0)
                with torch.no_grad():
                    n = B * L
                    mom = norm.momentum if norm.momentum is not None else 1.0 / float(norm.num_batches_tracked + 1)
                    norm.running_mean.mul_(1 - mom).add_(m, alpha=mom)
                    norm.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
                    norm.num_batches_tracked += 1
            else:
                m, var = norm.running_mean, norm.running_var
            scale = (norm.weight.detach() / torch.sqrt(var + norm.eps)).contiguous()
            shift = (norm.bias.detach() - m * scale).contiguous()
            lib.affine_act(y, Cc, scale, shift, 0, y, Cc, L, rows, Cc, act, 0.01)
        elif kind == 'group':      # stats = (E[y], E[y^2]) per (b, c): pool over the channels of a group (equal lengths)
            gn = norm.norm
            G = gn.num_groups
            e1 = stats[..., 0].view(B, G, Cc // G).mean(2, keepdim=True)
            e2 = stats[..., 1].view(B, G, Cc // G).mean(2, keepdim=True)
            rstd = 1.0 / torch.sqrt((e2 - e1 * e1).clamp_min(0) + gn.eps)
            scale = (rstd.expand(B, G, Cc // G).reshape(B, Cc) * gn.weight.detach()[None, :]).contiguous()
            shift = (gn.bias.detach()[None, :] - e1.expand(B, G, Cc // G).reshape(B, Cc) * scale).contiguous()
            lib.affine_act(y, Cc, scale, shift, Cc, y, Cc, L, rows, Cc, act, 0.01)
        elif kind == 'layer':
            lib.rownorm_fwd(y, Cc, norm.weight.detach().reshape(Cc), norm.bias.detach().reshape(Cc), y, Cc, rows, Cc, norm.eps, False, act, 0.01)
        elif kind == 'rms':
            lib.rownorm_fwd(y, Cc, norm.weight.detach().reshape(Cc), None, y, Cc, rows, Cc, norm.eps, True, act, 0.01)
        else:
            raise ValueError(f'Normalisation with name={kind} unknown.')
        return y

    def _dropout_(self, x, p):
        if self.training and p > 0.0:
            self._site += 1
            lib.eltwise(lib.ELT_DROP, x, None, x, x.numel(), p, ((self.seed & 0xFFFFFFFF) << 16) ^ (self._site * 0x9E3779B1 & 0xFFFFFFFF))
        return x

    def conv_layer(self, layer, x):
        """ConvLayer1D.forward on channels-last x [B, L, Cin] (blocks.py:173-186)."""
        conv = layer.conv
        k, stride, dil = conv.kernel_size[0], conv.stride[0], conv.dilation[0]
        if conv.groups != 1:
            raise NotImplementedError('grouped convolutions')
        B, L, _ = x.shape
        if layer.causal:   # symmetric padding (k-1)*dil, then the right trim of blocks.py:178-182
            pad = (k - 1) * dil
            L_out = (L + 2 * pad - dil * (k - 1) - 1) // stride + 1 - max(pad - (stride - 1), 0)
        else:
            pad = conv.padding[0]
            L_out = (L + 2 * pad - dil * (k - 1) - 1) // stride + 1
        kind = layer.norm_name
        want = {'instance': 0, 'group': 1}.get(kind)
        if kind == 'batch' and self.training and layer.norm.training:
            want = 1
        eps = layer.norm.eps if kind == 'instance' else 0.0
        bias = conv.bias.detach() if conv.bias is not None else None
        if bias is not None and want is not None:
            raise NotImplementedError('a convolution bias in front of a statistics-based norm')
        y, stats = self._conv(x, conv.weight.detach(), bias, L_out, stride=stride, pad=pad, dil=dil, want_stats=want, eps=eps)
        self._norm_act(layer, y, stats, layer.activation_name)
        return self._dropout_(y, layer.dropout_p)

    def conv_block(self, block, x):
        """ConvBlock1D.forward (blocks.py:57-71)."""
        out = self.conv_layer(block.conv3, self.conv_layer(block.conv2, self.conv_layer(block.conv1, x)))
        if block.use_residual:
            B, L, _ = x.shape
            r, _ = self._conv(x, block.downsample.weight.detach(), None, out.shape[1], stride=2, pad=0, dil=1)
            lib.eltwise(lib.ELT_ADD, out, r, out, out.numel())
        return act_(out, block.activation_name)

    def dilated_block(self, block, x):
        """DilatedConvBlock.forward (blocks.py:115-126) on [B, S, F]."""
        out = x
        for layer in block.conv_layers:
            out = self.conv_layer(layer, out)
        out = self._dropout_(out, block.dropout.p)
        if out is x:
            out = x.clone()
        lib.eltwise(lib.ELT_ADD, out, x, out, out.numel())
        return act_(out, block.activation_name)

    # ------------------------------------------------------------------ dense layers / GEMMs
    def linear(self, x_rows, weight, bias, act_name='linear'):
        """y[rows, cout] = x[rows, cin] @ W^T + b, then the activation (any cin that is a power of two <= 128 or a multiple of 128)."""
        rows, cin = x_rows.shape
        cout = weight.shape[0]
        if cout % 16:
            raise NotImplementedError(f'{cout} output features: multiples of 16')
        ck = _pow2_chunk(cin)
        y = torch.empty(rows, cout, device=x_rows.device, dtype=torch.float32)
        w = weight.detach()
        for q in range(cin // ck):
            wq = w[:, q * ck:(q + 1) * ck].contiguous()
            xq = x_rows if cin == ck else x_rows[:, q * ck:]
            lib.conv_forward(lib.conv_args(x=xq, w=wq, y=y, B=1, L_in=rows, L_out=rows, cin=ck, cout=cout, taps=1, stride=1, pad=0, ldx=cin,
                                           epi=lib.EPI_BIAS if (bias is not None and q == 0) else lib.EPI_PLAIN,
                                           bias=bias.detach() if (bias is not None and q == 0) else None, accumulate=q > 0))
        return act_(y, act_name)

    # ------------------------------------------------------------------ the modules
    def signal_encoder(self, enc, x_BT):
        """SignalEncoder.forward (wav2sleep.py:235-267): [B, T] -> [B, S, feature_dim]."""
        spe = enc.samples_per_epoch
        B, T = x_BT.shape
        if T % spe != 0:
            raise ValueError(f'Input length {T} must be divisible by samples_per_epoch={spe}.')
        S = T // spe
        x = x_BT.contiguous().float()
        y = x.view(B * S, spe, 1) if (enc.causal and enc.chunk_causal) else x.view(B, T, 1)
        for block in enc.cnn:
            y = self.conv_block(block, y)
        # [B(*S), 4(*S), C] -> [B, S, 4C]: feature index = t_local * C + c, which in channels-last layout is the memory order
        feat = y.reshape(B * S, enc.epoch_dim)
        z = self.linear(feat, enc.linear.weight, enc.linear.bias, enc.activation_name)
        if isinstance(enc.output_norm, nn.LayerNorm):
            F_ = enc.feature_dim
            lib.rownorm_fwd(z, F_, enc.output_norm.weight.detach(), enc.output_norm.bias.detach(), z, F_, B * S, F_, enc.output_norm.eps, False, 0, 0.01)
        return z.view(B, S, enc.feature_dim)

    def signal_encoders(self, mod, x: dict) -> dict:
        """SignalEncoders.forward (wav2sleep.py:146-161)."""
        z = {}
        for name, x_BT in x.items():
            if name not in mod.signal_map:
                raise ValueError(f'Unknown signal {name}')
            mask_B = torch.isinf(x_BT[:, 0])
            xs = torch.where(torch.isinf(x_BT), torch.zeros_like(x_BT), x_BT)
            z_BSF = self.signal_encoder(mod.get_encoder(name), xs)
            z_BSF = torch.where(mask_B[:, None, None], float('-inf'), z_BSF)
            if mod.embed_signals:
                z_BSF = z_BSF + mod.embedder.weight.detach()[mod.sig_to_embedding_idx[name]][None, None, :]
            z[name] = z_BSF
        return z

    def mixer(self, mod, z_dict: dict) -> torch.Tensor:
        """MultiModalAttentionEmbedder.forward (wav2sleep.py:301-346) -> CLS features [B, S, F]."""
        signals = sorted(z_dict.keys())
        if len(signals) == 0:
            raise ValueError('No signals provided to MultiModalAttentionEmbedder.')
        first = z_dict[signals[0]]
        B, S, F_ = first.shape
        if F_ != mod.feature_dim:
            raise ValueError(f'Feature dimension {F_} does not match {mod.feature_dim=}.')
        dev = first.device
        R1 = mod.num_register_tokens + 1
        D, N = len(signals) + R1, B * S
        if D > 16:
            raise ValueError(f'{D} tokens per epoch: the generic attention kernel holds at most 16')
        tokens = torch.empty(N, D, F_, device=dev, dtype=torch.float32)
        tokens[:, :R1, :] = mod.register_tokens.detach()[0, 0].t()[None]          # [R1, F]
        pads = [torch.zeros(B, dtype=torch.bool, device=dev)] * R1
        for m, sig in enumerate(signals):   # plumbing on [B, S, F]: mask detection, zero fill, token slot copy
            z = z_dict[sig].float()
            m_B = torch.isinf(z).any(dim=2).any(dim=1)
            tokens[:, R1 + m, :] = torch.where(m_B[:, None, None], torch.zeros_like(z), z).reshape(N, F_)
            pads.append(m_B)
        keypad = torch.stack(pads, dim=1).to(torch.uint8)[:, None, :].expand(B, S, D).reshape(N, D).contiguous()
        H = mod.nhead
        hd = F_ // H
        X = tokens.view(N * D, F_)
        for layer in mod.transformer_encoder.layers:
            sa = layer.self_attn

            def attn(h):
                qkv = self.linear(h, sa.in_proj_weight, sa.in_proj_bias)
                ao = torch.empty(N * D, F_, device=dev, dtype=torch.float32)
                lib.attn_generic_fwd(qkv, keypad, ao, N, D, H, hd)
                return self._dropout_(self.linear(ao, sa.out_proj.weight, sa.out_proj.bias), layer.dropout1.p)

            def ff(h):
                a1 = self._dropout_(self.linear(h, layer.linear1.weight, layer.linear1.bias, mod.activation_name), layer.dropout.p)
                return self._dropout_(self.linear(a1, layer.linear2.weight, layer.linear2.bias), layer.dropout2.p)

            def ln(norm, t):
                o = torch.empty_like(t)
                lib.rownorm_fwd(t, F_, norm.weight.detach(), norm.bias.detach(), o, F_, N * D, F_, norm.eps, False, 0, 0.01)
                return o

            def add(a_, b_):
                o = torch.empty_like(a_)
                lib.eltwise(lib.ELT_ADD, a_, b_, o, a_.numel())
                return o

            if layer.norm_first:
                X = add(X, attn(ln(layer.norm1, X)))
                X = add(X, ff(ln(layer.norm2, X)))
            else:
                X = ln(layer.norm1, add(X, attn(X)))
                X = ln(layer.norm2, add(X, ff(X)))
        return X.view(N, D, F_)[:, 0, :].reshape(B, S, F_).contiguous()

    def sequence_cnn(self, mod, x_BSF):
        """SequenceCNN.forward (wav2sleep.py:379-390); channels-last [B, S, F] is already the layout the blocks run on."""
        x = x_BSF.contiguous().float()
        for block in mod.dilated_convs:
            x = self.dilated_block(block, x)
        return x

    def classifier(self, lin: nn.Linear, x_BSF):
        B, S, F_ = x_BSF.shape
        nc = lin.out_features
        logits = torch.empty(B, S, nc, device=x_BSF.device, dtype=torch.float32)
        lib.head_fwd(x_BSF.contiguous(), F_, lin.weight.detach(), lin.bias.detach(), logits, B * S, F_, nc, False)
        return logits

    def wav2sleep(self, model, x: dict) -> torch.Tensor:
        """Wav2Sleep.forward (wav2sleep.py:48-67)."""
        z = self.signal_encoders(model.signal_encoders, x)
        m = self.mixer(model.epoch_mixer, z)
        s = self.sequence_cnn(model.sequence_mixer, m)
        return self.classifier(model.classifier, s)

    def ppgnet(self, model, x_BT) -> torch.Tensor:
        """SleepPPGNet.forward (ppgnet.py:50-80): [B, 1 228 800] -> [B, 1200, n_classes]."""
        if x_BT.size(1) != model.INPUT_LENGTH:
            raise ValueError(f'Input tensor had unexpected shape: {x_BT.size()}')
        B = x_BT.shape[0]
        y = x_BT.contiguous().float().view(B, -1, 1)
        for block in model.conv_block.model:
            y = self.conv_block(block, y)                      # [B, 4800, 256]
        feat = y.reshape(B * 1200, 1024)                       # transpose(-1, -2).reshape(-1, 1200, 1024): memory order in channels-last
        z = self.linear(feat, model.dense.linear.weight, model.dense.linear.bias, model.dense.activation_name).view(B, 1200, model.feature_dim)
        for block in model.dilated_convs:
            z = self.dilated_block(block, z)
        return self.classifier(model.classifier, z)
