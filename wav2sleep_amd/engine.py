"""Launch plan of the wav2sleep hot path on MI355X: which C-ABI kernel runs when, on which buffers.

This is host logic only (no arithmetic): every FLOP of the forward/backward is in libw2s_hip.so.
Activation layout is channels-last [B, L, C]; every tensor between two encoder convs is stored PRE-norm /
PRE-activation exactly once and the consumer normalises + activates while staging it into LDS, so a
ConvBlock1D (models/blocks.py:57-71) costs 4 launches forward and reads/writes each tensor once.

Reference call sites mirrored here: Wav2Sleep.forward (models/wav2sleep.py:48-67), SignalEncoders.forward (:146-161),
SignalEncoder.forward (:235-267), MultiModalAttentionEmbedder.forward (:301-346), SequenceCNN.forward (:379-390),
DilatedConvBlock.forward (blocks.py:115-126), and autograd's backward of all of them.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import torch

from . import lib
from .settings import COLS_TO_SAMPLES_PER_EPOCH

# Settled scheduling choices (each was an environment switch while its A/B ran: docs/lab_notes_r1-3.md, r4): the encoder with the longest signal is
# enqueued first, the encoder streams are fed round-robin (one block per turn), the trunk's weight gradients are deferred to run beside the
# encoder backward.  _CLS_ONLY stays a module attribute because tests/test_r3_parity_gpu.py checks the all-rows form against it.
_CLS_ONLY = True      # last transformer layer: attention queries and the row-wise tail on the CLS rows only

FIRST_TILE = 1024  # positions per statistics partial of the Cin=1 layer


@dataclass
class EngineSpec:
    signal_map: dict
    feature_dim: int = 128
    num_classes: int = 4
    initial_channels: int = 16
    max_channels: int = 128
    mixer_layers: int = 2
    mixer_nhead: int = 8
    mixer_dim_ff: int = 512
    mixer_dropout: float = 0.0
    seq_blocks: int = 2
    seq_dilations: int = 6
    seq_kernel: int = 7
    seq_dropout: float = 0.0
    instance_eps: float = 1e-2
    layer_eps: float = 1e-5
    causal: bool = False       # encoders: causal-padded convolutions (blocks.py:150-152,178-182; `chunk_causal: False`)
    embed_signals: bool = False  # nn.Embedding row per signal added to its encoder output (wav2sleep.py:127-133,155-159)
    output_norm: bool = False   # nn.LayerNorm(feature_dim) on every encoder's output (wav2sleep.py:232-233,266)
    use_residual: bool = True   # ConvBlock1D(use_residual=False): no 1x1/stride-2 branch, no `downsample` parameter (blocks.py:49-55,67-68)
    register_tokens: int = 0    # R learnable tokens next to CLS (wav2sleep.py:299,330): D = R + 1 + C tokens per epoch
    chunk_causal: bool = False  # with causal: encode every 30-s epoch on its own ([B*S, 1, spe], wav2sleep.py:248-255), symmetric padding
    seq_causal: bool = False   # SequenceCNN: causal dilated convolutions (wav2sleep.py:355, blocks.py:150-152)
    enc_sig: dict = field(default_factory=dict)  # encoder name -> first signal that created it

    def __post_init__(self):
        for sig, enc in self.signal_map.items():
            if sig not in COLS_TO_SAMPLES_PER_EPOCH:
                raise ValueError(f"Column {sig} unrecognised. Doesn't have a sampling rate.")
            self.enc_sig.setdefault(enc, sig)
        if self.feature_dim != 128 or self.mixer_nhead * 16 != self.feature_dim:
            raise ValueError('kernels are built for feature_dim=128, head_dim=16 (scripts/config/model/wav2sleep.yaml)')
        if not 0 <= self.register_tokens <= 5:
            raise ValueError('register_tokens must be in 0..5 (the attention kernels hold up to 12 tokens per epoch: 6 signals + CLS + 5)')
        if self.mixer_dim_ff not in (384, 512) or self.seq_kernel != 7 or self.initial_channels != 16 or self.max_channels not in (16, 32, 64, 128):
            raise ValueError('unsupported hyper-parameters for the fused gfx950 kernels (dim_ff 384 / 512, kernel_size 7, channels 16 -> 128): '
                             'other configurations run on the generic inference path (wav2sleep_amd/generic.py)')

    def channels(self, enc: str) -> list[int]:
        """models/wav2sleep.py:198-201"""
        spe = COLS_TO_SAMPLES_PER_EPOCH[self.enc_sig[enc]]
        nb = int(math.log2(spe)) - 2
        return [min(self.initial_channels * 2 ** (i // 2), self.max_channels) for i in range(nb)]


def _cdiv(a, b):
    return (a + b - 1) // b


class Engine:
    """Owns nothing but scratch: parameters/gradients are views handed in by the caller (flat buffers)."""

    def __init__(self, spec: EngineSpec, params: dict[str, torch.Tensor], grads: dict[str, torch.Tensor] | None = None):
        self.spec = spec
        self.P = params
        self.G = grads
        self.PF: dict[str, torch.Tensor] = {}
        self.PB: dict[str, torch.Tensor] = {}
        self._pack_key = None
        self.step_seed = 0
        self._written: set[str] = set()
        self.ctx = None
        self.taps = None  # set to {} to collect per-stage tensors (tests / debugging)
        self.multi_stream = os.environ.get('W2S_MULTI_STREAM', '1') != '0'
        # split-precision (bf16x3) matrix-core path for the >=64-channel GEMM-shaped layers; W2S_EXACT_FP32=1 keeps fp32 MFMA
        self.split_precision = os.environ.get('W2S_EXACT_FP32', '0') != '1'
        self._bf = {}  # data_ptr of a weight operand -> (hi plane, lo plane)
        self._bfbuf = {}
        self._streams = {}
        self._rjobs = []
        self._deferred = None
        # structural choices that won their A/B (attributes, not environment switches: a test may flip one to check the other form)
        self.fused_forward = True   # <= 32-channel forward convs on the persistent kernel
        self.fold_gp = True         # conv3-backward statistics of the previous block ride in the residual-fold conv1 kernel
        self.fold_w1 = True         # block 0: conv1's weight gradient inside conv2's backward kernel, gn1 never stored
        # W2S_GRAD_FP16=1: fp16 storage (one power-of-two scale per tensor, fp32 arithmetic) of the gradient tensors between the fused-backward
        # launches of the <= 32-channel blocks.  Measured round 3, full suite green with it: -14 GB of traffic,
        # 33.3 -> 32.8 ms per step (-1.6 %); worst full-size gradient tensor 4.4e-4 -> 1.1e-3 relative L2 (EOG 8.3e-4 -> 1.25e-3; bar 2e-3).
        # OFF by default: 2.5 x the gradient error for 1.6 % is a trade a user should choose, not inherit
        self.grad_fp16 = os.environ.get('W2S_GRAD_FP16', '0') == '1'
        self.bwd_wide = True        # one-pass backward of the 64-channel convs (csrc/bwd_wide.hip)
        # SequenceCNN: dilated conv + channel LayerNorm (+ GELU) in ONE launch, forward and backward (csrc/seq_conv.hip; split precision only)
        self.seq_fused = os.environ.get('W2S_NO_SEQCONV', '0') != '1'
        # workgroup (= slab) caps of the weight-gradient launches with >= 64 x 128 channels: encoder convs (k = 3) / trunk linears (k = 1, or 4 strided taps)
        self.enc_wgrad_cap = int(os.environ.get('W2S_ENC_WGRAD_CAP', '128'))
        self.trunk_wgrad_cap = int(os.environ.get('W2S_TRUNK_WGRAD_CAP', '256'))   # (256 since the trunk's launches run on the pipelined kernel: lab notes r6)
        self.bwd_wide_rd = True     # 64-channel conv1: residual branch folded into the one-pass kernel
        self._cnt = {}   # measured neutral (its extra read ~ the pre-pass it saves): off
        self._cjobs = []
        if not spec.use_residual:
            # The kernels keep their residual inputs: a zero 1x1 weight that is no parameter (its gradient goes to a scratch buffer that
            # nothing reads, and the data gradient it contributes, Wd^T g, is exactly zero).
            for enc in dict.fromkeys(spec.signal_map.values()):
                cin = 1
                for i, c in enumerate(spec.channels(enc)):
                    name = f'signal_encoders.encoders.{enc}.cnn.{i}.downsample.weight'
                    some = next(iter(params.values()))
                    self.P[name] = torch.zeros(c, cin, 1, device=some.device, dtype=torch.float32)
                    if self.G is not None:
                        self.G[name] = torch.zeros(c, cin, 1, device=some.device, dtype=torch.float32)
                    cin = c
        # causal padding (scripts/config/main.yaml:22 `causal`): out[j] reads x[j*stride - (k-1-tap)*dil], zeros before the start.  Same
        # kernels, different pad: forward pad (k-1)*dil, data-gradient (flipped taps) pad 0; the fused / persistent kernels take the
        # left padding as a parameter (window origin t0*stride - pad; the stride-2 transposed forms swap the roles of the parities).
        self.chunk = bool(spec.causal and spec.chunk_causal)
        self.causal = bool(spec.causal) and not self.chunk   # = `_causal_conv_mode`, wav2sleep.py:204
        self.seq_causal = bool(spec.seq_causal)
        self.kpad = 2 if self.causal else 1
        self._fused_bwd_ok = self.split_precision or not self.causal   # the exact-fp32 fused backward is written for symmetric padding
        lib.load()

    # ------------------------------------------------------------------ weights
    def _pack_list(self):
        sp = self.spec
        out = []  # (name, cout, cin, taps, need_fwd, need_bwd)
        for enc in dict.fromkeys(sp.signal_map.values()):
            ch = sp.channels(enc)
            cin = 1
            for i, c in enumerate(ch):
                p = f'signal_encoders.encoders.{enc}.cnn.{i}.'
                if i > 0:
                    out.append((p + 'conv1.conv.weight', c, cin, 3, True, True))
                    out.append((p + 'downsample.weight', c, cin, 1, False, True))
                out.append((p + 'conv2.conv.weight', c, c, 3, True, True))
                out.append((p + 'conv3.conv.weight', c, c, 3, True, True))
                cin = c
            out.append((f'signal_encoders.encoders.{enc}.linear.weight', sp.feature_dim, 4 * ch[-1], 1, False, True))
        F = sp.feature_dim
        for l in range(sp.mixer_layers):
            p = f'epoch_mixer.transformer_encoder.layers.{l}.'
            out.append((p + 'self_attn.in_proj_weight', 3 * F, F, 1, False, True))
            out.append((p + 'self_attn.out_proj.weight', F, F, 1, False, True))
            out.append((p + 'linear1.weight', sp.mixer_dim_ff, F, 1, False, True))
            out.append((p + 'linear2.weight', F, sp.mixer_dim_ff, 1, False, True))
        for b in range(sp.seq_blocks):
            for j in range(sp.seq_dilations):
                out.append((f'sequence_mixer.dilated_convs.{b}.conv_layers.{j}.conv.weight', F, F, sp.seq_kernel, True, True))
        return out

    def pack(self, need_bwd: bool = True):
        """torch-layout weights -> kernel layouts ([cout][taps][cin] forward, [cin][taps][cout] data-gradient)."""
        jobs = []
        for name, cout, cin, taps, nf, nb in self._pack_list():
            if name not in self.P:  # stand-alone sub-module engines hold only their own parameters
                continue
            w = self.P[name]
            if nf and name not in self.PF:
                self.PF[name] = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
            if nb and need_bwd and name not in self.PB:
                self.PB[name] = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
            f = self.PF.get(name) if nf else None
            bw = self.PB.get(name) if (nb and need_bwd) else None
            fh = fl = bh = bl = None
            if self.split_precision:
                # forward operand (packed copy, or the torch tensor itself when taps == 1) and data-gradient operand
                fop = f if f is not None else w
                want_f = (cin >= 32 and cout >= 32) or (cin == 16 and taps in (1, 3))   # 16 channels: two taps per K step
                want_b = bw is not None and cout >= 32 and cin >= 32
                planes = self._bfbuf.setdefault(name, {})
                for kind, want in (('f', want_f), ('b', want_b)):
                    if want and kind not in planes:
                        n = cout * 32 * ((taps + 1) // 2) if (kind == 'f' and cin == 16) else w.numel()   # padded K, zero tail
                        planes[kind] = (torch.zeros(n, device=w.device, dtype=torch.bfloat16),
                                        torch.zeros(n, device=w.device, dtype=torch.bfloat16))
                fh, fl = planes['f'] if want_f else (None, None)
                bh, bl = planes['b'] if want_b else (None, None)
                if want_f:
                    self._bf[fop.data_ptr()] = (fh, fl)
                if want_b:
                    self._bf[bw.data_ptr()] = (bh, bl)
            if f is not None or bw is not None or fh is not None or bh is not None:
                jobs.append((w, f, bw, fh, fl, bh, bl, cout, cin, taps))
        lib.repack_batch(jobs)  # every layer of the model in ceil(n / 48) launches

    def ensure_packed(self, key, need_bwd: bool):
        key = (key, need_bwd or (self._pack_key is not None and self._pack_key[1]))
        if self._pack_key != key:
            self.pack(need_bwd=key[1])
            self._pack_key = key

    # ------------------------------------------------------------------ helpers
    def _side_stream(self, enc: str, dev):
        if not self.multi_stream:
            return torch.cuda.current_stream(dev)
        key = (enc, dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=dev)
        return self._streams[key]

    def _finalize(self, part, B, ntiles, C, count, kind):
        out = torch.empty(B, C, 2, device=part.device, dtype=torch.float32)
        lib.stats_finalize(part, B, ntiles, C, count, self.spec.instance_eps, kind, out)
        return out

    def _conv(self, **kw):
        planes = self._bf.get(kw['w'].data_ptr()) if self.split_precision else None
        if planes is not None:
            kw['w_hi'], kw['w_lo'] = planes
        lib.conv_forward(lib.conv_args(**kw))

    def _conv_part(self, *, B, L_out, cout, kind, **kw):
        """A conv launch that also produces instance-norm statistics (per-tile partials, sized by the tile of the kernel that takes THIS
        descriptor) -> finalised statistics [B][cout][2] (kind 0: mean/rstd, 1: backward sums)."""
        planes = self._bf.get(kw['w'].data_ptr()) if self.split_precision else None
        if planes is not None:
            kw['w_hi'], kw['w_lo'] = planes
        dev = kw['x'].device
        a = lib.conv_args(B=B, L_out=L_out, cout=cout, part=None, **kw)
        nt = _cdiv(L_out, lib.conv_tile_of(a))
        part = torch.empty(B, nt, 2, cout, device=dev, dtype=torch.float32)
        lib.set_part(a, part)
        lib.conv_forward(a)
        return self._finalize(part, B, nt, cout, L_out, kind)

    def _conv_stats(self, *, x, w, B, L_in, L_out, cin, cout, stride, pro, pro_stats=None, x2=None):
        """k=3 encoder conv writing the pre-norm tensor + instance-norm statistics (blocks.py:174-183)."""
        dev = x.device
        y = torch.empty(B, L_out, cout, device=dev, dtype=torch.float32)
        ftile = lib.conv_fwd_fused_tile(cin, cout, stride) if (self.split_precision and self.fused_forward) else 0
        if ftile and pro in (lib.PRO_GELU, lib.PRO_IN_GELU, lib.PRO_FIRST):
            # <= 32 channels: persistent split-precision forward kernel (prefetch + LDS-resident weights)
            nt = _cdiv(L_out, ftile)
            # persistent workgroups (grid sweep of round 5, `gpurun_out/r5p`): first-layer form 1024, 16 -> 16 stride 2 768, everything else 512
            nwg = 1024 if pro == lib.PRO_FIRST else 768 if (cin == 16 and stride == 2) else 512
            part = torch.empty(B, nt, 2, cout, device=dev, dtype=torch.float32)
            lib.conv_fwd_fused(x=x, w=w, st_in=pro_stats, w1=x2, y=y, part=part, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, stride=stride,
                               pro=pro, pad=self.kpad, nwg=nwg)
            return y, self._finalize(part, B, nt, cout, L_out, 0)
        return y, self._conv_part(x=x, x2=x2, w=w, y=y, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, taps=3, stride=stride, pad=self.kpad, pro=pro,
                                  pro_stats=pro_stats, epi=lib.EPI_STATS, kind=0, **({'ldx': 4} if pro == lib.PRO_FIRST else {}))

    def _linear(self, x, w, bias, rows, cin, cout, ldx=None, y=None, ldy=None, **fuse):
        """y[rows, cout] = x[rows, cin(*k)] @ w^T + bias; cin > 128 runs as k = cin/128 strided taps.
        fuse: epilogue fusions of the transformer layer (lib.conv_args: fuse=FUSE_*, aux, y2, drop_p, drop_seed)."""
        if y is None:
            y = torch.empty(rows, cout, device=x.device, dtype=torch.float32)
        if cin <= 128:
            self._conv(x=x, w=w, y=y, B=1, L_in=rows, L_out=rows, cin=cin, cout=cout, taps=1, stride=1, pad=0, ldx=ldx, ldy=ldy,
                       epi=lib.EPI_BIAS, bias=bias, **fuse)
        else:
            k = cin // 128
            assert cin % 128 == 0 and k in (3, 4) and ldx is None
            self._conv(x=x, w=w, y=y, B=1, L_in=rows * k, L_out=rows, cin=128, cout=cout, taps=k, stride=k, pad=0, mode=lib.MODE_DILATED,
                       ldy=ldy, epi=lib.EPI_BIAS, bias=bias, **fuse)
        return y

    def _slab(self, dev, nslab, n):
        return torch.empty(nslab * n, device=dev, dtype=torch.float32)

    def _wgrad(self, name, *, g, x, B, L_in, L_out, cin, cout, taps, stride, pad, dil=1, layout=0, **kw):
        """weight gradient -> slabs -> deterministic reduce into G[name] (accumulating if already written).
        During the trunk's backward the call may be DEFERRED (queued as a closure over g / x and run after the encoder streams have forked):
        the caller must not write g or x in place afterwards -- every trunk gradient tensor is a fresh allocation written exactly once."""
        if self._deferred is not None:   # trunk backward: leaf work, enqueued after the encoder streams have been forked (backward())
            self._deferred.append(lambda: self._wgrad(name, g=g, x=x, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, taps=taps, stride=stride,
                                                      pad=pad, dil=dil, layout=layout, **kw))
            return
        gy = lib.wgrad_grid_y(cin, cout, taps, dil)
        work = _cdiv(B * L_out, 256)
        args = dict(g=g, x=x, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, taps=taps, stride=stride, pad=pad, dil=dil,
                    split_precision=self.split_precision, **kw)
        gx = max(1, min(work, max(1, lib.wgrad_max_blocks(slab=None, nslab=0, **args) // gy)))
        if cin >= 64 and cout >= 128 and L_out >= 1024:
            # 128-channel encoder convs: a slab is 100-200 KB, written once and read once by the reduce -- at 256 workgroups that is 40-80 % on top
            # of the kernel's own input.  128 workgroups halve it; the other streams use the CUs left over (docs/lab_notes_r5.md section 11).
            # The same predicate matches the trunk's 76 800-row linears (in_proj, out_proj, linear1, linear2: slabs of 64-256 KB); their cap is
            # a separate attribute so that the two can be measured apart (docs/lab_notes_r6.md)
            gx = min(gx, self.enc_wgrad_cap if taps == 3 else self.trunk_wgrad_cap)
        nslab = gx * lib.wgrad_slabs_per_block_of(slab=None, nslab=0, **args)
        slab = self._slab(g.device, nslab, cout * cin * taps)
        lib.wgrad(slab=slab, nslab=nslab, **args)
        self._rjobs.append((slab, nslab, self.G[name], cout, cin, taps, dil, name in self._written, layout))
        self._written.add(name)

    def _bwd_fused(self, name, *, g, y, st_k, bst_k, pro, xin, st_in, add_even, gout, want_part, B, Lg, Lh, cg, ch, stride,
                   gpre=None, down=None, w1=None, y3p=None, st3p=None, gmode=0, hdr_g=None, hdr_p=None, hdr_o=None, part_w1=None, x0=None, down0=None):
        """dgrad + wgrad of one k=3 encoder conv in one pass (<= 32 channels); returns the backward statistics or None.
        gpre / down (conv1 of a residual block): fold the 1x1/stride-2 residual branch `down` in as well."""
        dev = g.device
        tile = lib.bwd_fused_tile(cg, ch, stride, gpre is not None, self.split_precision)
        nt = _cdiv(Lh, tile)
        # persistent workgroups = what fits a CU x 256 (round 5: the stride-2 forms and the 16-channel residual-fold form fit three)
        nslab = max(1, min(B * nt, 768 if ((cg == 16 and ch == 16 and gpre is not None) or (stride == 2 and cg == ch)) else 512))
        slab = self._slab(dev, nslab, cg * ch * 3)
        slab_d = self._slab(dev, nslab, cg * ch) if gpre is not None else None
        part = torch.empty(B, nt, 2, ch, device=dev, dtype=torch.float32) if want_part else None
        part_wd = torch.empty(nslab, 16, device=dev, dtype=torch.float32) if x0 is not None else None   # block 0's downsample weight gradient (down0)
        lib.bwd_fused(g=g, y=y, st_k=st_k, bst_k=bst_k, pro=pro, xin=xin, st_in=st_in, add_even=add_even, wb=self.PB[name], gout=gout,
                      part=part, slab=slab, nslab=nslab, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch, stride=stride, pad=self.kpad, split_precision=self.split_precision,
                      gpre=gpre, wd=self.PB[down] if gpre is not None else None, slab_d=slab_d, w1=w1, y3p=y3p, st3p=st3p,
                      gmode=gmode, hdr_g=hdr_g, hdr_p=hdr_p, hdr_o=hdr_o, part_w1=part_w1, x0=x0, part_wd=part_wd)
        self._rjobs.append((slab, nslab, self.G[name], cg, ch, 3, 1, name in self._written, 0))
        self._written.add(name)
        if gpre is not None:
            self._rjobs.append((slab_d, nslab, self.G[down], cg, ch, 1, 1, down in self._written, 0))
            self._written.add(down)
        if part_wd is not None:
            self._colsum(part_wd, nslab, 16, self.G[down0], accumulate=down0 in self._written)
            self._written.add(down0)
        if not want_part:
            return None
        return self._bstats(part, B, nt, ch, Lh)

    def _bwd_wide_ok(self, B, L, cg, ch, stride=1, hst=True):
        """the one-pass backward of a 64-channel conv (csrc/bwd_wide.hip): split precision, symmetric or causal padding; L = input-side length"""
        return self.bwd_wide and self.split_precision and lib.bwd_wide_takes(B, L, cg, ch, stride, hst)

    def _bwd_wide(self, name, *, g, y, st_k, bst_k, xin, st_in, add_even, gout, want_part, B, L, cg, ch, stride=1, y3p=None, st3p=None,
                  gpre=None, down=None):
        """dgrad + wgrad + GELU' + backward statistics of one 64-channel k=3 conv in one pass; returns the statistics or None."""
        dev = g.device
        tile, groups = lib.bwd_wide_tile(cg, ch, stride), lib.bwd_wide_groups(cg, ch, stride)
        nt = _cdiv(L, tile)
        nslab = max(1, min(B * nt, 256))   # one workgroup per CU (LDS)
        slab = self._slab(dev, nslab, cg * ch * 3)
        part = torch.empty(B, nt * groups, 2, ch, device=dev, dtype=torch.float32) if want_part else None
        wh, wl = self._bf[self.PB[name].data_ptr()]
        dh, dl = self._bf[self.PB[down].data_ptr()] if gpre is not None else (None, None)
        slab_d = self._slab(dev, nslab, cg * ch) if gpre is not None else None
        lib.bwd_wide(g=g, y=y, st_k=st_k, bst_k=bst_k, xin=xin, st_in=st_in, add_even=add_even, w_hi=wh, w_lo=wl, gout=gout, part=part, slab=slab,
                     nslab=nslab, B=B, L=L, cg=cg, ch=ch, stride=stride, y3p=y3p, st3p=st3p, gpre=gpre, wd_hi=dh, wd_lo=dl, slab_d=slab_d, pad=self.kpad)
        self._rjobs.append((slab, nslab, self.G[name], cg, ch, 3, 1, name in self._written, 0))
        self._written.add(name)
        if gpre is not None:
            self._rjobs.append((slab_d, nslab, self.G[down], cg, ch, 1, 1, down in self._written, 0))
            self._written.add(down)
        return self._bstats(part, B, nt * groups, ch, L) if want_part else None

    def _colsum(self, part, nparts, C, out, accumulate=False, ld=None):
        """queued column sum (flushed with the slab reductions): out[c] (+)= sum_p part[p*ld + c]"""
        self._cjobs.append((part, nparts, C, out, accumulate, ld))

    def _flush_reduce(self):
        """Deterministic slab sums of every weight gradient queued since the last flush, in one launch per 48 layers
        (on the current stream; a weight appears at most once per flush)."""
        jobs, self._rjobs = self._rjobs, []
        lib.wgrad_reduce_batch(jobs)
        cjobs, self._cjobs = self._cjobs, []
        lib.colsum_batch(cjobs)

    def _interleave(self, tasks, trunk=None):
        """tasks: encoder -> (stream, [generators]).  Enqueue the encoders round-robin, one block per turn, each on its own stream: the
        streams then start together and progress together, instead of the first encoder's whole pass being enqueued (and mostly executed)
        before the second one's first kernel.  Signals that share an encoder share its stream and run one after the other.  The
        queues of pending slab / column reductions are per encoder (they are flushed on that encoder's stream)."""
        live = {e: [st, list(gens), [], []] for e, (st, gens) in tasks.items()}
        if trunk is not None:   # (stream, generator): the trunk's deferred leaf work; it inherits the reductions queued so far
            live['_trunk'] = [trunk[0], [trunk[1]], self._rjobs, self._cjobs]
            self._rjobs, self._cjobs = [], []
        saved = (self._rjobs, self._cjobs)
        while live:
            for e in list(live):
                st, gens, rj, cj = live[e]
                self._rjobs, self._cjobs = rj, cj
                with torch.cuda.stream(st):
                    try:
                        next(gens[0])
                    except StopIteration:
                        gens.pop(0)
                live[e][2], live[e][3] = self._rjobs, self._cjobs   # (_flush_reduce replaces the lists)
                if not gens:
                    if self._rjobs or self._cjobs:
                        raise RuntimeError('encoder pass ended with reductions still queued')
                    del live[e]
        self._rjobs, self._cjobs = saved

    def _colgrad(self, name, g, rows, C, ldg=None):
        """G[name][c] = sum_rows g[row, c]  (bias / CLS gradients)."""
        if self._deferred is not None:
            self._deferred.append(lambda: self._colgrad(name, g, rows, C, ldg))
            return
        nparts = max(1, min(1024, _cdiv(rows, 64)))
        part = torch.empty(nparts, C, device=g.device, dtype=torch.float32)
        lib.bias_grad(g, rows, C, C if ldg is None else ldg, part, nparts)
        self._colsum(part, nparts, C, self.G[name], accumulate=name in self._written)
        self._written.add(name)

    def _seed(self, site: int) -> int:
        return ((self.step_seed & 0xFFFFFFFF) << 16) ^ (site * 0x9E3779B1 & 0xFFFFFFFF)

    # ------------------------------------------------------------------ encoder forward
    def _encoder_forward(self, sig, x, keep, tok_slice, ldtok, save):
        """Generator: yields after every block so that the caller can enqueue several encoders round-robin (`_interleave`); the
        context dict (or None) is its return value."""
        sp, P, PF = self.spec, self.P, self.PF
        enc = sp.signal_map[sig]
        ch = sp.channels(enc)
        spe = COLS_TO_SAMPLES_PER_EPOCH[sig]
        B, T = x.shape
        if T % spe:
            raise ValueError(f'Input length {T} must be divisible by samples_per_epoch={spe}.')
        S = T // spe
        dev = x.device
        Bfull = B
        if self.chunk:   # every epoch is its own sample for the conv stack; the [B*S, 4, C] result IS the [B, 4S, C] map the dense layer reads
            if B * S > 65535:
                raise ValueError(f'chunk-causal encoding launches one grid row per epoch: batch*epochs = {B * S} exceeds 65535')
            x = x.contiguous().view(B * S, spe)
            B, T = B * S, spe
        pfx = f'signal_encoders.encoders.{enc}.'
        blocks = []
        # ---- block 0 (Cin = 1)
        c, L = ch[0], T
        # The Cin = 1 conv1 output (16 x T per recording: the largest tensor of the model) is only materialised when someone
        # needs it (debug taps, the exact-fp32 backward kernels); otherwise its consumers recompute it from the raw signal
        # (W2S_PRO_FIRST: 3 FMAs per element) and only its instance-norm statistics are computed here.
        recompute = self.split_precision and self.taps is None and c == 16
        w1 = P[pfx + 'cnn.0.conv1.conv.weight']
        y1 = None if recompute else torch.empty(B, L, c, device=dev, dtype=torch.float32)
        ft = FIRST_TILE
        nt = _cdiv(L, ft)
        part = torch.empty(B, nt, 2, c, device=dev, dtype=torch.float32)
        xmom = torch.empty(B, nt, 9, device=dev, dtype=torch.float32) if (recompute and save and self.fold_w1) else None
        if xmom is not None:   # (the signal's raw moments per tile: what the folded first-layer weight gradient needs in backward)
            lib.enc_first_stats(x, w1, part, xmom, B, L, ft, causal=self.causal)
        else:
            lib.enc_first_fwd(x, w1, y1, part, B, L, c, ft, causal=self.causal)
        st1 = self._finalize(part, B, nt, c, L, 0)
        if recompute:
            y2, st2 = self._conv_stats(x=x, x2=w1, w=PF[pfx + 'cnn.0.conv2.conv.weight'], B=B, L_in=L, L_out=L, cin=c, cout=c, stride=1,
                                       pro=lib.PRO_FIRST, pro_stats=st1)
        else:
            y2, st2 = self._conv_stats(x=y1, w=PF[pfx + 'cnn.0.conv2.conv.weight'], B=B, L_in=L, L_out=L, cin=c, cout=c, stride=1,
                                       pro=lib.PRO_IN_GELU, pro_stats=st1)
        y3, st3 = self._conv_stats(x=y2, w=PF[pfx + 'cnn.0.conv3.conv.weight'], B=B, L_in=L, L_out=L // 2, cin=c, cout=c, stride=2,
                                   pro=lib.PRO_IN_GELU, pro_stats=st2)
        pre = torch.empty(B, L // 2, c, device=dev, dtype=torch.float32)
        lib.enc_first_join(x, P[pfx + 'cnn.0.downsample.weight'], y3, st3, pre, B, L, c)
        if self.taps is not None:
            self.taps.update({f'{sig}.y1.0': y1, f'{sig}.st1.0': st1, f'{sig}.y2.0': y2, f'{sig}.y3.0': y3, f'{sig}.pre.0': pre})
        if save:
            blocks.append(dict(y1=y1, st1=st1, y2=y2, st2=st2, y3=y3, st3=st3, pin=None, L=L, cin=1, c=c))
        pin, cin, L = pre, c, L // 2
        yield
        # ---- blocks 1..
        for i in range(1, len(ch)):
            c = ch[i]
            p = f'{pfx}cnn.{i}.'
            y1, st1 = self._conv_stats(x=pin, w=PF[p + 'conv1.conv.weight'], B=B, L_in=L, L_out=L, cin=cin, cout=c, stride=1, pro=lib.PRO_GELU)
            y2, st2 = self._conv_stats(x=y1, w=PF[p + 'conv2.conv.weight'], B=B, L_in=L, L_out=L, cin=c, cout=c, stride=1,
                                       pro=lib.PRO_IN_GELU, pro_stats=st1)
            y3, st3 = self._conv_stats(x=y2, w=PF[p + 'conv3.conv.weight'], B=B, L_in=L, L_out=L // 2, cin=c, cout=c, stride=2,
                                       pro=lib.PRO_IN_GELU, pro_stats=st2)
            pre = torch.empty(B, L // 2, c, device=dev, dtype=torch.float32)
            self._conv(x=pin, w=P[p + 'downsample.weight'], y=pre, B=B, L_in=L, L_out=L // 2, cin=cin, cout=c, taps=1, stride=2, pad=0,
                       pro=lib.PRO_GELU, epi=lib.EPI_AUX_INGELU_ADD, aux=y3, aux_stats=st3)
            if save:
                blocks.append(dict(y1=y1, st1=st1, y2=y2, st2=st2, y3=y3, st3=st3, pin=pin, L=L, cin=cin, c=c))
            if self.taps is not None:
                self.taps[f'{sig}.pre.{i}'] = pre
            pin, cin, L = pre, c, L // 2
            yield
        # ---- time-distributed dense + GELU (wav2sleep.py:261-265): taps=4/stride=4 over the [B,4S,C] map
        F = sp.feature_dim
        zpre = torch.empty(Bfull, S, F, device=dev, dtype=torch.float32)
        zact = rs_out = None
        if sp.output_norm:   # GELU output to its own tensor, LayerNorm from there into the token slot (masked samples: LN(0) = beta,
            zact = torch.empty(Bfull, S, F, device=dev, dtype=torch.float32)   # never read by another token: their keys are padded out)
            rs_out = torch.empty(Bfull * S, 2, device=dev, dtype=torch.float32)
        self._conv(x=pin, w=P[pfx + 'linear.weight'], y=zpre, B=Bfull, L_in=4 * S, L_out=S, cin=cin, cout=F, taps=4, stride=4, pad=0,
                   mode=lib.MODE_DILATED, pro=lib.PRO_GELU, epi=lib.EPI_BIAS, bias=P[pfx + 'linear.bias'], rowkeep=keep,
                   y2=zact if sp.output_norm else tok_slice, ldy2=F if sp.output_norm else ldtok)
        if sp.output_norm:
            lib.layernorm_fwd(zact, F, P[pfx + 'output_norm.weight'], P[pfx + 'output_norm.bias'], tok_slice, ldtok, rs_out, Bfull * S, F, sp.layer_eps)
        if self.taps is not None:
            self.taps[f'{sig}.zpre'] = zpre
        return dict(sig=sig, enc=enc, x=x, xmom=xmom, keep=keep, blocks=blocks, plast=pin, zpre=zpre, zact=zact, rs_out=rs_out, S=S, B=Bfull, Bc=B) if save else None

    # ------------------------------------------------------------------ full forward
    def _validate(self, x: dict[str, torch.Tensor]):
        sp = self.spec
        if len(x) == 0:
            raise ValueError('No signals provided to MultiModalAttentionEmbedder.')
        for s in x:
            if s not in sp.signal_map:
                raise ValueError(f'Unknown signal {s}')
        for s, v in x.items():  # wav2sleep.py:243-244, before anything is launched
            spe = COLS_TO_SAMPLES_PER_EPOCH[s]
            if v.dim() != 2 or v.shape[1] == 0 or v.shape[1] % spe:
                raise ValueError(f'Input length {tuple(v.shape)} of {s} must be [B, T] with T divisible by samples_per_epoch={spe}.')
        if len({(v.shape[0], v.shape[1] // COLS_TO_SAMPLES_PER_EPOCH[s]) for s, v in x.items()}) != 1:
            raise ValueError('all signals must share batch size and number of epochs')
        if len(x) + sp.register_tokens + 1 > 12:
            raise ValueError(f'{len(x)} signals + {sp.register_tokens + 1} CLS/register tokens: the attention kernels hold at most 12 tokens per epoch')

    def _encode_begin(self, x: dict[str, torch.Tensor], save: bool, cls: bool = True):
        """Token tensor (CLS / register rows written on the current stream) and the encoder passes as generators, one list per encoder
        stream; nothing of the encoders is enqueued yet.  Returns (state, tasks) for `_interleave`."""
        sp, P = self.spec, self.P
        sigs = sorted(x.keys())  # wav2sleep.py:311
        first = x[sigs[0]]
        dev = first.device
        B = first.shape[0]
        S = first.shape[1] // COLS_TO_SAMPLES_PER_EPOCH[sigs[0]]
        R1 = sp.register_tokens + 1   # CLS + register tokens occupy token slots 0..R
        F, D, N = sp.feature_dim, len(sigs) + R1, B * S

        tokens = torch.empty(N, D, F, device=dev, dtype=torch.float32)
        if cls:
            for r in range(R1):   # column r of the [1, 1, F, R+1] parameter
                lib.add_rows(tokens.view(-1)[r * F:], D * F, P['epoch_mixer.register_tokens'].view(-1)[r:], R1, None, 1, N, F, False)
        enc_ctx = [None] * len(sigs)
        xf = []
        for s in sigs:
            xs = x[s]
            if xs.dtype != torch.float32 or not xs.is_contiguous():
                xs = xs.float().contiguous()
            xf.append(xs)
        # which sample has which modality (a `-inf` row = missing, wav2sleep.py:150): the per-signal keep masks of the encoder epilogues and
        # the key-padding mask of the set-fusion transformer in ONE launch on this stream, before the encoder streams fork
        keep_all = torch.empty(len(sigs), B, device=dev, dtype=torch.float32)
        keypad = torch.empty(N, D, device=dev, dtype=torch.uint8)
        lib.token_masks(xf, R1, B, S, keep_all, keypad)
        keeps = [keep_all[m] for m in range(len(sigs))]
        # launch order: longest encoder first (the streams run side by side; the 1024-samples-per-epoch encoders take 4x the time of the
        # 256 ones and set the end of this phase) -- token slot m stays the sorted position
        tasks = {}   # encoder -> (stream, [generators]): signals sharing an encoder run one after the other on its stream
        for m, s in sorted(enumerate(sigs), key=lambda ms: -COLS_TO_SAMPLES_PER_EPOCH[ms[1]]):
            xs = xf[m]
            st = self._side_stream(sp.signal_map[s], dev)

            def run(m=m, s=s, xs=xs):
                keep = keeps[m]
                slot = tokens.view(-1)[(R1 + m) * F:]
                enc_ctx[m] = yield from self._encoder_forward(s, xs, keep, slot, D * F, save)
                if sp.embed_signals:   # + embedding row of this signal on the samples that have it (wav2sleep.py:155-159)
                    lib.add_rows(slot, D * F, P['signal_encoders.embedder.weight'][sorted(sp.signal_map).index(s)], 1, keep, S, N, F, True)
            tasks.setdefault(sp.signal_map[s], (st, []))[1].append(run())
        return dict(tokens=tokens, keeps=keeps, keypad=keypad, enc=enc_ctx, sigs=sigs, B=B, S=S, D=D, N=N, R1=R1), tasks

    def _encode_keypad(self, e):
        """(the key-padding mask is made with the keep masks in `_encode_begin`: w2s_token_masks)"""
        return e

    def encode(self, x: dict[str, torch.Tensor], save: bool = False, pack_key=None, cls: bool = True):
        """SignalEncoders.forward (models/wav2sleep.py:146-161) + token assembly (:319-330): returns the set-fusion input
        tokens [N, D, F] (token 0 = CLS, tokens 1.. = modalities in sorted order, zero rows for missing ones), the per-signal
        keep masks [B] and, if `save`, what backward needs."""
        self._validate(x)
        self.ensure_packed(pack_key, need_bwd=save)
        e, tasks = self._encode_begin(x, save, cls)
        # The encoders are independent until the set-fusion transformer: each runs on its own HIP stream, so the
        # matrix-core-bound 64/128-channel layers of one modality overlap the bandwidth-bound 16/32-channel layers of
        # another and grid tails are filled (signals sharing an encoder share a stream: their weight gradients accumulate).
        main = torch.cuda.current_stream(e['tokens'].device)
        for st, _ in tasks.values():
            st.wait_stream(main)
        self._encode_keypad(e)
        self._interleave(tasks)
        for st, _ in tasks.values():
            main.wait_stream(st)
        return e

    def mix(self, tokens: torch.Tensor, keypad: torch.Tensor, pm: float = 0.0, save: bool = False):
        """MultiModalAttentionEmbedder's transformer (models/wav2sleep.py:341-345) on tokens [N, D, F]; returns the final
        token tensor [N*D, F] (row n*D is the CLS output) and the per-layer tensors backward needs."""
        sp, P = self.spec, self.P
        N, D, F = tokens.shape
        dev = tokens.device
        # ---- set-fusion transformer (nn.TransformerEncoderLayer norm_first, wav2sleep.py:286-296)
        R = N * D
        X = tokens.view(R, F)
        layers = []
        for l in range(sp.mixer_layers):
            p = f'epoch_mixer.transformer_encoder.layers.{l}.'
            h = torch.empty(R, F, device=dev, dtype=torch.float32)
            rs1 = torch.empty(R, 2, device=dev, dtype=torch.float32)
            lib.layernorm_fwd(X, F, P[p + 'norm1.weight'], P[p + 'norm1.bias'], h, F, rs1, R, F, sp.layer_eps)
            qkv = self._linear(h, P[p + 'self_attn.in_proj_weight'], P[p + 'self_attn.in_proj_bias'], R, F, 3 * F)
            # Only token 0 of the LAST layer's output is read (wav2sleep.py:345 returns the CLS token): behind that layer's attention, every
            # row-wise op -- out_proj, the residual adds, norm2, the feed-forward block -- runs on the N CLS rows instead of all N x D token
            # rows (row stride D*F in, compact [N][F] out); the other tokens' outputs were never used and their gradients are exactly zero.
            # Round 5: so is the attention itself -- token 0 is the only QUERY of that layer (keys and values still come from every token)
            cls = _CLS_ONLY and l == sp.mixer_layers - 1 and D > 1
            ao = torch.empty(R, F, device=dev, dtype=torch.float32)
            lib.attn_fwd(qkv, keypad, ao, N, D, sp.mixer_nhead, pm, self._seed(10 * l + 1), nq=1 if cls else D)
            Rr, ldr = (N, D * F) if cls else (R, F)
            # x + Dropout(out_proj(attention)): the residual add and the dropout ride in the projection's epilogue (no `proj` tensor)
            X1 = self._linear(ao, P[p + 'self_attn.out_proj.weight'], P[p + 'self_attn.out_proj.bias'], Rr, F, F, ldx=ldr, fuse=lib.FUSE_ADD_DROP, aux=X,
                              ld_aux=ldr, drop_p=pm, drop_seed=self._seed(10 * l + 2))
            h2 = torch.empty(Rr, F, device=dev, dtype=torch.float32)
            rs2 = torch.empty(Rr, 2, device=dev, dtype=torch.float32)
            lib.layernorm_fwd(X1, F, P[p + 'norm2.weight'], P[p + 'norm2.bias'], h2, F, rs2, Rr, F, sp.layer_eps)
            FF = sp.mixer_dim_ff
            a1 = torch.empty(Rr, FF, device=dev, dtype=torch.float32)   # Dropout(GELU(linear1)): second output of linear1's epilogue
            f1 = self._linear(h2, P[p + 'linear1.weight'], P[p + 'linear1.bias'], Rr, F, FF, fuse=lib.FUSE_Y2_GELU_DROP, y2=a1, ldy2=FF, drop_p=pm,
                              drop_seed=self._seed(10 * l + 3))
            X2 = self._linear(a1, P[p + 'linear2.weight'], P[p + 'linear2.bias'], Rr, FF, F, fuse=lib.FUSE_ADD_DROP, aux=X1, drop_p=pm,
                              drop_seed=self._seed(10 * l + 4))
            if save:
                layers.append(dict(X=X, rs1=rs1, h=h, qkv=qkv, ao=ao, X1=X1, rs2=rs2, h2=h2, f1=f1, a1=a1, cls=cls))
            X = X2

        return X, layers   # [N*D, F] (row n*D = CLS), or the CLS rows alone [N, F] when the last layer ran on those only

    def seq(self, xin: torch.Tensor, ldin: int, B: int, S: int, ps: float = 0.0, save: bool = False):
        """SequenceCNN.forward (models/wav2sleep.py:379-390) on rows [B*S] of `xin` (row stride ldin); returns the last block's
        PRE-activation output [B, S, F] (the module output is GELU of it) and the saved tensors."""
        sp, P = self.spec, self.P
        F = sp.feature_dim
        N, D = B * S, ldin // F
        dev = xin.device
        X = xin
        xin, ldin = X, D * F
        seq = []
        pre_out = None
        for b in range(sp.seq_blocks):
            hcur, ldh = xin, ldin
            convs = []
            for j in range(sp.seq_dilations):
                d = 2 ** j
                p = f'sequence_mixer.dilated_convs.{b}.conv_layers.{j}.'
                y = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                hn = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                rs = torch.empty(B * S, 2, device=dev, dtype=torch.float32)
                pad = (sp.seq_kernel - 1) * d if self.seq_causal else (sp.seq_kernel // 2) * d
                if self._seq_fused_ok(d):   # conv + LayerNorm + GELU in one launch (a tile holds all 128 channels of its positions)
                    wh, wl = self._bf[self.PF[p + 'conv.weight'].data_ptr()]
                    lib.seq_conv(x=hcur, w_hi=wh, w_lo=wl, B=B, S=S, ldx=ldh, dil=d, pad=pad, mode=1, y=y, out=hn, rs=rs, gamma=P[p + 'norm.weight'],
                                 beta=P[p + 'norm.bias'], eps=sp.layer_eps)
                else:
                    self._conv(x=hcur, w=self.PF[p + 'conv.weight'], y=y, B=B, L_in=S, L_out=S, cin=F, cout=F, taps=sp.seq_kernel, stride=1,
                               dil=d, pad=pad, mode=lib.MODE_DILATED, ldx=ldh)
                    lib.layernorm_fwd(y, F, P[p + 'norm.weight'], P[p + 'norm.bias'], hn, F, rs, B * S, F, sp.layer_eps, gelu=True)
                if save:
                    convs.append(dict(hin=hcur, ldh=ldh, y=y, rs=rs))
                hcur, ldh = hn, F
            pre_out = torch.empty(B, S, F, device=dev, dtype=torch.float32)
            if ldin == F:
                lib.eltwise(lib.ELT_ADD_DROP, xin, hcur, pre_out, B * S * F, ps, self._seed(100 + b))
            else:  # block 0 reads the strided CLS rows: gather them once (small)
                xg = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                lib.copy_rows(xg, F, xin, D * F, N, F)   # the CLS rows of the token tensor
                lib.eltwise(lib.ELT_ADD_DROP, xg, hcur, pre_out, B * S * F, ps, self._seed(100 + b))
            if save:
                seq.append(dict(convs=convs, pre_out=pre_out))
            if b + 1 < sp.seq_blocks:
                act = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                lib.eltwise(lib.ELT_GELU, pre_out, None, act, B * S * F)
                xin, ldin = act, F

        return pre_out, seq

    def _seq_fused_ok(self, dil: int) -> bool:
        """the fused SequenceCNN kernel takes this layer: split precision (it has no fp32-MFMA form), 128 channels, 7 taps, a window that fits LDS"""
        sp = self.spec
        return self.seq_fused and self.split_precision and sp.feature_dim == 128 and sp.seq_kernel == 7 and 1 <= dil <= 32

    def _trunk_forward(self, e, pm, ps, save, logits=None):
        """set-fusion transformer + SequenceCNN + classifier on the tokens of `e`; returns (logits [B, S, nc], saved context or None)"""
        sp, P = self.spec, self.P
        F = sp.feature_dim
        tokens, keypad, B, S, D, N = e['tokens'], e['keypad'], e['B'], e['S'], e['D'], e['N']
        dev = tokens.device
        X, layers = self.mix(tokens, keypad, pm, save)
        ldX = X.numel() // N   # D*F, or F when the last transformer layer produced the CLS rows only
        if self.taps is not None:
            self.taps['tokens'] = tokens
            self.taps['mixer'] = X.view(N, ldX)[:, :F].reshape(B, S, F)
        pre_out, seq = self.seq(X, ldX, B, S, ps, save)
        if self.taps is not None:
            self.taps['seq_pre'] = pre_out
        if logits is None:
            logits = torch.empty(B, S, sp.num_classes, device=dev, dtype=torch.float32)
        lib.head_fwd(pre_out, F, P['classifier.weight'], P['classifier.bias'], logits, B * S, F, sp.num_classes, True)
        # seed: the dropout masks of THIS forward (counter-based RNG, regenerated in backward) -- part of the saved context, because
        # `self.step_seed` belongs to whichever forward ran last (several forwards may be in flight before one backward: ops.py)
        ctx = dict(B=B, S=S, D=D, N=N, sigs=e['sigs'], enc=e['enc'], keypad=keypad, layers=layers, seq=seq, pre_out=pre_out, pm=pm,
                   ps=ps, tokens=tokens, seed=self.step_seed) if save else None
        return logits, ctx

    def forward(self, x: dict[str, torch.Tensor], train: bool = False, save: bool = False, pack_key=None) -> torch.Tensor:
        sp = self.spec
        e = self.encode(x, save=save, pack_key=pack_key)
        logits, ctx = self._trunk_forward(e, sp.mixer_dropout if train else 0.0, sp.seq_dropout if train else 0.0, save)
        if save:
            self.ctx = ctx
        return logits

    # ------------------------------------------------------------------ backward
    def backward(self, glogits: torch.Tensor, accumulate: bool = False, hook=None):
        """Gradients of everything saved by forward(save=True) into self.G (overwrite unless accumulate)."""
        try:
            self._backward(glogits, accumulate, hook)
        finally:   # an exception mid-way must not leave queued closures / reductions (and the tensors they hold) behind
            self._deferred = None
            self._rjobs, self._cjobs = [], []

    def _backward(self, glogits, accumulate, hook):
        c = self.ctx
        if c is None:
            raise RuntimeError('backward() needs forward(save=True) first')
        self._backward_begin(accumulate)
        dev = glogits.device
        gX = self._backward_trunk(c, glogits, accumulate)
        deferred, self._deferred = self._deferred, None
        # ---- encoders
        main = torch.cuda.current_stream(dev)
        tasks = self._backward_tasks(c, gX, hook)
        for st, _ in tasks.values():
            st.wait_stream(main)
        self._interleave(tasks, trunk=(main, self._trunk_leaves(deferred, [c], hook)))
        for st, _ in tasks.values():
            main.wait_stream(st)
        self._backward_end(accumulate)
        self.ctx = None

    def _backward_begin(self, accumulate):
        if self.G is None:
            raise RuntimeError('engine was built without gradient buffers')
        self._written = set(self.G.keys()) if accumulate else set()
        self._rjobs = []
        self._cjobs = []
        # weight / bias gradients of the trunk are leaves of the backward graph: they are queued here and enqueued on this stream AFTER the
        # encoder streams have been forked, so that these small-grid kernels run beside the encoder backward instead of before it
        self._deferred = [] if self.multi_stream else None

    def _backward_end(self, accumulate):
        if not accumulate:
            missing = [name for name in self.G if name not in self._written]
            if missing:
                raise RuntimeError(f'backward left gradients unwritten: {missing[:4]}')

    def _backward_trunk(self, c, glogits, accumulate):
        """classifier, SequenceCNN and set-fusion transformer backward of one saved context on the current stream (the data-gradient
        chain; weight / bias gradients go to `self._deferred` when that is a list).  Returns the token gradient gX [N*D, F]."""
        sp, P, PB = self.spec, self.P, self.PB
        B, S, D, N, F = c['B'], c['S'], c['D'], c['N'], sp.feature_dim
        dev = glogits.device
        nc = sp.num_classes
        rows = B * S
        pm, ps = c['pm'], c['ps']
        self.step_seed = c['seed']   # regenerate the masks of the forward that saved this context, not of the latest one
        glogits = glogits.reshape(rows, nc).contiguous()

        # ---- classifier
        g_pre = torch.empty(B, S, F, device=dev, dtype=torch.float32)
        nparts = max(1, min(1024, _cdiv(rows, 16)))   # 16 rows per block: two batches of eight rows in flight (head_optim.hip)
        part = torch.empty(nparts, nc * F + nc, device=dev, dtype=torch.float32)
        lib.head_bwd(c['pre_out'], F, P['classifier.weight'], glogits, g_pre, F, part, nparts, rows, F, nc, True)
        self._colsum(part, nparts, nc * F, self.G['classifier.weight'], accumulate='classifier.weight' in self._written, ld=nc * F + nc)
        self._colsum(part.view(-1)[nc * F:], nparts, nc, self.G['classifier.bias'], accumulate='classifier.bias' in self._written, ld=nc * F + nc)
        self._written.update(('classifier.weight', 'classifier.bias'))

        # ---- SequenceCNN
        for b in reversed(range(sp.seq_blocks)):
            blk = c['seq'][b]
            gh = torch.empty(B, S, F, device=dev, dtype=torch.float32)
            lib.eltwise(lib.ELT_DROP, g_pre, None, gh, rows * F, ps, self._seed(100 + b))
            gy = None   # gradient w.r.t. layer j's conv output, when the fused kernel of layer j + 1 has already produced it
            for j in reversed(range(sp.seq_dilations)):
                d = 2 ** j
                cv = blk['convs'][j]
                p = f'sequence_mixer.dilated_convs.{b}.conv_layers.{j}.'
                if gy is None:   # the block's last layer (or the unfused path): LayerNorm + GELU backward as a launch of its own
                    gy = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                    npl = max(1, min(1024, _cdiv(rows, 32)))
                    pg = torch.empty(npl, F, device=dev, dtype=torch.float32)
                    pb = torch.empty(npl, F, device=dev, dtype=torch.float32)
                    lib.layernorm_bwd(gh, F, cv['y'], F, P[p + 'norm.weight'], P[p + 'norm.bias'], cv['rs'], None, gy, F, pg, pb, rows, F, True, npl)
                    self._colsum(pg, npl, F, self.G[p + 'norm.weight'], accumulate=(p + 'norm.weight') in self._written)
                    self._colsum(pb, npl, F, self.G[p + 'norm.bias'], accumulate=(p + 'norm.bias') in self._written)
                    self._written.update((p + 'norm.weight', p + 'norm.bias'))
                pad = (sp.seq_kernel - 1) * d if self.seq_causal else (sp.seq_kernel // 2) * d
                self._wgrad(p + 'conv.weight', g=gy, x=cv['hin'], ldx=cv['ldh'], B=B, L_in=S, L_out=S, cin=F, cout=F, taps=sp.seq_kernel,
                            stride=1, pad=pad, dil=d)
                if self._seq_fused_ok(d):
                    # data gradient of this conv with the LayerNorm + GELU backward of the layer BELOW in its epilogue (layer 0: the plain
                    # data gradient w.r.t. the block's input); the lower LayerNorm's weight / bias gradient partials come with it
                    wh, wl = self._bf[PB[p + 'conv.weight'].data_ptr()]
                    gnext = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                    if j > 0:
                        lo = blk['convs'][j - 1]
                        pl = f'sequence_mixer.dilated_convs.{b}.conv_layers.{j - 1}.'
                        ntl = B * _cdiv(S, 64)
                        part = torch.empty(ntl, 2, F, device=dev, dtype=torch.float32)
                        lib.seq_conv(x=gy, w_hi=wh, w_lo=wl, B=B, S=S, ldx=F, dil=d, pad=(sp.seq_kernel - 1) * d - pad, flip=1, mode=2, out=gnext,
                                     rs=lo['rs'], gamma=P[pl + 'norm.weight'], beta=P[pl + 'norm.bias'], yl=lo['y'], part=part, eps=sp.layer_eps)
                        self._colsum(part, ntl, F, self.G[pl + 'norm.weight'], accumulate=(pl + 'norm.weight') in self._written, ld=2 * F)
                        self._colsum(part.view(-1)[F:], ntl, F, self.G[pl + 'norm.bias'], accumulate=(pl + 'norm.bias') in self._written, ld=2 * F)
                        self._written.update((pl + 'norm.weight', pl + 'norm.bias'))
                        gy = gnext
                    else:
                        lib.seq_conv(x=gy, w_hi=wh, w_lo=wl, B=B, S=S, ldx=F, dil=d, pad=(sp.seq_kernel - 1) * d - pad, flip=1, mode=0, y=gnext)
                        gh = gnext
                else:
                    gh = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                    self._conv(x=gy, w=PB[p + 'conv.weight'], y=gh, B=B, L_in=S, L_out=S, cin=F, cout=F, taps=sp.seq_kernel, stride=1, dil=d,
                               pad=(sp.seq_kernel - 1) * d - pad, flip=1, mode=lib.MODE_DILATED)
                    gy = None
            gx = torch.empty(B, S, F, device=dev, dtype=torch.float32)
            lib.eltwise(lib.ELT_ADD, g_pre, gh, gx, rows * F)
            if b > 0:
                g_pre = torch.empty(B, S, F, device=dev, dtype=torch.float32)
                lib.eltwise(lib.ELT_GELU_BWD, c['seq'][b - 1]['pre_out'], gx, g_pre, rows * F)
            else:
                g_pre = gx  # gradient w.r.t. the CLS rows of the transformer output

        # ---- set-fusion transformer
        R = N * D
        FF = sp.mixer_dim_ff
        gX = None   # [R, F] gradient w.r.t. a layer's output; None: only the CLS rows (g_pre [N, F]) carry a gradient so far
        for l in reversed(range(sp.mixer_layers)):
            L = c['layers'][l]
            p = f'epoch_mixer.transformer_encoder.layers.{l}.'
            cls = L.get('cls', False)   # the forward ran this layer's row-wise tail on the CLS rows only: so does the backward
            if gX is None and not cls:
                gX = torch.empty(N, D, F, device=dev, dtype=torch.float32)
                lib.cls_scatter(gX, g_pre.view(N, F), N, D, F)   # only token 0 is returned (wav2sleep.py:345): zero elsewhere
                gX = gX.view(R, F)
            gin = g_pre.view(N, F) if cls else gX
            Rr, ldr = (N, D * F) if cls else (R, F)
            gf2 = torch.empty(Rr, F, device=dev, dtype=torch.float32)
            lib.eltwise(lib.ELT_DROP, gin, None, gf2, Rr * F, pm, self._seed(10 * l + 4))
            self._colgrad(p + 'linear2.bias', gf2, Rr, F)
            self._wgrad(p + 'linear2.weight', g=gf2, x=L['a1'], B=1, L_in=Rr * (FF // 128), L_out=Rr, cin=128, cout=F, taps=FF // 128,
                        stride=FF // 128, pad=0, layout=1)
            # d/d(linear1 output) = (W2^T g) * dropmask * GELU'(f1): in the data-gradient GEMM's epilogue
            gf1 = self._linear(gf2, PB[p + 'linear2.weight'], None, Rr, F, FF, fuse=lib.FUSE_GELU_BWD_DROP, aux=L['f1'], ld_aux=FF, drop_p=pm,
                               drop_seed=self._seed(10 * l + 3))
            self._colgrad(p + 'linear1.bias', gf1, Rr, FF)
            self._wgrad(p + 'linear1.weight', g=gf1, x=L['h2'], B=1, L_in=Rr, L_out=Rr, cin=F, cout=FF, taps=1, stride=1, pad=0)
            gh2 = self._linear(gf1, PB[p + 'linear1.weight'], None, Rr, FF, F)
            gX1 = torch.empty(Rr, F, device=dev, dtype=torch.float32)
            self._ln_bwd(p + 'norm2', gh2, L['X1'], L['rs2'], gin, gX1, Rr)
            gproj = torch.empty(Rr, F, device=dev, dtype=torch.float32)
            lib.eltwise(lib.ELT_DROP, gX1, None, gproj, Rr * F, pm, self._seed(10 * l + 2))
            self._colgrad(p + 'self_attn.out_proj.bias', gproj, Rr, F)
            self._wgrad(p + 'self_attn.out_proj.weight', g=gproj, x=L['ao'], B=1, L_in=Rr, L_out=Rr, cin=F, cout=F, taps=1, stride=1, pad=0,
                        **(dict(ldx=ldr) if cls else {}))
            if cls:   # back to all token rows: the attention spreads the CLS query's gradient over every token's keys and values
                gao = torch.empty(R, F, device=dev, dtype=torch.float32)
                lib.cls_scatter(gao, None, N, D, F)   # zero rows 1 .. D-1; the projection below writes the CLS rows (row stride D*F)
                self._linear(gproj, PB[p + 'self_attn.out_proj.weight'], None, N, F, F, y=gao, ldy=D * F)
                gX1f = torch.empty(N, D, F, device=dev, dtype=torch.float32)
                lib.cls_scatter(gX1f, gX1, N, D, F)
                gX1 = gX1f.view(R, F)
            else:
                gao = self._linear(gproj, PB[p + 'self_attn.out_proj.weight'], None, R, F, F)
            gqkv = torch.empty(R, 3 * F, device=dev, dtype=torch.float32)
            lib.attn_bwd(L['qkv'], c['keypad'], gao, gqkv, N, D, sp.mixer_nhead, pm, self._seed(10 * l + 1), nq=1 if cls else D)
            self._colgrad(p + 'self_attn.in_proj_bias', gqkv, R, 3 * F)
            self._wgrad(p + 'self_attn.in_proj_weight', g=gqkv, x=L['h'], B=1, L_in=R, L_out=R, cin=F, cout=3 * F, taps=1, stride=1, pad=0)
            gh = self._linear(gqkv, PB[p + 'self_attn.in_proj_weight'], None, R, 3 * F, F)
            gXn = torch.empty(R, F, device=dev, dtype=torch.float32)
            self._ln_bwd(p + 'norm1', gh, L['X'], L['rs1'], gX1, gXn, R)
            gX = gXn
        # CLS / register-token parameter [1, 1, F, R+1]: column r = sum of the token-r rows' gradients
        R1 = sp.register_tokens + 1
        if R1 == 1:
            self._colgrad('epoch_mixer.register_tokens', gX, N, F, ldg=D * F)
        else:
            c['rt_tmp'] = tmp = torch.empty(R1, F, device=dev, dtype=torch.float32)
            for r in range(R1):
                nparts = max(1, min(1024, _cdiv(N, 64)))
                part = torch.empty(nparts, F, device=dev, dtype=torch.float32)
                lib.bias_grad(gX.view(-1)[r * F:], N, F, D * F, part, nparts)
                self._colsum(part, nparts, F, tmp[r])
        if sp.embed_signals:
            # embedding rows: sum over the rows of the samples that have the signal (absent signals and missing samples get zero)
            ew = 'signal_encoders.embedder.weight'
            if ew not in self._written:
                lib.zero_(self.G[ew])
                self._written.add(ew)
            order = sorted(sp.signal_map)
            for m, ec in enumerate(c['enc']):
                gk = gX.view(N, D, F)[:, R1 + m, :].reshape(B, S, F) * ec['keep'][:, None, None]
                nparts = max(1, min(1024, _cdiv(N, 64)))
                part = torch.empty(nparts, F, device=dev, dtype=torch.float32)
                lib.bias_grad(gk, N, F, F, part, nparts)
                self._colsum(part, nparts, F, self.G[ew][order.index(ec['sig'])], accumulate=True)
        encs = [ec['enc'] for ec in c['enc']]
        if not accumulate:
            # encoders whose signals are not in this batch get no backward: their gradient is zero.  Written here, BEFORE any range is
            # handed to the reducer, so that nothing touches a range on the compute stream once its all-reduce may be in flight.
            for name, g in self.G.items():
                if name.startswith('signal_encoders.encoders.') and name.split('.')[2] not in encs and name not in self._written:
                    lib.zero_(g)
                    self._written.add(name)
        return gX

    def _trunk_leaves(self, deferred, ctxs, hook):
        """Generator: the queued weight / bias gradients of the trunk (of every context in `ctxs`, in order), three per turn."""
        sp = self.spec
        F, R1 = sp.feature_dim, sp.register_tokens + 1
        k = 0
        while deferred:
            deferred.pop(0)()   # popped before it runs: the closure (and the trunk gradient tensors it holds) is released right after
            k += 1
            if k % 3 == 0:
                yield
        self._flush_reduce()
        if R1 > 1:
            rt = 'epoch_mixer.register_tokens'
            for c in ctxs:
                g = self.G[rt].view(F, R1)
                g.add_(c['rt_tmp'].t()) if rt in self._written else g.copy_(c['rt_tmp'].t())
                self._written.add(rt)
        if hook is not None:
            encs = [ec['enc'] for ec in ctxs[0]['enc']]
            hook('_tail')  # mixer + sequence CNN + classifier gradients are final: their all-reduce can start
            for e in dict.fromkeys(sp.signal_map.values()):
                if e not in encs:
                    hook(e)   # absent encoder: its (zero) range still takes part in the collective, the bucket layout is static
        yield

    def _backward_tasks(self, c, gX, hook):
        """encoder -> (stream, [generators]) of one context's encoder backward (reads the token gradient gX)"""
        sp = self.spec
        B, S, D, F, R1 = c['B'], c['S'], c['D'], sp.feature_dim, sp.register_tokens + 1
        dev = gX.device
        encs = [ec['enc'] for ec in c['enc']]
        tasks = {}
        for m, ec in sorted(enumerate(c['enc']), key=lambda me: -COLS_TO_SAMPLES_PER_EPOCH[me[1]['sig']]):
            st = self._side_stream(ec['enc'], dev)

            def run(m=m, ec=ec):
                yield from self._encoder_backward(ec, gX.view(-1)[(R1 + m) * F:], D * F)
                self._flush_reduce()
                if hook is not None and ec['enc'] not in encs[m + 1:]:
                    hook(ec['enc'])
            tasks.setdefault(ec['enc'], (st, []))[1].append(run())
        return tasks

    def train_waves(self, x: dict[str, torch.Tensor], ce, bounds, seeds, logits: torch.Tensor, pack_key=None, accumulate: bool = False,
                    hook=None):
        """Forward + backward of one batch as SAMPLE WAVES (`bounds`: [(b0, b1), ...]) in a software pipeline.  Every sample is independent
        up to the loss (instance norm in the encoders, layer norm in the trunk), so the trunk of wave w -- a serial chain of ~100 small
        launches that leaves most of the device idle -- runs on the main stream BESIDE the encoder forward of wave w+1 / the encoder
        backward of wave w-1 on the encoder streams, instead of between the two with nothing beside it:

            encoder streams:  F(0) F(1) ... F(n-1) | B(0)       B(1) ...        B(n-1)
            main stream:           T(0) ... T(n-2)   T(n-1) + trunk leaves (weight gradients of every wave)      -> joined

        T(w) = set-fusion transformer + SequenceCNN + classifier + loss + their data-gradient chain of wave w; it waits for the event
        recorded behind F(w) on every encoder stream, and B(w) waits for the event recorded behind T(w).  `ce(logits_w, b0, b1)` enqueues
        the loss of one wave and returns d loss / d logits_w; the logits go to `logits[b0:b1]`.  Gradients of wave 0 overwrite (unless
        `accumulate`), the later waves add -- the sums are those of gradient accumulation over the waves, in a fixed order."""
        try:
            self._train_waves(x, ce, bounds, seeds, logits, pack_key, accumulate, hook)
        finally:
            self._deferred = None
            self._rjobs, self._cjobs = [], []

    def _train_waves(self, x, ce, bounds, seeds, logits, pack_key, accumulate, hook):
        sp = self.spec
        self._validate(x)
        self.ensure_packed(pack_key, need_bwd=True)
        dev = logits.device
        main = torch.cuda.current_stream(dev)
        pm, ps = sp.mixer_dropout, sp.seq_dropout
        # ---- encoder forward of every wave (tokens are allocated and their CLS rows written on the main stream BEFORE the fork)
        waves = []
        for (b0, b1), seed in zip(bounds, seeds):
            self.step_seed = seed
            waves.append(self._encode_begin({s: v[b0:b1] for s, v in x.items()}, True))
        streams = list(dict.fromkeys(st for _, tasks in waves for st, _ in tasks.values()))
        for st in streams:
            st.wait_stream(main)
        for e, _ in waves:
            self._encode_keypad(e)
        fwd_done = []
        for (e, tasks), seed in zip(waves, seeds):
            self.step_seed = seed
            self._interleave(tasks)
            evs = []
            for st in streams:
                if st != main:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
            fwd_done.append(evs)
        # ---- trunk of every wave on the main stream
        self._backward_begin(accumulate)
        ctxs, gXs, trunk_done = [], [], []
        for w, ((e, _), seed, (b0, b1)) in enumerate(zip(waves, seeds, bounds)):
            self.step_seed = seed
            for ev in fwd_done[w]:
                main.wait_event(ev)
            lg, c = self._trunk_forward(e, pm, ps, True, logits=logits[b0:b1])
            gX = self._backward_trunk(c, ce(lg, b0, b1), accumulate or w > 0)
            # a weight may appear once per reduction flush: close this wave's queue (behind its own leaves when those are deferred)
            if self._deferred is not None:
                rj, cj, self._rjobs, self._cjobs = self._rjobs, self._cjobs, [], []

                def flush(rj=rj, cj=cj):
                    self._rjobs[:0] = rj
                    self._cjobs[:0] = cj
                    self._flush_reduce()
                self._deferred.append(flush)
            else:
                self._flush_reduce()
            ev = torch.cuda.Event()
            ev.record(main)
            ctxs.append(c)
            gXs.append(gX)
            trunk_done.append(ev)
        deferred, self._deferred = self._deferred, None
        # ---- encoder backward of every wave; the trunk's leaves ride beside the first one
        for w, (c, gX, seed) in enumerate(zip(ctxs, gXs, seeds)):
            self.step_seed = seed
            tasks = self._backward_tasks(c, gX, hook if w == len(ctxs) - 1 else None)
            for st, _ in tasks.values():
                if st != main:
                    st.wait_event(trunk_done[w])
            self._interleave(tasks, trunk=(main, self._trunk_leaves(deferred, ctxs, hook)) if w == 0 else None)
        for st in streams:
            main.wait_stream(st)   # every tensor that crossed streams (tokens, gX, keeps) is still referenced here
        self._backward_end(accumulate)

    def _ln_bwd(self, pfx, g, x, rstat, gadd, gx, rows):
        F = self.spec.feature_dim
        npl = max(1, min(1024, _cdiv(rows, 32)))
        pg = torch.empty(npl, F, device=g.device, dtype=torch.float32)
        pb = torch.empty(npl, F, device=g.device, dtype=torch.float32)
        lib.layernorm_bwd(g, F, x, F, self.P[pfx + '.weight'], self.P[pfx + '.bias'], rstat, gadd, gx, F, pg, pb, rows, F, False, npl)
        self._colsum(pg, npl, F, self.G[pfx + '.weight'], accumulate=(pfx + '.weight') in self._written)
        self._colsum(pb, npl, F, self.G[pfx + '.bias'], accumulate=(pfx + '.bias') in self._written)
        self._written.update((pfx + '.weight', pfx + '.bias'))

    def _bstats(self, part, B, nt, C, count):
        return self._finalize(part, B, nt, C, count, 1)

    def _encoder_backward(self, ec, gtok, ldtok):
        """Generator (see _encoder_forward): yields after every block."""
        sp, P, PB = self.spec, self.P, self.PB
        enc, B, S, F = ec['enc'], ec['B'], ec['S'], sp.feature_dim
        dev = gtok.device
        pfx = f'signal_encoders.encoders.{enc}.'
        ch = sp.channels(enc)
        cl = ch[-1]
        if sp.output_norm:   # token = LayerNorm(z): back through it first
            gact = torch.empty(B, S, F, device=dev, dtype=torch.float32)
            npl = max(1, min(1024, _cdiv(B * S, 32)))
            pg = torch.empty(npl, F, device=dev, dtype=torch.float32)
            pb = torch.empty(npl, F, device=dev, dtype=torch.float32)
            wn, bn = pfx + 'output_norm.weight', pfx + 'output_norm.bias'
            lib.layernorm_bwd(gtok, ldtok, ec['zact'], F, P[wn], P[bn], ec['rs_out'], None, gact, F, pg, pb, B * S, F, False, npl)
            self._colsum(pg, npl, F, self.G[wn], accumulate=wn in self._written)
            self._colsum(pb, npl, F, self.G[bn], accumulate=bn in self._written)
            self._written.update((wn, bn))
            gtok, ldtok = gact, F
        # z = keep * GELU(zpre): g_zpre
        gz = torch.empty(B, S, F, device=dev, dtype=torch.float32)
        lib.gelu_bwd_rows(gtok, ldtok, ec['zpre'], ec['keep'], S, gz, B * S, F)
        self._colgrad(pfx + 'linear.bias', gz, B * S, F)
        self._wgrad(pfx + 'linear.weight', g=gz, x=ec['plast'], B=B, L_in=4 * S, L_out=S, cin=cl, cout=F, taps=4, stride=4, pad=0,
                    pro_h=lib.PRO_GELU, layout=1)
        # data gradient through the dense layer, times GELU'(plast): rows = B*S, cout' = 4*cl
        gpre = torch.empty(B, 2 * 2 * S, cl, device=dev, dtype=torch.float32)
        self._conv(x=gz, w=PB[pfx + 'linear.weight'], y=gpre, B=1, L_in=B * S, L_out=B * S, cin=F, cout=4 * cl, taps=1, stride=1, pad=0,
                   epi=lib.EPI_GP, aux=ec['plast'], ld_aux=4 * cl)
        B = ec['Bc']   # chunk-causal: the conv stack ran on B*S one-epoch samples; gpre [B, 4S, C] is its [B*S, 4, C] gradient
        bs3_folded = None   # conv3 backward statistics of block i produced by block i+1's fused conv1 kernel (no gp_stats pre-pass)

        def fused(j):
            return lib.bwd_fused_supported(ch[j], ch[j]) and self._fused_bwd_ok

        def will_fold(j):
            bj = ec['blocks'][j]
            return (j > 0 and self.split_precision and lib.bwd_fused_supported(bj['c'], bj['cin']) and lib.bwd_fused_folds_residual(bj['c'], bj['cin'])
                    and not (bj['L'] & 1))
        # fp16 gradient chain: from its ENTRY block (the topmost fused block whose conv1 is not a residual-fold kernel: its gpre arrives
        # fp32 from outside the chain and only its conv3 kernel reads it) down to block 0, provided every block below is one the fp16
        # kernels cover (fused, residual-fold conv1; block 0: the first-layer kernels).  Production: blocks 3 .. 0 of every encoder.
        entry = -1
        if self.grad_fp16 and self.split_precision and self._fused_bwd_ok and fused(0):
            for j in reversed(range(1, len(ch))):
                if fused(j) and not will_fold(j) and (ch[j], ec['blocks'][j]['cin']) in ((16, 16), (32, 32)) and all(fused(k) and will_fold(k) for k in range(1, j)):
                    entry = j
                    break
        hdrs = torch.zeros(3 * (entry + 1) + 1, 2, device=dev, dtype=torch.float32) if entry >= 0 else None   # {scale, max} per chain tensor
        nh = [0]

        def new_hdr():
            nh[0] += 1
            return hdrs[nh[0] - 1]
        wd_folded = False   # block 0's downsample weight gradient already produced by block 1's conv1 kernel
        gpre_hdr = None   # header of gpre: set once gpre is a chain tensor (fp16) or the chain's fp32 entry (gp_stats publishes its maximum)
        for i in reversed(range(len(ch))):
            blk = ec['blocks'][i]
            p = f'{pfx}cnn.{i}.'
            c, cin, L = blk['c'], blk['cin'], blk['L']
            Lh = L // 2
            h16 = i <= entry                       # this block's gn2 / gn1 (and its gprev, if i > 0) are fp16
            gdt = torch.float16 if h16 else torch.float32
            ghalf = gpre.dtype == torch.float16
            # conv3 (stride 2): pre-pass for the instance-norm backward sums, then data + weight gradient
            if bs3_folded is not None:
                bs3, bs3_folded = bs3_folded, None
            else:
                tile = 512
                while tile > 64 and B * _cdiv(Lh, tile) < 1024:   # deep layers: enough workgroups to fill the device
                    tile //= 2
                if h16 or ghalf:
                    nt = _cdiv(Lh, tile)
                    part = torch.empty(B, nt, 2, c, device=dev, dtype=torch.float32)
                    if not ghalf:
                        gpre_hdr = new_hdr()
                    lib.gp_stats(gpre, blk['y3'], blk['st3'], part, B, Lh, c, tile, hdr_g=gpre_hdr if ghalf else None, hdr_amax=None if ghalf else gpre_hdr)
                    bs3 = self._bstats(part, B, nt, c, Lh)
                else:
                    nt = _cdiv(Lh, tile)
                    part = torch.empty(B, nt, 2, c, device=dev, dtype=torch.float32)
                    lib.gp_stats(gpre, blk['y3'], blk['st3'], part, B, Lh, c, tile)
                    bs3 = self._bstats(part, B, nt, c, Lh)
            # block 0 in the first-layer recompute flow: conv1's weight gradient rides in conv2's backward kernel (per-tile sums of gn1 x signal
            # taps), so gn1 -- which only that weight gradient would read -- is never stored (fp32 chain, split-precision kernels)
            fold_w1 = (i == 0 and blk['y1'] is None and ec.get('xmom') is not None and not h16 and self.split_precision and c == 16
                       and lib.bwd_fused_supported(c, c) and self._fused_bwd_ok)
            gn2 = torch.empty(B, L, c, device=dev, dtype=gdt)
            gn1 = None if fold_w1 else torch.empty(B, L, c, device=dev, dtype=gdt)
            h2 = new_hdr() if h16 else None
            h1 = new_hdr() if h16 else None
            if lib.bwd_fused_supported(c, c) and self._fused_bwd_ok:
                bs2 = self._bwd_fused(p + 'conv3.conv.weight', g=gpre, y=blk['y3'], st_k=blk['st3'], bst_k=bs3, pro=lib.PRO_INBWD_GP,
                                      xin=blk['y2'], st_in=blk['st2'], add_even=None, gout=gn2, want_part=True, B=B, Lg=Lh, Lh=L, cg=c, ch=c, stride=2,
                                      gmode=(2 if ghalf else 1) if h16 else 0, hdr_g=gpre_hdr, hdr_o=h2)
                first = i == 0 and blk['y1'] is None   # block 0's conv1 output is recomputed from the raw signal
                if fold_w1:
                    part_w1 = torch.empty(B, _cdiv(L, lib.bwd_fused_tile(c, c, 1, False, True)), 48, device=dev, dtype=torch.float32)
                bs1 = self._bwd_fused(p + 'conv2.conv.weight', g=gn2, y=blk['y2'], st_k=blk['st2'], bst_k=bs2, pro=lib.PRO_INBWD,
                                      xin=ec['x'] if first else blk['y1'], st_in=blk['st1'], add_even=None, gout=gn1, want_part=True, B=B,
                                      Lg=L, Lh=L, cg=c, ch=c, stride=1, w1=P[p + 'conv1.conv.weight'] if first else None,
                                      gmode=2 if h16 else 0, hdr_g=h2, hdr_o=h1, part_w1=part_w1 if fold_w1 else None)
            else:
                if not (L & 1) and self._bwd_wide_ok(B, L, c, c, stride=2):
                    bs2 = self._bwd_wide(p + 'conv3.conv.weight', g=gpre, y=blk['y3'], st_k=blk['st3'], bst_k=bs3, xin=blk['y2'], st_in=blk['st2'],
                                         add_even=None, gout=gn2, want_part=True, B=B, L=L, cg=c, ch=c, stride=2)
                else:
                    bs2 = self._conv_part(x=gpre, x2=blk['y3'], w=PB[p + 'conv3.conv.weight'], y=gn2, B=B, L_in=Lh, L_out=L, cin=c, cout=c, taps=3, stride=2,
                                          pad=self.kpad, mode=lib.MODE_UP2, pro=lib.PRO_INBWD_GP, pro_stats=blk['st3'], pro_bstats=bs3, epi=lib.EPI_GP,
                                          aux=blk['y2'], aux_stats=blk['st2'], kind=1)
                    self._wgrad(p + 'conv3.conv.weight', g=gpre, g2=blk['y3'], g_stats=blk['st3'], g_bstats=bs3, pro_g=lib.PRO_INBWD_GP, x=blk['y2'],
                                x_stats=blk['st2'], pro_h=lib.PRO_IN_GELU, B=B, L_in=L, L_out=Lh, cin=c, cout=c, taps=3, stride=2, pad=self.kpad)
                if self._bwd_wide_ok(B, L, c, c):
                    bs1 = self._bwd_wide(p + 'conv2.conv.weight', g=gn2, y=blk['y2'], st_k=blk['st2'], bst_k=bs2, xin=blk['y1'], st_in=blk['st1'],
                                         add_even=None, gout=gn1, want_part=True, B=B, L=L, cg=c, ch=c)
                else:
                    bs1 = self._conv_part(x=gn2, x2=blk['y2'], w=PB[p + 'conv2.conv.weight'], y=gn1, B=B, L_in=L, L_out=L, cin=c, cout=c, taps=3, stride=1,
                                          pad=2 - self.kpad, flip=1, pro=lib.PRO_INBWD, pro_stats=blk['st2'], pro_bstats=bs2, epi=lib.EPI_GP,
                                          aux=blk['y1'], aux_stats=blk['st1'], kind=1)
                    self._wgrad(p + 'conv2.conv.weight', g=gn2, g2=blk['y2'], g_stats=blk['st2'], g_bstats=bs2, pro_g=lib.PRO_INBWD, x=blk['y1'],
                                x_stats=blk['st1'], pro_h=lib.PRO_IN_GELU, B=B, L_in=L, L_out=L, cin=c, cout=c, taps=3, stride=1, pad=self.kpad)
            del gn2
            hp = new_hdr() if (h16 and i > 0) else None
            h16_next = (i - 1) <= entry   # the block below is the fp16 chain's entry: its gp_stats launch also publishes max |gpre| (keep it)
            if i > 0 and self.split_precision and lib.bwd_fused_supported(c, cin) and lib.bwd_fused_folds_residual(c, cin) and not (L & 1):
                # conv1 + the whole residual branch (its data gradient AND its weight gradient) in one pass over the tensors
                gprev = torch.empty(B, L, cin, device=dev, dtype=gdt)
                prev = ec['blocks'][i - 1] if self.fold_gp else dict(y3=None, st3=None)   # fold its conv3-backward statistics pre-pass in
                # block 1: this kernel's gout is block 0's gpre -- block 0's downsample weight gradient (gpre x every other signal sample)
                # rides in its epilogue instead of a pass of its own over that tensor
                fold_wd = False
                if i == 1 and ec.get('xmom') is not None and entry < 0 and c == 16 and cin == 16 and ec['blocks'][0]['y1'] is None and self._fused_bwd_ok:
                    fold_wd = wd_folded = True
                bs3_folded = self._bwd_fused(p + 'conv1.conv.weight', g=gn1, y=blk['y1'], st_k=blk['st1'], bst_k=bs1, pro=lib.PRO_INBWD,
                                             xin=blk['pin'], st_in=None, add_even=None, gout=gprev, want_part=self.fold_gp, B=B, Lg=L, Lh=L, cg=c,
                                             ch=cin, stride=1, gpre=gpre, down=p + 'downsample.weight', y3p=prev['y3'], st3p=prev['st3'],
                                             gmode=2 if h16 else 0, hdr_g=h1, hdr_p=gpre_hdr, hdr_o=hp, x0=ec['x'] if fold_wd else None,
                                             down0=f'{pfx}cnn.0.downsample.weight')
                if not self.fold_gp:
                    bs3_folded = None
                gpre, gpre_hdr = gprev, hp
            elif (i > 0 and self.bwd_wide_rd and self.bwd_wide and self.split_precision and not (L & 1) and (p + 'downsample.weight') in self.G
                  and c >= 64 and self.PB[p + 'downsample.weight'].data_ptr() in self._bf
                  and self.PB[p + 'conv1.conv.weight'].data_ptr() in self._bf and lib.bwd_wide_takes(B, L, c, cin, 1, False, rd=True)):
                # 64-channel conv1: the whole residual branch (Wd^T gpre into the data gradient, the downsample weight gradient) and the
                # previous block's conv3-backward statistics in the one-pass kernel -- no R tensor, no 1x1 conv launch, no separate weight gradient
                gprev = torch.empty(B, L, cin, device=dev, dtype=torch.float32)
                prev = ec['blocks'][i - 1] if (self.fold_gp and not h16_next) else None
                bs3_folded = self._bwd_wide(p + 'conv1.conv.weight', g=gn1, y=blk['y1'], st_k=blk['st1'], bst_k=bs1, xin=blk['pin'], st_in=None,
                                            add_even=None, gout=gprev, want_part=prev is not None, B=B, L=L, cg=c, ch=cin,
                                            y3p=prev['y3'] if prev else None, st3p=prev['st3'] if prev else None, gpre=gpre, down=p + 'downsample.weight')
                gpre, gpre_hdr = gprev, hp
            elif i > 0:
                # residual 1x1/stride-2 branch: R = Wd^T gpre, added at even positions inside conv1's data-gradient epilogue
                Rr = torch.empty(B, Lh, cin, device=dev, dtype=torch.float32)
                self._conv(x=gpre, w=PB[p + 'downsample.weight'], y=Rr, B=B, L_in=Lh, L_out=Lh, cin=c, cout=cin, taps=1, stride=1, pad=0)
                gprev = torch.empty(B, L, cin, device=dev, dtype=gdt)
                if lib.bwd_fused_supported(c, cin) and self._fused_bwd_ok:
                    self._bwd_fused(p + 'conv1.conv.weight', g=gn1, y=blk['y1'], st_k=blk['st1'], bst_k=bs1, pro=lib.PRO_INBWD, xin=blk['pin'],
                                    st_in=None, add_even=Rr, gout=gprev, want_part=False, B=B, Lg=L, Lh=L, cg=c, ch=cin, stride=1,
                                    gmode=2 if h16 else 0, hdr_g=h1, hdr_o=hp)
                elif self._bwd_wide_ok(B, L, c, cin, hst=False):
                    # ... with the previous block's conv3-backward statistics pre-pass folded in (its gp_stats launch read gprev and y3 again)
                    prev = ec['blocks'][i - 1] if (self.fold_gp and not h16_next) else None
                    bs3_folded = self._bwd_wide(p + 'conv1.conv.weight', g=gn1, y=blk['y1'], st_k=blk['st1'], bst_k=bs1, xin=blk['pin'], st_in=None,
                                                add_even=Rr, gout=gprev, want_part=prev is not None, B=B, L=L, cg=c, ch=cin,
                                                y3p=prev['y3'] if prev else None, st3p=prev['st3'] if prev else None)
                else:
                    self._conv(x=gn1, x2=blk['y1'], w=PB[p + 'conv1.conv.weight'], y=gprev, B=B, L_in=L, L_out=L, cin=c, cout=cin, taps=3,
                               stride=1, pad=2 - self.kpad, flip=1, pro=lib.PRO_INBWD, pro_stats=blk['st1'], pro_bstats=bs1, epi=lib.EPI_GP, aux=blk['pin'],
                               add_even=Rr)
                    self._wgrad(p + 'conv1.conv.weight', g=gn1, g2=blk['y1'], g_stats=blk['st1'], g_bstats=bs1, pro_g=lib.PRO_INBWD,
                                x=blk['pin'], pro_h=lib.PRO_GELU, B=B, L_in=L, L_out=L, cin=cin, cout=c, taps=3, stride=1, pad=self.kpad)
                self._wgrad(p + 'downsample.weight', g=gpre, x=blk['pin'], pro_h=lib.PRO_GELU, B=B, L_in=L, L_out=Lh, cin=cin, cout=c,
                            taps=1, stride=2, pad=0)
                gpre, gpre_hdr = gprev, hp
            elif gn1 is None:   # (fold_w1) conv1's weight gradient from the folded partials; the downsample weight gradient on its own
                n1, nd = p + 'conv1.conv.weight', p + 'downsample.weight'
                dw1 = torch.empty(B, 48, device=dev, dtype=torch.float32)
                lib.enc_first_wgrad(ec['xmom'], ec['xmom'].shape[1], P[n1], part_w1, blk['st1'], bs1, dw1, B, part_w1.shape[1])
                self._colsum(dw1, B, 48, self.G[n1], accumulate=n1 in self._written)
                if not wd_folded:
                    nslab = max(1, min(1024, _cdiv(B * Lh, 2048)))
                    slab = torch.empty(nslab, 16, device=dev, dtype=torch.float32)
                    lib.enc_first_dwd(ec['x'], gpre, slab, nslab, B, L)
                    self._colsum(slab, nslab, 16, self.G[nd], accumulate=nd in self._written)
                self._written.update((n1, nd))
            else:
                nslab = max(1, min(1024, _cdiv(B * L, 4096)))
                slab = torch.empty(nslab, 64, device=dev, dtype=torch.float32)
                lib.enc_first_bwd(ec['x'], gn1, blk['y1'], blk['st1'], bs1, gpre, slab, nslab, B, L, c, w1=P[p + 'conv1.conv.weight'], causal=self.causal,
                                  hdr_n=h1, hdr_p=gpre_hdr)
                n1, nd = p + 'conv1.conv.weight', p + 'downsample.weight'
                self._colsum(slab, nslab, 48, self.G[n1], accumulate=n1 in self._written, ld=64)
                self._colsum(slab.view(-1)[48:], nslab, 16, self.G[nd], accumulate=nd in self._written, ld=64)
                self._written.update((n1, nd))
            yield
