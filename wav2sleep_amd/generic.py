"""Generic (untuned) path: every module variant the reference can be configured into besides its shipped production model.

`models/utils.py:26-96` offers batch / layer / rms / group / instance / no normalisation and ReLU / LeakyReLU / GELU / SiLU / linear
activations; `MultiModalAttentionEmbedder` and `SequenceCNN` take any feature size and head count (the reference's own
tests/model/test_causality.py builds feature_dim 16, ReLU, BatchNorm, 4 layers x 4 heads); `models/ppgnet.py` is a ninth-of-a-kind CNN
from the same blocks.  The production model (GELU, instance / layer norm, 128 features, 16-wide heads) runs on the fused kernels
(engine.py); everything else runs here: the reference's call graph, layer by layer, on channels-last device tensors, every convolution and
GEMM through `w2s_conv_forward` (no prologue; contractions wider than 128 channels as accumulating launches) and the norms / activations /
attention core through the kernels of csrc/generic.hip.  With `grad=True` the walker also records a TAPE -- one entry per launch group
(output, inputs, backward closure) -- and `backward()` replays it in reverse: data gradients are `w2s_conv_forward` again (flipped taps,
W2S_MODE_UP2 for the stride-2 convs, transposed weights for the GEMMs), weight gradients `w2s_wgrad` (cut into the channel blocks its
kernels take), norm / activation / attention backward the kernels of csrc/generic.hip; that is what `SleepPPGNet` (which the reference
trains, models/ppgnet.py) and the non-production `Wav2Sleep` configurations train on (ops.py wraps it as one autograd node).  Nothing of
torch's autograd runs underneath.  Host-side torch is used for plumbing only: folding per-(sample, channel) statistics and affine parameters into
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


def _chunks(c: int):
    """Contraction widths of the launches that cover c channels: the kernels take a power-of-two channel count in [16, 128], so c (a
    multiple of 16) is cut greedily -- 256 -> 128 + 128, 48 -> 32 + 16 -- and the launches after the first accumulate.  [(offset, width)]"""
    if c < 16 or c % 16:
        raise NotImplementedError(f'{c} input channels: the generic kernels contract over multiples of 16')
    out, o = [], 0
    while o < c:
        w = 128
        while w > c - o:
            w >>= 1
        out.append((o, w))
        o += w
    return out


class _Lazy:
    """act(y * scale + shift) NOT materialised: the output of a ConvLayer1D whose consumer is another convolution.  That convolution and its
    weight gradient apply it on load (W2S_PRO_AFFINE + act, `ss` = (scale, shift) per (sample, channel)), so the tensor is never written.
    `mat` caches the materialised form for any other consumer."""
    __slots__ = ('y', 'ss', 'scale', 'shift', 'act', 'mat')

    def __init__(self, y, ss, scale, shift, act):
        self.y, self.ss, self.scale, self.shift, self.act, self.mat = y, ss, scale, shift, act, None

    @property
    def shape(self):
        return self.y.shape


class _LazyG:
    """The gradient of a conv output y behind a per-channel norm + activation, NOT materialised: gy = (scale g) act'(y scale + shift) + z c + d.
    The convolution's data-gradient launch and weight gradient form it on load (W2S_PRO_AFFINE_BWD + act; g = gradient of the activation's
    output, ss = (scale, shift), cd = (c, d) per (sample, channel))."""
    __slots__ = ('g', 'y', 'ss', 'cd', 'act')

    def __init__(self, g, y, ss, cd, act):
        self.g, self.y, self.ss, self.cd, self.act = g, y, ss, cd, act


class GenericForward:
    """Stateless walker over the parameter containers of wav2sleep.py (same attribute names as the reference modules).
    grad=True: every launch group is recorded on a tape and nothing is overwritten in place; `backward(out, g)` then returns
    {parameter: gradient}."""

    def __init__(self, training: bool = False, seed: int = 0, grad: bool = False):
        self.training = training
        self.seed = seed
        self.grad = grad
        self._site = 0
        self._x16 = None    # (the one-channel input, its 16-channel zero-padded copy)
        self.tape = []      # (output tensor, input tensors, fn: gradient of the output -> gradients of the inputs)
        self.pgrads = {}    # nn.Parameter -> gradient (the parameter's shape)
        lib.load()

    # ------------------------------------------------------------------ tape
    def _rec(self, out, inputs, fn, share=None, acc_ok=False):
        if self.grad:
            if share is not None:
                fn.share = share
            if acc_ok:
                fn.acc_ok = True
            self.tape.append((out, tuple(inputs), fn))

    def _add(self, a, b):
        if a.numel() & 3:   # (a classifier bias of 5 classes: the elementwise kernel works on quads)
            return a + b
        o = torch.empty_like(a)
        lib.eltwise(lib.ELT_ADD, a, b.contiguous(), o, a.numel())
        return o

    def _pgrad(self, param, g):
        g = g.reshape(param.shape)
        self.pgrads[param] = self._add(self.pgrads[param], g) if param in self.pgrads else g.contiguous()

    def _view(self, t, *shape):
        v = t.view(*shape)
        self._rec(v, (t,), lambda g: (g.reshape(t.shape),), share='view')
        return v

    def _mat(self, t):
        """the tensor behind t (a _Lazy output is written out now, once)"""
        if not isinstance(t, _Lazy):
            return t
        if t.mat is None:
            B, L, Cc = t.y.shape
            out = torch.empty_like(t.y)
            lib.affine_act(t.y, Cc, t.scale, t.shift, Cc if (t.scale is not None and t.scale.shape[0] > 1) else 0, out, Cc, L, B * L, Cc, t.act, 0.01)
            t.mat = out
            self._rec(out, (t,), lambda g: (g,), share='view')
        return t.mat

    def backward(self, out, g_out) -> dict:
        """Replay the tape in reverse from d(loss)/d(out); returns {parameter: gradient} (a parameter used twice: the sum)."""
        if not self.grad:
            raise RuntimeError('GenericForward(grad=False) keeps no tape')
        grads = {id(out): g_out.contiguous().float()}
        owned = set()   # ids of gradient buffers that exactly one tape entry still refers to (fresh outputs of a closure): safe to add into
        for o, inputs, fn in reversed(self.tape):
            g = grads.pop(id(o), None)
            if g is None:
                continue
            g_owned = id(g) in owned
            owned.discard(id(g))
            k0 = id(inputs[0]) if len(inputs) == 1 else None
            if getattr(fn, 'acc_ok', False) and k0 in grads and id(grads[k0]) in owned:
                fn(g, acc=grads[k0])   # a convolution whose input already has a gradient (the residual branch ran first): y += in the data-gradient launch
                continue
            share = getattr(fn, 'share', None)   # 'all': the outputs ARE g (fan-out of an add); 'view': a view of g
            for t, gi in zip(inputs, fn(g)):
                if gi is None:
                    continue
                k = id(t)
                if k in grads:
                    owned.discard(id(grads[k]))
                    grads[k] = self._add(grads[k], gi)
                    owned.add(id(grads[k]))
                else:
                    grads[k] = gi
                    if share is None or (share == 'view' and g_owned):
                        owned.add(id(gi))
        self.tape = []
        return self.pgrads

    def _rowsum(self, g, rows, Cc, ld):
        """sum over `rows` rows of a [rows][ld] gradient -> [Cc] (two fixed-order stages)"""
        nparts = max(1, min(1024, _cdiv(rows, 64)))
        part = torch.empty(nparts, Cc, device=g.device, dtype=torch.float32)
        lib.bias_grad(g, rows, Cc, ld, part, nparts)
        out = torch.empty(Cc, device=g.device, dtype=torch.float32)
        lib.colsum(part, nparts, Cc, out)
        return out

    # ------------------------------------------------------------------ weight / data gradients of a convolution
    def _wgrad(self, g, x, *, B, L_in, L_out, cin, cout, taps, stride, pad, dil=1, pro_h=0, x_stats=None, lg=None):
        """dW [cout][cin][taps] = sum_{b,t} g[b,t,:] (x) x[b, t*stride + j*dil - pad, :]: w2s_wgrad per (<= 128 input channels) x
        (128 / 64 / 32 / 16 output channels) block -- the shapes its kernels are instantiated for -- then the deterministic slab sum."""
        dev = g.device
        dW = torch.empty(cout, cin, taps, device=dev, dtype=torch.float32)
        for ci0, ck in _chunks(cin):
            co0 = 0
            while co0 < cout:
                cp = next(c for c in (128, 64, 32, 16) if c <= cout - co0)
                kw = dict(g=g if co0 == 0 else g[..., co0:], x=x if ci0 == 0 else x[..., ci0:], B=B, L_in=L_in, L_out=L_out, cin=ck, cout=cp,
                          taps=taps, stride=stride, pad=pad, dil=dil, ldg=cout, ldx=cin, split_precision=True, pro_h=pro_h, x_stats=x_stats)
                if lg is not None:   # (one output block: _norm_act allows the lazy gradient only then)
                    kw.update(pro_g=lib.PRO_AFFINE_BWD + lg.act, g2=lg.y, g_stats=lg.ss, g_bstats=lg.cd)
                gy = lib.wgrad_grid_y(ck, cp, taps, dil)
                gx = max(1, min(_cdiv(B * L_out, 256), max(1, lib.wgrad_max_blocks(slab=None, nslab=0, **kw) // gy)))
                nslab = gx * lib.wgrad_slabs_per_block_of(slab=None, nslab=0, **kw)
                slab = torch.empty(nslab * cp * ck * taps, device=dev, dtype=torch.float32)
                lib.wgrad(slab=slab, nslab=nslab, **kw)
                piece = dW if (cp == cout and ck == cin) else torch.empty(cp, ck, taps, device=dev, dtype=torch.float32)
                lib.wgrad_reduce(slab, nslab, piece, cp, ck, taps, dil, False, 0)
                if piece is not dW:
                    dW[co0:co0 + cp, ci0:ci0 + ck] = piece
                co0 += cp
        return dW

    def _dgrad(self, gy, w, *, B, L_x, L_out, stride, pad, dil, acc=None, lg=None, part_for=None):
        """gradient of the conv input: gy [B, L_out, cout], w [cout, cin, k] -> gx [B, L_x, cin] (w2s_conv_forward on the transposed weights:
        flipped taps for stride 1, W2S_MODE_UP2 for the k=3 / stride-2 conv, the even rows of gx for the 1x1 / stride-2 residual conv)."""
        cout, cin, k = w.shape
        dev = gy.device
        if cin % 16:
            raise NotImplementedError(f'data gradient towards {cin} channels: multiples of 16')
        wb = w.permute(1, 2, 0).contiguous()   # [cin][k][cout]
        chunks = _chunks(cout)
        nchunk = len(chunks)
        if stride == 2 and k == 1:
            if pad != 0:
                raise NotImplementedError('1x1 / stride-2 conv with padding')
            gx = acc if acc is not None else torch.zeros(B, L_x, cin, device=dev, dtype=torch.float32)
            samples = [(gy, gx, B)] if L_x == 2 * L_out else [(gy[b], gx[b], 1) for b in range(B)]
            for gs, xs, nb in samples:   # row t of the gradient lands on row 2t of gx: a GEMM whose output row stride is two rows of gx
                for q, (c0, ck) in enumerate(chunks):
                    wq = wb if nchunk == 1 else wb[:, :, c0:c0 + ck].contiguous()
                    lib.conv_forward(lib.conv_args(x=gs if q == 0 else gs[..., c0:], w=wq, y=xs, B=1, L_in=nb * L_out, L_out=nb * L_out, cin=ck, cout=cin,
                                                   taps=1, stride=1, pad=0, ldx=cout, ldy=2 * cin, accumulate=q > 0 or acc is not None))
            return gx
        gx = acc if acc is not None else torch.empty(B, L_x, cin, device=dev, dtype=torch.float32)
        pk = dict(pro=lib.PRO_AFFINE_BWD + lg.act, x2=lg.y, pro_stats=lg.ss, pro_bstats=lg.cd) if lg is not None else {}
        for q, (c0, ck) in enumerate(chunks):
            wq = wb if nchunk == 1 else wb[:, :, c0:c0 + ck].contiguous()
            xq = gy if q == 0 else gy[..., c0:]
            if stride == 1:
                mode = lib.MODE_DILATED if k == 7 else lib.MODE_CONTIG
                a = lib.conv_args(x=xq, w=wq, y=gx, B=B, L_in=L_out, L_out=L_x, cin=ck, cout=cin, taps=k, stride=1, pad=(k - 1) * dil - pad, dil=dil,
                                  flip=1, mode=mode, ldx=cout, accumulate=q > 0 or acc is not None, **pk)
            elif stride == 2 and k == 3 and dil == 1:
                a = lib.conv_args(x=xq, w=wq, y=gx, B=B, L_in=L_out, L_out=L_x, cin=ck, cout=cin, taps=3, stride=2, pad=pad, mode=lib.MODE_UP2,
                                  ldx=cout, accumulate=q > 0 or acc is not None, **pk)
            else:
                raise NotImplementedError(f'data gradient of kernel_size={k}, stride={stride}, dilation={dil}')
            if part_for is not None and nchunk == 1 and acc is None:
                # gx is the gradient of a _Lazy tensor act(y' scale + shift): the reduction pass of THAT layer's norm backward (sums of
                # ga = gx act'(z) and ga y') rides in this launch's epilogue (W2S_EPI_AFFINE_PART) instead of re-reading gx and y'
                a.epi, a.ld_aux = lib.EPI_AFFINE_PART + part_for.act, cin
                a.aux, a.aux_stats = lib._f(part_for.y), lib._f(part_for.ss)
                nt = _cdiv(L_x, lib.conv_tile_of(a))
                part = torch.empty(B, nt, 2, cin, device=dev, dtype=torch.float32)
                lib.set_part(a, part)
                gx._w2s_part = (part, nt)
            lib.conv_forward(a)
        return gx

    # ------------------------------------------------------------------ convolution + normalisation + activation
    def _col_stats(self, y, B, L, Cc, eps, kind):
        """per-(sample, channel) statistics of y [B, L, C] in a pass of their own (layers whose contraction is split over several launches)"""
        tile = 1024
        nt = _cdiv(L, tile)
        part = torch.empty(B, nt, 2, Cc, device=y.device, dtype=torch.float32)
        lib.norm_act_bwd_part(y, Cc, y, Cc, None, 0, None, None, L, B, Cc, 0, 0.0, tile, part)   # (sum y, sum y*y): g = y, no norm, linear
        stats = torch.empty(B, Cc, 2, device=y.device, dtype=torch.float32)
        lib.stats_finalize(part, B, nt, Cc, L, eps, kind, stats)
        return stats

    def _conv(self, x, wp, bp, L_out, *, stride, pad, dil, want_stats=None, eps=1e-5, x_needs_grad=True):
        """x [B, L_in, Cin] -> y [B, L_out, Cout]; left padding `pad` (zeros), taps j read x[t*stride + j*dil - pad].
        wp / bp: the weight / bias PARAMETERS ([cout, cin, k] / [cout] or None).  want_stats: None | 0 (mean, rstd) | 1 (E[y], E[y^2]) -> [B, Cout, 2]."""
        w = wp.detach()
        bias = bp.detach() if bp is not None else None
        pro, ss = lib.PRO_NONE, None
        x_in = x
        if isinstance(x, _Lazy):
            if len(_chunks(x.shape[2])) == 1:   # the previous layer's norm + activation on load (its output was never written)
                pro, ss, x = lib.PRO_AFFINE + x.act, x.ss, x.y
            else:
                x = x_in = self._mat(x)
        B, L_in, cin = x.shape
        cout, cin_w, k = w.shape
        if cout % 16:
            raise NotImplementedError(f'{cout} output channels: the generic kernels produce multiples of 16')
        dev = x.device
        if cin == 1 and k <= 3 and dil == 1 and cout <= 256:
            return self._conv1(x, wp, bp, L_out, stride=stride, pad=pad, want_stats=want_stats, eps=eps)
        if cin == 1:   # zero-pad the one-channel input to the narrowest tile the matrix path takes (once per input: conv1 and the residual conv share it)
            if self._x16 is None or self._x16[0] is not x:
                x16 = torch.zeros(B, L_in, 16, device=dev, dtype=torch.float32)
                x16[..., 0] = x[..., 0]
                self._x16 = (x, x16)
            x16 = self._x16[1]
            w16 = torch.zeros(cout, 16, k, device=dev, dtype=torch.float32)
            w16[:, 0] = w[:, 0]
            x, w, cin = x16, w16, 16
        chunks = _chunks(cin)
        nchunk = len(chunks)
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
        fused_stats = want_stats is not None and nchunk == 1
        for q, (c0, ck) in enumerate(chunks):
            wq = w[:, c0:c0 + ck, :].permute(0, 2, 1).contiguous()             # [cout][k][ck]
            xq = x if q == 0 else x[..., c0:]                                   # pointer offset; row stride stays the full width
            a = lib.conv_args(x=xq, w=wq, y=y, B=B, L_in=L_in, L_out=L_out, cin=ck, cout=cout, taps=k, stride=stride, pad=pad, dil=dil, mode=mode,
                              ldx=cin, epi=lib.EPI_BIAS if (bias is not None and q == 0) else (lib.EPI_STATS if fused_stats else lib.EPI_PLAIN),
                              bias=bias if q == 0 else None, accumulate=q > 0, pro=pro, pro_stats=ss)
            if fused_stats:
                nt = _cdiv(L_out, lib.conv_tile_of(a))
                part = torch.empty(B, nt, 2, cout, device=dev, dtype=torch.float32)
                lib.set_part(a, part)
                lib.conv_forward(a)
                stats = torch.empty(B, cout, 2, device=dev, dtype=torch.float32)
                lib.stats_finalize(part, B, nt, cout, L_out, eps, want_stats, stats)
            else:
                lib.conv_forward(a)
        if want_stats is not None and not fused_stats:
            stats = self._col_stats(y, B, L_out, cout, eps, want_stats)
        if self.grad:
            xs = x   # the (padded) operand of the weight gradient

            def bw(gy, acc=None):
                lg = gy if isinstance(gy, _LazyG) else None   # the norm + activation backward still to be applied (on load, below)
                if lg is not None:
                    gy = lg.g
                if bp is not None:
                    self._pgrad(bp, self._rowsum(gy, B * L_out, cout, cout))
                dW = self._wgrad(gy, xs, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, taps=k, stride=stride, pad=pad, dil=dil, pro_h=pro, x_stats=ss, lg=lg)
                self._pgrad(wp, dW[:, :cin_w] if cin_w != cin else dW)
                if not x_needs_grad or cin_w == 1:
                    return (None,)
                return (self._dgrad(gy, wp.detach(), B=B, L_x=L_in, L_out=L_out, stride=stride, pad=pad, dil=dil, acc=acc, lg=lg,
                                    part_for=x_in if (isinstance(x_in, _Lazy) and pro != lib.PRO_NONE and x_in.scale is not None) else None),)
            self._rec(y, (x_in,), bw, acc_ok=x_needs_grad and cin_w != 1)
        return y, stats

    def _conv1(self, x, wp, bp, L_out, *, stride, pad, want_stats, eps):
        """Convolution of the one-channel input [B, L_in, 1] (block 0's conv1 and residual conv) on the vector ALU: csrc/generic.hip conv1_*."""
        B, L_in, _ = x.shape
        cout, _, k = wp.shape
        dev = x.device
        x1 = x.reshape(B, L_in)
        y = torch.empty(B, L_out, cout, device=dev, dtype=torch.float32)
        nt = _cdiv(L_out, lib.C1_TILE)
        part = torch.empty(B, nt, 2, cout, device=dev, dtype=torch.float32) if want_stats is not None else None
        lib.conv1_fwd(x1, wp.detach().reshape(cout, k), bp.detach() if bp is not None else None, y, part, B, L_in, L_out, cout, k, stride, pad)
        stats = None
        if want_stats is not None:
            stats = torch.empty(B, cout, 2, device=dev, dtype=torch.float32)
            lib.stats_finalize(part, B, nt, cout, L_out, eps, want_stats, stats)

        def bw(gy, acc=None):
            lg = gy if isinstance(gy, _LazyG) else None
            g = lg.g if lg is not None else gy
            if bp is not None:
                self._pgrad(bp, self._rowsum(g, B * L_out, cout, cout))   # (a conv bias means no norm behind it: the gradient is materialised)
            nparts = lib.conv1_wgrad_parts(B, L_out)
            wpart = torch.empty(nparts, cout * k, device=dev, dtype=torch.float32)
            lib.conv1_wgrad(g, lg.y if lg is not None else None, lg.ss if lg is not None else None, lg.cd if lg is not None else None, x1, wpart,
                            B, L_in, L_out, cout, k, stride, pad, lg.act if lg is not None else 0)
            dW = torch.empty(cout * k, device=dev, dtype=torch.float32)
            lib.colsum(wpart, nparts, cout * k, dW)
            self._pgrad(wp, dW)
            return (None,)
        self._rec(y, (x,), bw)
        return y, stats

    def _act(self, x, name: str, slope: float = 0.01):
        """activation of a contiguous [..., C] tensor (in place unless a gradient is wanted)"""
        code = _act_code(name)
        if not code:
            return x
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        if not self.grad:
            lib.affine_act(x, Cc, None, None, 0, x, Cc, 1, rows, Cc, code, slope)
            return x
        out = torch.empty_like(x)
        lib.affine_act(x, Cc, None, None, 0, out, Cc, 1, rows, Cc, code, slope)

        def bw(g):
            gx = torch.empty_like(g)
            lib.norm_act_bwd_apply(g, Cc, x, Cc, None, 0, None, None, None, 0, gx, Cc, rows, rows, Cc, code, slope)
            return (gx,)
        self._rec(out, (x,), bw)
        return out

    def _stat_norm_bwd(self, kind, g, y, mr, gamma, beta, act, B, L, Cc, G, lazy_ss=None):
        """backward of (statistics-based norm -> affine -> activation) over y [B, L, C]: per-(sample, channel) means of ga and ga * xh
        (two launches), the norm's own averaging + (A, B, Cx) + the affine parameters' gradients (w2s_norm_bwd_coef), gy = A ga + B + Cx xh.
        Returns gy, dgamma, dbeta (None without affine parameters)."""
        dev = y.device
        per_sample = kind in (0, 3)
        pre = getattr(g, '_w2s_part', None)   # the sums came with g, out of the data-gradient launch that produced it (sums of ga and ga y)
        if pre is not None:
            part, nt = pre
        else:
            tile = 1024
            nt = _cdiv(L, tile)
            part = torch.empty(B, nt, 2, Cc, device=dev, dtype=torch.float32)
            lib.norm_act_bwd_part(g, Cc, y, Cc, mr, 2 * Cc if per_sample else 0, gamma, beta, L, B, Cc, act, 0.01, tile, part)
        means = torch.empty(B, Cc, 2, device=dev, dtype=torch.float32)
        lib.stats_finalize(part, B, nt, Cc, L, 0.0, 1, means)
        coef = torch.empty(B if per_sample else 1, 3, Cc, device=dev, dtype=torch.float32)
        dgam = torch.empty(Cc, device=dev, dtype=torch.float32) if gamma is not None else None
        dbet = torch.empty(Cc, device=dev, dtype=torch.float32) if gamma is not None else None
        cd = torch.empty(B, Cc, 2, device=dev, dtype=torch.float32) if lazy_ss is not None else None
        lib.norm_bwd_coef(kind, means, mr, B, Cc, G, gamma, beta, float(L), coef, dgam, dbet, cd, y_sums=pre is not None)
        if lazy_ss is not None:   # the convolution's data / weight gradient launches apply it on load: gy is never written
            return _LazyG(g, y, lazy_ss, cd, act), dgam, dbet
        gy = torch.empty_like(y)
        lib.norm_act_bwd_apply(g, Cc, y, Cc, mr, 2 * Cc if per_sample else 0, gamma, beta, coef, 3 * Cc if per_sample else 0, gy, Cc, L, B * L, Cc, act, 0.01)
        return gy, dgam, dbet

    def _norm_act(self, layer, y, stats, act_name, lazy=False, lazy_g=False):
        """norm -> activation of one ConvLayer1D output y [B, L, C] (blocks.py:183-185); in place unless a gradient is wanted.
        lazy (the consumer is a convolution over <= 128 channels): per-channel norms return a _Lazy instead of writing the result."""
        B, L, Cc = y.shape
        lazy = lazy and len(_chunks(Cc)) == 1
        rows = B * L
        act = _act_code(act_name)
        norm = layer.norm
        kind = layer.norm_name
        dev = y.device
        lazy = lazy and kind in (None, 'instance', 'batch', 'group')
        # lazy_g (the producing conv has one <= 128-channel output block and no bias): the BACKWARD of norm + activation is applied on load too
        lazy_g = lazy_g and self.grad and kind in ('instance', 'batch', 'group') and len(_chunks(Cc)) == 1
        out = None if lazy else (torch.empty_like(y) if self.grad else y)
        bw = None
        if kind is None or kind == 'weight':
            if lazy:
                ss = torch.zeros(B, Cc, 2, device=dev, dtype=torch.float32)
                ss[..., 0] = 1.0
                out = _Lazy(y, ss, None, None, act)
            else:
                lib.affine_act(y, Cc, None, None, 0, out, Cc, L, rows, Cc, act, 0.01)

            def bw(g):
                gy = torch.empty_like(g)
                lib.norm_act_bwd_apply(g, Cc, y, Cc, None, 0, None, None, None, 0, gy, Cc, L, rows, Cc, act, 0.01)
                return (gy,)
        elif kind in ('instance', 'batch', 'group'):
            # statistics -> (scale, shift) for the forward and (mean, rstd) for the backward, in one launch (w2s_norm_fold)
            gam = bet = rm = rv = None
            G, mom, pnorm = 1, 0.0, None
            if kind == 'instance':   # stats = (mean, rstd) per (b, c)
                if getattr(norm, 'affine', False):
                    raise NotImplementedError('InstanceNorm1d(affine=True)')
                code, eps = 0, norm.eps
            elif kind == 'batch':    # stats = (E[y], E[y^2]) per (b, c) in training, unused in eval mode
                pnorm = norm
                gam, bet, eps = norm.weight.detach(), norm.bias.detach(), norm.eps
                if self.training and norm.training:
                    code = 1
                    if norm.track_running_stats:
                        rm, rv = norm.running_mean, norm.running_var
                        mom = norm.momentum if norm.momentum is not None else 1.0 / float(norm.num_batches_tracked + 1)
                        norm.num_batches_tracked += 1
                else:
                    code, rm, rv = 2, norm.running_mean, norm.running_var
            else:                    # GroupNorm: stats = (E[y], E[y^2]) per (b, c), pooled over the channels of a group
                pnorm = norm.norm
                gam, bet, eps, G = pnorm.weight.detach(), pnorm.bias.detach(), pnorm.eps, pnorm.num_groups
                code = 3
            nset = B if code in (0, 3) else 1
            scale = torch.empty(nset, Cc, device=dev, dtype=torch.float32)
            shift = torch.empty(nset, Cc, device=dev, dtype=torch.float32)
            mr = torch.empty(nset, Cc, 2, device=dev, dtype=torch.float32)
            ss = torch.empty(B, Cc, 2, device=dev, dtype=torch.float32) if (lazy or lazy_g) else None
            lib.norm_fold(code, stats, B, Cc, G, gam, bet, rm, rv, eps, mom, float(B * L), scale, shift, mr, ss)
            if lazy:
                out = _Lazy(y, ss, scale, shift, act)
            else:
                lib.affine_act(y, Cc, scale, shift, Cc if nset > 1 else 0, out, Cc, L, rows, Cc, act, 0.01)

            def bw(g):
                gy, dgam, dbet = self._stat_norm_bwd(code, g, y, mr, gam, bet, act, B, L, Cc, G, lazy_ss=ss if lazy_g else None)
                if pnorm is not None:
                    self._pgrad(pnorm.weight, dgam)
                    self._pgrad(pnorm.bias, dbet)
                return (gy,)
        elif kind in ('layer', 'rms'):
            rms = kind == 'rms'
            gam = norm.weight.detach().reshape(Cc)
            bet = None if rms else norm.bias.detach().reshape(Cc)
            lib.rownorm_fwd(y, Cc, gam, bet, out, Cc, rows, Cc, norm.eps, rms, act, 0.01)

            def bw(g):
                return (self._rownorm_bwd(g, y, norm.weight, None if rms else norm.bias, rows, Cc, norm.eps, rms, act),)
        else:
            raise ValueError(f'Normalisation with name={kind} unknown.')
        self._rec(out, (y,), bw)
        return out

    def _rownorm_bwd(self, g, x, wp, bp, rows, Cc, eps, rms, act):
        nb = lib.rownorm_bwd_blocks(rows)
        part = torch.empty(nb, 2, Cc, device=g.device, dtype=torch.float32)
        gx = torch.empty_like(g)
        lib.rownorm_bwd(g, Cc, x, Cc, wp.detach().reshape(Cc), bp.detach().reshape(Cc) if bp is not None else None, gx, Cc, part, rows, Cc, eps, rms, act, 0.01)
        dg = torch.empty(Cc, device=g.device, dtype=torch.float32)
        lib.colsum(part, nb, Cc, dg, ld=2 * Cc)
        self._pgrad(wp, dg)
        if bp is not None:
            db = torch.empty(Cc, device=g.device, dtype=torch.float32)
            lib.colsum(part.view(-1)[Cc:], nb, Cc, db, ld=2 * Cc)
            self._pgrad(bp, db)
        return gx

    def _rownorm(self, t, wp, bp, eps, rms=False, act=0):
        """LayerNorm / RMS norm over the last dimension of t [rows, C] (out of place)"""
        rows, Cc = t.shape
        o = torch.empty_like(t)
        lib.rownorm_fwd(t, Cc, wp.detach().reshape(Cc), bp.detach().reshape(Cc) if bp is not None else None, o, Cc, rows, Cc, eps, rms, act, 0.01)
        self._rec(o, (t,), lambda g: (self._rownorm_bwd(g, t, wp, bp, rows, Cc, eps, rms, act),))
        return o

    def _dropout_(self, x, p):
        if not (self.training and p > 0.0):
            return x
        self._site += 1
        seed = ((self.seed & 0xFFFFFFFF) << 16) ^ (self._site * 0x9E3779B1 & 0xFFFFFFFF)
        out = torch.empty_like(x) if self.grad else x
        lib.eltwise(lib.ELT_DROP, x, None, out, x.numel(), p, seed)

        def bw(g):
            gx = torch.empty_like(g)
            lib.eltwise(lib.ELT_DROP, g, None, gx, g.numel(), p, seed)   # the same (seed, element) mask
            return (gx,)
        self._rec(out, (x,), bw)
        return out

    def _sum(self, a, b):
        """a + b (in place into a unless a gradient is wanted)"""
        out = torch.empty_like(a) if self.grad else a
        lib.eltwise(lib.ELT_ADD, a, b, out, a.numel())
        self._rec(out, (a, b), lambda g: (g, g), share='all')
        return out

    def conv_layer(self, layer, x, x_needs_grad=True, lazy_out=False):
        """ConvLayer1D.forward on channels-last x [B, L, Cin] (blocks.py:173-186).  x may be the _Lazy output of the layer before;
        lazy_out: the caller feeds the result to another conv_layer (and nothing else)."""
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
        if conv.bias is not None and want is not None:
            raise NotImplementedError('a convolution bias in front of a statistics-based norm')
        y, stats = self._conv(x, conv.weight, conv.bias, L_out, stride=stride, pad=pad, dil=dil, want_stats=want, eps=eps, x_needs_grad=x_needs_grad)
        drop = self.training and layer.dropout_p > 0.0
        out = self._norm_act(layer, y, stats, layer.activation_name, lazy=lazy_out and not drop, lazy_g=conv.bias is None)
        return self._dropout_(out, layer.dropout_p) if drop else out

    def conv_block(self, block, x, x_needs_grad=True):
        """ConvBlock1D.forward (blocks.py:57-71)."""
        out = self.conv_layer(block.conv3, self.conv_layer(block.conv2, self.conv_layer(block.conv1, x, x_needs_grad, lazy_out=True), lazy_out=True),
                              lazy_out=block.use_residual)
        if block.use_residual:
            r, _ = self._conv(x, block.downsample.weight, None, out.shape[1], stride=2, pad=0, dil=1, x_needs_grad=x_needs_grad)
            if isinstance(out, _Lazy):
                return self._join(out, r, block.activation_name)
            out = self._sum(out, r)
        return self._act(out, block.activation_name)

    def _join(self, lz, r, act_name):
        """act2(act3(y3 * scale + shift) + r) in one pass (w2s_affine_act_join): conv3's activated output and the sum are never written"""
        B, L, Cc = lz.y.shape
        rows = B * L
        act2 = _act_code(act_name)
        stride = Cc if (lz.scale is not None and lz.scale.shape[0] > 1) else 0
        out = torch.empty_like(r) if self.grad else r
        lib.affine_act_join(lz.y, Cc, lz.scale, lz.shift, stride, r, Cc, out, Cc, L, rows, Cc, lz.act, act2)

        def bw(g):
            gs = torch.empty_like(g)
            lib.affine_act_join_bwd(g, Cc, lz.y, Cc, lz.scale, lz.shift, stride, r, Cc, gs, Cc, L, rows, Cc, lz.act, act2)
            return (gs, gs)
        self._rec(out, (lz, r), bw, share='all')
        return out

    def dilated_block(self, block, x):
        """DilatedConvBlock.forward (blocks.py:115-126) on [B, S, F]."""
        out = x
        n = len(block.conv_layers)
        for i, layer in enumerate(block.conv_layers):
            out = self.conv_layer(layer, out, lazy_out=i + 1 < n)
        out = self._dropout_(out, block.dropout.p)
        if out is x:
            out = x.clone()
        return self._act(self._sum(out, x), block.activation_name)

    # ------------------------------------------------------------------ dense layers / GEMMs
    def linear(self, x_rows, wp, bp, act_name='linear'):
        """y[rows, cout] = x[rows, cin] @ W^T + b, then the activation (any cin that is a power of two <= 128 or a multiple of 128).
        wp / bp: the weight [cout, cin] / bias [cout] parameters (or tensors, when no gradient is recorded)."""
        rows, cin = x_rows.shape
        cout = wp.shape[0]
        if cout % 16:
            raise NotImplementedError(f'{cout} output features: multiples of 16')
        y = torch.empty(rows, cout, device=x_rows.device, dtype=torch.float32)
        w = wp.detach()
        bias = bp.detach() if bp is not None else None
        for q, (c0, ck) in enumerate(_chunks(cin)):
            wq = w[:, c0:c0 + ck].contiguous()
            xq = x_rows if q == 0 else x_rows[:, c0:]
            lib.conv_forward(lib.conv_args(x=xq, w=wq, y=y, B=1, L_in=rows, L_out=rows, cin=ck, cout=cout, taps=1, stride=1, pad=0, ldx=cin,
                                           epi=lib.EPI_BIAS if (bias is not None and q == 0) else lib.EPI_PLAIN,
                                           bias=bias if q == 0 else None, accumulate=q > 0))

        def bw(gy):
            if bp is not None:
                self._pgrad(bp, self._rowsum(gy, rows, cout, cout))
            self._pgrad(wp, self._wgrad(gy, x_rows, B=1, L_in=rows, L_out=rows, cin=cin, cout=cout, taps=1, stride=1, pad=0))
            return (self._dgrad(gy.view(1, rows, cout), w.view(cout, cin, 1), B=1, L_x=rows, L_out=rows, stride=1, pad=0, dil=1).view(rows, cin),)
        self._rec(y, (x_rows,), bw)
        return self._act(y, act_name)

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
        for i, block in enumerate(enc.cnn):
            y = self.conv_block(block, y, x_needs_grad=i > 0)
        # [B(*S), 4(*S), C] -> [B, S, 4C]: feature index = t_local * C + c, which in channels-last layout is the memory order
        feat = self._view(y, B * S, enc.epoch_dim)
        z = self.linear(feat, enc.linear.weight, enc.linear.bias, enc.activation_name)
        if isinstance(enc.output_norm, nn.LayerNorm):
            z = self._rownorm(z, enc.output_norm.weight, enc.output_norm.bias, enc.output_norm.eps)
        return self._view(z, B, S, enc.feature_dim)

    def signal_encoders(self, mod, x: dict) -> dict:
        """SignalEncoders.forward (wav2sleep.py:146-161)."""
        z = {}
        for name, x_BT in x.items():
            if name not in mod.signal_map:
                raise ValueError(f'Unknown signal {name}')
            mask_B = torch.isinf(x_BT[:, 0])
            xs = torch.where(torch.isinf(x_BT), torch.zeros_like(x_BT), x_BT)
            z_enc = self.signal_encoder(mod.get_encoder(name), xs)
            z_BSF = torch.where(mask_B[:, None, None], float('-inf'), z_enc)
            self._rec(z_BSF, (z_enc,), lambda g, mask_B=mask_B: (g.masked_fill(mask_B[:, None, None], 0.0),))
            if mod.embed_signals:
                idx = mod.sig_to_embedding_idx[name]
                z_in = z_BSF
                z_BSF = z_in + mod.embedder.weight.detach()[idx][None, None, :]

                def bw(g, idx=idx, mask_B=mask_B):
                    Bq, Sq, Fq = g.shape
                    gm = g.masked_fill(mask_B[:, None, None], 0.0)   # -inf rows stay -inf whatever the embedding is
                    ge = torch.zeros_like(mod.embedder.weight)
                    ge[idx] = self._rowsum(gm, Bq * Sq, Fq, Fq)
                    self._pgrad(mod.embedder.weight, ge)
                    return (g,)
                self._rec(z_BSF, (z_in,), bw, share='view')   # (the gradient passes through unchanged)
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

        def bw_tokens(gX):
            g3 = gX.view(N, D, F_)
            greg = torch.empty(F_, R1, device=dev, dtype=torch.float32)
            for r in range(R1):   # CLS / register tokens are broadcast over the N sentences: their gradient is the column sum of slot r
                col = torch.empty(F_, device=dev, dtype=torch.float32)
                lib.colsum(gX.view(-1)[r * F_:], N, F_, col, ld=D * F_)
                greg[:, r] = col
            self._pgrad(mod.register_tokens, greg)
            outs = []
            for m in range(len(signals)):
                gz = g3[:, R1 + m, :].reshape(B, S, F_)
                outs.append(gz.masked_fill(pads[R1 + m][:, None, None], 0.0))
            return outs
        self._rec(X, [z_dict[sig] for sig in signals], bw_tokens)
        p_attn = 0.0
        for layer in mod.transformer_encoder.layers:
            sa = layer.self_attn
            if self.training and sa.dropout > 0.0:
                p_attn = float(sa.dropout)

            def attn(h, sa=sa, layer=layer):
                qkv = self.linear(h, sa.in_proj_weight, sa.in_proj_bias)
                ao = torch.empty(N * D, F_, device=dev, dtype=torch.float32)
                seed = 0
                if p_attn > 0.0:
                    self._site += 1
                    seed = ((self.seed & 0xFFFFFFFF) << 16) ^ (self._site * 0x9E3779B1 & 0xFFFFFFFF)
                lib.attn_generic_fwd(qkv, keypad, ao, N, D, H, hd, p_attn, seed)

                def bw(g):
                    gqkv = torch.empty_like(qkv)
                    lib.attn_generic_bwd(qkv, keypad, g, gqkv, N, D, H, hd, p_attn, seed)
                    return (gqkv,)
                self._rec(ao, (qkv,), bw)
                return self._dropout_(self.linear(ao, sa.out_proj.weight, sa.out_proj.bias), layer.dropout1.p)

            def ff(h, layer=layer):
                a1 = self._dropout_(self.linear(h, layer.linear1.weight, layer.linear1.bias, mod.activation_name), layer.dropout.p)
                return self._dropout_(self.linear(a1, layer.linear2.weight, layer.linear2.bias), layer.dropout2.p)

            def ln(norm, t):
                return self._rownorm(t, norm.weight, norm.bias, norm.eps)

            def add(a_, b_):
                o = torch.empty_like(a_)
                lib.eltwise(lib.ELT_ADD, a_, b_, o, a_.numel())
                self._rec(o, (a_, b_), lambda g: (g, g), share='all')
                return o

            if layer.norm_first:
                X = add(X, attn(ln(layer.norm1, X)))
                X = add(X, ff(ln(layer.norm2, X)))
            else:
                X = ln(layer.norm1, add(X, attn(X)))
                X = ln(layer.norm2, add(X, ff(X)))
        cls = X.view(N, D, F_)[:, 0, :].reshape(B, S, F_).contiguous()

        def bw_cls(g):
            gX = torch.zeros(N, D, F_, device=dev, dtype=torch.float32)
            gX[:, 0, :] = g.reshape(N, F_)
            return (gX.view(N * D, F_),)
        self._rec(cls, (X,), bw_cls)
        return cls

    def sequence_cnn(self, mod, x_BSF):
        """SequenceCNN.forward (wav2sleep.py:379-390); channels-last [B, S, F] is already the layout the blocks run on."""
        x = x_BSF if (x_BSF.is_contiguous() and x_BSF.dtype == torch.float32) else x_BSF.contiguous().float()
        for block in mod.dilated_convs:
            x = self.dilated_block(block, x)
        return x

    def classifier(self, lin: nn.Linear, x_BSF):
        B, S, F_ = x_BSF.shape
        nc = lin.out_features
        rows = B * S
        logits = torch.empty(B, S, nc, device=x_BSF.device, dtype=torch.float32)
        xc = x_BSF if x_BSF.is_contiguous() else x_BSF.contiguous()
        lib.head_fwd(xc, F_, lin.weight.detach(), lin.bias.detach(), logits, rows, F_, nc, False)

        def bw(g):
            gl = g.reshape(rows, nc).contiguous()
            gx = torch.empty(B, S, F_, device=g.device, dtype=torch.float32)
            nparts = max(1, min(1024, _cdiv(rows, 16)))
            part = torch.empty(nparts, nc * F_ + nc, device=g.device, dtype=torch.float32)
            lib.head_bwd(xc, F_, lin.weight.detach(), gl, gx, F_, part, nparts, rows, F_, nc, False)
            dw = torch.empty(nc * F_, device=g.device, dtype=torch.float32)
            db = torch.empty(nc, device=g.device, dtype=torch.float32)
            lib.colsum(part, nparts, nc * F_, dw, ld=nc * F_ + nc)
            lib.colsum(part.view(-1)[nc * F_:], nparts, nc, db, ld=nc * F_ + nc)
            self._pgrad(lin.weight, dw)
            self._pgrad(lin.bias, db)
            return (gx,)
        self._rec(logits, (x_BSF,), bw)
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
        for i, block in enumerate(model.conv_block.model):
            y = self.conv_block(block, y, x_needs_grad=i > 0)  # [B, 4800, 256]
        feat = self._view(y, B * 1200, 1024)                   # transpose(-1, -2).reshape(-1, 1200, 1024): memory order in channels-last
        z = self._view(self.linear(feat, model.dense.linear.weight, model.dense.linear.bias, model.dense.activation_name), B, 1200, model.feature_dim)
        for block in model.dilated_convs:
            z = self.dilated_block(block, z)
        return self.classifier(model.classifier, z)


class _GenericFn(torch.autograd.Function):
    """One autograd node around a generic-path forward: logits = run(walker); the backward replays the walker's tape (HIP kernels only) and
    hands torch the parameter gradients.  The inputs (signals) get no gradient, as in training."""

    @staticmethod
    def forward(ctx, run, training, seed, device, *params):
        with torch.cuda.device(device):
            gf = GenericForward(training=training, seed=seed, grad=True)
            out = run(gf)
        ctx.gf, ctx.out, ctx.params, ctx.device = gf, out, params, device
        return out.detach().clone()   # the tape keys tensors by identity: keep the walker's own output out of the caller's hands

    @staticmethod
    def backward(ctx, g):
        if ctx.gf is None:
            raise RuntimeError('the generic path keeps ONE backward per forward (its tape is consumed): run the forward again')
        with torch.cuda.device(ctx.device):
            pg = ctx.gf.backward(ctx.out, g)
        ctx.gf = ctx.out = None
        return (None, None, None, None) + tuple(pg.get(p) if p.requires_grad else None for p in ctx.params)


def differentiable(model: nn.Module, run, training: bool, seed: int = 0) -> torch.Tensor:
    """run(walker) -> logits as ONE autograd node over model.parameters() (ppgnet.py / wav2sleep.py call this when a gradient is wanted)."""
    params = [p for p in model.parameters()]
    return _GenericFn.apply(run, training, seed, params[0].device, *params)
