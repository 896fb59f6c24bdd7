"""Progressive GPU check of libw2s_hip.so against CPU torch / the oracle.  Prints a table; never stops early.
Run on the GPU box:  python tests/gpu_check.py [stage ...]"""
import os, sys, math, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from wav2sleep_amd import lib
from oracle import wav2sleep_oracle as O

dev = 'cuda'
torch.manual_seed(0)
RES = []

def report(name, got, want, tol=1e-4):
    got = got.detach().float().cpu(); want = want.detach().float().cpu()
    err = (got - want).abs().max().item(); scale = want.abs().max().item() + 1e-12
    ok = err <= tol * max(1.0, scale)
    RES.append((name, ok))
    print(f'{"OK  " if ok else "FAIL"} {name:55s} maxerr {err:.3e} (scale {scale:.3e})', flush=True)

def cl(x):  # [B,C,L] -> channels-last [B,L,C]
    return x.transpose(1, 2).contiguous()

def pack_fwd(w):  # torch [o][c][j] -> [o][j][c]
    return w.permute(0, 2, 1).contiguous()

def run(fn):
    try:
        fn()
    except Exception as e:
        traceback.print_exc(); RES.append((fn.__name__, False)); print('FAIL', fn.__name__, 'exception', e, flush=True)

def t_conv_plain():
    for (cin, cout, taps, stride, L) in [(16, 16, 3, 1, 700), (16, 32, 3, 1, 512), (32, 32, 3, 2, 1000), (64, 64, 3, 1, 300), (64, 128, 3, 1, 260),
                                         (128, 128, 3, 1, 200), (128, 128, 3, 2, 256), (16, 32, 1, 2, 512), (128, 128, 1, 2, 128), (128, 384, 1, 1, 333), (128, 64, 1, 1, 100)]:
        B = 2
        x = torch.randn(B, cin, L); w = torch.randn(cout, cin, taps) / math.sqrt(cin * taps)
        pad = 1 if taps == 3 else 0
        want = F.conv1d(x, w, stride=stride, padding=pad)
        Lo = want.shape[-1]
        xd, wd = cl(x).to(dev), pack_fwd(w).to(dev)
        y = torch.zeros(B, Lo, cout, device=dev)
        lib.conv_forward(lib.conv_args(x=xd, w=wd, y=y, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride, pad=pad))
        report(f'conv {cin}->{cout} k{taps} s{stride} L{L}', y, cl(want))

def t_conv_stats_pro():
    B, cin, cout, L = 2, 16, 16, 1000
    x = torch.randn(B, cin, L) * 2 + 0.5; w = torch.randn(cout, cin, 3) / 7
    mean = x.mean(2); var = x.var(2, unbiased=False); rstd = 1 / torch.sqrt(var + 1e-2)
    st = torch.stack([mean, rstd], -1).to(dev)
    h = F.gelu(F.instance_norm(x, eps=1e-2))
    want = F.conv1d(h, w, padding=1)
    y = torch.zeros(B, L, cout, device=dev)
    tile = lib.conv_tile(cin, cout, 3, 1, lib.MODE_CONTIG, B, L); nt = (L + tile - 1) // tile
    part = torch.zeros(B, nt, 2, cout, device=dev)
    lib.conv_forward(lib.conv_args(x=cl(x).to(dev), w=pack_fwd(w).to(dev), y=y, B=B, L_in=L, L_out=L, cin=cin, cout=cout, taps=3, stride=1, pad=1,
                                   pro=lib.PRO_IN_GELU, pro_stats=st, epi=lib.EPI_STATS, part=part))
    report('conv IN_GELU prologue', y, cl(want))
    out = torch.zeros(B, cout, 2, device=dev)
    lib.stats_finalize(part, B, nt, cout, L, 1e-2, 0, out)
    report('stats mean', out[..., 0], want.mean(2))
    report('stats rstd', out[..., 1], 1 / torch.sqrt(want.var(2, unbiased=False) + 1e-2))

def t_conv_split_precision():
    """bf16x3 matrix-core path (w = hi + lo planes): same contract as fp32, error <= ~2^-16 per product."""
    B = 2
    for (cin, cout, taps, stride, L, mode, dil) in [(64, 64, 3, 1, 300, 0, 1), (32, 64, 3, 1, 300, 0, 1), (128, 128, 3, 2, 256, 0, 1), (128, 384, 1, 1, 333, 0, 1),
                                                   (128, 128, 7, 1, 200, 1, 4), (64, 128, 1, 2, 256, 0, 1), (16, 16, 3, 1, 300, 0, 1), (16, 32, 3, 2, 256, 0, 1), (16, 16, 1, 2, 256, 0, 1),
                                                   (16, 32, 1, 2, 512, 0, 1)]:
        pad = (taps // 2) * dil
        x = torch.randn(B, cin, L); w = torch.randn(cout, cin, taps) / math.sqrt(cin * taps)
        want = F.conv1d(x.double(), w.double(), stride=stride, padding=pad, dilation=dil).float()
        Lo = want.shape[-1]
        wp = pack_fwd(w).to(dev); wh, wl = lib.frag_major_planes(wp.view(cout, taps * cin))
        y = torch.zeros(B, Lo, cout, device=dev)
        lib.conv_forward(lib.conv_args(x=cl(x).to(dev), w=wp, w_hi=wh, w_lo=wl, y=y, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride,
                                       pad=pad, dil=dil, mode=mode))
        report(f'bf16x3 conv {cin}->{cout} k{taps} s{stride} d{dil}', y, cl(want), tol=5e-5)
    # data gradient through the transposed stride-2 mode
    cin, cout, L = 128, 128, 256
    x = torch.randn(B, cin, L, requires_grad=True); w = torch.randn(cout, cin, 3) / 20
    yv = F.conv1d(x, w, stride=2, padding=1); gy = torch.randn_like(yv); yv.backward(gy)
    wb = w.permute(1, 2, 0).contiguous().to(dev); wh, wl = lib.frag_major_planes(wb.view(cin, 3 * cout))
    gx = torch.zeros(B, L, cin, device=dev)
    lib.conv_forward(lib.conv_args(x=cl(gy).to(dev), w=wb, w_hi=wh, w_lo=wl, y=gx, B=B, L_in=L // 2, L_out=L, cin=cout, cout=cin, taps=3, stride=2, pad=1,
                                   mode=lib.MODE_UP2))
    report('bf16x3 dgrad 128 s2', gx, cl(x.grad), tol=5e-5)

def t_conv_wide():
    """conv_wide_kernel (>= 64-channel k=3 layers, persistent, weights in registers) against the CPU ops it replaces: forward with the
    GELU / IN+GELU prologues + statistics partials (stride 1 and 2), and the stride-1 data gradient with the instance-norm-backward
    prologue, GELU' epilogue, residual add at even positions and backward statistics.  Lengths that are no multiple of the 64-position tile."""
    B = 3
    for (cin, cout, stride, pro, L) in [(32, 64, 1, lib.PRO_GELU, 1000), (64, 64, 1, lib.PRO_IN_GELU, 777), (64, 64, 2, lib.PRO_IN_GELU, 1302), (64, 128, 1, lib.PRO_GELU, 450),
                                        (128, 128, 1, lib.PRO_IN_GELU, 333), (128, 128, 2, lib.PRO_IN_GELU, 514), (64, 64, 1, lib.PRO_GELU, 64), (128, 128, 1, lib.PRO_GELU, 130)]:
        x = torch.randn(B, cin, L) * 1.5 + 0.3; w = torch.randn(cout, cin, 3) / math.sqrt(3 * cin)
        if pro == lib.PRO_IN_GELU:
            st = torch.stack([x.mean(2), 1 / torch.sqrt(x.var(2, unbiased=False) + 1e-2)], -1).to(dev)
            h = F.gelu(F.instance_norm(x, eps=1e-2))
        else:
            st, h = None, F.gelu(x)
        want = F.conv1d(h.double(), w.double(), stride=stride, padding=1).float()
        Lo = want.shape[-1]
        wp = pack_fwd(w).to(dev); wh, wl = lib.frag_major_planes(wp.view(cout, 3 * cin))
        y = torch.zeros(B, Lo, cout, device=dev)
        a = lib.conv_args(x=cl(x).to(dev), w=wp, w_hi=wh, w_lo=wl, y=y, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=3, stride=stride, pad=1, pro=pro,
                          pro_stats=st, epi=lib.EPI_STATS)
        tile = lib.conv_tile_of(a); nt = (Lo + tile - 1) // tile
        part = torch.full((B, nt, 2, cout), float('nan'), device=dev); lib.set_part(a, part)
        lib.conv_forward(a)
        tag = f'wide fwd {cin}->{cout} s{stride} pro{pro} L{L}'
        report(tag + f' (tile {tile})', y, cl(want), tol=5e-5)
        RES.append((tag + ' uses the persistent wide kernel', tile in (64, 128)))
        out = torch.zeros(B, cout, 2, device=dev); lib.stats_finalize(part, B, nt, cout, Lo, 1e-2, 0, out)
        report(tag + ' mean', out[..., 0], want.mean(2), tol=2e-5); report(tag + ' rstd', out[..., 1], 1 / torch.sqrt(want.var(2, unbiased=False) + 1e-2), tol=5e-5)
    # data gradient of y_k = conv(h), h = GELU(n_in): gout = (W^T gy [+ add_even]) * GELU'(n_in), gy = IN-backward(g; y_k)
    for (cg, ch, L, aux_norm, add_even) in [(64, 64, 700, True, False), (128, 128, 260, True, False), (64, 32, 500, False, True), (128, 64, 322, False, True),
                                            (64, 64, 130, False, True), (128, 128, 64, False, False)]:
        g = torch.randn(B, cg, L) * 0.1; yk = torch.randn(B, cg, L) * 2 + 0.2
        w = torch.randn(cg, ch, 3) / math.sqrt(3 * ch)            # forward weight [cout = cg][cin = ch][3]
        aux = torch.randn(B, ch, L) * 1.3 - 0.1                   # the layer's input-side pre-norm / pre-activation tensor
        mean, rstd = yk.mean(2, keepdim=True), 1 / torch.sqrt(yk.var(2, unbiased=False, keepdim=True) + 1e-2)
        n = (yk - mean) * rstd
        s1, s2 = g.mean(2, keepdim=True), (g * n).mean(2, keepdim=True)
        gy = rstd * (g - s1 - n * s2)
        d = F.conv_transpose1d(gy.double(), w.double(), stride=1, padding=1).float()   # W^T gy
        ev = torch.randn(B, ch, L // 2) * 0.05 if add_even else None
        if add_even:
            d[:, :, 0:2 * (L // 2):2] += ev
        if aux_norm:
            am, ar = aux.mean(2, keepdim=True), 1 / torch.sqrt(aux.var(2, unbiased=False, keepdim=True) + 1e-2)
            na = (aux - am) * ar
        else:
            na = aux
        nad = na.double().requires_grad_(True); F.gelu(nad).sum().backward()
        want = d * nad.grad.float()
        wb = w.permute(1, 2, 0).contiguous().to(dev); wh, wl = lib.frag_major_planes(wb.view(ch, 3 * cg))
        st = torch.stack([mean.squeeze(2), rstd.squeeze(2)], -1).to(dev); bst = torch.stack([s1.squeeze(2), s2.squeeze(2)], -1).to(dev)
        ast = torch.stack([am.squeeze(2), ar.squeeze(2)], -1).to(dev) if aux_norm else None
        gout = torch.zeros(B, L, ch, device=dev)
        a = lib.conv_args(x=cl(g).to(dev), x2=cl(yk).to(dev), w=wb, w_hi=wh, w_lo=wl, y=gout, B=B, L_in=L, L_out=L, cin=cg, cout=ch, taps=3, stride=1, pad=1, flip=1,
                          pro=lib.PRO_INBWD, pro_stats=st, pro_bstats=bst, epi=lib.EPI_GP, aux=cl(aux).to(dev), aux_stats=ast,
                          add_even=cl(ev).to(dev) if add_even else None)
        tile = lib.conv_tile_of(a); nt = (L + tile - 1) // tile
        part = torch.full((B, nt, 2, ch), float('nan'), device=dev); lib.set_part(a, part)
        lib.conv_forward(a)
        tag = f'wide dgrad {cg}->{ch} L{L} norm{int(aux_norm)} even{int(add_even)}'
        report(tag + f' (tile {tile})', gout, cl(want), tol=5e-5)
        RES.append((tag + ' uses the persistent wide kernel', tile in (32, 64, 128)))   # (128 -> 128: 32-position tiles since round 4)
        out = torch.zeros(B, ch, 2, device=dev); lib.stats_finalize(part, B, nt, ch, L, 0.0, 1, out)
        report(tag + ' sum g', out[..., 0], want.mean(2), tol=2e-5); report(tag + ' sum g*n', out[..., 1], (want * na).mean(2), tol=2e-5)


def t_conv_dilated():
    B, C, S = 2, 128, 200
    for d in (1, 4, 32):
        x = torch.randn(B, C, S); w = torch.randn(C, C, 7) / 30
        want = F.conv1d(x, w, padding=3 * d, dilation=d)
        y = torch.zeros(B, S, C, device=dev)
        lib.conv_forward(lib.conv_args(x=cl(x).to(dev), w=pack_fwd(w).to(dev), y=y, B=B, L_in=S, L_out=S, cin=C, cout=C, taps=7, stride=1, dil=d,
                                       pad=3 * d, mode=lib.MODE_DILATED))
        report(f'dilated conv d={d}', y, cl(want))
    # taps=4/stride=4 linear over [B,4S,C]
    Cc = 64
    x = torch.randn(B, 4 * S, Cc); W = torch.randn(128, 4 * Cc) / 16; bias = torch.randn(128)
    want = F.linear(x.reshape(B, S, 4 * Cc), W, bias)
    y = torch.zeros(B, S, 128, device=dev)
    lib.conv_forward(lib.conv_args(x=x.to(dev), w=W.to(dev), y=y, B=B, L_in=4 * S, L_out=S, cin=Cc, cout=128, taps=4, stride=4, pad=0,
                                   mode=lib.MODE_DILATED, epi=lib.EPI_BIAS, bias=bias.to(dev)))
    report('linear as taps4/stride4', y, want)

def t_dgrad():
    B = 2
    for (cin, cout, stride, L) in [(16, 16, 1, 600), (32, 64, 1, 300), (16, 16, 2, 600), (128, 128, 2, 256), (64, 64, 2, 520)]:
        x = torch.randn(B, cin, L, requires_grad=True); w = torch.randn(cout, cin, 3) / 7
        y = F.conv1d(x, w, stride=stride, padding=1)
        gy = torch.randn_like(y)
        y.backward(gy)
        Lo = y.shape[-1]
        wb = w.permute(1, 2, 0).contiguous().to(dev)  # [cin][taps][cout]
        gx = torch.zeros(B, L, cin, device=dev)
        if stride == 1:
            a = lib.conv_args(x=cl(gy).to(dev), w=wb, y=gx, B=B, L_in=Lo, L_out=L, cin=cout, cout=cin, taps=3, stride=1, pad=1, flip=1)
        else:
            a = lib.conv_args(x=cl(gy).to(dev), w=wb, y=gx, B=B, L_in=Lo, L_out=L, cin=cout, cout=cin, taps=3, stride=2, pad=1, mode=lib.MODE_UP2)
        lib.conv_forward(a)
        report(f'dgrad {cin}->{cout} s{stride}', gx, cl(x.grad))

def t_wgrad():
    B = 2
    for (cin, cout, taps, stride, dil, L) in [(16, 16, 3, 1, 1, 700), (16, 32, 3, 1, 1, 300), (32, 32, 3, 2, 1, 600), (64, 64, 3, 1, 1, 300), (64, 128, 3, 2, 1, 256),
                                              (128, 128, 3, 1, 1, 200), (128, 128, 7, 1, 4, 200), (16, 32, 1, 2, 1, 512), (128, 384, 1, 1, 1, 300), (64, 128, 1, 2, 1, 256)]:
        pad = (taps // 2) * dil
        x = torch.randn(B, cin, L); w = (torch.randn(cout, cin, taps) / 7).requires_grad_(True)
        y = F.conv1d(x, w, stride=stride, padding=pad, dilation=dil)
        gy = torch.randn_like(y); y.backward(gy)
        Lo = y.shape[-1]
        gyd = lib.wgrad_grid_y(cin, cout, taps, dil)
        nslab = 8 * lib.wgrad_slabs_per_block(cin, cout, taps, dil)
        slab = torch.zeros(nslab * cout * cin * taps, device=dev)
        lib.wgrad(g=cl(gy).to(dev), x=cl(x).to(dev), slab=slab, nslab=nslab, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride,
                  pad=pad, dil=dil)
        gw = torch.zeros(cout, cin, taps, device=dev)
        lib.wgrad_reduce(slab, nslab, gw, cout, cin, taps, dil)
        report(f'wgrad {cin}->{cout} k{taps} s{stride} d{dil}', gw, w.grad, tol=3e-4)
        if cin >= 64 and cout >= 64:  # split-precision (bf16x3) variant of the tile-split kernel
            slab.zero_(); gw.zero_()
            lib.wgrad(g=cl(gy).to(dev), x=cl(x).to(dev), slab=slab, nslab=nslab, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride,
                      pad=pad, dil=dil, split_precision=True)
            lib.wgrad_reduce(slab, nslab, gw, cout, cin, taps, dil)
            report(f'wgrad bf16x3 {cin}->{cout} k{taps} s{stride} d{dil}', gw, w.grad, tol=3e-4)

def t_wgrad_pipelined():
    """Round 6: the software-pipelined form of the split-precision tile-split weight gradient (wgrad_bf_pf_kernel: k = 1 / tap-split launches
    without a gradient-side transform -- the trunk's linears, the SequenceCNN convs, the encoders' dense layer and 1x1 joins) against the
    unpipelined kernel (W2S_NO_WGRAD_PF=1, read per launch): same products in the same order => bit for bit; and against fp64."""
    cases = [  # cin, cout, taps, stride, dil, pad, B, L_out, pro_h, grid.x
        (128, 384, 1, 1, 1, 0, 1, 3000, lib.PRO_NONE, 7), (128, 512, 1, 1, 1, 0, 1, 2049, lib.PRO_NONE, 5), (128, 128, 1, 1, 1, 0, 1, 4100, lib.PRO_NONE, 9),
        (128, 128, 4, 4, 1, 0, 1, 1500, lib.PRO_NONE, 6), (128, 128, 7, 1, 4, 12, 3, 333, lib.PRO_NONE, 4), (128, 128, 7, 1, 32, 96, 2, 960, lib.PRO_NONE, 8),
        (128, 128, 4, 4, 1, 0, 3, 200, lib.PRO_GELU, 3), (64, 128, 1, 2, 1, 0, 2, 700, lib.PRO_GELU, 5), (128, 256, 1, 1, 1, 0, 1, 1000, lib.PRO_NONE, 2),
        (64, 64, 1, 2, 1, 0, 2, 900, lib.PRO_GELU, 5), (128, 128, 1, 1, 1, 0, 1, 20, lib.PRO_NONE, 1)]
    for (cin, cout, taps, stride, dil, pad, B, Lo, pro_h, gx) in cases:
        L_in = Lo * stride
        g = torch.randn(B, Lo, cout, device=dev); x = torch.randn(B, L_in, cin, device=dev) * 1.5
        gyd = lib.wgrad_grid_y(cin, cout, taps, dil)
        nslab = gx * lib.wgrad_slabs_per_block(cin, cout, taps, dil)
        outs = []
        for nopf in ('1', None):
            if nopf:
                os.environ['W2S_NO_WGRAD_PF'] = nopf
            else:
                os.environ.pop('W2S_NO_WGRAD_PF', None)
            slab = torch.full((nslab * cout * cin * taps,), float('nan'), device=dev)
            lib.wgrad(g=g, x=x, slab=slab, nslab=nslab, B=B, L_in=L_in, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride, pad=pad, dil=dil, pro_h=pro_h,
                      split_precision=True)
            gw = torch.zeros(cout, cin, taps, device=dev)
            lib.wgrad_reduce(slab, nslab, gw, cout, cin, taps, dil)
            torch.cuda.synchronize()
            outs.append((slab.clone(), gw))
        os.environ.pop('W2S_NO_WGRAD_PF', None)
        tag = f'wgrad pipelined {cin}->{cout} k{taps} s{stride} d{dil} B{B} L{Lo} pro_h{pro_h}'
        RES.append((tag + ' slabs bit-equal', bool(torch.equal(outs[0][0], outs[1][0]))))
        print(f'{"OK  " if RES[-1][1] else "FAIL"} {tag} slabs bit-equal to the unpipelined kernel', flush=True)
        h = (F.gelu(x.double().cpu()) if pro_h == lib.PRO_GELU else x.double().cpu())
        hp = F.pad(h, (0, 0, pad, pad + taps * dil * stride))
        want = torch.zeros(cout, cin, taps, dtype=torch.float64)
        for j in range(taps):
            rows = torch.arange(Lo) * stride + j * dil        # window row of output position t, tap j (origin -pad)
            want[:, :, j] = torch.einsum('bto,btc->oc', g.double().cpu(), hp[:, rows, :])
        report(tag + ' vs fp64', outs[1][1], want.float(), tol=3e-4)

def t_linear_pipelined():
    """Round 6: the persistent pipelined GEMM of the transformer's row-wise linears (csrc/linear_pf.hip) against the generic tile kernel it
    replaces (W2S_NO_LINEAR_PF=1, read per launch): every fusion of the bias epilogue, the strided CLS-row layouts, K = 384 / 512 as 3 / 4
    chunks, ragged row counts -- same products in the same order => bit for bit -- and one case against fp64."""
    cases = [  # rows, K, N, fuse, bias, drop_p, ldx, ldy, ld_aux
        (3000, 128, 384, 0, True, 0.0, 128, 384, 0),                          # in_proj
        (2049, 128, 128, lib.FUSE_ADD_DROP, True, 0.1, 128, 128, 128),        # out_proj + residual + dropout
        (1500, 128, 128, lib.FUSE_ADD_DROP, True, 0.1, 640, 128, 640),        # ... on the CLS rows of a [N, 5, 128] tensor
        (4100, 128, 512, lib.FUSE_Y2_GELU_DROP, True, 0.1, 128, 512, 0),      # linear1 -> (f1, dropout(GELU))
        (2500, 512, 128, lib.FUSE_ADD_DROP, True, 0.0, 128, 128, 128),        # linear2 (four 128-wide chunks) + residual
        (2500, 128, 512, lib.FUSE_GELU_BWD_DROP, False, 0.1, 128, 512, 512),  # data gradient of linear2 x GELU'(f1) x mask
        (1000, 512, 128, 0, False, 0.0, 128, 128, 0),                         # data gradient of linear1
        (1300, 384, 128, 0, False, 0.0, 128, 128, 0),                         # data gradient of in_proj (three chunks)
        (700, 128, 128, 0, False, 0.0, 128, 640, 0),                          # data gradient of out_proj into the CLS rows (strided output)
        (257, 128, 256, 0, True, 0.0, 128, 256, 0)]
    for (rows, K, N, fuse, bias, drop_p, ldx, ldy, ld_aux) in cases:
        taps = K // 128
        x = torch.randn(rows * taps, ldx, device=dev) if taps > 1 else torch.randn(rows, ldx, device=dev)
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).contiguous()
        wh, wl = lib.frag_major_planes(w)
        b = torch.randn(N, device=dev) if bias else None
        aux = torch.randn(rows, ld_aux, device=dev) if ld_aux else None
        outs = []
        for nopf in ('1', None):
            if nopf:
                os.environ['W2S_NO_LINEAR_PF'] = nopf
            else:
                os.environ.pop('W2S_NO_LINEAR_PF', None)
            y = torch.full((rows, ldy), float('nan'), device=dev)
            y2 = torch.full((rows, N), float('nan'), device=dev) if fuse & lib.FUSE_Y2_GELU_DROP else None
            a = lib.conv_args(x=x, w=w, w_hi=wh, w_lo=wl, y=y, y2=y2, ldy2=N, B=1, L_in=rows * taps, L_out=rows, cin=128, cout=N, taps=taps, stride=taps, pad=0,
                              mode=lib.MODE_DILATED if taps > 1 else lib.MODE_CONTIG, ldx=ldx, ldy=ldy, epi=lib.EPI_BIAS, bias=b, aux=aux, ld_aux=ld_aux,
                              fuse=fuse, drop_p=drop_p, drop_seed=12345)
            takes = bool(lib.load().w2s_linear_pf_takes(__import__('ctypes').byref(a)))
            lib.conv_forward(a)
            torch.cuda.synchronize()
            outs.append((y[:, :N].clone(), None if y2 is None else y2.clone(), takes))
        os.environ.pop('W2S_NO_LINEAR_PF', None)
        tag = f'linear pipelined rows {rows} K {K} N {N} fuse {fuse} ldx {ldx} ldy {ldy}'
        RES.append((tag + ' is taken', outs[1][2] and not outs[0][2]))
        ok = bool(torch.equal(outs[0][0], outs[1][0])) and (outs[0][1] is None or bool(torch.equal(outs[0][1], outs[1][1]))) and not bool(torch.isnan(outs[1][0]).any())
        RES.append((tag + ' bit-equal', ok))
        print(f'{"OK  " if ok and RES[-2][1] else "FAIL"} {tag}: bit-equal to the generic kernel (taken: {outs[1][2]})', flush=True)
        if fuse == 0 and ldx == 128:
            want = x.double().cpu().view(rows, K) @ w.double().cpu().t() + (b.double().cpu() if bias else 0.0)
            report(tag + ' vs fp64', outs[1][0], want.float(), tol=1e-4)

def t_seq_conv():
    """Round 6: the SequenceCNN kernel with the channel LayerNorm in its epilogue (csrc/seq_conv.hip) against the launches it replaces --
    generic dilated conv -> w2s_layernorm_fwd(gelu) forward, generic data gradient -> w2s_layernorm_bwd(gelu) backward -- and against fp64:
    every dilation, symmetric and causal padding, ragged lengths (a last tile of 8 positions, a sample shorter than the halo), the strided
    CLS-row input of block 0.  The conv part has the same products in the same order: bit for bit; the LayerNorm sums differ in order."""
    cases = [(2, 960, 1, False, 128), (3, 200, 32, False, 128), (2, 130, 4, True, 128), (2, 64, 8, False, 640), (1, 1000, 16, False, 128), (2, 50, 2, False, 128),
             (2, 200, 32, True, 128), (1, 72, 16, False, 128)]
    C = 128
    for (B, S, d, causal, ldx) in cases:
        pad = 6 * d if causal else 3 * d
        xs = torch.randn(B, S, ldx, device=dev)
        x = xs if ldx == C else None   # (ldx > 128: the kernel reads the first 128 floats of every ldx-wide row)
        w = torch.randn(C, C, 7) / math.sqrt(7 * C)                      # torch layout [cout][cin][taps]
        wf = w.permute(0, 2, 1).contiguous().to(dev)                       # forward packing [cout][taps][cin]
        wb = w.permute(1, 2, 0).contiguous().to(dev)                       # data-gradient packing [cin][taps][cout]
        fh, fl = lib.frag_major_planes(wf.view(C, 7 * C)); bh, bl = lib.frag_major_planes(wb.view(C, 7 * C))
        gamma = (torch.rand(C, device=dev) + 0.5); beta = torch.randn(C, device=dev) * 0.1
        tag = f'seq conv B{B} S{S} d{d} {"causal" if causal else "sym"} ldx{ldx}'
        # ---- forward
        y0 = torch.zeros(B, S, C, device=dev); hn0 = torch.zeros(B, S, C, device=dev); rs0 = torch.zeros(B * S, 2, device=dev)
        lib.conv_forward(lib.conv_args(x=xs, w=wf, w_hi=fh, w_lo=fl, y=y0, B=B, L_in=S, L_out=S, cin=C, cout=C, taps=7, stride=1, dil=d, pad=pad, mode=lib.MODE_DILATED, ldx=ldx))
        lib.layernorm_fwd(y0, C, gamma, beta, hn0, C, rs0, B * S, C, 1e-5, gelu=True)
        y1 = torch.full((B, S, C), float('nan'), device=dev); hn1 = torch.full((B, S, C), float('nan'), device=dev); rs1 = torch.full((B * S, 2), float('nan'), device=dev)
        lib.seq_conv(x=xs, w_hi=fh, w_lo=fl, B=B, S=S, ldx=ldx, dil=d, pad=pad, mode=1, y=y1, out=hn1, rs=rs1, gamma=gamma, beta=beta, eps=1e-5)
        RES.append((tag + ' y bit-equal', bool(torch.equal(y0, y1))))
        print(f'{"OK  " if RES[-1][1] else "FAIL"} {tag}: conv output bit-equal to the generic kernel', flush=True)
        report(tag + ' GELU(LN) vs the LayerNorm kernel', hn1, hn0, tol=3e-6)
        report(tag + ' row statistics', rs1, rs0, tol=3e-6)
        x64 = xs[:, :, :C].double().cpu().transpose(1, 2)
        want_y = F.conv1d(F.pad(x64, (pad, 6 * d - pad)), w.double(), dilation=d).transpose(1, 2)
        want_hn = F.gelu(F.layer_norm(want_y, (C,), gamma.double().cpu(), beta.double().cpu(), 1e-5))
        report(tag + ' GELU(LN(conv)) vs fp64', hn1, want_hn.float(), tol=1e-4)
        # ---- backward: gy = gradient w.r.t. THIS layer's conv output; the layer below has pre-norm output yl, statistics rsl
        gy = torch.randn(B, S, C, device=dev) * 0.1
        yl = torch.randn(B, S, C, device=dev) * 0.7 + 0.1
        rsl = torch.zeros(B * S, 2, device=dev); tmp = torch.zeros(B, S, C, device=dev)
        lib.layernorm_fwd(yl, C, gamma, beta, tmp, C, rsl, B * S, C, 1e-5, gelu=True)
        gh0 = torch.zeros(B, S, C, device=dev)
        lib.conv_forward(lib.conv_args(x=gy, w=wb, w_hi=bh, w_lo=bl, y=gh0, B=B, L_in=S, L_out=S, cin=C, cout=C, taps=7, stride=1, dil=d, pad=6 * d - pad, flip=1, mode=lib.MODE_DILATED))
        npl = max(1, min(1024, (B * S + 31) // 32))
        out0 = torch.zeros(B, S, C, device=dev); pg = torch.zeros(npl, C, device=dev); pb = torch.zeros(npl, C, device=dev)
        lib.layernorm_bwd(gh0, C, yl, C, gamma, beta, rsl, None, out0, C, pg, pb, B * S, C, True, npl)
        g0 = torch.full((B, S, C), float('nan'), device=dev)
        lib.seq_conv(x=gy, w_hi=bh, w_lo=bl, B=B, S=S, ldx=C, dil=d, pad=6 * d - pad, flip=1, mode=0, y=g0)
        RES.append((tag + ' data gradient bit-equal', bool(torch.equal(g0, gh0))))
        print(f'{"OK  " if RES[-1][1] else "FAIL"} {tag}: plain data gradient bit-equal to the generic kernel', flush=True)
        nt = B * ((S + 63) // 64)
        out1 = torch.full((B, S, C), float('nan'), device=dev); part = torch.full((nt, 2, C), float('nan'), device=dev)
        lib.seq_conv(x=gy, w_hi=bh, w_lo=bl, B=B, S=S, ldx=C, dil=d, pad=6 * d - pad, flip=1, mode=2, out=out1, rs=rsl, gamma=gamma, beta=beta, yl=yl, part=part, eps=1e-5)
        report(tag + ' LayerNorm backward in the epilogue', out1, out0, tol=2e-5)
        report(tag + ' LayerNorm weight gradient', part[:, 0].sum(0), pg.sum(0), tol=2e-5)
        report(tag + ' LayerNorm bias gradient', part[:, 1].sum(0), pb.sum(0), tol=2e-5)
        # fp64 autograd of GELU(LN(yl)) against the incoming gradient gh0
        yl64 = yl.double().cpu().requires_grad_(True); gm64 = gamma.double().cpu().requires_grad_(True); bt64 = beta.double().cpu().requires_grad_(True)
        F.gelu(F.layer_norm(yl64, (C,), gm64, bt64, 1e-5)).backward(gh0.double().cpu())
        report(tag + ' LayerNorm backward vs fp64 autograd', out1, yl64.grad.float(), tol=1e-4)
        report(tag + ' LayerNorm weight gradient vs fp64 autograd', part[:, 0].sum(0), gm64.grad.float(), tol=3e-4)

def t_conv_wide_up2():
    """transposed stride-2 form of conv_wide_kernel (data gradient of the stride-2 conv3, >= 64 channels), symmetric and causal padding:
    gout = (W^T (x) gy) * GELU'(IN(aux)), gy = IN-backward(g * GELU'(n3); y3) -- against the autograd of F.conv1d in fp64."""
    B = 3
    for (c, Lg, pad) in [(64, 500, 1), (128, 161, 1), (64, 333, 2), (128, 96, 2), (64, 32, 1)]:
        L = 2 * Lg
        g = torch.randn(B, c, Lg) * 0.1; y3 = torch.randn(B, c, Lg) * 2 + 0.2
        w = torch.randn(c, c, 3) / math.sqrt(3 * c)              # forward weight of the stride-2 conv [cout][cin][3]
        aux = torch.randn(B, c, L) * 1.3 - 0.1                    # y2: the conv's pre-norm input
        mean, rstd = y3.mean(2, keepdim=True), 1 / torch.sqrt(y3.var(2, unbiased=False, keepdim=True) + 1e-2)
        n = ((y3 - mean) * rstd).double().requires_grad_(True); F.gelu(n).sum().backward()
        gn = (g.double() * n.grad).float(); n = n.detach().float()
        s1, s2 = gn.mean(2, keepdim=True), (gn * n).mean(2, keepdim=True)
        gy = rstd * (gn - s1 - n * s2)
        # forward: y3[u] = sum_j W_j h[2u + j - pad] (pad 1: symmetric; pad 2: causal, left padding only); d = its data gradient
        hz = torch.zeros(B, c, L, dtype=torch.float64, requires_grad=True)
        F.conv1d(F.pad(hz, (pad, 2 - pad)), w.double(), stride=2).backward(gy.double())
        d = hz.grad.float()
        am, ar = aux.mean(2, keepdim=True), 1 / torch.sqrt(aux.var(2, unbiased=False, keepdim=True) + 1e-2)
        na = ((aux - am) * ar).double().requires_grad_(True); F.gelu(na).sum().backward()
        want = d * na.grad.float(); na = na.detach().float()
        wb = w.permute(1, 2, 0).contiguous().to(dev); wh, wl = lib.frag_major_planes(wb.view(c, 3 * c))
        st = torch.stack([mean.squeeze(2), rstd.squeeze(2)], -1).to(dev); bst = torch.stack([s1.squeeze(2), s2.squeeze(2)], -1).to(dev)
        ast = torch.stack([am.squeeze(2), ar.squeeze(2)], -1).to(dev)
        gout = torch.zeros(B, L, c, device=dev)
        a = lib.conv_args(x=cl(g).to(dev), x2=cl(y3).to(dev), w=wb, w_hi=wh, w_lo=wl, y=gout, B=B, L_in=Lg, L_out=L, cin=c, cout=c, taps=3, stride=2, pad=pad,
                          mode=lib.MODE_UP2, pro=lib.PRO_INBWD_GP, pro_stats=st, pro_bstats=bst, epi=lib.EPI_GP, aux=cl(aux).to(dev), aux_stats=ast)
        tile = lib.conv_tile_of(a); nt = (L + tile - 1) // tile
        part = torch.full((B, nt, 2, c), float('nan'), device=dev); lib.set_part(a, part)
        lib.conv_forward(a)
        tag = f'wide up2 {c} Lg{Lg} pad{pad}'
        report(tag + f' (tile {tile})', gout, cl(want), tol=5e-5)
        RES.append((tag + ' uses the persistent wide kernel', tile in (64, 128)))
        out = torch.zeros(B, c, 2, device=dev); lib.stats_finalize(part, B, nt, c, L, 0.0, 1, out)
        report(tag + ' sum g', out[..., 0], want.mean(2), tol=2e-5); report(tag + ' sum g*n', out[..., 1], (want * na).mean(2), tol=2e-5)


def t_wgrad_wide():
    """role-split weight gradient of the >= 64-channel k=3 layers (wgrad_wide.hip) with its on-load transforms vs fp64 torch."""
    B = 3
    for (cin, cout, stride, pro_h, L) in [(64, 64, 1, lib.PRO_IN_GELU, 1000), (64, 64, 1, lib.PRO_GELU, 333), (64, 64, 2, lib.PRO_IN_GELU, 1026),
                                          (128, 128, 1, lib.PRO_IN_GELU, 500), (128, 128, 1, lib.PRO_GELU, 97), (128, 128, 2, lib.PRO_IN_GELU, 258),
                                          (64, 128, 1, lib.PRO_GELU, 640), (32, 64, 1, lib.PRO_GELU, 777), (64, 64, 1, lib.PRO_IN_GELU, 40), (128, 128, 2, lib.PRO_IN_GELU, 64)]:
        Lo = L // stride
        x = torch.randn(B, L, cin, device=dev); g = torch.randn(B, Lo, cout, device=dev); y = torch.randn(B, Lo, cout, device=dev)
        st = torch.stack([torch.randn(B, cout, device=dev) * 0.1, torch.rand(B, cout, device=dev) + 0.5], dim=-1).contiguous()
        bst = (torch.randn(B, cout, 2, device=dev) * 0.05).contiguous()
        sti = torch.stack([torch.randn(B, cin, device=dev) * 0.1, torch.rand(B, cin, device=dev) + 0.5], dim=-1).contiguous()
        pro_g = lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP
        kw = dict(g=g, g2=y, g_stats=st, g_bstats=bst, x=x, x_stats=sti if pro_h == lib.PRO_IN_GELU else None, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout,
                  taps=3, stride=stride, pad=1, pro_g=pro_g, pro_h=pro_h, split_precision=True)
        assert lib.wgrad_max_blocks(slab=None, nslab=0, **kw) == 256 and lib.wgrad_slabs_per_block_of(slab=None, nslab=0, **kw) == 1, 'not taken by the role-split kernel'
        nslab = min(5, (B * Lo + 255) // 256)   # the engine's rule: never more workgroups than 256-position chunks
        slab = torch.full((nslab * cout * cin * 3,), float('nan'), device=dev)
        lib.wgrad(slab=slab, nslab=nslab, **kw)
        gw = torch.zeros(cout, cin, 3, device=dev)
        lib.wgrad_reduce(slab, nslab, gw, cout, cin, 3, 1)
        # fp64 reference
        xd, gd, yd = x.double(), g.double(), y.double()
        n = (yd - st[:, None, :, 0].double()) * st[:, None, :, 1].double()
        gp = 0.5 * (1 + torch.erf(n / 2 ** 0.5)) + n * torch.exp(-0.5 * n * n) / (2 * math.pi) ** 0.5
        gn = gd * gp if stride == 2 else gd
        GY = st[:, None, :, 1].double() * (gn - bst[:, None, :, 0].double() - n * bst[:, None, :, 1].double())
        hn = (xd - sti[:, None, :, 0].double()) * sti[:, None, :, 1].double() if pro_h == lib.PRO_IN_GELU else xd
        H = 0.5 * hn * (1 + torch.erf(hn / 2 ** 0.5))
        w = torch.zeros(cout, cin, 3, device=dev, dtype=torch.float64, requires_grad=True)
        F.conv1d(H.transpose(1, 2), w, stride=stride, padding=1).backward(GY.transpose(1, 2)[:, :, :Lo])
        report(f'wgrad wide {cin}->{cout} s{stride} pro_h={pro_h} L{L}', gw, w.grad.float(), tol=3e-4)
        # many workgroups (one tile each) and the generic kernel must agree with it too
        nslab2 = min(B * ((Lo + 63) // 64), 256)
        slab2 = torch.full((nslab2 * cout * cin * 3,), float('nan'), device=dev)
        lib.wgrad(slab=slab2, nslab=nslab2, **kw)
        gw2 = torch.zeros(cout, cin, 3, device=dev)
        lib.wgrad_reduce(slab2, nslab2, gw2, cout, cin, 3, 1)
        report(f'wgrad wide {cin}->{cout} s{stride} pro_h={pro_h} L{L} ({nslab2} workgroups)', gw2, w.grad.float(), tol=3e-4)


def t_causal():
    """Causal padding (blocks.py:150-152,178-182) through the C-ABI: forward pad (k-1)*dil, flipped-tap data gradient pad 0, the
    transposed stride-2 kernel with pad 2, weight gradient pad (k-1)*dil, and the Cin = 1 first layer with its `causal` shift."""
    B = 2
    for (cin, cout, taps, stride, dil, L) in [(16, 16, 3, 1, 1, 700), (32, 32, 3, 2, 1, 600), (64, 64, 3, 1, 1, 300), (128, 128, 3, 2, 1, 256),
                                              (64, 64, 3, 2, 1, 258), (128, 128, 7, 1, 1, 200), (128, 128, 7, 1, 8, 200), (128, 128, 7, 1, 32, 300)]:
        x = torch.randn(B, cin, L, requires_grad=True); w = (torch.randn(cout, cin, taps) / math.sqrt(cin * taps)).requires_grad_(True)
        y = O.causal_conv1d(x, w, stride, dil)
        gy = torch.randn_like(y); y.backward(gy)
        Lo, pad = y.shape[-1], (taps - 1) * dil
        mode = lib.MODE_DILATED if taps == 7 else lib.MODE_CONTIG
        yd = torch.zeros(B, Lo, cout, device=dev)
        lib.conv_forward(lib.conv_args(x=cl(x.detach()).to(dev), w=pack_fwd(w.detach()).to(dev), y=yd, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps,
                                       stride=stride, pad=pad, dil=dil, mode=mode))
        report(f'causal conv {cin}->{cout} k{taps} s{stride} d{dil}', yd, cl(y))
        wb = w.detach().permute(1, 2, 0).contiguous().to(dev)
        gx = torch.zeros(B, L, cin, device=dev)
        if stride == 1:
            a = lib.conv_args(x=cl(gy).to(dev), w=wb, y=gx, B=B, L_in=Lo, L_out=L, cin=cout, cout=cin, taps=taps, stride=1, pad=0, dil=dil, flip=1, mode=mode)
        else:
            a = lib.conv_args(x=cl(gy).to(dev), w=wb, y=gx, B=B, L_in=Lo, L_out=L, cin=cout, cout=cin, taps=3, stride=2, pad=2, mode=lib.MODE_UP2)
        lib.conv_forward(a)
        report(f'causal dgrad {cin}->{cout} k{taps} s{stride} d{dil}', gx, cl(x.grad))
        nslab = 8 * lib.wgrad_slabs_per_block(cin, cout, taps, dil)
        slab = torch.zeros(nslab * cout * cin * taps, device=dev)
        lib.wgrad(g=cl(gy).to(dev), x=cl(x.detach()).to(dev), slab=slab, nslab=nslab, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride,
                  pad=pad, dil=dil)
        gw = torch.zeros(cout, cin, taps, device=dev)
        lib.wgrad_reduce(slab, nslab, gw, cout, cin, taps, dil)
        report(f'causal wgrad {cin}->{cout} k{taps} s{stride} d{dil}', gw, w.grad, tol=3e-4)
    # first layer (Cin = 1): values + statistics partials, and its weight gradients
    L, c, tile = 3000, 16, 1024
    x = torch.randn(B, L); x[1, 100:130] = float('inf')
    xs = torch.where(torch.isinf(x), 0.0, x)
    w1 = (torch.randn(c, 1, 3) / 2).requires_grad_(True)
    y1 = O.causal_conv1d(xs[:, None, :], w1)
    nt = (L + tile - 1) // tile
    yd = torch.zeros(B, L, c, device=dev); part = torch.zeros(B, nt, 2, c, device=dev)
    lib.enc_first_fwd(x.to(dev), w1.detach().to(dev), yd, part, B, L, c, tile, causal=True)
    report('causal first layer values', yd, cl(y1))
    report('causal first layer sum', part[:, :, 0].sum(1), y1.sum(2), tol=1e-4)
    report('causal first layer sumsq', part[:, :, 1].sum(1), (y1 * y1).sum(2), tol=1e-4)
    mean = y1.detach().mean(2); rstd = 1 / torch.sqrt(y1.detach().var(2, unbiased=False) + 1e-2)
    n = (y1 - mean[:, :, None].detach()) * rstd[:, :, None].detach()   # gy1 = rstd * (g - q1 - n q2): the instance-norm backward with given sums
    g = torch.randn(B, c, L); q1 = torch.randn(B, c) * 0.01; q2 = torch.randn(B, c) * 0.01
    gy1 = rstd[:, :, None] * (g - q1[:, :, None] - n.detach() * q2[:, :, None])
    (y1 * gy1).sum().backward()
    gpre = torch.randn(B, c, L // 2)
    want_d = torch.einsum('bou,bu->o', gpre, xs[:, 0::2])
    slab = torch.zeros(8, 64, device=dev)
    lib.enc_first_bwd(x.to(dev), cl(g).to(dev), yd, torch.stack([mean, rstd], -1).to(dev), torch.stack([q1, q2], -1).to(dev), cl(gpre).to(dev), slab, 8, B, L, c,
                      causal=True)
    tot = slab.sum(0)
    report('causal first layer dW1', tot[:48].view(c, 3), w1.grad.view(c, 3), tol=3e-4)
    report('causal first layer dWd', tot[48:], want_d, tol=3e-4)

def t_rowops():
    rows, C = 1000, 128
    x = torch.randn(rows, C, requires_grad=True); g = torch.randn(C) + 1; b = torch.randn(C)
    g.requires_grad_(True); b.requires_grad_(True)
    for gelu in (False, True):
        x.grad = g.grad = b.grad = None
        y = F.layer_norm(x, (C,), g, b, 1e-5)
        if gelu: y = F.gelu(y)
        go = torch.randn_like(y); y.backward(go)
        yd = torch.zeros(rows, C, device=dev); rs = torch.zeros(rows, 2, device=dev)
        lib.layernorm_fwd(x.detach().to(dev), C, g.detach().to(dev), b.detach().to(dev), yd, C, rs, rows, C, 1e-5, gelu)
        report(f'layernorm fwd gelu={gelu}', yd, y)
        gx = torch.zeros(rows, C, device=dev); npl = 16
        pg = torch.zeros(npl, C, device=dev); pb = torch.zeros(npl, C, device=dev)
        lib.layernorm_bwd(go.to(dev), C, x.detach().to(dev), C, g.detach().to(dev), b.detach().to(dev), rs, None, gx, C, pg, pb, rows, C, gelu, npl)
        report(f'layernorm bwd gx gelu={gelu}', gx, x.grad)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        lib.colsum(pg, npl, C, dg); lib.colsum(pb, npl, C, db)
        report(f'layernorm bwd dgamma gelu={gelu}', dg, g.grad, tol=3e-4); report(f'layernorm bwd dbeta gelu={gelu}', db, b.grad, tol=3e-4)

def t_attn():
    for D in (5, 2, 3, 6, 7, 12):   # 2 .. 6: four lanes per (n, head); 7 .. 12: the sixteen-lane form (rowops.hip)
        _attn_case(D)

def _attn_case(D):
    N, H = 50, 8; Fd = 128
    qkv = torch.randn(N * D, 3 * Fd, requires_grad=True)
    pad = torch.zeros(N, D, dtype=torch.bool); pad[::3, min(2, D - 1)] = True; pad[1::4, D - 1] = True
    q, k, v = qkv.view(N, D, 3, H, 16).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / 4
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(N * D, Fd)
    go = torch.randn_like(o); o.backward(go)
    od = torch.zeros(N * D, Fd, device=dev)
    kp = pad.to(torch.uint8).to(dev)
    lib.attn_fwd(qkv.detach().to(dev), kp, od, N, D, H)
    report(f'attn fwd D={D}', od, o)
    gq = torch.zeros(N * D, 3 * Fd, device=dev)
    lib.attn_bwd(qkv.detach().to(dev), kp, go.to(dev), gq, N, D, H)
    report(f'attn bwd D={D}', gq, qkv.grad)
    # token 0 as the only query (nq = 1; the last layer of the stack): its output row as before, the other rows untouched; backward = the full
    # kernel on a gradient that is zero off the CLS rows, with dropout on (same counter-based masks in both)
    for pdrop in (0.0, 0.1):
        full = torch.zeros(N * D, Fd, device=dev); lib.attn_fwd(qkv.detach().to(dev), kp, full, N, D, H, pdrop, 77)
        o1 = torch.full((N * D, Fd), 7.0, device=dev); lib.attn_fwd(qkv.detach().to(dev), kp, o1, N, D, H, pdrop, 77, nq=1)
        RES.append((f'attn fwd D={D} nq=1 p={pdrop}: CLS rows bit-equal, other rows untouched',
                    torch.equal(o1.view(N, D, Fd)[:, 0], full.view(N, D, Fd)[:, 0]) and bool((o1.view(N, D, Fd)[:, 1:] == 7.0).all())))
        gcls = torch.zeros(N, D, Fd, device=dev); gcls[:, 0] = go.view(N, D, Fd)[:, 0].to(dev)
        g_full = torch.zeros(N * D, 3 * Fd, device=dev); lib.attn_bwd(qkv.detach().to(dev), kp, gcls.view(N * D, Fd), g_full, N, D, H, pdrop, 77)
        gcls[:, 1:] = float('nan')   # nq = 1 must not read the other rows
        g1 = torch.full((N * D, 3 * Fd), float('nan'), device=dev); lib.attn_bwd(qkv.detach().to(dev), kp, gcls.view(N * D, Fd), g1, N, D, H, pdrop, 77, nq=1)
        RES.append((f'attn bwd D={D} nq=1 p={pdrop}: equals the full kernel on a CLS-only gradient', torch.equal(g1, g_full)))

def t_head_optim():
    rows, Fd, nc = 700, 128, 5
    pre = torch.randn(rows, Fd, requires_grad=True); W = (torch.randn(nc, Fd) / 10).requires_grad_(True); b = torch.randn(nc).requires_grad_(True)
    y = torch.randint(0, nc, (rows,)).float(); y[::7] = -1
    logits = F.linear(F.gelu(pre), W, b)
    loss = F.cross_entropy(logits, y.long(), ignore_index=-1); loss.backward()
    lg = torch.zeros(rows, nc, device=dev)
    lib.head_fwd(pre.detach().to(dev), Fd, W.detach().to(dev), b.detach().to(dev), lg, rows, Fd, nc, True)
    report('head logits', lg, logits)
    part = torch.zeros((rows + 255) // 256, 2, device=dev); lo = torch.zeros(2, device=dev); gl = torch.zeros(rows, nc, device=dev)
    cm = torch.zeros(nc, nc, dtype=torch.int64, device=dev)
    lib.ce_fwd_bwd(lg, y.to(dev), rows, nc, part, lo, gl, cm, 1.0)
    report('ce loss', lo[0], loss); 
    report('cmat', cm.float(), O.confusion_matrix(logits.argmax(-1), y, nc).float())
    gp = torch.zeros(rows, Fd, device=dev); npart = 8; pp = torch.zeros(npart, nc * Fd + nc, device=dev)
    lib.head_bwd(pre.detach().to(dev), Fd, W.detach().to(dev), gl, gp, Fd, pp, npart, rows, Fd, nc, True)
    report('head gpre', gp, pre.grad, tol=1e-5)
    gw = torch.zeros(nc, Fd, device=dev); gb = torch.zeros(nc, device=dev)
    lib.colsum(pp, npart, nc * Fd, gw, ld=nc * Fd + nc); lib.colsum(pp.view(-1)[nc * Fd:], npart, nc, gb, ld=nc * Fd + nc)
    report('head dW', gw, W.grad, tol=1e-5); report('head db', gb, b.grad, tol=1e-5)
    # adamw + clip
    n = 10007
    p = torch.randn(n); g = torch.randn(n) * 3
    sd = {'p': p.clone()}; st = {}
    gg = {'p': g.clone()}; gn = O.clip_grad_norm(gg, 1.0); O.adamw_step(sd, gg, st, 1e-3)   # lr 1e-3: the update (1e-3 per element) is 1000x the tolerance
    pd, gd, m, v = p.to(dev), g.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    parts = torch.zeros(64, device=dev); lib.sumsq_partial(gd, n, parts, 64)
    hyper = torch.tensor([1e-3, 1e-4, 0.9, 0.999, 1e-8, 1 - 0.9, 1 - 0.999, 1.0], device=dev); nc2 = torch.zeros(2, device=dev)
    lib.clip_coef(parts, 64, hyper, nc2); lib.adamw(pd, gd, m, v, n, hyper, nc2)
    report('grad norm', nc2[0], torch.tensor(gn)); report('adamw param', pd, sd['p'], tol=1e-6)

def model_case(name, signal_map, nc, B, S, missing):
    from wav2sleep_amd.wav2sleep import Wav2Sleep, SignalEncoders, MultiModalAttentionEmbedder, SequenceCNN
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=nc)
    sd = O.make_state_dict(cfg, seed=3)
    x, y = O.make_inputs(cfg, B, S, seed=5, missing=missing)
    model = Wav2Sleep(SignalEncoders(signal_map, 128, 'gelu', chunk_causal=False), MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, nhead=8),
                      SequenceCNN(128, dropout=0.0, norm='layer'), nc)
    model.load_state_dict(sd, strict=True)
    model.to(dev).train()
    model._ensure_flat(); model._engine.taps = {}
    xd = {k: v.to(dev) for k, v in x.items()}
    logits = model(xd)
    taps = {}
    want = O.forward(sd, cfg, x, taps)
    T = model._engine.taps
    for sig in signal_map:
        enc = signal_map[sig]
        for i in range(len(cfg.encoder_channels(sig))):
            k = f'signal_encoders.encoders.{enc}.cnn.{i}.out'
            if sum(1 for s2 in signal_map if signal_map[s2] == enc) == 1:
                # split-precision products: ~1e-5 relative per conv layer, 30 layers deep for the 10-block EOG encoders
                report(f'{name} {sig} block{i}', F.gelu(T[f'{sig}.pre.{i}']), cl(taps[k]), tol=4e-4)
    report(f'{name} mixer', T['mixer'], taps['mixer'], tol=2e-4)
    report(f'{name} seq', F.gelu(T['seq_pre']), taps['seq'], tol=2e-4)
    report(f'{name} logits', logits, want, tol=2e-4)
    agree = (logits.argmax(-1).cpu() == want.argmax(-1)).float().mean().item()
    print(f'     argmax agreement {agree:.4f}')
    loss = F.cross_entropy(logits.view(-1, nc), y.to(dev).view(-1).long(), ignore_index=-1)
    loss.backward()
    l0, _, grads = O.loss_and_grads(sd, cfg, x, y)
    report(f'{name} loss', loss, torch.tensor(l0), tol=1e-4)
    worst = 0
    for k, p in model.named_parameters():
        g = p.grad.cpu(); w = grads[k]
        rel = ((g - w).norm() / (w.norm() + 1e-12)).item()
        worst = max(worst, rel)
        if rel > 2e-3:
            print(f'     grad mismatch {k}: rel {rel:.3e} |want| {w.norm():.3e} |got| {g.norm():.3e}')
    RES.append((f'{name} grads', worst <= 2e-3)); print(f'{"OK  " if worst <= 2e-3 else "FAIL"} {name} grads worst rel-L2 {worst:.3e}', flush=True)

def t_model_c1(): model_case('c1', {'ECG': 'UNI'}, 4, 2, 4, None)
def t_model_c2(): model_case('c2', {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}, 4, 3, 4, {'ABD': [1], 'PPG': [2]})
def t_model_c4(): model_case('c4', {'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 2, 2, {'EOG-R': [0]})

def t_fused_split_precision():
    """bf16x3 fused backward (32 gradient channels) vs the fp32 fused kernel on the same tensors, ragged lengths."""
    for cg, ch in ((16, 16), (32, 32), (32, 16)):
        for stride, Lh in ((1, 1000), (2, 1322), (1, 128), (2, 4096)):
            B, Lg = 3, Lh // stride
            g = torch.randn(B, Lg, cg, device=dev); y = torch.randn(B, Lg, cg, device=dev); xin = torch.randn(B, Lh, ch, device=dev)
            st = torch.rand(B, cg, 2, device=dev) + 0.5; bst = torch.rand(B, cg, 2, device=dev) * 0.01; sti = torch.rand(B, ch, 2, device=dev) + 0.5
            add_even = torch.randn(B, Lh // 2, ch, device=dev)
            wb = torch.randn(ch, 3, cg, device=dev) / 7
            outs = []
            for sp in (False, True):
                tile = lib.bwd_fused_tile(cg, ch, stride, False, sp); nt = (Lh + tile - 1) // tile   # (256 / 128 vs the split-precision kernels' 254 / 126)
                gout = torch.zeros(B, Lh, ch, device=dev); part = torch.zeros(B, nt, 2, ch, device=dev)
                ns = min(B * nt, 5)
                slab = torch.zeros(ns * cg * ch * 3, device=dev); grad = torch.zeros(cg, ch, 3, device=dev)
                lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP, xin=xin, st_in=sti,
                              add_even=add_even, wb=wb, gout=gout, part=part, slab=slab, nslab=ns, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch,
                              stride=stride, split_precision=sp)
                lib.wgrad_reduce(slab, ns, grad, cg, ch, 3, 1, accumulate=False, layout=0)
                outs.append((gout, part.sum(1), grad))   # (the two kernels' tiles differ: the partials are compared as sums over a sample's tiles)
            for nm, a, b in zip(('gout', 'part', 'wgrad'), outs[1], outs[0]):
                # statistics partials (sums of gout and gout * n over a sample's tiles): since round 5 n = IN(x) stays in fp32 registers (rounds 3-4
                # kept an fp16 / LDS copy and needed 2e-4 ... 5e-4 here).  Measured worst case, round 6: 2.6e-6 of the scale at
                # 16 -> 16, 4.6e-6 at 32 -> 32, 6.0e-6 at 32 -> 16 -- fp32 summation order over the two kernels' different tiles.  Bar: 2e-5
                report(f'fused bf16x3 {cg}->{ch} s{stride} L{Lh} {nm}', a, b, tol=2e-5 if nm == 'part' else 2e-4)

def t_fused_residual_fold():
    """conv1 fused backward with the residual branch folded in vs (1x1 conv + add_even) and the separate downsample wgrad."""
    for cg, ch in ((16, 16), (32, 16)):
        for Lh in (1000, 256, 4098):
            B, Lg = 3, Lh
            g = torch.randn(B, Lg, cg, device=dev); y = torch.randn(B, Lg, cg, device=dev); xin = torch.randn(B, Lh, ch, device=dev)
            st = torch.rand(B, cg, 2, device=dev) + 0.5; bst = torch.rand(B, cg, 2, device=dev) * 0.01
            gpre = torch.randn(B, Lh // 2, cg, device=dev)
            wb = torch.randn(ch, 3, cg, device=dev) / 7; wd = torch.randn(ch, 1, cg, device=dev) / 5
            tile = lib.bwd_fused_tile(cg, ch, 1, True, True); nt = (Lh + tile - 1) // tile; ns = min(B * nt, 5)
            # reference arm: R = Wd^T gpre via a 1x1 conv, then add_even; downsample wgrad via w2s_wgrad on GELU(xin)
            Rr = torch.zeros(B, Lh // 2, ch, device=dev)
            lib.conv_forward(lib.conv_args(x=gpre, w=wd.view(ch, cg), y=Rr, B=B, L_in=Lh // 2, L_out=Lh // 2, cin=cg, cout=ch, taps=1, stride=1, pad=0))
            outs = []
            for fold in (False, True):
                gout = torch.zeros(B, Lh, ch, device=dev); slab = torch.zeros(ns * cg * ch * 3, device=dev); grad = torch.zeros(cg, ch, 3, device=dev)
                slab_d = torch.zeros(ns * cg * ch, device=dev) if fold else None
                lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD, xin=xin, st_in=None, add_even=None if fold else Rr, wb=wb, gout=gout,
                              part=None, slab=slab, nslab=ns, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch, stride=1, split_precision=True,
                              gpre=gpre if fold else None, wd=wd.view(ch, cg) if fold else None, slab_d=slab_d)
                lib.wgrad_reduce(slab, ns, grad, cg, ch, 3, 1, accumulate=False, layout=0)
                gd = torch.zeros(cg, ch, 1, device=dev)
                if fold:
                    lib.wgrad_reduce(slab_d, ns, gd, cg, ch, 1, 1, accumulate=False, layout=0)
                outs.append((gout, grad, gd))
            h = F.gelu(xin)[:, 0:2 * (Lh // 2):2, :]                                  # h[2u]
            want_gd = torch.einsum('buo,buc->oc', gpre.double().cpu(), h.double().cpu()).float().view(cg, ch, 1)
            report(f'fold {cg}->{ch} L{Lh} gout', outs[1][0], outs[0][0], tol=2e-4)
            report(f'fold {cg}->{ch} L{Lh} wgrad', outs[1][1], outs[0][1], tol=2e-4)
            report(f'fold {cg}->{ch} L{Lh} downsample wgrad', outs[1][2], want_gd, tol=2e-4)
            # ... and the previous block's conv3-backward statistics folded into the same kernel vs the w2s_gp_stats pre-pass
            y3p = torch.randn(B, Lh, ch, device=dev)
            st3 = torch.stack([torch.randn(B, ch, device=dev) * 0.1, torch.rand(B, ch, device=dev) + 0.5], dim=-1).contiguous()
            gout2 = torch.zeros(B, Lh, ch, device=dev); pt = torch.zeros(B, nt, 2, ch, device=dev)
            slab = torch.zeros(ns * cg * ch * 3, device=dev); slab_d = torch.zeros(ns * cg * ch, device=dev)
            lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD, xin=xin, st_in=None, add_even=None, wb=wb, gout=gout2, part=pt, slab=slab,
                          nslab=ns, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch, stride=1, split_precision=True, gpre=gpre, wd=wd.view(ch, cg), slab_d=slab_d,
                          y3p=y3p, st3p=st3)
            ntg = (Lh + 511) // 512
            pg = torch.zeros(B, ntg, 2, ch, device=dev)
            lib.gp_stats(gout2, y3p, st3, pg, B, Lh, ch, 512)
            report(f'fold {cg}->{ch} L{Lh} gout (with stats fold)', gout2, outs[1][0], tol=0)
            report(f'fold {cg}->{ch} L{Lh} conv3 statistics', pt.sum(1), pg.sum(1), tol=2e-4)
            if (cg, ch) == (16, 16):
                # ... and block 0's downsample weight gradient (this kernel's gout x every other sample of the raw signal) vs w2s_enc_first_dwd,
                # with and without the statistics partials (the two barrier paths of the epilogue)
                x0 = torch.randn(B, 2 * Lh, device=dev); x0[1, 4] = float('inf'); x0[2, 2 * Lh - 2] = float('-inf')
                ref = torch.zeros(6, 16, device=dev)
                lib.enc_first_dwd(x0, gout2, ref, 6, B, 2 * Lh)
                for with_part in (True, False):
                    go = torch.zeros(B, Lh, ch, device=dev); pt2 = torch.zeros(B, nt, 2, ch, device=dev); pw = torch.full((ns, 16), float('nan'), device=dev)
                    sl2 = torch.zeros_like(slab); sd2 = torch.zeros_like(slab_d)
                    lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD, xin=xin, st_in=None, add_even=None, wb=wb, gout=go, part=pt2 if with_part else None,
                                  slab=sl2, nslab=ns, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch, stride=1, split_precision=True, gpre=gpre, wd=wd.view(ch, cg), slab_d=sd2,
                                  y3p=y3p if with_part else None, st3p=st3 if with_part else None, x0=x0, part_wd=pw)
                    assert torch.equal(go, gout2) and torch.equal(sl2, slab) and torch.equal(sd2, slab_d) and (not with_part or torch.equal(pt2, pt)), 'fold changed other outputs'
                    report(f'fold 16->16 L{Lh} block-0 downsample wgrad (part={with_part})', pw.sum(0), ref.sum(0), tol=2e-5)

def _gelu_grad64(n):
    return 0.5 * (1 + torch.erf(n / 2 ** 0.5)) + n * torch.exp(-0.5 * n * n) / (2 * math.pi) ** 0.5

def Rr_fp64(gpre, wd):
    """data gradient of the 1x1 / stride-2 residual conv at the even input rows: R[u] = Wd^T gpre[u]  (wd: [ch][cg], the kernels' layout)"""
    return torch.einsum('buc,oc->buo', gpre.double().cpu(), wd.double().cpu())

def _bwd_wide_fp64(g, y, st, bst, x, sti, ev, w, stride, pad=1):
    """fp64 restatement of one encoder conv's backward as blocks.py:174-183 + autograd define it (channels-last device tensors in, CPU out):
    gy = IN-backward of g (x GELU'(n) first for the stride-2 conv3) with the GIVEN statistics / backward sums; H = GELU(IN(x)) (or GELU(x));
    dH, dW = autograd of F.conv1d(H, W, stride, padding=1); gout = (dH + ev at the even rows) x GELU'; sums = means of gout, gout x n_in."""
    g, y, st, bst, x = (t.double().cpu() for t in (g, y, st, bst, x))
    n = (y - st[:, None, :, 0]) * st[:, None, :, 1]
    gn = g * _gelu_grad64(n) if stride == 2 else g
    GY = st[:, None, :, 1] * (gn - bst[:, None, :, 0] - n * bst[:, None, :, 1])
    hn = (x - sti[:, None, :, 0].double().cpu()) * sti[:, None, :, 1].double().cpu() if sti is not None else x
    H = (0.5 * hn * (1 + torch.erf(hn / 2 ** 0.5))).transpose(1, 2).contiguous().requires_grad_(True)
    W = w.double().cpu().clone().requires_grad_(True)   # [cg][ch][3]
    Hp = F.pad(H, (pad, 2 - pad))   # pad = 2: causal (blocks.py's left-only padding)
    F.conv1d(Hp, W, stride=stride)[:, :, :x.shape[1] // stride].backward(GY.transpose(1, 2)[:, :, :x.shape[1] // stride].contiguous())
    dH = H.grad.transpose(1, 2).clone()
    if ev is not None:
        dH[:, 0:2 * ev.shape[1]:2, :] += ev.double().cpu()
    gout = dH * _gelu_grad64(hn)
    sums = torch.stack([gout.mean(1), (gout * hn).mean(1)], dim=1)   # [B][2][ch]
    return gout.float(), W.grad.float(), sums.float()

def t_bwd_wide():
    """One-pass backward of the 64-channel convs (bwd_wide.hip; stride 1 and the stride-2 conv3) against the two kernels it replaces on the
    same tensors: the conv_wide data gradient (instance-norm-backward prologue, GELU' epilogue, residual add, backward statistics) and the
    wgrad_wide weight gradient.  Same split-precision products in the same order: the bars are tight."""
    B = 3
    cases = [(64, 64, 1000, True, False, 1), (64, 64, 777, False, True, 1), (64, 32, 500, False, True, 1), (64, 64, 64, True, False, 1),
             (64, 32, 130, False, True, 1), (64, 64, 4098, True, False, 1), (64, 32, 2050, True, False, 1),
             (64, 64, 1000, True, False, 2), (64, 64, 64, True, False, 2), (64, 64, 4098, True, False, 2), (64, 64, 130, True, False, 2)]
    # pad = 2: the causal variant (left-only padding), every case again
    for (cg, ch, L, hst, add_even, stride), pad in [(c, 1) for c in cases] + [(c, 2) for c in cases]:
        Lg = L // stride
        g = torch.randn(B, Lg, cg, device=dev) * 0.1; y = torch.randn(B, Lg, cg, device=dev) * 2 + 0.2; x = torch.randn(B, L, ch, device=dev) * 1.3 - 0.1
        st = torch.stack([torch.randn(B, cg, device=dev) * 0.1, torch.rand(B, cg, device=dev) + 0.5], dim=-1).contiguous()
        bst = (torch.randn(B, cg, 2, device=dev) * 0.01).contiguous()
        sti = torch.stack([torch.randn(B, ch, device=dev) * 0.1, torch.rand(B, ch, device=dev) + 0.5], dim=-1).contiguous() if hst else None
        ev = torch.randn(B, L // 2, ch, device=dev) * 0.05 if add_even else None
        w = torch.randn(cg, ch, 3) / math.sqrt(3 * ch)
        wb = w.permute(1, 2, 0).contiguous().to(dev); wh, wl = lib.frag_major_planes(wb.view(ch, 3 * cg))
        pro_g = lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP
        # the separate kernels
        gout0 = torch.zeros(B, L, ch, device=dev)
        if stride == 1:
            a = lib.conv_args(x=g, x2=y, w=wb, w_hi=wh, w_lo=wl, y=gout0, B=B, L_in=L, L_out=L, cin=cg, cout=ch, taps=3, stride=1, pad=2 - pad, flip=1, pro=pro_g,
                              pro_stats=st, pro_bstats=bst, epi=lib.EPI_GP, aux=x, aux_stats=sti, add_even=ev)
        else:
            a = lib.conv_args(x=g, x2=y, w=wb, w_hi=wh, w_lo=wl, y=gout0, B=B, L_in=Lg, L_out=L, cin=cg, cout=ch, taps=3, stride=2, pad=pad, mode=lib.MODE_UP2,
                              pro=pro_g, pro_stats=st, pro_bstats=bst, epi=lib.EPI_GP, aux=x, aux_stats=sti)
        if True:
            t0 = lib.conv_tile_of(a); nt0 = (L + t0 - 1) // t0
            part0 = torch.zeros(B, nt0, 2, ch, device=dev); lib.set_part(a, part0)
            lib.conv_forward(a)
            kw = dict(g=g, g2=y, g_stats=st, g_bstats=bst, x=x, x_stats=sti, B=B, L_in=L, L_out=Lg, cin=ch, cout=cg, taps=3, stride=stride, pad=pad, pro_g=pro_g,
                      pro_h=lib.PRO_IN_GELU if hst else lib.PRO_GELU, split_precision=True)
            ns0 = min(5, (B * Lg + 255) // 256)
            slab0 = torch.zeros(ns0 * cg * ch * 3, device=dev); lib.wgrad(slab=slab0, nslab=ns0, **kw)
            gw0 = torch.zeros(cg, ch, 3, device=dev); lib.wgrad_reduce(slab0, ns0, gw0, cg, ch, 3, 1)
        # the fused kernel, with few and with many workgroups
        tile, groups = lib.bwd_wide_tile(cg, ch, stride), lib.bwd_wide_groups(cg, ch, stride)
        nt = (L + tile - 1) // tile
        RES.append((f'bwd_wide {cg}->{ch} s{stride} pad{pad} L{L} is taken', lib.bwd_wide_takes(B, L, cg, ch, stride, hst) and tile == 64))
        for ns in (min(5, B * nt), min(256, B * nt)):
            gout = torch.full((B, L, ch), float('nan'), device=dev); part = torch.full((B, nt * groups, 2, ch), float('nan'), device=dev)
            slab = torch.full((ns * cg * ch * 3,), float('nan'), device=dev)
            lib.bwd_wide(g=g, y=y, st_k=st, bst_k=bst, xin=x, st_in=sti, add_even=ev, w_hi=wh, w_lo=wl, gout=gout, part=part, slab=slab, nslab=ns, B=B, L=L,
                         cg=cg, ch=ch, stride=stride, pad=pad)
            gw = torch.zeros(cg, ch, 3, device=dev); lib.wgrad_reduce(slab, ns, gw, cg, ch, 3, 1)
            tag = f'bwd_wide {cg}->{ch} s{stride} pad{pad} L{L} hst{int(hst)} ev{int(add_even)} wgs{ns}'
            loose = 1
            report(tag + ' gout', gout, gout0, tol=2e-6 * loose)
            report(tag + ' statistics sums', part.sum(1), part0.sum(1), tol=2e-5 * loose)
            report(tag + ' wgrad', gw, gw0, tol=2e-5 * loose)
        if (cg, ch, L, stride) in ((64, 64, 1000, 1), (64, 64, 1000, 2), (64, 32, 500, 1)):
            # CPU arm (round-3 verdict): the same outputs against fp64 autograd of F.conv1d, independent of every sibling HIP kernel
            want_gout, want_gw, want_s = _bwd_wide_fp64(g, y, st, bst, x, sti, ev, w, stride, pad)
            report(tag + ' gout vs fp64 autograd', gout, want_gout, tol=1e-4)
            report(tag + ' wgrad vs fp64 autograd', gw, want_gw, tol=3e-4)
            if hst:
                report(tag + ' statistics sums vs fp64', part.sum(1) / L, want_s, tol=5e-5)
        if stride == 1 and not hst:   # conv1 of a block: the previous block's conv3-backward statistics folded in vs the w2s_gp_stats pre-pass
            y3p = torch.randn(B, L, ch, device=dev)
            st3 = torch.stack([torch.randn(B, ch, device=dev) * 0.1, torch.rand(B, ch, device=dev) + 0.5], dim=-1).contiguous()
            gout2 = torch.full((B, L, ch), float('nan'), device=dev); part2 = torch.full((B, nt * groups, 2, ch), float('nan'), device=dev)
            slab2 = torch.zeros(ns * cg * ch * 3, device=dev)
            lib.bwd_wide(g=g, y=y, st_k=st, bst_k=bst, xin=x, st_in=sti, add_even=ev, w_hi=wh, w_lo=wl, gout=gout2, part=part2, slab=slab2, nslab=ns, B=B, L=L,
                         cg=cg, ch=ch, stride=1, y3p=y3p, st3p=st3, pad=pad)
            ntg = (L + 511) // 512
            pg = torch.zeros(B, ntg, 2, ch, device=dev)
            lib.gp_stats(gout2, y3p, st3, pg, B, L, ch, 512)
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} gout with the statistics fold', gout2, gout, tol=0)
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} folded conv3 statistics', part2.sum(1), pg.sum(1), tol=2e-5)
        if stride == 1 and not hst and not (L & 1):   # ... and the block's residual branch folded in vs (1x1 conv -> add_even) + downsample wgrad
            gpre = torch.randn(B, L // 2, cg, device=dev) * 0.1
            wd = (torch.randn(ch, cg) / math.sqrt(cg)).to(dev); dh, dl = lib.frag_major_planes(wd)
            Rr = torch.zeros(B, L // 2, ch, device=dev)
            lib.conv_forward(lib.conv_args(x=gpre, w=wd, y=Rr, B=B, L_in=L // 2, L_out=L // 2, cin=cg, cout=ch, taps=1, stride=1, pad=0))
            outs = []
            for fold in (False, True):
                go = torch.full((B, L, ch), float('nan'), device=dev); pt = torch.full((B, nt * groups, 2, ch), float('nan'), device=dev)
                sl = torch.zeros(ns * cg * ch * 3, device=dev); sd = torch.zeros(ns * cg * ch, device=dev) if fold else None
                lib.bwd_wide(g=g, y=y, st_k=st, bst_k=bst, xin=x, st_in=None, add_even=None if fold else Rr, w_hi=wh, w_lo=wl, gout=go, part=pt, slab=sl, nslab=ns,
                             B=B, L=L, cg=cg, ch=ch, stride=1, gpre=gpre if fold else None, wd_hi=dh if fold else None, wd_lo=dl if fold else None, slab_d=sd, pad=pad)
                gwf = torch.zeros(cg, ch, 3, device=dev); lib.wgrad_reduce(sl, ns, gwf, cg, ch, 3, 1)
                gd = torch.zeros(cg, ch, 1, device=dev)
                if fold:
                    lib.wgrad_reduce(sd, ns, gd, cg, ch, 1, 1)
                outs.append((go, pt, gwf, gd))
            hcl = F.gelu(x)[:, 0:2 * (L // 2):2, :]
            want_gd = torch.einsum('buo,buc->oc', gpre.double().cpu(), hcl.double().cpu()).float().view(cg, ch, 1)
            RES.append((f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold is taken', lib.bwd_wide_takes(B, L, cg, ch, 1, False, rd=True)))
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold gout', outs[1][0], outs[0][0], tol=2e-5)
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold statistics', outs[1][1].sum(1), outs[0][1].sum(1), tol=2e-4)
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold conv wgrad identical', outs[1][2], outs[0][2], tol=0)
            report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold downsample wgrad', outs[1][3], want_gd, tol=3e-4)
            if (cg, ch, L) == (64, 32, 500):   # CPU arm of the residual-fold form: (W^T gy + Wd^T gpre at the even rows) x GELU'(pin), fp64
                want_gout, want_gw, _ = _bwd_wide_fp64(g, y, st, bst, x, None, Rr_fp64(gpre, wd), w, 1, pad)
                report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold gout vs fp64 autograd', outs[1][0], want_gout, tol=1e-4)
                report(f'bwd_wide {cg}->{ch} pad{pad} L{L} residual fold wgrad vs fp64 autograd', outs[1][2], want_gw, tol=3e-4)
    RES.append(('bwd_wide refuses a batch whose statistics tables do not fit its LDS', not lib.bwd_wide_takes(48, 640, 64, 64, 1, True) and lib.bwd_wide_takes(40, 640, 64, 64, 1, False)))

def t_offset_channels():
    """Channels with a large DC offset against their spread.  Since round 5 the on-load transforms use per-tile coefficient
    forms -- n = x r + (-m r), gy = r g + (-r^2 s2) y + r (r s2 m - s1) -- whose absolute error in n is ~|m| r 2^-24 instead of the textbook
    (x - m) r's ~|n| 2^-24.  An fp32 MEAN already carries |m| 2^-24, i.e. the same |m| r 2^-24 in n, whatever form follows: the coefficient
    form can at most double what the fp32 statistics contract costs.  Measured here against fp64 for the forward statistics consumers
    (persistent <= 32-channel kernel, role-split 64-channel kernel, generic kernel) and the instance-norm backward of the fused backward kernel,
    beside the textbook fp32 form evaluated by torch on the same fp32 statistics; bar: <= 4 x the textbook form's error + the split-precision floor."""
    B = 2
    for ratio, mean, std in ((0, 0.0, 1.0), (30, 3.0, 0.1), (1e3, 100.0, 0.1), (1e4, 1000.0, 0.1)):
        for (cin, cout, L, kind) in ((16, 16, 1500, 'fused'), (64, 64, 700, 'wide'), (16, 32, 600, 'generic')):
            sign = torch.where(torch.arange(cin) % 2 == 0, 1.0, -1.0)
            x = (torch.randn(B, L, cin) * std + mean * sign).to(dev)
            x64 = x.double().cpu()
            m64, v64 = x64.mean(1), x64.var(1, unbiased=False)
            r64 = 1 / torch.sqrt(v64 + 1e-2)
            st = torch.stack([m64, r64], -1).float().contiguous().to(dev)     # what w2s_stats_finalize hands a consumer: fp32 (mean, rstd)
            w = torch.randn(cout, cin, 3) / math.sqrt(3 * cin)
            conv64 = lambda n: cl(F.conv1d(F.gelu(n).transpose(1, 2), w.double(), padding=1))
            want = conv64((x64 - m64[:, None]) * r64[:, None])
            text = conv64(((x.cpu() - st[..., 0].cpu()[:, None]) * st[..., 1].cpu()[:, None]).double())   # textbook form in fp32 on the fp32 statistics
            y = torch.zeros(B, L, cout, device=dev)
            wp = pack_fwd(w).to(dev)
            if kind == 'fused':
                t = lib.conv_fwd_fused_tile(cin, cout, 1); part = torch.zeros(B, (L + t - 1) // t, 2, cout, device=dev)
                lib.conv_fwd_fused(x=x, w=wp, st_in=st, w1=None, y=y, part=part, B=B, L_in=L, L_out=L, cin=cin, cout=cout, stride=1, pro=lib.PRO_IN_GELU, nwg=7)
            else:
                wh, wl = lib.frag_major_planes(wp.view(cout, 3 * cin)) if cin >= 32 else (None, None)
                a = lib.conv_args(x=x, w=wp, w_hi=wh, w_lo=wl, y=y, B=B, L_in=L, L_out=L, cin=cin, cout=cout, taps=3, stride=1, pad=1, pro=lib.PRO_IN_GELU,
                                  pro_stats=st, epi=lib.EPI_STATS)
                t = lib.conv_tile_of(a); part = torch.zeros(B, (L + t - 1) // t, 2, cout, device=dev); lib.set_part(a, part)
                lib.conv_forward(a)
            scale = float(want.abs().max())
            ek, et = float((y.double().cpu() - want).abs().max()), float((text - want).abs().max())
            ok = ek <= 4 * et + 5e-5 * scale
            RES.append((f'offset fwd {kind} ratio {ratio:g}', ok))
            print(f'{"OK  " if ok else "FAIL"} offset |mean|/std {ratio:<6g} forward {kind:8s} {cin}->{cout}: kernel err {ek:.3e}, textbook-fp32 err {et:.3e} (scale {scale:.3e})', flush=True)
        # instance-norm backward (W2S_PRO_INBWD) + the input-side IN + GELU of the fused backward kernel, both sides offset
        cg = ch = 16; L = 1500
        sign = torch.where(torch.arange(cg) % 2 == 0, 1.0, -1.0)
        g = torch.randn(B, L, cg, device=dev) * 0.1
        yk = (torch.randn(B, L, cg) * std + mean * sign).to(dev); xin = (torch.randn(B, L, ch) * std - mean * sign).to(dev)

        def stats(t):
            t64 = t.double().cpu()
            return torch.stack([t64.mean(1), 1 / torch.sqrt(t64.var(1, unbiased=False) + 1e-2)], -1)
        st64, sti64 = stats(yk), stats(xin)
        st, sti = st64.float().contiguous().to(dev), sti64.float().contiguous().to(dev)
        n64 = (yk.double().cpu() - st64[:, None, :, 0]) * st64[:, None, :, 1]
        g64 = g.double().cpu()
        bst64 = torch.stack([g64.mean(1), (g64 * n64).mean(1)], -1)          # the backward sums autograd's instance norm uses
        bst = bst64.float().contiguous().to(dev)
        w = torch.randn(cg, ch, 3) / math.sqrt(3 * ch)
        wb = w.permute(1, 2, 0).contiguous().to(dev)
        tile = lib.bwd_fused_tile(cg, ch, 1, False, True); nt = (L + tile - 1) // tile
        gout = torch.zeros(B, L, ch, device=dev); part = torch.zeros(B, nt, 2, ch, device=dev)
        ns = min(B * nt, 5); slab = torch.zeros(ns * cg * ch * 3, device=dev); gw = torch.zeros(cg, ch, 3, device=dev)
        lib.bwd_fused(g=g, y=yk, st_k=st, bst_k=bst, pro=lib.PRO_INBWD, xin=xin, st_in=sti, add_even=None, wb=wb, gout=gout, part=part, slab=slab, nslab=ns,
                      B=B, Lg=L, Lh=L, cg=cg, ch=ch, stride=1, split_precision=True)
        lib.wgrad_reduce(slab, ns, gw, cg, ch, 3, 1, accumulate=False, layout=0)
        want_gout, want_gw, _ = _bwd_wide_fp64(g, yk, st64, bst64, xin, sti64, None, w, 1)                          # fp64 statistics: the true values
        text_gout, text_gw, _ = _bwd_wide_fp64(g, yk, st.cpu(), bst.cpu(), xin, sti.cpu(), None, w, 1)              # fp32-rounded statistics, exact arithmetic
        for nm, got, want, text in (('gout', gout, want_gout, text_gout), ('wgrad', gw, want_gw, text_gw)):
            scale = float(want.abs().max())
            ek, et = float((got.cpu() - want).abs().max()), float((text - want).abs().max())
            ok = ek <= 4 * et + 3e-4 * scale
            RES.append((f'offset bwd {nm} ratio {ratio:g}', ok))
            print(f'{"OK  " if ok else "FAIL"} offset |mean|/std {ratio:<6g} fused backward 16->16 {nm:5s}: kernel err {ek:.3e}, fp32-statistics err {et:.3e} (scale {scale:.3e})', flush=True)


def t_grad_fp16_chain():
    """The fp16 gradient-chain forms of the fused backward kernels (gmode 1 / 2, include/w2s.h) against the fp32 kernels: a power-of-two
    scale and fp32 arithmetic make the relation EXACT -- fed with the dequantised fp16 gradient the fp32 kernel must give bit-identical
    statistics partials and weight-gradient slabs, and an output whose fp16 rounding (x scale) is the fp16 kernel's output bit for bit;
    the output header holds the scale derived from the input maxima and the output's own maximum."""
    import math

    def quant(t, ref=None):
        a = float(t.abs().max()) if ref is None else ref
        s = 2.0 ** (9 - math.frexp(a)[1])
        h = (t * s).half()
        return h, h.float() / s, torch.tensor([s, float((h.float() / s).abs().max())], device=dev)

    cases = [(16, 16, 1, 0, 0, 2), (16, 16, 2, 0, 0, 2), (32, 32, 1, 0, 0, 2), (32, 32, 2, 0, 0, 2), (32, 32, 2, 0, 0, 1), (16, 16, 2, 0, 0, 1),
             (16, 16, 1, 1, 0, 2), (32, 16, 1, 1, 0, 2), (16, 16, 1, 0, 1, 2)]
    for cg, ch, stride, rd, first, gmode in cases:
        for Lh in (1000, 4098):
            B, Lg = 3, Lh // stride
            g = torch.randn(B, Lg, cg, device=dev) * 3e-4; y = torch.randn(B, Lg, cg, device=dev)
            xin = torch.randn(B, Lh, device=dev) if first else torch.randn(B, Lh, ch, device=dev)
            st = torch.rand(B, cg, 2, device=dev) + 0.5; bst = torch.rand(B, cg, 2, device=dev) * 1e-6; sti = torch.rand(B, ch, 2, device=dev) + 0.5
            wb = torch.randn(ch, 3, cg, device=dev) / 7; wd = torch.randn(ch, cg, device=dev) / 5; w1 = torch.randn(16, 1, 3, device=dev) / 2
            gpre = torch.randn(B, Lh // 2, cg, device=dev) * 1e-3 if rd else None
            add_even = torch.randn(B, Lh // 2, ch, device=dev) * 1e-3 if (not rd and stride == 1 and not first) else None
            if gmode == 2:
                g16, g32, hg = quant(g)
            else:
                g16, g32, hg = None, g, torch.tensor([1.0, float(g.abs().max())], device=dev)
            p16 = p32 = hp = None
            if rd:
                p16, p32, hp = quant(gpre)
            tile = lib.bwd_fused_tile(cg, ch, stride, bool(rd), True); nt = (Lh + tile - 1) // tile; ns = min(B * nt, 5)
            pro = lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP
            res = []
            for mode in (0, gmode):
                gout = torch.zeros(B, Lh, ch, device=dev, dtype=torch.float16 if mode else torch.float32)
                part = torch.zeros(B, nt, 2, ch, device=dev); slab = torch.zeros(ns * cg * ch * 3, device=dev)
                slab_d = torch.zeros(ns * cg * ch, device=dev) if rd else None
                ho = torch.zeros(2, device=dev)
                lib.bwd_fused(g=(g16 if mode == 2 else g32), y=y, st_k=st, bst_k=bst, pro=pro, xin=xin, st_in=None if rd else sti, add_even=add_even, wb=wb,
                              gout=gout, part=part, slab=slab, nslab=ns, B=B, Lg=Lg, Lh=Lh, cg=cg, ch=ch, stride=stride, split_precision=True,
                              gpre=(p16 if mode == 2 else p32) if rd else None, wd=wd if rd else None, slab_d=slab_d, w1=w1 if first else None,
                              gmode=mode, hdr_g=hg if mode else None, hdr_p=hp if (mode and rd) else None, hdr_o=ho if mode else None)
                res.append((gout, part, slab, slab_d, ho))
            tag = f'grad fp16 {cg}->{ch} s{stride} rd{rd} first{first} gmode{gmode} L{Lh}'
            (o32, pt32, sl32, sd32, _), (o16, pt16, sl16, sd16, ho) = res
            ref = max(float(hg[1]), float(hp[1]) if rd else 0.0)
            s_out = 2.0 ** (9 - math.frexp(ref)[1])
            report(tag + ' partials identical', pt16, pt32, tol=0)
            report(tag + ' weight-gradient slabs identical', sl16, sl32, tol=0)
            if rd:
                report(tag + ' downsample slabs identical', sd16, sd32, tol=0)
            report(tag + ' gout = fp16(scale * fp32 gout)', o16.float(), (o32 * s_out).half().float(), tol=0)
            report(tag + ' header {scale, max}', ho, torch.tensor([s_out, float(o32.abs().max())]), tol=0)
    # the two small readers of chain tensors: the conv3 pre-pass (also the chain's entry: publishes {1, max}) and the first-layer weight gradients
    B, L, c = 2, 3000, 32
    g = torch.randn(B, L, c, device=dev) * 2e-5; y = torch.randn(B, L, c, device=dev); st = torch.rand(B, c, 2, device=dev) + 0.5
    g16, g32, hg = quant(g)
    nt = (L + 511) // 512
    pa = torch.zeros(B, nt, 2, c, device=dev); pb = torch.zeros_like(pa); pc = torch.zeros_like(pa); hm = torch.zeros(2, device=dev)
    lib.gp_stats(g32, y, st, pa, B, L, c, 512)
    lib.gp_stats(g16, y, st, pb, B, L, c, 512, hdr_g=hg)
    lib.gp_stats(g32, y, st, pc, B, L, c, 512, hdr_amax=hm)
    report('grad fp16 gp_stats (fp16 in) identical', pb, pa, tol=0)
    report('grad fp16 gp_stats (entry) identical + header', torch.cat([pc.flatten(), hm]), torch.cat([pa.flatten(), torch.tensor([1.0, float(g32.abs().max())], device=dev)]), tol=0)
    B, L, c = 2, 1500, 16
    x = torch.randn(B, L, device=dev); w1 = torch.randn(16, 1, 3, device=dev) / 2
    st1 = torch.rand(B, c, 2, device=dev) + 0.5; bs1 = torch.rand(B, c, 2, device=dev) * 1e-6
    n16, n32, hn = quant(torch.randn(B, L, c, device=dev) * 1e-4); q16, q32, hq = quant(torch.randn(B, L // 2, c, device=dev) * 1e-3)
    sa = torch.zeros(8, 64, device=dev); sb = torch.zeros(8, 64, device=dev)
    lib.enc_first_bwd(x, n32, None, st1, bs1, q32, sa, 8, B, L, c, w1=w1)
    lib.enc_first_bwd(x, n16, None, st1, bs1, q16, sb, 8, B, L, c, w1=w1, hdr_n=hn, hdr_p=hq)
    report('grad fp16 first-layer weight gradients identical', sb, sa, tol=0)

def t_first_layer_recompute():
    """W2S_PRO_FIRST flow: the consumers of block 0's conv1 output recompute it from the raw signal -- against the same
    kernels fed with the stored tensor (bit-level arithmetic differs only in the 3-FMA conv itself)."""
    B, L, c = 2, 1500, 16
    x = torch.randn(B, L, device=dev); x[0, 7] = float('inf')
    w1 = torch.randn(16, 1, 3, device=dev) / 2
    w2 = torch.randn(16, 16, 3, device=dev) / 7
    tile = 1024
    nt = (L + tile - 1) // tile
    y1 = torch.zeros(B, L, c, device=dev); part = torch.zeros(B, nt, 2, c, device=dev)
    lib.enc_first_fwd(x, w1, y1, part, B, L, c, tile)
    part2 = torch.zeros_like(part)
    lib.enc_first_fwd(x, w1, None, part2, B, L, c, tile)                       # statistics only
    report('first: stats-only partials (closed form from 9 signal sums)', part2, part, tol=2e-5)
    st1 = torch.zeros(B, c, 2, device=dev)
    lib.stats_finalize(part, B, nt, c, L, 1e-2, 0, st1)
    wp = pack_fwd(w2).to(dev)
    outs = []
    for first in (False, True):
        for planes in (None, lib.frag_major_planes(wp.view(c, 3 * c))):
            y2 = torch.zeros(B, L, c, device=dev)
            t2 = lib.conv_tile(c, c, 3, 1, lib.MODE_CONTIG, B, L)
            p2 = torch.zeros(B, (L + t2 - 1) // t2, 2, c, device=dev)
            kw = dict(w_hi=planes[0], w_lo=planes[1]) if planes else {}
            lib.conv_forward(lib.conv_args(x=x if first else y1, x2=w1 if first else None, w=wp, y=y2, B=B, L_in=L, L_out=L, cin=c, cout=c, taps=3,
                                           stride=1, pad=1, pro=lib.PRO_FIRST if first else lib.PRO_IN_GELU, pro_stats=st1, epi=lib.EPI_STATS,
                                           part=p2, ldx=4 if first else None, **kw))
            outs.append((y2, p2))
    report('first: conv2 fp32 recompute vs stored', outs[2][0], outs[0][0], tol=2e-5)
    report('first: conv2 bf16x3 recompute vs stored', outs[3][0], outs[1][0], tol=2e-5)
    report('first: conv2 bf16x3 vs fp32', outs[1][0], outs[0][0], tol=5e-5)
    report('first: conv2 stats partials', outs[3][1], outs[1][1], tol=1e-4)
    # fused conv2 backward: xin = stored y1 vs raw signal
    g = torch.randn(B, L, c, device=dev); y2 = outs[0][0]
    st2 = torch.rand(B, c, 2, device=dev) + 0.5; bst = torch.rand(B, c, 2, device=dev) * 0.01
    wb = torch.randn(c, 3, c, device=dev) / 7
    tilef = lib.bwd_fused_tile(c, c, 1, False, True); ntf = (L + tilef - 1) // tilef; ns = min(B * ntf, 5)
    res = []
    for first in (False, True):
        gout = torch.zeros(B, L, c, device=dev); pt = torch.zeros(B, ntf, 2, c, device=dev)
        slab = torch.zeros(ns * c * c * 3, device=dev); grad = torch.zeros(c, c, 3, device=dev)
        lib.bwd_fused(g=g, y=y2, st_k=st2, bst_k=bst, pro=lib.PRO_INBWD, xin=x if first else y1, st_in=st1, add_even=None, wb=wb, gout=gout, part=pt,
                      slab=slab, nslab=ns, B=B, Lg=L, Lh=L, cg=c, ch=c, stride=1, split_precision=True, w1=w1 if first else None)
        lib.wgrad_reduce(slab, ns, grad, c, c, 3, 1, accumulate=False, layout=0)
        res.append((gout, pt, grad))
    for nm, a, b in zip(('gout', 'part', 'wgrad'), res[1], res[0]):
        report(f'first: fused conv2 backward {nm}', a, b, tol=5e-5)
    # first-layer weight gradients: stored vs recomputed y1
    gpre = torch.randn(B, L // 2, c, device=dev); bs1 = torch.rand(B, c, 2, device=dev) * 0.01
    sl = []
    for first in (False, True):
        slab = torch.zeros(8, 64, device=dev)
        lib.enc_first_bwd(x, g, None if first else y1, st1, bs1, gpre, slab, 8, B, L, c, w1=w1)
        sl.append(slab.sum(0))
    report('first: conv1 / downsample weight gradients', sl[1], sl[0], tol=5e-5)
    # conv1's weight gradient folded into conv2's backward kernel (w2s_bwd_fused_w1 + w2s_enc_first_wgrad; the gradient tensor gn1 is never
    # stored) against the two-pass form: w2s_bwd_fused writes gn1, its statistics are finalised, w2s_enc_first_bwd reads it back
    for pad, causal in ((1, False), (2, True)):
        for Lx in (1500, 256, 70000):
            xx = torch.randn(B, Lx, device=dev) + 0.3; xx[0, 7] = float('inf'); xx[1, Lx - 1] = float('-inf')
            gg = torch.randn(B, Lx, c, device=dev); yy2 = torch.randn(B, Lx, c, device=dev)
            ntx = (Lx + tile - 1) // tile
            p1 = torch.zeros(B, ntx, 2, c, device=dev)
            xm = torch.zeros(B, ntx, 9, device=dev)
            lib.enc_first_stats(xx, w1, p1, xm, B, Lx, tile, causal=causal)
            p1b = torch.zeros_like(p1)
            lib.enc_first_fwd(xx, w1, None, p1b, B, Lx, c, tile, causal=causal)
            assert torch.equal(p1, p1b), 'w2s_enc_first_stats differs from the statistics-only w2s_enc_first_fwd'
            s1 = torch.zeros(B, c, 2, device=dev)
            lib.stats_finalize(p1, B, ntx, c, Lx, 1e-2, 0, s1)
            ntf = (Lx + tilef - 1) // tilef; ns = min(B * ntf, 5)
            gout = torch.zeros(B, Lx, c, device=dev); pt = torch.zeros(B, ntf, 2, c, device=dev); slab = torch.zeros(ns * c * c * 3, device=dev)
            lib.bwd_fused(g=gg, y=yy2, st_k=st2, bst_k=bst, pro=lib.PRO_INBWD, xin=xx, st_in=s1, add_even=None, wb=wb, gout=gout, part=pt, slab=slab, nslab=ns,
                          B=B, Lg=Lx, Lh=Lx, cg=c, ch=c, stride=1, split_precision=True, w1=w1, pad=pad)
            b1 = torch.zeros(B, c, 2, device=dev)
            lib.stats_finalize(pt, B, ntf, c, Lx, 0.0, 1, b1)
            gp = torch.randn(B, Lx // 2, c, device=dev)
            ref = torch.zeros(8, 64, device=dev)
            lib.enc_first_bwd(xx, gout, None, s1, b1, gp, ref, 8, B, Lx, c, w1=w1, causal=causal)
            ref = ref.sum(0)
            pt2 = torch.zeros_like(pt); slab2 = torch.zeros_like(slab); pw = torch.full((B, ntf, 48), float('nan'), device=dev)
            lib.bwd_fused(g=gg, y=yy2, st_k=st2, bst_k=bst, pro=lib.PRO_INBWD, xin=xx, st_in=s1, add_even=None, wb=wb, gout=None, part=pt2, slab=slab2, nslab=ns,
                          B=B, Lg=Lx, Lh=Lx, cg=c, ch=c, stride=1, split_precision=True, w1=w1, pad=pad, part_w1=pw)
            assert torch.equal(pt2, pt) and torch.equal(slab2, slab), 'the fold changed the kernel\'s other outputs'
            dw = torch.zeros(B, 48, device=dev)
            lib.enc_first_wgrad(xm, ntx, w1, pw, s1, b1, dw, B, ntf)
            report(f'first: folded conv1 weight gradient pad={pad} L={Lx}', dw.sum(0), ref[:48], tol=2e-4)
            if not (Lx & 1):
                sd = torch.zeros(7, 16, device=dev)
                lib.enc_first_dwd(xx, gp, sd, 7, B, Lx)
                report(f'first: downsample weight gradient alone pad={pad} L={Lx}', sd.sum(0), ref[48:], tol=2e-5)


def t_batched_entry_points():
    """w2s_wgrad_reduce_batch / w2s_colsum_batch / w2s_repack_batch against their single-job forms."""
    jobs, want = [], []
    for (cout, cin, taps, ns) in ((16, 16, 3, 7), (64, 64, 3, 9), (128, 128, 1, 5), (32, 16, 1, 33), (128, 64, 3, 4)):
        slab = torch.randn(ns * cout * cin * taps, device=dev)
        g1 = torch.randn(cout, cin, taps, device=dev); g2 = g1.clone()
        acc = (cout == 64)
        lib.wgrad_reduce(slab, ns, g1, cout, cin, taps, 1, accumulate=acc, layout=0)
        jobs.append((slab, ns, g2, cout, cin, taps, 1, acc, 0)); want.append((g1, g2))
    lib.wgrad_reduce_batch(jobs)
    for i, (a, b) in enumerate(want):
        report(f'batch: slab reduce job {i}', b, a, tol=0)
    cj, cw = [], []
    for (nparts, C, ld) in ((40, 128, 128), (7, 4, 516), (300, 512, 512), (64, 48, 64)):
        part = torch.randn(nparts, ld, device=dev); o1 = torch.randn(C, device=dev); o2 = o1.clone()
        lib.colsum(part, nparts, C, o1, accumulate=(C == 4), ld=ld)
        cj.append((part, nparts, C, o2, C == 4, ld)); cw.append((o1, o2))
    lib.colsum_batch(cj)
    for i, (a, b) in enumerate(cw):
        report(f'batch: column sum job {i}', b, a, tol=0)
    rj, rw = [], []
    for (cout, cin, taps) in ((64, 32, 3), (128, 128, 7), (32, 16, 3), (16, 16, 1), (128, 64, 1)):
        w = torch.randn(cout, cin, taps, device=dev)
        f = torch.zeros(cout * cin * taps, device=dev); bw = torch.zeros_like(f)
        kf = 32 * ((taps + 1) // 2) if cin == 16 else taps * cin
        fh = torch.zeros(cout * kf, device=dev, dtype=torch.bfloat16); fl = torch.zeros_like(fh)
        bh = bl = None
        if cin >= 32 and cout >= 32:
            bh = torch.zeros(cin * taps * cout, device=dev, dtype=torch.bfloat16); bl = torch.zeros_like(bh)
        rj.append((w, f, bw, fh, fl, bh, bl, cout, cin, taps)); rw.append(w)
    lib.repack_batch(rj)
    for (w, f, bw, fh, fl, bh, bl, cout, cin, taps) in rj:
        wf = w.permute(0, 2, 1).contiguous()                     # [o][j][c]
        wb_ = w.permute(1, 2, 0).contiguous()                    # [c][j][o]
        report(f'batch: repack fwd {cin}->{cout} k{taps}', f.view_as(wf), wf, tol=0)
        report(f'batch: repack bwd {cin}->{cout} k{taps}', bw.view_as(wb_), wb_, tol=0)
        h, l = lib.frag_major_planes(wf.view(cout, taps * cin))
        report(f'batch: fragment-major hi plane {cin}->{cout} k{taps}', fh.float(), h.float(), tol=0)
        report(f'batch: fragment-major lo plane {cin}->{cout} k{taps}', fl.float(), l.float(), tol=0)
        if bh is not None:
            h, l = lib.frag_major_planes(wb_.view(cin, taps * cout))
            report(f'batch: fragment-major dgrad hi plane {cin}->{cout} k{taps}', bh.float(), h.float(), tol=0)


def t_fwd_fused():
    """persistent split-precision forward conv (<= 32 channels) vs w2s_conv_forward with the same prologue / EPI_STATS."""
    B = 3
    for (cin, cout, stride, L) in ((16, 16, 1, 1000), (16, 16, 2, 1322), (16, 32, 1, 700), (32, 32, 1, 515), (32, 32, 2, 1026), (16, 16, 1, 64)):
        Lo = L // stride
        x = torch.randn(B, L, cin, device=dev)
        w = torch.randn(cout, cin, 3, device=dev) / math.sqrt(3 * cin)
        st = torch.stack([torch.randn(B, cin, device=dev) * 0.1, torch.rand(B, cin, device=dev) + 0.5], dim=-1).contiguous()
        wp = pack_fwd(w).to(dev)
        for pro in (lib.PRO_GELU, lib.PRO_IN_GELU):
            t1 = lib.conv_tile(cin, cout, 3, stride, lib.MODE_CONTIG, B, Lo)
            y1 = torch.zeros(B, Lo, cout, device=dev); p1 = torch.zeros(B, (Lo + t1 - 1) // t1, 2, cout, device=dev)
            lib.conv_forward(lib.conv_args(x=x, w=wp, y=y1, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=3, stride=stride, pad=1, pro=pro,
                                           pro_stats=st, epi=lib.EPI_STATS, part=p1))
            t2 = lib.conv_fwd_fused_tile(cin, cout, stride)
            y2 = torch.zeros(B, Lo, cout, device=dev); p2 = torch.zeros(B, (Lo + t2 - 1) // t2, 2, cout, device=dev)
            lib.conv_fwd_fused(x=x, w=wp, st_in=st if pro != lib.PRO_GELU else None, w1=None, y=y2, part=p2, B=B, L_in=L, L_out=Lo, cin=cin,
                               cout=cout, stride=stride, pro=pro, nwg=5)
            report(f'fwd fused {cin}->{cout} s{stride} L{L} pro{pro}', y2, y1, tol=5e-5)
            report(f'fwd fused {cin}->{cout} s{stride} L{L} pro{pro} stats', p2.sum(1), p1.sum(1), tol=2e-4)
    # first-layer recompute flavour
    L, c = 1500, 16
    xs = torch.randn(B, L, device=dev); xs[1, 3] = float('-inf')
    w1 = torch.randn(16, 1, 3, device=dev) / 2; w2 = torch.randn(16, 16, 3, device=dev) / 7
    st = torch.stack([torch.randn(B, c, device=dev) * 0.1, torch.rand(B, c, device=dev) + 0.5], dim=-1).contiguous()
    wp = pack_fwd(w2).to(dev)
    t1 = lib.conv_tile(c, c, 3, 1, lib.MODE_CONTIG, B, L)
    y1 = torch.zeros(B, L, c, device=dev); p1 = torch.zeros(B, (L + t1 - 1) // t1, 2, c, device=dev)
    lib.conv_forward(lib.conv_args(x=xs, x2=w1, w=wp, y=y1, B=B, L_in=L, L_out=L, cin=c, cout=c, taps=3, stride=1, pad=1, pro=lib.PRO_FIRST,
                                   pro_stats=st, epi=lib.EPI_STATS, part=p1, ldx=4))
    t2 = lib.conv_fwd_fused_tile(c, c, 1)
    y2 = torch.zeros(B, L, c, device=dev); p2 = torch.zeros(B, (L + t2 - 1) // t2, 2, c, device=dev)
    lib.conv_fwd_fused(x=xs, w=wp, st_in=st, w1=w1, y=y2, part=p2, B=B, L_in=L, L_out=L, cin=c, cout=c, stride=1, pro=lib.PRO_FIRST, nwg=7)
    report('fwd fused first-layer recompute', y2, y1, tol=5e-5)
    report('fwd fused first-layer recompute stats', p2.sum(1), p1.sum(1), tol=2e-4)


def t_stats_partition():
    """The persistent statistics producers take their tiles BLOCKED (a workgroup owns a contiguous run of the (sample, tile) list): the
    per-tile partials -- and everything else they write -- must not depend on how many workgroups share the list (1 ... more than there are
    tiles; runs that straddle samples), and the finalised statistics must agree with torch.  (Rounds 1 and 4 also finalised inside the
    producers; removed in round 5 -- docs/lab_notes_r5.md.)"""
    B = 5
    for (cin, cout, L, stride) in ((16, 16, 5000, 1), (32, 32, 3001, 1), (16, 32, 2560, 1), (32, 32, 4096, 2), (16, 16, 300, 1)):
        Lo = L // stride
        if L % stride:
            continue
        x = torch.randn(B, L, cin, device=dev) * 2 + 0.3
        w = torch.randn(cout, 3, cin, device=dev) / math.sqrt(3 * cin)
        st = torch.stack([torch.randn(B, cin, device=dev) * 0.1, torch.rand(B, cin, device=dev) + 0.5], dim=-1).contiguous()
        tile = lib.conv_fwd_fused_tile(cin, cout, stride); nt = (Lo + tile - 1) // tile
        outs = []
        for nwg in (1, 3, 7, 64, 1024):
            y = torch.zeros(B, Lo, cout, device=dev); part = torch.full((B, nt, 2, cout), float('nan'), device=dev)
            lib.conv_fwd_fused(x=x, w=w, st_in=st, w1=None, y=y, part=part, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, stride=stride, pro=lib.PRO_IN_GELU, nwg=nwg)
            outs.append((y, part))
        tag = f'blocked partition fwd {cin}->{cout} L{L} s{stride}'
        RES.append((tag + ': y and partials independent of the grid', all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])))
        ref = torch.zeros(B, cout, 2, device=dev); lib.stats_finalize(outs[0][1], B, nt, cout, Lo, 1e-2, 0, ref)
        y0 = outs[0][0]
        report(tag + ' statistics vs torch', ref, torch.stack([y0.mean(1), 1 / torch.sqrt(y0.var(1, unbiased=False) + 1e-2)], dim=-1), tol=2e-5)
    for (cg, ch, L, stride, wide) in ((16, 16, 3000, 1, False), (32, 32, 2050, 2, False), (32, 16, 1000, 1, False), (64, 64, 1000, 1, True), (64, 64, 1026, 2, True), (64, 32, 500, 1, True)):
        Lg = L // stride
        g = torch.randn(B, Lg, cg, device=dev) * 0.1; y = torch.randn(B, Lg, cg, device=dev) * 2 + 0.2; xi = torch.randn(B, L, ch, device=dev) * 1.3 - 0.1
        st = torch.stack([torch.randn(B, cg, device=dev) * 0.1, torch.rand(B, cg, device=dev) + 0.5], dim=-1).contiguous()
        bst = (torch.randn(B, cg, 2, device=dev) * 0.01).contiguous()
        sti = torch.stack([torch.randn(B, ch, device=dev) * 0.1, torch.rand(B, ch, device=dev) + 0.5], dim=-1).contiguous()
        wb = (torch.randn(cg, ch, 3) / math.sqrt(3 * ch)).permute(1, 2, 0).contiguous().to(dev); wh, wl = lib.frag_major_planes(wb.view(ch, 3 * cg))
        pro_g = lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP
        if wide:
            tile, groups = lib.bwd_wide_tile(cg, ch, stride), lib.bwd_wide_groups(cg, ch, stride)
        else:
            tile, groups = lib.bwd_fused_tile(cg, ch, stride, False, True), 1
        nt = (L + tile - 1) // tile
        outs = []
        for ns in (1, 4, min(37, B * nt), min(256, B * nt)):
            gout = torch.zeros(B, L, ch, device=dev); part = torch.full((B, nt * groups, 2, ch), float('nan'), device=dev); slab = torch.zeros(ns * cg * ch * 3, device=dev)
            if wide:
                lib.bwd_wide(g=g, y=y, st_k=st, bst_k=bst, xin=xi, st_in=sti, add_even=None, w_hi=wh, w_lo=wl, gout=gout, part=part, slab=slab, nslab=ns, B=B, L=L,
                             cg=cg, ch=ch, stride=stride)
            else:
                lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=pro_g, xin=xi, st_in=sti, add_even=None, wb=wb, gout=gout, part=part, slab=slab, nslab=ns, B=B,
                              Lg=Lg, Lh=L, cg=cg, ch=ch, stride=stride, split_precision=True)
            outs.append((gout, part))
        tag = f'blocked partition {"bwd_wide" if wide else "bwd_fused"} {cg}->{ch} L{L} s{stride}'
        RES.append((tag + ': gout and partials independent of the grid', all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])))
        n_in = (xi - sti[:, None, :, 0]) * sti[:, None, :, 1]
        ref = torch.zeros(B, ch, 2, device=dev); lib.stats_finalize(outs[0][1], B, nt * groups, ch, L, 0.0, 1, ref)
        report(tag + ' backward sums vs torch', ref, torch.stack([outs[0][0].mean(1), (outs[0][0] * n_in).mean(1)], dim=-1), tol=2e-4)


def t_plumbing():
    """w2s_token_masks / w2s_cls_scatter / w2s_copy_rows / w2s_zero (round 4: the stock element-wise launches of a step as library kernels)
    against the torch expressions they replace (models/wav2sleep.py:150,319-345)."""
    B, S, R1 = 5, 7, 2
    Ts = [S * 8, S * 16, S * 4]
    xs = [torch.randn(B, T, device=dev) for T in Ts]
    xs[0][1] = float('-inf'); xs[2][3] = float('-inf'); xs[1][4, 0] = float('inf'); xs[1][2, 5] = float('-inf')   # (only the FIRST value of a row decides)
    D = R1 + len(xs)
    keep = torch.full((len(xs), B), float('nan'), device=dev); keypad = torch.full((B * S, D), 77, device=dev, dtype=torch.uint8)
    lib.token_masks(xs, R1, B, S, keep, keypad)
    want_keep = torch.stack([(~torch.isinf(x[:, 0])).float() for x in xs])
    miss = torch.stack([torch.zeros(B, dtype=torch.bool, device=dev)] * R1 + [torch.isinf(x[:, 0]) for x in xs], dim=1)
    want_pad = miss.to(torch.uint8)[:, None, :].expand(B, S, D).reshape(B * S, D)
    RES.append(('token_masks keep', torch.equal(keep, want_keep)))
    RES.append(('token_masks key padding', torch.equal(keypad, want_pad)))
    N, F_ = 37, 128
    src = torch.randn(N, F_, device=dev)
    dst = torch.full((N, D, F_), float('nan'), device=dev)
    lib.cls_scatter(dst, src, N, D, F_)
    want = torch.zeros(N, D, F_, device=dev); want[:, 0] = src
    RES.append(('cls_scatter with source', torch.equal(dst, want)))
    dst2 = torch.randn(N, D, F_, device=dev); keep0 = dst2[:, 0].clone()
    lib.cls_scatter(dst2, None, N, D, F_)
    want2 = torch.zeros(N, D, F_, device=dev); want2[:, 0] = keep0
    RES.append(('cls_scatter without source leaves the CLS rows', torch.equal(dst2, want2)))
    tok = torch.randn(N, D * F_, device=dev); out = torch.zeros(N, F_, device=dev)
    lib.copy_rows(out, F_, tok, D * F_, N, F_)
    RES.append(('copy_rows gathers the CLS rows', torch.equal(out, tok[:, :F_])))
    cm = torch.randint(1, 9, (5, 5), device=dev, dtype=torch.int64); g = torch.randn(1001 * 4, device=dev)[:1000 * 4 + 4]
    lib.zero_(cm); lib.zero_(g)
    RES.append(('zero_', int(cm.abs().sum()) == 0 and float(g.abs().sum()) == 0.0))


STAGES = dict(plumb=t_plumbing, offset=t_offset_channels, wgpf=t_wgrad_pipelined, linpf=t_linear_pipelined, seqconv=t_seq_conv, bwdwide=t_bwd_wide, gradh=t_grad_fp16_chain, wideup2=t_conv_wide_up2, wgwide=t_wgrad_wide, wide=t_conv_wide, causal=t_causal, part=t_stats_partition, fwdfused=t_fwd_fused, first=t_first_layer_recompute, batch=t_batched_entry_points, fusedbf=t_fused_split_precision, fold=t_fused_residual_fold, conv=t_conv_plain, split=t_conv_split_precision, stats=t_conv_stats_pro, dil=t_conv_dilated, dgrad=t_dgrad, wgrad=t_wgrad, rowops=t_rowops, attn=t_attn,
              head=t_head_optim, c1=t_model_c1, c2=t_model_c2, c4=t_model_c4)
if __name__ == '__main__':
    print(lib.version(), torch.cuda.get_device_name(0))
    for s in (sys.argv[1:] or list(STAGES)):
        run(STAGES[s]); torch.cuda.synchronize()
    bad = [n for n, ok in RES if not ok]
    print(f'SUMMARY: {len(RES) - len(bad)}/{len(RES)} ok; failed: {bad}')
