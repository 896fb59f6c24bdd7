"""Single-kernel micro-benchmarks (HIP-event timed) for tuning.  python tools/kbench.py [name ...] [--iters N]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2sleep_amd import lib

dev = 'cuda'

def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def conv_case(cin, cout, L, stride=1, pro=lib.PRO_IN_GELU, epi=lib.EPI_STATS, B=16, taps=3, mode=lib.MODE_CONTIG, flip=0, dil=1, pad=None):
    Lo = L // stride if mode != lib.MODE_UP2 else L * 2
    x = torch.randn(B, L, cin, device=dev); x2 = torch.randn(B, L, cin, device=dev)
    w = torch.randn(cout, taps, cin, device=dev) / (cin * taps) ** 0.5
    y = torch.empty(B, Lo, cout, device=dev)
    st = torch.rand(B, cin, 2, device=dev) + 0.5; bst = torch.rand(B, cin, 2, device=dev) * 0.01
    ost = torch.rand(B, cout, 2, device=dev) + 0.5
    aux = torch.randn(B, Lo, cout, device=dev) if epi in (lib.EPI_GP, lib.EPI_AUX_INGELU_ADD) else None
    wh = wl = None
    if os.environ.get('BF') == '1':
        wh, wl = lib.frag_major_planes(w.view(cout, taps * cin))
    a = lib.conv_args(w_hi=wh, w_lo=wl, x=x, x2=x2 if pro >= lib.PRO_INBWD else None, w=w, y=y, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, taps=taps, stride=stride,
                      pad=(1 if taps == 3 else 0) if pad is None else pad, dil=dil, bias=torch.randn(cout, device=dev) if epi == lib.EPI_BIAS else None, flip=flip, mode=mode, pro=pro, epi=epi, pro_stats=st, pro_bstats=bst, aux=aux, aux_stats=ost if aux is not None else None,
                      part=None)
    tile = lib.conv_tile_of(a)
    part = torch.empty(B, (Lo + tile - 1) // tile, 2, cout, device=dev)
    if epi in (lib.EPI_STATS, lib.EPI_GP):
        lib.set_part(a, part)
    keep = (x, x2, w, y, st, bst, ost, aux, part, wh, wl)   # the descriptor holds raw pointers
    global LAST_PART
    LAST_PART = part
    nbytes = 4 * (B * L * cin * (2 if pro >= lib.PRO_INBWD else 1) + B * Lo * cout * (2 if aux is not None else 1))
    flops = 2 * B * Lo * cout * cin * (1.5 if mode == lib.MODE_UP2 else taps)
    return (lambda keep=keep: lib.conv_forward(a)), nbytes, flops

def wgrad_case(cin, cout, L, stride=1, taps=3, B=16, pro_g=lib.PRO_INBWD, pro_h=lib.PRO_IN_GELU):
    Lo = L // stride
    g = torch.randn(B, Lo, cout, device=dev); g2 = torch.randn(B, Lo, cout, device=dev); x = torch.randn(B, L, cin, device=dev)
    st = torch.rand(B, cout, 2, device=dev) + 0.5; bst = torch.rand(B, cout, 2, device=dev) * 0.01; xst = torch.rand(B, cin, 2, device=dev) + 0.5
    gy = lib.wgrad_grid_y(cin, cout, taps, 1)
    gx = max(1, min((B * Lo + 255) // 256, max(1, int(os.environ.get('WGCAP', 512)) // gy))); nslab = gx * lib.wgrad_slabs_per_block(cin, cout, taps, 1)
    slab = torch.empty(nslab * cout * cin * taps, device=dev)
    fn = lambda: lib.wgrad(g=g, g2=g2, g_stats=st, g_bstats=bst, x=x, x_stats=xst, slab=slab, nslab=nslab, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout,
                           taps=taps, stride=stride, pad=1 if taps == 3 else 0, pro_g=pro_g, pro_h=pro_h, split_precision=os.environ.get('BF') == '1')
    return fn, 4 * (B * Lo * cout * 2 + B * L * cin), 2 * B * Lo * cout * cin * taps

def wgrad_trunk_case(cin, cout, rows, taps=1, stride=1, B=1, dil=1, pad=0, pro_h=lib.PRO_NONE, cap=None):
    """weight gradients of the trunk (no gradient-side transform): the transformer's linears over `rows` token rows (taps = 1, or the 512-wide
    input of linear2 as 4 strided taps), the SequenceCNN's dilated convs (B x rows, 7 taps)"""
    L_in = rows * stride if taps == stride and taps > 1 else rows
    g = torch.randn(B, rows, cout, device=dev); x = torch.randn(B, L_in, cin, device=dev)
    gy = lib.wgrad_grid_y(cin, cout, taps, dil)
    cap = int(os.environ.get('WGCAP', 128 if cap is None else cap))
    gx = max(1, min((B * rows + 255) // 256, max(1, 512 // gy), cap)); nslab = gx * lib.wgrad_slabs_per_block(cin, cout, taps, dil)
    slab = torch.empty(nslab * cout * cin * taps, device=dev)
    fn = lambda: lib.wgrad(g=g, x=x, slab=slab, nslab=nslab, B=B, L_in=L_in, L_out=rows, cin=cin, cout=cout, taps=taps, stride=stride, pad=pad, dil=dil,
                           pro_h=pro_h, split_precision=True)
    return fn, 4 * (B * rows * cout + B * L_in * cin), 2 * B * rows * cout * cin * taps


def seqconv_case(dil, mode, B=16, S=960):
    """SequenceCNN conv with the LayerNorm in its epilogue (w2s_seq_conv): mode 1 = forward (conv + LN + GELU), 2 = backward (data gradient + LN backward)"""
    C = 128
    x = torch.randn(B, S, C, device=dev); w = torch.randn(C, 7 * C, device=dev) / (7 * C) ** 0.5
    wh, wl = lib.frag_major_planes(w)
    y = torch.empty(B, S, C, device=dev); out = torch.empty(B, S, C, device=dev); rs = torch.rand(B * S, 2, device=dev) + 0.5
    gm = torch.rand(C, device=dev) + 0.5; bt = torch.randn(C, device=dev) * 0.1; yl = torch.randn(B, S, C, device=dev)
    part = torch.empty(B * ((S + 63) // 64), 2, C, device=dev)
    fn = lambda: lib.seq_conv(x=x, w_hi=wh, w_lo=wl, B=B, S=S, ldx=C, dil=dil, pad=3 * dil, mode=mode, flip=int(mode == 2), y=y, out=out, rs=rs, gamma=gm, beta=bt,
                              yl=yl, part=part)
    return fn, 4 * B * S * C * 3, 2 * B * S * C * C * 7


def elt_case(op, n=16 * 983040 * 16):
    a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); y = torch.empty(n, device=dev)
    nb = 4 * n * (3 if op == lib.ELT_ADD else 2)
    return (lambda: lib.eltwise(op, a, b if op == lib.ELT_ADD else None, y, n)), nb, n

def fused_case(cg, ch, L, stride=1, B=16, nslab=None):
    Lg = L // stride
    g = torch.randn(B, Lg, cg, device=dev); y = torch.randn(B, Lg, cg, device=dev); xin = torch.randn(B, L, ch, device=dev)
    st = torch.rand(B, cg, 2, device=dev) + 0.5; bst = torch.rand(B, cg, 2, device=dev) * 0.01; sti = torch.rand(B, ch, 2, device=dev) + 0.5
    wb = torch.randn(ch, 3, cg, device=dev) / 7; gout = torch.empty(B, L, ch, device=dev)
    tile = lib.bwd_fused_tile(cg, ch, stride, False, os.environ.get('BF') == '1'); nt = (L + tile - 1) // tile
    part = torch.empty(B, nt, 2, ch, device=dev)
    ns = nslab or int(os.environ.get('NSLAB', 1024))
    slab = torch.empty(ns * cg * ch * 3, device=dev)
    global LAST_PART
    LAST_PART = part
    fn = lambda: lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD if stride == 1 else lib.PRO_INBWD_GP, xin=xin, st_in=sti, add_even=None,
                               wb=wb, gout=gout, part=None if os.environ.get('NOPART') else part, slab=slab, nslab=ns, B=B, Lg=Lg, Lh=L, cg=cg, ch=ch, stride=stride,
                               split_precision=os.environ.get('BF') == '1')   # NOPART=1: without the statistics partials (their per-tile reduction + barrier)
    return fn, 4 * (2 * B * Lg * cg + 2 * B * L * ch), 2 * B * Lg * cg * ch * 3 * 2

def ffwd_case(cin, cout, L, stride=1, B=16, pro=lib.PRO_IN_GELU):
    Lo = L // stride
    x = torch.randn(B, L, cin, device=dev); w = torch.randn(cout, 3, cin, device=dev) / (3 * cin) ** 0.5
    st = torch.rand(B, cin, 2, device=dev) + 0.5
    t = lib.conv_fwd_fused_tile(cin, cout, stride)
    y = torch.empty(B, Lo, cout, device=dev); part = torch.empty(B, (Lo + t - 1) // t, 2, cout, device=dev)
    nwg = int(os.environ.get('NWG', 1024 if cin == 16 else 512))
    fn = lambda: lib.conv_fwd_fused(x=x, w=w, st_in=st, w1=None, y=y, part=part, B=B, L_in=L, L_out=Lo, cin=cin, cout=cout, stride=stride, pro=pro, nwg=nwg)
    return fn, 4 * (B * L * cin + B * Lo * cout), 2 * B * Lo * cout * cin * 3

def bfirst_case(L=983040, B=16):
    """block 0's conv2 backward in the first-layer recompute form (bwd_fused_bf_kernel<1,1,4,0,0,1,0>): input side rebuilt from the raw
    signal, conv1's weight-gradient partials in the epilogue, gn1 not stored -- the step's third-largest kernel"""
    c = 16
    g = torch.randn(B, L, c, device=dev); y = torch.randn(B, L, c, device=dev); x = torch.randn(B, L, device=dev)
    st = torch.rand(B, c, 2, device=dev) + 0.5; bst = torch.rand(B, c, 2, device=dev) * 0.01; sti = torch.rand(B, c, 2, device=dev) + 0.5
    wb = torch.randn(c, 3, c, device=dev) / 7; w1 = torch.randn(c, 1, 3, device=dev) / 2
    tile = lib.bwd_fused_tile(c, c, 1); nt = (L + tile - 1) // tile
    part = torch.empty(B, nt, 2, c, device=dev); part_w1 = torch.empty(B, nt, 48, device=dev)
    ns = int(os.environ.get('NSLAB', 512)); slab = torch.empty(ns * c * c * 3, device=dev)
    fn = lambda: lib.bwd_fused(g=g, y=y, st_k=st, bst_k=bst, pro=lib.PRO_INBWD, xin=x, st_in=sti, add_even=None, wb=wb, gout=None, part=part, slab=slab, nslab=ns,
                               B=B, Lg=L, Lh=L, cg=c, ch=c, stride=1, split_precision=True, w1=w1, part_w1=part_w1)
    return fn, 4 * (2 * B * L * c + B * L), 2 * B * L * c * c * 3 * 2

def ffirst_case(L=983040, B=16):
    """block 0's conv2 forward with the first-layer recompute prologue (conv_fwd_bf_kernel<1,1,4,1,6>)"""
    c = 16
    x = torch.randn(B, L, device=dev); w = torch.randn(c, 3, c, device=dev) / 7; w1 = torch.randn(c, 1, 3, device=dev) / 2
    st = torch.rand(B, c, 2, device=dev) + 0.5
    t = lib.conv_fwd_fused_tile(c, c, 1)
    y = torch.empty(B, L, c, device=dev); part = torch.empty(B, (L + t - 1) // t, 2, c, device=dev)
    nwg = int(os.environ.get('NWG', 512))
    fn = lambda: lib.conv_fwd_fused(x=x, w=w, st_in=st, w1=w1, y=y, part=part, B=B, L_in=L, L_out=L, cin=c, cout=c, stride=1, pro=lib.PRO_FIRST, nwg=nwg)
    return fn, 4 * (B * L + B * L * c), 2 * B * L * c * c * 3

def bwd_wide_case(L, stride=1, hst=True, rd=False, ch=64, B=16):
    """one-pass backward of a 64-channel k=3 conv (w2s_bwd_wide) at the benchmark's shapes: stride 1 (conv2: hst, conv1: not, optionally with the
    residual fold) or the block's stride-2 conv3"""
    cg = 64
    Lg = L // stride
    g = torch.randn(B, Lg, cg, device=dev); y = torch.randn(B, Lg, cg, device=dev); xin = torch.randn(B, L, ch, device=dev)
    st = torch.rand(B, cg, 2, device=dev) + 0.5; bst = torch.rand(B, cg, 2, device=dev) * 0.01; sti = torch.rand(B, ch, 2, device=dev) + 0.5
    w = torch.randn(ch, 3 * cg, device=dev) / (3 * cg) ** 0.5
    wh, wl = lib.frag_major_planes(w)
    gout = torch.empty(B, L, ch, device=dev)
    tile = lib.bwd_wide_tile(cg, ch, stride); nt = (L + tile - 1) // tile; grp = lib.bwd_wide_groups(cg, ch, stride)
    part = torch.empty(B, nt, grp, 2, ch, device=dev)
    ns = min(256, B * nt)
    slab = torch.empty(ns, cg * 3 * ch, device=dev)
    kw = {}
    if rd:
        wd = torch.randn(ch, cg, device=dev) / cg ** 0.5
        dh, dl = lib.frag_major_planes(wd)
        kw = dict(gpre=torch.randn(B, L // 2, cg, device=dev), wd_hi=dh, wd_lo=dl, slab_d=torch.empty(ns, cg * ch, device=dev))
    global LAST_PART
    LAST_PART = part
    fn = lambda: lib.bwd_wide(g=g, y=y, st_k=st, bst_k=bst, xin=xin, st_in=sti if hst else None, add_even=None, w_hi=wh, w_lo=wl, gout=gout, part=part, slab=slab,
                              nslab=ns, B=B, L=L, cg=cg, ch=ch, stride=stride, **kw)
    nbytes = 4 * (2 * B * Lg * cg + 2 * B * L * ch + (B * L * cg // 2 if rd else 0))
    return fn, nbytes, 2 * B * Lg * cg * ch * 3 * 2


LAST_PART = None
CASES = {
    'bw64': lambda: bwd_wide_case(61440),
    'bw64c1': lambda: bwd_wide_case(61440, hst=False),
    'bw64rd': lambda: bwd_wide_case(61440, hst=False, rd=True),
    'bw64u': lambda: bwd_wide_case(61440, stride=2),
    'bfirst': bfirst_case,
    'ffirst': ffirst_case,
    'ff16': lambda: ffwd_case(16, 16, 983040),
    'ff16s2': lambda: ffwd_case(16, 16, 983040, stride=2),
    'ff1632': lambda: ffwd_case(16, 32, 245760),
    'ff32': lambda: ffwd_case(32, 32, 245760),
    'ff32s2': lambda: ffwd_case(32, 32, 245760, stride=2),
    'b16': lambda: fused_case(16, 16, 983040),
    'b16u': lambda: fused_case(16, 16, 983040, stride=2),
    'b32': lambda: fused_case(32, 32, 245760),
    'b32u': lambda: fused_case(32, 32, 245760, stride=2),
    'b21': lambda: fused_case(32, 16, 245760),
    'add': lambda: elt_case(lib.ELT_ADD),
    'gelu': lambda: elt_case(lib.ELT_GELU),
    'tcopy': lambda: (lambda a, y: ((lambda: y.copy_(a)), 8 * a.numel(), a.numel()))(torch.randn(16 * 983040 * 16, device=dev), torch.empty(16 * 983040 * 16, device=dev)),
    'f16': lambda: conv_case(16, 16, 983040),
    'f16n': lambda: conv_case(16, 16, 983040, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN),
    'f16s2': lambda: conv_case(16, 16, 983040, stride=2),
    'd16': lambda: conv_case(16, 16, 983040, pro=lib.PRO_INBWD, epi=lib.EPI_GP, flip=1),
    'u16': lambda: conv_case(16, 16, 491520, stride=2, pro=lib.PRO_INBWD_GP, epi=lib.EPI_GP, mode=lib.MODE_UP2),
    'u64': lambda: conv_case(64, 64, 30720, stride=2, pro=lib.PRO_INBWD_GP, epi=lib.EPI_GP, mode=lib.MODE_UP2),
    'u128': lambda: conv_case(128, 128, 7680, stride=2, pro=lib.PRO_INBWD_GP, epi=lib.EPI_GP, mode=lib.MODE_UP2),
    'w16': lambda: wgrad_case(16, 16, 983040),
    'f32': lambda: conv_case(32, 32, 245760),
    'f32s2': lambda: conv_case(32, 32, 245760, stride=2),
    'f1632': lambda: conv_case(16, 32, 245760, pro=lib.PRO_GELU),
    'w32': lambda: wgrad_case(32, 32, 245760),
    'f64': lambda: conv_case(64, 64, 61440),
    'd64': lambda: conv_case(64, 64, 61440, pro=lib.PRO_INBWD, epi=lib.EPI_GP, flip=1),
    'w64': lambda: wgrad_case(64, 64, 61440),
    'f64s2': lambda: conv_case(64, 64, 61440, stride=2),
    'f3264': lambda: conv_case(32, 64, 61440, pro=lib.PRO_GELU),
    'f64128': lambda: conv_case(64, 128, 15360, pro=lib.PRO_GELU),
    'f128s2': lambda: conv_case(128, 128, 15360, stride=2),
    'f128': lambda: conv_case(128, 128, 15360),
    'f128n': lambda: conv_case(128, 128, 15360, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN),
    'f64n': lambda: conv_case(64, 64, 61440, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN),
    'f128g': lambda: conv_case(128, 128, 15360, pro=lib.PRO_GELU, epi=lib.EPI_PLAIN),
    'f128s': lambda: conv_case(128, 128, 15360, pro=lib.PRO_NONE, epi=lib.EPI_STATS),
    'd128': lambda: conv_case(128, 128, 15360, pro=lib.PRO_INBWD, epi=lib.EPI_GP, flip=1),
    'w128': lambda: wgrad_case(128, 128, 15360),
    'w64s2': lambda: wgrad_case(64, 64, 61440, stride=2, pro_g=lib.PRO_INBWD_GP),
    'w128s2': lambda: wgrad_case(128, 128, 15360, stride=2, pro_g=lib.PRO_INBWD_GP),
    **{f'sq{d}{"f" if m == 1 else "b"}': (lambda d=d, m=m: seqconv_case(d, m)) for d in (1, 2, 4, 8, 16, 32) for m in (1, 2)},
    'wqkv': lambda: wgrad_trunk_case(128, 384, 76800),
    'wff1': lambda: wgrad_trunk_case(128, 512, 76800),
    'wproj': lambda: wgrad_trunk_case(128, 128, 76800),
    'wff2': lambda: wgrad_trunk_case(128, 128, 76800, taps=4, stride=4),
    'wff1c': lambda: wgrad_trunk_case(128, 512, 15360),
    'wseq1': lambda: wgrad_trunk_case(128, 128, 960, taps=7, B=16, dil=1, pad=3, cap=512),
    'wseq32': lambda: wgrad_trunk_case(128, 128, 960, taps=7, B=16, dil=32, pad=96, cap=512),
    'wdense': lambda: wgrad_trunk_case(128, 128, 960, taps=4, stride=4, B=16, pro_h=lib.PRO_GELU, cap=512),
    'qkv': lambda: conv_case(128, 384, 76800, B=1, taps=1, pro=lib.PRO_NONE, epi=lib.EPI_BIAS),
    'ff1': lambda: conv_case(128, 512, 76800, B=1, taps=1, pro=lib.PRO_NONE, epi=lib.EPI_BIAS),
    'proj': lambda: conv_case(128, 128, 76800, B=1, taps=1, pro=lib.PRO_NONE, epi=lib.EPI_BIAS),
    'ff2': lambda: conv_case(128, 128, 76800 * 4, stride=4, B=1, taps=4, pro=lib.PRO_NONE, epi=lib.EPI_BIAS, mode=lib.MODE_DILATED),
    'seq1': lambda: conv_case(128, 128, 960, B=16, taps=7, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN, mode=lib.MODE_DILATED, dil=1, pad=3),
    'seq32': lambda: conv_case(128, 128, 960, B=16, taps=7, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN, mode=lib.MODE_DILATED, dil=32, pad=96),
    **{f'seq{d}{m}': (lambda d=d, m=m: conv_case(128, 128, 960, B=16, taps=7, pro=lib.PRO_NONE, epi=lib.EPI_PLAIN,
                                                 mode=lib.MODE_CONTIG if m == 'c' else lib.MODE_DILATED, dil=d, pad=3 * d))
       for d in (1, 2, 4, 8, 16, 32) for m in 'cd'},
}
if __name__ == '__main__':
    ap = argparse.ArgumentParser(); ap.add_argument('names', nargs='*'); ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    for n in (a.names or list(CASES)):
        fn, nb, fl = CASES[n]()
        ms = timeit(fn, a.iters)
        print(f'{n:8s} {ms*1e3:9.1f} us  {nb/ms/1e6:8.0f} GB/s  {fl/ms/1e9:7.1f} TF/s', flush=True)
        if os.environ.get('W2S_STAMP') and LAST_PART is not None and n.startswith('b') and not n.startswith('bw') and n != 'bfirst':
            v = LAST_PART.view(-1)[:16].tolist()   # bwd_fused_bf (-DW2S_WIDE_STAMP): phases of workgroup 0's first wave
            print('   per wave (commit, prefetch issue + second barrier): ' + ', '.join(f'w{w}: {v[8 + 2 * w]:.0f} / {v[9 + 2 * w]:.0f}' for w in range(4)), flush=True)
            print(f'   stamps (cycles, {v[6]:.0f} tiles): barrier A {v[0]:.0f}, wait for the prefetched loads {v[1]:.0f}, commit {v[2]:.0f}, prefetch issue + barrier B {v[3]:.0f}, '
                  f'data gradient + epilogue {v[4]:.0f}, weight gradient {v[5]:.0f}; in-kernel clock {v[7]:.0f} MHz', flush=True)
        elif os.environ.get('W2S_STAMP') and LAST_PART is not None:   # diagnostic builds (-DW2S_WIDE_STAMP): cycle stamps of workgroup 0 in part[0..7]
            v = LAST_PART.view(-1)[:8].tolist()
            print(f'   stamps (cycles of workgroup 0): consumer K loop {v[0]:.0f}, epilogue {v[1]:.0f}, barrier {v[2]:.0f}, tiles (bwd_wide: weight-gradient loop) {v[3]:.0f}; '
                  f'producer stage {v[4]:.0f}, barrier {v[5]:.0f}, rounds {v[6]:.0f}', flush=True)
