"""ctypes binding of libw2s_hip.so (include/w2s.h).  PyTorch only supplies device memory and the stream.

There is NO fallback: if the shared library is missing or a kernel returns an error, this raises.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('W2S_LIB') or os.path.join(_HERE, 'libw2s_hip.so')  # W2S_LIB: tuning builds only
CSRC = os.path.join(_HERE, 'csrc')

# enums (include/w2s.h)
PRO_NONE, PRO_SANITIZE, PRO_GELU, PRO_IN_GELU, PRO_INBWD, PRO_INBWD_GP, PRO_FIRST = range(7)
PRO_AFFINE = 7   # + activation code (ACT): act(x * scale + shift) on load, statistics operand = (scale, shift) per (b, c)
PRO_AFFINE_BWD = 12   # + activation code: its backward on load (x = g, x2 = y; statistics (scale, shift), backward statistics (c, d))
EPI_PLAIN, EPI_STATS, EPI_AUX_INGELU_ADD, EPI_BIAS, EPI_GP = range(5)
EPI_AFFINE_PART = 5   # + activation code: the reduction pass of the layer below's norm backward in this launch's epilogue (generic path)
MODE_CONTIG, MODE_DILATED, MODE_UP2 = range(3)
ELT_GELU, ELT_GELU_BWD, ELT_ADD, ELT_ADD_DROP, ELT_DROP, ELT_GELU_DROP, ELT_GELU_DROP_BWD = range(7)
FUSE_ADD_DROP, FUSE_Y2_GELU_DROP, FUSE_GELU_BWD_DROP = 2, 4, 8

_fp = C.c_void_p
_i32 = C.c_int32


class ConvArgs(C.Structure):
    _fields_ = [(n, _fp) for n in ('x', 'x2', 'w', 'y', 'y2', 'pro_stats', 'pro_bstats', 'aux', 'aux_stats', 'add_even', 'bias',
                                   'rowkeep', 'part', 'w_hi', 'w_lo')] + \
               [(n, _i32) for n in ('B', 'L_in', 'L_out', 'cin', 'cout', 'taps', 'stride', 'dil', 'pad', 'flip', 'mode',
                                    'ldx', 'ldy', 'ldy2', 'ld_aux', 'pro', 'epi')] + [('reserved', _i32), ('drop_p', C.c_float),
                                                                                            ('drop_seed', C.c_uint64)]


class ReduceJob(C.Structure):
    _fields_ = [('slab', _fp), ('grad', _fp)] + [(n, _i32) for n in ('nslab', 'cout', 'cin', 'taps', 'dil', 'accumulate', 'layout', 'reserved')]


class ColsumJob(C.Structure):
    _fields_ = [('part', _fp), ('out', _fp)] + [(n, _i32) for n in ('nparts', 'C', 'ld', 'accumulate')]


class RepackJob(C.Structure):
    _fields_ = [(n, _fp) for n in ('w', 'fwd', 'bwd', 'fwd_hi', 'fwd_lo', 'bwd_hi', 'bwd_lo')] + [(n, _i32) for n in ('cout', 'cin', 'taps', 'reserved')]


class WgradArgs(C.Structure):
    _fields_ = [(n, _fp) for n in ('g', 'g2', 'g_stats', 'g_bstats', 'x', 'x_stats', 'slab')] + \
               [(n, _i32) for n in ('B', 'L_in', 'L_out', 'cin', 'cout', 'taps', 'stride', 'dil', 'pad', 'ldg', 'ldx',
                                    'pro_g', 'pro_h', 'nslab', 'split_precision')]


EXPORTS = ['w2s_conv_tile', 'w2s_conv_cfg', 'w2s_linear_pf_takes', 'w2s_seq_conv', 'w2s_conv_forward', 'w2s_wgrad', 'w2s_wgrad_max_blocks', 'w2s_wgrad_slabs_per_block_of', 'w2s_wgrad_grid_y', 'w2s_wgrad_slabs_per_block', 'w2s_wgrad_reduce', 'w2s_wgrad_reduce_batch', 'w2s_repack', 'w2s_repack_batch', 'w2s_repack_bf16',
           'w2s_conv_fwd_fused', 'w2s_conv_fwd_fused_tile', 'w2s_bwd_fused', 'w2s_bwd_wide', 'w2s_bwd_wide_tile', 'w2s_bwd_wide_groups', 'w2s_bwd_fused_h', 'w2s_gp_stats_h', 'w2s_enc_first_bwd_h', 'w2s_bwd_fused_w1', 'w2s_bwd_fused_wd', 'w2s_enc_first_wgrad', 'w2s_enc_first_dwd', 'w2s_enc_first_stats', 'w2s_bwd_fused_tile', 'w2s_bwd_fused_folds_residual', 'w2s_stats_finalize', 'w2s_enc_first_fwd', 'w2s_enc_first_join', 'w2s_enc_first_bwd', 'w2s_gp_stats',
           'w2s_layernorm_fwd', 'w2s_layernorm_bwd', 'w2s_bias_grad', 'w2s_colsum', 'w2s_colsum_batch', 'w2s_gelu_bwd_rows', 'w2s_fill_rows', 'w2s_add_rows', 'w2s_causal_normalize_host', 'w2s_eltwise',
           'w2s_attn_fwd', 'w2s_attn_bwd', 'w2s_head_fwd', 'w2s_ce_fwd_bwd', 'w2s_ce_count', 'w2s_ce_wave', 'w2s_ce_final', 'w2s_head_bwd', 'w2s_sumsq_partial',
           'w2s_clip_coef', 'w2s_adamw', 'w2s_ema_update', 'w2s_swap', 'w2s_zscore', 'w2s_augment', 'w2s_map_labels', 'w2s_token_masks', 'w2s_cls_scatter', 'w2s_copy_rows', 'w2s_zero', 'w2s_affine_act', 'w2s_rownorm_fwd', 'w2s_attn_generic_fwd', 'w2s_norm_act_bwd_part', 'w2s_norm_act_bwd_apply', 'w2s_rownorm_bwd_blocks', 'w2s_rownorm_bwd', 'w2s_attn_generic_bwd', 'w2s_norm_fold', 'w2s_norm_bwd_coef', 'w2s_affine_act_join', 'w2s_affine_act_join_bwd', 'w2s_conv1_fwd', 'w2s_conv1_wgrad_parts', 'w2s_conv1_wgrad', 'w2s_version', 'w2s_abi_version']

ABI_VERSION = 8   # include/w2s.h W2S_ABI_VERSION
_lib = None


def build(verbose: bool = False) -> str:
    """Compile libw2s_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j8'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('building libw2s_hip.so failed:\n' + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    # the build must not contain the packed-fp32 form that misbehaves on gfx950 (isa_audit.py).  A gate that cannot see the code is no
    # gate: a missing disassembler, no gfx950 code object, or implausibly few packed instructions FAIL the build instead of passing it
    from .isa_audit import MIN_PACKED, audit, scratch_violations
    seen, bad = audit(LIB_PATH)   # raises if llvm-objdump or the code objects are not found
    if bad:
        raise RuntimeError('libw2s_hip.so contains packed-fp32 instructions whose low lane reads the high half of src1 (wrong results on gfx950 '
                           'beside bf16 MFMA waves, tools/pk_fma_opsel_repro.hip):\n' + '\n'.join(f'  {k}: {i}' for k, i in bad[:20]))
    if seen < MIN_PACKED and not os.environ.get('W2S_LIB'):
        raise RuntimeError(f'ISA audit saw only {seen} packed-fp32 instructions in {LIB_PATH} (expected > {MIN_PACKED}): the disassembly is '
                           f'not covering the library, so the gfx950 op_sel gate would pass vacuously')
    # ... nor register spills nobody asked for (several units are built with accumulators in architectural VGPRs: csrc/Makefile VGPR_FORM)
    sv = scratch_violations(LIB_PATH)
    if sv and not os.environ.get('W2S_LIB'):
        raise RuntimeError('kernels of libw2s_hip.so use scratch memory (register spills) beyond isa_audit.SCRATCH_ALLOWED:\n'
                           + '\n'.join(f'  {k}: {b} B per lane' for k, b in sv[:20]))
    return LIB_PATH


def load():
    """Load the HIP library; raise loudly if it is not there (no CPU path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                               f'(or `make -C wav2sleep_amd/csrc`). wav2sleep_amd has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f'{LIB_PATH} does not export {name}')
        lib.w2s_version.restype = C.c_char_p
        lib.w2s_abi_version.restype = C.c_int
        if lib.w2s_abi_version() != ABI_VERSION:   # a stale W2S_LIB / build_alt library would be called with shifted arguments
            raise RuntimeError(f'{LIB_PATH} has ABI {lib.w2s_abi_version()}, this host was written against {ABI_VERSION} (include/w2s.h): rebuild it')
        _lib = lib
    return _lib


class W2SError(RuntimeError):
    pass


def _chk(rc: int, what: str):
    if rc != 0:
        raise W2SError(f'{what} failed with code {rc} ({ {-1: "EINVAL", -2: "ELAUNCH"}.get(rc, "?")})')


import threading

_tls = threading.local()   # .dev: device index of the most recent tensor argument evaluated on THIS thread (checked in _stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise W2SError('wav2sleep_amd kernels need device tensors (no CPU path)')
    _tls.dev = t.device.index
    return C.c_void_p(t.data_ptr())


def _stream():
    """The current device's current stream.  Every entry point evaluates its tensor arguments first (on the calling thread: the record is
    thread-local, so models driven from different threads on different GPUs do not see each other's), so `_tls.dev` is the device the
    buffers live on: a launch onto another device's stream (cuda:1 tensors while cuda:0 is current) raises instead of faulting.
    Callers bracket their work with `torch.cuda.device(tensor.device)` (Wav2Sleep.forward, FusedTrainStep.step, inputs.*)."""
    s = torch.cuda.current_stream()
    dev = getattr(_tls, 'dev', None)
    if dev is not None and s.device_index != dev:
        raise W2SError(f'tensors live on cuda:{dev} but the current device is cuda:{s.device_index}: run under torch.cuda.device(...)')
    return C.c_void_p(s.cuda_stream)


def _f(t):
    assert t is None or (t.dtype == torch.float32), 'fp32 tensors only'
    return _p(t)


def _fp(t):
    """the `part` argument of a statistics producer: fp32 per-tile partial sums"""
    return _f(t)


def conv_args(*, x, w, y, B, L_in, L_out, cin, cout, taps, stride, pad, dil=1, flip=0, mode=MODE_CONTIG, ldx=None, ldy=None,
              pro=PRO_NONE, epi=EPI_PLAIN, x2=None, pro_stats=None, pro_bstats=None, aux=None, aux_stats=None, add_even=None,
              bias=None, rowkeep=None, part=None, y2=None, ldy2=0, ld_aux=0, w_hi=None, w_lo=None,
              accumulate=False, fuse=0, drop_p=0.0, drop_seed=0) -> ConvArgs:
    a = ConvArgs()
    a.reserved = (1 if accumulate else 0) | fuse   # fuse: FUSE_* bits (EPI_BIAS only)
    a.drop_p, a.drop_seed = drop_p, drop_seed
    a.w_hi, a.w_lo = _p(w_hi), _p(w_lo)
    a.x, a.x2, a.w, a.y, a.y2 = _f(x), _f(x2), _f(w), _f(y), _f(y2)
    a.pro_stats, a.pro_bstats, a.aux, a.aux_stats = _f(pro_stats), _f(pro_bstats), _f(aux), _f(aux_stats)
    a.add_even, a.bias, a.rowkeep, a.part = _f(add_even), _f(bias), _f(rowkeep), _fp(part)
    a.B, a.L_in, a.L_out, a.cin, a.cout, a.taps, a.stride, a.dil, a.pad, a.flip, a.mode = B, L_in, L_out, cin, cout, taps, stride, dil, pad, flip, mode
    a.ldx = cin if ldx is None else ldx
    a.ldy = cout if ldy is None else ldy
    a.ldy2 = ldy2 or cout
    a.ld_aux = ld_aux or cout
    a.pro, a.epi = pro, epi
    return a


def conv_tile(cin, cout, taps, stride, mode=MODE_CONTIG, B=0, L_out=0) -> int:
    """Positions per workgroup tile of the generic kernel (depends on the problem size for short sequences)."""
    a = ConvArgs()
    a.cin, a.cout, a.taps, a.stride, a.mode, a.B, a.L_out = cin, cout, taps, stride, mode, B, L_out
    return load().w2s_conv_tile(C.byref(a))


def conv_tile_of(a: ConvArgs) -> int:
    """Positions per tile of the kernel that conv_forward(a) will launch (sizes the statistics partials [B][ntiles][2][cout]): which
    kernel takes a launch depends on the whole descriptor (the >= 64-channel persistent kernel uses 64-position tiles)."""
    return load().w2s_conv_tile(C.byref(a))


def set_part(a: ConvArgs, part):
    a.part = _fp(part)


class LaunchTimer:
    """Optional per-launch HIP-event timing of the two GEMM-shaped kernels (bench.py roofline leg).  Events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (key, algorithmic bytes, flops, start event, end event)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, nbytes, flops, e0, e1 in self.records:
            d = agg.setdefault(key, dict(launches=0, ms=0.0, bytes=0, flops=0))
            d['launches'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['bytes'] += nbytes
            d['flops'] += flops
        return agg


TIMER: LaunchTimer | None = None
DETAIL = bool(os.environ.get('W2S_TIMER_DETAIL'))


def _timed(key, nbytes, flops, fn):
    if TIMER is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    TIMER.records.append((key, nbytes, flops, e0, e1))


def conv_forward(a: ConvArgs):
    def run():
        _chk(load().w2s_conv_forward(C.byref(a), _stream()), f'w2s_conv_forward(cin={a.cin},cout={a.cout},taps={a.taps},stride={a.stride},mode={a.mode})')
    if TIMER is None:
        return run()
    out_el = a.B * a.L_out * a.cout
    in_el = a.B * a.L_in if a.pro == PRO_FIRST else a.B * a.L_in * a.cin * (2 if a.x2 else 1)
    nbytes = 4 * (in_el + out_el * (2 if a.y2 else 1) + (out_el if a.aux else 0) + (out_el // 2 if a.add_even else 0))
    taps_eff = 1.5 if a.mode == MODE_UP2 else a.taps
    flops = int(2 * out_el * a.cin * taps_eff)
    wide_up2 = a.mode == MODE_UP2 and a.w_hi and a.cin >= 64 and a.cin == a.cout and a.epi == EPI_GP and a.pro == PRO_INBWD_GP and not a.add_even
    if a.taps == 3 and a.w_hi and load().w2s_conv_tile(C.byref(a)) in (32, 64, 128) and a.cout >= 32 and max(a.cin, a.cout) >= 64 \
            and ((a.mode == MODE_CONTIG and (a.epi, bool(a.flip)) in ((EPI_STATS, False), (EPI_GP, True))) or wide_up2):
        key = f'conv_wide_kernel<{a.cin // 16}, {a.cout // 16}, {1 if wide_up2 else a.stride}, {a.pro}, {a.epi}>'
        if DETAIL:
            key += f' L{a.L_out}'
        return _timed(key, nbytes, flops, run)
    if load().w2s_linear_pf_takes(C.byref(a)):   # the transformer's row-wise linears (csrc/linear_pf.hip)
        return _timed(f'linear_pf_kernel<1, {a.taps}>' + (f' {a.cin * a.taps}->{a.cout} L{a.L_out}' if DETAIL else ''), nbytes, flops, run)   # <NB, KC>: the name rocprofv3 reports
    cfg = (C.c_int32 * 4)()
    _chk(load().w2s_conv_cfg(C.byref(a), cfg), 'w2s_conv_cfg')
    nt, mt, wn, mode = cfg[0], cfg[1], cfg[2], cfg[3]
    # key == the kernel name rocprofv3 reports, so bench.py's average can be checked against profiles/
    spec = (-1, -1)
    if (a.pro, a.epi) == (PRO_NONE, EPI_BIAS) and not a.rowkeep and not (a.reserved & 1) and (not a.y2 or (a.reserved & FUSE_Y2_GELU_DROP)) and \
            ((mode, a.taps, a.stride) == (0, 1, 1) or (mode == 1 and a.taps == a.stride)):
        spec = (a.pro, a.epi)
    elif not a.y2 and not a.rowkeep and not a.reserved:
        hot = {(0, 3, 1): [(2, 1), (3, 1), (4, 4), (6, 1)], (0, 3, 2): [(3, 1)], (0, 1, 2): [(2, 2)], (1, 1, 2): [(2, 2)], (2, 3, 2): [(5, 4)],
               (0, 1, 1): [(0, 3), (0, 0)], (1, 4, 4): [(0, 3)], (1, 3, 3): [(0, 3)], (0, 7, 1): [(0, 0)], (1, 7, 1): [(0, 0)]}
        if (a.pro, a.epi) in hot.get((mode, a.taps, a.stride), []):
            spec = (a.pro, a.epi)
    bf = 1 if (a.w_hi and a.w_lo and ((a.cin >= 32 and nt >= 2) or (a.cin == 16 and mode == MODE_CONTIG and a.taps in (1, 3) and a.dil <= 1))) else 0
    key = f'conv_cl_kernel<{nt}, {mt}, {a.taps}, {a.stride}, {mode}, {wn}, {spec[0]}, {spec[1]}, {bf}>'
    if DETAIL:
        key += f' {a.cin}->{a.cout} pro{a.pro} epi{a.epi} L{a.L_out}'
    _timed(key, nbytes, flops, run)


class SeqConvArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w_hi', C.c_void_p), ('w_lo', C.c_void_p), ('y', C.c_void_p), ('out', C.c_void_p), ('rs', C.c_void_p),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('yl', C.c_void_p), ('part', C.c_void_p),
                ('B', C.c_int32), ('S', C.c_int32), ('ldx', C.c_int32), ('dil', C.c_int32), ('pad', C.c_int32), ('flip', C.c_int32), ('mode', C.c_int32),
                ('eps', C.c_float)]


def seq_conv(*, x, w_hi, w_lo, B, S, ldx, dil, pad, mode, flip=0, y=None, out=None, rs=None, gamma=None, beta=None, yl=None, part=None, eps=1e-5):
    """SequenceCNN dilated conv (128 -> 128, k 7) with the channel LayerNorm in its epilogue (include/w2s.h w2s_seq_conv; csrc/seq_conv.hip)."""
    a = SeqConvArgs()
    a.x, a.w_hi, a.w_lo, a.y, a.out, a.rs = _f(x), _p(w_hi), _p(w_lo), _f(y), _f(out), _f(rs)
    a.gamma, a.beta, a.yl, a.part = _f(gamma), _f(beta), _f(yl), _f(part)
    a.B, a.S, a.ldx, a.dil, a.pad, a.flip, a.mode, a.eps = B, S, ldx, dil, pad, flip, mode, eps

    def run():
        _chk(load().w2s_seq_conv(C.byref(a), _stream()), f'w2s_seq_conv(mode={mode},dil={dil})')
    el = B * S * 128
    _timed(f'seq_conv_kernel<{mode}>', 4 * el * (2, 3, 3)[mode], 2 * el * 128 * 7, run)


def _wgrad_args(*, g, x, slab, nslab, B, L_in, L_out, cin, cout, taps, stride, pad, dil=1, ldg=None, ldx=None, pro_g=PRO_NONE,
                pro_h=PRO_NONE, g2=None, g_stats=None, g_bstats=None, x_stats=None, split_precision=False):
    a = WgradArgs()
    a.g, a.g2, a.g_stats, a.g_bstats, a.x, a.x_stats, a.slab = _f(g), _f(g2), _f(g_stats), _f(g_bstats), _f(x), _f(x_stats), _f(slab)
    a.B, a.L_in, a.L_out, a.cin, a.cout, a.taps, a.stride, a.dil, a.pad = B, L_in, L_out, cin, cout, taps, stride, dil, pad
    a.ldg = cout if ldg is None else ldg
    a.ldx = cin if ldx is None else ldx
    a.pro_g, a.pro_h, a.nslab = pro_g, pro_h, nslab
    a.split_precision = int(bool(split_precision))
    return a


def wgrad_max_blocks(**kw) -> int:
    """grid.x the caller should not exceed for this launch (same keyword arguments as `wgrad`; slab / nslab may be None / 0)."""
    return load().w2s_wgrad_max_blocks(C.byref(_wgrad_args(**kw)))


def wgrad_slabs_per_block_of(**kw) -> int:
    """slabs one grid.x block of this launch writes (same keyword arguments as `wgrad`)."""
    return load().w2s_wgrad_slabs_per_block_of(C.byref(_wgrad_args(**kw)))


def wgrad(*, g, x, slab, nslab, B, L_in, L_out, cin, cout, taps, stride, pad, dil=1, ldg=None, ldx=None, pro_g=PRO_NONE,
          pro_h=PRO_NONE, g2=None, g_stats=None, g_bstats=None, x_stats=None, split_precision=False):
    a = _wgrad_args(g=g, x=x, slab=slab, nslab=nslab, B=B, L_in=L_in, L_out=L_out, cin=cin, cout=cout, taps=taps, stride=stride, pad=pad, dil=dil,
                    ldg=ldg, ldx=ldx, pro_g=pro_g, pro_h=pro_h, g2=g2, g_stats=g_stats, g_bstats=g_bstats, x_stats=x_stats, split_precision=split_precision)

    def run():
        _chk(load().w2s_wgrad(C.byref(a), _stream()), f'w2s_wgrad(cin={cin},cout={cout},taps={taps},stride={stride})')
    nbytes = 4 * (B * L_out * cout * (2 if g2 is not None else 1) + B * L_in * cin)
    ntc = cin // 16
    if TIMER is not None and load().w2s_wgrad_max_blocks(C.byref(a)) == 256:   # the role-split kernel takes it (wgrad_wide.hip)
        return _timed(f'wgrad_wide_kernel<{cout // 16}, {cin // 16}, {stride}, {pro_g}, {pro_h}>', nbytes, 2 * B * L_out * cout * cin * taps, run)
    if cin >= 64 and cout >= 64:  # tile-split kernel (wg_cfg in wgrad.hip)
        tapst = 3 if (taps == 3 and dil == 1) else 1
        nw = 8 if cout >= 128 else 4
        ot = (cout // 16) // nw
        while ot > 1 and (ot > 32 // (ntc * tapst) or ((cout // 16) // nw) % ot):
            ot -= 1
        spec = (-1, -1)
        if tapst == 3 and ((stride == 1 and pro_g == PRO_INBWD and pro_h in (PRO_IN_GELU, PRO_GELU)) or
                           (stride == 2 and pro_g == PRO_INBWD_GP and pro_h == PRO_IN_GELU and not (ntc == 4 and nw == 8))):
            spec = (pro_g, pro_h)
        if split_precision and nw >= ntc and nw % ntc == 0:
            key = f'wgrad_bf_kernel<{(nw * ot) // (nw // ntc)}, {tapst}, {ntc}, {nw}, {stride}, {spec[0]}, {spec[1]}>'
        else:
            key = f'wgrad_ts_kernel<{ot}, {tapst}, {ntc}, {nw}, {stride}, {spec[0]}, {spec[1]}>'
    else:
        tapst = 3 if (taps == 3 and dil == 1 and ntc <= 4) else 1
        nto = 8
        while nto > 1 and (nto > 32 // (ntc * tapst) or (cout // 16) % nto):
            nto //= 2
        key = f'wgrad_kernel<{nto}, {ntc}, {tapst}, {stride}>'
    if DETAIL:
        key += f' {cin}->{cout} k{taps} L{L_out}'
    _timed(key, nbytes, 2 * B * L_out * cout * cin * taps, run)


def conv_fwd_fused_tile(cin, cout, stride) -> int:
    return load().w2s_conv_fwd_fused_tile(cin, cout, stride)


def conv_fwd_fused(*, x, w, st_in, w1, y, part, B, L_in, L_out, cin, cout, stride, pro, nwg, pad=1):
    def run():
        _chk(load().w2s_conv_fwd_fused(_f(x), _f(w), _f(st_in), _f(w1), _f(y), _fp(part), B, L_in, L_out, cin, cout, stride, pad, pro, nwg, _stream()),
             f'w2s_conv_fwd_fused(cin={cin},cout={cout},stride={stride},pro={pro})')
    nbytes = 4 * (B * L_in * (1 if pro == PRO_FIRST else cin) + B * L_out * cout)
    mt = (conv_fwd_fused_tile(cin, cout, stride) + 2) // 64   # tiles are 64*MT - 2 (stride 1) or - 1 (stride 2) positions
    key = f'conv_fwd_bf_kernel<{cin // 16}, {cout // 16}, {mt}, {stride}, {pro}>'
    if DETAIL:
        key += f' L{L_out}'
    _timed(key, nbytes, 2 * B * L_out * cout * cin * 3, run)


def bwd_fused_supported(cg, ch) -> bool:
    return (cg, ch) in ((16, 16), (32, 16), (32, 32))


def bwd_fused_tile(cg, ch, stride=1, rd=False, split_precision=True) -> int:
    return load().w2s_bwd_fused_tile(cg, ch, stride, int(bool(rd)), int(bool(split_precision)))


def bwd_fused_folds_residual(cg, ch) -> bool:
    return bool(load().w2s_bwd_fused_folds_residual(cg, ch))


def _h(t):
    assert t is None or t.dtype == torch.float16, 'fp16 tensor expected'
    return _p(t)


def bwd_fused(*, g, y, st_k, bst_k, pro, xin, st_in, add_even, wb, gout, part, slab, nslab, B, Lg, Lh, cg, ch, stride, split_precision=False, pad=1,
              gpre=None, wd=None, slab_d=None, w1=None, y3p=None, st3p=None, gmode=0, hdr_g=None, hdr_p=None, hdr_o=None,
              part_w1=None, x0=None, part_wd=None):
    """part_w1 (conv2 of block 0, w1 given): also leave the first layer's weight-gradient partials; gout may then be None.
    gmode 1 / 2: the fp16 gradient chain (include/w2s.h): gout (and, gmode 2, g / gpre) are fp16 tensors with the headers hdr_*."""
    gin = _h if gmode == 2 else _f
    gou = _h if gmode else _f

    def run():
        if gmode:
            _chk(load().w2s_bwd_fused_h(gin(g), _f(y), _f(st_k), _f(bst_k), pro, _f(xin), _f(st_in), _f(add_even), _f(wb), gou(gout), _f(part),
                                        _f(slab), nslab, B, Lg, Lh, cg, ch, stride, pad, gin(gpre), _f(wd), _f(slab_d), _f(w1), _f(y3p), _f(st3p),
                                        gmode, _f(hdr_g), _f(hdr_p), _f(hdr_o), _stream()), f'w2s_bwd_fused_h(cg={cg},ch={ch},stride={stride},gmode={gmode})')
            return
        if part_wd is not None:
            _chk(load().w2s_bwd_fused_wd(_f(g), _f(y), _f(st_k), _f(bst_k), _f(xin), _f(wb), _f(gout), _fp(part), _f(slab), nslab, B, Lh, pad, _f(gpre), _f(wd),
                                         _f(slab_d), _f(y3p), _f(st3p), _f(x0), _f(part_wd), _stream()), 'w2s_bwd_fused_wd')
            return
        if part_w1 is not None:
            _chk(load().w2s_bwd_fused_w1(_f(g), _f(y), _f(st_k), _f(bst_k), _f(xin), _f(st_in), _f(wb), _f(gout), _fp(part), _f(part_w1), _f(slab), nslab, B, Lh,
                                         pad, _f(w1), _stream()), 'w2s_bwd_fused_w1')
            return
        _chk(load().w2s_bwd_fused(_f(g), _f(y), _f(st_k), _f(bst_k), pro, _f(xin), _f(st_in), _f(add_even), _f(wb), _f(gout), _fp(part),
                                  _f(slab), nslab, B, Lg, Lh, cg, ch, stride, pad, int(bool(split_precision)), _f(gpre), _f(wd), _f(slab_d), _f(w1), _f(y3p), _f(st3p), _stream()),
             f'w2s_bwd_fused(cg={cg},ch={ch},stride={stride})')
    wg, wo = (2 if gmode == 2 else 4), (2 if gmode else 4)   # bytes per stored gradient element in / out
    nbytes = (B * Lg * cg * (wg + 4) + (B * Lh * (4 * 1 + (wo * ch if gout is not None else 0)) if w1 is not None else B * Lh * ch * (4 + wo)) + (4 * B * Lh * ch // 2 if add_even is not None else 0)
              + (wg * B * Lh * cg // 2 if gpre is not None else 0) + (4 * B * Lh * ch if y3p is not None else 0))
    flops = 2 * B * Lg * cg * ch * 3 * 2
    if split_precision or gmode:
        key = (f'bwd_fused_bf_kernel<{cg // 16}, {ch // 16}, {(bwd_fused_tile(cg, ch, stride, gpre is not None) + 2) // 64}, {1 if stride == 2 else 0}, '
               f'{1 if gpre is not None else 0}, {1 if w1 is not None else 0}, {gmode}>')
    else:
        key = f'bwd_fused_kernel<{cg // 16}, {ch // 16}, {bwd_fused_tile(cg, ch, stride, False, False) // 64}, {1 if stride == 2 else 0}, 1>'
    if DETAIL:
        key += f' L{Lh}'
    _timed(key, nbytes, flops, run)


def bwd_wide_tile(cg, ch, stride=1) -> int:
    return load().w2s_bwd_wide_tile(cg, ch, stride)


def bwd_wide_groups(cg, ch, stride=1) -> int:
    return load().w2s_bwd_wide_groups(cg, ch, stride)


def bwd_wide_takes(B, L, cg, ch, stride=1, hst=True, rd=False) -> bool:
    """Would w2s_bwd_wide take this launch (instance exists, the statistics tables of B samples fit its LDS)?  L: input-side length;
    rd: with the residual branch folded in (gpre given)."""
    one = C.c_void_p(1)
    return load().w2s_bwd_wide(None, None, None, None, None, one if hst else None, None, None, None, None, None, None, 0, B, L, cg, ch, stride,
                               None, None, one if rd else None, None, None, None, 1, 1, None) == 0


def bwd_wide(*, g, y, st_k, bst_k, xin, st_in, add_even, w_hi, w_lo, gout, part, slab, nslab, B, L, cg, ch, stride=1, y3p=None, st3p=None,
             gpre=None, wd_hi=None, wd_lo=None, slab_d=None, pad=1):
    def run():
        _chk(load().w2s_bwd_wide(_f(g), _f(y), _f(st_k), _f(bst_k), _f(xin), _f(st_in), _f(add_even), _p(w_hi), _p(w_lo), _f(gout), _fp(part), _f(slab), nslab,
                                 B, L, cg, ch, stride, _f(y3p), _f(st3p), _f(gpre), _p(wd_hi), _p(wd_lo), _f(slab_d), pad, 0, _stream()),
             f'w2s_bwd_wide(cg={cg},ch={ch},stride={stride})')
    nbytes = 4 * (2 * B * (L // stride) * cg + 2 * B * L * ch + (B * L * ch // 2 if add_even is not None else 0) + (B * L * ch if y3p is not None else 0)
                  + (B * L * cg // 2 if gpre is not None else 0))
    key = f'bwd_wide_kernel<{cg // 16}, {ch // 16}, {1 if st_in is not None else 0}, {stride}{", rd" if gpre is not None else ""}>'
    if DETAIL:
        key += f' L{L}'
    _timed(key, nbytes, 2 * B * (L // stride) * cg * ch * 3 * 2, run)


def wgrad_slabs_per_block(cin, cout, taps, dil=1) -> int:
    return load().w2s_wgrad_slabs_per_block(cin, cout, taps, dil)


def wgrad_grid_y(cin, cout, taps, dil=1) -> int:
    return load().w2s_wgrad_grid_y(cin, cout, taps, dil)


def frag_major_planes(a_RK: torch.Tensor):
    """[rows, K] fp32 GEMM operand -> (hi, lo) bf16 planes in the fragment-major order the split-precision conv reads
    (include/w2s.h, w2s_repack_bf16); rows % 16 == 0, K % 32 == 0.  Tooling / tests; the engine uses w2s_repack_batch."""
    R, K = a_RK.shape
    if K % 32 and K % 16 == 0:   # 16 input channels: K padded with zeros to a multiple of 32 (two taps per K step)
        a_RK = torch.cat([a_RK, a_RK.new_zeros(R, 32 - K % 32)], dim=1)
        K = a_RK.shape[1]
    if K % 32 or R % 16:
        return None, None
    hi = a_RK.bfloat16()
    lo = (a_RK - hi.float()).bfloat16()
    f = lambda t: t.view(R // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    return f(hi), f(lo)


def wgrad_reduce_batch(jobs):
    """jobs: list of (slab, nslab, grad, cout, cin, taps, dil, accumulate, layout); one launch per 48 jobs."""
    if not jobs:
        return
    arr = (ReduceJob * len(jobs))()
    for q, (slab, nslab, grad, cout, cin, taps, dil, accumulate, layout) in zip(arr, jobs):
        q.slab, q.grad = _f(slab), _f(grad)
        q.nslab, q.cout, q.cin, q.taps, q.dil, q.accumulate, q.layout = nslab, cout, cin, taps, dil, int(accumulate), layout
    _chk(load().w2s_wgrad_reduce_batch(arr, len(jobs), _stream()), 'w2s_wgrad_reduce_batch')


def repack_batch(jobs):
    """jobs: list of (w, fwd, bwd, fwd_hi, fwd_lo, bwd_hi, bwd_lo, cout, cin, taps); tensors or None."""
    if not jobs:
        return
    arr = (RepackJob * len(jobs))()
    for q, (w, fwd, bwd, fh, fl, bh, bl, cout, cin, taps) in zip(arr, jobs):
        q.w, q.fwd, q.bwd = _f(w), _f(fwd), _f(bwd)
        q.fwd_hi, q.fwd_lo, q.bwd_hi, q.bwd_lo = _p(fh), _p(fl), _p(bh), _p(bl)
        q.cout, q.cin, q.taps = cout, cin, taps
    _chk(load().w2s_repack_batch(arr, len(jobs), _stream()), 'w2s_repack_batch')


def wgrad_reduce(slab, nslab, grad, cout, cin, taps, dil=1, accumulate=False, layout=0):
    _chk(load().w2s_wgrad_reduce(_f(slab), nslab, _f(grad), cout, cin, taps, dil, int(accumulate), layout, _stream()), 'w2s_wgrad_reduce')


def repack(w, fwd, bwd, cout, cin, taps):
    _chk(load().w2s_repack(_f(w), _f(fwd), _f(bwd), cout, cin, taps, _stream()), 'w2s_repack')


def repack_bf16(w, fwd_hi, fwd_lo, bwd_hi, bwd_lo, cout, cin, taps):
    _chk(load().w2s_repack_bf16(_f(w), _p(fwd_hi), _p(fwd_lo), _p(bwd_hi), _p(bwd_lo), cout, cin, taps, _stream()), 'w2s_repack_bf16')


def stats_finalize(part, B, ntiles, Cc, count, eps, kind, out):
    _chk(load().w2s_stats_finalize(_f(part), B, ntiles, Cc, C.c_long(count), C.c_float(eps), kind, _f(out), _stream()), 'w2s_stats_finalize')


def enc_first_fwd(x, w, y, part, B, L, cout, tile, causal=False):
    _chk(load().w2s_enc_first_fwd(_f(x), _f(w), _f(y), _fp(part), B, L, cout, tile, int(causal), _stream()),
         'w2s_enc_first_fwd')


def enc_first_join(x, wd, y3, stats3, pre, B, L, cout):
    _chk(load().w2s_enc_first_join(_f(x), _f(wd), _f(y3), _f(stats3), _f(pre), B, L, cout, _stream()), 'w2s_enc_first_join')


def enc_first_bwd(x, gn1, y1, stats1, bstats1, gpre, slab, nslab, B, L, cout, w1=None, causal=False, hdr_n=None, hdr_p=None):
    if gn1.dtype == torch.float16:   # the fp16 gradient chain
        _chk(load().w2s_enc_first_bwd_h(_f(x), _h(gn1), _f(hdr_n), _f(y1), _f(stats1), _f(bstats1), _h(gpre), _f(hdr_p), _f(slab), nslab, B, L, cout, _f(w1),
                                        int(causal), _stream()), 'w2s_enc_first_bwd_h')
        return
    _chk(load().w2s_enc_first_bwd(_f(x), _f(gn1), _f(y1), _f(stats1), _f(bstats1), _f(gpre), _f(slab), nslab, B, L, cout, _f(w1), int(causal), _stream()),
         'w2s_enc_first_bwd')


def enc_first_wgrad(xmom, ntx, w1, part_w1, stats1, bstats1, out, B, ntiles):
    _chk(load().w2s_enc_first_wgrad(_f(xmom), ntx, _f(w1), _f(part_w1), _f(stats1), _f(bstats1), _f(out), B, ntiles, _stream()), 'w2s_enc_first_wgrad')


def enc_first_stats(x, w, part, xmom, B, L, tile, causal=False):
    _chk(load().w2s_enc_first_stats(_f(x), _f(w), _fp(part), _f(xmom), B, L, tile, int(causal), _stream()),
         'w2s_enc_first_stats')


def enc_first_dwd(x, gpre, slab, nslab, B, L):
    _chk(load().w2s_enc_first_dwd(_f(x), _f(gpre), _f(slab), nslab, B, L, _stream()), 'w2s_enc_first_dwd')


def gp_stats(g, y, stats, part, B, L, Cc, tile, hdr_g=None, hdr_amax=None):
    """hdr_g: g is an fp16 chain tensor with that header; hdr_amax (fp32 g): also publish {1, max |g|} there (the chain's entry)."""
    if hdr_g is not None or hdr_amax is not None:
        half = g.dtype == torch.float16
        _chk(load().w2s_gp_stats_h(_h(g) if half else _f(g), int(half), _f(hdr_g), _f(hdr_amax), _f(y), _f(stats), _f(part), B, L, Cc, tile, _stream()),
             'w2s_gp_stats_h')
        return
    _chk(load().w2s_gp_stats(_f(g), _f(y), _f(stats), _fp(part), B, L, Cc, tile, _stream()), 'w2s_gp_stats')


def layernorm_fwd(x, ldx, gamma, beta, y, ldy, rstat, rows, Cc, eps, gelu=False):
    _chk(load().w2s_layernorm_fwd(_f(x), ldx, _f(gamma), _f(beta), _f(y), ldy, _f(rstat), rows, Cc, C.c_float(eps), int(gelu), _stream()),
         'w2s_layernorm_fwd')


def layernorm_bwd(g, ldg, x, ldx, gamma, beta, rstat, gadd, gx, ldgx, part_gamma, part_beta, rows, Cc, gelu, nparts):
    _chk(load().w2s_layernorm_bwd(_f(g), ldg, _f(x), ldx, _f(gamma), _f(beta), _f(rstat), _f(gadd), _f(gx), ldgx, _f(part_gamma),
                                  _f(part_beta), rows, Cc, int(gelu), nparts, _stream()), 'w2s_layernorm_bwd')


def bias_grad(g, rows, Cc, ldg, part, nparts):
    _chk(load().w2s_bias_grad(_f(g), rows, Cc, ldg, _f(part), nparts, _stream()), 'w2s_bias_grad')


def colsum_batch(jobs):
    """jobs: list of (part, nparts, C, out, accumulate, ld); one launch per 64 jobs."""
    if not jobs:
        return
    arr = (ColsumJob * len(jobs))()
    for q, (part, nparts, Cc, out, accumulate, ld) in zip(arr, jobs):
        q.part, q.out = _f(part), _f(out)
        q.nparts, q.C, q.ld, q.accumulate = nparts, Cc, Cc if ld is None else ld, int(accumulate)
    _chk(load().w2s_colsum_batch(arr, len(jobs), _stream()), 'w2s_colsum_batch')


def colsum(part, nparts, Cc, out, accumulate=False, ld=None):
    _chk(load().w2s_colsum(_f(part), nparts, Cc, Cc if ld is None else ld, _f(out), int(accumulate), _stream()), 'w2s_colsum')


def gelu_bwd_rows(g, ldg, pre, keep, rows_per_sample, out, rows, Cc):
    _chk(load().w2s_gelu_bwd_rows(_f(g), ldg, _f(pre), _f(keep), rows_per_sample, _f(out), rows, Cc, _stream()), 'w2s_gelu_bwd_rows')


def fill_rows(dst, ld, src, rows, Cc):
    _chk(load().w2s_fill_rows(_f(dst), ld, _f(src), rows, Cc, _stream()), 'w2s_fill_rows')


def add_rows(dst, ld, src, src_stride, keep, rows_per_sample, rows, Cc, accumulate):
    _chk(load().w2s_add_rows(_f(dst), ld, _f(src), src_stride, _f(keep), rows_per_sample, rows, Cc, int(accumulate), _stream()), 'w2s_add_rows')


def token_masks(xs, R1, B, S, keep, keypad):
    """keep [nsig][B] float and keypad [B*S][R1 + nsig] uint8 from the first value of every signal row (`-inf` row = missing modality)"""
    n = len(xs)
    for t in xs:
        assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    ptrs = (C.c_void_p * n)(*[_p(t).value for t in xs])
    lds = (C.c_long * n)(*[t.stride(0) for t in xs])
    assert keypad.dtype == torch.uint8
    _chk(load().w2s_token_masks(ptrs, lds, n, R1, B, S, _f(keep), _p(keypad), _stream()), 'w2s_token_masks')


def cls_scatter(dst, src, N, D, F):
    _chk(load().w2s_cls_scatter(_f(dst), _f(src), C.c_long(N), D, F, _stream()), 'w2s_cls_scatter')


def copy_rows(dst, ld_dst, src, ld_src, rows, Cc):
    _chk(load().w2s_copy_rows(_f(dst), C.c_long(ld_dst), _f(src), C.c_long(ld_src), C.c_long(rows), Cc, _stream()), 'w2s_copy_rows')


def zero_(t):
    """t.zero_() as a launch of this library (t contiguous, a multiple of 4 bytes)"""
    assert t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0, 'w2s_zero: contiguous storage, a multiple of 4 bytes'
    _chk(load().w2s_zero(_p(t), C.c_long(t.numel() * t.element_size()), _stream()), 'w2s_zero')


def eltwise(op, a, b, y, n, p=0.0, seed=0):
    _chk(load().w2s_eltwise(op, _f(a), _f(b), _f(y), C.c_long(n), C.c_float(p), C.c_uint64(seed), _stream()), 'w2s_eltwise')


def attn_fwd(qkv, keypad, out, N, D, H, p=0.0, seed=0, nq=None):
    _chk(load().w2s_attn_fwd(_f(qkv), _p(keypad), _f(out), N, D, H, D if nq is None else nq, C.c_float(p), C.c_uint64(seed), _stream()), 'w2s_attn_fwd')


def attn_bwd(qkv, keypad, gout, gqkv, N, D, H, p=0.0, seed=0, nq=None):
    _chk(load().w2s_attn_bwd(_f(qkv), _p(keypad), _f(gout), _f(gqkv), N, D, H, D if nq is None else nq, C.c_float(p), C.c_uint64(seed), _stream()), 'w2s_attn_bwd')


def head_fwd(pre, ld, w, bias, logits, rows, F, nc, gelu_in):
    _chk(load().w2s_head_fwd(_f(pre), ld, _f(w), _f(bias), _f(logits), rows, F, nc, int(gelu_in), _stream()), 'w2s_head_fwd')


def ce_fwd_bwd(logits, labels, rows, nc, part, loss_out, glogits, cmat, gscale=1.0):
    _chk(load().w2s_ce_fwd_bwd(_f(logits), _f(labels), rows, nc, _f(part), _f(loss_out), _f(glogits), _p(cmat), C.c_float(gscale), _stream()),
         'w2s_ce_fwd_bwd')


def ce_count(labels, rows, nc, count):
    _chk(load().w2s_ce_count(_f(labels), rows, nc, _f(count), _stream()), 'w2s_ce_count')


def ce_wave(logits, labels, rows, nc, part, count, glogits, cmat, gscale=1.0):
    _chk(load().w2s_ce_wave(_f(logits), _f(labels), rows, nc, _f(part), _f(count), _f(glogits), _p(cmat), C.c_float(gscale), _stream()), 'w2s_ce_wave')


def ce_final(part, nblocks, loss_out):
    _chk(load().w2s_ce_final(_f(part), nblocks, _f(loss_out), _stream()), 'w2s_ce_final')


def head_bwd(pre, ld, w, glogits, gpre, ldg, part, nparts, rows, F, nc, gelu_in):
    _chk(load().w2s_head_bwd(_f(pre), ld, _f(w), _f(glogits), _f(gpre), ldg, _f(part), nparts, rows, F, nc, int(gelu_in), _stream()),
         'w2s_head_bwd')


def sumsq_partial(g, n, part, nparts):
    _chk(load().w2s_sumsq_partial(_f(g), C.c_long(n), _f(part), nparts, _stream()), 'w2s_sumsq_partial')


def clip_coef(part, nparts, hyper, normcoef):
    _chk(load().w2s_clip_coef(_f(part), nparts, _f(hyper), _f(normcoef), _stream()), 'w2s_clip_coef')


def adamw(p, g, m, v, n, hyper, normcoef):
    _chk(load().w2s_adamw(_f(p), _f(g), _f(m), _f(v), C.c_long(n), _f(hyper), _f(normcoef), _stream()), 'w2s_adamw')


def ema_update(ema, p, n, decay):
    _chk(load().w2s_ema_update(_f(ema), _f(p), C.c_long(n), C.c_float(decay), _stream()), 'w2s_ema_update')


def swap(a, b, n):
    _chk(load().w2s_swap(_f(a), _f(b), C.c_long(n), _stream()), 'w2s_swap')


def zscore(x, y, rows, T, part, nblk, eps, stats_out=None):
    assert part.dtype == torch.float64
    _chk(load().w2s_zscore(_f(x), _f(y), rows, C.c_long(T), _p(part), nblk, C.c_float(eps), _f(stats_out), _stream()), 'w2s_zscore')


def augment(x, B, T, sign, keep):
    _chk(load().w2s_augment(_f(x), B, C.c_long(T), _f(sign), _p(keep), _stream()), 'w2s_augment')


def map_labels(src, dst, n, num_classes):
    _chk(load().w2s_map_labels(_f(src), _f(dst), C.c_long(n), num_classes, _stream()), 'w2s_map_labels')


ACT = {'linear': 0, 'relu': 1, 'leaky': 2, 'gelu': 3, 'silu': 4, 'swish': 4}


def affine_act(x, ldx, scale, shift, sample_stride, y, ldy, rows_per_sample, rows, Cc, act, slope=0.01):
    _chk(load().w2s_affine_act(_f(x), ldx, _f(scale), _f(shift), sample_stride, _f(y), ldy, rows_per_sample, C.c_long(rows), Cc, act, C.c_float(slope),
                               _stream()), 'w2s_affine_act')


def affine_act_join(x, ldx, scale, shift, sample_stride, add, ld_add, y, ldy, rows_per_sample, rows, Cc, act, act2, slope=0.01):
    _chk(load().w2s_affine_act_join(_f(x), ldx, _f(scale), _f(shift), sample_stride, _f(add), ld_add, _f(y), ldy, rows_per_sample, C.c_long(rows), Cc,
                                    act, act2, C.c_float(slope), _stream()), 'w2s_affine_act_join')


def affine_act_join_bwd(g, ldg, x, ldx, scale, shift, sample_stride, add, ld_add, gs, ldgs, rows_per_sample, rows, Cc, act, act2, slope=0.01):
    _chk(load().w2s_affine_act_join_bwd(_f(g), ldg, _f(x), ldx, _f(scale), _f(shift), sample_stride, _f(add), ld_add, _f(gs), ldgs, rows_per_sample,
                                        C.c_long(rows), Cc, act, act2, C.c_float(slope), _stream()), 'w2s_affine_act_join_bwd')


C1_TILE = 1024   # positions per statistics / weight-gradient partial of the one-channel convolutions (csrc/generic.hip W2S_C1_TILE)


def conv1_fwd(x, w, bias, y, part, B, L_in, L_out, Cc, K, stride, pad):
    _chk(load().w2s_conv1_fwd(_f(x), _f(w), _f(bias), _f(y), _f(part), B, L_in, L_out, Cc, K, stride, pad, _stream()), 'w2s_conv1_fwd')


def conv1_wgrad_parts(B, L_out) -> int:
    return load().w2s_conv1_wgrad_parts(B, L_out)


def conv1_wgrad(g, y2, ss, cd, x, part, B, L_in, L_out, Cc, K, stride, pad, act=0):
    _chk(load().w2s_conv1_wgrad(_f(g), _f(y2), _f(ss), _f(cd), _f(x), _f(part), B, L_in, L_out, Cc, K, stride, pad, act, _stream()), 'w2s_conv1_wgrad')


def rownorm_fwd(x, ldx, gamma, beta, y, ldy, rows, Cc, eps, rms=False, act=0, slope=0.01):
    _chk(load().w2s_rownorm_fwd(_f(x), ldx, _f(gamma), _f(beta), _f(y), ldy, C.c_long(rows), Cc, C.c_float(eps), int(rms), act, C.c_float(slope),
                                _stream()), 'w2s_rownorm_fwd')


def attn_generic_fwd(qkv, keypad, out, N, D, H, hd, p=0.0, seed=0):
    _chk(load().w2s_attn_generic_fwd(_f(qkv), _p(keypad), _f(out), C.c_long(N), D, H, hd, C.c_float(p), C.c_uint64(seed), _stream()),
         'w2s_attn_generic_fwd')


def attn_generic_bwd(qkv, keypad, gout, gqkv, N, D, H, hd, p=0.0, seed=0):
    _chk(load().w2s_attn_generic_bwd(_f(qkv), _p(keypad), _f(gout), _f(gqkv), C.c_long(N), D, H, hd, C.c_float(p), C.c_uint64(seed), _stream()),
         'w2s_attn_generic_bwd')


def norm_act_bwd_part(g, ldg, y, ldy, stats, stats_stride, gamma, beta, rows_per_sample, nsamples, Cc, act, slope, tile, part):
    _chk(load().w2s_norm_act_bwd_part(_f(g), ldg, _f(y), ldy, _f(stats), stats_stride, _f(gamma), _f(beta), rows_per_sample, nsamples, Cc, act,
                                      C.c_float(slope), tile, _f(part), _stream()), 'w2s_norm_act_bwd_part')


def norm_act_bwd_apply(g, ldg, y, ldy, stats, stats_stride, gamma, beta, coef, coef_stride, gy, ldgy, rows_per_sample, rows, Cc, act, slope=0.01):
    _chk(load().w2s_norm_act_bwd_apply(_f(g), ldg, _f(y), ldy, _f(stats), stats_stride, _f(gamma), _f(beta), _f(coef), coef_stride, _f(gy), ldgy,
                                       rows_per_sample, C.c_long(rows), Cc, act, C.c_float(slope), _stream()), 'w2s_norm_act_bwd_apply')


NORM_KIND = {'instance': 0, 'batch_train': 1, 'batch_eval': 2, 'group': 3}


def norm_fold(kind, stats, B, Cc, G, gamma, beta, run_mean, run_var, eps, momentum, count, scale, shift, mr, ss=None):
    _chk(load().w2s_norm_fold(kind, _f(stats), B, Cc, G, _f(gamma), _f(beta), _f(run_mean), _f(run_var), C.c_float(eps), C.c_float(momentum),
                              C.c_double(count), _f(scale), _f(shift), _f(mr), _f(ss), _stream()), 'w2s_norm_fold')


def norm_bwd_coef(kind, means, mr, B, Cc, G, gamma, beta, L, coef, dgamma, dbeta, cd=None, y_sums=False):
    _chk(load().w2s_norm_bwd_coef(kind, _f(means), _f(mr), B, Cc, G, _f(gamma), _f(beta), C.c_double(L), _f(coef), _f(dgamma), _f(dbeta), _f(cd),
                                  int(y_sums), _stream()), 'w2s_norm_bwd_coef')


def rownorm_bwd_blocks(rows) -> int:
    return load().w2s_rownorm_bwd_blocks(C.c_long(rows))


def rownorm_bwd(g, ldg, x, ldx, gamma, beta, gx, ldgx, part, rows, Cc, eps, rms=False, act=0, slope=0.01):
    _chk(load().w2s_rownorm_bwd(_f(g), ldg, _f(x), ldx, _f(gamma), _f(beta), _f(gx), ldgx, _f(part), C.c_long(rows), Cc, C.c_float(eps), int(rms),
                                act, C.c_float(slope), _stream()), 'w2s_rownorm_bwd')


def version() -> str:
    return load().w2s_version().decode()
