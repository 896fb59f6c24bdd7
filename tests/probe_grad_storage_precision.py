"""CPU experiment (oracle arithmetic, no GPU): what does storing the encoder's inter-layer GRADIENT tensors in reduced precision do to
the parameter gradients?  The three gradient tensors the backward kernels write per ConvBlock1D -- d/d(xhat1), d/d(xhat2) (`gn1`, `gn2`:
gradient w.r.t. the instance-norm outputs of conv1 / conv2) and d/d(pre) (`gpre`: w.r.t. the block's pre-activation output) -- are
rounded at exactly those points by autograd hooks; everything else stays fp32 (fp32 accumulate, fp32 statistics).  Logits and arg-max
never see a gradient tensor, so the forward parity bar cannot move.

    python tests/probe_grad_storage_precision.py [S] [B] [eog]

Modes: bf16 (round to nearest even); fp16 with ONE power-of-two scale per tensor chosen from the tensor's own max (the best a per-tensor
scale can do); fp16 with the scale derived from the PREVIOUS tensor of the chain, placed as the kernels place it (what a producer kernel
can know when it writes; w2s_gscale_for in csrc/w2s_common.h); `chain32`: that rule on exactly the tensors the shipped kernels store as
fp16 -- gn1 / gn2 of the <= 32-channel blocks 0..3 and the gpre tensors between them (W2S_GRAD_FP16, engine._encoder_backward).
Reported: worst / median relative L2 error over all parameter-gradient tensors, and the dynamic range the chain tensors need."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import wav2sleep_oracle as O

MODE = {'kind': None, 'prev_amax': None, 'ranges': []}


def _round(g, tag):
    k = MODE['kind']
    amax = float(g.abs().max())
    if amax == 0.0 or k is None:
        return g
    nz = g[g != 0].abs()
    MODE['ranges'].append((tag, math.log2(amax / float(nz.min())), math.log2(amax / float(nz.median()))))
    if k == 'bf16':
        return g.bfloat16().float()
    if k == 'fp16_own':
        s = 2.0 ** (14 - math.ceil(math.log2(amax)))   # max lands in [2^13, 2^14]: 4x headroom below fp16's 65504
    else:   # 'fp16_prev' / 'chain32': scale from the previous chain tensor's max (first one: its own), the kernels' placement
        ref = MODE['prev_amax'] or amax
        MODE['prev_amax'] = amax
        s = 2.0 ** (9 - math.frexp(ref)[1])
    return (g * s).half().float() / s   # no clamp: an overflow shows as inf, as in the kernels


def conv_layer_in(x, w, stride, eps, causal=False, hook=True):
    y = O.causal_conv1d(x, w, stride) if causal else F.conv1d(x, w, None, stride=stride, padding=1)
    xhat = O.instance_norm(y, eps)
    if MODE['kind'] == 'chain32' and w.size(0) > 32:
        hook = False
    if hook and xhat.requires_grad:
        xhat.register_hook(lambda g: _round(g, 'gn'))
    return O.gelu(xhat)


def conv_block(sd, p, x, eps, taps=None, causal=False):
    first = x.size(1) == 1
    h1 = conv_layer_in(x, sd[p + 'conv1.conv.weight'], 1, eps, causal)
    h2 = conv_layer_in(h1, sd[p + 'conv2.conv.weight'], 1, eps, causal)
    h3 = conv_layer_in(h2, sd[p + 'conv3.conv.weight'], 2, eps, causal, hook=False)   # conv3's GELU' is applied on load from gpre
    pre = h3 + F.conv1d(x, sd[p + 'downsample.weight'], None, stride=2)
    blk = int(p.split('.')[-2])
    if pre.requires_grad and not (MODE['kind'] == 'chain32' and blk > 2):   # chain32: gpre_0 .. gpre_2 (gpre_3 comes from the 64-channel kernels, fp32)
        pre.register_hook(lambda g: _round(g, 'gpre'))
    return O.gelu(pre)


O.conv_layer_in, O.conv_block = conv_layer_in, conv_block

if __name__ == '__main__':
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    eog = len(sys.argv) > 3 and sys.argv[3] == 'eog'
    sm = {'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'} if eog else {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
    nc = 5 if eog else 4
    cfg = O.ModelConfig(signal_map=sm, num_classes=nc)
    from tests.test_r2_pins_cpu import default_init_model
    sds = {'trained-like (make_state_dict)': O.make_state_dict(cfg, seed=12)}
    if not eog:
        sds['default init seed 42'] = {k: v.detach().clone() for k, v in default_init_model().state_dict().items()}
    x, y = O.make_inputs(cfg, B, S, seed=5)
    for name, sd in sds.items():
        MODE['kind'] = None
        _, _, ref = O.loss_and_grads(sd, cfg, x, y)
        print(f'{name}  ({"EOG pair" if eog else "4 modalities"}, B={B}, S={S})')
        for kind in ('bf16', 'fp16_own', 'fp16_prev', 'chain32'):
            MODE.update(kind=kind, prev_amax=None, ranges=[])
            _, _, got = O.loss_and_grads(sd, cfg, x, y)
            rels = sorted(((float((got[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)), k) for k in ref if float(ref[k].norm()) > 0), reverse=True)
            enc = [r for r in rels if r[1].startswith('signal_encoders')]
            rng = MODE['ranges']
            print(f'   {kind:10s} worst rel-L2 {rels[0][0]:.2e} ({rels[0][1]}); median over encoder tensors {enc[len(enc) // 2][0]:.2e}; '
                  f'tensors over 1e-3: {sum(r[0] > 1e-3 for r in rels)}, over 2e-3: {sum(r[0] > 2e-3 for r in rels)} of {len(rels)}; '
                  f'chain tensors span up to 2^{max(r[1] for r in rng):.0f} (max/min non-zero), 2^{max(r[2] for r in rng):.0f} (max/median)')
