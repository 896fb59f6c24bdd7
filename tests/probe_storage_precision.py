"""CPU experiment (oracle arithmetic, no GPU): what does storing the encoder's inter-layer tensors (y1, y2, y3 pre-norm conv outputs and
the pre-activation block output) in bf16 / fp16 do to the logits?  Statistics from the unrounded fp32 values (as the kernels compute them
from their accumulators), consumers read the rounded tensor.  `python tests/probe_storage_precision.py [S] [B]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import wav2sleep_oracle as O

MODE = {'dt': None}

def rnd(t):
    return t if MODE['dt'] is None else t.to(MODE['dt']).float()

def conv_layer_in(x, w, stride, eps, causal=False):
    y = F.conv1d(x, w, None, stride=stride, padding=1)
    mu = y.mean(-1, keepdim=True); var = y.var(-1, unbiased=False, keepdim=True)
    return O.gelu((rnd(y) - mu) / torch.sqrt(var + eps))

def conv_block(sd, p, x, eps, taps=None, causal=False):
    h1 = conv_layer_in(x, sd[p + 'conv1.conv.weight'], 1, eps)
    h2 = conv_layer_in(h1, sd[p + 'conv2.conv.weight'], 1, eps)
    h3 = conv_layer_in(h2, sd[p + 'conv3.conv.weight'], 2, eps)
    return O.gelu(rnd(h3 + F.conv1d(x, sd[p + 'downsample.weight'], None, stride=2)))

O.conv_layer_in = conv_layer_in; O.conv_block = conv_block
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
from tests.test_r2_pins_cpu import default_init_model
sds = {'trained-like (make_state_dict)': O.make_state_dict(cfg, seed=12), 'default init seed 42': {k: v.detach().clone() for k, v in default_init_model().state_dict().items()}}
x, _ = O.make_inputs(cfg, B, S, seed=5)
for name, sd in sds.items():
    MODE['dt'] = None
    ref = O.forward(sd, cfg, x)
    srt = ref.sort(-1).values
    print(f'{name}: |logit| max {float(ref.abs().max()):.3f}, min top-2 margin {float((srt[..., -1] - srt[..., -2]).min()):.2e}')
    for dt in (torch.bfloat16, torch.float16):
        MODE['dt'] = dt
        got = O.forward(sd, cfg, x)
        err = (got - ref).abs()
        print(f'   {str(dt):16s} max abs err {float(err.max()):.3e} = {float(err.max() / ref.abs().max()):.2e} of max |logit|; rms {float(err.pow(2).mean().sqrt()):.2e}; '
              f'arg-max agreement {float((got.argmax(-1) == ref.argmax(-1)).float().mean()):.4f}')
