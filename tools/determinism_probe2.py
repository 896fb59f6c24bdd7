"""Which input of the first-layer weight-gradient kernel differs between two runs of the same step (multi-stream, full size)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wav2sleep_amd as W
from wav2sleep_amd import lib
from oracle import wav2sleep_oracle as O
SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').train()
cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
x, y = O.make_inputs(cfg, 2, 960, seed=123, missing={'THX': [1]})
x = {k: v.to('cuda') for k, v in x.items()}; y = y.to('cuda')
log = []
real = lib.enc_first_bwd
def spy(xs, gn1, y1, st1, bs1, gpre, slab, nslab, B, L, c, w1=None, causal=False):
    real(xs, gn1, y1, st1, bs1, gpre, slab, nslab, B, L, c, w1=w1, causal=causal)
    cs = lambda t: t.double().sum() + 3 * t.double().abs().sum()     # (enqueued on the current stream: no sync here)
    log.append((L, dict(gn1=cs(gn1), st1=cs(st1), bs1=cs(bs1), gpre=cs(gpre), slab48=cs(slab.view(nslab, 64)[:, :48]), slab16=cs(slab.view(nslab, 64)[:, 48:]))))
lib.enc_first_bwd = spy
runs = []
for r in range(6):
    log.clear()
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(x).reshape(-1, 4), y.reshape(-1).long(), ignore_index=-1)
    loss.backward(); torch.cuda.synchronize()
    runs.append([(L, {k: float(v) for k, v in d.items()}) for L, d in log])
for i in range(len(runs[0])):
    L = runs[0][i][0]
    for k in runs[0][i][1]:
        vals = {runs[r][i][1][k] for r in range(len(runs))}
        if len(vals) > 1:
            print(f'call {i} (L={L}): {k} differs across runs: {sorted(vals)[:3]} ...')
print('done')
