"""Is the full-size gradient bit-reproducible?  Two backward passes of the same step; report tensors that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wav2sleep_amd as W
from oracle import wav2sleep_oracle as O
SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').train()
cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
B, S = int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[1]) if len(sys.argv) > 1 else 960
x, y = O.make_inputs(cfg, B, S, seed=123, missing={'THX': [1]})
x = {k: v.to('cuda') for k, v in x.items()}; y = y.to('cuda')
runs = []
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for r in range(NR):
    model.zero_grad(set_to_none=True)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.reshape(-1).long(), ignore_index=-1)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
bad = 0
for n in runs[0]:
    d = max(float((runs[0][n] - runs[k][n]).abs().max()) for k in range(1, NR))
    if d > 0:
        bad += 1
        print(f'{n:60s} max |diff| over NR runs {d:.3e}  (|g|max {float(runs[0][n].abs().max()):.3e})')
print(f'S={S} B={B}: {bad} of {len(runs[0])} gradient tensors are not bit-reproducible')
