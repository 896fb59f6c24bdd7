"""The benchmark's own shape: 4 modalities x 960 epochs, B = 16, default initialisation.  Every gradient tensor of ONE backward pass
against the oracle (accumulated over 8 micro-batches of 2 on the host: sum_mb (valid_mb / valid_total) * grad(mean loss of mb))."""
import os, sys, time
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
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
B, S = 16, 960
x, y = O.make_inputs(cfg, B, S, seed=77, missing={'ABD': [3], 'PPG': [3, 7], 'ECG': [11], 'THX': [0, 15]})
logits = model({k: v.to('cuda') for k, v in x.items()})
loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.to('cuda').reshape(-1).long(), ignore_index=-1)
loss.backward(); torch.cuda.synchronize()
total = int((y >= 0).sum())
want = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in sd.items()}
wl = 0.0
t0 = time.time()
for b0 in range(0, B, 2):
    xm = {k: v[b0:b0 + 2] for k, v in x.items()}; ym = y[b0:b0 + 2]
    l, _, g = O.loss_and_grads(sd, cfg, xm, ym)
    w = int((ym >= 0).sum()) / total
    wl += w * l
    for k in want: want[k] += w * g[k].double()
print(f'oracle: {time.time() - t0:.0f} s; loss {float(loss):.6f} vs {wl:.6f}')
worst = ('', 0.0); over = []
for name, p in model.named_parameters():
    rel = float((p.grad.detach().cpu().double() - want[name]).norm() / (want[name].norm() + 1e-30))
    if rel > worst[1]: worst = (name, rel)
    if rel > 1e-3: over.append((name, rel))
print(f'B=16 full size: worst gradient tensor {worst[0]} rel-L2 {worst[1]:.2e}; tensors over 1e-3: {over}')
