"""Full-size gradient of the first-layer conv weight: build (fp32 kernels) vs the oracle in fp32 vs the oracle in fp64 (truth)."""
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
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
B, S = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 960
x, y = O.make_inputs(cfg, B, S, seed=123, missing={'THX': [1]})
logits = model({k: v.to('cuda') for k, v in x.items()})
loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.to('cuda').reshape(-1).long(), ignore_index=-1)
loss.backward()
_, _, g32 = O.loss_and_grads(sd, cfg, x, y)
sd64 = {k: v.double() for k, v in sd.items()}
x64 = {k: v.double() for k, v in x.items()}
_, _, g64 = O.loss_and_grads(sd64, cfg, x64, y)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
worst_b = worst_o = 0
for name, p in model.named_parameters():
    eb, eo = rel(p.grad.cpu(), g64[name]), rel(g32[name], g64[name])
    worst_b, worst_o = max(worst_b, eb if 'cnn.0.conv1' not in name else 0), max(worst_o, eo if 'cnn.0.conv1' not in name else 0)
    if 'cnn.0.conv1' in name or 'cnn.0.downsample' in name:
        print(f'{name:60s} build vs fp64 {eb:.2e}   oracle-fp32 vs fp64 {eo:.2e}   build vs oracle-fp32 {rel(p.grad.cpu(), g32[name]):.2e}   |g| {float(g64[name].norm()):.3e}')
print(f'all other tensors: worst build vs fp64 {worst_b:.2e}, worst oracle-fp32 vs fp64 {worst_o:.2e}')
