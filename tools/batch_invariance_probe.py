"""A recording's logits must not depend on its batch neighbours (instance / layer norms only): B = 32 against two halves of 16, and
B = 5 against single recordings; full length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').eval()
dev = torch.device('cuda')
with torch.no_grad():
    x, _ = bench.make_batch(32, 960, 4, dev, 99)
    x['ECG'][5] = float('-inf'); x['ABD'][20] = float('-inf')
    full = model(x)
    halves = torch.cat([model({k: v[:16] for k, v in x.items()}), model({k: v[16:] for k, v in x.items()})])
    print(f'B=32 vs 2 x B=16: max |diff| {float((full - halves).abs().max()):.3e} (max |logit| {float(full.abs().max()):.3f}); identical bits: {torch.equal(full, halves)}')
    x5 = {k: v[:5] for k, v in x.items()}
    f5 = model(x5)
    singles = torch.cat([model({k: v[i:i + 1] for k, v in x5.items()}) for i in range(5)])
    print(f'B=5 vs 5 x B=1: max |diff| {float((f5 - singles).abs().max()):.3e}; identical bits: {torch.equal(f5, singles)}; arg-max equal: {bool((f5.argmax(-1) == singles.argmax(-1)).all())}')
