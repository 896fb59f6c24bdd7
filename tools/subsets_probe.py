"""forward_subsets (encoders once, trunk per subset) against separate forwards of each subset, full length."""
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
x, _ = bench.make_batch(4, 960, 4, dev, 5)
x['PPG'][1] = float('-inf')
subs = [None, ('ECG',), ('ECG', 'THX'), ('PPG',), ('PPG', 'THX')]
with torch.no_grad():
    got = model.forward_subsets(x, subs)
    for s in subs:
        want = model(x if s is None else {k: x[k] for k in s})
        g = got[s]
        print(s, 'identical bits:', torch.equal(g, want), f'max |diff| {float((g - want).abs().max()):.2e}')
