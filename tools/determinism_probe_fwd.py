"""Are the full-size forward logits bit-reproducible (B = 16, four encoder streams)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').eval()
x, y = bench.make_batch(16, 960, 4, torch.device('cuda'), 1234)
x['PPG'][3] = float('-inf')
outs = []
with torch.no_grad():
    for _ in range(8):
        outs.append(model(x).clone()); torch.cuda.synchronize()
print('forward logits identical over 8 runs:', all(torch.equal(outs[0], o) for o in outs[1:]), ' max |logit|', float(outs[0].abs().max()))
tr = W.FusedTrainStep(model.train())
losses = []
for rep in range(2):
    torch.manual_seed(42)
    m2 = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                     W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                     W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').train()
    t2 = W.FusedTrainStep(m2)
    for _ in range(6):
        out = t2.step(x, y)
    torch.cuda.synchronize()
    losses.append((float(out['loss']), m2._flat.clone()))
print('two 6-step training runs from the same seed: final loss', losses[0][0], losses[1][0], ' parameters identical:', torch.equal(losses[0][1], losses[1][1]))
