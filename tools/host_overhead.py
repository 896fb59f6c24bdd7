"""How far ahead of the GPU does the host run?  Times the enqueue of K train steps (no sync) against their completion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W

dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', norm='instance', chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4).to(dev).train()
tr = W.FusedTrainStep(model)
x, y = bench.make_batch(16, 960, 4, dev, 1234)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3 * (t1 - t0) / K:.2f} ms/step   complete {1e3 * (t2 - t0) / K:.2f} ms/step   host share {(t1 - t0) / (t2 - t0):.2f}')
