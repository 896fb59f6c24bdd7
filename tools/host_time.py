"""How long does the HOST need to enqueue one train step (no synchronisation), against the device time of the step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W
dev = torch.device('cuda', 0)
torch.manual_seed(42)
SM = bench.SIGNAL_MAP
model = W.Wav2Sleep(W.SignalEncoders(SM, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to(dev).train()
tr = W.FusedTrainStep(model)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y = bench.make_batch(B, 960, 4, dev, 1234)
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
print(f'batch {B}: host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step; until the device is done {1e3 * (t2 - t0) / K:.2f} ms/step; device still busy after the last enqueue for {1e3 * (t2 - t1):.2f} ms')
