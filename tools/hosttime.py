import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import wav2sleep_amd as W
dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer'), 4).to(dev).train()
tr = W.FusedTrainStep(model)
x, y = bench.make_batch(16, 960, 4, dev, 1)
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(x, y); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('host ms / total ms per step:', [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in ts])
print('peak mem GB', torch.cuda.max_memory_allocated() / 2**30, 'reserved', torch.cuda.memory_reserved() / 2**30)
