"""Inference forward throughput (eval mode, no saved activations) on the bench workload: recordings/s at batch 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import wav2sleep_amd as W
dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer'), 4).to(dev).eval()
x, y = bench.make_batch(16, 960, 4, dev, 1)
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): out = model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'inference forward: {dt * 1e3:.2f} ms per batch of 16 eight-hour recordings = {16 / dt:.0f} recordings/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
