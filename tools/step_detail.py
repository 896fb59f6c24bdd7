"""Per-launch table of one single-stream train step (batch 16, 8 h, 4 modalities): time, algorithmic bytes, TB/s, TFLOP/s
and the excess over a 4.5 TB/s streaming floor.  Run with W2S_TIMER_DETAIL=1 on the GPU box."""
import os, sys
os.environ.setdefault('W2S_TIMER_DETAIL', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import wav2sleep_amd as W
from wav2sleep_amd import lib
dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer'), 4).to(dev).train()
tr = W.FusedTrainStep(model)
tr.eng.multi_stream = False
x, y = bench.make_batch(16, 960, 4, dev, 1)
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize()
lib.TIMER = lib.LaunchTimer()
tr.step(x, y)
torch.cuda.synchronize()
agg = {}
for key, nbytes, flops, e0, e1 in lib.TIMER.records:
    a = agg.setdefault(key, [0, 0.0, 0, 0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += nbytes; a[3] += flops
lib.TIMER = None
tot = sum(a[1] for a in agg.values())
print(f'{"kernel":90s} {"n":>3s} {"us":>8s} {"GB":>7s} {"TB/s":>6s} {"TF/s":>6s} {"excess":>7s}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    floor = a[2] / 4.5e12 * 1e6
    print(f'{k[:90]:90s} {a[0]:3d} {a[1]:8.0f} {a[2] / 1e9:7.3f} {a[2] / a[1] / 1e6:6.2f} {a[3] / a[1] / 1e6:6.1f} {a[1] - floor:7.0f}')
print('total us', tot, 'bytes GB', sum(a[2] for a in agg.values()) / 1e9)
