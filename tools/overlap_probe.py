"""Feasibility probe for micro-batch pipelining: two independent batch-8 train steps on two streams vs back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import wav2sleep_amd as W
dev = torch.device('cuda', 0)
def mk(seed):
    torch.manual_seed(seed)
    m = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer'), 4).to(dev).train()
    return W.FusedTrainStep(m)
t1, t2 = mk(1), mk(2)
x1, y1 = bench.make_batch(8, 960, 4, dev, 1)
x2, y2 = bench.make_batch(8, 960, 4, dev, 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    t1.step(x1, y1); t2.step(x2, y2)
def par():
    with torch.cuda.stream(s1): t1.step(x1, y1)
    with torch.cuda.stream(s2): t2.step(x2, y2)
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('two batch-8 steps back to back: %.2f ms' % timeit(seq))
print('two batch-8 steps on two streams: %.2f ms' % timeit(par))
t16 = mk(3); x, y = bench.make_batch(16, 960, 4, dev, 3)
print('one batch-16 step: %.2f ms' % timeit(lambda: t16.step(x, y)))
