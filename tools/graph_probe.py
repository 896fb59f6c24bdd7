"""Is the inference forward hipGraph-capturable as it is (ctypes launches on torch's streams, four encoder streams), and what does a
captured forward buy at small batch, where the host (~250 launches x ~18 us) is slower than the device?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W
dev = torch.device('cuda')
torch.manual_seed(42)
model = W.Wav2Sleep(W.SignalEncoders(bench.SIGNAL_MAP, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to(dev).eval()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for B in (1, 2, 4):
    x, _ = bench.make_batch(B, 960, 4, dev, 7 + B)
    with torch.no_grad():
        ref = model(x).clone()
        eager_ms = timeit(lambda: model(x))
        static_x = {k: v.clone() for k, v in x.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(static_x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_out = model(static_x)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(static_out, ref)
        x2, _ = bench.make_batch(B, 960, 4, dev, 70 + B)
        for k in static_x: static_x[k].copy_(x2[k])
        g.replay(); torch.cuda.synchronize()
        same2 = torch.equal(static_out, model(x2))
        graph_ms = timeit(lambda: g.replay())
    print(f'B={B}: eager {eager_ms:.2f} ms, captured graph {graph_ms:.2f} ms per forward; replay == eager bits: {same}, after new inputs: {same2}', flush=True)
