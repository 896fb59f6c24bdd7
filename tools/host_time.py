"""Host side of the train step: how long does the Python / ctypes enqueue of one step take, against the GPU's step time?
(tools/host_time.py on the GPU box)  If the two are close the step is launch-bound wherever its kernels are short."""
import sys, time, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import wav2sleep_amd as W

dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
torch.manual_seed(42)
model, trainer = bench.build_trainer(W, dict(bench.SIGNAL_MAP), 4, False, dev)
x, y = bench.make_batch(16, 960, 4, dev, 1234)
for _ in range(4):
    trainer.step(x, y)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); trainer.step(x, y); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'host enqueue per step: {1e3 * sum(host) / 10:.2f} ms (min {1e3 * min(host):.2f}, max {1e3 * max(host):.2f}); wall per step {1e3 * (t2 - t0) / 10:.2f} ms; '
      f'host finished {1e3 * (t2 - t1):.2f} ms before the GPU')
# the host alone: enqueue a step while the GPU is still busy with earlier ones is what the numbers above show; now with the GPU idle first
torch.cuda.synchronize()
a = time.perf_counter(); trainer.step(x, y); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
print(f'one step from an idle GPU: host enqueue {1e3 * (b - a):.2f} ms, GPU done {1e3 * (c - a):.2f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    trainer.step(x, y)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue()[:6000])
