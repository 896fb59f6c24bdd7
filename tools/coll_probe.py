"""Where does the cost of the collective path go at world size 1?  (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
import bench
import wav2sleep_amd as W
from wav2sleep_amd import ddp
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
torch.manual_seed(42)
SM = bench.SIGNAL_MAP
model = W.Wav2Sleep(W.SignalEncoders(SM, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                    W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                    W.SequenceCNN(128, dropout=0.1, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to(dev).train()
x, y = bench.make_batch(16, 960, 4, dev, 1234)

def run(tag, force, patch=None, side=True):
    if force: os.environ['W2S_FORCE_COLLECTIVES'] = '1'
    else: os.environ.pop('W2S_FORCE_COLLECTIVES', None)
    tr = W.FusedTrainStep(model)
    if not side: tr.reducer.stream = None
    real = dist.all_reduce
    if patch == 'noop':
        dist.all_reduce = lambda *a, **k: None
    for _ in range(3): tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    dist.all_reduce = real
    print(f'{tag:45s} {1e3 * dt:.2f} ms/step', flush=True)

run('no collectives', False)
run('collectives (side stream)', True)
run('hooks on, all_reduce patched to a no-op', True, patch='noop')
run('collectives, no side stream (async_op)', True, side=False)
run('no collectives', False)
dist.destroy_process_group()
