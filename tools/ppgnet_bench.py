"""SleepPPGNet train step on the generic path (wav2sleep_amd/generic.py tape + GenericTrainStep): ms per step and recordings/s at batch B on
synthetic 10-hour PPG (the model of models/ppgnet.py as the reference trains it: BatchNorm on batch statistics, LeakyReLU, dropout 0.2).

    python tools/ppgnet_bench.py [--batch 8] [--steps 10] [--warmup 3] [--infer]
    rocprofv3 --kernel-trace --stats -d gpurun_out/ppg_prof -- python3 tools/ppgnet_bench.py --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--infer', action='store_true', help='time the eval-mode forward instead of the train step')
    a = ap.parse_args()
    import wav2sleep_amd as W
    from wav2sleep_amd.trainer import GenericTrainStep
    torch.manual_seed(0)
    model = W.SleepPPGNet().to('cuda')
    g = torch.Generator().manual_seed(1)
    x = torch.randn(a.batch, model.INPUT_LENGTH, generator=g).to('cuda')
    y = torch.randint(0, 4, (a.batch, 1200), generator=g).float().to('cuda')
    if a.infer:
        model.eval()
        run = lambda: model(x)
    else:
        model.train()
        step = GenericTrainStep(model, lr=1e-3, scheduler=False)
        run = lambda: step.step(x, y)
    with torch.no_grad() if a.infer else torch.enable_grad():
        for _ in range(a.warmup):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            out = run()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / a.steps
    ms = e0.elapsed_time(e1) / a.steps
    rec = {'workload': f"SleepPPGNet {'eval forward' if a.infer else 'train step'} batch {a.batch} x 10 h PPG (1 228 800 samples)", 'ms_per_step': round(ms, 3),
           'ms_per_step_host': round(wall, 3), 'recordings_per_s': round(a.batch / ms * 1e3, 2), 'steps': a.steps, 'warmup': a.warmup,
           'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    if not a.infer:
        rec['loss'] = float(out['loss'])
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
