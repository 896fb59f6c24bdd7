"""Does any kernel read memory nobody wrote?  (round 6: one of 32 exact-fp32 forwards of tests/child_checks.py argmax_sweep came back with a logit
error of 0.3 -- bit-identical inputs and weights had given 3e-5 in the previous session.)

torch.empty hands out whatever the caching allocator's free blocks hold.  In a steady loop that is the previous, identical launch's data, which
hides a read of unwritten memory perfectly.  Here the allocator's free pool is POISONED before every run (large tensors filled with NaN, or
with 1e30, released back to the pool -- not to the driver), and every run is compared bit for bit with the first one:

    python tools/poison_probe.py [--mode bf16x3|exact|both] [--batch 1] [--epochs 960] [--trials 6] [--train]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def poison(value, gb):
    torch.cuda.synchronize()
    blocks = []
    try:
        for _ in range(int(gb)):
            blocks.append(torch.full((1 << 28,), value, device='cuda', dtype=torch.float32))   # 1 GiB each
    except RuntimeError:
        pass
    # small blocks too: the allocator keeps separate pools for allocations below 1 MB
    small = [torch.full((n,), value, device='cuda', dtype=torch.float32) for n in (64, 256, 1024, 4096, 16384, 65536, 200000) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='both')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--epochs', type=int, default=960)
    ap.add_argument('--trials', type=int, default=6)
    ap.add_argument('--gb', type=int, default=24)
    ap.add_argument('--train', action='store_true')
    ap.add_argument('--seed', type=int, default=1009)
    a = ap.parse_args()
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    x, y = O.make_inputs(cfg, a.batch, a.epochs, seed=7009, missing={'THX': [0]} if a.batch > 1 else None)
    xd = {k: v.to('cuda') for k, v in x.items()}
    yd = y.to('cuda')
    bad = 0
    for mode in (['bf16x3', 'exact'] if a.mode == 'both' else [a.mode]):
        if mode == 'exact':
            os.environ['W2S_EXACT_FP32'] = '1'
        else:
            os.environ.pop('W2S_EXACT_FP32', None)
        torch.manual_seed(a.seed)
        sd = None
        ref = None
        for trial in range(a.trials):
            value = [0.0, float('nan'), 1e30, float('nan'), -1e30, float('nan')][trial % 6]
            torch.cuda.empty_cache()
            if trial:
                poison(value, a.gb)
            model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                                W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                                W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)
            if sd is None:
                sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            model.load_state_dict(sd)
            model.to('cuda')
            if a.train:
                model.train()
                logits = model(xd)
                loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), yd.reshape(-1).long(), ignore_index=-1)
                loss.backward()
                torch.cuda.synchronize()
                out = torch.cat([logits.detach().flatten(), model._flat_grad.detach().flatten()]).cpu()
            else:
                model.eval()
                with torch.no_grad():
                    out = model(xd).flatten().cpu()
            nan = int(torch.isnan(out).sum())
            if ref is None:
                ref = out
                print(f'[{mode}] trial 0 (clean pool): {out.numel()} values, NaN {nan}, max |v| {float(out.abs().max()):.4f}', flush=True)
            else:
                diff = int((out != ref).sum()) if not nan else -1
                worst = float((out - ref).abs().max()) if not nan else float('nan')
                ok = nan == 0 and diff == 0
                bad += not ok
                print(f'[{mode}] trial {trial} (pool poisoned with {value}): NaN {nan}, elements differing from trial 0: {diff}, max |d| {worst:.3e}  {"ok" if ok else "<-- READS UNWRITTEN MEMORY"}',
                      flush=True)
            del model
    os.environ.pop('W2S_EXACT_FP32', None)
    print('RESULT:', 'clean' if not bad else f'{bad} poisoned runs differ')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
