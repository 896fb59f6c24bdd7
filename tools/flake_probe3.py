"""The arg-max sweep's GPU side, self-checking: every inference forward of tests/child_checks.py argmax_sweep is run TWICE back to back and the two
outputs compared bit for bit (round 6: transient gross mismatches -- 1 exact-fp32 forward in one sweep, 2 bf16x3 forwards in another, none in
~900 forwards of the simpler probes).  No oracle result is needed to see a transient; the oracle's worker processes can still be started to
reproduce the sweep's host load.

    python tools/flake_probe3.py [--seeds 16] [--pool 4] [--no-train] [--repeat 2]
    W2S_MULTI_STREAM=0 / W2S_NO_LINEAR_PF=1 / W2S_NO_SEQCONV=1 python tools/flake_probe3.py ...      # bisecting
"""
import argparse
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def _oracle_forward(job):
    sd, cfg, x, threads = job
    torch.set_num_threads(threads)
    from oracle import wav2sleep_oracle as O
    return float(O.forward(sd, cfg, x).abs().max())


def build(W, dropout=0.0):
    return W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=16)
    ap.add_argument('--pool', type=int, default=4)
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--keep-models', action='store_true', help='build every seed\'s model on the CPU first, as the sweep does')
    ap.add_argument('--share', action='store_true', help='hand the SAME sd / x tensors to the pool and to the model (the harness race of round 6); default: copies')
    a = ap.parse_args()
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    S = 960
    pool = cf.ProcessPoolExecutor(max_workers=a.pool, mp_context=mp.get_context('spawn')) if a.pool else None
    models, xs = [], []
    for seed in range(a.seeds):
        torch.manual_seed(1000 + seed)
        models.append(build(W, dropout=0.1) if a.keep_models else None)
        xs.append(O.make_inputs(cfg, 1, S, seed=7000 + seed)[0])
    futs = []
    if pool:
        for seed in range(min(a.seeds, 8)):
            torch.manual_seed(1000 + seed)
            m0 = models[seed] if a.keep_models else build(W, dropout=0.1)
            futs.append(pool.submit(_oracle_forward, ({k: v.detach().clone() for k, v in m0.state_dict().items()}, cfg, xs[seed], 16)))

    def run_mode(exact, sd, xd):
        if exact:
            os.environ['W2S_EXACT_FP32'] = '1'
        else:
            os.environ.pop('W2S_EXACT_FP32', None)
        m = build(W)
        m.load_state_dict(sd)
        m.to('cuda').eval()
        with torch.no_grad():
            out = m(xd).cpu()
        del m
        return out

    bad, total = [], 0
    for seed in range(a.seeds):
        if a.keep_models:
            model = models[seed].to('cuda').train()
        else:
            torch.manual_seed(1000 + seed)
            model = build(W, dropout=0.1).to('cuda').train()
        xd = {k: v.to('cuda') for k, v in xs[seed].items()}
        for state in ('init', 'trained'):
            if state == 'trained':
                if a.no_train:
                    continue
                tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False)
                xb, yb = O.make_inputs(cfg, 2, S, seed=8000 + seed)
                xb = {k: v.to('cuda') for k, v in xb.items()}
                for _ in range(10):
                    tr.step(xb, yb.to('cuda'))
                torch.cuda.synchronize()
                del tr
            sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            if pool:
                futs.append(pool.submit(_oracle_forward, (sd, cfg, xs[seed], 16) if a.share else ({k: v.clone() for k, v in sd.items()}, cfg, {k: v.clone() for k, v in xs[seed].items()}, 16)))
            for exact in (False, True):
                outs = [run_mode(exact, sd, xd) for _ in range(a.repeat)]
                total += a.repeat
                for k in range(1, a.repeat):
                    if not torch.equal(outs[0], outs[k]):
                        d = (outs[0] - outs[k]).abs()
                        bad.append((seed, state, 'exact' if exact else 'bf16x3', k, int((outs[0] != outs[k]).sum()), float(d.max())))
                        print('MISMATCH', bad[-1], flush=True)
        models[seed] = None
        del model
    os.environ.pop('W2S_EXACT_FP32', None)
    if pool:
        for f in futs:
            f.result()
        pool.shutdown()
    print(f'RESULT [{os.environ.get("W2S_MULTI_STREAM", "ms")}, pool {a.pool}, train {not a.no_train}, keep {a.keep_models}]: {len(bad)} mismatching pairs in {total} forwards', bad[:10])
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
