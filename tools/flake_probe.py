"""Is the inference forward bit-reproducible over many FRESH model builds?  (round 6: one of 32 exact-fp32 forwards of the arg-max sweep came
back 0.3 off; tools/poison_probe.py found no read of unwritten memory.)

    python tools/flake_probe.py [--mode exact|bf16x3] [--trials 200] [--load N] [--mix]

--load N : N worker processes burning host cores (torch CPU GEMMs, started before this process touches the GPU) -- the sweep ran its oracle
           forwards beside the GPU work, which stretches the gaps between launches
--mix    : between two probed forwards run what the sweep ran there: a forward in the OTHER arithmetic mode and ten train steps
"""
import argparse
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def burn(stop):
    torch.set_num_threads(16)
    a = torch.randn(2048, 2048)
    while not stop.is_set():
        a = (a @ a).clamp_(-1, 1)


def build(W, dropout=0.0):
    return W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='exact')
    ap.add_argument('--trials', type=int, default=200)
    ap.add_argument('--load', type=int, default=0)
    ap.add_argument('--mix', action='store_true')
    ap.add_argument('--epochs', type=int, default=960)
    a = ap.parse_args()
    stop, procs = None, []
    if a.load:
        ctx = mp.get_context('spawn')
        stop = ctx.Event()
        procs = [ctx.Process(target=burn, args=(stop,)) for _ in range(a.load)]
        for p in procs:
            p.start()
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    x, _ = O.make_inputs(cfg, 1, a.epochs, seed=7009)
    xd = {k: v.to('cuda') for k, v in x.items()}
    xb, yb = O.make_inputs(cfg, 2, a.epochs, seed=8009)
    xb = {k: v.to('cuda') for k, v in xb.items()}
    yb = yb.to('cuda')
    torch.manual_seed(1009)
    sd = {k: v.detach().clone() for k, v in build(W).state_dict().items()}

    def fwd(exact):
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

    exact = a.mode == 'exact'
    ref = fwd(exact)
    bad = []
    trainer_model = None
    for t in range(a.trials):
        if a.mix:
            fwd(not exact)
            if trainer_model is None:
                trainer_model = build(W, dropout=0.1).to('cuda').train()
                tr = W.FusedTrainStep(trainer_model, lr=1e-3, scheduler=False)
            for _ in range(3):
                tr.step(xb, yb)
            torch.cuda.synchronize()
        out = fwd(exact)
        if not torch.equal(out, ref):
            d = (out - ref).abs()
            bad.append((t, int((out != ref).sum()), float(d.max())))
            print(f'trial {t}: {bad[-1][1]} of {out.numel()} logits differ, max |d| {bad[-1][2]:.3e}', flush=True)
    if stop is not None:
        stop.set()
        for p in procs:
            p.join(timeout=30)
    print(f'RESULT [{a.mode}, load {a.load}, mix {a.mix}]: {len(bad)} of {a.trials} forwards differ from the first', bad[:10])
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
