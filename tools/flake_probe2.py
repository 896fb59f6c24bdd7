"""The arg-max sweep's own sequence for ONE seed, repeated: build -> forwards (both modes) -> 10 train steps -> forwards (both modes), every
output compared bit for bit with the first repetition's (round 6: one exact-fp32 forward of the sweep came back 0.3 off on trained weights).

    python tools/flake_probe2.py [--seed 9] [--reps 30] [--load 4]
"""
import argparse
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.flake_probe import SM4, build, burn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=9)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--load', type=int, default=4)
    ap.add_argument('--extra', type=int, default=4, help='additional exact-mode forwards of the trained weights per repetition')
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
    S = 960
    x, _ = O.make_inputs(cfg, 1, S, seed=7000 + a.seed)
    xd = {k: v.to('cuda') for k, v in x.items()}
    xb, yb = O.make_inputs(cfg, 2, S, seed=8000 + a.seed)
    xb = {k: v.to('cuda') for k, v in xb.items()}
    yb = yb.to('cuda')

    def run_mode(exact, sd):
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

    ref = {}
    bad = []
    for rep in range(a.reps):
        torch.manual_seed(1000 + a.seed)
        model = build(W, dropout=0.1).to('cuda').train()
        outs = {}
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        outs['init/bf16x3'] = run_mode(False, sd)
        outs['init/exact'] = run_mode(True, sd)
        tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False)   # (built while W2S_EXACT_FP32=1 is set, as in the sweep)
        for _ in range(10):
            tr.step(xb, yb)
        torch.cuda.synchronize()
        del tr
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        outs['trained/weights'] = torch.cat([v.flatten().float() for v in sd.values()])
        outs['trained/bf16x3'] = run_mode(False, sd)
        outs['trained/exact'] = run_mode(True, sd)
        for k in range(a.extra):
            outs[f'trained/exact+{k}'] = run_mode(True, sd)
        del model
        for key, v in outs.items():
            base = key.split('+')[0]
            if base not in ref:
                ref[base] = v
            elif not torch.equal(v, ref[base]):
                d = (v - ref[base]).abs()
                bad.append((rep, key, int((v != ref[base]).sum()), float(d.max())))
                print(f'rep {rep} {key}: {bad[-1][2]} of {v.numel()} values differ, max |d| {bad[-1][3]:.3e}', flush=True)
                torch.save(dict(sd=sd, out=v, ref=ref[base]), f'gpurun_out/flake_{rep}_{key.replace("/", "_")}.pt')
    if stop is not None:
        stop.set()
        for p in procs:
            p.join(timeout=30)
    print(f'RESULT [seed {a.seed}, {a.reps} reps, load {a.load}]: {len(bad)} mismatches', bad[:12])
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
