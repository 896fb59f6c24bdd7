"""Localise the transient of tools/flake_probe3.py: the FIRST split-precision forward after exact-fp32 activity is sometimes wrong (all logits),
the next ones -- fresh models too -- are right.  Loop: [one exact-fp32 forward] [three bf16x3 forwards on fresh models, stage by stage]; the
bf16x3 forwards run the engine's stages by hand and keep every stage's output (token tensor per modality slot, transformer output,
SequenceCNN output, logits), so a mismatch names the first stage that differs.

    python tools/flake_probe4.py [--cycles 60] [--no-exact] [--train]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def build(W, dropout=0.0):
    return W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cycles', type=int, default=60)
    ap.add_argument('--no-exact', action='store_true', help='no exact-fp32 forward between the probed ones (control)')
    ap.add_argument('--train', action='store_true', help='three train steps (exact-fp32 engine) before the probed forwards as well')
    a = ap.parse_args()
    import wav2sleep_amd as W
    from wav2sleep_amd import lib
    from oracle import wav2sleep_oracle as O
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    S = 960
    bad = []
    trainer = None
    for cyc in range(a.cycles):
        seed = cyc % 16
        torch.manual_seed(1000 + seed)
        sd = {k: v.detach().clone() for k, v in build(W).state_dict().items()}
        x, _ = O.make_inputs(cfg, 1, S, seed=7000 + seed)
        xd = {k: v.to('cuda') for k, v in x.items()}

        def fresh(exact):
            if exact:
                os.environ['W2S_EXACT_FP32'] = '1'
            else:
                os.environ.pop('W2S_EXACT_FP32', None)
            m = build(W)
            m.load_state_dict(sd)
            m.to('cuda').eval()
            m._ensure_flat()
            return m

        if not a.no_exact:
            m = fresh(True)
            with torch.no_grad():
                m(xd).cpu()
            del m
            if a.train:
                xb, yb = O.make_inputs(cfg, 2, S, seed=8000 + seed)
                mt = build(W, dropout=0.1)
                mt.load_state_dict(sd)
                mt.to('cuda').train()
                tr = W.FusedTrainStep(mt, lr=1e-3, scheduler=False)
                for _ in range(3):
                    tr.step({k: v.to('cuda') for k, v in xb.items()}, yb.to('cuda'))
                torch.cuda.synchronize()
                del tr, mt
        runs = []
        for _ in range(3):
            m = fresh(False)
            eng = m._engine
            with torch.no_grad(), torch.cuda.device(0):
                e = eng.encode(xd, save=False, pack_key=m.param_version())
                tokens = e['tokens'].clone()
                X, _ = eng.mix(e['tokens'], e['keypad'], 0.0, False)
                Xc = X.clone()
                pre, _ = eng.seq(X, X.numel() // e['N'], e['B'], e['S'], 0.0, False)
                logits = torch.empty(e['B'], e['S'], 4, device='cuda')
                lib.head_fwd(pre, 128, eng.P['classifier.weight'], eng.P['classifier.bias'], logits, e['B'] * e['S'], 128, 4, True)
                torch.cuda.synchronize()
            runs.append(dict(tokens=tokens.cpu(), mixer=Xc.cpu(), seq=pre.cpu().clone(), logits=logits.cpu(), sigs=e['sigs']))
            del m, eng, e
        for k in (1, 2):
            diffs = {}
            for name in ('tokens', 'mixer', 'seq', 'logits'):
                if not torch.equal(runs[0][name], runs[k][name]):
                    diffs[name] = float((runs[0][name] - runs[k][name]).abs().max())
            if diffs:
                tk0, tkk = runs[0]['tokens'], runs[k]['tokens']   # [N, D, F]: slot 0 = CLS, 1.. = signals in sorted order
                slots = [d for d in range(tk0.shape[1]) if not torch.equal(tk0[:, d], tkk[:, d])]
                nrows = {d: int((tk0[:, d] != tkk[:, d]).any(dim=-1).sum()) for d in slots}
                bad.append((cyc, k, diffs, 'token slots differing: ' + str(slots), runs[0]['sigs'], nrows))
                print('MISMATCH', bad[-1], flush=True)
    os.environ.pop('W2S_EXACT_FP32', None)
    print(f'RESULT [exact between: {not a.no_exact}, train: {a.train}]: {len(bad)} mismatching comparisons in {a.cycles} cycles')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
