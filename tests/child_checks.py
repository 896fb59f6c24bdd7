"""Full-size / multi-process checks that run in a CHILD process of the pytest run (tests/test_a_children_gpu.py starts them before the
pytest process has touched the GPU, each with a hard timeout that kills the child's process group: a hang costs one test, not the box).

    python tests/child_checks.py <check> <result.json>

Each check writes a JSON dict of measured values; the tolerances live in the pytest file.  The oracle is imported as the checker only.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
SM5 = dict(SM4, **{'EOG-L': 'EOG-L'})
SM6 = dict(SM5, **{'EOG-R': 'EOG-R'})
EOG = {'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}
DEV = 'cuda'
CHECKS = {}


def check(fn):
    CHECKS[fn.__name__] = fn
    return fn


def build(W, signal_map, nc, dropout=0.0):
    return W.Wav2Sleep(W.SignalEncoders(signal_map, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                       W.MultiModalAttentionEmbedder(128, layers=2, dropout=dropout, dim_ff=512, nhead=8),
                       W.SequenceCNN(128, dropout=dropout, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), nc)


def _grad_errors(model, want):
    worst, over = ('', 0.0), []
    for name, p in model.named_parameters():
        w = want[name].double()
        rel = float((p.grad.detach().cpu().double() - w).norm() / (w.norm() + 1e-30))
        if rel > worst[1]:
            worst = (name, rel)
        if rel > 1e-3:
            over.append((name, rel))
    return worst, over


@check
def eog_fullsize_grad():
    """BASELINE configs[3] at full size (EOG-L + EOG-R, 4096 samples per epoch: ten-block encoders, 3.9 M samples per recording, 5 classes),
    B = 1: loss and every gradient tensor vs the oracle's autograd; two runs bit-identical (the 10-block encoders' reproducibility)."""
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    torch.manual_seed(42)
    model = build(W, EOG, 5).to(DEV).train()
    cfg = O.ModelConfig(signal_map=EOG, num_classes=5)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, y = O.make_inputs(cfg, 1, 960, seed=321)
    runs = []
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        logits = model({k: v.to(DEV) for k, v in x.items()})
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 5), y.to(DEV).reshape(-1).long(), ignore_index=-1)
        loss.backward()
        torch.cuda.synchronize()
        runs.append(model._flat_grad.clone())
    want_loss, _, want = O.loss_and_grads(sd, cfg, x, y)
    worst, over = _grad_errors(model, want)
    return dict(bit_reproducible=all(torch.equal(runs[0], r) for r in runs[1:]), loss=float(loss), want_loss=float(want_loss), worst_tensor=worst[0],
                worst_rel_l2=worst[1], over_1e3=over)


@check
def causal_fullsize_grad():
    """`causal: True` (scripts/config/main.yaml:22; causal-padded convolutions, chunk_causal=False) at the benchmark's full length: 4 modalities x
    960 epochs, B = 2, one missing (sample, modality) pair -- loss, arg-max labels and every gradient tensor vs the oracle.  Since round 5 the
    64-channel layers of this variant run on the one-pass backward kernel (w2s_bwd_wide, pad = 2): this is its full-length case (61 440 / 30 720
    positions per sample; the stage checks and goldens stop at a few thousand)."""
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    torch.manual_seed(42)
    model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=True, chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', causal=True, num_layers=2, kernel_size=7, num_dilations=6), 4).to(DEV).train()
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4, causal=True)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, y = O.make_inputs(cfg, 2, 960, seed=99, missing={'THX': [1]})
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        logits = model({k: v.to(DEV) for k, v in x.items()})
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.to(DEV).reshape(-1).long(), ignore_index=-1)
        loss.backward()
        torch.cuda.synchronize()
        runs.append(model._flat_grad.clone())
    want_loss, want_logits, want = O.loss_and_grads(sd, cfg, x, y)
    worst, over = _grad_errors(model, want)
    agree = float((logits.argmax(-1).cpu() == want_logits.argmax(-1)).float().mean())
    return dict(bit_reproducible=torch.equal(runs[0], runs[1]), loss=float(loss), want_loss=float(want_loss), worst_tensor=worst[0],
                worst_rel_l2=worst[1], over_1e3=over, argmax_agreement=agree,
                max_abs_logit_err=float((logits.detach().cpu() - want_logits).abs().max()), max_abs_logit=float(want_logits.abs().max()))


def _oracle_microbatch(job):
    """Runs in a spawned worker process that never touches the GPU: the oracle's forward + autograd backward of one micro-batch."""
    sd, cfg, xm, ym, threads = job
    torch.set_num_threads(threads)
    from oracle import wav2sleep_oracle as O
    return O.loss_and_grads(sd, cfg, xm, ym)


def _logit_errors(got, want):
    """(tests/test_r2_parity_gpu.py logit_errors) max-norm bar 1e-3 of the scale AND element-wise |d| <= 1e-3 |want| + 2e-4 scale"""
    err, scale = (got - want).abs().double(), float(want.abs().max())
    return dict(max_abs=float(err.max()), scale=scale, ok_maxnorm=bool(float(err.max()) <= 1e-3 * scale),
                ok_elementwise=bool((err <= 1e-3 * want.abs().double() + 2e-4 * scale).all()))


@check
def b16_fullsize_grad():
    """The benchmark's own shape (4 modalities x 960 epochs, B = 16, default init, 6 missing (sample, modality) pairs) in ONE child (round 6:
    this check, the default-initialisation B = 16 forward in both arithmetic modes and nothing else rebuild this model):
      * every gradient tensor of ONE backward pass vs the oracle accumulated over 8 micro-batches of 2 (sum_mb (valid_mb / valid_total) *
        grad(mean loss of mb)), loss, arg-max labels;
      * the same weights and batch through the inference forward in the default split-precision mode and under W2S_EXACT_FP32=1: logit
        bars and label flips per recording against the same oracle logits.
    The oracle's micro-batches run in worker processes started BEFORE this process touches the GPU (the pool refuses to start a program
    from a process that has initialised it), side by side with the GPU work."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    torch.manual_seed(42)
    model = build(W, SM4, 4)                     # on the CPU: the reference's default initialisation
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B, S = 16, 960
    x, y = O.make_inputs(cfg, B, S, seed=77, missing={'ABD': [3], 'PPG': [3, 7], 'ECG': [11], 'THX': [0, 15]})
    try:
        mem_gb = os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_AVPHYS_PAGES') / 2 ** 30
    except (ValueError, OSError):
        mem_gb = 64.0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
    workers = max(1, min(4, int(mem_gb // 40), cores // 8))    # a micro-batch of 2 holds ~25 GB of autograd state at its peak
    threads = max(1, min(16, cores // workers))
    pool = cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn'))
    # (copies: a tensor handed to the pool is moved into shared memory in place by the executor's feeder thread -- never share one this thread still uses)
    futs = [pool.submit(_oracle_microbatch, ({k: v.clone() for k, v in sd.items()}, cfg, {k: v[b0:b0 + 2].clone() for k, v in x.items()}, y[b0:b0 + 2].clone(), threads))
            for b0 in range(0, B, 2)]
    # ---- GPU side (the workers are running)
    model.to(DEV).train()
    xd = {k: v.to(DEV) for k, v in x.items()}
    logits = model(xd)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.to(DEV).reshape(-1).long(), ignore_index=-1)
    loss.backward()
    torch.cuda.synchronize()
    got_modes = {}
    for mode in ('bf16x3', 'exact_fp32'):
        if mode == 'exact_fp32':
            os.environ['W2S_EXACT_FP32'] = '1'
        else:
            os.environ.pop('W2S_EXACT_FP32', None)
        m = build(W, SM4, 4)
        m.load_state_dict(sd)
        m.to(DEV).eval()
        with torch.no_grad():
            got_modes[mode] = m(xd).cpu()
        assert m._engine.split_precision == (mode == 'bf16x3')
        del m
    os.environ.pop('W2S_EXACT_FP32', None)
    # ---- the oracle's side
    total = int((y >= 0).sum())
    want = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in sd.items()}
    wl = 0.0
    want_logits = []
    for f, b0 in zip(futs, range(0, B, 2)):
        l, lg, g = f.result()
        want_logits.append(lg)
        w = int((y[b0:b0 + 2] >= 0).sum()) / total
        wl += w * l
        for k in want:
            want[k] += w * g[k].double()
    pool.shutdown()
    want_logits = torch.cat(want_logits)
    worst, over = _grad_errors(model, want)
    agree = float((logits.argmax(-1).cpu() == want_logits.argmax(-1)).float().mean())
    srt = want_logits.sort(-1).values
    gap = srt[..., -1] - srt[..., -2]
    modes = {}
    for mode, got in got_modes.items():
        flips = got.argmax(-1) != want_logits.argmax(-1)
        per = [_logit_errors(got[b], want_logits[b]) for b in range(B)]
        modes[mode] = dict(flips=int(flips.sum()), worst_flip_gap=float(gap[flips].max()) if bool(flips.any()) else 0.0,
                           ok_maxnorm=all(e['ok_maxnorm'] for e in per), ok_elementwise=all(e['ok_elementwise'] for e in per),
                           max_abs=max(e['max_abs'] for e in per), scale=max(e['scale'] for e in per))
    return dict(loss=float(loss), want_loss=float(wl), worst_tensor=worst[0], worst_rel_l2=worst[1], over_1e3=over, argmax_agreement=agree, modes=modes,
                oracle_workers=workers, oracle_threads=threads)


@check
def eog_b16_consistency():
    """BASELINE configs[3] at the batch the metric is quoted on (EOG-L + EOG-R, 16 recordings x 960 epochs = 3.9 M samples each): the 16
    recordings as ONE batch against the same 16 one by one.  Logits bit for bit (batch invariance at this length); the batch's flat
    gradient = sum_b (valid_b / valid_total) * (gradient of recording b alone) -- together with eog_fullsize_grad (B = 1 against the
    oracle's autograd) that pins the B = 16 step the bench's `extra.configs3_eog_b16` leg times."""
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    torch.manual_seed(42)
    model = build(W, EOG, 5).to(DEV).train()   # dropout 0: train mode only selects the saved-forward path
    cfg = O.ModelConfig(signal_map=EOG, num_classes=5)
    B = 16
    x, y = O.make_inputs(cfg, B, 960, seed=1616, missing={'EOG-R': [2, 9], 'EOG-L': [13]})
    xd = {k: v.to(DEV) for k, v in x.items()}
    yd = y.to(DEV)

    def fwd_bwd(xs, ys):
        model.zero_grad(set_to_none=True)
        logits = model(xs)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 5), ys.reshape(-1).long(), ignore_index=-1)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), float(loss), model._flat_grad.detach().double().clone()

    lb, loss_b, gb = fwd_bwd(xd, yd)
    total = int((y >= 0).sum())
    acc = torch.zeros_like(gb)
    singles, wl = [], 0.0
    for i in range(B):
        li, lo, gi = fwd_bwd({k: v[i:i + 1] for k, v in xd.items()}, yd[i:i + 1])
        w = int((y[i] >= 0).sum()) / total
        acc += w * gi
        wl += w * lo
        singles.append(li)
    ls = torch.cat(singles)
    return dict(logits_equal=torch.equal(lb, ls), logits_max_diff=float((lb - ls).abs().max()), loss_batch=loss_b, loss_singles=wl,
                grad_rel_l2=float((gb - acc).norm() / (acc.norm() + 1e-300)), grad_norm=float(acc.norm()), finite=bool(torch.isfinite(gb).all()))


@check
def batch_invariance():
    """A recording's logits must not depend on its batch neighbours (instance / layer norms only): B = 32 vs two halves of 16, B = 5 vs
    single recordings; full length, bit for bit."""
    import bench
    import wav2sleep_amd as W
    torch.manual_seed(42)
    model = build(W, SM4, 4, dropout=0.1).to(DEV).eval()
    dev = torch.device(DEV)
    with torch.no_grad():
        x, _ = bench.make_batch(32, 960, 4, dev, 99)
        x['ECG'][5] = float('-inf')
        x['ABD'][20] = float('-inf')
        full = model(x)
        halves = torch.cat([model({k: v[:16] for k, v in x.items()}), model({k: v[16:] for k, v in x.items()})])
        x5 = {k: v[:5] for k, v in x.items()}
        f5 = model(x5)
        singles = torch.cat([model({k: v[i:i + 1] for k, v in x5.items()}) for i in range(5)])
    return dict(b32_vs_halves_equal=torch.equal(full, halves), b32_max_diff=float((full - halves).abs().max()),
                b5_vs_singles_equal=torch.equal(f5, singles), b5_max_diff=float((f5 - singles).abs().max()))


@check
def five_mod_fullsize_forward():
    """BASELINE configs[4]'s modality set at full length: {ABD, THX, ECG, PPG, EOG-L} (D = 6 tokens; 6-, 8- and 10-block encoders side by
    side), 960 epochs, B = 2, masks drawn by SignalMasker; forward vs oracle."""
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    torch.manual_seed(42)
    model = build(W, SM5, 4).to(DEV).eval()
    cfg = O.ModelConfig(signal_map=SM5, num_classes=4)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, y = O.make_inputs(cfg, 2, 960, seed=555, missing={'EOG-L': [1], 'ABD': [0]})
    with torch.no_grad():
        got = model({k: v.to(DEV) for k, v in x.items()}).cpu()
    want = O.forward(sd, cfg, x)
    err = (got - want).abs()
    scale = float(want.abs().max())
    return dict(max_abs_err=float(err.max()), max_abs_logit=scale, argmax_agreement=float((got.argmax(-1) == want.argmax(-1)).float().mean()),
                worst_floored_rel=float((err / (want.abs() + 0.2 * scale)).max()))


def _oracle_forward(job):
    """Runs in a spawned worker process that never touches the GPU: the oracle's inference forward of one recording."""
    sd, cfg, x, threads = job
    torch.set_num_threads(threads)
    from oracle import wav2sleep_oracle as O
    return O.forward(sd, cfg, x)


@check
def argmax_sweep():
    """How much room do the arg-max labels have?    16 seeds -- each its own default initialisation AND its own full-length
    recording -- at two states of the weights (as initialised; after 10 AdamW steps at lr 1e-3, which move every weight by ~1e-2), in the
    default split-precision mode and under W2S_EXACT_FP32=1: epochs whose label differs from the oracle's (flips), the oracle's top-2 gap at
    every flip, epochs within 10 x the error of a tie, max |d logit|.  The oracle's forward is the same for both modes; the 32 oracle
    forwards run in worker processes (started before this process touches the GPU) beside the GPU work."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    S, NSEED = 960, 16
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
    workers = max(1, min(16, cores // 8))   # (inference forwards of one recording: ~3 GB each; many narrow workers scale better than a few wide ones)
    threads = max(1, min(8, cores // workers))
    pool = cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn'))
    # ---- CPU only so far: every seed's default initialisation and recording; their oracle forwards start now (this spawns all the workers)
    models, xs, futs = [], [], {}
    for seed in range(NSEED):
        torch.manual_seed(1000 + seed)
        models.append(build(W, SM4, 4, dropout=0.1))
        x, _ = O.make_inputs(cfg, 1, S, seed=7000 + seed)
        xs.append(x)
    # What goes to the pool is a COPY (`_to_pool`): torch registers its tensor reductions with multiprocessing, so a submitted CPU tensor's
    # storage is MOVED INTO SHARED MEMORY IN PLACE by the executor's feeder thread, some time after submit() returns -- while this thread may be
    # reading the very same tensor (load_state_dict, .to(device)).  The first version of this check shared `sd` with the pool and then loaded
    # it into a model at once: 1-3 forwards per sweep came out wrong (docs/lab_notes_r6.md section 10: a harness race, not a kernel's)
    def _to_pool(sd, x):
        return {k: v.clone() for k, v in sd.items()}, {k: v.clone() for k, v in x.items()}

    for seed in range(NSEED):
        sd = {k: v.detach().clone() for k, v in models[seed].state_dict().items()}
        sdj, xj = _to_pool(sd, xs[seed])
        futs[(seed, 'init')] = (pool.submit(_oracle_forward, (sdj, cfg, xj, threads)), sd)
    assert len(futs) >= workers   # (every worker process exists before the first GPU call below)

    def run_mode(exact, sd, xd):
        if exact:
            os.environ['W2S_EXACT_FP32'] = '1'
        else:
            os.environ.pop('W2S_EXACT_FP32', None)
        m = build(W, SM4, 4)
        m.load_state_dict(sd)
        m.to(DEV).eval()
        with torch.no_grad():
            out = m(xd).cpu()
        assert m._engine.split_precision == (not exact)
        del m
        return out

    # ---- GPU side
    got = {}
    for seed in range(NSEED):
        model = models[seed].to(DEV).train()
        xd = {k: v.to(DEV) for k, v in xs[seed].items()}
        for state in ('init', 'trained'):
            if state == 'trained':
                tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False)
                xb, yb = O.make_inputs(cfg, 2, S, seed=8000 + seed)
                xb = {k: v.to(DEV) for k, v in xb.items()}
                for _ in range(10):
                    tr.step(xb, yb.to(DEV))
                torch.cuda.synchronize()
                del tr
                sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
                sdj, xj = _to_pool(sd, xs[seed])
                futs[(seed, state)] = (pool.submit(_oracle_forward, (sdj, cfg, xj, threads)), sd)
            sd = futs[(seed, state)][1]
            for exact in (False, True):
                got[(seed, state, exact)] = run_mode(exact, sd, xd)
        models[seed] = None
        del model
    os.environ.pop('W2S_EXACT_FP32', None)
    # ---- compare
    rows, transients = [], []
    for seed in range(NSEED):
        for state in ('init', 'trained'):
            want = futs[(seed, state)][0].result()
            top2 = want.topk(2, dim=-1).values
            gap = (top2[..., 0] - top2[..., 1]).flatten()
            for exact in (False, True):
                g = got[(seed, state, exact)]
                err = float((g - want).abs().max())
                if err > 1e-3 * float(want.abs().max()):
                    # a GROSS mismatch: run the same forward again before judging it, and REPORT it as a transient if the re-run is right (the test
                    # then fails either way: round 6's transients were this harness sharing tensors with the pool -- see `_to_pool` above -- and
                    # none is expected any more)
                    g2 = run_mode(exact, futs[(seed, state)][1], {k: v.to(DEV) for k, v in xs[seed].items()})
                    err2 = float((g2 - want).abs().max())
                    transients.append(dict(seed=seed, state=state, mode='exact_fp32' if exact else 'bf16x3', first_err=err, rerun_err=err2,
                                           rerun_bit_equal_to_first=bool(torch.equal(g, g2))))
                    g, err = g2, err2
                flip = (g.argmax(-1) != want.argmax(-1)).flatten()
                rows.append(dict(seed=seed, state=state, mode='exact_fp32' if exact else 'bf16x3', flips=int(flip.sum()),
                                 flip_gaps=[float(v) for v in gap[flip]], max_abs_err=err, max_abs_logit=float(want.abs().max()),
                                 near_ties=int((gap < 10 * err).sum()), min_gap=float(gap.min()), median_gap=float(gap.median())))
    pool.shutdown()
    agg = {}
    for mode in ('bf16x3', 'exact_fp32'):
        for state in ('init', 'trained'):
            rs = [r for r in rows if r['mode'] == mode and r['state'] == state]
            agg[f'{mode}/{state}'] = dict(epochs=S * len(rs), flips=sum(r['flips'] for r in rs), near_ties=sum(r['near_ties'] for r in rs),
                                          max_abs_err=max(r['max_abs_err'] for r in rs), max_rel_err=max(r['max_abs_err'] / r['max_abs_logit'] for r in rs),
                                          worst_flip_gap_over_err=max([g / r['max_abs_err'] for r in rs for g in r['flip_gaps']], default=0.0),
                                          min_gap=min(r['min_gap'] for r in rs))
    return dict(summary=agg, rows=rows, transients=transients, oracle_workers=workers, oracle_threads=threads)


@check
def nccl_forced_collectives():
    """The RCCL + side-stream path on one GPU: FusedTrainStep with backend 'nccl', world size 1, W2S_FORCE_COLLECTIVES=1 against the same
    steps without any collective: flat gradient and parameters bit for bit (a SUM over one rank is the identity)."""
    import torch.distributed as dist
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29591')
    os.environ['RANK'], os.environ['WORLD_SIZE'] = '0', '1'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = O.make_state_dict(cfg, seed=61)
    batches = [O.make_inputs(cfg, 3, 40, seed=600 + k, missing={'THX': [1]} if k == 1 else None) for k in range(3)]

    def run(force):
        if force:
            os.environ['W2S_FORCE_COLLECTIVES'] = '1'
        else:
            os.environ.pop('W2S_FORCE_COLLECTIVES', None)
        model = build(W, SM4, 4)
        model.load_state_dict(sd)
        model.to(dev).train()
        tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False)
        assert tr.reducer.force == force and (tr.reducer.stream is not None) == force
        grads = []
        for x, y in batches:
            tr.step({k: v.to(dev) for k, v in x.items()}, y.to(dev))
            grads.append(model._flat_grad.clone())
        _, _, cm = tr.metrics()
        torch.cuda.synchronize()
        return grads, model._flat.clone(), cm.clone()

    g0, p0, cm0 = run(False)
    dist.init_process_group('nccl', device_id=dev)
    g1, p1, cm1 = run(True)
    dist.destroy_process_group()
    return dict(grads_equal=all(torch.equal(a, b) for a, b in zip(g0, g1)), params_equal=torch.equal(p0, p1), cm_equal=torch.equal(cm0, cm1),
                backend='nccl', steps=len(batches))


if __name__ == '__main__':
    name, out = sys.argv[1], sys.argv[2]
    t0 = time.time()
    res = CHECKS[name]()
    res['seconds'] = round(time.time() - t0, 1)
    with open(out, 'w') as f:
        json.dump(res, f)
    print(json.dumps(res))
