"""Headline benchmark: overnight-recordings/sec for a full train step (fwd + masked CE + bwd + clip + AdamW
[+ RCCL all-reduce]) at per-GPU batch 16 -- BASELINE.json `metric`, workload = configs[1] (4-modality
ABD+THX+ECG+PPG, 8 h = 960 epochs, 4 classes, batch 16), synthetic z-scored-like inputs resident in HBM,
reference weights initialisation (random), fp32 end to end (the reference trains `32-true`).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  Extra legs, outside the timed region: `roofline` (per-launch HIP-event timing of
the dominant kernel in one extra step) and `cpu_baseline` (the CPU oracle = the stock-ATen path the reference runs,
timed on this box's host cores on a bounded sample: the same workload at micro-batch 2).
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

SIGNAL_MAP = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
SPE = {'ABD': 256, 'THX': 256, 'ECG': 1024, 'PPG': 1024}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # MI355X_MICROARCH.md: what a float4 grid-stride copy reaches on this part (the practical ceiling)
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA peak (the W2S_EXACT_FP32=1 kernels)
MFMA_SPLIT_PEAK_TF = 2500.0 / 3   # split precision (the default): every product is three dense-bf16 MFMAs (2.5 PFLOP/s dense) -> fp32-equivalent peak


def pmc_key(name):
    """rocprofv3 kernel name -> the LaunchTimer key (conv_wide: <CI, NW, NP, STRIDE, PRO, EPI, MT, PD> -> <CI, NW, STRIDE, PRO, EPI>).
    Since round 4 the four persistent statistics producers carry a trailing FIN template argument (1: the instantiation with the in-kernel
    statistics finalisation); the keys do not."""
    name = name.replace('conv_wide_np_kernel', 'conv_wide_kernel')   # (the forward instances' entry point without packed-fp32 selection)
    m = re.match(r'wgrad_bf_pf_kernel<(\d+), (\d+), (\d+), (\d+), \d+>$', name)   # round 6: the pipelined form <OW, CT, NW, STRIDE, TM> of wgrad_bf_kernel<OW, 1, CT, NW, STRIDE, -1, -1>
    if m:
        return f'wgrad_bf_kernel<{m[1]}, 1, {m[2]}, {m[3]}, {m[4]}, -1, -1>'
    m = re.match(r'(conv_fwd_bf|bwd_fused_bf|bwd_fused|conv_wide|bwd_wide)_kernel<(.*), [01]>$', name)
    if m and m[2].count(',') + 2 == {'conv_fwd_bf': 6, 'bwd_fused_bf': 8, 'bwd_fused': 6, 'conv_wide': 11, 'bwd_wide': 11}[m[1]]:   # (names of this round: drop FIN)
        name = f'{m[1]}_kernel<{m[2]}>'
    m = re.match(r'conv_wide_kernel<(\d+), (\d+), \d+, (\d+), (\d+), (\d+), .*>', name)   # <CI, NW, NP, STRIDE, PRO, EPI, MT, PD, UP2, CZ>
    if m:
        return f'conv_wide_kernel<{m[1]}, {m[2]}, {m[3]}, {m[4]}, {m[5]}>'
    m = re.match(r'bwd_wide_kernel<(\d+), (\d+), (\d+), \d+, \d+, \d+, \d+, \d+, (\d+), (\d+)(?:, \d+)?>', name)   # <CO, CI, HST, MT, NWC, IB, CB, PD, UP2, RD[, CZ]>
    if m:
        return f'bwd_wide_kernel<{m[1]}, {m[2]}, {m[3]}, {int(m[4]) + 1}{", rd" if m[5] == "1" else ""}>'
    m = re.match(r'wgrad_wide_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), .*>', name)   # <CO, CI, STRIDE, PG, PH, MT, NWC, IB, CB, PD>
    return f'wgrad_wide_kernel<{m[1]}, {m[2]}, {m[3]}, {m[4]}, {m[5]}>' if m else name


def family_of(key):
    """Kernel family of a LaunchTimer key (DESIGN.md section 4 names)."""
    for prefix, fam in (('bwd_fused_bf_kernel', 'fused backward <=32ch (dgrad+wgrad)'), ('conv_fwd_bf_kernel', 'persistent forward <=32ch'),
                        ('conv_wide_kernel', 'wide conv >=64ch (fwd + dgrad)'), ('bwd_wide_kernel', 'fused backward 64ch (dgrad+wgrad)'),
                        ('wgrad', 'weight gradient >=64ch / k1 / dilated'),
                        ('conv_cl_kernel', 'generic conv (1x1, dilated, downsample, UP2 dgrad)'), ('linear_pf_kernel', 'generic conv (1x1, dilated, downsample, UP2 dgrad)'),
                        ('seq_conv_kernel', 'generic conv (1x1, dilated, downsample, UP2 dgrad)')):
        if key.startswith(prefix):
            return fam
    return 'other'


FAMILY_KEYS = {'fused backward <=32ch (dgrad+wgrad)': 'bwd_le32', 'persistent forward <=32ch': 'fwd_le32', 'wide conv >=64ch (fwd + dgrad)': 'wide_ge64',
               'fused backward 64ch (dgrad+wgrad)': 'bwd64_onepass', 'weight gradient >=64ch / k1 / dilated': 'wgrad', 'generic conv (1x1, dilated, downsample, UP2 dgrad)': 'generic_conv'}


def step_traffic(profiles):
    """HBM bytes one step moves: the committed PMC bytes/launch x the committed kernel-trace launch counts (all kernels, not only the
    GEMM-shaped ones).  None when the two files are not both there."""
    import csv
    import glob
    tr = sorted(glob.glob(os.path.join(profiles, 'r*_pmc_traffic.json')))
    ks = sorted(glob.glob(os.path.join(profiles, 'r*_kernel_stats_bench_b16_single_stream.csv')))
    if not tr or not ks:
        return None
    per = json.load(open(tr[-1]))
    rows = list(csv.DictReader(ln for ln in open(ks[-1]) if not ln.startswith('#')))
    steps = None
    for ln in open(ks[-1]):
        m = re.search(r'\((\d+) train steps', ln) if ln.startswith('#') else None
        if m:
            steps = int(m[1])
    if steps is None:
        return None
    tot = 0.0
    for r in rows:
        k = r['Name'].split('(')[0].replace('void ', '').strip()
        if k in per:
            tot += per[k]['hbm_bytes_per_launch'] * int(r['Calls']) / steps
    return int(tot)


def make_batch(batch, epochs, num_classes, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    x = {s: torch.randn(batch, epochs * SPE[s], device=device, generator=g) for s in SIGNAL_MAP}
    y = torch.randint(0, num_classes, (batch, epochs), device=device, generator=g).float()
    y[torch.rand(batch, epochs, device=device, generator=g) < 0.1] = -1.0
    return x, y


def host_cpu():
    """(model string, physical cores, logical cpus) of this box from /proc/cpuinfo."""
    model, cores, logical = 'unknown', set(), 0
    try:
        phys = core = None
        for ln in open('/proc/cpuinfo'):
            k, _, v = ln.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'processor':
                logical += 1
            elif k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or logical or os.cpu_count() or 1), (logical or os.cpu_count() or 1)


def cpu_baseline(epochs, num_classes, signal_map, budget_s=60.0):
    """CPU 'port' baseline: the oracle's full train step -- stock ATen CPU kernels, fp32, what the reference runs -- on this
    box's host cores, at micro-batch 2 with gradient accumulation (batch 16 x 8 h does not fit host memory in fp32: ~5.3 GB per
    recording), as the reference itself reaches its effective batch (scripts/train.py:59-76).  Bounded sample: 1 warm-up micro-batch
    (oneDNN primitive caching) + >= 3 timed micro-batches of the same workload; recordings/s = 2 / median micro-batch time (the
    clip + AdamW of the 16-recording step is < 0.1 % of it and is included in one of them)."""
    from oracle import wav2sleep_oracle as O
    model, cores, logical = host_cpu()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else logical
    # thread count: "all physical cores" over-subscribes the memory-bound CPU path on a 128-core host (SURVEY measured 0.26 recordings/s on
    # 8 cores, round 3 0.124 on 128).  Probe 4 / 8 / 16 / 32 / 64 / all cores once on a short sample of the same workload (1/8 of the epochs,
    # forward + backward) and run the bounded sample at the fastest count: the baseline is the reference's CPU path at ITS best here.
    cfg_p = O.ModelConfig(signal_map=signal_map, num_classes=num_classes)
    sd_p = O.make_state_dict(cfg_p, seed=42)
    xp, yp = O.make_inputs(cfg_p, 2, max(8, epochs // 8), seed=99)
    probe = {}
    for th in sorted({min(t, cores, avail) for t in (4, 8, 16, 32, 64, cores)}):
        torch.set_num_threads(max(1, th))
        O.loss_and_grads(sd_p, cfg_p, xp, yp)   # warm-up at this count
        t1 = time.time()
        O.loss_and_grads(sd_p, cfg_p, xp, yp)
        probe[max(1, th)] = round(time.time() - t1, 3)
    del sd_p, xp, yp
    threads = min(probe, key=probe.get)
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=num_classes)
    sd = O.make_state_dict(cfg, seed=42)
    mb = 2
    x, y = O.make_inputs(cfg, mb, epochs, seed=1234)
    t0 = time.time()
    # the short probe only ranks the counts roughly: its two fastest are timed once at the FULL length (the first run doubles as warm-up)
    full = {}
    for th in sorted(probe, key=probe.get)[:2]:
        torch.set_num_threads(th)
        O.loss_and_grads(sd, cfg, x, y)
        t1 = time.time()
        O.loss_and_grads(sd, cfg, x, y)
        full[th] = round(time.time() - t1, 2)
    threads = min(full, key=full.get)
    torch.set_num_threads(threads)
    times = []
    state = {}
    while len(times) < 3 or (time.time() - t0 < budget_s and len(times) < 8):
        t1 = time.time()
        if len(times) == 0:
            O.train_step(sd, cfg, x, y, state)   # forward + backward + clip + AdamW
        else:
            O.loss_and_grads(sd, cfg, x, y)      # an accumulation micro-batch: forward + backward
        times.append(time.time() - t1)
    times.sort()
    med = times[len(times) // 2]
    return {'value': round(mb / med, 4), 'unit': 'recordings/s', 'cores': threads, 'kind': 'port', 'cpu': model, 'physical_cores': cores,
            'logical_cpus': logical, 'thread_probe_s': probe, 'thread_probe_full_length_s': full,
            'sample': f'full train step of the same {len(signal_map)}-modality {epochs}-epoch workload as micro-batches of {mb} with gradient accumulation '
                      f'(scripts/train.py:59-76): 1 warm-up + median of {len(times)} timed micro-batches ({med:.2f} s each), {threads} threads = the fastest of the probed counts {sorted(probe)} (short probe, its two fastest re-timed at full length); '
                      f'oracle/wav2sleep_oracle.py on stock torch CPU ops'}


def kappa_parity(model, signal_map, num_classes, epochs, dev, causal=False):
    """Cohen's kappa between the HIP path's and the CPU oracle's stage predictions on ONE synthetic overnight recording (inference
    forward, same weights, same input), plus the logit errors: BASELINE.json's metric names kappa parity.  Checker use of oracle/."""
    from oracle import wav2sleep_oracle as O
    from wav2sleep_amd.stats import cohens_kappa
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=num_classes, causal=causal)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, _ = O.make_inputs(cfg, 1, epochs, seed=4321)
    was_training = model.training
    model.eval()
    xd = {k: v.to(dev) for k, v in x.items()}
    with torch.no_grad():
        got = model(xd).cpu()
    model.train(was_training)
    want = O.forward(sd, cfg, x)
    top2 = want.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]).flatten()
    pw = want.argmax(-1).flatten()

    def figures(got):
        pg = got.argmax(-1).flatten()
        cm = torch.zeros(num_classes, num_classes, dtype=torch.int64)
        cm.index_put_((pw, pg), torch.ones_like(pw), accumulate=True)
        err = (got - want).abs()
        # the label margin: how close the oracle's own top-2 logits are, in units of this build's logit error -- an epoch whose gap is within
        # 10 x the error is one the next forward change could flip (the suite demands identical arg-max labels)
        emax = float(err.max())
        agree = pg == pw
        return {'near_ties': int((gap < 10.0 * emax).sum()), 'near_tie_threshold': 10.0 * emax,
                'min_top2_gap_among_agreeing': float(gap[agree].min()) if bool(agree.any()) else None,
                'kappa_build_vs_oracle': round(float(cohens_kappa(cm.numpy(), num_classes)), 6), 'argmax_agreement': round(float(agree.float().mean()), 6),
                'flips': int((~agree).sum()), 'max_abs_logit_err': emax,
                'max_rel_logit_err_elementwise': float((err / want.abs().clamp_min(1e-3 * float(want.abs().max()))).max())}

    res = figures(got)
    # the same weights and recording through the fp32-MFMA kernels (W2S_EXACT_FP32=1 is read when a model's engine is built): the mode whose
    # error is ~5 x smaller, beside the default one (tests/child_checks.py `argmax_sweep` runs 16 seeds x 2 states of both)
    import wav2sleep_amd as W
    prev = os.environ.get('W2S_EXACT_FP32')
    os.environ['W2S_EXACT_FP32'] = '1'
    try:
        m2 = W.Wav2Sleep(W.SignalEncoders(signal_map, 128, 'gelu', norm='instance', causal=causal, chunk_causal=False),
                         W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                         W.SequenceCNN(128, dropout=0.1, norm='layer', causal=causal, num_layers=2, kernel_size=7, num_dilations=6), num_classes)
        m2.load_state_dict(sd)
        m2.to(dev).eval()
        with torch.no_grad():
            got2 = m2(xd).cpu()
        assert not m2._engine.split_precision
        exact = figures(got2)
        del m2
    except Exception as e:   # (a causal bench model etc.: the default figures stand alone)
        exact = {'error': repr(e)[:200]}
    finally:
        if prev is None:
            os.environ.pop('W2S_EXACT_FP32', None)
        else:
            os.environ['W2S_EXACT_FP32'] = prev
    return {**res, 'exact_fp32': exact, 'min_top2_gap': float(gap.min()), 'median_top2_gap': float(gap.median()),
            'epochs_compared': int(pw.numel()), 'max_abs_logit': float(want.abs().max()),
            'weights': 'the bench model (reference default init, seed 42) after the timed steps', 'sample': f'1 recording x {epochs} epochs, inference forward'}


def build_trainer(W, signal_map, nc, causal, dev):
    model = W.Wav2Sleep(W.SignalEncoders(signal_map, 128, 'gelu', norm='instance', causal=causal, chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.1, norm='layer', causal=causal, num_layers=2, kernel_size=7, num_dilations=6), nc).to(dev).train()
    return model, W.FusedTrainStep(model)   # world > 1: broadcasts rank 0's parameters (FusedTrainStep.sync_parameters)


ELEMS_FWD = {'ABD': 62.2e6, 'THX': 62.2e6, 'ECG': 266.4e6, 'PPG': 266.4e6, 'EOG-L': 1083.3e6, 'EOG-R': 1083.3e6}   # SURVEY 8d, per recording


def host_batches(trainer, x, y, dev, warmup=2, steps=8):
    """The headline step with every batch starting in PINNED HOST memory, as a DataLoader(pin_memory=True) hands it to Lightning (which moves
    it with `.to(device, non_blocking=True)` before `on_after_batch_transfer`): the PCIe-inclusive rate -- reported under `extra`, never as
    `value` (the metric is quoted on inputs resident in HBM).  Two forms: the copy on the step's own stream (Lightning's default), and the
    next batch copied on a side stream while the current step runs (a prefetching loader)."""
    hx = [{k: v.cpu().pin_memory() for k, v in x.items()} for _ in range(2)]
    hy = [y.cpu().pin_memory() for _ in range(2)]
    nbytes = sum(v.numel() * 4 for v in hx[0].values()) + hy[0].numel() * 4
    up = lambda i: ({k: v.to(dev, non_blocking=True) for k, v in hx[i % 2].items()}, hy[i % 2].to(dev, non_blocking=True))
    for i in range(warmup):
        trainer.step(*up(i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.step(*up(i))
    torch.cuda.synchronize()
    same = (time.perf_counter() - t0) / steps
    side = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)

    def prefetch(i):
        with torch.cuda.stream(side):
            b = up(i)
        ev = torch.cuda.Event()
        ev.record(side)
        return b, ev
    nxt = prefetch(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        (bx, by), ev = nxt
        cur.wait_event(ev)
        for t in list(bx.values()) + [by]:
            t.record_stream(cur)
        nxt = prefetch(i + 1)
        trainer.step(bx, by)
    torch.cuda.synchronize()
    pre = (time.perf_counter() - t0) / steps
    B = y.shape[0]
    return {'workload': 'the headline train step with each batch copied from pinned host memory first (PCIe-inclusive; not the metric)',
            'host_bytes_per_batch': nbytes, 'ms_per_step_copy_on_step_stream': round(1000 * same, 3), 'recordings_per_s_copy_on_step_stream': round(B / same, 2),
            'ms_per_step_prefetched_on_side_stream': round(1000 * pre, 3), 'recordings_per_s_prefetched': round(B / pre, 2), 'steps': steps, 'warmup': warmup}


def ppgnet_leg(W, batch, dev, warmup=2, steps=6):
    """The reference's second trainable model, SleepPPGNet (models/ppgnet.py; 10-hour PPG inputs, BatchNorm + LeakyReLU), one full train step
    on the generic path (wav2sleep_amd/generic.py: walker + tape over HIP kernels, GenericTrainStep) -- reported under `extra`, never as
    `value`; a failure here is reported in place and does not touch the headline line."""
    try:
        from wav2sleep_amd.trainer import GenericTrainStep
        torch.manual_seed(42)
        model = W.SleepPPGNet().to(dev).train()
        step = GenericTrainStep(model, lr=1e-3, scheduler=False)
        g = torch.Generator(device=dev).manual_seed(99)
        x = torch.randn(batch, model.INPUT_LENGTH, device=dev, generator=g)
        y = torch.randint(0, 4, (batch, 1200), device=dev, generator=g).float()
        for _ in range(warmup):
            step.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step.step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = {'workload': f'SleepPPGNet, 10-hour PPG (1 228 800 samples), batch {batch}, full train step on the generic path', 'ms_per_step': round(1000 * dt, 3),
               'value': round(batch / dt, 3), 'unit': 'recordings/s', 'steps': steps, 'warmup': warmup, 'final_loss': round(float(out['loss']), 5),
               'peak_mem_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
        del model, step, x, y, out
        torch.cuda.empty_cache()
        return res
    except Exception as e:   # noqa: BLE001  (an extra leg must not cost the headline record)
        torch.cuda.empty_cache()
        return {'error': f'{type(e).__name__}: {e}'}


def extra_config(W, signal_map, spe, nc, causal, batch, epochs, dev, warmup=3, steps=8):
    """One more configuration of BASELINE.json, timed the same way as the headline (synthetic inputs resident in HBM, full train step),
    AFTER the headline's timed region and on a model of its own -- reported under `extra`, never as `value`."""
    SIGNAL_MAP_SAVE, SPE_SAVE = dict(SIGNAL_MAP), dict(SPE)
    SIGNAL_MAP.clear(); SIGNAL_MAP.update(signal_map); SPE.clear(); SPE.update(spe)
    try:
        torch.manual_seed(42)
        model, trainer = build_trainer(W, signal_map, nc, causal, dev)
        x, y = make_batch(batch, epochs, nc, dev, 4321)
        for _ in range(warmup):
            trainer.step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = trainer.step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        step_bytes = 3 * 4 * sum(ELEMS_FWD[s] for s in signal_map) * (epochs / 960) * batch
        res = {'workload': f'{"+".join(signal_map)} {epochs}-epoch synthetic, {nc}-class, batch {batch}, full train step' + (', causal convolutions' if causal else ''),
               'ms_per_step': round(1000 * dt, 3), 'value': round(batch / dt, 3), 'unit': 'recordings/s', 'steps': steps, 'warmup': warmup,
               'final_loss': round(float(out['loss']), 5), 'step_frac_of_hbm_peak': round(step_bytes / dt / 1e9 / HBM_PEAK_GBS, 4)}
        del model, trainer, x, y, out
        torch.cuda.empty_cache()
        return res
    finally:
        SIGNAL_MAP.clear(); SIGNAL_MAP.update(SIGNAL_MAP_SAVE); SPE.clear(); SPE.update(SPE_SAVE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)   # >= 20 timed steps after >= 5 warm-up
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--epochs', type=int, default=960)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the `extra` legs (BASELINE configs[3] and the causal variant at batch 16); --no-cpu skips them too')
    ap.add_argument('--variant', choices=['cardio', 'eog'], default='cardio',
                    help="'eog': BASELINE.json configs[3] (EOG-L+EOG-R @256 Hz, 5 classes); a parity-test configuration, not the headline line")
    ap.add_argument('--causal', action='store_true', help="the reference's `causal: True` variant (causal-padded convolutions); not the headline config")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes (nothing has touched the GPU in this
        # process yet, and a process that had must not exec) and leave with their exit code.
        import subprocess
        port = os.environ.get('MASTER_PORT', '29533')
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and os.environ.get('W2S_DIST_BACKEND', 'nccl') == 'nccl':
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or run `python bench.py --gpus N` directly)')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('W2S_FORCE_COLLECTIVES') == '1':
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        # W2S_DIST_BACKEND=gloo: control-flow test of the multi-rank path with several ranks sharing one GPU (RCCL refuses that)
        backend = os.environ.get('W2S_DIST_BACKEND', 'nccl')
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    import wav2sleep_amd as W
    from wav2sleep_amd import lib
    torch.manual_seed(42)  # scripts/config/main.yaml:35
    nc = 4
    if args.variant == 'eog':   # wav2sleep-eog (hub.py:17-22)
        SIGNAL_MAP.clear(); SIGNAL_MAP.update({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'})
        SPE.clear(); SPE.update({'EOG-L': 4096, 'EOG-R': 4096})
        nc = 5
    model, trainer = build_trainer(W, dict(SIGNAL_MAP), nc, args.causal, dev)
    x, y = make_batch(args.batch, args.epochs, nc, dev, 1234 + rank)

    for _ in range(args.warmup):
        trainer.step(x, y)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region: EXACTLY `steps` steps between two barrier + synchronize pairs (the driver's contract), measured twice over the same
    # steps -- the host clock around the synchronised region (`value` / `ms_per_step`: it contains everything, host enqueue included) and a HIP
    # event pair on the stream the steps are enqueued on (every step starts and ends on that stream: the encoder streams fork
    # from it and join it) -- the device's own time for the same steps, reported beside it as a cross-check.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = trainer.step(x, y)
    ev1.record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_ev = ev0.elapsed_time(ev1) * 1e-3
    if world > 1:
        t = torch.tensor([dt, dt_ev], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_ev = float(t[0]), float(t[1])
    loss = float(out['loss'])

    line = {'metric': 'overnight-recordings/sec (train step, bs=16)', 'value': round(args.batch * world * args.steps / dt, 3), 'unit': 'recordings/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000 * dt / args.steps, 3),
            'ms_per_step_hip_events': round(1000 * dt_ev / args.steps, 3), 'timing': 'host clock around barrier + synchronize (value, ms_per_step); HIP event pair on the launch stream over the same steps beside it',
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{len(SIGNAL_MAP)}-modality ({"+".join(SIGNAL_MAP)}) {args.epochs}-epoch ({args.epochs // 120} h) synthetic, {nc}-class, per-GPU batch {args.batch}, '
                                   f'full train step fwd+CE+bwd+clip+AdamW' + (', causal convolutions (causal: True)' if args.causal else ''), 'global_batch': args.batch * world, 'epochs': args.epochs,
                       'parallelism': f'dp{world}', 'final_loss': round(loss, 5),
                       'precision': "fp32 storage + fp32 accumulate; >=32-channel GEMMs as bf16x3 split products on the matrix cores "
                                    "(= the reference's float32_matmul_precision('high')); W2S_EXACT_FP32=1 for fp32 MFMA throughout" +
                                    ("; W2S_GRAD_FP16=1: inter-kernel gradient tensors of the <=32-channel encoder blocks stored as fp16 with "
                                     "per-tensor power-of-two scales" if os.environ.get('W2S_GRAD_FP16') == '1' else '')}}

    agg = None
    if not args.no_roofline:
        # two extra, untimed steps, the second with a HIP event pair around every GEMM-shaped launch on rank 0.  EVERY rank runs
        # them: a step contains the gradient all-reduce, so rank 0 stepping alone would leave its collectives unmatched.
        # (single stream for these steps: the timed steps overlap the four encoders on separate HIP streams, which inflates
        #  every kernel's own duration; isolated durations are what a roofline fraction is about.  profiles/ holds the
        #  rocprofv3 summaries of both: `bench.py` as is, and with W2S_MULTI_STREAM=0 which this leg agrees with.)
        ms_flag = trainer.eng.multi_stream
        trainer.eng.multi_stream = False
        trainer.step(x, y)  # settle allocator / packs in single-stream mode
        if rank == 0:
            lib.TIMER = lib.LaunchTimer()
        trainer.step(x, y)
        if rank == 0:
            agg = lib.TIMER.summary()
            lib.TIMER = None
        trainer.eng.multi_stream = ms_flag
        torch.cuda.synchronize()
    if rank == 0 and agg is not None:
        total_ms = sum(d['ms'] for d in agg.values())
        key, d = max(agg.items(), key=lambda kv: kv[1]['ms'])
        avg_s = d['ms'] / d['launches'] / 1e3
        b_per, f_per = d['bytes'] / d['launches'], d['flops'] / d['launches']
        ai = f_per / b_per
        # the matrix-core peak of the arithmetic the kernels actually run: bf16 MFMAs, three per product (split precision), or fp32 MFMAs
        mfma_peak = MFMA_SPLIT_PEAK_TF if trainer.eng.split_precision else MFMA_F32_PEAK_TF
        if ai > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
            ach = f_per / avg_s / 1e12
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(mfma_peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / mfma_peak, 4)}
        else:
            ach = b_per / avg_s / 1e9
            roof = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                    'frac_of_achievable': round(ach / HBM_ACHIEVABLE_GBS, 4), 'achievable': HBM_ACHIEVABLE_GBS}
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); only valid for the workload
        # they were collected on (the default one)
        pmc = {}
        if args.batch == 16 and args.epochs == 960 and args.variant == 'cardio' and not args.causal:
            for fn in sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))[-1:]:
                pmc = {pmc_key(k): v['hbm_bytes_per_launch'] for k, v in json.load(open(fn)).items()}
        traffic = int(pmc[key]) if key in pmc else None
        pmc_files = sorted(__import__('glob').glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
        roof.update({'traffic': traffic, 'traffic_source': (f'profiles/{os.path.basename(pmc_files[-1])} (builder-run rocprofv3 PMC passes, committed; not measured '
                                                            f'inside this run: rocprofv3 cannot nest in bench.py)') if traffic is not None and pmc_files else None,
                     'kernel': key, 'measured': 'HIP events, one extra single-stream step', 'launches_per_step': d['launches'], 'avg_us': round(avg_s * 1e6, 1),
                     'share_of_gemm_kernel_time': round(d['ms'] / total_ms, 3), 'algorithmic_bytes_per_launch': int(b_per),
                     'flops_per_launch': int(f_per)})
        # whole-step view against the SURVEY 8d convention (3x forward algorithmic bytes, fp32 storage)
        step_bytes = 3 * 4 * sum(ELEMS_FWD[s] for s in SIGNAL_MAP) * (args.epochs / 960) * args.batch
        roof['step_algorithmic_GBps'] = round(step_bytes / (dt / args.steps) / 1e9, 1)
        roof['step_frac_of_hbm_peak'] = round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
        roof['step_frac_of_achievable'] = round(step_bytes / (dt / args.steps) / 1e9 / HBM_ACHIEVABLE_GBS, 4)
        # every GEMM-shaped kernel family of the step, same measurement: time, launches, achieved algorithmic GB/s, and the ratio of
        # measured HBM traffic (committed PMC) to algorithmic bytes where a counter value exists for every kernel of the family
        fams = {}
        for k, v in agg.items():
            f = fams.setdefault(family_of(k), dict(ms=0.0, launches=0, bytes=0, traffic=0, covered=True))
            f['ms'] += v['ms']; f['launches'] += v['launches']; f['bytes'] += v['bytes']
            if k in pmc:
                f['traffic'] += pmc[k] * v['launches']
            else:
                f['covered'] = False
        roof['families'] = {n: {'ms': round(f['ms'], 3), 'launches': f['launches'], 'GBps': round(f['bytes'] / f['ms'] / 1e6, 1),
                                'frac': round(f['bytes'] / f['ms'] / 1e6 / HBM_PEAK_GBS, 4),
                                'frac_of_achievable': round(f['bytes'] / f['ms'] / 1e6 / HBM_ACHIEVABLE_GBS, 4),
                                'traffic_over_algorithmic': round(f['traffic'] / f['bytes'], 3) if f['covered'] and pmc else None}
                            for n, f in sorted(fams.items(), key=lambda kv: -kv[1]['ms'])}
        roof['gemm_kernel_ms_per_step'] = round(total_ms, 3)
        # the same facts as top-level scalars (a parser that drops nested objects keeps these): the family furthest below the HBM roof, the
        # largest family, and every family's time / fraction / traffic ratio under a flat key
        named = {n: f for n, f in roof['families'].items() if n != 'other'}
        if named:
            wn, wf = min(named.items(), key=lambda kv: kv[1]['frac'])
            ln, lf = max(named.items(), key=lambda kv: kv[1]['ms'])
            roof.update({'worst_family': wn, 'worst_family_ms': wf['ms'], 'worst_family_frac': wf['frac'], 'worst_family_traffic_ratio': wf['traffic_over_algorithmic'],
                         'largest_family': ln, 'largest_family_ms': lf['ms'], 'largest_family_frac': lf['frac'], 'largest_family_traffic_ratio': lf['traffic_over_algorithmic']})
            for n, f in named.items():
                k = FAMILY_KEYS.get(n, re.sub(r'[^a-z0-9]+', '_', n.lower()).strip('_'))
                roof[f'fam_{k}_ms'], roof[f'fam_{k}_frac'], roof[f'fam_{k}_traffic_ratio'] = f['ms'], f['frac'], f['traffic_over_algorithmic']
        # what the memory system gives a plain 1-read : 1-write stream on THIS box (torch's device copy of 1 GiB, HIP events): the
        # practical ceiling for these kernels -- the 8 TB/s `peak` is the spec number
        src = torch.empty(1 << 28, device=dev, dtype=torch.float32).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        # a note, not a headline: torch's own device copy on this box (a softer yardstick than the guide's 6.29 TB/s)
        roof['note_device_copy_GBps'] = round(copy_gbps, 1)
        roof['step_traffic_bytes'] = step_traffic(os.path.join(ROOT, 'profiles')) if pmc else None
        line['roofline'] = roof
        top = sorted(agg.items(), key=lambda kv: -kv[1]['ms'])
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_launch_breakdown.json'), 'w') as f:
            json.dump({k: v for k, v in top}, f, indent=1)
    if rank == 0 and world == 1 and not args.no_cpu:
        line['kappa_parity'] = kappa_parity(model, dict(SIGNAL_MAP), nc, args.epochs, dev, causal=args.causal)
    if rank == 0 and world == 1 and not args.no_extra and not args.no_cpu and args.variant == 'cardio' and not args.causal:   # (tuning runs pass --no-cpu)
        # BASELINE.json configs[3] (wav2sleep-eog: EOG-L + EOG-R at 4096 samples per epoch, ten-block encoders, 5 classes; hub.py:17-22,
        # settings.py:19-26) and the `causal: True` variant of the headline shape (scripts/config/main.yaml:22), each at batch 16 x 8 h on a
        # model of its own, after the headline's model has been released.  Driver-visible perf for the configs the suite only checks for parity
        host_leg = host_batches(trainer, x, y, dev)
        # inference forward of the headline model (api.predict -> model(x), api.py:179-183; eval mode, nothing saved), same batch
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                model(x)
            torch.cuda.synchronize()
            dti = (time.perf_counter() - t0) / 6
        fwd_bytes = 4 * sum(ELEMS_FWD[s] for s in SIGNAL_MAP) * (args.epochs / 960) * args.batch
        inference = {'workload': f'{"+".join(SIGNAL_MAP)} {args.epochs}-epoch synthetic, batch {args.batch}, inference forward (eval mode)',
                     'ms_per_batch': round(1000 * dti, 3), 'value': round(args.batch / dti, 3), 'unit': 'recordings/s', 'steps': 6, 'warmup': 2,
                     'frac_of_hbm_peak': round(fwd_bytes / dti / 1e9 / HBM_PEAK_GBS, 4)}
        del trainer, model, x, y, out
        torch.cuda.empty_cache()
        line['extra'] = {
            'sleep_ppgnet_b16': ppgnet_leg(W, args.batch, dev),
            'host_batches_b16': host_leg,
            'inference_b16': inference,
            'configs3_eog_b16': extra_config(W, {'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, {'EOG-L': 4096, 'EOG-R': 4096}, 5, False, args.batch, args.epochs, dev, 2, 6),
            'causal_b16': extra_config(W, dict(SIGNAL_MAP), dict(SPE), 4, True, args.batch, args.epochs, dev, 3, 8)}
    if rank == 0 and world == 1 and not args.no_cpu:
        line['cpu_baseline'] = cpu_baseline(args.epochs, nc, dict(SIGNAL_MAP))
    # what the collective layer saw: a SCALE record then proves RCCL ran with N ranks (dist_world_size is torch.distributed's own answer)
    line['dist_backend'] = dist.get_backend() if dist.is_initialized() else None
    line['dist_world_size'] = dist.get_world_size() if dist.is_initialized() else 1
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
