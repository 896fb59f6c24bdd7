"""Generate golden vectors by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box).  Usage:

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

The reference's hot-path modules are imported through a stub parent package (SURVEY.md App. B:
`wav2sleep/__init__.py` pulls hydra/lightning, which are not installed).  Weights and inputs are
NOT the reference's: they come from `oracle.wav2sleep_oracle.make_state_dict / make_inputs`
(deterministic CPU generators) and are loaded into the reference modules with
`load_state_dict(strict=True)`, so the fixtures only need to hold the reference's OUTPUTS plus a
checksum of the generated weights/inputs.  Large tensors are stored as (first-k, strided-k, sum,
abs-sum, l2) summaries; small ones in full.
"""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = '/root/reference/src/wav2sleep'
pkg = types.ModuleType('wav2sleep'); pkg.__path__ = [REF]; sys.modules['wav2sleep'] = pkg
tr = types.ModuleType('wav2sleep.trainer'); tr.__path__ = [REF + '/trainer']; sys.modules['wav2sleep.trainer'] = tr
from wav2sleep.models.wav2sleep import MultiModalAttentionEmbedder, SequenceCNN, SignalEncoders, Wav2Sleep  # noqa: E402
from wav2sleep.stats import cohens_kappa, confusion_accuracy  # noqa: E402
from wav2sleep.trainer.masker import SignalMasker  # noqa: E402
from wav2sleep.trainer.scheduler import ExpWarmUpScheduler  # noqa: E402

from oracle import wav2sleep_oracle as O  # noqa: E402

torch.backends.mha.set_fastpath_enabled(False)  # scripts/train.py:24
torch.manual_seed(0)


def summarize(t: torch.Tensor, k: int = 64) -> np.ndarray:
    """[sum, abs-sum, l2, first-k..., strided-k...] in float64 -- mirrored by tests/golden_util.py."""
    f = t.detach().double().flatten()
    n = f.numel()
    idx = torch.linspace(0, n - 1, k).long()
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), f.norm()]), f[:k] if n >= k else torch.cat([f, f.new_zeros(k - n)]), f[idx]]).numpy()


def checksum(d: dict) -> float:
    return float(sum(v.double().abs().sum() for v in d.values() if torch.isfinite(v).all()))


def build_reference(cfg: O.ModelConfig) -> Wav2Sleep:
    enc = SignalEncoders(signal_map=dict(cfg.signal_map), feature_dim=cfg.feature_dim, activation='gelu', norm='instance',
                         causal=cfg.causal, chunk_causal=cfg.chunk_causal, embed_signals=cfg.embed_signals, initial_channels=cfg.initial_channels,
                         max_channels=cfg.max_channels, output_norm=cfg.output_norm, use_residual=cfg.use_residual)
    mix = MultiModalAttentionEmbedder(feature_dim=cfg.feature_dim, dropout=0.0, activation='gelu', layers=cfg.mixer_layers,
                                      dim_ff=cfg.mixer_dim_ff, nhead=cfg.mixer_nhead, register_tokens=cfg.register_tokens)
    seq = SequenceCNN(feature_dim=cfg.feature_dim, dropout=0.0, activation='gelu', norm='layer', causal=cfg.causal,
                      num_layers=cfg.seq_blocks, kernel_size=cfg.seq_kernel, num_dilations=cfg.seq_dilations)
    return Wav2Sleep(enc, mix, seq, num_classes=cfg.num_classes)


CASES = {
    # name: (signal_map, num_classes, B, S, missing, weight seed, input seed)
    'c1_ecg_only': ({'ECG': 'UNI'}, 4, 2, 16, None, 11, 101),
    'c2_four_mod': ({'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}, 4, 3, 8, {'ABD': [1], 'PPG': [2], 'ECG': [1]}, 12, 102),
    'c4_eog_pair': ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 2, 4, {'EOG-R': [0]}, 14, 104),
    'c5_shared_enc': ({'ABD': 'RESP', 'THX': 'RESP', 'ECG': 'ECG'}, 4, 2, 4, {'THX': [0]}, 15, 105),
    # `causal: True` of scripts/config/main.yaml:22 with the model yaml's `chunk_causal: False`: causal-padded convolutions
    'c6_causal': ({'ABD': 'ABD', 'ECG': 'ECG'}, 4, 2, 8, {'ABD': [1]}, 16, 106),
    # SignalEncoders' own default for causal models: chunk_causal=True (per-epoch encoding, wav2sleep.py:248-255)
    'c7_chunk_causal': ({'THX': 'THX', 'PPG': 'PPG'}, 4, 2, 6, {'THX': [0]}, 17, 107),
    # shared encoder told apart by signal embeddings (embed_signals=True) + two register tokens next to CLS
    'c8_embed_reg': ({'ABD': 'RESP', 'THX': 'RESP', 'ECG': 'ECG'}, 4, 2, 4, {'THX': [1]}, 18, 108),
    'c9_no_residual': ({'ABD': 'ABD', 'PPG': 'PPG'}, 4, 2, 4, None, 19, 109),
    # BASELINE configs[4] as written: a map over {ABD, THX, ECG, PPG, EOG} (settings.py:19-26; mixed 6/8/10-block encoders, D = 6 tokens),
    # ragged: sample 1 keeps only its backup channel (masker.py:30-48), sample 2 lacks the EOG
    'c10_five_mod': ({'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG', 'EOG-L': 'EOG-L'}, 4, 3, 3,
                     {'ABD': [1], 'THX': [1], 'PPG': [1], 'EOG-L': [1, 2], 'ECG': [0]}, 20, 110),
    # ... and with EOG-R: D = 7 tokens, the attention kernels' limit; 5 classes
    'c11_six_mod': ({'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG', 'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 2, 2,
                    {'THX': [0], 'EOG-R': [1], 'ECG': [1]}, 21, 111),
}
CAUSAL_CASES = {'c6_causal', 'c7_chunk_causal'}
CHUNK_CASES = {'c7_chunk_causal'}
EXTRA = {'c8_embed_reg': dict(embed_signals=True, register_tokens=2, output_norm=True), 'c9_no_residual': dict(use_residual=False)}


def run_case(name: str):
    signal_map, nc, B, S, missing, wseed, iseed = CASES[name]
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=nc, causal=name in CAUSAL_CASES, chunk_causal=name in CHUNK_CASES, **EXTRA.get(name, {}))
    sd = O.make_state_dict(cfg, seed=wseed)
    x, y = O.make_inputs(cfg, B, S, seed=iseed, missing=missing)
    model = build_reference(cfg)
    model.load_state_dict(sd, strict=True)
    out = {'weights_checksum': np.float64(checksum(sd)), 'inputs_checksum': np.float64(checksum(x) + float(y.sum()))}

    # ---- eval forward with stage taps (hooks on the reference modules) ----
    model.eval()
    taps = {}
    hooks = []
    for enc_name, enc in model.signal_encoders.encoders.items():
        for i, blk in enumerate(enc.cnn):
            key = f'signal_encoders.encoders.{enc_name}.cnn.{i}.out'
            hooks.append(blk.register_forward_hook(lambda m, a, o, key=key: taps.setdefault(key, []).append(o.detach())))
    hooks.append(model.epoch_mixer.register_forward_hook(lambda m, a, o: taps.__setitem__('mixer', o.detach())))
    hooks.append(model.sequence_mixer.register_forward_hook(lambda m, a, o: taps.__setitem__('seq', o.detach())))
    hooks.append(model.signal_encoders.register_forward_hook(lambda m, a, o: taps.__setitem__('z', {k: v.detach() for k, v in o.items()})))
    with torch.no_grad():
        logits = model({k: v.clone() for k, v in x.items()})
    z = taps.pop('z')
    for h in hooks:
        h.remove()
    out['logits'] = logits.numpy()
    out['pred'] = model.predict({k: v.clone() for k, v in x.items()}).numpy()
    out['mixer'] = taps['mixer'].numpy()
    out['seq'] = taps['seq'].numpy()
    for s, v in z.items():
        out[f'z.{s}'] = v.numpy()
    for k, v in taps.items():
        if isinstance(v, list):  # block outputs; shared encoders are called once per signal (dict order)
            for j in range(len(v)):
                out[f'tap.{k}.{j}'] = summarize(v[j])

    # ---- subset equivalence (masked sample == that sample with the subset only) is an oracle test ----

    # ---- train steps: CE(ignore -1) + backward + clip 1.0 + AdamW(1e-3, 1e-4) + ExpWarmUp ----
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    sched = ExpWarmUpScheduler(opt, lr_max=1e-3, warmup_steps=2000, tau=10000)
    crit = torch.nn.CrossEntropyLoss(reduction='mean', label_smoothing=0.0, ignore_index=-1)
    for step in range(2):
        xs, ys = O.make_inputs(cfg, B, S, seed=iseed + 1000 * step, missing=missing)
        opt.zero_grad()
        lg = model(xs)
        loss = crit(lg.view(-1, nc), ys.view(-1).long())
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[f'grad0.{k}'] = g.numpy().copy() if g.numel() <= 4096 else summarize(g)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        out[f'lr{step}'] = np.float64(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()
        out[f'loss{step}'] = np.float64(loss.item())
        out[f'gnorm{step}'] = np.float64(float(gn))
    for k, p in model.state_dict().items():
        out[f'param2.{k}'] = p.numpy() if p.numel() <= 4096 else summarize(p)
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(name, 'logits', tuple(logits.shape), 'loss', [out['loss0'], out['loss1']], 'gnorm', [out['gnorm0'], out['gnorm1']])


def run_misc():
    out = {}
    # scheduler table (scheduler.py:23-32); lr used by optimiser step k
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sched = ExpWarmUpScheduler(opt, lr_max=1e-3, warmup_steps=2000, tau=10000)
    lrs = {}
    for k in range(1, 12002):
        if k in (1, 2, 3, 4, 5, 1999, 2000, 2001, 2002, 12000, 12001):
            lrs[k] = opt.param_groups[0]['lr']
        opt.step(); sched.step()
    out['lr_steps'] = np.array(list(lrs.keys()), dtype=np.int64)
    out['lr_values'] = np.array(list(lrs.values()), dtype=np.float64)
    # kappa / accuracy (stats.py)
    cm = np.array([[50, 3, 0, 2], [4, 80, 5, 6], [0, 6, 20, 0], [1, 5, 0, 30]])
    out['cm'] = cm
    out['kappa'] = np.float64(cohens_kappa(cm, 4))
    out['acc'] = np.float64(confusion_accuracy(cm))
    cm5 = np.array([[10, 1, 0, 0, 2], [3, 22, 4, 0, 1], [0, 5, 40, 2, 0], [0, 0, 3, 9, 0], [1, 2, 0, 0, 17]])
    out['cm5'] = cm5
    out['kappa5'] = np.float64(cohens_kappa(cm5, 5))
    # masker: record the masks the reference produces for a fixed torch seed, plus its inputs
    torch.manual_seed(7)
    B = 64
    x = {s: torch.randn(B, 8) for s in ('ABD', 'THX', 'ECG', 'PPG')}
    x['ECG'][:8] = float('-inf'); x['PPG'][8:16] = float('-inf'); x['ABD'][16:24] = float('-inf')
    avail = np.stack([~torch.isinf(v[:, 0]).numpy() for v in x.values()], -1)
    masker = SignalMasker({'ABD': 0.7, 'THX': 0.7, 'ECG': 0.5, 'PPG': 0.1}, backups=['ECG', 'PPG'])
    keep = np.zeros((50, B, 4), dtype=bool)
    for r in range(50):
        xm = masker({k: v.clone() for k, v in x.items()})
        keep[r] = np.stack([~torch.isinf(v[:, 0]).numpy() for v in xm.values()], -1)
    out['masker_avail'] = avail
    out['masker_keep'] = keep
    np.savez_compressed(os.path.join(HERE, 'misc.npz'), **out)
    print('misc: kappa', out['kappa'], 'acc', out['acc'], 'lrs', out['lr_values'][:5])


def run_causal_norm():
    """Golden vectors of `causal_rolling_normalize` from the reference module itself.  numba is not installed here; its only use in
    data/normalization.py is the `@njit(cache=True)` decorator on the loop, so the module is imported with `numba.njit` bound to an
    identity decorator and the reference's own loop body runs as plain Python (same arithmetic, fp64)."""
    shim = types.ModuleType('numba')
    shim.njit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault('numba', shim)
    data = types.ModuleType('wav2sleep.data'); data.__path__ = [REF + '/data']; sys.modules['wav2sleep.data'] = data
    from wav2sleep.data.normalization import causal_rolling_normalize as ref_norm
    rng = np.random.default_rng(5)
    out = {}
    cases = {
        # name: (samples per epoch, epochs, dtype, kwargs)
        'ecg_drift': (1024, 24, np.float32, dict(tau_seconds=900.0, baseline_tau_seconds=120.0, min_sigma=0.1, outlier_threshold_sigma=4.0)),
        'abd_spikes': (256, 40, np.float32, dict(tau_seconds=900.0, baseline_tau_seconds=120.0, min_sigma=0.1, outlier_threshold_sigma=4.0)),
        'eog_default': (4096, 6, np.float64, dict()),
        'flat_then_active': (256, 30, np.float32, dict(tau_seconds=300.0, min_sigma=0.1)),
    }
    for name, (spe, epochs, dtype, kw) in cases.items():
        n = spe * epochs
        t = np.arange(n) / (spe / 30.0)
        x = rng.standard_normal(n) * (1.0 + 0.5 * np.sin(2 * np.pi * t / 300.0)) + 0.002 * t
        if name == 'abd_spikes':
            x[rng.integers(0, n, 40)] += rng.standard_normal(40) * 25.0
        if name == 'flat_then_active':
            x[: n // 3] = 0.25
        x = x.astype(dtype)
        y, m = ref_norm(x.copy(), sampling_freq=spe / 30.0, return_outlier_mask=True, **kw)
        out[f'{name}.x'] = x
        out[f'{name}.y'] = np.asarray(y)
        out[f'{name}.mask'] = np.asarray(m)
        out[f'{name}.spe'] = np.int64(spe)
        for k, v in kw.items():
            out[f'{name}.kw.{k}'] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, 'causal_norm.npz'), **out)
    print('causal_norm:', {k: int(out[k + '.mask'].sum()) for k in cases})


if __name__ == '__main__':
    only = sys.argv[1:]
    for name in CASES:
        if not only or name in only:
            run_case(name)
    if not only or 'misc' in only:
        run_misc()
    if not only or 'causal_norm' in only:
        run_causal_norm()
