"""Round-2 golden vectors, produced by running the REAL reference (imported from /root/reference) in the build container.

    python tests/golden/make_goldens_r2.py [default_init] [optim] [train10] [dataset] [ema]

* default_init.npz -- the reference model of scripts/config/model/wav2sleep.yaml built under `torch.manual_seed(42)`
  (scripts/config/main.yaml:35): per-tensor checksums of its default initialisation, and its eval logits / arg-max on
  seeded inputs.  Default-init logits are nearly tied, so this is the arg-max case with the least margin.
* optim.npz -- torch.optim.AdamW + clip_grad_norm_(1.0) (+ the reference's ExpWarmUpScheduler) driven with a seeded gradient
  sequence at lr 1e-3: parameters after steps 1, 2, 5, 10.  The fused clip+AdamW kernel is compared on the same gradients.
* train10.npz -- ten full train steps (CE + backward + clip + AdamW, scheduler off, lr 1e-3) of the reference model.
* dataset.npz + dataset/*.parquet -- the reference `ParquetDataset` (data/dataset.py; `numba.njit` bound to the identity, its
  only use is a decorator in data/normalization.py) on committed parquet files: `_zscore_normalize`, `__getitem__` outputs.
* ema.npz -- the reference `EMACallback` (trainer/callbacks.py; `lightning` replaced by an empty base class: the callback only
  subclasses `lightning.pytorch.callbacks.Callback` and calls `state_dict()` / `load_state_dict()` on the module it is given).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_goldens as MG  # noqa: E402  (sets up the stub parent packages and imports the reference model classes)
from make_goldens import MultiModalAttentionEmbedder, SequenceCNN, SignalEncoders, Wav2Sleep, ExpWarmUpScheduler, summarize, REF  # noqa: E402
from oracle import wav2sleep_oracle as O  # noqa: E402

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


def production_model(signal_map=SM4, nc=4, dropout=0.1):
    """scripts/config/model/wav2sleep.yaml + inputs/cardiorespiratory/all.yaml."""
    enc = SignalEncoders(signal_map=dict(signal_map), feature_dim=128, activation='gelu', norm='instance', causal=False, chunk_causal=False,
                         initial_channels=16, max_channels=128, output_norm=False, use_residual=True)
    mix = MultiModalAttentionEmbedder(feature_dim=128, dropout=dropout, activation='gelu', layers=2, dim_ff=512, nhead=8)
    seq = SequenceCNN(feature_dim=128, dropout=dropout, activation='gelu', norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6)
    return Wav2Sleep(enc, mix, seq, num_classes=nc)


def run_default_init():
    torch.manual_seed(42)
    model = production_model()
    sd = model.state_dict()
    out = {'names': np.array(list(sd.keys())), 'abs_sums': np.array([float(v.double().abs().sum()) for v in sd.values()]),
           'sums': np.array([float(v.double().sum()) for v in sd.values()]),
           'first': np.array([float(v.flatten()[0]) for v in sd.values()])}
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    model.eval()
    for tag, (B, S, seed, missing) in {'a': (2, 16, 4242, None), 'b': (3, 8, 4243, {'ABD': [0], 'ECG': [1], 'PPG': [2]})}.items():
        x, y = O.make_inputs(cfg, B, S, seed=seed, missing=missing)
        with torch.no_grad():
            lg = model({k: v.clone() for k, v in x.items()})
        out[f'logits_{tag}'] = lg.numpy()
        out[f'pred_{tag}'] = lg.argmax(-1).numpy()
        srt = lg.sort(-1).values
        out[f'margin_{tag}'] = (srt[..., -1] - srt[..., -2]).numpy()
    np.savez_compressed(os.path.join(HERE, 'default_init.npz'), **out)
    print('default_init: |logit| max', float(np.abs(out['logits_a']).max()), 'min top-2 margin', float(out['margin_a'].min()), float(out['margin_b'].min()))


from tests.golden_util import OPT_SHAPES, grad_sequence as _grad_sequence  # noqa: E402


def run_optim():
    out = {}
    for variant in ('const', 'sched', 'wd'):
        g = torch.Generator().manual_seed(77)
        params = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.2) for s in OPT_SHAPES]
        out['init'] = np.concatenate([p.detach().flatten().numpy() for p in params])
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-2 if variant == 'wd' else 1e-4)   # optimizer/adamw.yaml ('wd': 100x the decay, so that the decoupled-decay term is far above fp32 round-off)
        sched = ExpWarmUpScheduler(opt, lr_max=1e-3, warmup_steps=4, tau=5.0) if variant == 'sched' else None
        for k, grads in enumerate(_grad_sequence(OPT_SHAPES, 10, 78), start=1):
            opt.zero_grad()
            for p, gr in zip(params, grads):
                p.grad = gr.clone()
            gn = torch.nn.utils.clip_grad_norm_(params, 1.0)   # training/main.yaml:21-22
            out[f'{variant}.lr{k}'] = np.float64(opt.param_groups[0]['lr'])
            out[f'{variant}.gnorm{k}'] = np.float64(float(gn))
            opt.step()
            if sched is not None:
                sched.step()
            if k in (1, 2, 5, 10):
                out[f'{variant}.param{k}'] = np.concatenate([p.detach().flatten().double().numpy() for p in params])
    np.savez_compressed(os.path.join(HERE, 'optim.npz'), **out)
    d = out['const.param10'] - out['init']
    print('optim: |dparam| after 10 steps', float(np.abs(d).max()), 'lrs', [out[f'sched.lr{k}'] for k in (1, 2, 4, 5, 10)])


def run_train10():
    """Ten reference train steps with the scheduler off at lr 1e-3 (weights move by ~1e-2: the optimiser is resolved)."""
    signal_map = {'ABD': 'ABD', 'ECG': 'ECG'}
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=4)
    sd = O.make_state_dict(cfg, seed=31)
    model = MG.build_reference(cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)
    out = {}
    for step in range(10):
        xs, ys = O.make_inputs(cfg, 2, 8, seed=3100 + step, missing={'ABD': [1]} if step % 2 else None)
        opt.zero_grad()
        loss = crit(model(xs).view(-1, 4), ys.view(-1).long())
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        out[f'loss{step}'] = np.float64(loss.item())
        out[f'gnorm{step}'] = np.float64(float(gn))
        if step in (0, 9):
            for k, p in model.state_dict().items():
                d = p.double() - sd[k].double()
                out[f'dparam{step}.{k}'] = d.numpy() if d.numel() <= 4096 else summarize(d)
                out[f'dnorm{step}.{k}'] = np.float64(float(d.norm()))
    np.savez_compressed(os.path.join(HERE, 'train10.npz'), **out)
    print('train10: losses', [round(float(out[f'loss{s}']), 4) for s in range(10)])


def _write_recording(path, epochs, cols, labels=True, seed=0, constant=None, stage_nan=False):
    import pandas as pd
    SPE = O.SAMPLES_PER_EPOCH
    rng = np.random.default_rng(seed)
    frames = []
    for c in cols:
        n = epochs * SPE[c]
        t = np.arange(n) * (30.0 / SPE[c])
        v = rng.standard_normal(n).astype(np.float32) * (2.5 if c == 'ECG' else 0.3) + (1.0 if c == 'THX' else -0.4)
        if constant == c:
            v[:] = 0.75
        frames.append(pd.Series(v, index=t, name=c))
    if labels:
        st = rng.integers(0, 5, epochs).astype(np.float64)
        st[1] = 7.0   # an out-of-map stage: `.map()` gives NaN -> -1
        frames.append(pd.Series(st, index=np.arange(epochs) * 30.0 + 1e-3, name='Stage'))
    df = pd.concat(frames, axis=1).sort_index()
    df.index.name = 'Timestamp'
    os.makedirs(os.path.dirname(path), exist_ok=True)
    df.to_parquet(path)


def run_dataset():
    shim = types.ModuleType('numba')
    shim.njit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault('numba', shim)
    data = types.ModuleType('wav2sleep.data'); data.__path__ = [REF + '/data']; sys.modules['wav2sleep.data'] = data
    from wav2sleep.data.dataset import ParquetDataset
    ddir = os.path.join(HERE, 'dataset')
    files = {'full': (6, ('ECG', 'PPG', 'THX', 'ABD'), True, 1, None), 'no_ppg': (5, ('ECG', 'THX'), True, 2, None),
             'flat_thx': (4, ('ECG', 'THX'), False, 3, 'THX')}
    for name, (epochs, cols, labels, seed, const) in files.items():
        _write_recording(os.path.join(ddir, name + '.parquet'), epochs, cols, labels=labels, seed=seed, constant=const)
    out = {}
    # _zscore_normalize on its own (dataset.py:76-87): finite, constant (std < eps), -inf and empty inputs
    g = torch.Generator().manual_seed(5)
    sig = {'a': torch.randn(5000, generator=g) * 3 + 2, 'const': torch.full((300,), 1.25), 'inf': torch.full((64,), float('-inf')),
           'empty': torch.zeros(0), 'tiny': torch.randn(257, generator=g) * 1e-7, 'one_nan': torch.tensor([1.0, float('nan'), 2.0])}
    z = ParquetDataset._zscore_normalize({k: v.clone() for k, v in sig.items()})
    for k in sig:
        out[f'zs.in.{k}'] = sig[k].numpy()
        out[f'zs.out.{k}'] = z[k].numpy()
    for name, (epochs, cols, labels, seed, const) in files.items():
        for nc in (4, 5):
            for mlh in (None, 0):
                ds = ParquetDataset([os.path.join(ddir, name + '.parquet')], columns=['ABD', 'THX', 'ECG', 'PPG'], num_classes=nc,
                                    require_labels=labels, max_length_hours=mlh)
                x, y = ds[0]
                tag = f'{name}.nc{nc}.mlh{mlh}'
                out[f'ds.{tag}.keys'] = np.array(list(x.keys()))
                for k, v in x.items():
                    out[f'ds.{tag}.x.{k}'] = v.numpy()
                out[f'ds.{tag}.y'] = y.numpy()
    np.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)
    print('dataset:', len(out), 'arrays;', {k: out[f'ds.{k}.nc4.mlhNone.y'].tolist() for k in files})



def run_save_predictions():
    """The reference `api.save_predictions` (api.py:193-221) on the committed parquet recordings + one with a DatetimeIndex: the CSV
    text it writes.  api.py imports hydra / omegaconf / pyedflib at module level (absent here, none of them touched by this function):
    empty stand-in modules; `numba.njit` bound to the identity as for the dataset goldens."""
    import tempfile
    import pandas as pd

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return type(name, (), {})
    for n in ('hydra', 'hydra.utils', 'omegaconf', 'pyedflib', 'mne'):
        sys.modules.setdefault(n, _Any(n))
    shim = types.ModuleType('numba'); shim.njit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault('numba', shim)
    data = types.ModuleType('wav2sleep.data'); data.__path__ = [REF + '/data']; sys.modules['wav2sleep.data'] = data
    models = types.ModuleType('wav2sleep.models'); models.__path__ = [REF + '/models']; sys.modules.setdefault('wav2sleep.models', models)
    from wav2sleep.data.dataset import ParquetDataset
    import wav2sleep.api as api
    ddir = os.path.join(HERE, 'dataset')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'in'); os.makedirs(os.path.join(src, 'sub'))
        import shutil
        shutil.copy(os.path.join(ddir, 'full.parquet'), os.path.join(src, 'a.parquet'))
        shutil.copy(os.path.join(ddir, 'no_ppg.parquet'), os.path.join(src, 'sub', 'b.parquet'))
        # a recording indexed by absolute time (api.py:214-215)
        df = pd.read_parquet(os.path.join(ddir, 'flat_thx.parquet'))
        df.index = pd.Timestamp('2021-03-04 22:15:00') + pd.to_timedelta(df.index.values, unit='s')
        df.index.name = 'Timestamp'
        df.to_parquet(os.path.join(src, 'sub', 'c_abs.parquet'))
        df.to_parquet(os.path.join(ddir, 'abs_time.parquet'))
        files = sorted([os.path.join(src, 'a.parquet'), os.path.join(src, 'sub', 'b.parquet'), os.path.join(src, 'sub', 'c_abs.parquet')])
        ds = ParquetDataset(files, columns=['ABD', 'THX', 'ECG', 'PPG'], num_classes=4, require_labels=False)
        g = torch.Generator().manual_seed(17)
        preds = [torch.randint(0, 4, (n,), generator=g) for n in (6, 5, 4)]
        labels = [torch.randint(-1, 4, (n,), generator=g).float() for n in (6, 5, 4)]
        for tag, lab in (('with_labels', labels), ('no_labels', None)):
            dst = os.path.join(tmp, 'out_' + tag)
            api.save_predictions(preds, src, dst, ds, labels=lab)
            for fp in files:
                rel = os.path.relpath(fp, src)
                csv = os.path.join(dst, os.path.splitext(rel)[0] + '.preds.csv')
                out[f'{tag}.{rel}'] = np.array(open(csv).read())
        out['files'] = np.array([os.path.relpath(f, src) for f in files])
        # api.load_dataset (api.py:141-159): which files, in which order, which columns
        lds = api.load_dataset(src, ['ECG', 'THX'], num_classes=5, max_length_hours=None)
        out['load_dataset.files'] = np.array([os.path.relpath(f, src) for f in lds.files])
        out['load_dataset.columns'] = np.array(list(lds.columns))
        x0, y0 = lds[0]
        out['load_dataset.keys0'] = np.array(list(x0.keys())); out['load_dataset.y0'] = y0.numpy()
        for i in range(3):
            out[f'pred{i}'] = preds[i].numpy(); out[f'label{i}'] = labels[i].numpy()
    np.savez_compressed(os.path.join(HERE, 'save_predictions.npz'), **out)
    print('save_predictions:', {k: len(str(v)) for k, v in out.items() if k.startswith('with')})


def run_cli():
    """The argument parser of the reference's scripts/predict.py (its module imports torchmetrics / wav2sleep.api: stand-ins; the parser is
    captured by intercepting ArgumentParser.parse_args): every option with its destination, default, type, action and required flag."""
    import argparse
    import importlib.util
    import json

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return type(name, (), {})
    for n in ('torchmetrics', 'torchmetrics.classification', 'hydra', 'hydra.utils', 'omegaconf', 'pyedflib', 'mne'):
        sys.modules.setdefault(n, _Any(n))
    shim = types.ModuleType('numba'); shim.njit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault('numba', shim)
    data = types.ModuleType('wav2sleep.data'); data.__path__ = [REF + '/data']; sys.modules['wav2sleep.data'] = data
    models = types.ModuleType('wav2sleep.models'); models.__path__ = [REF + '/models']; sys.modules.setdefault('wav2sleep.models', models)
    spec = importlib.util.spec_from_file_location('ref_predict_cli', os.path.join(os.path.dirname(os.path.dirname(REF)), 'scripts', 'predict.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    captured = {}
    real = argparse.ArgumentParser.parse_args

    def spy(self, *a, **k):
        captured['parser'] = self
        raise SystemExit(0)
    argparse.ArgumentParser.parse_args = spy
    try:
        mod.parse_args()
    except SystemExit:
        pass
    finally:
        argparse.ArgumentParser.parse_args = real
    opts = []
    for act in captured['parser']._actions:
        if not act.option_strings or act.dest == 'help':
            continue
        opts.append(dict(flags=list(act.option_strings), dest=act.dest, default=act.default, required=bool(act.required),
                         type=getattr(act.type, '__name__', None) if act.type else None, action=type(act).__name__, nargs=act.nargs))
    with open(os.path.join(HERE, 'predict_cli.json'), 'w') as f:
        json.dump(opts, f, indent=1)
    print('cli:', [o['flags'][0] for o in opts])

def run_ema():
    class _Any(types.ModuleType):   # any other name the module touches at import time (base classes of callbacks that are not used here)
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return type(name, (), {})
    lt = _Any('lightning'); ltp = _Any('lightning.pytorch'); ltc = _Any('lightning.pytorch.callbacks')
    ltpr = _Any('lightning.pytorch.callbacks.progress'); ltq = _Any('lightning.pytorch.callbacks.progress.tqdm_progress')
    ltc.Callback = type('Callback', (), {})
    lt.pytorch = ltp; ltp.callbacks = ltc; ltc.progress = ltpr; ltpr.tqdm_progress = ltq
    lt.Trainer = object; lt.LightningModule = object
    for n, m in {'lightning': lt, 'lightning.pytorch': ltp, 'lightning.pytorch.callbacks': ltc, 'lightning.pytorch.callbacks.progress': ltpr,
                 'lightning.pytorch.callbacks.progress.tqdm_progress': ltq}.items():
        sys.modules.setdefault(n, m)
    from wav2sleep.trainer.callbacks import EMACallback
    out = {}
    g = torch.Generator().manual_seed(9)
    mod = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    mod.device = torch.device('cpu')
    for name, (decay, start) in {'d999_s0': (0.999, 0), 'd9_s3': (0.9, 3), 'd0_s0': (0.0, 0), 'd1_s0': (1.0, 0)}.items():
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(10)))
        cb = EMACallback(decay=decay, start_step=start)
        cb.setup(None, mod, 'fit')
        out[f'{name}.init'] = torch.cat([p.detach().flatten() for p in mod.parameters()]).numpy()
        traj = []
        for step in range(6):
            with torch.no_grad():
                for p in mod.parameters():
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
            traj.append(torch.cat([p.detach().flatten() for p in mod.parameters()]).numpy())
            cb.on_train_batch_end(None, mod, None, None, step)
            out[f'{name}.ema{step}'] = torch.cat([v.flatten() for v in cb.state_dict()['ema_state_dict'].values()]).numpy()
        out[f'{name}.params'] = np.stack(traj)
        cb.on_validation_epoch_start(None, mod)
        out[f'{name}.during_val'] = torch.cat([p.detach().flatten() for p in mod.parameters()]).numpy()
        cb.on_validation_epoch_end(None, mod)
        out[f'{name}.after_val'] = torch.cat([p.detach().flatten() for p in mod.parameters()]).numpy()
        cb.on_train_end(None, mod)
        out[f'{name}.train_end'] = torch.cat([p.detach().flatten() for p in mod.parameters()]).numpy()
        out[f'{name}.step_count'] = np.int64(cb.state_dict()['step_count'])
    np.savez_compressed(os.path.join(HERE, 'ema.npz'), **out)
    print('ema:', len(out), 'arrays')


def run_variants():
    """Eval (and one train-mode BatchNorm) forwards of the reference's modules in configurations outside the production family, default
    initialisation under a seed + tests.golden_util.perturb_state; and SleepPPGNet on one 10-h input."""
    from tests.golden_util import VARIANTS, VARIANT_SEEDS, perturb_state, variant_index, variant_inputs
    from wav2sleep.models.ppgnet import SleepPPGNet
    out = {}
    for name in VARIANT_SEEDS:   # the round-2 set (round 6 added entries of its own: make_goldens_r6.py)
        v = VARIANTS[name]
        torch.manual_seed(4000 + variant_index(name))
        enc = SignalEncoders(**v['enc'])
        model = Wav2Sleep(enc, MultiModalAttentionEmbedder(**v['mix']), SequenceCNN(**v['seq']), num_classes=v['nc'])
        sd = perturb_state(model.state_dict(), seed=77)
        model.load_state_dict(sd, strict=True)
        out[f'{name}.keys'] = np.array(list(sd.keys()))
        out[f'{name}.checksum'] = np.array([float(t.double().abs().sum()) for t in sd.values()])
        model.train(bool(v.get('train')))
        x = variant_inputs(name)
        with torch.no_grad():
            lg = model({k: t.clone() for k, t in x.items()})
        out[f'{name}.logits'] = lg.numpy()
        if v.get('train'):
            after = model.state_dict()
            for k in after:
                if k.endswith('running_mean') or k.endswith('running_var'):
                    out[f'{name}.after.{k}'] = after[k].numpy()
        print('variant', name, tuple(lg.shape), 'max |logit|', float(lg.abs().max()))
    torch.manual_seed(4100)
    ppg = SleepPPGNet(n_classes=4, feature_dim=128, dropout=0.2, activation='leaky', norm='batch')
    sd = perturb_state(ppg.state_dict(), seed=78)
    ppg.load_state_dict(sd, strict=True)
    ppg.eval()
    out['ppgnet.keys'] = np.array(list(sd.keys()))
    out['ppgnet.checksum'] = np.array([float(t.double().abs().sum()) for t in sd.values()])
    x = torch.randn(1, 1228800, generator=torch.Generator().manual_seed(4101))
    with torch.no_grad():
        lg = ppg(x)
    out['ppgnet.logits'] = lg.numpy()
    print('ppgnet', tuple(lg.shape), 'max |logit|', float(lg.abs().max()))
    np.savez_compressed(os.path.join(HERE, 'variants.npz'), **out)


if __name__ == '__main__':
    only = sys.argv[1:]
    for name, fn in (('default_init', run_default_init), ('optim', run_optim), ('train10', run_train10), ('dataset', run_dataset), ('ema', run_ema), ('variants', run_variants), ('save_predictions', run_save_predictions), ('cli', run_cli)):
        if not only or name in only:
            fn()
