"""Round-6 golden vectors: GRADIENTS of the reference's modules in configurations outside the production family, produced by running the
REAL reference (imported from /root/reference) with torch autograd on the CPU of the build container.

    python tests/golden/make_goldens_r6.py

variants_grad.npz -- for each configuration of tests.golden_util.VARIANTS, in eval mode and (with BatchNorm) in train mode (every dropout at
0, BatchNorm on batch statistics): logits, the reference criterion's loss (CrossEntropyLoss(ignore_index=-1) over the flattened epochs,
trainer/main.py:76-77) on seeded labels, and d(loss)/d(parameter) for every parameter -- whole up to 2048 elements, an evenly strided
sample beyond, plus each gradient's L2 norm.  The gradients are those of the reference modules run in FLOAT64; the deviation of the same
step in float32 (the precision the reference trains in) is stored beside each as `ref32`.  And SleepPPGNet (models/ppgnet.py) in train mode, dropout 0, on two 10-hour inputs.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_goldens as MG  # noqa: E402,F401  (sets up the stub parent packages and imports the reference model classes)
from make_goldens import MultiModalAttentionEmbedder, SequenceCNN, SignalEncoders, Wav2Sleep  # noqa: E402
from tests.golden_util import VARIANTS, grad_sample_index, perturb_state, variant_cfg, variant_index, variant_inputs, variant_labels  # noqa: E402


def store_grads(out, tag, model, model32):
    """model: the float64 run (the stored gradients); model32: the same step in float32, the reference's own precision -- its deviation from
    the float64 gradient (relative L2, per parameter) is stored as the yardstick the tests size their tolerance with: BatchNorm on batch
    statistics makes some of these gradients ill-conditioned (differences of nearly equal sums), up to 6e-2 in float32."""
    p32 = dict(model32.named_parameters())
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        f = g.detach().flatten()
        out[f'{tag}.grad.{k}'] = f[torch.from_numpy(grad_sample_index(f.numel()))].numpy().astype(np.float32)
        out[f'{tag}.norm.{k}'] = np.float64(f.norm())
        g32 = p32[k].grad if p32[k].grad is not None else torch.zeros_like(p32[k])
        out[f'{tag}.ref32.{k}'] = np.float64((g32.double().flatten() - f).norm() / f.norm().clamp_min(1e-30))


def variant_run(name, train, dtype):
    v = variant_cfg(name, train)
    torch.manual_seed(4000 + variant_index(name))
    model = Wav2Sleep(SignalEncoders(**v['enc']), MultiModalAttentionEmbedder(**v['mix']), SequenceCNN(**v['seq']), num_classes=v['nc'])
    model.load_state_dict(perturb_state(model.state_dict(), seed=77), strict=True)
    model = model.to(dtype).train(train)
    x, y = variant_inputs(name), variant_labels(name)
    lg = model({k: t.clone().to(dtype) for k, t in x.items()})
    loss = F.cross_entropy(lg.flatten(0, 1), y.flatten().long(), ignore_index=-1)
    loss.backward()
    return model, lg.detach(), loss.detach()


def ppgnet_run(dtype):
    from wav2sleep.models.ppgnet import SleepPPGNet
    torch.manual_seed(4100)
    ppg = SleepPPGNet(n_classes=4, feature_dim=128, dropout=0.0, activation='leaky', norm='batch')
    ppg.load_state_dict(perturb_state(ppg.state_dict(), seed=78), strict=True)
    ppg = ppg.to(dtype).train()
    x = torch.randn(2, 1228800, generator=torch.Generator().manual_seed(4102)).to(dtype)
    y = torch.randint(0, 4, (2, 1200), generator=torch.Generator().manual_seed(4103))
    y[torch.rand(2, 1200, generator=torch.Generator().manual_seed(4104)) < 0.1] = -1
    lg = ppg(x)
    loss = F.cross_entropy(lg.flatten(0, 1), y.flatten().long(), ignore_index=-1)
    loss.backward()
    return ppg, lg.detach(), loss.detach(), y


def main():
    out = {}
    for name in VARIANTS:
        if name == 'causality_train':   # the same model as 'causality' (its train-mode run is below)
            continue
        for train in (False, True):
            if train and 'batch' not in (VARIANTS[name]['enc'].get('norm'), VARIANTS[name]['seq'].get('norm')):
                continue   # no BatchNorm and no dropout: the train-mode run is the eval-mode run
            model, lg, loss = variant_run(name, train, torch.float64)
            model32, lg32, _ = variant_run(name, train, torch.float32)
            tag = f"{name}.{'train' if train else 'eval'}"
            out[f'{tag}.logits'] = lg32.numpy()          # float32 logits (the forward is well-conditioned)
            out[f'{tag}.loss'] = np.float64(loss)
            store_grads(out, tag, model, model32)
            print(tag, 'loss', float(loss), 'worst float32 deviation', max(float(out[k]) for k in out if k.startswith(f'{tag}.ref32.')))
    ppg, lg, loss, y = ppgnet_run(torch.float64)
    ppg32, lg32, _, _ = ppgnet_run(torch.float32)
    out['ppgnet.train.logits'] = lg32.numpy()
    out['ppgnet.train.loss'] = np.float64(loss)
    out['ppgnet.train.labels'] = y.numpy()
    store_grads(out, 'ppgnet.train', ppg, ppg32)
    after = ppg32.state_dict()
    for k in ('conv_block.model.0.conv1.norm.running_mean', 'conv_block.model.7.conv3.norm.running_var', 'dilated_convs.1.conv_layers.5.norm.running_var'):
        out[f'ppgnet.train.after.{k}'] = after[k].numpy()
    print('ppgnet.train loss', float(loss), 'worst float32 deviation', max(float(out[k]) for k in out if k.startswith('ppgnet.train.ref32.')))
    np.savez_compressed(os.path.join(HERE, 'variants_grad.npz'), **out)
    print(len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'variants_grad.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
