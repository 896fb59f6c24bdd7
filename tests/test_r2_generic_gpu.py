"""GPU tests of the generic inference path (wav2sleep_amd/generic.py, csrc/generic.hip): the configurations of the reference's modules
outside the production family -- BatchNorm / GroupNorm / RMS / layer / no norm, ReLU / LeakyReLU / SiLU, feature_dim 16..64, any head
count, post-norm transformer layers, SleepPPGNet -- against logits the REFERENCE modules produced (tests/golden/variants.npz), and the
reference's own tests/model/test_causality.py mirrored at its own size.  fp32 matrix cores throughout: tolerance 2e-4 of the logit scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import wav2sleep_amd as W  # noqa: E402
from tests.golden_util import VARIANTS, load, variant_inputs  # noqa: E402
from tests.test_r2_pins_cpu import build_ppgnet, build_variant  # noqa: E402

DEV = 'cuda'


def close(got, want, tol=2e-4):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err, scale = np.abs(got - want).max(), np.abs(want).max()
    assert err <= tol * max(scale, 1e-3), (err, scale)
    return err / max(scale, 1e-30)


@pytest.mark.parametrize('name', ['causality', 'leaky_auto_rms', 'silu_group', 'relu_nonorm'])
def test_variant_forward_matches_reference(name):
    g = load('variants')
    model = build_variant(name).to(DEV).eval()
    x = variant_inputs(name)
    with torch.no_grad():
        lg = model({k: v.to(DEV) for k, v in x.items()})
    assert lg.shape == g[f'{name}.logits'].shape and not lg.requires_grad
    close(lg.cpu().numpy(), g[f'{name}.logits'])
    assert np.array_equal(lg.argmax(-1).cpu().numpy(), g[f'{name}.logits'].argmax(-1))
    # stand-alone sub-module calls compose to the same result (SignalEncoders -> MultiModalAttentionEmbedder -> SequenceCNN -> classifier)
    z = model.signal_encoders({k: v.to(DEV) for k, v in x.items()})
    m = model.epoch_mixer(z)
    s = model.sequence_mixer(m)
    lg2 = F.linear(s, model.classifier.weight, model.classifier.bias)
    close(lg2.detach().cpu().numpy(), g[f'{name}.logits'], tol=3e-4)


def test_batchnorm_train_mode_forward_and_running_statistics():
    """model.train(): BatchNorm normalises with the statistics of the batch and updates its running mean / (unbiased) variance."""
    g = load('variants')
    name = 'causality_train'
    model = build_variant(name).to(DEV).train()
    x = variant_inputs(name)
    lg = model({k: v.to(DEV) for k, v in x.items()}).detach()   # (grad mode: the differentiable forward; the tape is simply dropped)
    close(lg.cpu().numpy(), g[f'{name}.logits'], tol=5e-4)
    after = model.state_dict()
    checked = 0
    for k in after:
        if k.endswith('running_mean') or k.endswith('running_var'):
            np.testing.assert_allclose(after[k].cpu().numpy(), g[f'{name}.after.{k}'], rtol=2e-4, atol=2e-5, err_msg=k)
            checked += 1
    assert checked > 50 and int(after['signal_encoders.encoders.ECG.cnn.0.conv1.norm.num_batches_tracked']) == 1


def test_sleep_ppgnet_matches_reference():
    g = load('variants')
    model = build_ppgnet().to(DEV).eval()
    x = torch.randn(1, 1228800, generator=torch.Generator().manual_seed(4101))
    with torch.no_grad():
        lg = model(x.to(DEV))
    assert lg.shape == (1, 1200, 4)
    close(lg.cpu().numpy(), g['ppgnet.logits'], tol=5e-4)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 1024, device=DEV))


def test_reference_causality_test_mirrored():
    """tests/model/test_causality.py of the reference, unchanged in what it builds and asserts: SignalEncoders(feature_dim=16, 'relu',
    norm='batch', causal=True), MultiModalAttentionEmbedder(16), SequenceCNN(16, causal=True, norm='batch'), eval mode, 1 228 800 samples
    of ECG = PPG against the first half: the logits of the prefix do not depend on what follows."""
    causal, norm = True, 'batch'
    encoders = W.SignalEncoders(signal_map={'ECG': 'ECG', 'PPG': 'PPG'}, feature_dim=16, activation='relu', norm=norm, causal=causal)
    model = W.Wav2Sleep(signal_encoders=encoders, epoch_mixer=W.MultiModalAttentionEmbedder(feature_dim=16),
                        sequence_mixer=W.SequenceCNN(feature_dim=16, causal=causal, norm=norm), num_classes=4)
    model = model.to(DEV)
    model.eval()
    L = 1_228_800
    x = torch.randn(1, L, device=DEV)
    x2 = x[:, : L // 2]
    y = model({'ECG': x, 'PPG': x})
    y2 = model({'ECG': x2, 'PPG': x2})
    L_out = y2.shape[1]
    assert L_out == 600
    assert torch.allclose(y[:, :L_out], y2[:, :L_out])


def test_block_forwards_against_torch_cpu():
    """ConvLayer1D / ConvBlock1D / DilatedConvBlock / SignalEncoder are callable on their own (channels-first, like the reference's):
    a BatchNorm + ReLU layer and a LeakyReLU block against the same arithmetic written with torch CPU ops."""
    torch.manual_seed(3)
    layer = W.ConvLayer1D(32, 64, kernel_size=3, stride=2, padding=1, activation='relu', norm='batch')
    with torch.no_grad():
        layer.norm.running_mean.normal_(0, 0.3); layer.norm.running_var.uniform_(0.5, 1.5); layer.norm.weight.normal_(1, 0.2); layer.norm.bias.normal_(0, 0.2)
    x = torch.randn(2, 32, 301)
    want = F.relu(F.batch_norm(F.conv1d(x, layer.conv.weight, None, stride=2, padding=1), layer.norm.running_mean, layer.norm.running_var,
                               layer.norm.weight, layer.norm.bias, False, 0.1, layer.norm.eps))
    got = layer.to(DEV).eval()(x.to(DEV))
    close(got.cpu().numpy(), want.detach().numpy(), tol=1e-4)
    block = W.ConvBlock1D(16, 32, activation='leaky', norm='layer', causal=True)
    x = torch.randn(2, 16, 256)

    def cl_t(l, t):   # causal ConvLayer1D with ConvLayerNorm, on CPU
        k, st = l.conv.kernel_size[0], l.conv.stride[0]
        pad = k - 1
        o = F.conv1d(t, l.conv.weight, None, stride=st, padding=pad)
        trim = max(pad - (st - 1), 0)
        o = o[:, :, :-trim] if trim else o
        mu = o.mean(1, keepdim=True); var = (o - mu).pow(2).mean(1, keepdim=True)
        return F.leaky_relu(l.norm.weight * ((o - mu) / torch.sqrt(var + l.norm.eps)) + l.norm.bias)
    want = F.leaky_relu(cl_t(block.conv3, cl_t(block.conv2, cl_t(block.conv1, x))) + F.conv1d(x, block.downsample.weight, None, stride=2))
    got = block.to(DEV).eval()(x.to(DEV))
    close(got.cpu().numpy(), want.detach().numpy(), tol=1e-4)
    enc = W.SignalEncoder(feature_dim=32, activation='silu', samples_per_epoch=256, norm='rms').to(DEV).eval()
    z = enc(torch.randn(3, 5 * 256, device=DEV))
    assert z.shape == (3, 5, 32) and torch.isfinite(z).all()
    with pytest.raises(ValueError):
        enc(torch.randn(1, 300, device=DEV))
