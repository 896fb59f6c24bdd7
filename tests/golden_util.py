"""Helpers shared by the golden-vector tests (mirror of tests/golden/make_goldens.py:summarize)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CASES = {
    # name: (signal_map, num_classes, B, S, missing, weight seed, input seed) -- as in make_goldens.py
    'c1_ecg_only': ({'ECG': 'UNI'}, 4, 2, 16, None, 11, 101),
    'c2_four_mod': ({'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}, 4, 3, 8, {'ABD': [1], 'PPG': [2], 'ECG': [1]}, 12, 102),
    'c4_eog_pair': ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 2, 4, {'EOG-R': [0]}, 14, 104),
    'c5_shared_enc': ({'ABD': 'RESP', 'THX': 'RESP', 'ECG': 'ECG'}, 4, 2, 4, {'THX': [0]}, 15, 105),
    'c6_causal': ({'ABD': 'ABD', 'ECG': 'ECG'}, 4, 2, 8, {'ABD': [1]}, 16, 106),
    'c7_chunk_causal': ({'THX': 'THX', 'PPG': 'PPG'}, 4, 2, 6, {'THX': [0]}, 17, 107),
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
CAUSAL_CASES = {'c6_causal', 'c7_chunk_causal'}  # `causal: True` (scripts/config/main.yaml:22)
EXTRA = {'c8_embed_reg': dict(embed_signals=True, register_tokens=2, output_norm=True), 'c9_no_residual': dict(use_residual=False)}  # SignalEncoders(embed_signals=True), MultiModalAttentionEmbedder(register_tokens=2)
CHUNK_CASES = {'c7_chunk_causal'}  # chunk_causal=True (SignalEncoders' default) instead of the model yaml's `chunk_causal: False`


def case_config(name: str):
    """oracle ModelConfig of a golden case."""
    from oracle import wav2sleep_oracle as O
    signal_map, nc = CASES[name][:2]
    return O.ModelConfig(signal_map=signal_map, num_classes=nc, causal=name in CAUSAL_CASES, chunk_causal=name in CHUNK_CASES, **EXTRA.get(name, {}))


def summarize(t: torch.Tensor, k: int = 64) -> np.ndarray:
    f = t.detach().double().flatten().cpu()
    n = f.numel()
    idx = torch.linspace(0, n - 1, k).long()
    head = f[:k] if n >= k else torch.cat([f, f.new_zeros(k - n)])
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), f.norm()]), head, f[idx]]).numpy()


def checksum(d: dict) -> float:
    return float(sum(v.double().abs().sum() for v in d.values() if torch.isfinite(v).all()))


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'))


def assert_summary_close(got: torch.Tensor, want: np.ndarray, rtol: float, atol: float, what: str = ''):
    """Compare a tensor against either its full golden array or its golden summary vector."""
    if want.shape == tuple(got.shape):
        np.testing.assert_allclose(got.detach().cpu().double().numpy(), want, rtol=rtol, atol=atol, err_msg=what)
        return
    s = summarize(got)
    assert s.shape == want.shape, f'{what}: neither full nor summary shape ({got.shape} vs {want.shape})'
    # sums: scale atol by the abs-sum; pointwise entries: plain tolerances
    np.testing.assert_allclose(s[3:], want[3:], rtol=rtol, atol=atol, err_msg=what + ' [samples]')
    scale = max(want[1], 1e-30)
    assert abs(s[0] - want[0]) <= rtol * scale + atol, f'{what} [sum] {s[0]} vs {want[0]}'
    assert abs(s[1] - want[1]) <= rtol * scale + atol, f'{what} [abs-sum] {s[1]} vs {want[1]}'
    assert abs(s[2] - want[2]) <= rtol * max(want[2], 1e-30) + atol, f'{what} [l2] {s[2]} vs {want[2]}'


# ---- seeded gradient sequence of the optimiser golden (tests/golden/optim.npz): shared by the generator script and the GPU test ----
OPT_SHAPES = [(16, 1, 3), (32, 16, 3), (48, 96), (128,), (1, 1, 128, 1), (4, 128), (4,), (1, 37, 1)]


def grad_sequence(shapes, steps, seed):
    g = torch.Generator().manual_seed(seed)
    seq = []
    for k in range(steps):
        scale = [3.0, 0.02, 1.0, 0.3, 5.0, 0.05, 0.7, 1.5, 0.01, 2.0][k % 10]   # global norms above and below the clip threshold
        seq.append([scale * torch.randn(s, generator=g) / (float(np.prod(s)) ** 0.5) for s in shapes])
    return seq


# ---- generic-path variants (tests/golden/variants.npz): configurations of the reference's modules outside the production family ----
VARIANTS = {
    # tests/model/test_causality.py of the reference: feature_dim 16, ReLU, BatchNorm, causal (chunked) encoders, mixer / SequenceCNN defaults
    'causality': dict(enc=dict(signal_map={'ECG': 'ECG', 'PPG': 'PPG'}, feature_dim=16, activation='relu', norm='batch', causal=True),
                      mix=dict(feature_dim=16), seq=dict(feature_dim=16, causal=True, norm='batch'), nc=4, B=2, S=6, missing={'PPG': [1]}),
    'causality_train': dict(enc=dict(signal_map={'ECG': 'ECG', 'PPG': 'PPG'}, feature_dim=16, activation='relu', norm='batch', causal=True),
                            mix=dict(feature_dim=16), seq=dict(feature_dim=16, causal=True, norm='batch', dropout=0.0), nc=4, B=2, S=6, missing=None, train=True),
    'leaky_auto_rms': dict(enc=dict(signal_map={'ABD': 'ABD', 'ECG': 'ECG'}, feature_dim=32, activation='leaky', norm='auto', max_channels=64),
                           mix=dict(feature_dim=32, layers=1, nhead=2, dim_ff=128, activation='relu', norm_first=False, register_tokens=1),
                           seq=dict(feature_dim=32, norm='rms', activation='silu', num_layers=1, num_dilations=3), nc=5, B=2, S=5, missing={'ABD': [0]}),
    'silu_group': dict(enc=dict(signal_map={'THX': 'THX'}, feature_dim=64, activation='silu', norm='group', causal=True, chunk_causal=False, use_residual=False,
                                output_norm=True),
                       mix=dict(feature_dim=64, layers=2, nhead=4, dim_ff=256, activation='gelu'),
                       seq=dict(feature_dim=64, norm='group', activation='leaky', causal=True), nc=4, B=3, S=4, missing=None),
    'relu_nonorm': dict(enc=dict(signal_map={'ABD': 'RESP', 'THX': 'RESP'}, feature_dim=16, activation='relu', norm=None, embed_signals=True),
                        mix=dict(feature_dim=16, layers=1, nhead=1, dim_ff=64), seq=dict(feature_dim=16, norm='layer', activation='gelu', num_layers=1), nc=4, B=2, S=4,
                        missing=None),
}


def perturb_state(sd: dict, seed: int) -> dict:
    """Deterministic non-trivial values for everything default initialisation leaves at 0 / 1 (norm affine parameters, BatchNorm running
    statistics, biases of the norm-free convolutions) so that the variants' goldens exercise them; applied to the reference's and to the
    build's state dict alike (same keys, same order)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith('running_mean'):
            out[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith('running_var'):
            out[k] = 0.6 + torch.rand(v.shape, generator=g)
        elif k.endswith('num_batches_tracked'):
            out[k] = v.clone()
        elif ('norm' in k and k.endswith('.weight')) or k.endswith('output_norm.weight'):
            out[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif 'norm' in k and k.endswith('.bias'):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = v.clone()
    return out


def variant_inputs(name: str):
    """Seeded inputs of a variant: dict signal -> [B, S * spe] with the listed samples' rows set to -inf."""
    v = VARIANTS[name]
    spe = {'ABD': 256, 'THX': 256, 'ECG': 1024, 'PPG': 1024, 'EOG-L': 4096, 'EOG-R': 4096}
    g = torch.Generator().manual_seed(900 + variant_index(name))
    x = {s: torch.randn(v['B'], v['S'] * spe[s], generator=g) for s in v['enc']['signal_map']}
    for s, rows in (v.get('missing') or {}).items():
        x[s][rows] = float('-inf')
    return x


# ---- gradients of the generic-path variants (tests/golden/variants_grad.npz, made by make_goldens_r6.py from the REFERENCE modules) ----
GRAD_SAMPLE = 2048   # gradient tensors above this size are stored as an evenly strided sample of this many elements (+ their L2 norm)


# round-6 additions (seeded on their own so that the round-2 entries keep theirs): per-epoch (chunk-causal) encoders with register tokens and
# three classes; BatchNorm in train mode with an odd number of epochs and samples, a shared encoder and signal embeddings
VARIANTS_R6 = {
    'chunk_regs': dict(enc=dict(signal_map={'ECG': 'ECG', 'ABD': 'ABD'}, feature_dim=32, activation='gelu', norm='instance', causal=True, chunk_causal=True),
                       mix=dict(feature_dim=32, layers=1, nhead=4, dim_ff=64, register_tokens=2),
                       seq=dict(feature_dim=32, norm='layer', causal=True, num_layers=1, num_dilations=3, dropout=0.0), nc=3, B=3, S=5, missing={'ABD': [2]}, seed=4300),
    'batch_shared_odd': dict(enc=dict(signal_map={'ABD': 'RESP', 'THX': 'RESP', 'PPG': 'PPG'}, feature_dim=16, activation='leaky', norm='batch', embed_signals=True),
                             mix=dict(feature_dim=16, layers=2, nhead=2, dim_ff=32, activation='silu'),
                             seq=dict(feature_dim=16, norm='batch', activation='relu', num_layers=2, num_dilations=2, dropout=0.0), nc=5, B=3, S=7, missing={'THX': [0]},
                             seed=4301, train=True),
}
VARIANTS.update(VARIANTS_R6)   # (looked up by name everywhere; the seeds of the round-2 entries come from VARIANT_SEEDS below, not from this dict's order)
VARIANT_SEEDS = {'causality': 0, 'causality_train': 1, 'leaky_auto_rms': 2, 'relu_nonorm': 3, 'silu_group': 4}   # = sorted() index of the round-2 set


def variant_index(name: str) -> int:
    return VARIANT_SEEDS[name] if name in VARIANT_SEEDS else VARIANTS[name]['seed'] - 4000


def variant_labels(name: str):
    """Seeded stage labels [B, S] of a variant with ~15 % unscored (-1) epochs."""
    v = VARIANTS[name]
    g = torch.Generator().manual_seed(1900 + variant_index(name))
    y = torch.randint(0, v['nc'], (v['B'], v['S']), generator=g)
    y[torch.rand(v['B'], v['S'], generator=g) < 0.15] = -1
    return y


def variant_cfg(name: str, train: bool):
    """The variant's constructor arguments for a gradient run: train=True -> train mode with every dropout at 0 (BatchNorm uses the batch's
    statistics; nothing random), train=False -> eval mode (BatchNorm uses its running statistics)."""
    v = dict(VARIANTS[name])
    if train:
        v['seq'] = dict(v['seq'], dropout=0.0)
    return v


def grad_sample_index(n: int):
    """indices of the stored sample of an n-element gradient (all of it up to GRAD_SAMPLE elements)"""
    if n <= GRAD_SAMPLE:
        return np.arange(n)
    return np.linspace(0, n - 1, GRAD_SAMPLE).astype(np.int64)
