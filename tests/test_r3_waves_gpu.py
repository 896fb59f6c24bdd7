"""The pipelined train step (FusedTrainStep(waves=n) / W2S_WAVES, engine.train_waves): one batch as n sample waves, the trunk of one wave
beside the encoders of its neighbours.  It must be the same optimiser step: the reference goldens with their tolerances, and against the
one-pass step of this library -- logits and confusion counts bit for bit (every sample is computed independently of its batch), the loss to
fp32 rounding, every gradient to the rounding of a two-term fp32 sum (the waves' partial gradients are added in a fixed order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import wav2sleep_amd as W  # noqa: E402
from oracle import wav2sleep_oracle as O  # noqa: E402  (checker only)
from tests.golden_util import CASES, assert_summary_close, case_config, load  # noqa: E402
from tests.test_parity_gpu import build, to_dev  # noqa: E402

DEV = 'cuda'
SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}


@pytest.mark.parametrize('name', [n for n in CASES if CASES[n][2] >= 2])
def test_train_steps_in_two_waves_match_reference_goldens(name):
    """test_train_steps_match_reference_goldens with the batch split into two sample waves: same goldens, same tolerances."""
    signal_map, nc, B, S, missing, wseed, iseed = CASES[name]
    g = load(name)
    cfg = case_config(name)
    model = build(signal_map, nc, causal=cfg.causal, chunk_causal=cfg.chunk_causal, embed_signals=cfg.embed_signals, register_tokens=cfg.register_tokens,
                  output_norm=cfg.output_norm, use_residual=cfg.use_residual)
    model.load_state_dict(O.make_state_dict(cfg, seed=wseed))
    model.to(DEV).train()
    tr = W.FusedTrainStep(model, waves=2)
    for step in range(2):
        xs, ys = O.make_inputs(cfg, B, S, seed=iseed + 1000 * step, missing=missing)
        out = tr.step(to_dev(xs), ys.to(DEV))
        assert float(out['loss']) == pytest.approx(float(g[f'loss{step}']), rel=1e-4)
        assert float(out['grad_norm']) == pytest.approx(float(g[f'gnorm{step}']), rel=1e-3)
        if step == 0:
            for k, p in model._engine.G.items():
                want = g[f'grad0.{k}'] if f'grad0.{k}' in g else None
                if want is None or want.shape != tuple(p.shape):
                    continue
                got = p.detach().cpu().double().numpy()
                rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
                assert rel <= 2e-3, (k, rel)
    sd = model.state_dict()
    for k in sd:
        assert_summary_close(sd[k], g[f'param2.{k}'], rtol=1e-5, atol=1.1e-6, what=f'param2.{k}')


def _one_step(cfg, sd, x, y, multi_stream=True, **kw):
    model = build(cfg.signal_map, cfg.num_classes, causal=cfg.causal, chunk_causal=cfg.chunk_causal)
    model.load_state_dict(sd)
    model.to(DEV).train()
    model._ensure_flat()
    model._engine.multi_stream = multi_stream
    tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False, **kw)
    outs = []
    k = kw.get('accumulate', 1)
    B = y.shape[0]
    for i in range(k):
        sl = slice(i * B // k, (i + 1) * B // k)
        out = tr.step({s: v[sl] for s, v in x.items()}, y[sl])
        outs.append({n: (v.clone() if torch.is_tensor(v) else v) for n, v in out.items()})
    torch.cuda.synchronize()
    return model, outs


@pytest.mark.parametrize('signal_map,nc,B,S,missing,waves,causal', [
    (SM4, 4, 4, 6, {'ABD': [0, 1], 'PPG': [3]}, 2, False),          # wave 0 has no ABD at all
    (SM4, 4, 5, 3, {'THX': [1], 'ECG': [4]}, 3, False),            # ragged waves: 1 + 2 + 2 samples
    ({'ECG': 'UNI'}, 4, 3, 40, None, 8, False),                      # more waves than samples: one sample per wave
    ({'EOG-L': 'EOG-L', 'EOG-R': 'EOG-R'}, 5, 4, 5, {'EOG-L': [2]}, 2, True),   # ten-block encoders, causal padding
    ({'ABD': 'RESP', 'THX': 'RESP', 'ECG': 'ECG'}, 4, 4, 5, None, 2, 'chunk'),  # shared encoder (two passes per wave on one stream), chunk-causal
])
def test_wave_step_equals_one_pass_step(signal_map, nc, B, S, missing, waves, causal):
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=nc, causal=bool(causal), chunk_causal=causal == 'chunk')
    sd = O.make_state_dict(cfg, seed=33)
    x, y = O.make_inputs(cfg, B, S, seed=34, missing=missing)
    y[0, 0] = -1
    xd, yd = to_dev(x), y.to(DEV)
    m1, (o1,) = _one_step(cfg, sd, xd, yd, waves=1)
    m2, (o2,) = _one_step(cfg, sd, xd, yd, waves=waves)
    assert torch.equal(o1['logits'], o2['logits'])
    assert torch.equal(o1['cmat'], o2['cmat'])
    assert float(o1['count']) == float(o2['count'])
    assert float(o2['loss']) == pytest.approx(float(o1['loss']), rel=2e-6)
    assert float(o2['grad_norm']) == pytest.approx(float(o1['grad_norm']), rel=1e-5)
    for k, g1 in m1._engine.G.items():
        g2 = m2._engine.G[k]
        scale = float(g1.abs().max())
        if scale == 0.0:
            assert float(g2.abs().max()) == 0.0, k
            continue
        # (fp32 accumulation order: the waves cut the (sample, tile) list differently, so other workgroups sum other tiles -- measured up to
        #  2.3e-5 of the tensor's scale in the chunk-causal case, whose 'samples' are single epochs)
        assert float((g1 - g2).abs().max()) <= 4e-5 * scale, (k, float((g1 - g2).abs().max()), scale)
    # (the parameters after the step are not compared element-wise: AdamW's first update is lr * sign(g) wherever |g| >> eps, so a gradient
    # that is zero to rounding may move its weight by +-lr in either run; the goldens test above bounds the parameters after two steps)


def test_wave_step_single_stream_and_accumulation():
    """The same schedule on ONE stream (W2S_MULTI_STREAM=0: the waves then simply run one after the other) and under gradient accumulation
    (two micro-batches of two waves each) reproduce the multi-stream wave step bit for bit / the one-pass accumulation to rounding."""
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = O.make_state_dict(cfg, seed=35)
    x, y = O.make_inputs(cfg, 4, 4, seed=36, missing={'PPG': [1]})
    xd, yd = to_dev(x), y.to(DEV)
    m_ms, (o_ms,) = _one_step(cfg, sd, xd, yd, waves=2)
    m_ss, (o_ss,) = _one_step(cfg, sd, xd, yd, multi_stream=False, waves=2)
    assert torch.equal(m_ms._flat_grad, m_ss._flat_grad) and torch.equal(m_ms._flat, m_ss._flat)
    assert float(o_ms['loss']) == float(o_ss['loss'])
    m_a1, o_a1 = _one_step(cfg, sd, xd, yd, waves=1, accumulate=2)
    m_a2, o_a2 = _one_step(cfg, sd, xd, yd, waves=2, accumulate=2)
    assert [o['stepped'] for o in o_a2] == [False, True]
    for a, b in zip(o_a1, o_a2):
        assert torch.equal(a['logits'], b['logits'])
        assert float(b['loss']) == pytest.approx(float(a['loss']), rel=2e-6)
    scale = float(m_a1._flat_grad.abs().max())
    assert float((m_a1._flat_grad - m_a2._flat_grad).abs().max()) <= 2e-5 * scale


def test_wave_step_is_bit_reproducible_and_draws_fresh_dropout_masks():
    """Two runs of the wave step from the same state and seed give the same bits (fixed-order sums, no float atomics); with dropout on, every
    wave draws its own masks (a per-wave seed): the same two samples placed in both waves get different logits."""
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = O.make_state_dict(cfg, seed=37)
    x, y = O.make_inputs(cfg, 2, 4, seed=38)
    x = {s: torch.cat([v, v]) for s, v in x.items()}   # samples 2, 3 = samples 0, 1
    y = torch.cat([y, y])
    xd, yd = to_dev(x), y.to(DEV)
    runs = []
    for _ in range(2):
        model = build(SM4, 4, dropout=0.1)
        model.load_state_dict(sd)
        model.to(DEV).train()
        model._seed_base, model._seed_ctr = 1234, 0
        tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False, waves=2)
        out = tr.step(xd, yd)
        runs.append((out['logits'].clone(), model._flat_grad.clone(), float(out['loss'])))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]
    lg = runs[0][0]
    assert not torch.equal(lg[:2], lg[2:])
