"""Round-2 GPU parity tests (`-m gpu`): the cases round 1 left soft.

* the reference's DEFAULT initialisation under seed 42 (near-tied logits: the hardest arg-max case), small golden + full size B=16,
  in both arithmetic modes (bf16x3 split precision and W2S_EXACT_FP32=1);
* the fused clip + AdamW kernel on a seeded gradient sequence at lr 1e-3 against torch.optim.AdamW + clip_grad_norm_ (+ the
  reference's scheduler) run by make_goldens_r2.py -- ten steps, parameters move by ~1e-2, tolerance 1e-5 of the movement;
* ten end-to-end train steps with the scheduler off against the reference's own run;
* gradient accumulation (Lightning accumulate_grad_batches) against the oracle;
* EMACallback against the reference's callback.

Tolerance conventions: logits are checked element-wise as |d| <= 1e-3*|want| + 2e-4*max|want| (north_star: 1e-3 rtol; the absolute
floor is for logits that are themselves ~0) AND in the max norm as before; arg-max labels exactly equal.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import wav2sleep_amd as W  # noqa: E402
from oracle import wav2sleep_oracle as O  # noqa: E402  (checker only)
from tests.golden_util import load  # noqa: E402
from tests.test_r2_pins_cpu import SM4, default_init_model  # noqa: E402

DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def note(msg: str):
    """Measured parity figures -> stdout and gpurun_out/parity_notes.txt (DESIGN.md quotes them)."""
    print(msg)
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_notes.txt'), 'a') as f:
            f.write(msg + '\n')


def logit_errors(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err, scale = np.abs(got - want), np.abs(want).max()
    return dict(max_abs=float(err.max()), scale=float(scale), max_rel_elementwise=float((err / np.maximum(np.abs(want), 0.1 * scale)).max()),
                ok_elementwise=bool((err <= 1e-3 * np.abs(want) + 2e-4 * scale).all()), ok_maxnorm=bool(err.max() <= 1e-3 * scale))


@pytest.fixture(params=['bf16x3', 'exact_fp32'])
def mode(request, monkeypatch):
    monkeypatch.setenv('W2S_EXACT_FP32', '1' if request.param == 'exact_fp32' else '0')   # read when the engine is built
    return request.param


@pytest.mark.parametrize('tag,B,S,seed,missing', [('a', 2, 16, 4242, None), ('b', 3, 8, 4243, {'ABD': [0], 'ECG': [1], 'PPG': [2]})])
def test_default_init_forward_matches_reference_golden(mode, tag, B, S, seed, missing):
    g = load('default_init')
    model = default_init_model().to(DEV).eval()
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    x, _ = O.make_inputs(cfg, B, S, seed=seed, missing=missing)
    with torch.no_grad():
        lg = model({k: v.to(DEV) for k, v in x.items()}).cpu()
    assert model._engine.split_precision == (mode == 'bf16x3')
    e = logit_errors(lg.numpy(), g[f'logits_{tag}'])
    note(f'default-init golden {tag} [{mode}]: {e} min top-2 margin {float(g[f"margin_{tag}"].min()):.3e}')
    assert e['ok_maxnorm'] and e['ok_elementwise'], e
    assert np.array_equal(lg.argmax(-1).numpy(), g[f'pred_{tag}'])


# (test_default_init_full_size_batch16_matches_oracle[bf16x3 | exact_fp32] lives in tests/test_a_children_gpu.py since round 6: one child process
#  shares the oracle's eight micro-batches with the B = 16 gradient check)


def _flat_of(chunks, layout, total):
    flat = torch.zeros(total)
    for (o, n, _), c in zip(layout, chunks):
        flat[o:o + n] = c.flatten()
    return flat


class _FlatHolder:
    """What FusedTrainStep / EMACallback need from a model, over caller-provided tensors (no network): flat parameter and gradient buffers."""

    def __init__(self, shapes, init_flat, num_classes=4):
        from wav2sleep_amd.ddp import flat_layout
        self._layout, total = flat_layout(shapes)
        self._flat = torch.zeros(total, device=DEV)
        self._flat_grad = torch.zeros(total, device=DEV)
        o2 = 0
        for (o, n, _) in self._layout:
            self._flat[o:o + n] = init_flat[o2:o2 + n].to(DEV)
            o2 += n
        self._engine = None
        self.num_classes = num_classes
        self._dirty = 0

    def _ensure_flat(self):
        pass

    def mark_params_dirty(self):
        self._dirty += 1

    def named_parameters(self):
        return [(f'p{i}', self._flat[o:o + n].view(shape)) for i, (o, n, shape) in enumerate(self._layout)]

    def packed(self, t=None):
        t = self._flat if t is None else t
        return torch.cat([t[o:o + n] for (o, n, _) in self._layout]).cpu()


@pytest.mark.parametrize('variant', ['const', 'sched', 'wd'])
def test_fused_clip_adamw_matches_torch_adamw_over_ten_steps(variant):
    """w2s_sumsq_partial + w2s_clip_coef + w2s_adamw on the flat buffers vs clip_grad_norm_(1.0) + torch.optim.AdamW(lr 1e-3, wd 1e-4)
    [+ ExpWarmUpScheduler(warmup 4, tau 5)] on the same seeded gradients: bias correction, decoupled weight decay, the clip coefficient
    above and below the threshold, the scheduler's first step.  Parameters move by up to 9e-3; the check is 1e-5 of the movement."""
    from tests.golden_util import OPT_SHAPES, grad_sequence as _grad_sequence
    g = load('optim')
    holder = _FlatHolder(OPT_SHAPES, torch.from_numpy(g['init']))
    kw = dict(warmup_steps=4, tau=5.0, scheduler=True) if variant == 'sched' else dict(scheduler=False)
    tr = W.FusedTrainStep(holder, lr=1e-3, weight_decay=1e-2 if variant == 'wd' else 1e-4, **kw)
    init = torch.from_numpy(g['init']).double()
    for k, grads in enumerate(_grad_sequence(OPT_SHAPES, 10, 78), start=1):
        holder._flat_grad.copy_(_flat_of(grads, holder._layout, holder._flat.numel()))
        lr = tr.apply_optimizer()
        assert lr == pytest.approx(float(g[f'{variant}.lr{k}']), rel=1e-6)   # the step's lr travels to the device as fp32
        assert float(tr.normcoef[0]) == pytest.approx(float(g[f'{variant}.gnorm{k}']), rel=2e-6)
        if k in (1, 2, 5, 10):
            want = torch.from_numpy(g[f'{variant}.param{k}'])
            moved = (want - init).abs().max()
            err = (holder.packed().double() - want).abs().max()
            assert float(err) <= 1e-5 * float(moved) + 4e-7, (k, float(err), float(moved))   # 4e-7: fp32 round-off of ten updates of weights of size <= 1 (ulp 6e-8 each)
    assert float(moved) > 4e-3 and holder._dirty == 10


def test_fused_adamw_detects_what_round1_could_not():
    """Resolving power of the check above: a dropped first step, a missing bias correction and a missing weight decay are each far outside it."""
    g = load('optim')
    init, p1, p10 = (torch.from_numpy(g[k]).double() for k in ('init', 'const.param1', 'const.param10'))
    moved = float((p10 - init).abs().max())
    tol = 1e-5 * moved + 4e-7
    assert float((p1 - init).abs().max()) > 100 * tol                          # losing step 1 (1e-3 per element) is visible
    wd10 = torch.from_numpy(g['wd.param10']).double()
    assert float((wd10 - p10).abs().max()) > 50 * tol                          # the 'wd' variant (decay 1e-2) separates decoupled weight decay from no decay


def test_ten_train_steps_match_reference_run():
    """The reference model trained for ten steps at lr 1e-3 (scheduler off) by make_goldens_r2.py vs FusedTrainStep on the same batches."""
    g = load('train10')
    signal_map = {'ABD': 'ABD', 'ECG': 'ECG'}
    cfg = O.ModelConfig(signal_map=signal_map, num_classes=4)
    sd = O.make_state_dict(cfg, seed=31)
    model = W.Wav2Sleep(W.SignalEncoders(signal_map, 128, 'gelu', norm='instance', chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4)
    model.load_state_dict(sd)
    model.to(DEV).train()
    tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False)
    for step in range(10):
        xs, ys = O.make_inputs(cfg, 2, 8, seed=3100 + step, missing={'ABD': [1]} if step % 2 else None)
        out = tr.step({k: v.to(DEV) for k, v in xs.items()}, ys.to(DEV))
        # trajectories of two fp32 implementations drift apart slowly (Adam's update is ~sign(g) on the first steps)
        assert float(out['loss']) == pytest.approx(float(g[f'loss{step}']), rel=1e-4 if step == 0 else 2e-2), step
        assert float(out['grad_norm']) == pytest.approx(float(g[f'gnorm{step}']), rel=1e-3 if step == 0 else 5e-2), step
        if step in (0, 9):
            new = model.state_dict()
            num = den = 0.0
            for k in sd:
                d = (new[k].detach().cpu().double() - sd[k].double())
                want_norm = float(g[f'dnorm{step}.{k}'])
                assert float(d.norm()) == pytest.approx(want_norm, rel=0.05, abs=1e-9), (step, k)
                want = g[f'dparam{step}.{k}']
                if want.shape == tuple(d.shape):
                    num += float(((d - torch.from_numpy(want)) ** 2).sum()); den += float((torch.from_numpy(want) ** 2).sum())
            rel = (num / den) ** 0.5
            note(f'train10 step {step}: relative L2 error of the parameter MOVEMENT over the fully stored tensors = {rel:.3e}')
            assert rel <= (0.03 if step == 0 else 0.15), (step, rel)


@pytest.mark.parametrize('k', [3])   # (k = 2 against the oracle: tests/test_a_ddp_flow_gpu.py::test_two_ranks_one_gpu_train_step_matches_oracle[2])
def test_gradient_accumulation_matches_oracle(k):
    """FusedTrainStep(accumulate=k) = Lightning accumulate_grad_batches=k (scripts/train.py:59-76): k micro-batches, each loss / k,
    gradients summed, ONE clip + AdamW.  Oracle: mean of the k per-micro-batch mean-loss gradients."""
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    sd = O.make_state_dict(cfg, seed=61)
    model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4)
    model.load_state_dict(sd)
    model.to(DEV).train()
    tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False, accumulate=k)
    want = {n: torch.zeros_like(v) for n, v in sd.items()}
    for mb in range(k):
        x, y = O.make_inputs(cfg, 2, 6, seed=620 + mb, missing={'THX': [0]} if mb == 1 else {'PPG': [1], 'ABD': [1]})
        out = tr.step({s: v.to(DEV) for s, v in x.items()}, y.to(DEV))
        assert out['stepped'] == (mb == k - 1) and tr.step_count == int(mb == k - 1)
        if mb < k - 1:   # no optimiser step yet: weights untouched
            assert all(torch.equal(model.state_dict()[n].cpu(), sd[n]) for n in sd)
        loss_o, _, g = O.loss_and_grads(sd, cfg, x, y)
        assert float(out['loss']) == pytest.approx(float(loss_o), rel=1e-4)
        for n in want:
            want[n] += g[n] / k
    for n, p in model._engine.G.items():
        rel = float((p.detach().cpu() - want[n]).norm() / (want[n].norm() + 1e-20))
        assert rel <= 2e-3, (n, rel)
    gn = torch.sqrt(sum((v.double() ** 2).sum() for v in want.values()))
    assert float(out['grad_norm']) == pytest.approx(float(gn), rel=1e-3)
    # a second optimiser step starts from a clean accumulator
    x, y = O.make_inputs(cfg, 2, 6, seed=700)
    sd1 = {n: v.detach().cpu().clone() for n, v in model.state_dict().items()}
    tr.step({s: v.to(DEV) for s, v in x.items()}, y.to(DEV))
    _, _, g1 = O.loss_and_grads(sd1, cfg, x, y)
    n0 = 'classifier.weight'
    assert float((model._engine.G[n0].cpu() - g1[n0] / k).norm() / (g1[n0] / k).norm()) <= 2e-3


@pytest.mark.parametrize('name', ['d999_s0', 'd9_s3', 'd0_s0', 'd1_s0'])
def test_ema_callback_matches_reference_callback(name):
    """wav2sleep_amd.EMACallback (one kernel per update / swap on the flat buffer) against the REFERENCE EMACallback's trajectory
    (trainer/callbacks.py:12-128, run by make_goldens_r2.py): start_step gating, update rule, swap for validation, adoption at train end."""
    g = load('ema')
    decay, start = {'d999_s0': (0.999, 0), 'd9_s3': (0.9, 3), 'd0_s0': (0.0, 0), 'd1_s0': (1.0, 0)}[name]
    shapes = [(5, 7), (5,), (3, 5), (3,)]
    holder = _FlatHolder(shapes, torch.from_numpy(g[f'{name}.init']))
    cb = W.EMACallback(decay=decay, start_step=start)
    cb.setup(None, holder, 'fit')
    traj = torch.from_numpy(g[f'{name}.params'])
    for step in range(traj.shape[0]):
        o2 = 0
        for (o, n, _) in holder._layout:
            holder._flat[o:o + n] = traj[step, o2:o2 + n].to(DEV)
            o2 += n
        cb.on_train_batch_end(None, holder, None, None, step)
        np.testing.assert_allclose(holder.packed(cb._ema_flat).numpy(), g[f'{name}.ema{step}'], rtol=2e-6, atol=1e-8, err_msg=f'step {step}')
    assert cb.state_dict()['step_count'] == int(g[f'{name}.step_count'])
    cb.on_validation_epoch_start(None, holder)
    np.testing.assert_allclose(holder.packed().numpy(), g[f'{name}.during_val'], rtol=2e-6, atol=1e-8)
    cb.on_validation_epoch_end(None, holder)
    np.testing.assert_array_equal(holder.packed().numpy(), g[f'{name}.after_val'])
    cb.on_train_end(None, holder)
    np.testing.assert_allclose(holder.packed().numpy(), g[f'{name}.train_end'], rtol=2e-6, atol=1e-8)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs a second GPU')
def test_second_device_runs_on_its_own_stream():
    """tensors on cuda:1 while cuda:0 is the current device."""
    cfg = O.ModelConfig(signal_map={'ECG': 'UNI'}, num_classes=4)
    sd = O.make_state_dict(cfg, seed=2)
    model = W.Wav2Sleep(W.SignalEncoders({'ECG': 'UNI'}, 128, 'gelu', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, nhead=8),
                        W.SequenceCNN(128, norm='layer'), 4)
    model.load_state_dict(sd)
    model.to('cuda:1').eval()
    x, _ = O.make_inputs(cfg, 2, 4, seed=3)
    torch.cuda.set_device(0)
    with torch.no_grad():
        lg = model({k: v.to('cuda:1') for k, v in x.items()})
    assert lg.device.index == 1
    want = O.forward(sd, cfg, x)
    assert float((lg.cpu() - want).abs().max()) <= 1e-3 * float(want.abs().max())


def test_wrong_device_stream_is_refused(monkeypatch):
    """lib._stream(): buffers on one device, launch stream of another -> W2SError instead of a fault (exercised on one GPU by faking the record)."""
    from wav2sleep_amd import lib
    t = torch.zeros(8, device=DEV)
    lib._p(t)
    monkeypatch.setattr(lib._tls, 'dev', t.device.index + 1)
    with pytest.raises(lib.W2SError):
        lib._stream()


def test_augment_consumes_the_device_rng_like_the_reference():
    """inputs.augment_ (one fused pass per signal) draws polarity for every signal first, then the masker's Bernoulli / categorical samples:
    the order of trainer/main.py:131-138 (`invert_signals`, then `SignalMasker.__call__`), so under one seed both produce the same batch and
    leave the generator in the same state."""
    from wav2sleep_amd.inputs import augment_
    names = ('ABD', 'THX', 'ECG', 'PPG')
    g = torch.Generator().manual_seed(5)
    x0 = {s: torch.randn(7, 96, generator=g) for s in names}
    x0['ECG'][1] = float('-inf'); x0['ABD'][3] = float('-inf'); x0['PPG'][5] = float('-inf')
    masker = W.SignalMasker({'ABD': 0.7, 'THX': 0.7, 'ECG': 0.5, 'PPG': 0.1}, backups=['ECG', 'PPG'])
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        a = augment_({k: v.clone().to(DEV) for k, v in x0.items()}, flip_polarity=True, masker=masker)
        nxt_a = torch.rand(4, device=DEV)
        torch.manual_seed(seed)
        b = masker(W.invert_signals({k: v.clone().to(DEV) for k, v in x0.items()}))
        nxt_b = torch.rand(4, device=DEV)
        assert torch.equal(nxt_a, nxt_b)
        for k in names:
            ia, ib = torch.isinf(a[k]), torch.isinf(b[k])
            assert torch.equal(ia, ib), k
            assert torch.equal(torch.where(ia, torch.zeros_like(a[k]), a[k]), torch.where(ib, torch.zeros_like(b[k]), b[k])), k


def test_model_compile_traces_through_the_custom_operator():
    """tests/model/test_compile.py of the reference with torch.compile ACTUALLY invoked: `model.compile(mode='max-autotune', fullgraph=True)`
    traces Wav2Sleep.forward as one `w2s::wav2sleep_forward` node (fake implementation for shapes, registered backward for autograd); the
    compiled forward and a compiled training step give the eager results bit for bit."""
    import torch._dynamo
    sm = {'ECG': 'ECG', 'PPG': 'PPG'}
    cfg = O.ModelConfig(signal_map=sm, num_classes=4)
    sd = O.make_state_dict(cfg, seed=6)

    def make():
        m = W.Wav2Sleep(W.SignalEncoders(sm, 128, 'gelu', norm='instance', chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4)
        m.load_state_dict(sd)
        return m.to(DEV)
    x, y = O.make_inputs(cfg, 2, 6, seed=61)
    xd = {k: v.to(DEV) for k, v in x.items()}
    eager, comp = make().eval(), make().eval()
    torch._dynamo.reset()
    try:
        comp.compile(mode='max-autotune', fullgraph=True)
        with torch.no_grad():
            a, b = eager(xd), comp(xd)
        assert torch.equal(a, b) and comp.compiled_with['fullgraph'] is True
        eager.train(); comp.train()
        la = torch.nn.functional.cross_entropy(eager(xd).view(-1, 4), y.to(DEV).view(-1).long(), ignore_index=-1)
        lb = torch.nn.functional.cross_entropy(comp(xd).view(-1, 4), y.to(DEV).view(-1).long(), ignore_index=-1)
        la.backward(); lb.backward()
        assert float(la) == float(lb)
        for (k, p), q in zip(eager.named_parameters(), comp.parameters()):
            assert torch.equal(p.grad, q.grad), k
        # the op is visible to torch.library tooling: fake tensor propagation gives the logits' shape without running a kernel
        from torch._subclasses.fake_tensor import FakeTensorMode
        with FakeTensorMode(allow_non_fake_inputs=True) as mode:
            fx = [mode.from_tensor(v) for v in xd.values()]
            out, ticket = torch.ops.w2s.wav2sleep_forward(eager._handle, False, False, ','.join(xd), fx, [mode.from_tensor(p.detach()) for p in eager.parameters()])
            assert tuple(out.shape) == (2, 6, 4) and ticket.device.type == 'cpu'
    finally:
        torch._dynamo.reset()


def test_scheduling_does_not_change_a_bit():
    """Four encoder streams fed round-robin with the trunk's weight gradients deferred beside the encoder backward (the default), or everything
    on ONE stream in program order (`engine.multi_stream = False` / W2S_MULTI_STREAM=0): pure scheduling -- the flat gradient of a train
    step must be bit-identical (fixed-order reductions, no float atomics).  (Rounds 2-4 also had switches for the launch order, the
    round-robin enqueue and the deferral on their own; settled and removed in round 5.)"""
    import wav2sleep_amd as W
    sm = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
    torch.manual_seed(7)
    model = W.Wav2Sleep(W.SignalEncoders(sm, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to('cuda').train()
    g = torch.Generator(device='cuda').manual_seed(3)
    spe = {'ABD': 256, 'THX': 256, 'ECG': 1024, 'PPG': 1024}
    x = {s: torch.randn(3, 24 * spe[s], device='cuda', generator=g) for s in sm}
    x['THX'][1] = float('-inf')
    y = torch.randint(0, 4, (3, 24), device='cuda', generator=g).float()
    model._ensure_flat()
    grads = []
    saved = model._engine.multi_stream
    try:
        for ms in (True, False, True):
            model._engine.multi_stream = ms
            model.zero_grad(set_to_none=True)
            logits = model(x)
            loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 4), y.reshape(-1).long())
            loss.backward()
            grads.append(model._flat_grad.clone())
    finally:
        model._engine.multi_stream = saved
    assert float(grads[0].abs().max()) > 0
    for k in range(1, len(grads)):
        assert torch.equal(grads[0], grads[k]), f'schedule {k} changed {int((grads[0] != grads[k]).sum())} gradient elements'


# (round 6: test_full_size_gradients_match_oracle[False | True] -- one full-size step at B = 2 against the oracle's autograd -- removed as duplicates:
#  the symmetric case is a sub-case of tests/test_a_children_gpu.py::test_benchmark_shape_batch16_gradients_match_oracle (same model and lengths,
#  B = 16, six missing pairs), the causal case IS ::test_causal_variant_full_length_gradients_match_oracle (same seed, batch and mask))


def test_full_size_gradients_are_bit_reproducible():
    """Four backward passes of the same full-size step (4 modalities x 960 epochs, B = 2, encoders on their own streams) give the same
    bits in all 183 gradient tensors.  Round 2 found the first-layer weight-gradient kernel (w2s_enc_first_bwd) failing exactly this --
    only at this size, only with other kernels sharing the CUs, never in isolation (tools/determinism_probe*.py)."""
    torch.manual_seed(42)
    model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', causal=False, chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', causal=False, num_layers=2, kernel_size=7, num_dilations=6), 4).to(DEV).train()
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    x, y = O.make_inputs(cfg, 2, 960, seed=123, missing={'THX': [1]})
    x = {k: v.to(DEV) for k, v in x.items()}
    y = y.to(DEV)
    runs = []
    for _ in range(4):
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x).reshape(-1, 4), y.reshape(-1).long(), ignore_index=-1)
        loss.backward()
        torch.cuda.synchronize()
        runs.append(model._flat_grad.clone())
    for k in range(1, 4):
        diff = runs[0] != runs[k]
        assert not bool(diff.any()), f'run {k}: {int(diff.sum())} gradient elements differ'
