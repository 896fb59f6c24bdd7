"""Full-size and multi-process checks, each in its OWN child process with a hard time limit (tests/child_checks.py does the work).

Round 2 lost a GPU box during a suite run that held two of these checks in-process ("An unexplained lost box"); the verdict
asked for them back under `-m gpu` in a form where a hang costs one test, not the box: the child runs in its own session, is killed
(whole process group) at the limit and the test fails -- no re-exec, no retry.  This file sorts before every in-process GPU test on
purpose: the pool refuses to start a program from a process that has initialised the GPU, so the children are started while this
pytest process has not (if it already has, the test is skipped with that reason).

Covers: BASELINE configs[3] full-size gradients (+ bit-reproducibility of the ten-block encoders), the benchmark's own B = 16 shape
against the oracle, batch invariance, configs[4]'s five-modality set at full length, the RCCL path (`nccl`, forced collectives at world
size 1) and `bench.py --gpus 2` end to end over gloo."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(cmd, limit, env=None, tag='child'):
    """-> (returncode, combined output).  Kills the child's whole process group at `limit` seconds and fails the test."""
    try:
        p = subprocess.Popen(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             text=True, start_new_session=True, cwd=ROOT)
    except PermissionError as e:   # exec refused: this process had already initialised the GPU
        pytest.skip(f'child process could not be started from this process: {e}')
    try:
        out, _ = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        pytest.fail(f'{tag}: no result within {limit} s -- child killed.  Output so far:\n{(out or "")[-3000:]}')
    return p.returncode, out


def run_check(name, tmp_path, limit, env=None):
    res = os.path.join(tmp_path, f'{name}.json')
    rc, out = run_child([sys.executable, os.path.join(ROOT, 'tests', 'child_checks.py'), name, res], limit, env, tag=name)
    assert rc == 0, f'{name} failed (exit {rc}):\n{out[-4000:]}'
    return json.load(open(res))


def test_configs3_eog_full_size_gradients_match_oracle_and_are_bit_reproducible(tmp_path):
    """EOG-L + EOG-R, 3.9 M samples per recording, ten-block encoders, 5 classes, B = 1.  Bar: every gradient tensor <= 2e-3 relative L2."""
    r = run_check('eog_fullsize_grad', tmp_path, 900)
    assert r['bit_reproducible']
    assert r['loss'] == pytest.approx(r['want_loss'], rel=1e-4)
    assert r['worst_rel_l2'] <= 2e-3, (r['worst_tensor'], r['worst_rel_l2'])


def test_configs3_eog_batch16_equals_its_sixteen_recordings_one_by_one(tmp_path):
    """configs[3] at B = 16 (what `extra.configs3_eog_b16` times): logits bit-equal to the sixteen B = 1 runs, flat gradient = their
    valid-label-weighted mean within 1e-5 relative L2 (fp32 summation order is all that differs).  The B = 1 run is pinned to the oracle by
    test_configs3_eog_full_size_gradients_match_oracle_and_are_bit_reproducible."""
    r = run_check('eog_b16_consistency', tmp_path, 1200)
    assert r['finite'] and r['logits_equal'], r
    assert r['loss_batch'] == pytest.approx(r['loss_singles'], rel=1e-6)
    assert r['grad_rel_l2'] <= 1e-5, r


_B16 = {}


def _b16(tmp_path):
    """ONE child for everything that needs the benchmark-shape model at B = 16 (the oracle's eight micro-batches run once, in worker processes)."""
    if 'r' not in _B16:
        _B16['r'] = run_check('b16_fullsize_grad', tmp_path, 1500)
    return _B16['r']


def test_benchmark_shape_batch16_gradients_match_oracle(tmp_path):
    """4 modalities x 960 epochs, B = 16, ragged: one backward pass vs the oracle over 8 micro-batches of 2."""
    r = _b16(tmp_path)
    assert r['loss'] == pytest.approx(r['want_loss'], rel=1e-4)
    assert r['worst_rel_l2'] <= 2e-3, (r['worst_tensor'], r['worst_rel_l2'], r['over_1e3'])
    assert r['argmax_agreement'] >= 0.9999


@pytest.mark.parametrize('mode', ['bf16x3', 'exact_fp32'])
def test_default_init_full_size_batch16_matches_oracle(tmp_path, mode):
    """BASELINE configs[1] at full size -- 4 modalities, 960 epochs, batch 16, the weights scripts/train.py starts from -- inference forward in
    the default split-precision mode and under W2S_EXACT_FP32=1 (until round 5: tests/test_r2_parity_gpu.py, an oracle forward per mode):
    logits within 1e-3 of the scale in the max norm AND element-wise within 1e-3 |want| + 2e-4 scale, per recording; every label equal."""
    m = _b16(tmp_path)['modes'][mode]
    assert m['ok_maxnorm'] and m['ok_elementwise'], m
    assert m['flips'] == 0, m


def test_argmax_label_margin_over_sixteen_seeds_two_weight_states_both_modes(tmp_path):
    """the arg-max bar holds "by luck at the tightest epochs" unless someone counts.  16 seeds (own default initialisation, own
    full-length recording) x {as initialised, after 10 AdamW steps at lr 1e-3} x {bf16x3, W2S_EXACT_FP32=1} = 64 x 960 epochs against the
    oracle: the logit bar in every run, and label flips only where the oracle's own top-2 gap is inside twice the run's logit error (a tie
    to the precision of either side); the flip RATE of each mode is reported (gpurun_out/argmax_sweep.json)."""
    r = run_check('argmax_sweep', tmp_path, 900)
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        json.dump(r, open(os.path.join(out, 'argmax_sweep.json'), 'w'), indent=1)
    print(json.dumps(r['summary'], indent=1))
    # a gross mismatch that did not repeat when the same forward was run again would be a transient (round 6's were the harness's own race with
    # its worker pool, fixed in child_checks.py `_to_pool`): none is tolerated
    assert not r.get('transients'), r['transients']
    for key, a in r['summary'].items():
        assert a['max_rel_err'] <= 1e-3, (key, a)
        assert a['worst_flip_gap_over_err'] <= 2.0, (key, a)       # a flip further from a tie than the error allows would be a wrong result, not a tie
        assert a['flips'] <= 1e-3 * a['epochs'], (key, a)
    assert r['summary']['exact_fp32/init']['max_abs_err'] < r['summary']['bf16x3/init']['max_abs_err']


def test_causal_variant_full_length_gradients_match_oracle(tmp_path):
    """`causal: True` at 4 modalities x 960 epochs, B = 2: the full-length case of the causal convolutions (their 64-channel layers on the
    one-pass backward kernel with causal padding)."""
    r = run_check('causal_fullsize_grad', tmp_path, 1200)
    assert r['bit_reproducible'], r
    assert r['loss'] == pytest.approx(r['want_loss'], rel=1e-4)
    assert r['worst_rel_l2'] <= 2e-3, (r['worst_tensor'], r['worst_rel_l2'], r['over_1e3'])
    assert r['max_abs_logit_err'] <= 1e-3 * r['max_abs_logit'], r
    assert r['argmax_agreement'] >= 0.999, r


def test_logits_do_not_depend_on_batch_neighbours_full_length(tmp_path):
    r = run_check('batch_invariance', tmp_path, 600)
    assert r['b32_vs_halves_equal'] and r['b5_vs_singles_equal'], r


def test_configs4_five_modality_full_length_forward_matches_oracle(tmp_path):
    """{ABD, THX, ECG, PPG, EOG-L}: D = 6 tokens, 6/8/10-block encoders on five streams, 960 epochs, B = 2, ragged."""
    r = run_check('five_mod_fullsize_forward', tmp_path, 900)
    assert r['max_abs_err'] <= 1e-3 * r['max_abs_logit'], r
    assert r['argmax_agreement'] >= 0.9999, r


def test_rccl_forced_collectives_leave_every_bit_unchanged(tmp_path):
    """backend 'nccl' (= RCCL), world size 1, W2S_FORCE_COLLECTIVES=1: the ranged all-reduces on the side stream and the packed metric
    all-reduce run for real; a SUM over one rank is the identity, so gradients, parameters and confusion counts must not change."""
    r = run_check('nccl_forced_collectives', tmp_path, 600)
    assert r['grads_equal'] and r['params_equal'] and r['cm_equal'], r


def test_bench_two_ranks_end_to_end_over_gloo(tmp_path):
    """`python bench.py --gpus 2` (spawns its two ranks through torch.distributed.run, both on this box's one GPU: gloo, since RCCL refuses
    two ranks per device) -> ONE JSON line with n_gpus 2 and a positive whole-job value: the driver's multi-GPU launch cannot fail on plumbing."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    rc, out = run_child([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--epochs', '120',
                         '--no-cpu'], 900, env=dict(W2S_DIST_BACKEND='gloo', MASTER_PORT=str(port)), tag='bench --gpus 2')
    assert rc == 0, out[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['config']['global_batch'] == 4 and line['scaling'] == 'weak'
    assert line['config']['parallelism'] == 'dp2'
