"""One rank of the multi-rank train-step flow test (tests/test_a_ddp_flow_gpu.py starts two of these as fresh processes, both on GPU 0,
with the gloo backend: RCCL refuses two ranks on one device).  Each rank builds its OWN default initialisation (different seeds: the
trainer must broadcast rank 0's), draws its own ragged batch with the reference's SignalMasker rule (= BASELINE configs[4] semantics),
runs FusedTrainStep for `accumulate` micro-batches and writes what it saw and what it ended with to <out_dir>/rank<r>.pt.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/ddp_flow_worker.py <out_dir> <accumulate>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SM4 = {'ABD': 'ABD', 'THX': 'THX', 'ECG': 'ECG', 'PPG': 'PPG'}
B, S = 3, 6


def generic_model(W):
    """a configuration outside the production family (GroupNorm / LeakyReLU encoders, post-norm ReLU mixer, RMS SequenceCNN): GenericTrainStep"""
    return W.Wav2Sleep(W.SignalEncoders({'ECG': 'ECG', 'THX': 'THX'}, feature_dim=32, activation='leaky', norm='group'),
                       W.MultiModalAttentionEmbedder(32, layers=1, nhead=2, dim_ff=64, activation='relu', norm_first=False),
                       W.SequenceCNN(32, norm='rms', activation='silu', dropout=0.0, num_layers=1, num_dilations=2), 4)


def generic_batch(rank, mb):
    g = torch.Generator().manual_seed(900 + 10 * rank + mb)
    x = {'ECG': torch.randn(B, S * 1024, generator=g), 'THX': torch.randn(B, S * 256, generator=g)}
    x['THX' if rank == 0 else 'ECG'][1] = float('-inf')   # a different missing modality per rank
    y = torch.randint(0, 4, (B, S), generator=g).float()
    y[rank, :2 + rank] = -1                               # unequal label counts per rank
    return x, y


def main_generic(out_dir: str, accumulate: int):
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get('W2S_DIST_BACKEND', 'gloo'))
    import wav2sleep_amd as W
    from wav2sleep_amd.trainer import GenericTrainStep
    torch.manual_seed(2000 + rank)                        # every rank its own initialisation: the trainer must broadcast rank 0's
    model = generic_model(W).to('cuda:0').train()
    tr = GenericTrainStep(model, lr=1e-3, scheduler=False, accumulate=accumulate)
    start = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    outs = []
    for mb in range(accumulate):
        x, y = generic_batch(rank, mb)
        outs.append(tr.step({k: v.to('cuda:0') for k, v in x.items()}, y.to('cuda:0')))
    torch.cuda.synchronize()
    torch.save({'start': start, 'params': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'flat_grad': tr.flat_grad.detach().cpu(),
                'names': [n for n, _ in model.named_parameters()], 'layout': model._layout, 'loss': float(outs[-1]['loss']),
                'grad_norm': float(outs[-1]['grad_norm']), 'step_count': tr.step_count}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def main(out_dir: str, accumulate: int):
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get('W2S_DIST_BACKEND', 'gloo'))
    import wav2sleep_amd as W
    from oracle import wav2sleep_oracle as O   # seeded input / weight generators shared with the checking side
    cfg = O.ModelConfig(signal_map=SM4, num_classes=4)
    torch.manual_seed(1000 + rank)
    model = W.Wav2Sleep(W.SignalEncoders(SM4, 128, 'gelu', norm='instance', chunk_causal=False),
                        W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.0, dim_ff=512, nhead=8),
                        W.SequenceCNN(128, dropout=0.0, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), 4)
    if rank == 0:
        model.load_state_dict(O.make_state_dict(cfg, seed=51))
    model.to('cuda:0').train()
    tr = W.FusedTrainStep(model, lr=1e-3, scheduler=False, accumulate=accumulate)   # broadcasts rank 0's weights
    start = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    masker = W.SignalMasker({'ABD': 0.7, 'THX': 0.7, 'ECG': 0.5, 'PPG': 0.1}, backups=['ECG', 'PPG'])   # inputs/cardiorespiratory/all.yaml
    torch.manual_seed(70 + rank)
    batches, outs = [], []
    for mb in range(accumulate):
        x, y = O.make_inputs(cfg, B, S, seed=500 + 10 * rank + mb)
        x = {k: v.to('cuda:0') for k, v in x.items()}
        masker(x)   # -inf rows: a different modality subset per sample and per rank
        out = tr.step(x, y.to('cuda:0'))
        batches.append(({k: v.cpu() for k, v in x.items()}, y))
        outs.append(out)
    gmean, rmean, cm = tr.metrics()
    torch.cuda.synchronize()
    torch.save({'start': start, 'batches': batches, 'params': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                'flat_grad': model._flat_grad.detach().cpu(), 'layout': model._layout, 'names': [n for n, _ in model.named_parameters()],
                'loss': float(outs[-1]['loss']), 'grad_norm': float(outs[-1]['grad_norm']), 'stepped': [bool(o['stepped']) for o in outs],
                'cm': cm.cpu(), 'gmean': float(gmean), 'rmean': float(rmean), 'step_count': tr.step_count},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    (main_generic if len(sys.argv) > 3 and sys.argv[3] == 'generic' else main)(sys.argv[1], int(sys.argv[2]))
