"""Train wav2sleep on MI355X: the counterpart of the reference's `scripts/train.py` (the `trainer.fit` / `trainer.test` it drives through
Hydra + Lightning, /root/reference/scripts/train.py:27-106) as a plain loop over `SleepModule` -- no Hydra, MLflow or Hub (out of scope).

    python scripts/train.py --train-folder data/train --val-folder data/val --out runs/a [--signals ABD,THX,ECG,PPG] [--epochs 30]
    python -m torch.distributed.run --nproc-per-node 8 scripts/train.py ...        # one rank per GPU, RCCL all-reduce (wav2sleep_amd/ddp.py)
    python scripts/train.py --synthetic 64 --epochs 1 --out /tmp/run               # smoke: synthetic 8-hour recordings, no files needed
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--train-folder'); ap.add_argument('--val-folder'); ap.add_argument('--out', required=True)
    ap.add_argument('--signals', default='ABD,THX,ECG,PPG'); ap.add_argument('--num-classes', type=int, default=4)
    ap.add_argument('--model', default='wav2sleep', choices=['wav2sleep', 'ppgnet'], help='ppgnet: SleepPPGNet on the one signal given (scripts/config/model/ppgnet.yaml)')
    ap.add_argument('--epochs', type=int, default=30); ap.add_argument('--batch-size', type=int, default=16)      # scripts/config/main.yaml
    ap.add_argument('--accumulate', type=int, default=1); ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--synthetic', type=int, default=0, help='train on this many synthetic recordings (no folders needed)')
    ap.add_argument('--synthetic-epochs', type=int, default=960, help='30-s epochs per synthetic recording (960 = 8 h)')
    ap.add_argument('--max-length-hours', type=int, default=10); ap.add_argument('--num-workers', type=int, default=4); ap.add_argument('--seed', type=int, default=42)
    a = ap.parse_args(argv)
    import torch.distributed as dist
    world, rank, local = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    import wav2sleep_amd as W
    from wav2sleep_amd.checkpoint import save_lightning_checkpoint, save_model
    from wav2sleep_amd.data import ParquetDataset, _get_parquet_files
    torch.manual_seed(a.seed)   # utils.fix_seeds: every rank the same initialisation (rank 0's is broadcast anyway)
    sig = [s.strip() for s in a.signals.split(',')]
    smap = {s: s for s in sig}
    if a.model == 'ppgnet':   # 10-h inputs only (SleepPPGNet.INPUT_LENGTH); trained on the generic path's tape (wav2sleep_amd/generic.py)
        if len(sig) != 1:
            ap.error('--model ppgnet takes exactly one signal (--signals PPG)')
        model = W.SleepPPGNet(n_classes=a.num_classes).to('cuda')
    else:
        model = W.Wav2Sleep(W.SignalEncoders(smap, 128, 'gelu', norm='instance', chunk_causal=False), W.MultiModalAttentionEmbedder(128, layers=2, dropout=0.1, dim_ff=512, nhead=8),
                            W.SequenceCNN(128, dropout=0.1, norm='layer', num_layers=2, kernel_size=7, num_dilations=6), a.num_classes).to('cuda')
    drop = {'ABD': 0.7, 'THX': 0.7, 'ECG': 0.5, 'PPG': 0.1}   # scripts/config/inputs/cardiorespiratory/all.yaml:9-18 (other signal sets: no masking)
    masker = W.SignalMasker({s: drop[s] for s in sig}, backups=[s for s in ('ECG', 'PPG') if s in sig]) if set(sig) <= set(drop) and len(sig) > 1 and a.model == 'wav2sleep' else None
    mod = W.SleepModule(model, num_classes=a.num_classes, masker=masker, lr=a.lr, accumulate_grad_batches=a.accumulate)

    def loader(folder, train):
        if a.synthetic:   # z-scored-like noise and random stages with 10 % unscored epochs (SURVEY.md 8d's synthetic overnight batch)
            from wav2sleep_amd.settings import COLS_TO_SAMPLES_PER_EPOCH as SPE
            g = torch.Generator().manual_seed(a.seed + train)
            n = a.synthetic if train else max(2, a.synthetic // 8)
            x = {k: torch.randn(n, a.synthetic_epochs * SPE[k], generator=g) for k in sig}
            y = torch.randint(0, a.num_classes, (n, a.synthetic_epochs), generator=g).float()
            y[torch.rand(n, a.synthetic_epochs, generator=g) < 0.1] = -1.0
            return [({k: v[i:i + a.batch_size] for k, v in x.items()}, y[i:i + a.batch_size]) for i in range(rank * a.batch_size, n, world * a.batch_size)]
        ds = ParquetDataset(_get_parquet_files(folder), sig, num_classes=a.num_classes, max_length_hours=a.max_length_hours)
        smp = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=train) if world > 1 else None
        return torch.utils.data.DataLoader(ds, batch_size=a.batch_size, shuffle=train and smp is None, sampler=smp, num_workers=a.num_workers, drop_last=train)

    train, val = loader(a.train_folder, True), loader(a.val_folder or a.train_folder, False)
    dev = lambda b: ({k: v.to('cuda', non_blocking=True) for k, v in b[0].items()}, b[1].to('cuda', non_blocking=True))
    for epoch in range(a.epochs):
        for batch in train:
            loss = mod.training_step(mod.on_after_batch_transfer(dev(batch), training=True))
        for batch in val:
            mod.validation_step(dev(batch), mode='val')
        cm = mod.aux_outputs['val'][None if mod.unified else '_'.join(sig)]['all']   # (single-modality models log under the signal's name: trainer/main.py:165-170)
        if rank == 0:
            print(f'epoch {epoch}: train loss {float(loss):.4f}, val kappa {W.cohens_kappa(cm.cpu().numpy(), a.num_classes):.4f}, accuracy {W.confusion_accuracy(cm.cpu().numpy()):.4f}', flush=True)
            os.makedirs(a.out, exist_ok=True)
            save_lightning_checkpoint(os.path.join(a.out, 'last.ckpt'), mod, epoch=epoch)
        mod.aux_outputs['val'].clear(); mod.aux_outputs['train'].clear()
    if rank == 0:
        save_model(os.path.join(a.out, 'model'), model)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
