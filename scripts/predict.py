"""Sleep-stage predictions for a folder of recordings with a trained wav2sleep model on MI355X.

Command-line counterpart of the reference's `scripts/predict.py` (same flags): parquet files in, one `<name>.preds.csv` per
recording out (`Timestamp`, `Pred` [, `Stage`]), in a copy of the input folder's directory structure; when the inputs carry
labels, Cohen's kappa and accuracy of the run are logged.  Everything happens in `wav2sleep_amd.predict_on_folder`.

    python scripts/predict.py --input-folder recordings/ --output-folder preds/ --model-folder models/wav2sleep --no-preprocess
"""
import argparse
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

log = logging.getLogger('wav2sleep_amd.predict')

# flag, keyword arguments for add_argument
FLAGS = [
    ('--input-folder', dict(required=True, type=os.path.abspath, help='folder with the recordings (parquet; searched recursively)')),
    ('--output-folder', dict(required=True, type=os.path.abspath, help='where the .preds.csv files go (input tree is mirrored)')),
    ('--model-folder', dict(default=None, help='folder holding config.yaml + state_dict.pth (no network here: hf:// URIs are refused)')),
    ('--signals', dict(default=None, help='comma-separated subset of the signals the model knows, e.g. ECG,THX (default: all)')),
    ('--device', dict(type=str, default='auto', help="'auto', 'cuda' or 'cuda:N'")),
    ('--batch-size', dict(type=int, default=4)),
    ('--num-workers', dict(type=int, default=4)),
    ('--no-preprocess', dict(action='store_true', help='inputs are model-ready parquet files (EDF/CSV ingestion is not part of this build)')),
    ('--max-length-hours', dict(type=int, default=10, help='recordings are cropped to this many hours')),
    ('--overwrite', dict(action='store_true', help='replace prediction files that already exist')),
    ('--compile', dict(action='store_true', help='accepted for compatibility: the forward already is hand-written gfx950 code')),
]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog='predict', description=__doc__.split('\n')[0])
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    a = ap.parse_args(argv)
    if a.model_folder is None:
        ap.error('--model-folder is required (the reference defaults to a Hugging Face Hub URI; there is no network on this system)')
    logging.basicConfig(level=logging.INFO, format='%(message)s')

    import torch

    import wav2sleep_amd as W
    signals = [s.strip() for s in a.signals.split(',')] if a.signals else None
    preds, labels = W.predict_on_folder(input_folder=a.input_folder, output_folder=a.output_folder, model_folder=a.model_folder, signals=signals,
                                        device=a.device, batch_size=a.batch_size, num_workers=a.num_workers, preprocess=not a.no_preprocess,
                                        max_length_hours=a.max_length_hours, overwrite=a.overwrite, compile=a.compile, return_tensors=True)
    log.info('%d recordings, %d epochs each', preds.shape[0], preds.shape[1])
    if labels is not None:   # rows = true stage, columns = predicted stage; unscored epochs (-1) dropped
        nc = int(max(preds.max(), labels.max())) + 1
        scored = labels.reshape(-1) >= 0
        idx = labels.reshape(-1)[scored].long() * nc + preds.reshape(-1)[scored].long()
        cmat = torch.bincount(idx, minlength=nc * nc).reshape(nc, nc).numpy()
        log.info("Cohen's kappa %.4f, accuracy %.4f over %d scored epochs", W.cohens_kappa(cmat, n_classes=nc), W.confusion_accuracy(cmat), int(scored.sum()))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
