"""wav2sleep hot path (train step / inference forward) on AMD Instinct MI355X -- hand-written HIP kernels
(wav2sleep_amd/csrc -> libw2s_hip.so, C ABI in include/w2s.h) behind the reference's own module surface."""
from . import inputs, lib, settings, trainer, wav2sleep  # noqa: F401
from .api import load_model, predict  # noqa: F401
from .checkpoint import (EMACallback, lightning_checkpoint, load_lightning_checkpoint, save_lightning_checkpoint,  # noqa: F401
                         save_model)
from .data import ParquetDataset, load_dataset, predict_on_folder, save_predictions  # noqa: F401
from .inputs import causal_rolling_normalize  # noqa: F401
from .stats import cohens_kappa, confusion_accuracy  # noqa: F401
from .trainer import (ExpWarmUpScheduler, FusedTrainStep, SignalMasker, SleepLightningModule, SleepModule,  # noqa: F401
                      exp_warmup_lr, invert_signals)
from .ppgnet import SleepPPGNet  # noqa: F401
from .serving import GraphedForward  # noqa: F401
from .wav2sleep import (ConvBlock1D, ConvLayer1D, DilatedConvBlock, MultiModalAttentionEmbedder, SequenceCNN, SignalEncoder,  # noqa: F401
                        SignalEncoders, Wav2Sleep)

__all__ = ['Wav2Sleep', 'SleepPPGNet', 'GraphedForward', 'SignalEncoder', 'ConvBlock1D', 'ConvLayer1D', 'DilatedConvBlock', 'SignalEncoders', 'MultiModalAttentionEmbedder', 'SequenceCNN', 'load_model', 'predict', 'FusedTrainStep',
           'SleepModule', 'SleepLightningModule', 'SignalMasker', 'invert_signals', 'ExpWarmUpScheduler', 'exp_warmup_lr',
           'cohens_kappa', 'confusion_accuracy', 'EMACallback', 'lightning_checkpoint', 'save_lightning_checkpoint',
           'load_lightning_checkpoint', 'save_model', 'ParquetDataset', 'load_dataset', 'save_predictions', 'predict_on_folder', 'causal_rolling_normalize']
