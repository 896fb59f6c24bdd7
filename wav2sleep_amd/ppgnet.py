"""SleepPPG-Net (reference models/ppgnet.py:19-134; Kotzen et al. 2023): eight ConvBlock1D's down to 4800 x 256, a time-distributed
dense layer 1024 -> feature_dim, two DilatedConvBlocks, a linear classifier.  Same constructor, attribute names and state-dict keys as
the reference; forward AND backward run on the generic path (generic.py: BatchNorm / LeakyReLU, 256-channel layers as accumulating launches)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .wav2sleep import ConvBlock1D, DilatedConvBlock, _check_activation

__all__ = ('SleepPPGNet',)


class WindowEncoder(nn.Module):
    """ppgnet.py:83-110."""

    CHANNELS = [16, 16, 32, 32, 64, 64, 128, 256]

    def __init__(self, activation: str = 'leaky', norm: str = 'batch') -> None:
        super().__init__()
        blocks = []
        in_channels = 1
        for out_channels in self.CHANNELS:
            blocks.append(ConvBlock1D(in_channels, out_channels, activation=activation, norm=norm))
            in_channels = out_channels
        self.model = nn.Sequential(*blocks)


class DenseBlock(nn.Module):
    """ppgnet.py:113-134 (time-distributed dense layer)."""

    def __init__(self, in_dim: int = 1024, out_dim: int = 128, activation: str = 'leaky') -> None:
        super().__init__()
        _check_activation(activation)
        self.linear = nn.Linear(in_dim, out_dim)
        self.activation_name = activation


class SleepPPGNet(nn.Module):
    INPUT_LENGTH: int = 1228800  # 10 h @ 1024 samples per 30-s epoch

    def __init__(self, n_classes: int = 4, feature_dim: int = 128, dropout: float = 0.2, activation: str = 'leaky', norm: str = 'batch') -> None:
        super().__init__()
        self.feature_dim = feature_dim
        self.conv_block = WindowEncoder(activation=activation, norm=norm)
        self.dense = DenseBlock(in_dim=1024, out_dim=feature_dim)
        self.dilated_convs = nn.Sequential(
            DilatedConvBlock(feature_dim=feature_dim, dropout=dropout, activation=activation, norm=norm),
            DilatedConvBlock(feature_dim=feature_dim, dropout=dropout, activation=activation, norm=norm),
        )
        self.classifier = nn.Linear(in_features=feature_dim, out_features=n_classes)
        self._config = dict(n_classes=n_classes, norm=norm, feature_dim=feature_dim, activation=activation, dropout=dropout)

    def config_dict(self) -> dict:
        """The resolved `scripts/config/model/ppgnet.yaml` tree (Hydra `_target_` of the REFERENCE package) that rebuilds this model."""
        return {'_target_': 'wav2sleep.models.ppgnet.SleepPPGNet', **self._config}

    def forward(self, x_BT: Tensor) -> Tensor:
        """[N, 1 228 800] -> logits [N, 1200, n_classes].  With gradients enabled the result is ONE autograd node whose backward is the
        generic path's tape (generic.py: HIP kernels only), so `loss.backward()` fills `.grad` as the reference's training loop expects."""
        from .generic import GenericForward, differentiable
        from .lib import W2SError
        dev = next(self.parameters()).device
        if dev.type != 'cuda' or x_BT.device.type != 'cuda':
            raise W2SError('wav2sleep_amd runs on MI355X only: move the model and its input to a cuda device (there is no CPU fallback)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            self._seed = getattr(self, '_seed', 0) + 1
            return differentiable(self, lambda gf: gf.ppgnet(self, x_BT), self.training, self._seed)
        with torch.no_grad(), torch.cuda.device(dev):
            return GenericForward(training=self.training).ppgnet(self, x_BT)
