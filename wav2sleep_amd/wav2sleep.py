"""Host-side mirror of the reference's model surface (src/wav2sleep/models/{wav2sleep,blocks,utils}.py).

Same class names, constructor arguments, attribute names and `state_dict` keys/shapes as the reference, so
`load_model()` / Hydra `_target_` configs / reference checkpoints are drop-in.  The sub-modules below are
PARAMETER CONTAINERS built from the same torch.nn classes the reference uses (identical default init under
the same seed); torch's own forwards never run.  `Wav2Sleep.forward` hands the whole computation to the HIP
engine (wav2sleep_amd/engine.py -> libw2s_hip.so) and plugs into autograd as ONE node; `SignalEncoders`,
`MultiModalAttentionEmbedder` and `SequenceCNN` can also be called on their own (inference, same kernels).

There is no CPU path: CPU tensors raise (lib.W2SError).
"""

from __future__ import annotations

import os

import logging
import math

import torch
from torch import Tensor, nn

from .engine import Engine, EngineSpec
from .settings import COLS_TO_SAMPLES_PER_EPOCH

logger = logging.getLogger(__name__)

_NO_FORWARD = ('{} is a parameter container in wav2sleep_amd; run it through Wav2Sleep.forward '
               '(the HIP engine fuses across module boundaries).')


_ACTIVATIONS = ('relu', 'leaky', 'gelu', 'silu', 'swish', 'linear')


def _check_activation(name: str):
    """models/utils.py:61-74: the names get_activation accepts (GELU has fused kernels; the rest run on the generic path)."""
    if name not in _ACTIVATIONS:
        raise ValueError(f'{name=} is unsupported.')


def _gen(module: nn.Module):
    """Generic-path walker for a stand-alone sub-module call (inference; see generic.py)."""
    from .generic import GenericForward
    from .lib import W2SError
    for p in module.parameters():
        if p.device.type != 'cuda':
            raise W2SError('wav2sleep_amd runs on MI355X only: move the module to a cuda device (there is no CPU fallback)')
        break
    return GenericForward(training=module.training)


def _standalone_engine(module: nn.Module, prefix: str, spec: EngineSpec) -> Engine:
    """Engine over ONE sub-module's own parameters (inference only: the fused autograd path lives in Wav2Sleep)."""
    params = {prefix + k: p.detach() for k, p in module.named_parameters()}
    for k, p in params.items():
        if p.device.type != 'cuda':
            from .lib import W2SError
            raise W2SError('wav2sleep_amd runs on MI355X only: move the module to a cuda device (there is no CPU fallback)')
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise ValueError(f'{k}: fp32 contiguous parameters required')
    key = tuple((p.data_ptr(), p._version) for p in module.parameters())
    cached = getattr(module, '_w2s_engine', None)
    if cached is None or cached[0] != tuple(k[0] for k in key):
        module._w2s_engine = (tuple(k[0] for k in key), Engine(spec, params, None))
    return module._w2s_engine[1], hash(key)


class _SavedForwards(dict):
    """Saved activations of forwards whose backward has not run yet, keyed by ticket -- the COMPILED path's registry only: in eager mode
    the autograd node itself owns them from `setup_context` on (ops.py), so any number of forwards may precede a backward,
    `retain_graph=True` works, and a dropped graph frees its ~1.2 GB per recording with the node.  Under `torch.compile` the traced graph
    carries only the ticket, so the activations wait here; a forward whose graph is then dropped without a backward would pin them, hence a
    cap (W2S_SAVED_FORWARDS, default 2 most recent) with a warning when an entry is evicted."""

    def __init__(self, cap: int | None = None):
        super().__init__()
        import os
        self.cap = max(1, int(os.environ.get('W2S_SAVED_FORWARDS', 2) if cap is None else cap))

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        while len(self) > self.cap:
            old = next(iter(self))
            logger.warning('wav2sleep_amd: dropping the saved activations of forward #%d (more than %d compiled forwards are waiting for a '
                           'backward; raise W2S_SAVED_FORWARDS if that is intended) -- its backward will raise', old, self.cap)
            del self[old]


class _HandWritten:
    """`nn.Module.compile()` hook for the modules whose forward is a sequence of hand-written HIP launches: the reference calls
    `model.compile()` / `encoders.compile(fullgraph=True)` (api.py:96-97, tests/model/test_compile.py); there is nothing for
    Inductor to generate here, so the call is accepted, remembered and leaves the module as it is."""

    compiled_with: dict | None = None

    def compile(self, *args, **kwargs):
        self.compiled_with = dict(args=args, **kwargs)
        if isinstance(self, Wav2Sleep) and self.fused_ok():
            # The forward is one custom operator: Dynamo traces it (fullgraph=True holds), and there is nothing for Inductor to generate, so
            # the default backend is replaced by 'aot_eager' (fake tensors + the registered autograd formula, no code generation; `mode` /
            # `options` are Inductor switches and are dropped with it).  W2S_COMPILE_BACKEND selects another backend.
            import os
            kw = dict(kwargs)
            if 'backend' not in kw:
                kw['backend'] = os.environ.get('W2S_COMPILE_BACKEND', 'aot_eager')
                if kw['backend'] != 'inductor':
                    kw.pop('mode', None)
                    kw.pop('options', None)
            if next(self.parameters()).device.type == 'cuda':
                self._ensure_flat()   # parameter storage is settled before tracing (a model compiled on the host settles at its first forward)
            return nn.Module.compile(self, *args, **kw)
        logger.info('%s.compile(%s): the forward already is hand-written gfx950 code; nothing to compile', type(self).__name__, kwargs)


class ConvLayerNorm(nn.Module):
    """models/utils.py:9-23 (weights [1, C, 1])."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_features, 1))
        self.eps = eps


class ConvRMSNorm(nn.Module):
    """models/utils.py:26-38."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features, 1))
        self.eps = eps


class ConvGroupNorm(nn.Module):
    """models/utils.py:41-58 (the `nn.GroupNorm` lives under `.norm`, as in the reference's state dict)."""

    def __init__(self, num_features: int, num_groups: int = 8, channels_per_group: int | None = None, eps: float = 1e-5):
        super().__init__()
        if channels_per_group is not None:
            num_groups = num_features // channels_per_group
        if num_features < num_groups:
            logger.warning(f'{num_features=} is less than {num_groups=}. Will function as instance norm.')
            num_groups = num_features
        if num_features % num_groups != 0:
            raise ValueError(f'{num_features=} must be divisible by {num_groups=}.')
        self.norm = nn.GroupNorm(num_groups=num_groups, num_channels=num_features, eps=eps)


def get_norm(name: str | None = 'batch', causal: bool = False, *args, **kwargs) -> nn.Module:
    """models/utils.py:77-96: the parameter container of a normalisation layer (torch's forwards never run)."""
    norm_eps = kwargs.pop('norm_eps', None)
    if name == 'batch':
        return nn.BatchNorm1d(*args, **kwargs)
    elif name == 'layer':
        return ConvLayerNorm(*args, **kwargs)
    elif name == 'rms':
        return ConvRMSNorm(*args, **kwargs)
    elif name is None:
        return nn.Identity()
    elif name == 'instance':
        if norm_eps is not None:
            kwargs['eps'] = norm_eps
        return nn.InstanceNorm1d(*args, **kwargs)
    elif name == 'group':
        return ConvGroupNorm(*args, **kwargs)
    else:
        raise ValueError(f'Normalisation with {name=} and {causal=} unknown.')


def _to_cl(x_BCL: Tensor) -> Tensor:
    return x_BCL.transpose(1, 2).contiguous().float()


class ConvLayer1D(nn.Module):
    """models/blocks.py:129-186: conv -> (causal right trim) -> norm -> activation -> dropout.  Holds `.conv` and `.norm`."""

    def __init__(self, input_dim, output_dim, kernel_size=3, stride=1, padding=1, dilation=1, dropout: float = 0.0, causal: bool = False,
                 groups: int = 1, activation: str = 'relu', bias: bool = False, norm: str | None = 'batch', norm_eps: float | None = None):
        super().__init__()
        _check_activation(activation)
        self.causal = causal
        self.padding = (kernel_size - 1) * dilation if causal else padding
        self.conv = nn.Conv1d(input_dim, output_dim, kernel_size=kernel_size, stride=stride, padding=self.padding, groups=groups,
                              bias=bias or norm is None, dilation=dilation)
        if norm == 'weight':
            raise NotImplementedError("norm='weight' (weight-normalised convolution) has no gfx950 kernel")
        norm_kwargs = {'norm_eps': norm_eps} if norm_eps is not None else {}
        self.norm = get_norm(norm, causal=causal, num_features=output_dim, **norm_kwargs)
        self.norm_name = norm
        self.activation_name = activation
        self.dropout_p = dropout
        self.dropout = nn.Dropout(p=dropout)

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """[N, Cin, L] -> [N, Cout, L'] (channels-first like the reference; inference, generic path)."""
        return _gen(self).conv_layer(self, _to_cl(x)).transpose(1, 2)


class ConvBlock1D(nn.Module):
    """models/blocks.py:8-71."""

    def __init__(self, input_dim, output_dim, dropout: float = 0.0, activation: str = 'leaky', norm: str = 'batch', causal: bool = False,
                 norm_eps: float | None = None, use_residual: bool = True):
        super().__init__()
        self.use_residual = use_residual
        kw = dict(kernel_size=3, padding=1, activation=activation, norm=norm, dropout=dropout, causal=causal, norm_eps=norm_eps)
        self.conv1 = ConvLayer1D(input_dim, output_dim, stride=1, **kw)
        self.conv2 = ConvLayer1D(output_dim, output_dim, stride=1, **kw)
        self.conv3 = ConvLayer1D(output_dim, output_dim, stride=2, **kw)
        self.activation_name = activation
        if use_residual:
            self.downsample = nn.Conv1d(input_dim, output_dim, kernel_size=1, stride=2, padding=0, bias=False)
        else:
            self.register_parameter('downsample', None)   # blocks.py:53-55

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """[N, Cin, L] -> [N, Cout, L/2] (inference, generic path)."""
        return _gen(self).conv_block(self, _to_cl(x)).transpose(1, 2)


class DilatedConvBlock(nn.Module):
    """models/blocks.py:74-126."""

    def __init__(self, feature_dim=128, dropout=0.2, activation: str = 'leaky', norm: str = 'batch', kernel_size=7, causal: bool = False,
                 num_dilations=6):
        super().__init__()
        self.kernel_size = kernel_size
        self.dilations = [2 ** i for i in range(num_dilations)]
        blocks = []
        for d in self.dilations:
            k_eff = kernel_size + (kernel_size - 1) * (d - 1)
            blocks.append(ConvLayer1D(feature_dim, feature_dim, kernel_size=kernel_size, stride=1, dilation=d, padding=k_eff // 2,
                                      activation=activation, norm=norm, causal=causal))
        self.dropout = nn.Dropout(p=dropout)
        self.conv_layers = nn.Sequential(*blocks)
        self.activation_name = activation

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """[N, F, S] -> [N, F, S] (inference, generic path)."""
        return _gen(self).dilated_block(self, _to_cl(x)).transpose(1, 2)


class SignalEncoder(nn.Module):
    """models/wav2sleep.py:164-267: the whole-sequence path (:256-261), non-causal or -- `causal=True, chunk_causal=False`, what
    scripts/config/model/wav2sleep.yaml:10-11 builds for `causal: True` -- with causal-padded convolutions in the blocks; `chunk_causal=True`
    encodes every 30-s epoch on its own (:248-255)."""

    def __init__(self, input_dim=1, feature_dim=256, activation='gelu', samples_per_epoch=1024, norm='instance', initial_channels=16,
                 max_channels=128, causal=False, chunk_causal=True, output_norm=False, use_residual=True):
        super().__init__()
        _check_activation(activation)
        self.feature_dim = feature_dim
        self.samples_per_epoch = samples_per_epoch
        self.causal = causal
        self.chunk_causal = chunk_causal
        self.activation_name = activation
        self.norm_name = norm
        if samples_per_epoch & (samples_per_epoch - 1) != 0:
            raise ValueError(f'samples_per_epoch must be a power of 2, got {samples_per_epoch}')
        num_blocks = int(math.log2(samples_per_epoch)) - 2
        channels = [min(initial_channels * 2 ** (i // 2), max_channels) for i in range(num_blocks)]
        causal_conv = causal and not chunk_causal
        blocks = []
        for i, c in enumerate(channels):
            norm_i = ('instance' if i < 2 else 'layer') if norm == 'auto' else norm          # wav2sleep.py:206-212
            blocks.append(ConvBlock1D(input_dim, c, activation=activation, norm=norm_i, norm_eps=1e-2 if norm_i == 'instance' else None,
                                      causal=causal_conv, use_residual=use_residual))
            input_dim = c
        self.cnn = nn.Sequential(*blocks)
        self.epoch_dim = channels[-1] * 4
        self.linear = nn.Linear(self.epoch_dim, feature_dim)
        self.output_norm = nn.LayerNorm(feature_dim) if output_norm else nn.Identity()   # wav2sleep.py:232-233

    @torch.no_grad()
    def forward(self, x: Tensor) -> Tensor:
        """[B, T] -> [B, S, feature_dim] (inference, generic path; `SignalEncoders` / `Wav2Sleep` use the fused kernels for the production family)."""
        return _gen(self).signal_encoder(self, x)


class SignalEncoders(_HandWritten, nn.Module):
    """models/wav2sleep.py:83-161."""

    def __init__(self, signal_map: dict[str, str], feature_dim: int, activation: str, norm: str = 'instance', causal: bool = False,
                 chunk_causal: bool = True, embed_signals: bool = False, initial_channels: int = 16, max_channels: int = 128,
                 output_norm: bool = False, use_residual: bool = True) -> None:
        super().__init__()
        _check_activation(activation)
        self.feature_dim = feature_dim
        self.signal_map = dict(signal_map)
        self.causal = causal
        self.chunk_causal = chunk_causal
        self.activation_name = activation
        self.norm_name = norm
        self.use_output_norm = output_norm
        self.use_residual = use_residual
        self.initial_channels = initial_channels
        self.max_channels = max_channels
        encoders = {}
        for signal_name, encoder_name in self.signal_map.items():
            if encoder_name in encoders:
                continue
            if signal_name not in COLS_TO_SAMPLES_PER_EPOCH:
                raise ValueError(f"Column {signal_name} unrecognised. Doesn't have a sampling rate.")
            encoders[encoder_name] = SignalEncoder(input_dim=1, feature_dim=feature_dim, samples_per_epoch=COLS_TO_SAMPLES_PER_EPOCH[signal_name],
                                                   activation=activation, norm=norm, causal=causal, chunk_causal=chunk_causal,
                                                   initial_channels=initial_channels, max_channels=max_channels, output_norm=output_norm,
                                                   use_residual=use_residual)
        self.encoders = nn.ModuleDict(encoders)
        self.embed_signals = embed_signals
        self.sig_to_embedding_idx = {sig: i for i, sig in enumerate(sorted(signal_map.keys()))}
        if self.embed_signals:   # wav2sleep.py:127-133: one row per signal, added to its encoder's output
            self.embedder = nn.Embedding(num_embeddings=len(signal_map), embedding_dim=self.feature_dim)
        else:
            self.register_parameter('embedder', None)

    def __len__(self) -> int:
        return len(self.encoders)

    def get_encoder(self, signal_name: str) -> SignalEncoder:
        return self.encoders[self.signal_map[signal_name]]  # type: ignore

    def fused_ok(self) -> bool:
        """The production family the fused kernels are written for (scripts/config/model/wav2sleep.yaml): GELU, instance norm, 128 features,
        16 -> 128 channels.  Everything else runs on the generic path (generic.py; inference)."""
        return (self.activation_name == 'gelu' and self.norm_name == 'instance' and self.feature_dim == 128 and self.initial_channels == 16
                and self.max_channels in (16, 32, 64, 128))

    @torch.no_grad()
    def forward(self, x: dict[str, Tensor]) -> dict[str, Tensor]:
        """models/wav2sleep.py:146-161, inference only (training goes through Wav2Sleep.forward, one fused autograd node):
        dict signal -> [B, T]  ->  dict signal -> [B, S, feature_dim]; samples whose input row is -inf come back as -inf."""
        if not self.fused_ok():
            with torch.cuda.device(next(self.parameters()).device):
                return _gen(self).signal_encoders(self, x)
        spec = EngineSpec(signal_map=dict(self.signal_map), feature_dim=self.feature_dim, initial_channels=self.initial_channels,
                          max_channels=self.max_channels, causal=self.causal, chunk_causal=self.chunk_causal, embed_signals=self.embed_signals,
                          output_norm=self.use_output_norm, use_residual=self.use_residual)
        eng, ver = _standalone_engine(self, 'signal_encoders.', spec)
        with torch.cuda.device(next(self.parameters()).device):   # kernels go to the current device's stream: make it the parameters' device
            e = eng.encode(x, save=False, pack_key=ver, cls=False)
        B, S, F = e['B'], e['S'], self.feature_dim
        out = {}
        for m, sig in enumerate(e['sigs']):
            z = e['tokens'][:, e['R1'] + m, :].reshape(B, S, F).clone()
            out[sig] = torch.where(e['keeps'][m][:, None, None] == 0, float('-inf'), z)
        return {k: out[k] for k in x}


class MultiModalAttentionEmbedder(_HandWritten, nn.Module):
    """models/wav2sleep.py:270-346."""

    def __init__(self, feature_dim: int, layers: int = 4, dropout: float = 0.0, dim_ff: int = 512, activation: str = 'gelu',
                 norm_first: bool = True, nhead: int = 4, register_tokens: int = 0):
        super().__init__()
        _check_activation(activation)
        self.feature_dim = feature_dim
        self.activation_name = activation
        self.norm_first = norm_first
        self.dropout_p = dropout
        self.nhead = nhead
        self.dim_ff = dim_ff
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            encoder_layer = nn.TransformerEncoderLayer(d_model=feature_dim, dim_feedforward=dim_ff, activation={'relu': nn.ReLU(), 'leaky': nn.LeakyReLU(), 'gelu': nn.GELU(), 'silu': nn.SiLU(), 'swish': nn.SiLU(), 'linear': nn.Identity()}[activation], nhead=nhead,
                                                       batch_first=True, dropout=dropout, norm_first=norm_first)
            self.num_layers = layers
            self.transformer_encoder = nn.TransformerEncoder(encoder_layer, num_layers=layers)
        self.num_register_tokens = register_tokens
        self.register_tokens = nn.Parameter(torch.randn(1, 1, feature_dim, register_tokens + 1))

    def fused_ok(self) -> bool:
        """Pre-norm GELU layers with 128 features in 16-wide heads, FFN 384 or 512, at most 5 register tokens: the tuned attention / GEMM kernels."""
        return (self.activation_name == 'gelu' and self.norm_first and self.feature_dim == 128 and self.nhead * 16 == self.feature_dim
                and self.dim_ff in (384, 512) and 0 <= self.num_register_tokens <= 5)

    @torch.no_grad()
    def forward(self, z_dict: dict[str, Tensor]) -> Tensor:
        """models/wav2sleep.py:301-346, inference only: dict signal -> [B, S, F] (-inf rows = missing) -> CLS features [B, S, F]."""
        if not self.fused_ok():
            first_ = next(iter(z_dict.values())) if len(z_dict) else None
            if first_ is None:
                raise ValueError('No signals provided to MultiModalAttentionEmbedder.')
            with torch.cuda.device(first_.device):
                return _gen(self).mixer(self, z_dict)
        signals = sorted(z_dict.keys())
        if len(signals) == 0:
            raise ValueError('No signals provided to MultiModalAttentionEmbedder.')
        first = z_dict[signals[0]]
        B, S, F = first.shape
        if F != self.feature_dim:
            raise ValueError(f'Feature dimension {F} does not match {self.feature_dim=}.')
        spec = EngineSpec(signal_map={'ECG': 'ECG'}, feature_dim=self.feature_dim, mixer_layers=self.num_layers, mixer_nhead=self.nhead,
                          mixer_dim_ff=self.dim_ff, mixer_dropout=self.dropout_p, register_tokens=self.num_register_tokens)
        eng, ver = _standalone_engine(self, 'epoch_mixer.', spec)
        from . import lib
        R1 = self.num_register_tokens + 1
        N, D = B * S, len(signals) + R1
        if D > 12:
            raise ValueError(f'{len(signals)} signals + {R1} CLS/register tokens: the attention kernels hold at most 12 tokens per epoch')
        tokens = torch.empty(N, D, F, device=first.device, dtype=torch.float32)
        with torch.cuda.device(first.device):
            for r in range(R1):
                lib.add_rows(tokens.view(-1)[r * F:], D * F, eng.P['epoch_mixer.register_tokens'].view(-1)[r:], R1, None, 1, N, F, False)
        pads = [torch.zeros(B, dtype=torch.bool, device=first.device)] * R1
        for m, sig in enumerate(signals):  # host-side plumbing of [B,S,F] tensors: mask detection, zero fill, token slot copy
            z = z_dict[sig].float()
            m_B = torch.isinf(z).any(dim=2).any(dim=1)
            tokens[:, R1 + m, :] = torch.where(m_B[:, None, None], 0.0, z).reshape(N, F)
            pads.append(m_B)
        keypad = torch.stack(pads, dim=1).to(torch.uint8)[:, None, :].expand(B, S, D).reshape(N, D).contiguous()
        with torch.cuda.device(first.device):
            X, _ = eng.mix(tokens, keypad, self.dropout_p if self.training else 0.0, save=False)
        return X.view(N, X.numel() // N)[:, :F].reshape(B, S, F).clone()   # (row stride D*F, or F: the last layer may return the CLS rows alone)


class SequenceCNN(_HandWritten, nn.Module):
    """models/wav2sleep.py:349-390."""

    def __init__(self, feature_dim: int = 128, dropout: float = 0.2, num_layers: int = 2, activation: str = 'gelu', norm: str = 'batch',
                 causal: bool = False, num_dilations: int = 6, kernel_size: int = 7) -> None:
        super().__init__()
        _check_activation(activation)
        self.feature_dim = feature_dim
        self.activation_name = activation
        self.norm_name = norm
        self.causal = causal
        self.dropout_p = dropout
        self.num_layers = num_layers
        self.num_dilations = num_dilations
        self.kernel_size = kernel_size
        self.dilated_convs = nn.Sequential(*[DilatedConvBlock(feature_dim=feature_dim, dropout=dropout, activation=activation, norm=norm, causal=causal,
                                                              num_dilations=num_dilations, kernel_size=kernel_size) for _ in range(num_layers)])

    def fused_ok(self) -> bool:
        return self.activation_name == 'gelu' and self.norm_name == 'layer' and self.feature_dim == 128 and self.kernel_size == 7

    @torch.no_grad()
    def forward(self, x_BSF: Tensor) -> Tensor:
        """models/wav2sleep.py:379-390, inference only: [B, S, F] -> [B, S, F]."""
        B, S, F = x_BSF.shape
        if not self.fused_ok():
            with torch.cuda.device(x_BSF.device):
                return _gen(self).sequence_cnn(self, x_BSF)
        spec = EngineSpec(signal_map={'ECG': 'ECG'}, feature_dim=F, seq_blocks=self.num_layers, seq_dilations=self.num_dilations,
                          seq_kernel=self.kernel_size, seq_dropout=self.dropout_p, seq_causal=self.causal)
        eng, ver = _standalone_engine(self, 'sequence_mixer.', spec)
        from . import lib
        with torch.cuda.device(x_BSF.device):
            eng.ensure_packed(ver, need_bwd=False)
            pre, _ = eng.seq(x_BSF.float().contiguous(), F, B, S, self.dropout_p if self.training else 0.0, save=False)
            out = torch.empty_like(pre)
            lib.eltwise(lib.ELT_GELU, pre, None, out, pre.numel())
        return out


class Wav2Sleep(_HandWritten, nn.Module):
    """models/wav2sleep.py:16-80 -- same constructor, attributes and state_dict; compute on libw2s_hip.so."""

    def __init__(self, signal_encoders: SignalEncoders, epoch_mixer: MultiModalAttentionEmbedder, sequence_mixer: SequenceCNN,
                 num_classes: int):
        super().__init__()
        self.signal_encoders = signal_encoders
        self.epoch_mixer = epoch_mixer
        self.sequence_mixer = sequence_mixer
        self.feature_dim = self.epoch_mixer.feature_dim
        self.num_classes = num_classes
        self.classifier = nn.Linear(in_features=self.feature_dim, out_features=num_classes)
        self._flat = None
        self._flat_grad = None
        self._engine: Engine | None = None
        self._layout = []
        self._seed_base = torch.initial_seed() & 0x7FFFFFFF   # dropout masks follow torch.manual_seed (checkpointed as `w2s_seed_state`)
        self._seed_ctr = 0
        self._param_epoch = 0
        from . import ops
        self._handle = ops.register_model(self)
        self._saved_ctx = _SavedForwards()

    # ---------------------------------------------------------------- reference API
    @property
    def valid_signals(self) -> list[str]:
        return list(self.signal_encoders.signal_map.keys())

    def fused_ok(self) -> bool:
        """The production family runs on the fused kernels; W2S_FORCE_GENERIC=1 sends it down the generic path too (a cross-check of the two
        independent forward / backward implementations against each other: tests/test_r6_generic_grad_gpu.py)."""
        if os.environ.get('W2S_FORCE_GENERIC') == '1':
            return False
        return self.signal_encoders.fused_ok() and self.epoch_mixer.fused_ok() and self.sequence_mixer.fused_ok() and self.num_classes <= 8

    def forward(self, x: dict[str, Tensor]) -> Tensor:
        """dict[str -> [B, T_sig]] -> logits [B, S, num_classes].  The production family runs on the fused kernels (one autograd node);
        any other configuration of the reference's modules on the generic path (generic.py), also one autograd node when a gradient is wanted."""
        if not self.fused_ok():
            from .generic import GenericForward
            from .lib import W2SError
            dev = next(self.parameters()).device
            if dev.type != 'cuda':
                raise W2SError('wav2sleep_amd runs on MI355X only: move the model to a cuda device (there is no CPU fallback)')
            if any(v.device.type != 'cuda' for v in x.values()):
                raise W2SError('wav2sleep_amd runs on MI355X only: move the inputs to a cuda device (there is no CPU fallback)')
            seed = self._next_seed() if self.training else 0
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):   # one autograd node; backward = the walker's tape
                from .generic import differentiable
                return differentiable(self, lambda gf: gf.wav2sleep(self, x), self.training, seed)
            with torch.no_grad(), torch.cuda.device(dev):
                return GenericForward(training=self.training, seed=seed).wav2sleep(self, x)
        # ONE custom operator (ops.py: fake + autograd registered), so a caller's torch.compile traces through this forward
        params = list(self.parameters())
        if params[0].device.type != 'cuda' or any(v.device.type != 'cuda' for v in x.values()):
            from .lib import W2SError
            raise W2SError('wav2sleep_amd runs on MI355X only: move the model and its inputs to a cuda device (there is no CPU fallback)')
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return torch.ops.w2s.wav2sleep_forward(self._handle, self.training, save, ','.join(x.keys()), list(x.values()), params)[0]

    def predict(self, x: dict[str, Tensor]) -> Tensor:
        return self(x).argmax(axis=2)

    def config_dict(self) -> dict:
        """The resolved `model/wav2sleep.yaml` tree (Hydra `_target_` keys of the REFERENCE package) that rebuilds this
        model: what log.py:75-83 stores as `config.yaml` next to `state_dict.pth`."""
        se, em, sm = self.signal_encoders, self.epoch_mixer, self.sequence_mixer
        t = 'wav2sleep.models.wav2sleep.'
        return {
            '_target_': t + 'Wav2Sleep', 'num_classes': self.num_classes,
            'signal_encoders': {'_target_': t + 'SignalEncoders', 'signal_map': dict(se.signal_map), 'feature_dim': se.feature_dim,
                                'activation': 'gelu', 'norm': 'instance', 'causal': bool(se.causal), 'chunk_causal': bool(se.chunk_causal), 'embed_signals': bool(se.embed_signals),
                                'initial_channels': se.initial_channels, 'max_channels': se.max_channels, 'output_norm': bool(se.use_output_norm),
                                'use_residual': bool(se.use_residual)},
            'epoch_mixer': {'_target_': t + 'MultiModalAttentionEmbedder', 'feature_dim': em.feature_dim, 'dropout': em.dropout_p,
                            'activation': 'gelu', 'layers': em.num_layers, 'dim_ff': em.dim_ff, 'nhead': em.nhead,
                            'register_tokens': em.num_register_tokens},
            'sequence_mixer': {'_target_': t + 'SequenceCNN', 'feature_dim': self.feature_dim, 'dropout': sm.dropout_p, 'activation': 'gelu',
                               'norm': 'layer', 'causal': bool(sm.causal), 'num_layers': sm.num_layers, 'kernel_size': sm.kernel_size,
                               'num_dilations': sm.num_dilations},
        }

    @torch.no_grad()
    def forward_subsets(self, x: dict[str, Tensor], subsets) -> dict:
        """Logits for several signal subsets of ONE batch with every encoder run once (inference).

        The reference's validation / test / predict steps re-run the whole model per subset (trainer/main.py:188-240:
        all signals, ECG, ECG+THX, PPG, PPG+THX); the encoders are >95 % of the forward, and a subset only changes which
        tokens enter the set-fusion transformer, so the encoder outputs are computed once and re-used.
        subsets: iterable of tuples of signal names (None = all signals of `x`).  -> {subset: logits [B, S, nc]}"""
        self._ensure_flat()
        with torch.cuda.device(self._flat.device):
            return self._forward_subsets(x, subsets)

    def _forward_subsets(self, x, subsets) -> dict:
        from . import lib
        eng = self._engine
        e = eng.encode(x, save=False, pack_key=self.param_version())
        tokens, B, S, F = e['tokens'], e['B'], e['S'], self.feature_dim
        N, sigs = B * S, e['sigs']
        out = {}
        for sub in subsets:
            names = sorted(sub) if sub is not None else sigs
            for n in names:
                if n not in sigs:
                    raise ValueError(f'signal {n} is not in the batch')
            R1 = e['R1']
            cols = list(range(R1)) + [R1 + sigs.index(n) for n in names]
            D = len(cols)
            tok = tokens[:, cols, :].contiguous()  # [N, D, F] token gather (host-side plumbing on the small tensor)
            keep = torch.stack([torch.ones(B, device=tok.device)] * R1 + [e['keeps'][sigs.index(n)] for n in names], dim=1)
            keypad = (keep == 0).to(torch.uint8)[:, None, :].expand(B, S, D).reshape(N, D).contiguous()
            X, _ = eng.mix(tok, keypad, 0.0, save=False)
            pre, _ = eng.seq(X, X.numel() // N, B, S, 0.0, save=False)
            logits = torch.empty(B, S, self.num_classes, device=tok.device, dtype=torch.float32)
            lib.head_fwd(pre, F, eng.P['classifier.weight'], eng.P['classifier.bias'], logits, N, F, self.num_classes, True)
            out[tuple(sub) if sub is not None else None] = logits
        return out

    # ---------------------------------------------------------------- engine plumbing
    def spec(self) -> EngineSpec:
        se, em, sm = self.signal_encoders, self.epoch_mixer, self.sequence_mixer
        return EngineSpec(signal_map=dict(se.signal_map), feature_dim=self.feature_dim, num_classes=self.num_classes,
                          initial_channels=se.initial_channels, max_channels=se.max_channels, mixer_layers=em.num_layers,
                          mixer_nhead=em.nhead, mixer_dim_ff=em.dim_ff, mixer_dropout=em.dropout_p, seq_blocks=sm.num_layers,
                          seq_dilations=sm.num_dilations, seq_kernel=sm.kernel_size, seq_dropout=sm.dropout_p, causal=se.causal,
                          chunk_causal=se.chunk_causal, seq_causal=sm.causal, embed_signals=se.embed_signals,
                          register_tokens=em.num_register_tokens, output_norm=se.use_output_norm, use_residual=se.use_residual)

    def param_version(self) -> int:
        """Changes whenever any parameter was written (torch in-place ops bump `_version`; the fused AdamW kernel
        writes through raw pointers, so the trainer calls `mark_params_dirty()`).  Keys the packed-weight cache."""
        return self._param_epoch * (1 << 40) + sum(p._version for p in self.parameters())

    def mark_params_dirty(self):
        self._param_epoch += 1

    def _next_seed(self) -> int:
        self._seed_ctr += 1
        return (self._seed_base * 1000003 + self._seed_ctr) & 0x7FFFFFFF

    def _ensure_flat(self):
        """Move all parameters into ONE fp32 device buffer (16-B aligned slices); params become views of it.

        One buffer => one clip-norm reduction, one AdamW launch, one RCCL all-reduce for all 183 tensors.
        Re-done automatically if `.to()` / `load_state_dict(assign=True)` replaced the storage.
        """
        named = list(self.named_parameters())
        dev = named[0][1].device
        if dev.type != 'cuda':
            from .lib import W2SError
            raise W2SError('wav2sleep_amd runs on MI355X only: move the model to a cuda device (there is no CPU fallback)')
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            for (o, n, _), (_, p) in zip(self._layout, named):
                if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        from .ddp import flat_layout
        layout, off = flat_layout([p.shape for _, p in named])
        flat = torch.zeros(off, device=dev, dtype=torch.float32)
        gflat = torch.zeros(off, device=dev, dtype=torch.float32)
        P, G = {}, {}
        with torch.no_grad():
            for (o, n, shape), (name, p) in zip(layout, named):
                v = flat[o:o + n].view(shape)
                v.copy_(p.detach().to(torch.float32))
                p.data = v
                P[name] = v
                G[name] = gflat[o:o + n].view(shape)
        self._flat, self._flat_grad, self._layout = flat, gflat, layout
        self._engine = Engine(self.spec(), P, G)
