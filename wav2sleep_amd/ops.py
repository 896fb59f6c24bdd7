"""`torch.library` surface of the fused path: the whole-model forward as ONE custom operator (`w2s::wav2sleep_forward`) with a fake
(meta) implementation and a registered backward, so that a caller's `torch.compile` (scripts/train.py:40-43 compiles the Lightning module,
api.py:96-97 `model.compile()`, tests/model/test_compile.py `fullgraph=True`) traces straight through `Wav2Sleep.forward`: Dynamo sees one
opaque op instead of ctypes calls.  SURVEY 8b ("custom_op + register_autograd + register_fake").

The operator's tensor arguments are the signal tensors and the parameters (so autograd routes the parameter gradients); the model object
itself travels as an integer handle into a weak registry.  What backward needs (the engine's saved activations: ~1.2 GB per recording)
stays on the model under a ticket number that the forward returns as a second, host-side output; the backward is a second opaque operator
(`w2s::wav2sleep_backward`), so compiled backward graphs see one node as well.
"""
from __future__ import annotations

import weakref
from typing import List

import torch
import torch.utils._pytree as pytree
from torch import Tensor

from .settings import COLS_TO_SAMPLES_PER_EPOCH

_MODELS: 'weakref.WeakValueDictionary[int, torch.nn.Module]' = weakref.WeakValueDictionary()
_NEXT = [1]


def register_model(model) -> int:
    h = _NEXT[0]
    _NEXT[0] += 1
    _MODELS[h] = model
    return h


def _model(handle: int):
    m = _MODELS.get(handle)
    if m is None:
        raise RuntimeError(f'w2s::wav2sleep_forward: model handle {handle} is no longer alive')
    return m


@torch.library.custom_op('w2s::wav2sleep_forward', mutates_args=(), device_types='cuda')
def wav2sleep_forward(handle: int, training: bool, save: bool, names: str, signals: List[Tensor], params: List[Tensor]) -> tuple[Tensor, Tensor]:
    """(logits [B, S, num_classes], ticket) of the model behind `handle` for the signals `names` (comma separated, same order as
    `signals`).  ticket: a host int64 scalar naming the saved activations of this forward (0: nothing saved) -- a tensor, so that it
    travels through traced graphs, on the host, so that reading it costs no device synchronisation."""
    model = _model(handle)
    model._ensure_flat()
    eng = model._engine
    x = dict(zip(names.split(','), signals))
    eng.step_seed = model._next_seed() if training else 0
    with torch.cuda.device(model._flat.device):   # launches take the CURRENT device's stream (lib._stream)
        logits = eng.forward(x, train=training, save=save, pack_key=model.param_version())
    ticket = 0
    if save:
        _NEXT[0] += 1
        ticket = _NEXT[0]
        model._saved_ctx[ticket] = eng.ctx
    eng.ctx = None
    return logits, torch.tensor(ticket, dtype=torch.int64)


@wav2sleep_forward.register_fake
def _(handle, training, save, names, signals, params):
    model = _model(handle)
    first = names.split(',')[0]
    B, T = signals[0].shape
    return signals[0].new_empty(B, T // COLS_TO_SAMPLES_PER_EPOCH[first], model.num_classes), torch.empty((), dtype=torch.int64, device='cpu')


@torch.library.custom_op('w2s::wav2sleep_backward', mutates_args=(), device_types='cuda')
def wav2sleep_backward(handle: int, ticket: Tensor, glogits: Tensor) -> Tensor:
    """Flat gradient buffer (the layout of `model._layout`) of the forward named by `ticket`."""
    model = _model(handle)
    saved = model._saved_ctx.pop(int(ticket), None)
    if saved is None:
        raise RuntimeError('w2s::wav2sleep_backward: no saved forward for this ticket (the forward ran with gradients disabled, its '
                           'activations were dropped from the registry of the compiled path -- W2S_SAVED_FORWARDS --, or a compiled backward ran twice)')
    eng = model._engine
    eng.ctx = saved
    with torch.cuda.device(model._flat.device):
        eng.backward(glogits.contiguous().float())
    return model._flat_grad.clone()   # fresh storage per backward: autograd may keep or accumulate views of it


@wav2sleep_backward.register_fake
def _(handle, ticket, glogits):
    return glogits.new_empty(_model(handle)._flat.numel())


def _setup(ctx, inputs, output):
    handle, training, save, names, signals, params = inputs
    ctx.handle = handle
    ctx.nsig = len(signals)
    ctx.w2s_spec = None
    saved = None
    if type(output[1]) is torch.Tensor:
        # Eager mode (a real tensor, not a tracing proxy): the autograd node owns the saved activations from here on, under autograd's own
        # lifetime rules -- every tensor of the engine's context goes through save_for_backward, so it is released right after this node's
        # backward unless the caller asked for retain_graph=True, freed with the graph if backward never runs, and there is no limit on
        # the number of forwards in flight.  Traced graphs cannot hold Python objects; there the activations stay in the model's ticket
        # registry (wav2sleep._SavedForwards).
        saved = _model(handle)._saved_ctx.pop(int(output[1]), None)
    if saved is None:
        ctx.save_for_backward(output[1])
        return
    leaves, spec = pytree.tree_flatten(saved)
    idx = [i for i, leaf in enumerate(leaves) if isinstance(leaf, torch.Tensor)]
    ctx.save_for_backward(output[1], *[leaves[i] for i in idx])
    for i in idx:
        leaves[i] = None
    ctx.w2s_spec = (spec, leaves, idx)


def _backward(ctx, glogits, gticket):
    ticket, *tensors = ctx.saved_tensors
    model = _model(ctx.handle)
    if ctx.w2s_spec is not None:
        spec, leaves, idx = ctx.w2s_spec
        leaves = list(leaves)
        for i, t in zip(idx, tensors):
            leaves[i] = t
        model._saved_ctx[int(ticket)] = pytree.tree_unflatten(leaves, spec)   # handed to the operator for this call (popped there)
    gflat = torch.ops.w2s.wav2sleep_backward(ctx.handle, ticket, glogits)
    grads = [gflat[o:o + n].view(shape) for (o, n, shape) in model._layout]
    return None, None, None, None, [None] * ctx.nsig, grads


wav2sleep_forward.register_autograd(_backward, setup_context=_setup)
