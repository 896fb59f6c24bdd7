"""Data-parallel exchange for the train step: one process per GPU, RCCL (`nccl` backend) over xGMI.

The reference relies on Lightning's DDP (scripts/config/training/main.yaml:14-20) plus a per-step
`barrier()+all_reduce` of the confusion matrix and a `sync_dist` loss all-reduce (trainer/main.py:41-46,170-182).
Here: gradients live in ONE flat buffer whose order is (encoders..., mixer, sequence CNN, classifier); backward
produces them in the reverse order, so the reducer all-reduces contiguous RANGES as they complete, on a side
stream, overlapped with the remaining encoder backward; the loss / count / confusion counts travel in one small
packed all-reduce.  The 1/world scaling is folded into the loss gradient (w2s_ce_fwd_bwd gscale), so SUM == mean.
Pure torch.distributed (device-agnostic: the gloo/CPU tests exercise exactly this code).
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def flat_layout(shapes) -> tuple[list, int]:
    """(offset, numel, shape) of every parameter in ONE fp32 buffer, each slice starting on a 16-byte boundary; total length."""
    layout, off = [], 0
    for shape in shapes:
        n = 1
        for d in shape:
            n *= int(d)
        layout.append((off, n, tuple(shape)))
        off += (n + 3) // 4 * 4
    return layout, off


def reduce_ranges(layout, names) -> dict:
    """Contiguous flat ranges in the order backward completes them: '_tail' = everything that is not inside an encoder (signal
    embedding, set-fusion transformer, SequenceCNN, classifier -- final first), then one range per encoder as its backward ends.
    The ranges are disjoint and cover the whole buffer (tests/test_ddp_gloo_cpu.py)."""
    ranges = {}
    for (o, n, _), name in zip(layout, names):
        key = name.split('.')[2] if name.startswith('signal_encoders.encoders.') else '_tail'
        lo, hi = ranges.get(key, (o, o))
        ranges[key] = (min(lo, o), max(hi, o + (n + 3) // 4 * 4))
    return ranges


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class FlatGradReducer:
    def __init__(self, flat_grad: torch.Tensor, group=None, side_stream: bool = True):
        self.flat = flat_grad
        self.group = group
        self.world = world_size(group)
        # W2S_FORCE_COLLECTIVES=1: issue the collectives even at world size 1 (exercises the RCCL + side-stream path on one GPU)
        self.force = os.environ.get('W2S_FORCE_COLLECTIVES') == '1' and dist.is_available() and dist.is_initialized()
        self.handles = []
        self.stream = None
        if (self.world > 1 or self.force) and flat_grad.is_cuda and side_stream:
            self.stream = torch.cuda.Stream(device=flat_grad.device)

    @property
    def grad_scale(self) -> float:
        """Fold into dLoss so that SUM over ranks equals the DDP mean (per-rank mean loss, then mean over ranks)."""
        return 1.0 / self.world

    def reduce_range(self, lo: int, hi: int):
        """All-reduce flat[lo:hi] (SUM).  Called as soon as that range's gradients are final."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        chunk = self.flat[lo:hi]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Make the compute stream wait for every outstanding range."""
        if self.world == 1 and not self.force:
            return
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        for h in self.handles:
            h.wait()
        self.handles = []


def reduce_metrics(loss_count: torch.Tensor, cmat: torch.Tensor, group=None):
    """One packed SUM all-reduce of [loss*count, count, cmat...] (fp64: integer counts are exact up to 2^53).

    Returns (global mean loss over all valid labels, per-rank-mean loss averaged over ranks as the reference
    logs it with sync_dist=True, summed confusion matrix).
    """
    nc = cmat.shape[0]
    w = world_size(group)
    loss, count = loss_count[0].double(), loss_count[1].double()
    packed = torch.cat([torch.stack([loss * count, count, loss]), cmat.reshape(-1).double()])
    if w > 1 or (os.environ.get('W2S_FORCE_COLLECTIVES') == '1' and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    cm = packed[3:].round().long().reshape(nc, nc)
    return packed[0] / packed[1], packed[2] / w, cm
