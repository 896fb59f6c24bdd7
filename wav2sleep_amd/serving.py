"""Low-latency inference: the whole forward as ONE hipGraph.

A forward is ~250 kernel launches through ctypes; the host needs ~4.5 ms to enqueue them, the device 1.7 ms to run them for a single
eight-hour recording -- at small batch the HOST sets the latency (tools/graph_probe.py: 4.51 -> 1.71 ms at B = 1, 4.46 -> 2.30 at
B = 2, 4.46 -> 3.54 at B = 4; from batch 8 up the device is the slower side and a graph buys nothing).  The C ABI makes no allocation,
no synchronisation and keeps no hidden state (include/w2s.h), so the launches -- including the four encoder streams forked from and
joined to the capturing stream -- record into a graph as they are.

    fwd = wav2sleep_amd.GraphedForward(model, example_batch)     # fixed signals / batch size / length
    logits = fwd(batch)                                           # copy into the static inputs, one hipGraphLaunch, copy out

Inference only: the weights' packed copies are captured by address (capture again after changing the weights: `recapture()`),
and train-mode dropout draws its seeds on the host, so a training step is not replayable as a graph (nor would it gain: at the
benchmark's batch 16 the host enqueues a step in 11.6 ms against 35 ms of device time).
"""
from __future__ import annotations

import torch
from torch import Tensor

from .lib import W2SError


class GraphedForward:
    def __init__(self, model, example: dict[str, Tensor], warmup: int = 2):
        if not example:
            raise ValueError('example batch is empty')
        dev = next(iter(example.values())).device
        if dev.type != 'cuda':
            raise W2SError('GraphedForward runs on MI355X only (there is no CPU fallback)')
        if model.training:
            raise ValueError('GraphedForward captures the inference forward: call model.eval() first')
        if not (hasattr(model, 'fused_ok') and model.fused_ok()):
            raise NotImplementedError('only the production (fused) configuration is captured')
        self.model = model
        self.static_in = {k: v.detach().to(torch.float32).contiguous().clone() for k, v in example.items()}
        self.warmup = warmup
        self.graph = None
        self.static_out = None
        self.recapture()

    def recapture(self):
        """(Re)record the graph: after construction, and after the model's weights have changed."""
        dev = next(iter(self.static_in.values())).device
        with torch.cuda.device(dev), torch.no_grad():
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)   # warm-up off the capture stream (allocator pools, weight packing, lazy attributes)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(max(1, self.warmup)):
                    self.model(self.static_in)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            self._version = self.model.param_version()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.model(self.static_in)
            self.graph, self.static_out = g, out

    def __call__(self, x: dict[str, Tensor]) -> Tensor:
        if set(x) != set(self.static_in):
            raise ValueError(f'captured for signals {sorted(self.static_in)}, got {sorted(x)}')
        if self.model.param_version() != self._version:
            raise RuntimeError('the model weights changed since the graph was captured: call recapture()')
        for k, v in x.items():
            s = self.static_in[k]
            if v.shape != s.shape:
                raise ValueError(f'{k}: captured for shape {tuple(s.shape)}, got {tuple(v.shape)}')
            s.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone()
