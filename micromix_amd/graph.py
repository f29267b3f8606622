"""hipGraph capture of the hot path for small token counts.

At decode-sized M the two kernels of `QLinearLayer.forward` take ~5 + ~6 us on the GPU while the Python/ctypes/allocator
work to launch them takes ~35 us on the host, so the layer is host-bound.  `GraphedForward` captures
`reorder_quantize_x -> matmul (+bias)` of one or several QLinearLayers sharing an input into ONE hipGraph (through
torch.cuda.CUDAGraph, which records the work our C ABI queues on torch's current stream) and replays it with a single
launch.  Buffers are static: call `run(x)` with tensors of the captured shape; the results are overwritten by the next run.
"""
from __future__ import annotations

from typing import Sequence

import torch


class GraphedForward:
    def __init__(self, layers: Sequence, example_input: torch.Tensor, warmup: int = 3):
        self.layers = list(layers)
        self.static_in = example_input.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # warm-up outside capture (library load, attribute calls)
            for _ in range(warmup):
                for layer in self.layers:
                    layer(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = [layer(self.static_in) for layer in self.layers]

    def run(self, x: torch.Tensor):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out if len(self.static_out) > 1 else self.static_out[0]

    __call__ = run
