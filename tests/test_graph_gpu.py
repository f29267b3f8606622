"""GPU test: the quantize + GEMM pair replayed from a captured hipGraph equals the eager result."""
import pytest

from micromix_amd.graph import GraphedForward
from micromix_amd.qlinear import QLinearLayer

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager(dev):
    import torch
    g = torch.Generator().manual_seed(3)
    k, split = 1024, (512, 128, 384)
    idx = torch.randperm(k, generator=g)
    layers = []
    for n, bias in ((512, True), (256, False)):
        lin = torch.nn.Linear(k, n, bias=bias, dtype=torch.bfloat16)
        layers.append(QLinearLayer(lin.to(dev), p8_num=split[2], p6_num=split[1], reorder_index=idx))
    x0 = torch.randn((1, 16, k), generator=g).to(torch.bfloat16).to(dev)
    graphed = GraphedForward(layers, x0)
    for seed in (4, 5):
        x = torch.randn((1, 16, k), generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16).to(dev)
        outs = graphed(x)
        torch.cuda.synchronize()
        for layer, y in zip(layers, outs):
            assert torch.equal(y, layer(x))
