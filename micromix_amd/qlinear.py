"""QLinearLayer -- host-side mirror of the reference operator (model/qLinearLayer.py:20-74).

Same constructor signature and forward contract; the weight is packed once at construction
with `reorder_quantize_w4` (production W4 mode, qLinearLayer.py:50) and every forward is
`reorder_quantize_x` + `matmul` (+ bias).  The bias add is fused into the GEMM epilogue with
the same two-rounding semantics as the reference's separate `y + bias`.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import mixedgemm


def find_qlinear_layers(module, name=""):
    """qLinearLayer.py:8-17."""
    if type(module) == QLinearLayer:
        if getattr(module, "enable_quant", True):
            return {name: module}
    res = {}
    for name1, child in module.named_children():
        res.update(find_qlinear_layers(child, name=name + "." + name1 if name != "" else name1))
    return res


class QLinearLayer(nn.Module):
    def __init__(self, originalLayer: nn.Linear, p8_num, p6_num, reorder_index, out_reorder_index=None,
                 weight_mode: str = "w4"):
        super().__init__()
        self.in_features = originalLayer.in_features
        self.out_features = originalLayer.out_features
        if originalLayer.bias is not None:
            self.register_buffer("bias", originalLayer.bias.detach().to(torch.bfloat16))
        else:
            self.bias = None
        self.p6_num = int(p6_num)  # p4_num, p6_num, p8_num must be multiples of 128 (qLinearLayer.py:40)
        self.p8_num = int(p8_num)
        self.p4_num = self.in_features - self.p8_num - self.p6_num
        w = originalLayer.weight.data
        if not w.is_cuda:
            w = w.cuda()
        w = w.to(torch.bfloat16).contiguous()
        self.reorder_index = reorder_index.to(torch.int16).to(w.device).contiguous()
        quant = mixedgemm.reorder_quantize_w4 if weight_mode == "w4" else mixedgemm.reorder_quantize_w
        self.BN, self.BS, self.BO, self.SFBN, self.SFBS, self.SFBO = quant(
            w, self.reorder_index, self.p4_num, self.p6_num, self.p8_num)

    @torch.no_grad()
    def forward(self, x):
        bsz, q_len, _ = x.shape
        x = x.reshape(bsz * q_len, -1).contiguous()
        AN, AS, AO, SFAN, SFAS, SFAO = mixedgemm.reorder_quantize_x(
            x, self.reorder_index, self.p4_num, self.p6_num, self.p8_num)
        bias = self.bias
        if bias is not None and bias.device != x.device:
            bias = bias.to(x.device)
        y = mixedgemm.matmul(AN, self.BN, AS, self.BS, AO, self.BO, SFAN, self.SFBN, SFAS, self.SFBS, SFAO, self.SFBO,
                             bias=bias)
        return y.reshape(bsz, q_len, -1)
