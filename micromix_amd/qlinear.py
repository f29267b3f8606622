"""QLinearLayer -- host-side mirror of the reference operator (model/qLinearLayer.py:20-74).

Same constructor signature and forward contract; the weight is packed once at construction
with `reorder_quantize_w4` (production W4 mode, qLinearLayer.py:50) and every forward is
`reorder_quantize_x` + `matmul` (+ bias).  The bias add is fused into the GEMM epilogue with
the same two-rounding semantics as the reference's separate `y + bias`.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import os

from . import mixedgemm

# decode-sized inputs (M <= 8 rows) run quantize + GEMM as one launch; MICROMIX_DECODE_FUSED=0 keeps the two-op path
_DECODE_FUSED = os.environ.get("MICROMIX_DECODE_FUSED", "1") != "0"


class _DecodePlan:
    """Everything about a layer that the C ABI needs and that does not change between calls (weight pointers, split, weight
    mode), validated once.  `run` is the fused decode kernel (one torch.empty + one ctypes call, ~8 us of host time instead of
    ~20); `run_two_op` is reorder_quantize_x + matmul with ONE scratch allocation for the six quantizer outputs instead of six
    (~12 us instead of ~34), which is what bounds the eager throughput for 8 < M < ~512."""
    __slots__ = ("lib", "args", "refs", "n", "k", "split", "wmode", "flags", "device", "index", "benefit", "ws_bytes")

    def __init__(self, layer):
        from . import _lib
        self.lib = _lib.load()
        self.n, self.k = layer.out_features, layer.in_features
        self.split = (layer.p4_num, layer.p6_num, layer.p8_num)
        self.device = layer.BN.device
        self.index = self.device.index
        same = layer.BS.size(1) == layer.p6_num // 4 * 3 and layer.BO.size(1) == layer.p8_num
        self.wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
        self.flags = _lib.MM_ROUND_ONCE if getattr(layer, "rounding", "reference") == "fused" else _lib.MM_ROUND_PER_SEGMENT
        ptr = lambda t: t.data_ptr() if t.numel() else None
        # the tensors themselves are kept (alive, and compared by identity on every call: a layer whose packed weights were
        # replaced gets a new plan instead of stale pointers)
        self.refs = (layer.reorder_index, layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS, layer.SFBO)
        self.args = (ptr(layer.reorder_index), ptr(layer.BN), ptr(layer.BS), ptr(layer.BO), ptr(layer.SFBN), ptr(layer.SFBS),
                     ptr(layer.SFBO))
        self.benefit = {}          # rows -> mm_qlinear_decode_supported(...) == 2
        self.ws_bytes = {}         # rows -> mm_matmul_workspace_bytes(...)

    def matches(self, layer):
        r = self.refs
        fused = getattr(layer, "rounding", "reference") == "fused"
        return (r[1] is layer.BN and r[0] is layer.reorder_index and r[2] is layer.BS and r[3] is layer.BO
                and r[4] is layer.SFBN and r[5] is layer.SFBS and r[6] is layer.SFBO and fused == (self.flags == 1))

    def wins(self, m):
        w = self.benefit.get(m)
        if w is None:
            w = self.benefit[m] = self.lib.mm_qlinear_decode_supported_w(m, self.n, *self.split, self.wmode) == 2
        return w

    def norm_wins(self, m):
        w = self.benefit.get(("norm", m))
        if w is None:
            w = self.benefit[("norm", m)] = self.lib.mm_rmsnorm_qlinear_decode_supported_w(m, self.n, *self.split, self.wmode) == 2
        return w

    def run(self, x2d, bias):
        out = torch.empty((x2d.size(0), self.n), dtype=torch.bfloat16, device=self.device)
        if torch.cuda.current_device() != self.index:
            with torch.cuda.device(self.index):
                return self.run(x2d, bias)
        st = self.lib.mm_qlinear_decode(x2d.data_ptr(), *self.args, x2d.size(0), self.n, *self.split, self.wmode, self.flags,
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        if st:
            from . import _lib
            _lib.check(st, "qlinear_decode")
        return out


    def run_two_op(self, x2d, bias, norm=None):
        """quantize (norm = (weight, eps): rmsnorm_quantize_x instead of reorder_quantize_x) + matmul, the split-K scratch from the plan"""
        m = x2d.size(0)
        kn, ks, ko = self.split
        sizes = (m * (kn // 2), m * (ks // 4 * 3), m * ko, mixedgemm._sf_bytes_x(m, kn), mixedgemm._sf_bytes_x(m, ks),
                 mixedgemm._sf_bytes_x(m, ko))
        offs, total = [], 0
        for sz in sizes:
            offs.append(total)
            total += (sz + 255) & ~255
        from . import _lib
        wflags = self.flags | _lib.MM_WS_TICKETS_ZEROED     # the split-K scratch is mixedgemm.split_workspace (ticket words kept zero)
        ws_bytes = self.ws_bytes.get(m)
        if ws_bytes is None:
            ws_bytes = self.ws_bytes[m] = self.lib.mm_matmul_workspace_bytes(m, self.n, kn, ks, ko, self.wmode, wflags) if m > 32 else 0
        if torch.cuda.current_device() != self.index:
            with torch.cuda.device(self.index):
                return self.run_two_op(x2d, bias, norm)
        # one scratch tensor for the quantizer outputs; it is released at return, which is safe because the caching allocator only
        # hands the block to later work on the same stream.  The split-K workspace is the stream's persistent one.
        scratch = torch.empty((total,), dtype=torch.uint8, device=self.device)
        ws = mixedgemm.split_workspace(self.device, ws_bytes) if ws_bytes else None
        base = scratch.data_ptr()
        q = [base + o if sz else None for o, sz in zip(offs, sizes)]
        out = torch.empty((m, self.n), dtype=torch.bfloat16, device=self.device)
        stream = torch.cuda.current_stream().cuda_stream
        idx, bn, bs, bo, sfbn, sfbs, sfbo = self.args
        if norm is None:
            st = self.lib.mm_reorder_quantize(x2d.data_ptr(), m, self.k, idx, kn, ks, ko, 0, *q, stream)
        else:
            st = self.lib.mm_rmsnorm_quantize(x2d.data_ptr(), norm[0].data_ptr(), float(norm[1]), m, self.k, idx, kn, ks, ko, 0, *q, stream)
        if st == 0:
            st = self.lib.mm_matmul_ws(q[0], bn, q[1], bs, q[2], bo, q[3], sfbn, q[4], sfbs, q[5], sfbo, m, self.n, kn, ks, ko,
                                       self.wmode, wflags, bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                       ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream)
        if st:
            _lib.check(st, "QLinearLayer.forward")
        return out


def _forward(layer, x):
    """shared by QLinearLayer and FusedQLinear: x tensor or pre-quantized tuple -> [M, N] bf16, bsz, q_len"""
    bias = layer.bias
    if isinstance(x, (tuple, list)):   # pre-quantized input (qMixtralLayer.py:292,359,509,517)
        AN, AS, AO, SFAN, SFAS, SFAO, bsz, q_len = x
        if AN.size(1) * 2 != layer.p4_num or AO.size(1) != layer.p8_num:
            raise RuntimeError("pre-quantized input was produced with a different (p4, p6, p8) split")
    else:
        bsz, q_len, k = x.shape
        m = bsz * q_len
        plan = _plan_of(layer)
        if x.dtype is not torch.bfloat16 or x.device != plan.device or k != plan.k:
            raise TypeError(f"input must be a bfloat16 tensor [bsz, q_len, {plan.k}] on {plan.device}")
        if bias is not None and bias.device != x.device:
            bias = bias.to(x.device)
        x2d = x.reshape(m, k).contiguous()
        if m == 0:
            return torch.empty((0, plan.n), dtype=torch.bfloat16, device=plan.device), bsz, q_len
        if _DECODE_FUSED and m <= 8 and plan.wins(m):
            return plan.run(x2d, bias), bsz, q_len
        return plan.run_two_op(x2d, bias), bsz, q_len
    if bias is not None and bias.device != AN.device:
        bias = bias.to(AN.device)
    y = mixedgemm.matmul(AN, layer.BN, AS, layer.BS, AO, layer.BO, SFAN, layer.SFBN, SFAS, layer.SFBS, SFAO, layer.SFBO,
                         bias=bias, rounding=getattr(layer, "rounding", "reference"))
    return y, bsz, q_len


def _plan_of(layer):
    plan = layer.__dict__.get("_decode_plan")
    if plan is None or not plan.matches(layer):
        plan = layer.__dict__["_decode_plan"] = _DecodePlan(layer)
    return plan


def _forward_norm(layer, x, norm_weight, eps):
    """RMSNorm(x; norm_weight, eps) -> layer: the reference's caller pattern `layer(rmsnorm_quantize_x(x, w, eps, idx, p4, p6, p8))`
    (model/qLlamaLayer.py: input_layernorm -> q/k/v, post_attention_layernorm -> gate/up; bindings.cpp:257-303).  At decode sizes where
    it is faster the norm, the quantization and the GEMM are ONE launch (`mixedgemm.rmsnorm_qlinear_decode`); the bytes are the same.
    x: [bsz, q_len, K], or [tokens, K] (the Mixtral caller's 2-D form: bsz comes back as None).  The same checks, the same empty-batch
    answer and the same per-layer plan (split-K workspace included) as `_forward`."""
    if x.dim() == 3:
        bsz, q_len, k = x.shape
    elif x.dim() == 2:
        bsz, (q_len, k) = None, x.shape
    else:
        raise TypeError("input must be [bsz, q_len, K] or [tokens, K]")
    m = q_len if bsz is None else bsz * q_len
    plan = _plan_of(layer)
    if x.dtype is not torch.bfloat16 or x.device != plan.device or k != plan.k:
        raise TypeError(f"input must be a bfloat16 tensor [bsz, q_len, {plan.k}] on {plan.device}")
    if norm_weight.dtype is not torch.bfloat16 or norm_weight.device != plan.device or norm_weight.numel() != plan.k:
        raise TypeError(f"norm_weight must be a bfloat16 tensor [{plan.k}] on {plan.device}")
    if m == 0:
        return torch.empty((0, plan.n), dtype=torch.bfloat16, device=plan.device), bsz, q_len
    x2d = x.reshape(m, k).contiguous()
    norm_weight = norm_weight.contiguous()
    bias = layer.bias
    if bias is not None and bias.device != x.device:
        bias = bias.to(x.device)
    if _DECODE_FUSED and m <= 8 and plan.norm_wins(m):
        split = plan.split
        return mixedgemm.rmsnorm_qlinear_decode(x2d, norm_weight, eps, layer.reorder_index, layer.BN, layer.BS, layer.BO, layer.SFBN, layer.SFBS,
                                                layer.SFBO, *split, bias=bias, rounding=getattr(layer, "rounding", "reference")), bsz, q_len
    return plan.run_two_op(x2d, bias, norm=(norm_weight, eps)), bsz, q_len


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
                 weight_mode: str = "w4", rounding: str = "reference"):
        """Same positional arguments as the reference (qLinearLayer.py:21-27).  Extensions: `weight_mode` "w4" (all-MXFP4 weights,
        the reference's deployment, qLinearLayer.py:50) or "w" (matching precisions); `rounding` "reference" (D rounded through
        bf16 after every segment, as the reference's three chained kernels do) or "fused" (one rounding: more accurate and
        ~1.3 us faster per segment boundary at 4096^3, but no longer the reference's rounding chain)."""
        super().__init__()
        if rounding not in ("reference", "fused"):
            raise ValueError("rounding must be 'reference' or 'fused'")
        self.rounding = rounding
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

    def quantize_input(self, x):
        """x [bsz, q_len, K] -> the 8-tuple (AN, AS, AO, SFAN, SFAS, SFAO, bsz, q_len) that `forward` also accepts.
        Layers fed by the same tensor with the same reorder index (q/k/v, gate/up) can share one quantization, which
        is what the reference's Mixtral caller does by hand (qMixtralLayer.py:289-295)."""
        bsz, q_len, _ = x.shape
        x = x.reshape(bsz * q_len, -1).contiguous()
        return (*mixedgemm.reorder_quantize_x(x, self.reorder_index, self.p4_num, self.p6_num, self.p8_num), bsz, q_len)

    @torch.no_grad()
    def forward(self, x):
        y, bsz, q_len = _forward(self, x)
        # the Mixtral expert caller passes bsz = None with 2-D token batches (qMixtralLayer.py:507-519)
        return y.reshape(bsz, q_len, y.size(-1)) if bsz is not None else y.reshape(q_len, y.size(-1))

    @torch.no_grad()
    def forward_norm(self, x, norm_weight, eps):
        """layer(RMSNorm(x)): x [bsz, q_len, K] bf16 -> [bsz, q_len, N]; see _forward_norm"""
        y, bsz, q_len = _forward_norm(self, x, norm_weight, eps)
        return y.reshape(bsz, q_len, y.size(-1)) if bsz is not None else y.reshape(q_len, y.size(-1))


class FusedQLinear(nn.Module):
    """Several QLinearLayers that read the SAME input (q/k/v, gate/up) as ONE GEMM.

    Output features are independent rows of the packed weights and 128-row tiles of the scale tensors, so the packed
    tensors of layers with identical (reorder_index, p4, p6, p8) concatenate along N (every out_features must be a multiple
    of 128 so that the scale-factor row tiles stay aligned).  `forward` returns the per-layer outputs (views of one [M, sum N]
    result): one quantization and one launch instead of len(layers) of each -- at decode sizes that is the difference
    between three ~6 us launches and one.  Not in the reference; results are bit-identical to the separate layers.
    """

    def __init__(self, layers):
        super().__init__()
        layers = list(layers)
        first = layers[0]
        for l in layers[1:]:
            if (l.p4_num, l.p6_num, l.p8_num) != (first.p4_num, first.p6_num, first.p8_num) or \
                    not torch.equal(l.reorder_index, first.reorder_index) or l.BS.size(1) != first.BS.size(1) or \
                    getattr(l, "rounding", "reference") != getattr(first, "rounding", "reference"):
                raise ValueError("fused layers must share reorder_index, the (p4, p6, p8) split, the weight mode and the rounding mode")
        if any(l.out_features % 128 for l in layers[:-1]):
            raise ValueError("out_features of every fused layer but the last must be a multiple of 128")
        self.in_features = first.in_features
        self.splits = [l.out_features for l in layers]
        self.out_features = sum(self.splits)
        self.p4_num, self.p6_num, self.p8_num = first.p4_num, first.p6_num, first.p8_num
        self.rounding = getattr(first, "rounding", "reference")
        self.reorder_index = first.reorder_index
        for name in ("BN", "BS", "BO", "SFBN", "SFBS", "SFBO"):
            setattr(self, name, torch.cat([getattr(l, name) for l in layers], dim=0).contiguous())
        if any(l.bias is not None for l in layers):
            self.register_buffer("bias", torch.cat([l.bias if l.bias is not None else
                                                    torch.zeros(l.out_features, dtype=torch.bfloat16, device=first.BN.device)
                                                    for l in layers]))
            self._has_bias = [l.bias is not None for l in layers]
        else:
            self.bias = None

    quantize_input = QLinearLayer.quantize_input

    @torch.no_grad()
    def forward(self, x):
        y, bsz, q_len = _forward(self, x)
        if bsz is None:
            return tuple(t.reshape(q_len, t.size(-1)) for t in y.split(self.splits, dim=1))
        return tuple(t.reshape(bsz, q_len, t.size(-1)) for t in y.split(self.splits, dim=1))

    @torch.no_grad()
    def forward_norm(self, x, norm_weight, eps):
        """the fused layers on RMSNorm(x) (input_layernorm -> q | k | v as one launch at decode sizes); see _forward_norm"""
        y, bsz, q_len = _forward_norm(self, x, norm_weight, eps)
        if bsz is None:
            return tuple(t.reshape(q_len, t.size(-1)) for t in y.split(self.splits, dim=1))
        return tuple(t.reshape(bsz, q_len, t.size(-1)) for t in y.split(self.splits, dim=1))


class FusedMLP(nn.Module):
    """gate_proj, up_proj, act_fn(gate) * up and down_proj of one decoder MLP (model/qLlamaLayer.py:336-387) in three launches
    (M > 64; at most four below) instead of the reference's five (quantize x once instead of twice; gate + up + silu * up + the quantization for down_proj as ONE
    GEMM launch, `mixedgemm.gate_up_activate`; down_proj).  `gate` and `up` are QLinearLayers over the same input with the same
    reorder index and split (fp4 weights), whose output features are already in down_proj's reordered order -- the reference folds
    that order into gate / up (`out_reorder_index`, qLlamaLayer.py:341,354); `w_down` [H, I] has its columns in that order and
    `down_split` = (p4, p6, p8) of the I intermediate features (packed with downproj_quantize_w4, bindings.cpp:363-387).
    Bit-identical to gate(x), up(x) -> activate_quantize_x -> matmul with the packed down weight."""

    def __init__(self, gate: QLinearLayer, up: QLinearLayer, w_down: torch.Tensor, down_split, rounding: str = "reference"):
        super().__init__()
        if (gate.p4_num, gate.p6_num, gate.p8_num) != (up.p4_num, up.p6_num, up.p8_num) or not torch.equal(gate.reorder_index, up.reorder_index) \
                or gate.out_features != up.out_features or gate.bias is not None or up.bias is not None:
            raise ValueError("gate and up must share input features, reorder index, split and have no bias")
        if gate.BS.size(1) != gate.p6_num // 2 or gate.BO.size(1) != gate.p8_num // 2:
            raise ValueError("the fused MLP needs fp4 weights (weight_mode='w4')")
        self.hidden, self.inter = gate.in_features, gate.out_features
        self.in_split = (gate.p4_num, gate.p6_num, gate.p8_num)
        self.down_split = tuple(int(v) for v in down_split)
        if sum(self.down_split) != self.inter or tuple(w_down.shape) != (self.hidden, self.inter):
            raise ValueError("down_split must sum to the intermediate size and w_down must be [hidden, intermediate]")
        self.rounding = rounding
        self.reorder_index = gate.reorder_index
        packed = mixedgemm.interleave_gate_up((gate.BN, gate.BS, gate.BO, gate.SFBN, gate.SFBS, gate.SFBO),
                                              (up.BN, up.BS, up.BO, up.SFBN, up.SFBS, up.SFBO))
        for name, t in zip(("BN", "BS", "BO", "SFBN", "SFBS", "SFBO"), packed):
            self.register_buffer("GU_" + name, t)
        down = mixedgemm.downproj_quantize_w4(w_down.to(gate.BN.device).to(torch.bfloat16).contiguous(), *self.down_split)
        for name, t in zip(("BN", "BS", "BO", "SFBN", "SFBS", "SFBO"), down):
            self.register_buffer("D_" + name, t)

    @torch.no_grad()
    def forward(self, x, norm_weight=None, eps=1e-5):
        """`norm_weight` given: the MLP of RMSNorm(x) (post_attention_layernorm fused into the quantizer, as the reference's
        rmsnorm_quantize_x caller does); at M = 1 the norm, the quantization and the gate | up GEMM are one launch"""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.hidden).contiguous()
        gu = (self.GU_BN, self.GU_BS, self.GU_BO, self.GU_SFBN, self.GU_SFBS, self.GU_SFBO)
        m = x2.size(0)
        down = (self.D_BN, self.D_BS, self.D_BO, self.D_SFBN, self.D_SFBS, self.D_SFBO)
        # (a wide layer: gate_up_activate is ONE launch at M <= 16 -- then the fused pairs below, whose workgroups all repeat the
        # quantization, only win at M <= 2)
        pair_rows = 2 if mixedgemm.gate_up_activate_decode_supported(1, self.inter, *self.in_split) == 2 else 4
        if norm_weight is not None:
            if m <= 2 and mixedgemm.rmsnorm_gate_up_activate_decode_supported(m, self.inter, *self.in_split) == 2:
                # round 6: norm, quantization, gate | up GEMM, silu * up and the quantization for down_proj in ONE launch; down_proj a plain GEMM
                qh = mixedgemm.rmsnorm_gate_up_activate_decode(x2, norm_weight, eps, self.reorder_index, gu, *self.down_split, rounding=self.rounding)
                y = mixedgemm.matmul(qh[0], self.D_BN, qh[1], self.D_BS, qh[2], self.D_BO, qh[3], self.D_SFBN, qh[4], self.D_SFBS, qh[5],
                                     self.D_SFBO, rounding=self.rounding)
                return y.reshape(*lead, self.hidden)
            if m <= pair_rows and mixedgemm.rmsnorm_qlinear_decode_supported(m, 2 * self.inter, *self.in_split) == 2 \
                    and mixedgemm.down_activate_decode_supported(m, self.hidden, *self.down_split) == 2:
                gub = mixedgemm.rmsnorm_qlinear_decode(x2, norm_weight, eps, self.reorder_index, *gu, *self.in_split, rounding=self.rounding)
                return mixedgemm.down_activate_decode(gub, down, *self.down_split, rounding=self.rounding).reshape(*lead, self.hidden)
            qx = mixedgemm.rmsnorm_quantize_x(x2, norm_weight, eps, self.reorder_index, *self.in_split)
            qh = mixedgemm.gate_up_activate(qx, gu, *self.down_split, rounding=self.rounding)
            y = mixedgemm.matmul(qh[0], self.D_BN, qh[1], self.D_BS, qh[2], self.D_BO, qh[3], self.D_SFBN, qh[4], self.D_SFBS, qh[5],
                                 self.D_SFBO, rounding=self.rounding)
            return y.reshape(*lead, self.hidden)
        if m <= 2 and mixedgemm.gate_up_activate_decode_supported(m, self.inter, *self.in_split) == 2:
            qh = mixedgemm.gate_up_activate_decode(x2, self.reorder_index, gu, *self.down_split, rounding=self.rounding)      # ONE launch (round 6)
            y = mixedgemm.matmul(qh[0], self.D_BN, qh[1], self.D_BS, qh[2], self.D_BO, qh[3], self.D_SFBN, qh[4], self.D_SFBS, qh[5],
                                 self.D_SFBO, rounding=self.rounding)
            return y.reshape(*lead, self.hidden)
        if m <= pair_rows and mixedgemm.qlinear_decode_supported(m, 2 * self.inter, *self.in_split) == 2 \
                and mixedgemm.down_activate_decode_supported(m, self.hidden, *self.down_split) == 2:
            # decode, TWO launches: reorder + quantize + the gate | up GEMM, then down_proj with silu * up + its quantization inside
            # every workgroup (the same bytes as the five-launch form)
            gub = mixedgemm.qlinear_decode(x2, self.reorder_index, *gu, *self.in_split, rounding=self.rounding)
            return mixedgemm.down_activate_decode(gub, down, *self.down_split, rounding=self.rounding).reshape(*lead, self.hidden)
        if pair_rows == 4 and m <= 8 and mixedgemm.qlinear_decode_supported(m, 2 * self.inter, *self.in_split) == 2:
            # decode: reorder + quantize + the gate | up GEMM in one launch, then the activation quantizer (same bytes, one launch fewer)
            qh = mixedgemm.gate_up_activate_decode(x2, self.reorder_index, gu, *self.down_split, rounding=self.rounding)
        else:
            qx = mixedgemm.reorder_quantize_x(x2, self.reorder_index, *self.in_split)
            qh = mixedgemm.gate_up_activate(qx, gu, *self.down_split, rounding=self.rounding)
        y = mixedgemm.matmul(qh[0], self.D_BN, qh[1], self.D_BS, qh[2], self.D_BO, qh[3], self.D_SFBN, qh[4], self.D_SFBS, qh[5],
                             self.D_SFBO, rounding=self.rounding)
        return y.reshape(*lead, self.hidden)
