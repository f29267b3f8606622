"""`mixedgemm` -- drop-in for the reference's pybind11 extension of the same name
(mgemm/src/bindings.cpp:682-742), MX hot path only, backed by libmicromix_hip.so.

Same function names, argument order, keyword names, return tuples, dtypes and shapes:

    matmul(AN, BN, AS, BS, AO, BO, SFAN, SFBN, SFAS, SFBS, SFAO, SFBO) -> Tensor[M, N] bf16
    reorder_quantize_x(X, reorder_index, KN, KS, KO)  -> (XN, XS, XO, SFXN, SFXS, SFXO)
    reorder_quantize_w(W, reorder_index, KN, KS, KO)  -> (WN, WS, WO, SFWN, SFWS, SFWO)
    reorder_quantize_w4(W, reorder_index, KN, KS, KO) -> (WN, WS, WO, SFWN, SFWS, SFWO)

Differences from the reference, all deliberate:
  * kernels are queued on torch's CURRENT stream of the tensors' device (the reference uses
    the legacy default stream and no device guard, reorder.cu:455, w4a4.cu:182);
  * any K that is a multiple of 128 (<= 32768) works, not only the ten compiled-in values
    (bindings.cpp:134-148); bad splits still raise RuntimeError("Value error in run_...");
  * inputs are validated (dtype/device/contiguity) instead of being reinterpreted blindly;
  * `matmul` takes optional keyword-only `bias` and `rounding` arguments (extensions).
    activate_quantize_x(A, B, KN, KS, KO)             -> (XN, XS, XO, SFXN, SFXS, SFXO)   (section 8f rank 1)
    downproj_quantize_w / _w4 (W, KN, KS, KO)         -> (WN, WS, WO, SFWN, SFWS, SFWO)
    rmsnorm_quantize_x(X, W, eps, reorder_index, KN, KS, KO) -> (XN, XS, XO, SFXN, SFXS, SFXO)    (section 8f rank 2)
The remaining exports of the reference module (FlashInfer KV ops) are outside the hot path; they raise
NotImplementedError.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["test_function", "matmul", "gate_up_activate", "interleave_gate_up", "reorder_quantize_x", "reorder_quantize_w", "reorder_quantize_w4", "activate_quantize_x",
           "downproj_quantize_w", "downproj_quantize_w4", "rmsnorm_quantize_x", "qlinear_decode", "qlinear_decode_supported", "matmul_grouped", "reorder_quantize_x_grouped"]


def test_function():
    """bindings.cpp:700: the module's liveness probe (answered by the HIP library, so it also proves the library loads)."""
    return _lib.load().mm_test_function().decode()


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_tensor(t, name, dtype, device=None):
    """Slow path: produces the precise error.  The ops call it only after the cheap combined test failed."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a device tensor (HIP); the MicroMix ops have no CPU path")
    if device is not None and t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _ok(t, dtype, index) -> bool:
    # one short-circuit expression per tensor: ~0.3 us instead of ~0.7 us for the descriptive checks
    return t.dtype is dtype and t.is_cuda and t.get_device() == index and t.is_contiguous()


def _ptr(t):
    return t.data_ptr() if t.numel() else None


def _sf_bytes_x(m, k):   # bindings.cpp:120-123
    return (m // 128 + 1) * 128 * (k // 32)


def _sf_bytes_w(n, k):   # bindings.cpp:170-172 (rows padded to 128)
    return (n + 127) // 128 * 128 * (k // 32)


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (saves ~2 us per call)."""
    __slots__ = ("idx", "prev")

    def __init__(self, idx):
        self.idx = idx
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _quantize(src, reorder_index, KN, KS, KO, mode, what, gather_subset=False):
    lib = _lib.load()
    if not (isinstance(src, torch.Tensor) and isinstance(reorder_index, torch.Tensor) and src.is_cuda
            and _ok(src, torch.bfloat16, src.get_device()) and _ok(reorder_index, torch.int16, src.get_device())):
        _check_tensor(src, "X" if mode == "x" else "W", torch.bfloat16)
        _check_tensor(reorder_index, "reorder_index", torch.int16, src.device)
    if src.dim() != 2:
        raise RuntimeError("input must be 2-D [rows, K]")
    KN, KS, KO = int(KN), int(KS), int(KO)
    rows, K = src.shape
    if reorder_index.numel() != KN + KS + KO:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, what)
    if KN < 0 or KS < 0 or KO < 0 or KN % 128 or KS % 128 or KO % 128 or K % 128:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, what)
    if (KN + KS + KO != K) if not gather_subset else (KN + KS + KO > K or KN + KS + KO == 0):
        _lib.check(_lib.MM_ERR_BAD_SPLIT, what)
    dev = src.device
    w4 = mode == "w4"
    u8 = torch.uint8
    oN = torch.empty((rows, KN // 2), dtype=u8, device=dev)
    oS = torch.empty((rows, KS // 2 if w4 else KS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((rows, KO // 2 if w4 else KO), dtype=u8, device=dev)
    sf_bytes = _sf_bytes_x if mode == "x" else _sf_bytes_w
    sfN = torch.empty((sf_bytes(rows, KN),), dtype=u8, device=dev)
    sfS = torch.empty((sf_bytes(rows, KS),), dtype=u8, device=dev)
    sfO = torch.empty((sf_bytes(rows, KO),), dtype=u8, device=dev)
    with _on_device(dev.index):
        entry = lib.mm_reorder_quantize_gather if gather_subset else lib.mm_reorder_quantize
        st = entry(
            _ptr(src), rows, K, _ptr(reorder_index), KN, KS, KO,
            _lib.MM_QUANT_W4 if w4 else _lib.MM_QUANT_MIXED,
            _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO), _stream_ptr(dev))
    if st:
        _lib.check(st, what)
    return oN, oS, oO, sfN, sfS, sfO


def reorder_quantize_x(X, reorder_index, KN, KS, KO):
    """bindings.cpp:104-151.  X [M,K] bf16 -> XN [M,KN/2], XS [M,3KS/4], XO [M,KO] + 3 SF tensors."""
    return _quantize(X, reorder_index, KN, KS, KO, "x", "reorder_quantize_x")


def reorder_quantize_w(W, reorder_index, KN, KS, KO):
    """bindings.cpp:155-202.  W [N,K] bf16 -> WN [N,KN/2], WS [N,3KS/4], WO [N,KO] + 3 SF tensors."""
    return _quantize(W, reorder_index, KN, KS, KO, "w", "reorder_quantize_w")


def reorder_quantize_w4(W, reorder_index, KN, KS, KO):
    """bindings.cpp:206-253.  W [N,K] bf16 -> WN [N,KN/2], WS [N,KS/2], WO [N,KO/2] + 3 SF tensors."""
    return _quantize(W, reorder_index, KN, KS, KO, "w4", "reorder_quantize_w4")


_SPLIT_WS = {}             # (device index, stream handle) -> workspace tensor; insertion order = age
_SPLIT_WS_MAX = 8          # eager workspaces kept alive at once (8-128 MiB each): the oldest stream's is released beyond that


def split_workspace(dev, nbytes):
    """The split-K scratch for a launch on `dev`'s current stream, its first MM_WS_TICKET_BYTES zero when that launch runs (the C
    ABI's MM_WS_TICKETS_ZEROED contract); the C ABI itself never allocates.

    Eager: one tensor per (device, stream), cleared when it is created and left zero by every launch, grown when a shape needs
    more; consecutive launches of the stream reuse it in stream order.  At most _SPLIT_WS_MAX are kept: dropping a tensor only
    returns its block to the caching allocator's pool of the stream it was allocated on, which hands it to LATER work of that
    stream, so queued launches are not disturbed.

    Under hipGraph capture: a tensor of the capture's own (private pool) with a clearing kernel node for its ticket words in front of the
    launch, on every call.  A captured launch therefore never shares ticket counters with eager work or with another graph --
    whichever stream the graphs are replayed on, in whatever order -- and nothing depends on a clearing that capture recorded
    instead of executing (the first capture used to create, and "zero", the capture stream's workspace)."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    if torch.cuda.is_current_stream_capturing():
        t = torch.empty((max(int(nbytes), _lib.MM_WS_TICKET_BYTES),), dtype=torch.uint8, device=dev)
        with _on_device(dev.index):
            st = _lib.load().mm_matmul_ws_reset(t.data_ptr(), t.numel(), stream)
        if st:
            _lib.check(st, "matmul(workspace)")
        return t
    key = (dev.index, stream)
    t = _SPLIT_WS.get(key)
    if t is None or t.numel() < nbytes:
        size = max(int(nbytes), 8 << 20, 2 * t.numel() if t is not None else 0)
        _SPLIT_WS.pop(key, None)
        t = torch.empty((size,), dtype=torch.uint8, device=dev)
        with _on_device(dev.index):
            st = _lib.load().mm_matmul_ws_reset(t.data_ptr(), t.numel(), stream)
        if st:
            _lib.check(st, "matmul(workspace)")
        _SPLIT_WS[key] = t
        while len(_SPLIT_WS) > _SPLIT_WS_MAX:
            _SPLIT_WS.pop(next(iter(_SPLIT_WS)))
    return t


def matmul(AN, BN, AS, BS, AO, BO, SFAN, SFBN, SFAS, SFBS, SFAO, SFBO, *, bias=None, rounding="reference", out=None,
           split_k=True, out_dtype=torch.bfloat16):
    """bindings.cpp:50-102.  Returns a new [M, N] bf16 tensor.

    Keyword-only extras (not in the reference): `bias` [N] bf16 fused into the epilogue, `rounding` "reference" (bf16 after
    each segment, as the chained reference kernels) or "fused", `out` to write into an existing tensor, `split_k=False`
    to forbid the K-split the library uses for shapes with few output tiles (it needs a scratch tensor; "force" splits
    wherever the shape allows, for tests), `out_dtype=torch.float32` for the unrounded fp32 accumulator (MM_OUT_F32: the partial
    products of a K-sharded tensor-parallel layer; needs rounding="fused" and no bias).

    Shapes are derived exactly as the reference does (bindings.cpp:66-70) and the weight mode
    from `AS.size(1) == BS.size(1) and AO.size(1) == BO.size(1)` (bindings.cpp:74,87).
    """
    lib = _lib.load()
    tensors = (AN, BN, AS, BS, AO, BO, SFAN, SFBN, SFAS, SFBS, SFAO, SFBO)
    dev = AN.device
    index = dev.index
    u8 = torch.uint8
    for t in tensors:
        if not _ok(t, u8, index):
            names = ("AN", "BN", "AS", "BS", "AO", "BO", "SFAN", "SFBN", "SFAS", "SFBS", "SFAO", "SFBO")
            for n, tt in zip(names, tensors):
                _check_tensor(tt, n, u8, dev)
    M, N = AN.size(0), BN.size(0)
    as1, ao1, bs1, bo1 = AS.size(1), AO.size(1), BS.size(1), BO.size(1)
    KN, KS, KO = AN.size(1) * 2, as1 * 4 // 3, ao1
    same = as1 == bs1 and ao1 == bo1
    wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
    if (BN.size(1) != KN // 2 or bs1 != (KS // 4 * 3 if same else KS // 2) or bo1 != (KO if same else KO // 2)
            or BS.size(0) != N or BO.size(0) != N or BN.dim() != 2):
        exp_b = (KN // 2, KS // 4 * 3 if same else KS // 2, KO if same else KO // 2)
        for n, t, w in zip(("BN", "BS", "BO"), (BN, BS, BO), exp_b):
            if t.dim() != 2 or t.size(0) != N or t.size(1) != w:
                raise RuntimeError(f"{n} has shape {tuple(t.shape)}, expected ({N}, {w})")
    if AS.size(0) != M or AO.size(0) != M:
        raise RuntimeError("AS and AO must be [M, bytes]")
    if (SFAN.numel() < _sf_bytes_w(M, KN) or SFAS.numel() < _sf_bytes_w(M, KS) or SFAO.numel() < _sf_bytes_w(M, KO)
            or SFBN.numel() < _sf_bytes_w(N, KN) or SFBS.numel() < _sf_bytes_w(N, KS) or SFBO.numel() < _sf_bytes_w(N, KO)):
        for n, t, need in (("SFAN", SFAN, _sf_bytes_w(M, KN)), ("SFAS", SFAS, _sf_bytes_w(M, KS)),
                           ("SFAO", SFAO, _sf_bytes_w(M, KO)), ("SFBN", SFBN, _sf_bytes_w(N, KN)),
                           ("SFBS", SFBS, _sf_bytes_w(N, KS)), ("SFBO", SFBO, _sf_bytes_w(N, KO))):
            if t.numel() < need:
                raise RuntimeError(f"{n} holds {t.numel()} scale bytes, needs at least {need}")
    if rounding == "reference":
        flags = _lib.MM_ROUND_PER_SEGMENT
    elif rounding == "fused":
        flags = _lib.MM_ROUND_ONCE
    else:
        raise ValueError("rounding must be 'reference' or 'fused'")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("out_dtype must be torch.bfloat16 or torch.float32")
    if out_dtype == torch.float32:
        if rounding != "fused" or bias is not None:
            raise ValueError("out_dtype=torch.float32 (fp32 partial sums) needs rounding='fused' and no bias")
        flags |= _lib.MM_OUT_F32
    if bias is not None:
        if not _ok(bias, torch.bfloat16, index):
            _check_tensor(bias, "bias", torch.bfloat16, dev)
        if bias.numel() != N:
            raise RuntimeError("bias must have N elements")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=dev)
    else:
        if not _ok(out, out_dtype, index):
            _check_tensor(out, "out", out_dtype, dev)
        if out.dim() != 2 or out.size(0) != M or out.size(1) != N:
            raise RuntimeError("out has the wrong shape")
    # the kernels move operands in 16-byte pieces (LDS-DMA, dwordx4 loads): views at odd offsets are rejected, not mis-read
    for t in tensors:
        if t.data_ptr() & 15 and t.numel():
            raise RuntimeError("matmul operands must be 16-byte aligned (got a view at an unaligned offset)")
    if out.data_ptr() & 15:
        raise RuntimeError("out must be 16-byte aligned")
    # shapes with few output tiles split K and need scratch for fp32 partial sums; it comes from torch's caching allocator
    # (stream-ordered, graph-capture safe) because the C ABI never allocates
    ws, ws_bytes = None, 0
    if split_k and M > 32:
        if split_k == "force":
            flags |= _lib.MM_SPLIT_K_ALWAYS
        flags |= _lib.MM_WS_TICKETS_ZEROED
        ws_bytes = lib.mm_matmul_workspace_bytes(M, N, KN, KS, KO, wmode, flags)
        if ws_bytes:
            ws = split_workspace(dev, ws_bytes)
            ws_bytes = ws.numel()
    with _on_device(index):
        st = lib.mm_matmul_ws(_ptr(AN), _ptr(BN), _ptr(AS), _ptr(BS), _ptr(AO), _ptr(BO), _ptr(SFAN), _ptr(SFBN),
                              _ptr(SFAS), _ptr(SFBS), _ptr(SFAO), _ptr(SFBO), M, N, KN, KS, KO, wmode, flags,
                              _ptr(bias) if bias is not None else None, _ptr(out), _ptr(ws) if ws is not None else None,
                              ws_bytes, _stream_ptr(dev))
    if st:
        _lib.check(st, "matmul")
    return out


def interleave_gate_up(gate, up):
    """(BN, BS, BO, SFBN, SFBS, SFBO) of gate_proj and of up_proj -- same shapes, packed with the SAME reorder index and split, fp4
    weights -- as the ONE packed weight of 2 I rows that `gate_up_activate` takes: 128 gate rows alternate with the 128 up rows of the
    same feature indices (one 256-feature GEMM tile then holds everything 128 columns of silu(gate) * up need); the scale tensors,
    whose 128-row tiles are contiguous blocks of (Kseg / 128) * 512 bytes, interleave the same way.  I must be a multiple of 128."""
    n = gate[0].size(0)
    if n % 128 or any(g.shape != u.shape for g, u in zip(gate, up)):
        raise RuntimeError("gate and up must have identical packed shapes and a multiple of 128 output features")
    out = []
    for g, u in zip(gate[:3], up[:3]):
        w = g.size(1)
        out.append(torch.stack((g.reshape(n // 128, 128, w), u.reshape(n // 128, 128, w)), dim=1).reshape(2 * n, w).contiguous())
    for g, u in zip(gate[3:], up[3:]):
        if g.numel() == 0:
            out.append(g.clone())
            continue
        per = g.numel() // (n // 128)
        out.append(torch.stack((g.reshape(n // 128, per), u.reshape(n // 128, per)), dim=1).reshape(-1).contiguous())
    return tuple(out)


def deinterleave_gate_up(packed):
    """inverse of interleave_gate_up: (gate 6-tuple, up 6-tuple), as contiguous copies"""
    n2 = packed[0].size(0)
    n = n2 // 2
    gate, up = [], []
    for t in packed[:3]:
        v = t.reshape(n // 128, 2, 128, t.size(1))
        gate.append(v[:, 0].reshape(n, t.size(1)).contiguous())
        up.append(v[:, 1].reshape(n, t.size(1)).contiguous())
    for t in packed[3:]:
        if t.numel() == 0:
            gate.append(t.clone())
            up.append(t.clone())
            continue
        v = t.reshape(n // 128, 2, -1)
        gate.append(v[:, 0].reshape(-1).contiguous())
        up.append(v[:, 1].reshape(-1).contiguous())
    return tuple(gate), tuple(up)


def gate_up_activate(A, B, DN, DS, DO, *, rounding="reference"):
    """silu(gate_proj(x)) * up_proj(x), quantized for down_proj, as ONE launch for M > 64 (the reference: model/qLlamaLayer.py:377-387
    through HBM; its activate_quantize_x, bindings.cpp:307-334).  A = (AN, AS, AO, SFAN, SFAS, SFAO) = reorder_quantize_x(x, ...);
    B = interleave_gate_up(gate packed, up packed); (DN, DS, DO) = down_proj's split of the I intermediate features in natural
    column order.  Returns (XN, XS, XO, SFXN, SFXS, SFXO), bit-identical to
    `activate_quantize_x(matmul(A, gate), matmul(A, up), DN, DS, DO)`.  Not an export of the reference module."""
    lib = _lib.load()
    dev = A[0].device
    index = dev.index
    u8 = torch.uint8
    for t in (*A, *B):
        if not _ok(t, u8, index):
            _check_tensor(t, "operand", u8, dev)
    M, N2 = A[0].size(0), B[0].size(0)
    KN, KS, KO = A[0].size(1) * 2, A[1].size(1) * 4 // 3, A[2].size(1)
    DN, DS, DO = int(DN), int(DS), int(DO)
    I = N2 // 2
    if N2 % 256 or B[0].size(1) != KN // 2 or B[1].size(1) != KS // 2 or B[2].size(1) != KO // 2 or B[1].size(0) != N2 or B[2].size(0) != N2:
        raise RuntimeError("B must be an interleaved fp4 gate/up weight (interleave_gate_up) matching the activations' split")
    if A[1].size(0) != M or A[2].size(0) != M:
        raise RuntimeError("AS and AO must be [M, bytes]")
    if DN < 0 or DS < 0 or DO < 0 or DN % 128 or DS % 128 or DO % 128 or DN + DS + DO != I:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "activate_quantize_x")
    for n, t, need in (("SFAN", A[3], _sf_bytes_w(M, KN)), ("SFAS", A[4], _sf_bytes_w(M, KS)), ("SFAO", A[5], _sf_bytes_w(M, KO)),
                       ("SFBN", B[3], _sf_bytes_w(N2, KN)), ("SFBS", B[4], _sf_bytes_w(N2, KS)), ("SFBO", B[5], _sf_bytes_w(N2, KO))):
        if t.numel() < need:
            raise RuntimeError(f"{n} holds {t.numel()} scale bytes, needs at least {need}")
    for t in (*A, *B):
        if t.data_ptr() & 15 and t.numel():
            raise RuntimeError("operands must be 16-byte aligned")
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    flags = _lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE
    oN = torch.empty((M, DN // 2), dtype=u8, device=dev)
    oS = torch.empty((M, DS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((M, DO), dtype=u8, device=dev)
    sfN = torch.empty((_sf_bytes_x(M, DN),), dtype=u8, device=dev)
    sfS = torch.empty((_sf_bytes_x(M, DS),), dtype=u8, device=dev)
    sfO = torch.empty((_sf_bytes_x(M, DO),), dtype=u8, device=dev)
    ws_bytes = lib.mm_gate_up_activate_workspace_bytes(M, I)
    ws = torch.empty((ws_bytes,), dtype=u8, device=dev) if ws_bytes else None      # stream-ordered scratch from the caching allocator
    with _on_device(index):
        st = lib.mm_gate_up_activate(_ptr(A[0]), _ptr(B[0]), _ptr(A[1]), _ptr(B[1]), _ptr(A[2]), _ptr(B[2]), _ptr(A[3]), _ptr(B[3]),
                                     _ptr(A[4]), _ptr(B[4]), _ptr(A[5]), _ptr(B[5]), M, I, KN, KS, KO, DN, DS, DO, flags,
                                     _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO),
                                     _ptr(ws) if ws is not None else None, ws_bytes, _stream_ptr(dev))
    if st:
        _lib.check(st, "gate_up_activate")
    return oN, oS, oO, sfN, sfS, sfO


def reorder_quantize_x_grouped(Xs, reorder_indices, KN, KS, KO):
    """`[reorder_quantize_x(X_g, idx_g, KN, KS, KO) for g]` for row sets that share K and the split (the tokens routed to each MoE
    expert, every expert with its own reorder index; qMixtralLayer.py:507-519), 8 groups per launch, bit-identical to the
    separate calls.  Returns a list of 6-tuples.  Not an export of the reference module."""
    lib = _lib.load()
    if len(Xs) != len(reorder_indices):
        raise ValueError("one reorder index per group")
    if not Xs:
        return []
    KN, KS, KO = int(KN), int(KS), int(KO)
    dev = Xs[0].device
    index = dev.index
    K = Xs[0].size(1)
    if KN < 0 or KS < 0 or KO < 0 or KN % 128 or KS % 128 or KO % 128 or KN + KS + KO != K:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "reorder_quantize_x")
    arr = (_lib.MMQuantGroup * len(Xs))()
    res = []
    u8 = torch.uint8
    for g, (X, idx) in enumerate(zip(Xs, reorder_indices)):
        if not (_ok(X, torch.bfloat16, index) and _ok(idx, torch.int16, index)) or X.dim() != 2 or X.size(1) != K or idx.numel() != K:
            _check_tensor(X, f"X[{g}]", torch.bfloat16, dev)
            _check_tensor(idx, f"reorder_index[{g}]", torch.int16, dev)
            raise RuntimeError(f"group {g}: X must be [rows, {K}] and its reorder index must have {K} entries")
        rows = X.size(0)
        out = (torch.empty((rows, KN // 2), dtype=u8, device=dev), torch.empty((rows, KS // 4 * 3), dtype=u8, device=dev),
               torch.empty((rows, KO), dtype=u8, device=dev), torch.empty((_sf_bytes_x(rows, KN),), dtype=u8, device=dev),
               torch.empty((_sf_bytes_x(rows, KS),), dtype=u8, device=dev), torch.empty((_sf_bytes_x(rows, KO),), dtype=u8, device=dev))
        res.append(out)
        e = arr[g]
        e.src_bf16, e.reorder_index = _ptr(X), _ptr(idx)
        e.oN, e.oS, e.oO, e.sfN, e.sfS, e.sfO = (_ptr(t) for t in out)
        e.rows = rows
    with _on_device(index):
        st = lib.mm_reorder_quantize_grouped(arr, len(Xs), K, KN, KS, KO, _lib.MM_QUANT_MIXED, _stream_ptr(dev))
    if st:
        _lib.check(st, "reorder_quantize_x")
    return res


def matmul_grouped(As, Bs, *, biases=None, rounding="reference"):
    """`[matmul(*interleave(A_g, B_g)) for g]` for groups that share N, the (KN, KS, KO) split and the weight mode -- MoE experts
    (the per-expert loop of qMixtralLayer.py:507-519) -- in as few launches as possible: groups of <= 64 token rows run 8 per
    launch.  As[g] = (AN, AS, AO, SFAN, SFAS, SFAO) as returned by reorder_quantize_x, Bs[g] = (BN, BS, BO, SFBN, SFBS, SFBO).
    Returns the list of [M_g, N] bf16 outputs, bit-identical to the separate calls.  Not an export of the reference module."""
    lib = _lib.load()
    if len(As) != len(Bs) or (biases is not None and len(biases) != len(As)):
        raise ValueError("As, Bs (and biases) must have one entry per group")
    if not As:
        return []
    dev = As[0][0].device
    index = dev.index
    u8 = torch.uint8
    BN0, BS0, BO0 = Bs[0][0], Bs[0][1], Bs[0][2]
    N = BN0.size(0)
    KN, KS, KO = As[0][0].size(1) * 2, As[0][1].size(1) * 4 // 3, As[0][2].size(1)
    same = As[0][1].size(1) == BS0.size(1) and As[0][2].size(1) == BO0.size(1)
    wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
    wb = (KN // 2, KS // 4 * 3 if same else KS // 2, KO if same else KO // 2)
    flags = _lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    arr = (_lib.MMGroup * len(As))()
    outs = []
    for g, (A, B) in enumerate(zip(As, Bs)):
        for t in (*A, *B):
            if not _ok(t, u8, index):
                _check_tensor(t, f"group {g} operand", u8, dev)
        M = A[0].size(0)
        if (A[0].size(1) * 2, A[1].size(1) * 4 // 3, A[2].size(1)) != (KN, KS, KO) or A[1].size(0) != M or A[2].size(0) != M:
            raise RuntimeError(f"group {g}: activation segments do not match the split of group 0")
        if tuple(t.size(0) for t in B[:3]) != (N, N, N) or tuple(t.size(1) for t in B[:3]) != wb:
            raise RuntimeError(f"group {g}: packed weights do not match N / split / weight mode of group 0")
        if (A[3].numel() < _sf_bytes_w(M, KN) or A[4].numel() < _sf_bytes_w(M, KS) or A[5].numel() < _sf_bytes_w(M, KO)
                or B[3].numel() < _sf_bytes_w(N, KN) or B[4].numel() < _sf_bytes_w(N, KS) or B[5].numel() < _sf_bytes_w(N, KO)):
            raise RuntimeError(f"group {g}: a scale tensor is too small")
        bias = biases[g] if biases is not None else None
        if bias is not None and (not _ok(bias, torch.bfloat16, index) or bias.numel() != N):
            raise RuntimeError(f"group {g}: bias must be a bfloat16 tensor with N elements")
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        outs.append(out)
        e = arr[g]
        e.AN, e.AS, e.AO, e.SFAN, e.SFAS, e.SFAO = (_ptr(t) for t in A)
        e.BN, e.BS, e.BO, e.SFBN, e.SFBS, e.SFBO = (_ptr(t) for t in B)
        e.bias_bf16 = _ptr(bias) if bias is not None else None
        e.D = _ptr(out)
        e.M = M
    with _on_device(index):
        st = lib.mm_matmul_grouped(arr, len(As), N, KN, KS, KO, wmode, flags, _stream_ptr(dev))
    if st:
        _lib.check(st, "matmul_grouped")
    return outs


def _wmode_of(weight_mode):
    if weight_mode not in ("w4", "w"):
        raise ValueError("weight_mode must be 'w4' (fp4 weights, reorder_quantize_w4) or 'w' (matching precision, reorder_quantize_w)")
    return _lib.MM_W_FP4 if weight_mode == "w4" else _lib.MM_W_MATCH


def qlinear_decode_supported(M, N, KN, KS, KO, weight_mode="w4"):
    """0: `qlinear_decode` cannot run this shape (needs 1 <= M <= 8 and the quantized rows in LDS); 1: it can; 2: it can and is
    expected to be faster than reorder_quantize_x + matmul.  `weight_mode`: the packing of the weights the call will get ("w4": the
    production mode; its ring in LDS is the smaller one, so longer K fit: mm_qlinear_decode_supported_w)."""
    return int(_lib.load().mm_qlinear_decode_supported_w(int(M), int(N), int(KN), int(KS), int(KO), _wmode_of(weight_mode)))


def gate_up_activate_decode(X, reorder_index, B, DN, DS, DO, *, rounding="reference"):
    """`gate_up_activate(reorder_quantize_x(X, reorder_index, KN, KS, KO), B, DN, DS, DO)` for decode-sized batches in TWO launches
    instead of three: reorder + quantize + the gate | up GEMM as one launch (`qlinear_decode` on the interleaved weight B), then
    silu(gate) * up + the quantization for down_proj.  X [M, K] bf16 with M <= 8 and `qlinear_decode_supported(M, 2 I, KN, KS, KO)`;
    (KN, KS, KO) are read off B.  Same bytes as the three-launch form.  Not an export of the reference module."""
    lib = _lib.load()
    dev = X.device
    index = dev.index
    if not (X.is_cuda and _ok(X, torch.bfloat16, index) and _ok(reorder_index, torch.int16, index)):
        _check_tensor(X, "X", torch.bfloat16)
        _check_tensor(reorder_index, "reorder_index", torch.int16, dev)
    for t in B:
        if not _ok(t, torch.uint8, index):
            _check_tensor(t, "operand", torch.uint8, dev)
    M, K = X.shape
    N2 = B[0].size(0)
    KN, KS, KO = B[0].size(1) * 2, B[1].size(1) * 2, B[2].size(1) * 2
    I = N2 // 2
    DN, DS, DO = int(DN), int(DS), int(DO)
    if N2 % 256 or K != KN + KS + KO or reorder_index.numel() != K or B[1].size(0) != N2 or B[2].size(0) != N2:
        raise RuntimeError("B must be an interleaved fp4 gate/up weight (interleave_gate_up) whose split adds up to X's columns")
    if DN < 0 or DS < 0 or DO < 0 or DN % 128 or DS % 128 or DO % 128 or DN + DS + DO != I:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "activate_quantize_x")
    for n, t, need in (("SFBN", B[3], _sf_bytes_w(N2, KN)), ("SFBS", B[4], _sf_bytes_w(N2, KS)), ("SFBO", B[5], _sf_bytes_w(N2, KO))):
        if t.numel() < need:
            raise RuntimeError(f"{n} holds {t.numel()} scale bytes, needs at least {need}")
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    flags = _lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE
    u8 = torch.uint8
    oN = torch.empty((M, DN // 2), dtype=u8, device=dev)
    oS = torch.empty((M, DS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((M, DO), dtype=u8, device=dev)
    sfN = torch.empty((_sf_bytes_x(M, DN),), dtype=u8, device=dev)
    sfS = torch.empty((_sf_bytes_x(M, DS),), dtype=u8, device=dev)
    sfO = torch.empty((_sf_bytes_x(M, DO),), dtype=u8, device=dev)
    ws = torch.empty((M * N2 * 2,), dtype=u8, device=dev)      # stream-ordered scratch from the caching allocator
    with _on_device(index):
        st = lib.mm_gate_up_activate_decode(_ptr(X), _ptr(reorder_index), _ptr(B[0]), _ptr(B[1]), _ptr(B[2]), _ptr(B[3]), _ptr(B[4]), _ptr(B[5]),
                                            M, I, KN, KS, KO, DN, DS, DO, flags, _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO),
                                            _ptr(ws), ws.numel(), _stream_ptr(dev))
    if st:
        _lib.check(st, "gate_up_activate_decode")
    return oN, oS, oO, sfN, sfS, sfO


def rmsnorm_gate_up_activate_decode_supported(M, I, KN, KS, KO):
    """0: cannot run; 1: runs; 2: ONE launch with everything inside and expected to be the fastest form of the MLP's first half (M <= 2 on
    a wide layer; mm_rmsnorm_gate_up_activate_decode_supported)"""
    return int(_lib.load().mm_rmsnorm_gate_up_activate_decode_supported(int(M), int(I), int(KN), int(KS), int(KO)))


def gate_up_activate_decode_supported(M, I, KN, KS, KO):
    """the same for `gate_up_activate_decode` (no norm)"""
    return int(_lib.load().mm_gate_up_activate_decode_supported(int(M), int(I), int(KN), int(KS), int(KO)))


def rmsnorm_gate_up_activate_decode(X, norm_weight, eps, reorder_index, B, DN, DS, DO, *, rounding="reference", integer_round=True):
    """`gate_up_activate(rmsnorm_quantize_x(X, norm_weight, eps, reorder_index, KN, KS, KO), B, DN, DS, DO)` for decode-sized batches: the
    post-attention RMSNorm, the quantization of x, the gate | up GEMM on the interleaved weight B, silu(gate) * up and the quantization for
    down_proj -- ONE launch on a wide layer at M <= 4, two otherwise (`rmsnorm_gate_up_activate_decode_supported(...) == 2`: where it is the
    fastest form).  Returns
    down_proj's activation operands (oN, oS, oO, sfN, sfS, sfO): `matmul` them with the packed down_proj weight.  Same bytes as the
    three-op form.  Not an export of the reference module."""
    lib = _lib.load()
    dev = X.device
    index = dev.index
    if not (X.is_cuda and _ok(X, torch.bfloat16, index) and _ok(reorder_index, torch.int16, index) and _ok(norm_weight, torch.bfloat16, index)):
        _check_tensor(X, "X", torch.bfloat16)
        _check_tensor(norm_weight, "norm_weight", torch.bfloat16, dev)
        _check_tensor(reorder_index, "reorder_index", torch.int16, dev)
    for t in B:
        if not _ok(t, torch.uint8, index):
            _check_tensor(t, "operand", torch.uint8, dev)
    M, K = X.shape
    N2 = B[0].size(0)
    KN, KS, KO = B[0].size(1) * 2, B[1].size(1) * 2, B[2].size(1) * 2
    I = N2 // 2
    DN, DS, DO = int(DN), int(DS), int(DO)
    if N2 % 256 or K != KN + KS + KO or reorder_index.numel() != K or norm_weight.numel() != K or B[1].size(0) != N2 or B[2].size(0) != N2:
        raise RuntimeError("B must be an interleaved fp4 gate/up weight (interleave_gate_up) whose split adds up to X's columns")
    if DN < 0 or DS < 0 or DO < 0 or DN % 128 or DS % 128 or DO % 128 or DN + DS + DO != I:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "activate_quantize_x")
    for n, t, need in (("SFBN", B[3], _sf_bytes_w(N2, KN)), ("SFBS", B[4], _sf_bytes_w(N2, KS)), ("SFBO", B[5], _sf_bytes_w(N2, KO))):
        if t.numel() < need:
            raise RuntimeError(f"{n} holds {t.numel()} scale bytes, needs at least {need}")
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    flags = (_lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE) | (0 if integer_round else _lib.MM_NORM_NO_INTEGER_ROUND)
    u8 = torch.uint8
    oN = torch.empty((M, DN // 2), dtype=u8, device=dev)
    oS = torch.empty((M, DS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((M, DO), dtype=u8, device=dev)
    sfN = torch.empty((_sf_bytes_x(M, DN),), dtype=u8, device=dev)
    sfS = torch.empty((_sf_bytes_x(M, DS),), dtype=u8, device=dev)
    sfO = torch.empty((_sf_bytes_x(M, DO),), dtype=u8, device=dev)
    ws = torch.empty((M * N2 * 2,), dtype=u8, device=dev)      # (scratch of the two-launch form; stream-ordered, from the caching allocator)
    with _on_device(index):
        st = lib.mm_rmsnorm_gate_up_activate_decode(_ptr(X), _ptr(norm_weight), float(eps), _ptr(reorder_index), _ptr(B[0]), _ptr(B[1]), _ptr(B[2]),
                                                    _ptr(B[3]), _ptr(B[4]), _ptr(B[5]), M, I, KN, KS, KO, DN, DS, DO, flags, _ptr(oN), _ptr(oS),
                                                    _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO), _ptr(ws), ws.numel(), _stream_ptr(dev))
    if st:
        _lib.check(st, "rmsnorm_gate_up_activate_decode")
    return oN, oS, oO, sfN, sfS, sfO


def down_activate_decode_supported(M, N, DN, DS, DO, weight_mode="w4"):
    """0: cannot run; 1: runs; 2: runs and is expected to beat activate_quantize_x + matmul (mm_down_activate_decode_supported_w)"""
    return int(_lib.load().mm_down_activate_decode_supported_w(int(M), int(N), int(DN), int(DS), int(DO), _wmode_of(weight_mode)))


def down_activate_decode(GU, B, DN, DS, DO, *, bias=None, rounding="reference"):
    """down_proj(act_fn(gate) * up) for decode-sized batches in ONE launch: GU [M, 2 I] bf16 holds 128 gate columns alternating with the
    128 up columns of the same indices (what `qlinear_decode` on an `interleave_gate_up` weight returns); B = down_proj packed with
    `downproj_quantize_w4` / `_w`; (DN, DS, DO) = its split of the I intermediate features.  Bit-identical to
    `matmul(activate_quantize_x(gate, up, DN, DS, DO), B)`.  M <= 4.  Not an export of the reference module."""
    lib = _lib.load()
    dev = GU.device
    index = dev.index
    if not (GU.is_cuda and _ok(GU, torch.bfloat16, index)):
        _check_tensor(GU, "GU", torch.bfloat16)
    for t in B:
        if not _ok(t, torch.uint8, index):
            _check_tensor(t, "operand", torch.uint8, dev)
    DN, DS, DO = int(DN), int(DS), int(DO)
    M, I2 = GU.shape
    I = DN + DS + DO
    N = B[0].size(0)
    if I2 != 2 * I or I % 128 or DN % 128 or DS % 128 or DO % 128:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "activate_quantize_x")
    same = B[1].size(1) == DS // 4 * 3 and B[2].size(1) == DO
    w4 = B[1].size(1) == DS // 2 and B[2].size(1) == DO // 2
    if B[0].size(1) != DN // 2 or not (same or w4) or B[1].size(0) != N or B[2].size(0) != N:
        raise RuntimeError("packed weights do not match (DN, DS, DO)")
    wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
    if B[3].numel() < _sf_bytes_w(N, DN) or B[4].numel() < _sf_bytes_w(N, DS) or B[5].numel() < _sf_bytes_w(N, DO):
        raise RuntimeError("weight scale tensors are too small")
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    flags = _lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE
    if bias is not None and (not _ok(bias, torch.bfloat16, index) or bias.numel() != N):
        _check_tensor(bias, "bias", torch.bfloat16, dev)
        raise RuntimeError("bias must have N elements")
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    with _on_device(index):
        st = lib.mm_down_activate_decode(_ptr(GU), _ptr(B[0]), _ptr(B[1]), _ptr(B[2]), _ptr(B[3]), _ptr(B[4]), _ptr(B[5]), M, N, DN, DS, DO, wmode,
                                         flags, _ptr(bias) if bias is not None else None, _ptr(out), _stream_ptr(dev))
    if st:
        _lib.check(st, "down_activate_decode")
    return out


def qlinear_decode(X, reorder_index, BN, BS, BO, SFBN, SFBS, SFBO, KN, KS, KO, *, bias=None, rounding="reference", out=None):
    """reorder_quantize_x + matmul (+ bias) of `QLinearLayer.forward` (qLinearLayer.py:58-74) as ONE launch for M <= 8 rows.

    Not an export of the reference module: it is what the reference's forward computes, fused for decode, and bit-identical
    to the two-op path.  X [M, K] bf16; B / SFB are the layer's packed weights; returns [M, N] bf16.
    """
    lib = _lib.load()
    dev = X.device
    index = dev.index
    if not (X.is_cuda and _ok(X, torch.bfloat16, index) and _ok(reorder_index, torch.int16, index)):
        _check_tensor(X, "X", torch.bfloat16)
        _check_tensor(reorder_index, "reorder_index", torch.int16, dev)
    for n, t in (("BN", BN), ("BS", BS), ("BO", BO), ("SFBN", SFBN), ("SFBS", SFBS), ("SFBO", SFBO)):
        if not _ok(t, torch.uint8, index):
            _check_tensor(t, n, torch.uint8, dev)
    KN, KS, KO = int(KN), int(KS), int(KO)
    M, K = X.shape
    N = BN.size(0)
    if K != KN + KS + KO or reorder_index.numel() != K:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "reorder_quantize_x")
    same = BS.size(1) == KS // 4 * 3 and BO.size(1) == KO
    w4 = BS.size(1) == KS // 2 and BO.size(1) == KO // 2
    if BN.size(1) != KN // 2 or not (same or w4) or BS.size(0) != N or BO.size(0) != N:
        raise RuntimeError("packed weights do not match (KN, KS, KO)")
    wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
    if (SFBN.numel() < _sf_bytes_w(N, KN) or SFBS.numel() < _sf_bytes_w(N, KS) or SFBO.numel() < _sf_bytes_w(N, KO)):
        raise RuntimeError("weight scale tensors are too small")
    flags = _lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    if bias is not None and (not _ok(bias, torch.bfloat16, index) or bias.numel() != N):
        _check_tensor(bias, "bias", torch.bfloat16, dev)
        raise RuntimeError("bias must have N elements")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    with _on_device(index):
        st = lib.mm_qlinear_decode(_ptr(X), _ptr(reorder_index), _ptr(BN), _ptr(BS), _ptr(BO), _ptr(SFBN), _ptr(SFBS), _ptr(SFBO),
                                   M, N, KN, KS, KO, wmode, flags, _ptr(bias) if bias is not None else None, _ptr(out),
                                   _stream_ptr(dev))
    if st:
        _lib.check(st, "qlinear_decode")
    return out


def rmsnorm_qlinear_decode_supported(M, N, KN, KS, KO, weight_mode="w4"):
    """0 / 1 / 2 as qlinear_decode_supported, for the launch with the RMSNorm inside (K <= 8192)"""
    return int(_lib.load().mm_rmsnorm_qlinear_decode_supported_w(int(M), int(N), int(KN), int(KS), int(KO), _wmode_of(weight_mode)))


def rmsnorm_qlinear_decode(X, norm_weight, eps, reorder_index, BN, BS, BO, SFBN, SFBS, SFBO, KN, KS, KO, *, bias=None, rounding="reference",
                           integer_round=True, out=None):
    """rmsnorm_quantize_x + matmul (+ bias) as ONE launch for M <= 8 rows: what a decoder layer of the reference runs in front of
    q/k/v and gate/up (qLlamaLayer.py: input_layernorm / post_attention_layernorm fused into the quantizer, rmsnorm.cu:95-352, then
    qLinearLayer.py:58-74).  Bit-identical to `rmsnorm_quantize_x` followed by `matmul`.  X [M, K] bf16, norm_weight [K] bf16."""
    lib = _lib.load()
    dev = X.device
    index = dev.index
    if not (X.is_cuda and _ok(X, torch.bfloat16, index) and _ok(reorder_index, torch.int16, index) and _ok(norm_weight, torch.bfloat16, index)):
        _check_tensor(X, "X", torch.bfloat16)
        _check_tensor(norm_weight, "norm_weight", torch.bfloat16, dev)
        _check_tensor(reorder_index, "reorder_index", torch.int16, dev)
    for n, t in (("BN", BN), ("BS", BS), ("BO", BO), ("SFBN", SFBN), ("SFBS", SFBS), ("SFBO", SFBO)):
        if not _ok(t, torch.uint8, index):
            _check_tensor(t, n, torch.uint8, dev)
    KN, KS, KO = int(KN), int(KS), int(KO)
    M, K = X.shape
    N = BN.size(0)
    if K != KN + KS + KO or reorder_index.numel() != K or norm_weight.numel() != K:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "rmsnorm_quantize_x")
    same = BS.size(1) == KS // 4 * 3 and BO.size(1) == KO
    w4 = BS.size(1) == KS // 2 and BO.size(1) == KO // 2
    if BN.size(1) != KN // 2 or not (same or w4) or BS.size(0) != N or BO.size(0) != N:
        raise RuntimeError("packed weights do not match (KN, KS, KO)")
    wmode = _lib.MM_W_MATCH if same else _lib.MM_W_FP4
    if (SFBN.numel() < _sf_bytes_w(N, KN) or SFBS.numel() < _sf_bytes_w(N, KS) or SFBO.numel() < _sf_bytes_w(N, KO)):
        raise RuntimeError("weight scale tensors are too small")
    if rounding not in ("reference", "fused"):
        raise ValueError("rounding must be 'reference' or 'fused'")
    flags = (_lib.MM_ROUND_PER_SEGMENT if rounding == "reference" else _lib.MM_ROUND_ONCE) | (0 if integer_round else _lib.MM_NORM_NO_INTEGER_ROUND)
    if bias is not None and (not _ok(bias, torch.bfloat16, index) or bias.numel() != N):
        _check_tensor(bias, "bias", torch.bfloat16, dev)
        raise RuntimeError("bias must have N elements")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    with _on_device(index):
        st = lib.mm_rmsnorm_qlinear_decode(_ptr(X), _ptr(norm_weight), float(eps), _ptr(reorder_index), _ptr(BN), _ptr(BS), _ptr(BO), _ptr(SFBN),
                                           _ptr(SFBS), _ptr(SFBO), M, N, KN, KS, KO, wmode, flags, _ptr(bias) if bias is not None else None,
                                           _ptr(out), _stream_ptr(dev))
    if st:
        _lib.check(st, "rmsnorm_qlinear_decode")
    return out


def _direct(src_a, src_b, KN, KS, KO, mode, what):
    """shared body of activate_quantize_x / downproj_quantize_w / downproj_quantize_w4 (bindings.cpp:307-387)."""
    lib = _lib.load()
    srcs = (src_a,) if src_b is None else (src_a, src_b)
    for t in srcs:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and _ok(t, torch.bfloat16, src_a.get_device())):
            _check_tensor(t, "input", torch.bfloat16, src_a.device if isinstance(src_a, torch.Tensor) and src_a.is_cuda else None)
    KN, KS, KO = int(KN), int(KS), int(KO)
    if src_a.dim() != 2 or (src_b is not None and tuple(src_b.shape) != tuple(src_a.shape)):
        raise RuntimeError("inputs must be 2-D [rows, K] of equal shape")
    rows, K = src_a.shape
    if KN < 0 or KS < 0 or KO < 0 or KN % 128 or KS % 128 or KO % 128 or KN + KS + KO != K:
        _lib.check(_lib.MM_ERR_BAD_SPLIT, what)
    dev = src_a.device
    u8 = torch.uint8
    w4 = mode == "w4"
    oN = torch.empty((rows, KN // 2), dtype=u8, device=dev)
    oS = torch.empty((rows, KS // 2 if w4 else KS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((rows, KO // 2 if w4 else KO), dtype=u8, device=dev)
    sfN = torch.empty((_sf_bytes_x(rows, KN),), dtype=u8, device=dev)    # (rows/128+1)*128 rows for all three ops
    sfS = torch.empty((_sf_bytes_x(rows, KS),), dtype=u8, device=dev)
    sfO = torch.empty((_sf_bytes_x(rows, KO),), dtype=u8, device=dev)
    with _on_device(dev.index):
        if src_b is not None:
            st = lib.mm_activate_quantize(_ptr(src_a), _ptr(src_b), rows, KN, KS, KO, _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN),
                                          _ptr(sfS), _ptr(sfO), _stream_ptr(dev))
        else:
            st = lib.mm_downproj_quantize(_ptr(src_a), rows, KN, KS, KO, _lib.MM_QUANT_W4 if w4 else _lib.MM_QUANT_MIXED,
                                          _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO), _stream_ptr(dev))
    if st:
        _lib.check(st, what)
    return oN, oS, oO, sfN, sfS, sfO


def activate_quantize_x(A, B, KN, KS, KO):
    """bindings.cpp:307-335.  silu(A) * B, natural column order -> (XN, XS, XO, SFXN, SFXS, SFXO)."""
    return _direct(A, B, KN, KS, KO, "x", "activate_quantize_x")


def downproj_quantize_w(W, KN, KS, KO):
    """bindings.cpp:336-362.  W [N,K] in natural column order -> fp4 | fp6 | fp8 segments."""
    return _direct(W, None, KN, KS, KO, "w", "downproj_quantize_w")


def downproj_quantize_w4(W, KN, KS, KO):
    """bindings.cpp:363-387.  W [N,K] in natural column order -> three fp4 segments."""
    return _direct(W, None, KN, KS, KO, "w4", "downproj_quantize_w4")


def _not_on_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"mixedgemm.{name} is outside the MX hot path built here (SURVEY.md section 8f)")
    f.__name__ = name
    return f


def rmsnorm_quantize_x(X, W, eps, reorder_index, KN, KS, KO, *, integer_round=True):
    """bindings.cpp:257-303 / rmsnorm.cu:95-312.  RMSNorm(X; W, eps) -> reorder -> mixed quantize, one kernel.

    X [M,K] bf16, W [K] bf16 norm weight -> (XN [M,KN/2], XS [M,3KS/4], XO [M,KO], SFXN, SFXS, SFXO), the tuple
    `QLinearLayer.forward` accepts in place of a tensor, so q/k/v (or gate/up) share one quantization.
    `integer_round=True` (default) reproduces the reference, which rounds the scaled value to an integer before the element
    conversion (rmsnorm.cu:262-267); `False` drops that step.
    """
    lib = _lib.load()
    if not (isinstance(X, torch.Tensor) and X.is_cuda and isinstance(W, torch.Tensor) and isinstance(reorder_index, torch.Tensor)
            and _ok(X, torch.bfloat16, X.get_device()) and _ok(W, torch.bfloat16, X.get_device())
            and _ok(reorder_index, torch.int16, X.get_device())):
        _check_tensor(X, "X", torch.bfloat16)
        _check_tensor(W, "W", torch.bfloat16, X.device)
        _check_tensor(reorder_index, "reorder_index", torch.int16, X.device)
    if X.dim() != 2:
        raise RuntimeError("X must be 2-D [rows, K]")
    KN, KS, KO = int(KN), int(KS), int(KO)
    rows, K = X.shape
    if (KN < 0 or KS < 0 or KO < 0 or KN % 128 or KS % 128 or KO % 128 or KN + KS + KO != K or reorder_index.numel() != K
            or W.numel() != K):
        _lib.check(_lib.MM_ERR_BAD_SPLIT, "rmsnorm_bf16_mixed")    # bindings.cpp:298 "Value error in run_rmsnorm_bf16_mixed"
    dev, u8 = X.device, torch.uint8
    oN = torch.empty((rows, KN // 2), dtype=u8, device=dev)
    oS = torch.empty((rows, KS // 4 * 3), dtype=u8, device=dev)
    oO = torch.empty((rows, KO), dtype=u8, device=dev)
    sfN = torch.empty((_sf_bytes_x(rows, KN),), dtype=u8, device=dev)
    sfS = torch.empty((_sf_bytes_x(rows, KS),), dtype=u8, device=dev)
    sfO = torch.empty((_sf_bytes_x(rows, KO),), dtype=u8, device=dev)
    with _on_device(dev.index):
        st = lib.mm_rmsnorm_quantize(_ptr(X), _ptr(W), float(eps), rows, K, _ptr(reorder_index), KN, KS, KO,
                                     _lib.MM_RMS_REFERENCE if integer_round else _lib.MM_RMS_NO_INTEGER_ROUND,
                                     _ptr(oN), _ptr(oS), _ptr(oO), _ptr(sfN), _ptr(sfS), _ptr(sfO), _stream_ptr(dev))
    if st:
        _lib.check(st, "rmsnorm_bf16_mixed")
    return oN, oS, oO, sfN, sfS, sfO


for _n in ("batch_decode_i4", "batch_decode_f16", "init_kv_i4", "init_kv_f16", "append_kv_i4", "append_kv_f16"):
    globals()[_n] = _not_on_path(_n)
