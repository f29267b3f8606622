"""GPU tests that pin the oracle and the kernel's layout assumptions against the CDNA4 hardware:
the scaled-MFMA operand / scale / accumulator register layouts, and AMD's own MX converter
instructions (an independent implementation of the OCP encodings)."""
import numpy as np
import pytest

import hw_layout as hl
from conftest import t_from_bits
from micromix_amd import _lib
from oracle import mx_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", (32, 16))
@pytest.mark.parametrize("el_a", hl.ELS)
@pytest.mark.parametrize("el_b", hl.ELS)
def test_scaled_mfma_register_layout(dev, shape, el_a, el_b):
    import torch
    lib = _lib.load_diag()
    rng = np.random.default_rng(shape * 100 + hl.ELS.index(el_a) * 10 + hl.ELS.index(el_b))
    # fp4 x fp4 / fp4 x fp6 products are summed exactly by the hardware; anything with fp8 or
    # fp6 x fp6 goes through a limited-precision adder tree (documented in tests/gemm_check.py)
    exact = {el_a, el_b} <= {"fp4", "fp6"} and (el_a, el_b) != ("fp6", "fp6")
    tol = 1e-6 if exact else 5e-4
    for opsel in range(4):
        for _ in range(3):
            err, _, _ = hl.run_case(lib, torch, dev, rng, shape, el_a, el_b, opsel)
            assert err < tol, (shape, el_a, el_b, opsel, err)


@pytest.mark.parametrize("el", hl.ELS)
def test_oracle_encoders_match_hardware_converters(dev, el):
    """v_cvt_scalef32_pk_{fp4,fp8}_bf16 / v_cvt_scalef32_pk32_bf6_bf16 (dst = cvt(src / scale)) produce
    exactly the oracle's codes for every finite bf16 whose scaled value is in range."""
    import torch
    lib = _lib.load_diag()
    allb = np.arange(65536, dtype=np.uint16)
    src = allb[np.isfinite(o.bf16_to_f32(allb))]
    src = src[: len(src) // 32 * 32]
    tsrc = t_from_bits(src, dev)
    x = o.bf16_to_f32(src).astype(np.float64)
    fm = o.FORMATS[el]["fmax"]
    # -127: the scale operand is read as E8M0 (exponent field only), so the bit pattern 0 -- and any fp32 denormal -- means 2^-127;
    # the quantizers rely on it for blocks whose absmax is below FMAX * 2^-127 (mx_group_convert.h: convert_group)
    for e, scale in ((-20, None), (-3, None), (0, None), (2, None), (17, None), (-127, 0.0), (-127, 2.0 ** -127), (-127, 2.0 ** -130)):
        out = torch.zeros(len(src), dtype=torch.uint8, device=dev)
        assert lib.mm_diag_hw_convert(tsrc.data_ptr(), len(src), float(2.0 ** e) if scale is None else scale, hl.ELS.index(el), out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        scaled = x / 2.0 ** e
        inr = np.abs(scaled) <= fm
        want = o.encode(np.clip(scaled, -2 * fm, 2 * fm).astype(np.float32), el)
        got = out.cpu().numpy()
        assert inr.sum() > (1000 if e > -127 else 500) and np.array_equal(got[inr], want[inr]), (el, e, scale)
