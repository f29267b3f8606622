"""One-hot probe: where does byte e of lane (h*R + 0) of an fp8 operand sit in K, relative to the (verified) fp4 order?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from micromix_amd import _lib
lib = _lib.load_diag()
dev = torch.device("cuda:0")

def run(shape, ea, eb, a, b, sa, sb):
    out = torch.zeros((64, 16 if shape == 32 else 4), dtype=torch.float32, device=dev)
    lib.mm_diag_mfma(shape, ea, eb, 0, a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()

for shape in (32, 16):
    R = shape; nh = 64 // R
    ones = torch.full((64,), 127, dtype=torch.int32, device=dev)
    # B = fp4 position coded: column j (lane j + R*h'), nibble j%32... for R=16 only 16 columns -> two passes
    for which in ("A", "B"):
        print(f"--- shape {shape}: fp8 as operand {which}; rows: (lane-half h, byte e) -> K position (h', nibble) in fp4 order")
        for h in range(nh):
            line = []
            for e in range(32):
                res = None
                for jpass in range(32 // R if R == 16 else 1):
                    f8 = np.zeros((64, 32), np.uint8)
                    f8[h * R + 0, e] = 0x38  # 1.0 in e4m3, row/col 0
                    f4 = np.zeros((64, 32), np.uint8)  # first 16 bytes used
                    for j in range(R):
                        nib = j + jpass * R
                        for hp in range(nh):
                            code = [0x2, 0x4, 0x5, 0x6][hp]  # 1, 2, 3, 4 in e2m1
                            f4[j + R * hp, nib // 2] |= code << (4 * (nib & 1))
                    t8 = torch.from_numpy(f8.view(np.int32).reshape(64, 8)).to(dev)
                    t4 = torch.from_numpy(f4.view(np.int32).reshape(64, 8)).to(dev)
                    if which == "A":
                        out = run(shape, 2, 0, t8, t4, ones, ones)
                    else:
                        out = run(shape, 0, 2, t4, t8, ones, ones)
                    # D[0][j] (A fp8 row 0) or D[j][0] (B fp8 col 0)
                    for j in range(R):
                        if which == "A":
                            if R == 32: v = out[j, 0]          # lane j (col j), reg 0 -> row 0
                            else: v = out[j, 0]                # lane j: col j, row 0 = reg 0
                        else:
                            if R == 32:
                                l = 0 + 32 * ((j >> 2) & 1); r = (j & 3) + 4 * (j >> 3)
                                v = out[l, r]
                            else:
                                l = 0 + 16 * (j >> 2); r = j & 3
                                v = out[l, r]
                        if v != 0:
                            res = (int(v) - 1, j + jpass * R)
                line.append(res)
            print(f"h={h}:", " ".join(f"{r[0]}.{r[1]:02d}" if r else "--.--" for r in line))
