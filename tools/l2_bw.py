import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from micromix_amd import _lib
lib = _lib.load_diag(); dev = torch.device("cuda:0")
sink = torch.zeros(4, device=dev)
buf = torch.randint(0, 255, (256 << 20,), dtype=torch.uint8, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
for region_mb in (2, 16):
    region = region_mb << 20
    for mode, name in ((1, "LDS-DMA contiguous"), (2, "LDS-DMA 8x128B rows stride 4096"), (4, "VGPR loads, 12 deep"), (5, "VGPR loads + ds_write")):
        for blocks in (128, 256, 512, 768):
            m, stride = (2, 4224) if mode == 3 else (mode, 4096)
            kb, iters = 48, 400
            lib.mm_diag_l2_bw(buf.data_ptr(), region, stride, kb, 20, m, blocks, sink.data_ptr(), st())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.mm_diag_l2_bw(buf.data_ptr(), region, stride, kb, iters, m, blocks, sink.data_ptr(), st())
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            tot = blocks * kb * 1024 * iters
            print(f"region {region_mb:3d} MB  {name:34s} blocks {blocks:4d}: {tot/ms/1e9:7.2f} TB/s  ({tot/ms/1e6/min(blocks,256):6.1f} GB/s per CU)", flush=True)
