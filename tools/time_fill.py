"""Write-only bandwidth probe: torch fill of a 32 MiB bf16 tensor, and a copy, with event timing."""
import torch
dev = torch.device("cuda:0")
for mb in (32, 64, 256):
    t = torch.empty(mb << 19, dtype=torch.bfloat16, device=dev)
    s = torch.empty_like(t)
    for name, f in (("fill", lambda: t.fill_(1.0)), ("copy", lambda: t.copy_(s))):
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1000
        print(f"{name} {mb} MiB: {us:.1f} us  {mb * 1.048576 / us * (2 if name == 'copy' else 1):.2f} TB/s", flush=True)
