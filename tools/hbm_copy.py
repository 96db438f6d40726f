"""Achievable HBM stream rate on this GPU: device-to-device copies (read + write bytes per second) at the pyramid's launch sizes."""
import torch
for mb in (16, 66, 133, 266, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy of %4d MB: %.1f us, %.2f TB/s (read + write)" % (mb, ms * 1e3, 2 * n * 4 / (ms * 1e-3) / 1e12))
