"""Achievable HBM rates for the traffic patterns of the pyramid kernels (torch ops, HIP events): write only, copy (1 read :
1 write), broadcast copy (1 read : 2 writes = the fused blur + response kernel's pattern), at the octave-0 launch size."""
import torch
n = 16 * 1920 * 1080          # one fp32 plane of a 16-image batch: 133 MB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty(n, dtype=torch.float32, device="cuda")
c = torch.empty(2, n, dtype=torch.float32, device="cuda")


def timeit(fn, bytes_moved, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-34s %7.1f us  %5.2f TB/s" % (name, ms * 1e3, bytes_moved / (ms * 1e-3) / 1e12))


timeit(lambda: b.fill_(1.0), 4 * n, "write 133 MB")
timeit(lambda: c.fill_(1.0), 8 * n, "write 265 MB")
timeit(lambda: b.copy_(a), 8 * n, "read 133 MB + write 133 MB")
timeit(lambda: c.copy_(a), 12 * n, "read 133 MB + write 265 MB")
timeit(lambda: torch.add(a, 1.0, out=b), 8 * n, "read 133 + write 133 (add)")
