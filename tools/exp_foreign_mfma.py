#!/usr/bin/env python3
"""Round 5: does the fp64 victim (tools/ubench/spin_victim.hip: svd_kernel, the detector's 2x2 fp64 Jacobi SVD repeated on fixed
inputs) see wrong rounds next to matrix-core kernels that are NOT this library's - the MIOpen convolutions of the three daemon
networks (zmq_daemon.py --device cuda), rocBLAS / hipBLASLt GEMMs of several types?  One aggressor at a time on a thread of its own
(torch stream), the victim launched from the main thread.  Prints one line per aggressor:
    <aggressor> launches <n> rounds <r> wrong <w> aggressor_calls <c>
usage: python tools/exp_foreign_mfma.py [seconds per aggressor]"""
import ctypes
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mods-light-zmq_amd"))


def build_victim():
    so = "/tmp/libspin_victim.so"
    src = os.path.join(ROOT, "tools", "ubench", "spin_victim.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-w", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def aggressors(torch):
    import zmq_daemon
    dev = "cuda"
    out = {}

    def gemm(dt, n=4096):
        a = torch.randn(n, n, device=dev).to(dt)
        b = torch.randn(n, n, device=dev).to(dt)
        return lambda: torch.matmul(a, b)
    out["gemm_bf16_4096"] = gemm(torch.bfloat16)
    out["gemm_f16_4096"] = gemm(torch.float16)
    out["gemm_f32_4096"] = gemm(torch.float32)
    out["gemm_f64_2048"] = gemm(torch.float64, 2048)
    out["gemm_bf16_512"] = gemm(torch.bfloat16, 512)
    x = torch.rand(2000, 1, 32, 32, device=dev) * 255.0
    for name in ("hardnet", "affnet", "orinet"):
        f = zmq_daemon.build_model(name, None, 0, dev)
        xs = x.cpu().numpy()
        out["daemon_" + name] = (lambda f=f, xs=xs: f(xs))
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
    xc = torch.randn(256, 64, 32, 32, device=dev)
    with torch.no_grad():
        out["conv3x3_f32"] = lambda: conv(xc)
        convh = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).half()
        xh = xc.half()
        out["conv3x3_f16"] = lambda: convh(xh)
    return out


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    import torch
    spin = build_victim()
    res = (ctypes.c_uint * 3)()
    assert spin.svd_launch(2048, 300, res) == 0
    base = list(res)
    print("victim alone: rounds %d wrong %d" % (base[2], base[0]), flush=True)
    aggs = aggressors(torch)
    for name, fn in aggs.items():
        stop = threading.Event()
        calls = [0]
        err = []

        def run():
            try:
                s = torch.cuda.Stream()
                with torch.cuda.stream(s), torch.no_grad():
                    while not stop.is_set():
                        for _ in range(8):
                            fn()
                        s.synchronize()
                        calls[0] += 8
            except Exception as e:   # pragma: no cover
                err.append(repr(e)[:200])

        with torch.no_grad():      # first calls choose / compile the library's kernels (MIOpen: seconds): not inside the measurement
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
        before = list(res)
        th = threading.Thread(target=run)
        th.start()
        t0 = time.time()
        n = 0
        while time.time() - t0 < secs:
            assert spin.svd_launch(2048, 300, res) == 0
            n += 1
        stop.set()
        th.join()
        print("%-18s launches %5d rounds %9d wrong %9d aggressor_calls %6d %s" % (name, n, res[2] - before[2], res[0] - before[0], calls[0], err[0] if err else ""), flush=True)


if __name__ == "__main__":
    main()
