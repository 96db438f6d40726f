#!/usr/bin/env python3
"""Several contexts on one GPU (one thread and stream each) repeat detect + describe + match of the same 1080p pairs and compare
every result with the first one: development aid for an intermittent difference in the order of a pipeline's inlier list."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import synth

pkg = ge.load_package()
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
redetect = 1
aggressor = sys.argv[3] if len(sys.argv) > 3 else ""      # "", "detect", "match", "full", "gemm", "stream", "poison1|2|3" [pattern]: what the threads 1.. run (thread 0 = the victim)
stop = threading.Event()
W, H = 1920, 1080
pairs = [synth.pair(W, H, seed=2000 + i)[:2] for i in range(3)]
dev = [torch.from_numpy(np.stack([np.round(a).clip(0, 255), np.round(b).clip(0, 255)]).astype(np.float32)).cuda() for a, b in pairs]
torch.cuda.synchronize()
lock = threading.Lock()
bad = [0]


def aggress_torch(k):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        if aggressor == "gemm":
            a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
        else:
            a = torch.randn(64 << 20, device="cuda")
        while not stop.is_set():
            if aggressor == "gemm":
                for _ in range(8): c = a @ b
            else:
                for _ in range(8): c = a * 1.5 + 2.0
            st.synchronize()


def aggress_poison(k):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libpoison.so"))
    mode = int(aggressor[len("poison"):])              # 1 registers, 2 LDS, 3 both
    pattern = int(sys.argv[4], 16) if len(sys.argv) > 4 else 0x7fc00000
    while not stop.is_set():
        rc = lib.poison_launch(mode, ctypes.c_uint(pattern), 1024, 1)
        if rc:
            print("poison_launch failed", rc); return


def aggress_top(k):
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libvgprtop.so"))
    mode = int(aggressor[len("top"):])      # tools/ubench/vgpr_top.hip: 0 ds_read_b128 v[124:127], 1 control v[120:123], 5 ds_read -> MFMA srcA v[124:127]
    while not stop.is_set():
        rc = lib.top_launch(mode, 4096, 400)
        if rc:
            print("top_launch failed", rc); return


def aggress(k):
    if aggressor.startswith("top"):
        return aggress_top(k)
    if aggressor in ("gemm", "stream"):
        return aggress_torch(k)
    if aggressor.startswith("poison"):
        return aggress_poison(k)
    ctx = pkg.Context(0, W, H, 2)
    ctx.detect_describe_dev(dev[0].data_ptr(), 2, W, H)
    n = 0
    while not stop.is_set():
        if aggressor == "detect":
            ctx.detect_hessian_affine_dev(dev[n % len(dev)].data_ptr(), 2, W, H, fetch=False)
        elif aggressor == "match":
            try:
                ctx.match_dev(0, 1)
            except Exception:
                pass
        else:
            ctx.detect_describe_dev(dev[n % len(dev)].data_ptr(), 2, W, H)
        n += 1
    ctx.close()


def worker_detect(k):
    """victim that runs the detector only (pyramid, NMS, localisation, Baumberg, export) and compares the keypoints"""
    ctx = pkg.Context(0, W, H, 2)
    ref = {}
    par = pkg.HessAffParams.default()
    if os.environ.get("VICTIM_NO_BAUMBERG"):
        par.doBaumberg = 0
    if os.environ.get("VICTIM_HESSIAN_BAUMBERG"):
        par.affBmbrgMethod = 1
    for it in range(reps):
        p = (it + k) % len(dev)
        keys = ctx.detect_hessian_affine_dev(dev[p].data_ptr(), 2, W, H, par, max_out=1 << 15)
        if p not in ref:
            ref[p] = keys
            continue
        msgs = []
        for s in (0, 1):
            if len(keys[s]) != len(ref[p][s]):
                msgs.append("key count of image %d: %d vs %d" % (s, len(keys[s]), len(ref[p][s])))
                continue
            for f in ("x", "y", "s", "a11", "a21", "a22", "response"):
                if not np.array_equal(keys[s][f], ref[p][s][f]):
                    msgs.append("image %d key field %s differs at %s" % (s, f, np.nonzero(keys[s][f] != ref[p][s][f])[0][:4]))
        if msgs:
            with lock:
                bad[0] += 1
                print("thread %d iteration %d pair %d: %s" % (k, it, p, "; ".join(msgs)), flush=True)
    ctx.close()


def worker(k):
    if os.environ.get("VICTIM") == "detect":
        return worker_detect(k)
    ctx = pkg.Context(0, W, H, 2)
    ref = {}
    for it in range(reps):
        p = (it + k) % len(dev)
        if redetect or p not in ref or True:
            ctx.detect_describe_dev(dev[p].data_ptr(), 2, W, H)
        tent, u6 = ctx.match_dev(0, 1) if not os.environ.get('MODS_MATCH_MASK') else (np.zeros(0, pkg.TENT_DTYPE), np.zeros((0, 6)))
        regs = (ctx.regions_fetch(0), ctx.regions_fetch(1))
        if p not in ref:
            ref[p] = (tent, u6, regs)
            continue
        rt, ru, rr = ref[p]
        msgs = []
        if len(tent) != len(rt):
            msgs.append("tentative count %d vs %d" % (len(tent), len(rt)))
        else:
            for f in tent.dtype.names:
                if f != "pad" and not np.array_equal(tent[f], rt[f]):
                    d = np.nonzero(tent[f] != rt[f])[0]
                    msgs.append("field %s differs at %s: %s vs %s" % (f, d[:4], tent[f][d[:4]], rt[f][d[:4]]))
            if not np.array_equal(u6, ru):
                msgs.append("u6 differs")
        if regs is not None and rr is not None:
            for s in (0, 1):
                if len(regs[s]) != len(rr[s]):
                    msgs.append("region count of image %d: %d vs %d" % (s, len(regs[s]), len(rr[s])))
                    continue
                for f in ("x", "y", "s", "a11", "a12", "a21", "a22", "response"):
                    if not np.array_equal(regs[s][f], rr[s][f]):
                        d = np.nonzero(regs[s][f] != rr[s][f])[0]
                        msgs.append("image %d region field %s differs at %s" % (s, f, d[:6]))
                dd = np.nonzero(np.any(regs[s]["desc"] != rr[s]["desc"], axis=1))[0]
                if len(dd):
                    r = dd[0]
                    e = np.nonzero(regs[s]["desc"][r] != rr[s]["desc"][r])[0]
                    msgs.append("image %d: %d descriptors differ, first region %d (x %.1f y %.1f s %.2f), %d of 128 bytes: idx %s got %s want %s"
                                % (s, len(dd), r, regs[s]["x"][r], regs[s]["y"][r], regs[s]["s"][r], len(e), e[:12], regs[s]["desc"][r][e[:12]], rr[s]["desc"][r][e[:12]]))
        if msgs:
            with lock:
                bad[0] += 1
                print("thread %d iteration %d pair %d: %s" % (k, it, p, "; ".join(msgs)), flush=True)
    ctx.close()


t0 = time.time()
if aggressor:
    ths = [threading.Thread(target=aggress, args=(k,)) for k in range(1, n_threads)]
    for t in ths: t.start()
    worker(0)
    stop.set()
    for t in ths: t.join()
else:
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    for t in ths: t.start()
    for t in ths: t.join()
print("%d threads x %d repetitions, %d differing results, %.1f s" % (n_threads, reps, bad[0], time.time() - t0))
