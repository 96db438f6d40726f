#!/usr/bin/env python3
"""Descriptor / affine-shape / orientation daemon for the MODS ZMQ protocol, computing on the MI355X through
PyTorch-ROCm (or on the CPU with --device cpu).

Serves what the reference's build/desc_server.py (HardNet, port 5555), build/affnet_server.py (AffNet, port 5556) and
build/orinet_server.py (OriNet, port 5557) serve: a REP socket; every request is the PNG of an 8-bit image holding n
32x32 patches in a column; the reply is n x dim float32:
    hardnet  dim 128, clip(210 * (d + 0.45), 0, 255) quantised to integers   (desc_server.py:44)
    affnet   dim 3,   (a11, a21, a22) with +1 on the first and the last       (affnet_server.py:80-84)
    orinet   dim 2,   (y, x) of the orientation; the client takes atan2        (orinet_server.py:80-82)
    stats    dim 4,   (mean, std, min, max) of the patch - a deterministic model for protocol tests
Socket handling and PNG decoding come from libmodszmq.so (no pyzmq / cv2 needed).  Weights: --weights FILE loads a
checkpoint with a 'state_dict' entry as the reference's servers do (HardNet++.pth, AffNet.pth, OriNet.pth); without it
the network is initialised from --seed (there is no network access for checkpoints in the build environment).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int))


def wire():
    lib = C.CDLL(os.path.join(HERE, "libmodszmq.so"))
    lib.mods_zmq_last_error.restype = C.c_char_p
    return lib


TRUST_CHECKPOINT = False   # --trust-checkpoint: load a pickled checkpoint with the full unpickler (runs code from the file)


def build_model(name, weights=None, seed=0, device="cpu"):
    """Returns f(patches float32 [n,1,ps,ps] in 0..255) -> float32 [n, dim]."""
    if name == "stats":
        def stats(p):
            f = p.reshape(p.shape[0], -1).astype(np.float64)
            return np.stack([f.mean(1), f.std(1), f.min(1), f.max(1)], axis=1).astype(np.float32)
        return stats
    import torch
    import torch.nn as nn

    def input_norm(x):                      # per-patch mean / std, as the three reference networks do
        flat = x.view(x.size(0), -1)
        mp = flat.mean(dim=1).view(-1, 1, 1, 1)
        sp = (flat.std(dim=1) + 1e-7).view(-1, 1, 1, 1)
        return (x - mp) / sp

    def block(cin, cout, stride=1):
        return [nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(cout, affine=False), nn.ReLU()]

    class HardNet(nn.Module):               # 32x32 -> 128-D, L2 normalised
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(*block(1, 32), *block(32, 32), *block(32, 64, 2), *block(64, 64), *block(64, 128, 2),
                                          *block(128, 128), nn.Dropout(0.3), nn.Conv2d(128, 128, kernel_size=8, bias=False),
                                          nn.BatchNorm2d(128, affine=False))

        def forward(self, x):
            y = self.features(input_norm(x)).view(x.size(0), -1)
            return y / torch.sqrt((y * y).sum(dim=1, keepdim=True) + 1e-10)

    class ShapeNet(nn.Module):              # AffNetFast (3 outputs, 8x8 head) / OriNetFast (2 outputs, PS/4 head with padding 1)
        def __init__(self, n_out, head_pad):
            super().__init__()
            self.features = nn.Sequential(*block(1, 16), *block(16, 16), *block(16, 32, 2), *block(32, 32), *block(32, 64, 2),
                                          *block(64, 64), nn.Dropout(0.25),
                                          nn.Conv2d(64, n_out, kernel_size=8, stride=1, padding=head_pad, bias=True), nn.Tanh(),
                                          nn.AdaptiveAvgPool2d(1))
            self.n_out = n_out

        def forward(self, x):
            y = self.features(input_norm(x)).view(-1, self.n_out)
            if self.n_out == 3:
                y = y + torch.tensor([1.0, 0.0, 1.0], device=y.device)
            return y

    torch.manual_seed(seed)
    net = {"hardnet": lambda: HardNet(), "affnet": lambda: ShapeNet(3, 0), "orinet": lambda: ShapeNet(2, 1)}[name]()
    if isinstance(weights, dict):           # tensors by name (e.g. the arrays of tests/golden/nets.npz)
        net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in weights.items()})
    elif weights:
        # a checkpoint is a pickle: the restricted unpickler (tensors and plain containers only - all the reference's
        # checkpoints need) unless the operator vouches for the file with --trust-checkpoint
        ck = torch.load(weights, map_location="cpu", weights_only=not TRUST_CHECKPOINT)
        net.load_state_dict(ck["state_dict"] if "state_dict" in ck else ck)   # strict: the architecture has to be the checkpoint's
    net = net.eval().to(device)

    on_gpu = str(device) != "cpu"
    BUCKETS = (64, 512)                                           # batch shapes the GPU ever sees (512 = BATCH_SIZE of the reference servers)

    def forward(chunk):
        # On the GPU a chunk is padded with zero patches to one of two batch sizes: the convolution library then selects (and on
        # first use tries) kernels for exactly two shapes, both at start-up (warm_up below) and never while a matcher's fp64 work
        # shares the GPU - the first calls of a new shape are where foreign matrix-core kernels disturbed fp64 results
        # (profiles/r05_foreign_mfma_first_calls.log) - and a patch's output does not depend on how
        # many patches its request held.
        n = len(chunk)
        if on_gpu:
            b = next(x for x in BUCKETS if n <= x)
            if n < b:
                chunk = np.concatenate([chunk, np.zeros((b - n,) + chunk.shape[1:], chunk.dtype)], axis=0)
        return net(torch.from_numpy(chunk).to(device)).float().cpu().numpy()[:n]

    def run(p):
        outs = []
        with torch.no_grad():
            for i in range(0, len(p), 512):
                outs.append(forward(p[i:i + 512]))
        out = np.concatenate(outs, axis=0) if outs else np.zeros((0, 1), np.float32)
        if name == "hardnet":
            out = np.clip(210 * (out.astype(np.float64) + 0.45), 0, 255).astype(np.uint8).astype(np.float32)
        return out.astype(np.float32)

    def warm_up(ps=32):
        """Every batch shape once (twice: the second call runs the selected kernels), before the first request is served."""
        if on_gpu:
            with torch.no_grad():
                for b in BUCKETS:
                    for _ in range(2):
                        forward(np.full((b, 1, ps, ps), 128.0, np.float32) + np.arange(ps, dtype=np.float32))
            torch.cuda.synchronize()
    run.warm_up = warm_up
    return run


def serve(endpoint, model, max_requests=0):
    lib = wire()

    def cb(user, patches, n, ps, out, cap, dim):
        try:
            p = np.ctypeslib.as_array(patches, shape=(n, 1, ps, ps)).astype(np.float32)
            res = np.ascontiguousarray(model(p), np.float32)
            if res.shape[0] != n or res.size > cap:
                return 1
            C.memmove(out, res.ctypes.data, res.nbytes)
            dim[0] = res.shape[1]
            return 0
        except Exception as e:                                    # a REP socket has to answer; the client sees an empty reply
            print("zmq_daemon: model failed: %r" % (e,), file=sys.stderr)
            return 1
    fn = MODEL_FN(cb)
    rc = lib.mods_zmq_serve(endpoint.encode(), fn, None, max_requests)
    if rc:
        raise RuntimeError("mods_zmq_serve: %s" % lib.mods_zmq_last_error().decode())


def main():
    ap = argparse.ArgumentParser(description="MODS patch daemon (HardNet / AffNet / OriNet wire protocol) on PyTorch-ROCm")
    ap.add_argument("--model", default="hardnet", choices=["hardnet", "affnet", "orinet", "stats"])
    ap.add_argument("--port", default=None, help="TCP port (default: 5555 hardnet, 5556 affnet, 5557 orinet)")
    ap.add_argument("--bind", default=None, help="full endpoint instead of tcp://*:PORT")
    ap.add_argument("--weights", default=None, help="checkpoint (.pth with a 'state_dict' entry, as the reference's servers load it); "
                                                    "FILE.npz: arrays named <model>.<parameter> (tests/golden/nets.npz)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default=None, help="cuda (MI355X through ROCm) or cpu; default: cuda when available")
    ap.add_argument("--max-requests", type=int, default=0)
    ap.add_argument("--trust-checkpoint", action="store_true",
                    help="load --weights with the unrestricted unpickler (executes code stored in the file: only for checkpoints "
                         "of known origin; the default accepts tensors and plain containers, which is all a state_dict holds; "
                         "a .npz file needs no pickle at all)")
    args = ap.parse_args()
    global TRUST_CHECKPOINT
    TRUST_CHECKPOINT = bool(args.trust_checkpoint)
    device = args.device
    if device is None and args.model != "stats":
        import torch
        device = "cuda" if torch.cuda.is_available() else "cpu"
    port = args.port or {"hardnet": "5555", "affnet": "5556", "orinet": "5557", "stats": "5558"}[args.model]
    endpoint = args.bind or "tcp://*:" + port
    weights = args.weights
    if weights and weights.endswith(".npz"):
        g = np.load(weights)
        weights = {k[len(args.model) + 1:]: g[k] for k in g.files if k.startswith(args.model + ".")}
        if not weights:
            raise SystemExit("zmq_daemon: %s holds no arrays named %s.*" % (args.weights, args.model))
    model = build_model(args.model, weights, args.seed, device or "cpu")
    if hasattr(model, "warm_up"):
        model.warm_up()
    print("zmq_daemon: %s on %s, serving %s" % (args.model, device or "cpu", endpoint), file=sys.stderr, flush=True)
    serve(endpoint, model, args.max_requests)


if __name__ == "__main__":
    main()
