#!/usr/bin/env python3
"""Pins the AffNet / OriNet daemons (mods-light-zmq_amd/zmq_daemon.py) against the reference's own networks.

Run in the build container (needs /root/reference).  The reference ships build/AffNet.pth and build/OriNet.pth together with
the server scripts build/affnet_server.py / orinet_server.py (which cannot be imported here: they need zmq, cv2 and a CUDA
device at import time).  This script takes the two network CLASSES out of those scripts (ast: the class definitions only, run
from where they lie, nothing is copied), loads the checkpoints into them exactly as the servers do
(load_state_dict(checkpoint['state_dict']), strict), evaluates them on fixed patches on the CPU in fp32, checks that the
daemon's own modules loaded from the same checkpoints (strict) give the same numbers, and writes tests/golden/nets.npz:
  patches            64 x 32 x 32 uint8 (crops of a synthetic texture, the wire format of the protocol)
  affnet_out, orinet_out   the REFERENCE classes' outputs (float32; AffNet with +1 on columns 0 and 2 as its forward does)
  affnet.<key>, orinet.<key>   the checkpoint tensors (data files of the reference: the .pth files do not travel to the GPU
                     box, the arrays do), so that the -m gpu test can run the daemon on the MI355X with the real weights
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/build"
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mods-light-zmq_amd"))
import synth        # noqa: E402
import zmq_daemon   # noqa: E402


def reference_class(script, name):
    """The class `name` of a reference server script, compiled from its own source text (class definition only)."""
    src = open(os.path.join(REF, script)).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"torch": torch, "nn": nn}
    exec(compile(mod, os.path.join(REF, script), "exec"), ns)
    return ns[name]


def main():
    img = synth.texture(512, 384, seed=77)
    rng = np.random.default_rng(77)
    patches = np.stack([img[y:y + 32, x:x + 32] for y, x in zip(rng.integers(0, 352, 64), rng.integers(0, 480, 64))]).astype(np.uint8)
    x = torch.from_numpy(patches.astype(np.float32)).unsqueeze(1)
    out = {"patches": patches}
    for tag, script, cls, pth in (("affnet", "affnet_server.py", "AffNetFast", "AffNet.pth"), ("orinet", "orinet_server.py", "OriNetFast", "OriNet.pth")):
        ck = torch.load(os.path.join(REF, pth), map_location="cpu", weights_only=False)
        ref = reference_class(script, cls)()
        ref.load_state_dict(ck["state_dict"])            # strict, as the server does
        ref.eval()
        with torch.no_grad():
            want = ref(x.clone()).float().numpy().copy()
        mine = zmq_daemon.build_model(tag, weights=os.path.join(REF, pth), device="cpu")(patches.astype(np.float32)[:, None])
        err = float(np.max(np.abs(mine - want)))
        print(tag, "reference class vs daemon module on the CPU: max abs difference", err, "outputs", want[:2])
        assert err < 1e-6, err
        out[tag + "_out"] = want.astype(np.float32)
        for k, v in ck["state_dict"].items():
            out[tag + "." + k] = v.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nets.npz"), **out)
    print("wrote tests/golden/nets.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "nets.npz")), "bytes")


if __name__ == "__main__":
    main()
