#!/usr/bin/env python3
"""Static instruction statistics of gfx950 kernels, from the compiler's assembly (no GPU needed): the per-keypoint kernels are bound by
VALU issue (DESIGN.md "Per-keypoint kernels"), so what counts is how many vector instructions a loop body holds and of which class.

  python tools/isa_stats.py <unit, e.g. sift> [kernel-name substring ...]      compiles csrc/<unit>.hip to assembly (cached in /tmp)
  options: --dump <substring>   print the kernel's assembly
           --flags "<extra hipcc flags>"

Per kernel: instructions, VALU, of those the half-rate classes (v_cndmask, v_cvt, v_min/max/med, shifts, v_mul_lo, compares, fp64), the
quarter-rate transcendentals, SGPR-spill traffic (v_readlane / v_writelane), LDS and vector-memory instructions, and the same per
basic block that a backward branch closes (= a loop body), largest first."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = re.compile(r"^v_(cndmask|cvt|min|max|med3|lshl|lshr|ashr|mul_lo|mul_hi|mad_u32|mad_i32|cmp|cmpx|floor|fract|ceil|trunc|rndne|bfe|bfi|perm|alignbit|readlane|writelane|readfirstlane)|_f64")
TRANS = re.compile(r"^v_(sqrt|rcp|rsq|exp|log|sin|cos)_")


def classify(op):
    if op.startswith("v_"):
        if TRANS.match(op):
            return "trans"
        if HALF.search(op):
            return "half"
        return "full"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def stats(lines):
    c = {"full": 0, "half": 0, "trans": 0, "lds": 0, "vmem": 0, "salu": 0, "other": 0, "spill": 0, "mfma": 0}
    for op in lines:
        c[classify(op)] += 1
        if op.startswith(("v_readlane", "v_writelane")):
            c["spill"] += 1
        if "mfma" in op:
            c["mfma"] += 1
    c["valu"] = c["full"] + c["half"] + c["trans"]
    # issue cycles by the rates of tools/ubench/valu_rate.hip (full 2.7, half 4.4, transcendental ~9 cycles per wave64 instruction)
    c["cyc"] = int(2.7 * c["full"] + 4.4 * c["half"] + 9 * c["trans"])
    return c


def fmt(c):
    return "valu %5d (full %5d half %5d trans %3d) ~%6d cyc | sgpr-spill %3d lds %4d vmem %4d salu %5d" % (
        c["valu"], c["full"], c["half"], c["trans"], c["cyc"], c["spill"], c["lds"], c["vmem"], c["salu"])


def main():
    args = sys.argv[1:]
    dump, flags = None, ""
    if "--dump" in args:
        i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
    if "--flags" in args:
        i = args.index("--flags"); flags = args[i + 1]; del args[i:i + 2]
    unit, pats = args[0], args[1:]
    src = os.path.join(ROOT, "mods-light-zmq_amd", "csrc", unit + ".hip")
    out = "/tmp/isa_%s.s" % unit
    extra = "-mllvm -amdgpu-mfma-vgpr-form=1" if unit == "match" else ""
    cmd = "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w --cuda-device-only -S %s %s %s -o %s" % (extra, flags, src, out)
    subprocess.check_call(cmd, shell=True)
    text = open(out).read().split("\n")
    # kernels: from "<name>:" (a .globl'd function symbol) to s_endpgm
    names = [l.split()[1] for l in text if l.startswith("\t.amdhsa_kernel ")]
    dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE).stdout.decode().split("\n")
    for name, pretty in zip(names, dem):
        if pats and not any(p in pretty or p in name for p in pats):
            continue
        a = next((i for i, l in enumerate(text) if l.startswith(name + ":")), None)
        if a is None:
            continue
        body = []
        for l in text[a + 1:]:
            body.append(l)
            if l.strip().startswith("s_endpgm"):
                break
        if dump and (dump in pretty or dump in name):
            print("\n".join(body))
            continue
        ops, labels, loops = [], {}, []
        for l in body:
            t = l.strip()
            ml = re.match(r"^(\.LBB\w+):", t)
            if ml:
                labels[ml.group(1)] = len(ops)
                continue
            if not t or t.startswith((";", ".", "//")):
                continue
            op = t.split()[0]
            ops.append(op)
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", t)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in labels:       # backward branch: a loop
                    loops.append((labels[tgt], len(ops)))
        short = pretty.split("(")[0].replace("void ", "").replace("mods::", "")
        print("%-50s %6d instr | %s" % (short[-50:], len(ops), fmt(stats(ops))))
        seen = []
        for a0, b0 in sorted(loops, key=lambda ab: ab[0] - ab[1]):
            if any(a0 >= x and b0 <= y and (b0 - a0) == (y - x) for x, y in seen):
                continue
            seen.append((a0, b0))
            if b0 - a0 < 12:
                continue
            print("      loop %5d..%5d %6d instr | %s" % (a0, b0, b0 - a0, fmt(stats(ops[a0:b0]))))


if __name__ == "__main__":
    main()
