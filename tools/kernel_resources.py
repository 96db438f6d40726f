#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of every gfx950 kernel in a built libmodsgpu.so, read from the code objects
the library carries (no recompilation): the .hip_fatbin section is a sequence of clang offload bundles, one per
translation unit; each holds one AMDGPU ELF whose note section lists the kernels (.vgpr_count, .agpr_count, ...), and
the disassembly says which kernels issue MFMA instructions.

  python tools/kernel_resources.py [path/to/libmodsgpu.so]     prints one line per kernel

Used by tests/test_cpu_host.py::test_matrix_core_kernels_own_their_simds (DESIGN.md "The matcher and its neighbours":
waves that issue independent MFMA chains disturb fp64 work of foreign waves on their SIMD, so the library's matrix-core
kernels allocate whole SIMDs)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """the gfx950 ELF images inside the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(td, "unused.so")])
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24: pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[base + off: base + off + size])
    return out


def waves_per_simd(vgprs_total):
    alloc = (max(vgprs_total, 1) + 7) // 8 * 8          # allocation granule of the unified VGPR/AGPR file
    return alloc, min(8, 512 // alloc)


def kernels(lib_path):
    """[{name, vgpr, agpr, sgpr, lds, scratch, alloc, waves, mfma}] for every kernel of the library"""
    res = []
    for elf in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf); f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], stdout=subprocess.PIPE, check=True).stdout.decode()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], stdout=subprocess.PIPE, check=True).stdout.decode()
        mfma = {}
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); mfma.setdefault(cur, 0)
            elif cur and "v_mfma" in line:
                mfma[cur] += 1
        # the metadata is YAML: one "- .agpr_count: ..." block per kernel
        for block in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
            block = ".agpr_count:" + block
            get = lambda key: (re.search(r"\." + key + r":\s*(\S+)", block) or [None, "0"])[1]
            name = get("name")
            sym = get("symbol").replace(".kd", "")
            vg, ag = int(get("vgpr_count")), int(get("agpr_count"))
            alloc, waves = waves_per_simd(vg)            # .vgpr_count is the unified count on gfx950 (AGPRs included)
            res.append(dict(name=name, vgpr=vg, agpr=ag, sgpr=int(get("sgpr_count")), lds=int(get("group_segment_fixed_size")),
                            scratch=int(get("private_segment_fixed_size")), wg_size=int(get("max_flat_workgroup_size")), alloc=alloc, waves=waves, mfma=mfma.get(sym, mfma.get(name, 0))))
    return res


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return names
    p = subprocess.run([tool], input="\n".join(names).encode(), stdout=subprocess.PIPE)
    return p.stdout.decode().splitlines() if p.returncode == 0 else names


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "mods-light-zmq_amd", "libmodsgpu.so")
    ks = kernels(lib)
    dn = demangle([k["name"] for k in ks])
    print("%-64s %5s %5s %6s %5s %7s %7s %5s" % ("kernel", "vgpr", "alloc", "waves", "sgpr", "lds", "scratch", "mfma"))
    for k, n in sorted(zip(ks, dn), key=lambda kn: kn[1]):
        flag = ""
        if k["mfma"]:
            flag = "  owns its SIMDs" if k["alloc"] * (k["wg_size"] // 256) == 512 else "  <-- MFMA kernel that shares its SIMDs"
        print("%-64s %5d %5d %6d %5d %7d %7d %5d%s" % (n.split("(")[0][-64:], k["vgpr"], k["alloc"], k["waves"], k["sgpr"], k["lds"], k["scratch"], k["mfma"], flag))
