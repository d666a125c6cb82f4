#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in libdsdiff.so, read from the gfx950 code objects embedded in the library
(no GPU needed): the AMDGPU metadata note of each object via llvm-readelf.
    python tools/kernel_resources.py [--spills]        # --spills: only kernels with scratch or spilled registers
Used by tests/test_abi.py: a kernel that hipcc made spill is a build regression (and, for kernels that keep asm-loaded
registers live, a correctness hazard — DESIGN.md §9)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
        i = m.start()
        if data[i + 18:i + 20] != b"\xe0\x00":      # e_machine = EM_AMDGPU
            continue
        e_shoff, = struct.unpack_from("<Q", data, i + 0x28)
        e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 0x3A)
        yield data[i:i + e_shoff + e_shentsize * e_shnum]


def kernels(path=None):
    """[{name, vgpr, sgpr, lds, scratch, vgpr_spills, sgpr_spills}] over all code objects of the library."""
    path = path or os.path.join(ROOT, "diffusion_models_dsdiff_amd", "libdsdiff.so")
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            def field(k, d=0):
                mm = re.search(r"\." + k + r":\s+(\S+)", blk)
                return mm.group(1) if mm else d
            out.append({"name": field("name", "?"), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                        "lds": int(field("group_segment_fixed_size")), "scratch": int(field("private_segment_fixed_size")),
                        "vgpr_spills": int(field("vgpr_spill_count")), "sgpr_spills": int(field("sgpr_spill_count"))})
    return out


if __name__ == "__main__":
    ks = kernels()
    only = "--spills" in sys.argv
    print(f"{len(ks)} kernels")
    for k in sorted(ks, key=lambda k: -k["vgpr"]):
        if only and not (k["scratch"] or k["vgpr_spills"] or k["sgpr_spills"]):
            continue
        print(f"{k['vgpr']:4d} vgpr {k['sgpr']:4d} sgpr {k['lds']:7d} lds {k['scratch']:6d} scratch {k['vgpr_spills']:4d}/{k['sgpr_spills']:<4d} spills  {k['name'][:110]}")
