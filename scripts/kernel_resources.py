#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in the built libnavhip.so (the HSA metadata notes of its gfx950 code
objects): what decides the waves per SIMD of a launch.
    python scripts/kernel_resources.py [lib] [substring...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    objdump, readelf = (os.path.join("/opt/rocm/lib/llvm/bin", t) for t in ("llvm-objdump", "llvm-readelf"))
    work = tempfile.mkdtemp()
    shutil.copy(lib, work)
    subprocess.run([objdump, "--offloading", os.path.basename(lib)], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in os.listdir(work):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", block) or [None, "?"])[1]
            out[g("name")] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                                  scratch=g("private_segment_fixed_size"), spill=g("vgpr_spill_count"))
    shutil.rmtree(work, ignore_errors=True)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "permafrost-engine_amd", "libnavhip.so")
    for name, r in sorted(kernels(lib).items()):
        if args and not any(a in name for a in args):
            continue
        short = re.sub(r"^_Z\d+", "", name)[:44]
        print("%-46s vgpr %4s  sgpr %4s  lds %6s  scratch %5s  spilled %s" % (short, r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["spill"]))
