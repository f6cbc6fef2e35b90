#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in libpddp_hip.so, read from
the code object's metadata (no GPU needed):

    python tools/kernel_resources.py [--csv]

A kernel with private_segment (scratch) bytes keeps part of its working set in
memory - how four BNN kernels lost 2 .. 8x in round 2."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out]


def kernel_resources(lib=None):
    lib = lib or os.path.join(ROOT, "pddp_amd", "lib", "libpddp_hip.so")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so],
                       capture_output=True, text=True, check=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run(
                [os.path.join(LLVM, "llvm-readelf"), "--notes",
                 os.path.join(tmp, f)], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
                if not m:
                    continue
                key, v = m.group(1), m.group(2)
                if key == "name" and v.startswith("_Z") and not v.endswith(".kd"):
                    cur["name"] = v
                elif key in ("vgpr_count", "agpr_count", "sgpr_count",
                             "vgpr_spill_count", "sgpr_spill_count",
                             "private_segment_fixed_size",
                             "group_segment_fixed_size"):
                    cur[key] = int(v)
                elif key == "wavefront_size" and "name" in cur:
                    rows.append(cur)
                    cur = {}
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["kernel"] = n
    return rows


if __name__ == "__main__":
    rows = kernel_resources()
    rows.sort(key=lambda r: r["kernel"])
    print("kernel,vgpr,agpr,sgpr,vgpr_spills,scratch_bytes,static_lds_bytes")
    for r in rows:
        print('"%s",%d,%d,%d,%d,%d,%d' % (
            r["kernel"], r.get("vgpr_count", 0), r.get("agpr_count", 0),
            r.get("sgpr_count", 0), r.get("vgpr_spill_count", 0),
            r.get("private_segment_fixed_size", 0),
            r.get("group_segment_fixed_size", 0)))
