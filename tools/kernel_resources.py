#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in libpddp_hip.so, read from
the code object's metadata (no GPU needed):

    python tools/kernel_resources.py            # the table (CSV)
    python tools/kernel_resources.py --check    # against the committed baseline

`--check` (also run by tests/test_host_cpu.py): every kernel that spills or
uses scratch must be in profiles/kernel_resources_allowed.csv with at least
that many spilled registers / scratch bytes - a NEW spilling kernel, or one
that spills more than it did, fails.  `--write-allowed` rewrites that file from
the current build (a deliberate act: review the diff).

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


ALLOWED = os.path.join(ROOT, "profiles", "kernel_resources_allowed.csv")


def spilling(rows):
    out = {}
    for r in rows:
        sp, sc = r.get("vgpr_spill_count", 0), \
            r.get("private_segment_fixed_size", 0)
        if sp or sc:
            out[r["kernel"]] = (sp, sc)
    return out


def read_allowed(path=ALLOWED):
    import csv
    out = {}
    with open(path) as fh:
        for row in csv.reader(fh):
            if row and not row[0].startswith("#") and row[0] != "kernel":
                out[row[0]] = (int(row[1]), int(row[2]))
    return out


def check(rows=None):
    """[(kernel, spills, scratch, allowed spills, allowed scratch)] of the
    kernels that spill without being allowed to, or more than allowed."""
    rows = kernel_resources() if rows is None else rows
    allowed = read_allowed()
    bad = []
    for k, (sp, sc) in sorted(spilling(rows).items()):
        a = allowed.get(k, (0, 0))
        if sp > a[0] or sc > a[1]:
            bad.append((k, sp, sc, a[0], a[1]))
    return bad


if __name__ == "__main__" and "--write-allowed" in sys.argv:
    with open(ALLOWED, "w") as fh:
        fh.write("# kernels that may spill registers / use scratch, with the "
                 "figures of the build\n# they were reviewed at "
                 "(tools/kernel_resources.py --check)\n"
                 "kernel,vgpr_spills,scratch_bytes\n")
        for k, (sp, sc) in sorted(spilling(kernel_resources()).items()):
            fh.write('"%s",%d,%d\n' % (k, sp, sc))
    sys.exit(0)

if __name__ == "__main__" and "--check" in sys.argv:
    bad = check()
    for k, sp, sc, a0, a1 in bad:
        print("NEW / GROWN: %s spills %d (allowed %d), scratch %d B (allowed "
              "%d)" % (k, sp, a0, sc, a1))
    print("%d kernels over their allowance" % len(bad))
    sys.exit(1 if bad else 0)

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
