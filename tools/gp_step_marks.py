"""Phase times of pddp_gp_step_* from a debug build (-DPDDP_GP_MARKS: rebuilds
csrc/gp_step.hip into a private library): s_memtime of wavefront 0 at the phase
boundaries A0 | A1 | A2 | B | C | A3 | end, rows 0-7 of a launch."""
import ctypes
import os
import subprocess
import sys

import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pddp_amd", "csrc")
out = "/tmp/libpddp_gp_marks.so"
subprocess.check_call(
    ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
     "-fPIC", "-shared", "-ffp-contract=fast", "-fno-fast-math",
     "-DPDDP_GP_MARKS", os.path.join(src, "gp_step.hip"), "-o", out])
from pddp_amd import _native  # noqa: E402
from pddp_amd import StateEncoding  # noqa: E402
from gp_native_check import make, rows  # noqa: E402

lib = ctypes.CDLL(out)
P = ctypes.c_void_p
for nm in ("pddp_gp_step_f32", "pddp_gp_step_f64"):
    getattr(lib, nm).argtypes = [P, ctypes.c_int, P, P, P, P, P, P]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for dtype in (torch.float32, torch.float64):
    model = make("double_cartpole", M, dtype)
    enc = StateEncoding.DEFAULT
    z, u = rows("double_cartpole", 2048, enc, dtype)
    g = model._native_model(dtype, z.device, enc)
    for jac in (False, True):
        if not model.native_ok(z, enc, jac):
            continue
        o = torch.empty_like(z)
        Fz = torch.empty(2048, 27, 27, dtype=dtype, device="cuda") if jac else None
        Fu = torch.empty(2048, 27, 1, dtype=dtype, device="cuda") if jac else None
        fn = lib.pddp_gp_step_f32 if dtype == torch.float32 else lib.pddp_gp_step_f64
        p = _native.ptr
        for _ in range(2):
            rc = fn(ctypes.byref(g), 2048, p(z), p(u), p(o), p(Fz), p(Fu),
                    _native.stream_handle(z.device))
            assert rc == 0, rc
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 64)()
        lib.pddp_debug_gp_marks(buf)
        names = ["A0", "A1", "A2", "B", "C", "A3"]
        for r in range(2):
            t = [buf[r * 8 + k] for k in range(7)]
            print("%s M=%d jac=%-5s row %d: " % (str(dtype)[6:], M, jac, r) +
                  "  ".join("%s %d" % (names[k], t[k + 1] - t[k])
                            for k in range(6)) + "  total %d" % (t[6] - t[0]))
