"""Phase times of the float64 network kernel (csrc/bnn_mlp_f64.hip) from a debug
build (-DPDDP_MLP64_MARKS into a private library): s_memtime of the four
wavefronts of workgroup 0 at the phase boundaries of their sixth tile:
layer 1 | wait A | layer 2 | layer 3 + partial sums | wait B | finisher.

    python tools/mlp64_marks.py [H]
"""
import ctypes
import os
import subprocess
import sys

import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "pddp_amd", "csrc")
out = "/tmp/libpddp_mlp64_marks.so"
subprocess.check_call(
    ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
     "-fPIC", "-shared", "-ffp-contract=fast", "-fno-fast-math",
     "-DPDDP_MLP64_MARKS", os.path.join(src, "bnn_mlp_f64.hip"), "-o", out])
lib = ctypes.CDLL(out)
P, I = ctypes.c_void_p, ctypes.c_int
lib.pddp_bnn_mlp_rows_f64.argtypes = [I] * 5 + [P] * 12
for H in ([int(a) for a in sys.argv[1:]] or [200, 128]):
    R, Pn, IN, OUT = 4096000, 100, 6, 8
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float64, generator=g)
    X, W1, b1, W2 = r(R, IN), r(H, IN), r(H), r(H, H) / H ** 0.5
    b2, W3, b3 = r(H), r(OUT, H), r(OUT)
    M1, M2 = (r(Pn, H) > 0).double(), (r(Pn, H) > 0).double()
    Y = torch.empty(R, OUT, device="cuda", dtype=torch.float64)
    p = lambda t: t.data_ptr()
    for _ in range(2):
        rc = lib.pddp_bnn_mlp_rows_f64(R, Pn, IN, H, OUT, p(X), p(W1), p(b1), p(M1),
                                       p(W2), p(b2), p(M2), p(W3), p(b3), p(Y),
                                       None, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    assert lib.pddp_debug_mlp64_marks(buf) == 0
    names = ["layer1", "waitA", "layer2", "layer3", "waitB", "finish"]
    t0 = min(buf[w * 8] for w in range(4))
    for w in range(4):
        t = [buf[w * 8 + k] for k in range(7)]
        print("H=%d wave %d: start +%d  " % (H, w, t[0] - t0) +
              "  ".join("%s %d" % (names[k], t[k + 1] - t[k]) for k in range(6)) +
              "  tile %d" % (t[6] - t[0]))
