"""Phase times of the float32 network kernel (csrc/bnn_mlp.hip, the balanced
H = 200 form) from a debug build (-DPDDP_MLP_MARKS into a private library):
s_memtime of the eight wavefronts of workgroup 0 at the phase boundaries of
their sixth iteration.  Consumers 0 .. 6: requests + epilogue of tile i - 1 |
layer 2 of tile i | layer 1 of tile i + 1 | barrier wait; finisher (7): layer 3
of tile i - 2 | its share of layer 2 | - | barrier wait.

    python tools/mlp_marks.py
"""
import ctypes
import os
import subprocess

import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "pddp_amd", "csrc")
out = "/tmp/libpddp_mlp_marks.so"
subprocess.check_call(
    ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950",
     "-fPIC", "-shared", "-ffp-contract=fast", "-fno-fast-math", "-mllvm",
     "-amdgpu-mfma-vgpr-form=1", "-DPDDP_MLP_MARKS",
     os.path.join(src, "bnn_mlp.hip"), "-o", out])
lib = ctypes.CDLL(out)
P, I = ctypes.c_void_p, ctypes.c_int
lib.pddp_bnn_mlp_rows_f32.argtypes = [I] * 5 + [P] * 12
H, R, Pn, IN, OUT = 200, 4096000, 100, 6, 8
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
X, W1, b1, W2 = r(R, IN), r(H, IN), r(H), r(H, H) / H ** 0.5
b2, W3, b3 = r(H), r(OUT, H), r(OUT)
M1, M2 = (r(Pn, H) > 0).float(), (r(Pn, H) > 0).float()
Y = torch.empty(R, OUT, device="cuda")
p = lambda t: t.data_ptr()
for _ in range(2):
    rc = lib.pddp_bnn_mlp_rows_f32(R, Pn, IN, H, OUT, p(X), p(W1), p(b1), p(M1),
                                   p(W2), p(b2), p(M2), p(W3), p(b3), p(Y),
                                   None, None)
    assert rc == 0, rc
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
assert lib.pddp_debug_mlp_marks(buf) == 0
names = ["requests+epilogue", "layer2", "layer1", "wait"]
t0 = min(buf[w * 8] for w in range(8))
for w in range(8):
    t = [buf[w * 8 + k] for k in range(5)]
    print("wave %d: start +%d  " % (w, t[0] - t0) +
          "  ".join("%s %d" % (names[k], t[k + 1] - t[k]) for k in range(4)) +
          "  iteration %d" % (t[4] - t[0]) +
          "  (%.2f us by the 100 MHz clock: shader clock %.2f GHz)" % (
              (buf[w * 8 + 6] - buf[w * 8 + 5]) / 100.0,
              (t[4] - t[0]) / max(buf[w * 8 + 6] - buf[w * 8 + 5], 1) / 10.0))
