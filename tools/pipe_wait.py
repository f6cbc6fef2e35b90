"""Debug build only (make HIPFLAGS+=-DPDDP_PIPE_TIMING): which role of the
decoupled sweep kernel waits at the step barrier, in cycles per step."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pddp_amd import _native
from tools.variant_ab import make
s = make(4096, 100, torch.float32)
s.derivs()
reg = torch.full((4096,), 1.0, dtype=torch.float64, device="cuda")
lib = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * 4)()
for v in (13,):
    s.backward(reg=reg, variant=v)
    lib.pddp_debug_pipe_wait(out, 1)
    s.backward(reg=reg, variant=v)
    lib.pddp_debug_pipe_wait(out, 1)
    w = 1024 * 100.0
    print("variant", v, "barrier wait cycles/step: role Q %.0f, role M %.0f; "
          "total cycles/step: Q %.0f, M %.0f" % (out[0] / w, out[1] / w, out[2] / w, out[3] / w))
