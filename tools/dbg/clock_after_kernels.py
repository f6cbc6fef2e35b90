"""Why the n = 27 sweep's launches are bimodal inside the BNN workloads
(VERDICT round 4, item 5): the shader clock around every launch.

    python tools/dbg/clock_after_kernels.py

configs[3]'s shard (double cartpole BNN, B = 1024, N = 150, n = 27): a round
(network kernels: matrix-core bound), then sweeps back to back with a clock
probe (pddp_debug_clock_probe: cycles per 100 MHz tick of a sleeping
wavefront) before each; the same after an idle gap and after a vector-bound
kernel.  Prints per launch: duration by events on the dispatch, and the GHz the
probe in front of it saw."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
import pddp_amd  # noqa: E402
from pddp_amd import _native  # noqa: E402
from pddp_amd.controllers.ilqr import fit_alphas  # noqa: E402
from pddp_amd.controllers.plugin import TorchProblem  # noqa: E402
from pddp_amd.controllers.solver import ILQRSolver  # noqa: E402
from pddp_amd.examples import double_cartpole as ex  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402

dev = torch.device("cuda:0")
lib = _native.lib()
raw = ctypes.CDLL(_native.LIB_PATH)
raw.pddp_debug_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
torch.manual_seed(0)
CM = ex.DoubleCartpoleDynamicsModel
B, N, D, m, P, H = 1024, 150, 6, 1, 100, 200
n = D + D * (D + 1) // 2
model = bnn_dynamics_model_factory(D, m, [H, H], CM.angular_indices,
                                   CM.non_angular_indices)(
    n_particles=P).to(dev).eval()
with torch.no_grad():
    model.model.out.weight.mul_(0.05)
    model.model.out.bias.mul_(0.05)
cost = ex.DoubleCartpoleCost().to(dev)
enc = pddp_amd.StateEncoding.DEFAULT
plugin = TorchProblem(model, cost, enc, {"use_predicted_std": False,
                                         "infer_noise_variables": True}, {})
s = ILQRSolver(None, B, N, torch.float32, dev, torch.tensor([-20.0]),
               torch.tensor([20.0]), fit_alphas(torch.float32, dev),
               plugin=plugin, n=n, m=m)
g = torch.Generator().manual_seed(0)
mean = torch.tensor([0.0, 0.0, 3.14159, 0.0, 3.14159, 0.0])
z0 = torch.stack([pddp_amd.GaussianVariable(
    mean + 1e-2 * torch.randn(D, generator=g),
    var=1e-2 * torch.ones(D)).encode(enc) for _ in range(B)]).to(dev)
s.set_nominal(z0, (0.1 * torch.randn(B, N, m, generator=g)).to(dev))
s.round(5e-6, 1e10, 1 << 30)
torch.cuda.synchronize()
probes = torch.zeros(64, 2, dtype=torch.int64, device=dev)


IN_KERNEL = hasattr(raw, "pddp_debug_mfma32s_clock")  # (-DPDDP_WG_TIMELINE)


def sweeps(count, label):
    pool = bench.EventPool(lib)
    inside = []
    for i in range(count):
        raw.pddp_debug_clock_probe(probes[i].data_ptr(),
                                   _native.stream_handle(dev))
        s.backward(active=None, variant=s.kernel_variant, events=pool.pair())
        if IN_KERNEL:  # (a host synchronisation per launch: only this build)
            b2 = (ctypes.c_longlong * 2)()
            raw.pddp_debug_mfma32s_clock(b2)
            inside.append(b2[0] / max(b2[1], 1) * 0.1)
    raw.pddp_debug_clock_probe(probes[count].data_ptr(),
                               _native.stream_handle(dev))
    torch.cuda.synchronize()
    d = np.array(pool.durations()) * 1e6
    pr = probes.cpu().numpy()[:count + 1]
    ghz = pr[:, 0] / np.maximum(pr[:, 1], 1) * 0.1
    print("%-34s us  %s" % (label, " ".join("%5.0f" % v for v in d)))
    print("%-34s GHz %s  | after the last: %.2f" % (
        "  clock in front of each", " ".join("%5.2f" % v for v in ghz[:-1]),
        ghz[-1]))
    if inside:
        print("%-34s GHz %s" % ("  clock DURING each (workgroup 0)",
                                " ".join("%5.2f" % v for v in inside)))


sweeps(8, "warm, back to back")
s.round(5e-6, 1e10, 1 << 30)      # (network launches, matrix-core bound)
sweeps(8, "right behind a round (network)")
torch.cuda.synchronize()
time.sleep(0.05)
sweeps(8, "after 50 ms of idle")
x = torch.randn(1 << 28, device=dev)
for _ in range(20):
    x = x * 1.0001 + 0.5            # (a memory-bound vector kernel, ~1 ms each)
sweeps(8, "behind 20 streaming launches")
acc = torch.zeros((), device=dev)
for _ in range(20):
    acc += x.sum()                  # (read-only streaming, 1 GB each)
sweeps(8, "behind 20 read-only launches")
a = torch.randn(8192, 8192, device=dev)
for _ in range(6):
    b = a @ a                       # (matrix-core bound, no HBM stream)
sweeps(8, "behind 6 large GEMMs")
# what precedes the sweep INSIDE a round: the derivative rollout, whose last
# launches (bnn_jvp moments, qr_cost_derivs) write the 970 MB of records
s.derivs(mask=None)
sweeps(8, "behind the derivative rollout")
s.round(5e-6, 1e10, 1 << 30)
sweeps(8, "behind a round, again")
