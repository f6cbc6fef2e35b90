"""One round of the native controller under a Gaussian encoding (known
dynamics, csrc/default_kernels.hip): ms per round and per kernel.
    python tools/default_encoding_bench.py [cartpole] [DEFAULT] [4096]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.controllers.ilqr import _make_solver, fit_alphas  # noqa: E402

problem = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
enc = pddp_amd.StateEncoding[sys.argv[2] if len(sys.argv) > 2 else "DEFAULT"]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = {"cartpole": 100, "pendulum": 50, "double_cartpole": 150}[problem]
mod = getattr(pddp_amd.examples, problem)
model = [getattr(mod, n) for n in dir(mod) if n.endswith("DynamicsModel")
         and n != "DynamicsModel"][0](0.1).cuda()
cost = [getattr(mod, n) for n in dir(mod) if n.endswith("Cost")
        and n != "AugmentedQRCost"][0]().cuda()
D, m = model.state_size, model.action_size
n = pddp_amd.utils.encoding.infer_encoded_state_size(D, enc)
bound = {"cartpole": 10.0, "pendulum": 2.5, "double_cartpole": 20.0}[problem]
s = _make_solver(model, cost, enc, B, N, n, torch.float32, "cuda",
                 torch.tensor([-bound]), torch.tensor([bound]), None)
assert s.plugin is None
g = torch.Generator().manual_seed(0)
mean = torch.zeros(D)
z0 = torch.stack([pddp_amd.GaussianVariable(
    mean + 1e-2 * torch.randn(D, generator=g), var=1e-2 * torch.ones(D)).encode(enc)
    for _ in range(64)]).repeat(B // 64 + 1, 1)[:B].cuda()
U = (0.1 * torch.randn(B, N, m, generator=g)).cuda()
s.set_nominal(z0, U)
for _ in range(3):
    s.round(5e-6, 1e10, 1 << 30)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    s.round(5e-6, 1e10, 1 << 30)
torch.cuda.synchronize()
print("%s %s n=%d B=%d N=%d: %.3f ms per round, live %d" % (
    problem, enc.name, n, B, N, (time.perf_counter() - t0) / K * 1e3,
    int(s.active.sum())))
