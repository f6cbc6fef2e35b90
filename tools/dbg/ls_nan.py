"""Debug: candidates whose cost is NaN on the device but finite in the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import oracle as orc
from test_gpu_parity import _setup
B, N = 4096, 100
s, op, z0, U, u_min, u_max = _setup("cartpole", "f32", B, N, seed=11)
s.set_nominal(torch.from_numpy(z0).cuda(), torch.from_numpy(U).cuda())
s.mu.fill_(1.0)
assert s.sweep_nominal()
k, K = s.gain_views(); k, K = k.cpu().numpy(), K.cpu().numpy()
Z0, U0 = s.Z.cpu().numpy().copy(), s.U.cpu().numpy().copy()
assert s.search_accept(5e-6, 1e10, 50, records=False)
Jc = s.Jc.cpu().numpy()
print("nan rows", int(np.isnan(Jc).any(1).sum()), "inf rows", int(np.isinf(Jc).any(1).sum()), "of", B)
o32, o64 = orc.load(np.float32), orc.load(np.float64)
al = s.alphas.cpu().numpy()
bad = np.where(~np.isfinite(Jc).all(1))[0][:6].tolist()
for b in [0] + bad:
    Zn, Un = o64.control_law(op, Z0[b], U0[b], k[b], K[b], al, u_min, u_max)
    J64 = o64.trajectory_cost(op, Zn, Un)
    Zn32, Un32 = o32.control_law(op, Z0[b], U0[b], k[b], K[b], al.astype(np.float32), u_min, u_max)
    J32 = o32.trajectory_cost(op, Zn32, Un32)
    print("b", b, "\n Jc ", Jc[b], "\n J32", J32, "\n J64", J64)
    print(" max|Z| per alpha (f64)", np.abs(Zn).max((0, 2)))
    print(" max|Z| per alpha (f32)", np.abs(Zn32).max((0, 2)))
    print(" state", int(s.state[b]), "argmin hip", np.argmin(np.nan_to_num(Jc[b], nan=-np.inf)), "argmin o32", np.argmin(J32))
# the separate line search kernel on the same gains
s2, *_ = _setup("cartpole", "f32", B, N, seed=11)
s2._nominal_sweep = False
s2.set_nominal(torch.from_numpy(z0).cuda(), torch.from_numpy(U).cuda())
s2.gains.copy_(s.gains)
s2.bwd_status.zero_()
s2.line_search()
J2 = s2.Jc.cpu().numpy()
print("separate line search: nan rows", int(np.isnan(J2).any(1).sum()))
for b in [0] + bad[:2]:
    print(b, J2[b])
