"""Per-row error of pddp_bnn_mlp_f64 against torch (H = 200)."""
import sys
import torch
sys.path[:0] = ["."]
from pddp_amd.models.bnn import BayesianMLP
for rows in (1, 37):
    torch.manual_seed(200 + rows)
    net = BayesianMLP(6, 8, [200, 200]).cuda().double().eval()
    x = torch.randn(rows, 100, 6, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        y = net(x)
        y2 = net(x)
        net.use_native = False
        ref = net(x)
    e = (y - ref).abs().reshape(-1, 8).max(1).values
    print("rows", rows, "repeatable", bool(torch.equal(y, y2)), "max err", float(e.max()))
    bad = (e > 1e-12).nonzero().flatten().tolist()
    print(" bad rows:", len(bad), bad[:40])
    eo = (y - ref).abs().reshape(-1, 8).max(0).values
    print(" per output:", ["%.1e" % v for v in eo.tolist()])
