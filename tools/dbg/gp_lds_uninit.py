"""Does any pddp_gp_* kernel read LDS it has not written?  Every launch is
preceded by tools/dbg/lds_poison (all CUs' LDS filled with a pattern): the
outputs after a NaN pattern and after zeros must agree bit for bit."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import torch
import pddp_amd.examples as ex
from pddp_amd import GaussianVariable, StateEncoding, _native
from pddp_amd.models.gp import gp_dynamics_model_factory
from pddp_amd.controllers.ilqr import fit_alphas
from pddp_amd.controllers.plugin import TorchProblem
from pddp_amd.controllers.solver import ILQRSolver

if not os.path.exists(os.path.join(HERE, "lds_poison.so")):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                           os.path.join(HERE, "lds_poison.hip"), "-o", os.path.join(HERE, "lds_poison.so")])
P = ctypes.CDLL(os.path.join(HERE, "lds_poison.so"))
P.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
scratch = torch.zeros(4, dtype=torch.int32, device="cuda")


def poison(pattern):
    if os.environ.get("NOPOISON"):
        return
    rc = P.lds_poison(pattern, scratch.data_ptr(), _native.stream_handle(scratch.device))
    assert rc == 0, rc


from pddp_amd.utils import angular, encoding as enc_mod
dev = torch.device("cuda", 0)
watch, marks = [], []
sent = [torch.full((64,), 1000 + i, dtype=torch.int64, device=dev) for i in range(128)]


def remember():
    for D in (2, 4, 6, 8, 9):
        watch.append((enc_mod._triu(D, dev), torch.triu_indices(D, D).to(dev), "_triu(%d)" % D))
        for dt in (torch.float32, torch.float64):
            watch.append((enc_mod._eye(D, dt, dev), torch.eye(D, dtype=dt).to(dev), "_eye(%d,%s)" % (D, dt)))
    for idx in ((0,), (1,), (2,), (0, 1), (1, 2), (0, 3), (2, 3), (1, 3), (0, 1, 4, 5), (0, 3, 4, 5), (0, 1, 3), (0, 1, 2, 3)):
        watch.append((angular._index_tensor(idx, dev), torch.tensor(idx).to(dev), "_index_tensor%s" % (idx,)))
    for i, t in enumerate(sent):
        watch.append((t, torch.full((64,), 1000 + i, dtype=torch.int64).to(dev), "sentinel %d" % i))


def mark(stage):
    """device-side comparison, no host synchronisation"""
    if not watch:
        return
    marks.append((stage, torch.stack([(t != r).any() for t, r, _ in watch])))


def report():
    torch.cuda.synchronize()
    for stage, bad in marks:
        b = bad.cpu()
        if bool(b.any()):
            names = [watch[i][2] for i in range(len(watch)) if bool(b[i])]
            print("CORRUPT first seen after", stage, ":", names[:10], flush=True)
            for i in range(len(watch)):
                if bool(b[i]):
                    print("  ", watch[i][2], watch[i][0].cpu().flatten()[:16].tolist(), flush=True)
                    break
            sys.exit(3)
    marks.clear()


if os.environ.get("WATCH"):
    remember()


def same(a, b):
    return all(torch.equal(torch.nan_to_num(x, nan=123.0), torch.nan_to_num(y, nan=123.0))
               and torch.equal(torch.isnan(x), torch.isnan(y)) for x, y in zip(a, b))


encs = [StateEncoding.DEFAULT, StateEncoding.VARIANCE_ONLY, StateEncoding.IGNORE_UNCERTAINTY]
for system in os.environ.get("SYSTEMS", "pendulum,cartpole,double_cartpole").split(","):
    mod = getattr(ex, system)
    MC = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel") and k != "DynamicsModel"][0]
    cost_cls = [getattr(mod, k) for k in dir(mod) if k.endswith("Cost") and k not in ("AugmentedQRCost", "QRCost")][0]
    E, m = MC.state_size, 1
    for Md in (24, 21, 5):
        for dtype in (torch.float64, torch.float32):
            g = torch.Generator().manual_seed(2)
            Xd = torch.randn(Md, E, generator=g, dtype=torch.float64)
            Ud = torch.randn(Md, m, generator=g, dtype=torch.float64)
            dXd = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
            model = gp_dynamics_model_factory(E, m, MC.angular_indices, MC.non_angular_indices)().double().cuda()
            model.fit(Xd.cuda(), Ud.cuda(), dXd.cuda())
            model = model.to(dtype).eval()
            mark("fit %s %s" % (system, dtype))
            for enc in encs:
                R = 70
                z = torch.stack([GaussianVariable(0.3 * torch.randn(E, generator=g, dtype=torch.float64),
                                                  var=1e-2 * torch.ones(E, dtype=torch.float64)).encode(enc)
                                 for _ in range(R)]).to(dtype).cuda()
                u = (0.3 * torch.randn(R, m, generator=g)).to(dtype).cuda()
                res = {}
                for pat in (0x7ff80000, 0, 0xffffffff, 0x3f800000):
                    poison(pat)
                    a = model.native_step(z, u, enc).clone()
                    mark("step %s %s enc %d" % (system, dtype, int(enc)))
                    poison(pat)
                    b = [t.clone() for t in model.native_step(z, u, enc, jacobian=True)] if not os.environ.get("NOJAC") else []
                    mark("jac %s %s enc %d" % (system, dtype, int(enc)))
                    res[pat] = [a] + b
                ok_f = all(same(res[0][:1], res[p][:1]) for p in res)
                ok_j = all(same(res[0][1:], res[p][1:]) for p in res)
                # the rollout kernel (DEFAULT only needs the cost's model class)
                ok_r = None
                if enc == StateEncoding.DEFAULT and not os.environ.get("NOSOLVER"):
                    n = z.shape[1]
                    B, N = 7, 3
                    bound = torch.tensor([2.0], dtype=dtype)
                    plugin = TorchProblem(model, cost_cls().to(dtype).cuda(), enc, {}, {})
                    s = ILQRSolver(None, B, N, dtype, "cuda", -bound, bound, fit_alphas(dtype, "cuda"),
                                   plugin=plugin, n=n, m=m)
                    U0 = (0.3 * torch.randn(B, N, m, generator=g)).to(dtype).cuda()
                    s.graph_rollout = bool(os.environ.get("GRAPH"))
                    s.set_nominal(z[:B], U0)
                    mark("set_nominal %s %s" % (system, dtype))
                    s.derivs()
                    mark("derivs %s %s" % (system, dtype))
                    s.mu.fill_(1.0)
                    s.backward(active=s.active)
                    mark("backward %s %s" % (system, dtype))
                    if plugin._gp_line_search_ok(s) and not os.environ.get("NOLS"):
                        rr = {}
                        for pat in (0x7ff80000, 0, 0xffffffff):
                            s.Zc.fill_(-7.0); s.Jc.fill_(-7.0); s.Uc.fill_(-7.0)
                            poison(pat)
                            s.line_search(active=s.active)
                            mark("line_search %s %s" % (system, dtype))
                            rr[pat] = (s.Zc.clone(), s.Uc.clone(), s.Jc.clone())
                        ok_r = all(same(rr[0], rr[p]) for p in rr)
                report()
                print("%-16s M %3d %-14s enc %d  step %s  jacobian %s  rollout %s" % (
                    system, Md, str(dtype), int(enc), ok_f, ok_j, ok_r), flush=True)
