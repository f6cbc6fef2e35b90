"""Hunts the writer that corrupts the small cached device tensors
(utils/encoding.py _triu / _eye, utils/angular.py _index_tensor) during GP
rounds.  The caches are wrapped so that every cached tensor is snapshotted on
the device after every stage (clones into a SIDE memory pool: the main pool's
layout - where the victim sits - is the unwatched run's), compared on the host
once per configuration; on a mismatch the victim's block and its surroundings
are dumped."""
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
from pddp_amd.utils import angular, encoding as enc_mod

dev = torch.device("cuda", 0)
tracked = {}   # name -> (device tensor, host reference)


origs = []


def wrap(mod, fname, host_ref):
    orig = getattr(mod, fname)

    origs.append(orig)

    def f(*a):
        t = orig(*a)
        name = "%s%s" % (fname, a[:-1])
        if name not in tracked:
            tracked[name] = (t, host_ref(*a))
        return t
    setattr(mod, fname, f)


wrap(enc_mod, "_triu", lambda D, d: torch.triu_indices(D, D))
wrap(enc_mod, "_eye", lambda D, dt, d: torch.eye(D, dtype=dt))
wrap(angular, "_index_tensor", lambda idx, d: torch.tensor(idx, dtype=torch.long))

pool = torch.cuda.MemPool()
snaps = []
live = {}      # name -> tensor, for the address map


def mark(stage):
    with torch.cuda.use_mem_pool(pool):
        snaps.append((stage, [(k, v[0].clone()) for k, v in tracked.items()]))


def hip():
    for l in open("/proc/self/maps"):
        if "libamdhip64" in l:
            return ctypes.CDLL(l.split()[-1])


def peek(ptr, nbytes):
    buf = (ctypes.c_ubyte * nbytes)()
    rc = hip().hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2)
    return rc, bytes(buf)


def report():
    torch.cuda.synchronize()
    for stage, items in snaps:
        for name, snap in items:
            ref = tracked[name][1]
            if not torch.equal(snap.cpu(), ref):
                t = tracked[name][0]
                print("CORRUPT first seen after [%s]: %s at 0x%x (%d bytes)" % (
                    stage, name, t.data_ptr(), t.numel() * t.element_size()), flush=True)
                print("  now   :", t.cpu().flatten()[:40].tolist(), flush=True)
                print("  wanted:", ref.flatten()[:40].tolist(), flush=True)
                base = t.data_ptr()
                lo = base - min(4096, base % (2 << 20))
                rc, raw = peek(lo, base - lo + 8192)
                print("  dump rc", rc, "from 0x%x" % lo, flush=True)
                import numpy as np
                a32 = np.frombuffer(raw, dtype=np.float32)
                a64 = np.frombuffer(raw, dtype=np.float64)
                i64 = np.frombuffer(raw, dtype=np.int64)
                for off in range(0, len(raw), 64):
                    print("  %+6d  f32 %s | f64 %s | i64 %s" % (
                        lo + off - base,
                        " ".join("%9.3g" % v for v in a32[off // 4: off // 4 + 16][:8]),
                        " ".join("%9.3g" % v for v in a64[off // 8: off // 8 + 8][:4]),
                        " ".join("%d" % v for v in i64[off // 8: off // 8 + 8][:4])), flush=True)
                print("  live tensors:", flush=True)
                for k, v in sorted(live.items(), key=lambda kv: kv[1].data_ptr()):
                    print("    0x%x +%-8d %s %s" % (v.data_ptr(), v.numel() * v.element_size(), k,
                                                    tuple(v.shape)), flush=True)
                sys.exit(3)
    snaps.clear()


enc = StateEncoding.DEFAULT
encs = [StateEncoding.DEFAULT, StateEncoding.VARIANCE_ONLY, StateEncoding.IGNORE_UNCERTAINTY]
for outer in range(int(os.environ.get("REPS", "10"))):
  print("== repetition", outer, flush=True)
  tracked.clear()
  for o in origs:
      o.cache_clear()
  for system in ("pendulum", "cartpole", "double_cartpole"):
      mod = getattr(ex, system)
      MC = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel") and k != "DynamicsModel"][0]
      cost_cls = [getattr(mod, k) for k in dir(mod) if k.endswith("Cost") and k not in ("AugmentedQRCost", "QRCost")][0]
      E, m = MC.state_size, 1
      for Md in (24, 21, 5):
          for dtype in (torch.float64, torch.float32):
              tag = "%s M %d %s" % (system, Md, dtype)
              g = torch.Generator().manual_seed(2)
              Xd = torch.randn(Md, E, generator=g, dtype=torch.float64)
              Ud = torch.randn(Md, m, generator=g, dtype=torch.float64)
              dXd = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
              model = gp_dynamics_model_factory(E, m, MC.angular_indices, MC.non_angular_indices)().double().cuda()
              model.fit(Xd.cuda(), Ud.cuda(), dXd.cuda())
              model = model.to(dtype).eval()
              mark(tag + " fit")
              for e2 in encs:
                  R = 70
                  z = torch.stack([GaussianVariable(0.3 * torch.randn(E, generator=g, dtype=torch.float64),
                                                    var=1e-2 * torch.ones(E, dtype=torch.float64)).encode(e2)
                                   for _ in range(R)]).to(dtype).cuda()
                  u = (0.3 * torch.randn(R, m, generator=g)).to(dtype).cuda()
                  for rep in range(4):
                      a = model.native_step(z, u, e2).clone()
                      mark(tag + " step enc %d" % int(e2))
                      b = [t.clone() for t in model.native_step(z, u, e2, jacobian=True)]
                      mark(tag + " jac enc %d" % int(e2))
                  if e2 == StateEncoding.DEFAULT:
                      n = z.shape[1]
                      B, N = 7, 3
                      bound = torch.tensor([2.0], dtype=dtype)
                      plugin = TorchProblem(model, cost_cls().to(dtype).cuda(), e2, {}, {})
                      s = ILQRSolver(None, B, N, dtype, "cuda", -bound, bound, fit_alphas(dtype, "cuda"),
                                     plugin=plugin, n=n, m=m)
                      U0 = (0.3 * torch.randn(B, N, m, generator=g)).to(dtype).cuda()
                      live.clear()
                      for k in ("Z", "U", "z0", "rec", "gains", "Zc", "Uc", "Jc", "L", "J_opt", "mu", "active",
                                "bwd_status", "alphas", "state"):
                          if torch.is_tensor(getattr(s, k, None)):
                              live["s." + k] = getattr(s, k)
                      live["z"], live["u"] = z, u
                      for k, v in model._native_cache.items():
                          for kk, vv in v[1].items():
                              live["gp." + kk] = vv
                      s.set_nominal(z[:B], U0)
                      mark(tag + " set_nominal")
                      s.derivs()
                      mark(tag + " derivs")
                      s.mu.fill_(1.0)
                      s.backward(active=s.active)
                      mark(tag + " backward")
                      if plugin._gp_line_search_ok(s):
                          for rep in range(3):
                              s.Zc.fill_(-7.0); s.Jc.fill_(-7.0); s.Uc.fill_(-7.0)
                              s.line_search(active=s.active)
                              mark(tag + " line_search")
                              rr = (s.Zc.clone(), s.Uc.clone(), s.Jc.clone())
                  report()
                  print(tag, "enc", int(e2), "clean;", len(tracked), "cached tensors", flush=True)
print("all clean")
