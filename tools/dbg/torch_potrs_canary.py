"""torch-only: do torch.linalg.cholesky / torch.cholesky_solve on small batched
f64 matrices (what GPDynamicsModel.condition runs on the device) write outside
their outputs?  The small-block pool is filled with canaries first."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
dev = torch.device("cuda", 0)
print(torch.__version__, torch.backends.cuda.preferred_linalg_library(), flush=True)


def canaries():
    keep = []
    for r in range(3000):
        n = (64, 30, 128, 256, 100, 512)[r % 6]
        keep.append(torch.full((n,), 3 + (r % 5), dtype=torch.int64, device=dev))
    # holes for the operations' own tensors
    return [t for i, t in enumerate(keep) if i % 3]


def check(keep, what):
    torch.cuda.synchronize()
    bad = [(i, t) for i, t in enumerate(keep) if not bool((t == t.flatten()[-1]).all()) or int(t.flatten()[-1]) not in (3, 4, 5, 6, 7)]
    if bad:
        i, t = bad[0]
        print("CORRUPT by", what, ":", len(bad), "canaries; first", i, t.numel(), t.cpu()[:32].tolist(), flush=True)
        return True
    return False


def problem(E, M, dtype, g):
    X = torch.randn(M, E + 3, generator=g, dtype=torch.float64)
    d = (X[:, None] - X[None]).unsqueeze(0) / (0.5 + torch.rand(E, 1, 1, E + 3, generator=g, dtype=torch.float64))
    K = 0.01 * torch.exp(-0.5 * (d ** 2).sum(-1)) + 1e-4 * torch.eye(M, dtype=torch.float64)
    Y = 0.1 * torch.randn(M, E, generator=g, dtype=torch.float64)
    return K.to(dtype).to(dev), Y.to(dtype).to(dev)


g = torch.Generator().manual_seed(0)
found = False
for which in sys.argv[1:] or ["chol", "solve_eye", "solve_vec", "all"]:
    keep = canaries()
    for it in range(300):
        for E, M in ((2, 24), (4, 24), (6, 24), (6, 21), (4, 5), (2, 10)):
            K, Y = problem(E, M, torch.float64, g)
            if which.startswith("magma:") or which.startswith("cusolver:"):
                lib, which_ = which.split(":")
                torch.backends.cuda.preferred_linalg_library(lib)
            else:
                which_ = which
            if which_ in ("chol", "all", "chol+eye", "chol+vec", "chol+clone+eye", "chol+tri", "chol+eye_c", "chol+sync+eye", "chol+inv"):
                L = torch.linalg.cholesky(K)
            else:
                L = torch.linalg.cholesky(K.cpu()).to(dev)
            if which_ == "chol+clone+eye":
                L = L.contiguous().clone()
            if which_ == "chol+sync+eye":
                torch.cuda.synchronize()
            if which_ in ("solve_eye", "all", "chol+eye", "chol+clone+eye", "chol+sync+eye"):
                eye = torch.eye(M, dtype=K.dtype, device=dev).expand(E, M, M)
                Kinv = torch.cholesky_solve(eye, L)
            if which_ == "chol+eye_c":
                eye = torch.eye(M, dtype=K.dtype, device=dev).expand(E, M, M).contiguous()
                Kinv = torch.cholesky_solve(eye, L)
            if which_ in ("solve_vec", "all", "chol+vec"):
                beta = torch.cholesky_solve(Y.t().unsqueeze(-1), L).squeeze(-1)
            if which_ == "helper":   # pddp_amd.utils.linalg.cholesky_solve, as GPDynamicsModel.condition calls it
                from pddp_amd.utils.linalg import cholesky_solve
                L = torch.linalg.cholesky(K)
                eye = torch.eye(M, dtype=K.dtype, device=dev).expand(E, M, M)
                Kinv = cholesky_solve(eye, L)
                beta = cholesky_solve(Y.t().unsqueeze(-1), L).squeeze(-1)
            if which_ == "others":   # the other dense solves the package runs on the device
                d = min(E + 3, M)
                S = K[:, :d, :d].unsqueeze(0).expand(5, E, d, d) + torch.eye(d, dtype=K.dtype, device=dev)
                sol = torch.linalg.solve(S, Y[:d, :1].expand(5, E, d, M).contiguous()[..., :7])
                det = torch.linalg.det(S)
                U, info = torch.linalg.cholesky_ex(S, upper=True)
                eps = torch.linalg.solve_triangular(U, sol, upper=True)
                A3 = S[..., :3, :3].reshape(-1, 3, 3)
                x3 = torch.linalg.solve(A3, torch.ones(A3.shape[0], 3, 1, dtype=K.dtype, device=dev))
                for dt in (torch.float32,):
                    torch.linalg.solve(S.to(dt), sol.to(dt)); torch.linalg.det(S.to(dt)); torch.linalg.cholesky_ex(S.to(dt))
            if which_ == "eye+vec":   # (L from the host)
                eye = torch.eye(M, dtype=K.dtype, device=dev).expand(E, M, M)
                Kinv = torch.cholesky_solve(eye, L)
                beta = torch.cholesky_solve(Y.t().unsqueeze(-1), L).squeeze(-1)
            if which_ == "chol+inv":
                Kinv = torch.cholesky_inverse(L)
                beta = (Kinv @ Y.t().unsqueeze(-1)).squeeze(-1)
            if which_ == "chol+tri":
                eye = torch.eye(M, dtype=K.dtype, device=dev).expand(E, M, M)
                W = torch.linalg.solve_triangular(L, eye, upper=False)
                Kinv = W.transpose(-1, -2) @ W
                w = torch.linalg.solve_triangular(L, Y.t().unsqueeze(-1), upper=False)
                beta = torch.linalg.solve_triangular(L.transpose(-1, -2), w, upper=True).squeeze(-1)
        if it % 20 == 19 and check(keep, which + " (iteration %d)" % it):
            found = True
            print("L strides", L.stride(), L.shape, flush=True)
            break
    print(which, "done", flush=True)
print("found" if found else "clean")
