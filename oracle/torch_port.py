"""Op-for-op PyTorch-CPU restatement of one reference iLQR iteration on ONE
trajectory - the second CPU leg of SURVEY 8(d) ("looping over trajectories like
the reference, torch.set_num_threads(1)").

TEST INFRASTRUCTURE ONLY: used by tests/ (pinned to the reference's golden
vectors) and by bench.py's `cpu_baseline` leg; never imported by the product.

It restates, with the torch calls the reference makes (matmul, cholesky,
cholesky_solve, autograd over replicated inputs):
  forward          pddp/controllers/ilqr.py:393-486 (+ utils/evaluation.py:134-288)
  Q, backward      ilqr.py:489-526, 529-674 (four gain branches)
  boxqp            pddp/utils/constraint.py:150-266
  control_law      ilqr.py:677-723          trajectory_cost  ilqr.py:764-791
  iteration        one pass of ilqr.py:318-362 with the accept test of :330-341
The dynamics / cost plugins are the caller's torch modules.  Eigen-decomposition
of Q_uu: symmetric (`linalg.eigh`) - the reference's non-symmetric `eig` is
LAPACK-internal for repeated eigenvalues; identical for m = 1.
"""
import torch


def _clamp(u, lo, hi):
    return torch.min(torch.max(u, lo), hi)


def _derivs_dynamics(model, z, u, t, encoding, **opts):
    n, m = z.shape[-1], u.shape[-1]
    rep = torch.cat([z, u]).detach().repeat(n, 1).requires_grad_()
    zn = model(rep[:, :n], rep[:, n:], t, encoding, identical_inputs=True,
               **opts)
    J, = torch.autograd.grad(zn, rep, torch.eye(n, dtype=z.dtype))
    return zn[0].detach(), J[:, :n], J[:, n:]


def _derivs_cost(cost, z, u, t, terminal, encoding, **opts):
    n = z.shape[-1]
    zu = z if terminal else torch.cat([z, u])
    d = zu.shape[-1]
    rep = zu.detach().repeat(d, 1).requires_grad_()
    l = cost(rep[:, :n], None if terminal else rep[:, n:], t,
             terminal=terminal, encoding=encoding, identical_inputs=True,
             **opts)
    g, = torch.autograd.grad(l, rep, torch.ones_like(l), create_graph=True)
    H, = torch.autograd.grad(g, rep, torch.eye(d, dtype=z.dtype),
                             allow_unused=True)
    H = torch.zeros(d, d, dtype=z.dtype) if H is None else H
    g = g[0].detach()
    if terminal:
        return l[0].detach(), g[:n], None, H[:n, :n], None, None
    return l[0].detach(), g[:n], g[n:], H[:n, :n], H[n:, :n], H[n:, n:]


def forward(z0, U, model, cost, encoding, u_min=None, u_max=None,
            model_opts=None, cost_opts=None):
    """Derivative rollout along the nominal actions (clamped before they are
    evaluated, ilqr.py:461-462)."""
    mo, co = model_opts or {}, cost_opts or {}
    N, m = U.shape
    n = z0.shape[-1]
    kw = dict(dtype=z0.dtype)
    Z = torch.zeros(N + 1, n, **kw)
    F_z, F_u = torch.zeros(N, n, n, **kw), torch.zeros(N, n, m, **kw)
    L, L_z, L_u = torch.zeros(N + 1, **kw), torch.zeros(N + 1, n, **kw), \
        torch.zeros(N, m, **kw)
    L_zz, L_uz, L_uu = torch.zeros(N + 1, n, n, **kw), \
        torch.zeros(N, m, n, **kw), torch.zeros(N, m, m, **kw)
    Z[0] = z0
    for t in range(N):
        u = U[t] if u_min is None else _clamp(U[t], u_min, u_max)
        L[t], L_z[t], L_u[t], L_zz[t], L_uz[t], L_uu[t] = _derivs_cost(
            cost, Z[t], u, t, False, encoding, **co)
        Z[t + 1], F_z[t], F_u[t] = _derivs_dynamics(model, Z[t], u, t,
                                                    encoding, **mo)
    L[N], L_z[N], _, L_zz[N], _, _ = _derivs_cost(cost, Z[N], None, N, True,
                                                  encoding, **co)
    return Z, F_z, F_u, L, L_z, L_u, L_zz, L_uz, L_uu


def Q(F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu, V_z, V_zz):
    Q_z = L_z + F_z.t().matmul(V_z)
    Q_u = L_u + F_u.t().matmul(V_z)
    Q_zz = L_zz + F_z.t().mm(V_zz).mm(F_z)
    Q_uz = L_uz + F_u.t().mm(V_zz).mm(F_z)
    Q_uu = L_uu + F_u.t().mm(V_zz).mm(F_u)
    return Q_z, Q_u, 0.5 * (Q_zz + Q_zz.t()), Q_uz, 0.5 * (Q_uu + Q_uu.t())


def boxqp(x0, Qm, c, lower, upper, max_iter=100, min_grad=1e-8, tol=1e-8,
          step_dec=0.6, min_step=1e-22, armijo=0.1):
    """min 0.5 x'Qx + c'x on a box by projected Newton steps -> (x, result,
    Ufree, free); result < 1 is a failure (constraint.py:163-173)."""
    obj = lambda x: 0.5 * x.matmul(Qm).matmul(x) + x.matmul(c)
    m = x0.shape[0]
    x = _clamp(x0, lower, upper)
    x = torch.where(torch.isinf(x), torch.zeros_like(x), x)
    clamped = torch.zeros(m, dtype=torch.bool)
    free = ~clamped
    Ufree = torch.zeros(m, m, dtype=x.dtype)
    f, old_f, result = obj(x), None, 0
    for it in range(max_iter):
        if result != 0:
            break
        if it > 0 and (old_f - f) < tol * old_f.abs():
            result = 4
            break
        old_f = f
        g = Qm.matmul(x) + c
        old_clamped = clamped
        clamped = ((x == lower) & (g > 0)) | ((x == upper) & (g < 0))
        free = ~clamped
        if bool(clamped.all()):
            result = 6
            break
        if it == 0 or bool((old_clamped != clamped).any()):
            Qf = Qm[free][:, free]
            Ufree, info = torch.linalg.cholesky_ex(Qf, upper=True)
            if int(info) != 0:
                result = -1
                break
        if float(g[free].norm()) < min_grad:
            result = 5
            break
        g_clamped = Qm.matmul(x * clamped.to(x.dtype)) + c
        search = torch.zeros_like(x)
        search[free] = -torch.cholesky_solve(
            g_clamped[free].unsqueeze(1), Ufree, upper=True).squeeze(1) - x[free]
        sdotg = (search * g).sum()
        step = 1.0
        xc = _clamp(x + step * search, lower, upper)
        fc = obj(xc)
        while (fc - old_f) / (step * sdotg) < armijo:
            step *= step_dec
            xc = _clamp(x + step * search, lower, upper)
            fc = obj(xc)
            if step < min_step:
                result = 2
                break
        x, f = xc, fc
    return x, result, Ufree, free


def backward(F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu, reg=0.0, V_zz_reg=False,
             u_min=None, u_max=None, U=None):
    """-> (k [N, m], K [N, m, n]); RuntimeError where the reference raises."""
    N, n, m = F_u.shape
    V_z, V_zz = L_z[N], L_zz[N]
    k = torch.zeros(N, m, dtype=F_z.dtype)
    K = torch.zeros(N, m, n, dtype=F_z.dtype)
    bounded = u_min is not None and u_max is not None
    I = torch.eye(n, dtype=F_z.dtype)
    for t in range(N - 1, -1, -1):
        args = (F_z[t], F_u[t], L_z[t], L_u[t], L_zz[t], L_uz[t], L_uu[t])
        Q_z, Q_u, Q_zz, Q_uz, Q_uu = Q(*args, V_z, V_zz)
        if V_zz_reg:
            _, Qu_g, _, Quz_g, Quu_g = Q(*args, V_z, V_zz + reg * I)
        else:
            if not bool(torch.isfinite(Q_uu).all()):
                raise RuntimeError("non-finite Q_uu")
            e, E = torch.linalg.eigh(Q_uu)
            e = torch.where(e < 0, torch.full_like(e, 1e-12), e) + reg
            Qu_g, Quz_g = Q_u, Q_uz
            Quu_g = (E * e).mm(E.t())
        if not bounded:
            if V_zz_reg:
                Uc, info = torch.linalg.cholesky_ex(Quu_g, upper=True)
                if int(info) != 0:
                    raise RuntimeError("Q_uu is not positive definite")
                kK = -torch.cholesky_solve(
                    torch.cat([Qu_g.unsqueeze(1), Quz_g], 1), Uc, upper=True)
            else:
                kK = -(E / e).mm(E.t()).mm(torch.cat([Q_u.unsqueeze(1), Q_uz], 1))
                if bool(torch.isnan(kK).any()):
                    raise RuntimeError("NaN gains")
            k[t], K[t] = kK[:, 0], kK[:, 1:]
        else:
            x0 = k[min(t + 1, N - 1)]
            kt, result, Ufree, free = boxqp(x0, Quu_g, Qu_g, u_min - U[t],
                                            u_max - U[t])
            k[t] = kt
            if result < 1:
                raise RuntimeError("boxqp failed: %d" % result)
            if bool(free.any()):
                Kf = -torch.cholesky_solve(Quz_g[free], Ufree, upper=True)
                Kt = torch.zeros(m, n, dtype=F_z.dtype)
                Kt[free] = Kf
                K[t] = Kt
        # value function with the UN-regularised Q_uu, Q_uz (ilqr.py:619-625)
        Kt, kt = K[t], k[t]
        V_z = Q_z + Kt.t().matmul(Q_u) + Kt.t().mm(Q_uu).matmul(kt) \
            + Q_uz.t().matmul(kt)
        V_zz = Q_zz + Kt.t().mm(Q_uu).mm(Kt) + Kt.t().mm(Q_uz) \
            + Q_uz.t().mm(Kt)
        V_zz = 0.5 * (V_zz + V_zz.t())
    return k, K


@torch.no_grad()
def control_law(model, Z, U, k, K, alphas, encoding, u_min=None, u_max=None,
                model_opts=None):
    """All step sizes rolled out together -> Z_new [N+1, A, n], U_new [N, A, m]."""
    mo = model_opts or {}
    N, m = U.shape
    A = alphas.shape[0]
    Zn = torch.zeros(N + 1, A, Z.shape[-1], dtype=Z.dtype)
    Un = torch.zeros(N, A, m, dtype=Z.dtype)
    Zn[0] = Z[0]
    for t in range(N):
        du = alphas.unsqueeze(1) * k[t] + (Zn[t] - Z[t]).matmul(K[t].t())
        u = U[t] + du
        if u_min is not None:
            u = _clamp(u, u_min, u_max)
        Un[t] = u
        Zn[t + 1] = model(Zn[t], u, t, encoding, **mo)
    return Zn, Un


@torch.no_grad()
def trajectory_cost(cost, Z, U, encoding, cost_opts=None):
    """J [A] of candidate rollouts Z [N+1, A, n], U [N, A, m]."""
    co = cost_opts or {}
    N = U.shape[0]
    J = cost(Z[N], None, N, terminal=True, encoding=encoding, **co)
    for t in range(N):
        J = J + cost(Z[t], U[t], t, terminal=False, encoding=encoding, **co)
    return J


def iteration(z0, U, model, cost, encoding, u_min, u_max, alphas, reg=1e-3):
    """One attempt of the fit loop on one trajectory: derivative rollout,
    sweep (the controller's branch: bounds, eig clamp), line search, accept.
    Returns (U_next, J_nominal, J_best, accepted)."""
    out = forward(z0, U, model, cost, encoding, u_min, u_max)
    Z, L = out[0], out[3]
    J_opt = L.sum()
    Uc = _clamp(U, u_min, u_max)
    k, K = backward(*out[1:3], *out[4:], reg=reg, u_min=u_min, u_max=u_max,
                    U=Uc)
    Zn, Un = control_law(model, Z, Uc, k, K, alphas, encoding, u_min, u_max)
    J = trajectory_cost(cost, Zn, Un, encoding)
    Jf = torch.where(torch.isnan(J), torch.full_like(J, float("inf")), J)
    a = int(torch.argmin(Jf))
    accepted = bool(Jf[a] < J_opt)
    return (Un[:, a] if accepted else U), float(J_opt), float(Jf[a]), accepted
