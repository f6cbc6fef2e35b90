"""Numpy prototype of the DEFERRED rank-one form of the n = 4, m = 1 backward
sweep (riccati_n4_defer.hpp restates exactly this schedule on the GPU).

The reference's step (pddp/controllers/ilqr.py:489-526, 629-672) is

    Q*_t from V_{t+1};  BoxQP -> k_t, s_t (K_t = -s_t Quz_t);
    V_t  = sym(Qzz_t) + c_t Quz_t^T Quz_t,   c_t = s_t (s_t Quu_t - 2)
    vz_t = Qz_t + w_t Quz_t,                 w_t = k_t - s_t (Qu_t + Quu_t k_t)

Everything before the BoxQP is LINEAR in (V_{t+1}, vz_{t+1}); the only
non-linear link between steps is the scalar pair (c_t, w_t).  Writing

    V_{t+1}  = W_{t+1} + sum_{j in P} c_j y_j y_j^T      P = {t+1 .. t+A}
    vz_{t+1} = r_{t+1} + sum_{j in P} w_j y_j            y_j = Quz_j carried to t+1

the 4x4 products (role M) run on W alone and absorb a rank-one term only A
steps after it was born, the vectors y_j are carried by role Y, and the scalar
chain (role Q) sees the young terms through dot products g_{j,t} = f_t . y_j:

    Quu_t = A00_t + sum_j c_j g_{j,t}^2      Qu_t = B00_t + sum_j w_j g_{j,t}
    Quz_t = Quz0_t + sum_j c_j g_{j,t} (F_t^T y_j)

This file checks that schedule (who knows what in which phase) against the
oracle's backward() in float64, and measures the float32 error of both forms
against float64.  Test infrastructure (imports oracle/): never on the product
path.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def boxqp1(x0, Q, c, lo, hi, dt):
    """utils/constraint.py:150-266 for D = 1 in dtype dt; returns x, free."""
    one = dt(1)
    x = min(max(dt(x0), lo), hi)
    if np.isinf(x):
        x = dt(0)
    f = dt(0.5) * ((x * Q) * x) + x * c
    old_f = dt(0)
    free = True
    result = 0
    U = np.sqrt(Q)
    for i in range(100):
        if i > 0 and (old_f - f) < dt(1e-8) * abs(old_f):
            result = 4
            break
        old_f = f
        g = Q * x + c
        clamped = (x == lo and g > 0) or (x == hi and g < 0)
        free = not clamped
        if clamped:
            result = 6
            break
        if i == 0 and not (Q > 0):
            result = -1
            break
        if abs(g) < dt(1e-8):
            result = 5
            break
        search = -((c / U) / U) - x
        sdotg = search * g
        step = one
        xc = min(max(x + step * search, lo), hi)
        fc = dt(0.5) * ((xc * Q) * xc) + xc * c
        while (fc - old_f) / (step * sdotg) < dt(0.1):
            step = step * dt(0.6)
            xc = min(max(x + step * search, lo), hi)
            fc = dt(0.5) * ((xc * Q) * xc) + xc * c
            if step < dt(1e-22):
                result = 2
                break
        if result == 2:
            break
        x, f = xc, fc
    return x, free, result


def scalars(kprev, Quu, Qu, Un, reg, umin, umax, dt):
    """BoxQP of one step and the rank-one coefficients (k, s, c, w)."""
    e = dt(1e-12) if Quu < 0 else Quu
    qpQ = e + reg
    k, free, res = boxqp1(kprev, qpQ, Qu, umin - Un, umax - Un, dt)
    s = one_over(qpQ, dt) if free else dt(0)
    c = s * (s * Quu - dt(2))
    w = k - s * (Qu + Quu * k)
    return k, s, c, w, res


def one_over(x, dt):
    return dt(1) / x


def plain(F, f, Lzz, Luz, Luu, Lz, Lu, U, LzN, LzzN, reg, umin, umax, dt):
    """The step as riccati_n4_qpipe.hpp runs it (one pending term)."""
    N = F.shape[0]
    V, vz = LzzN.astype(dt), LzN.astype(dt)
    k_out, K_out = np.zeros(N, dt), np.zeros((N, 4), dt)
    kprev = dt(0)
    for t in range(N - 1, -1, -1):
        T = V @ F[t]
        Qzz = Lzz[t] + F[t].T @ T
        Quz = Luz[t] + f[t] @ T
        Quu = Luu[t] + f[t] @ (V @ f[t])
        Qz = Lz[t] + F[t].T @ vz
        Qu = Lu[t] + f[t] @ vz
        k, s, c, w, _ = scalars(kprev, Quu, Qu, U[t], reg, umin, umax, dt)
        k_out[t], K_out[t] = k, -s * Quz
        V = dt(0.5) * (Qzz + Qzz.T) + c * np.outer(Quz, Quz)
        vz = Qz + w * Quz
        kprev = k
    return k_out, K_out


def deferred(F, f, Lzz, Luz, Luu, Lz, Lu, U, LzN, LzzN, reg, umin, umax, dt,
             A=2):
    """A pending rank-one terms.  Written per STEP (the phases of the GPU
    kernel only re-time these statements): for step t everything marked [M]
    uses W_{t+1} (absorbed up to c_{t+1+A}), [Y] the carried vectors, [Q] the
    scalars."""
    N = F.shape[0]
    W, r = LzzN.astype(dt), LzN.astype(dt)     # W_N, r_N
    # pending terms of V_{t+1}: list of (c_j, w_j, y_j at time t+1), youngest first
    pend = []
    k_out, K_out = np.zeros(N, dt), np.zeros((N, 4), dt)
    kprev = dt(0)
    for t in range(N - 1, -1, -1):
        Ft, ft = F[t], f[t]
        # [M] products on W_{t+1}
        T = W @ Ft
        S0 = dt(0.5) * ((Lzz[t] + Ft.T @ T) + (Lzz[t] + Ft.T @ T).T)
        Quz0 = Luz[t] + ft @ T
        A00 = Luu[t] + ft @ (W @ ft)
        # [Y] vectors on r_{t+1} and the pending y_j
        r0 = Lz[t] + Ft.T @ r
        B00 = Lu[t] + ft @ r
        g = [ft @ y for (_, _, y) in pend]           # g_{j,t}
        yh = [Ft.T @ y for (_, _, y) in pend]        # y_j carried to time t
        # [Q] scalars
        Quu, Qu, Quz = A00, B00, Quz0.copy()
        for (cj, wj, _), gj, yj in zip(pend, g, yh):
            Quu = Quu + cj * (gj * gj)
            Qu = Qu + wj * gj
            Quz = Quz + (cj * gj) * yj
        k, s, c, w, _ = scalars(kprev, Quu, Qu, U[t], reg, umin, umax, dt)
        k_out[t], K_out[t] = k, -s * Quz
        kprev = k
        # new representation at time t: absorb the oldest term once A are pending
        newp = [(c, w, Quz)] + [(cj, wj, yj) for (cj, wj, _), yj in zip(pend, yh)]
        W, r = S0, r0
        while len(newp) > A:
            cj, wj, yj = newp.pop()
            W = W + cj * np.outer(yj, yj)
            r = r + wj * yj
        pend = newp
    return k_out, K_out


class Exchange:
    """LDS exchange buffers of the kernel: a value published in phase p may
    be read in phase p + 1 only (one barrier per phase, two parities)."""

    def __init__(self):
        self.d = {}

    def put(self, name, p, v):
        self.d[name, p & 1] = (p, v)

    def get(self, name, p):
        stamp, v = self.d[name, (p - 1) & 1]
        assert stamp == p - 1, (name, p, stamp)
        return v


def scheduled(F, f, Lzz, Luz, Luu, Lz, Lu, U, LzN, LzzN, reg, umin, umax, dt):
    """deferred(A=2) re-timed into the kernel's phases.  In phase p role Q
    solves the BoxQP of step tq = N + 1 - p, role Y finalises the vector
    y_tq and prepares the scalars of step tq - 2, role M forms the products
    of step tq - 2; every role reads only what was published in phase p - 1
    (Exchange asserts it) or its own registers."""
    N = F.shape[0]
    z4, zero = np.zeros(4, dt), dt(0)
    X = Exchange()
    k_out, K_out = np.zeros(N, dt), np.zeros((N, 4), dt)

    def rec_ok(t):
        return 0 <= t <= N - 1
    # ---- registers
    # M
    S0_prev = LzzN.astype(dt)             # S0_{tm+1} ("S0_N" = W_N)
    # Y: y of the vector finalised last phase (y_{tq+1}), its carry to tq and
    # tq - 1 and dots; partial y'_{tq}; r
    Yr = dict(y1=z4, y1c=z4, y1cc=z4, g1a=zero, g1b=zero,  # vector tq+1
              yp=z4,                                        # y'_{tq}
              r0=LzN.astype(dt),                            # r0_{tq-1}... see below
              )
    r0_next = LzN.astype(dt)   # r0_{t+1} of the package being prepared
    # Q
    Qr = dict(kprev=zero, c1=zero, w1=zero, c2=zero, w2=zero,
              A0p=zero, g1=zero, B0p=zero, gprev=zero,
              have=False)
    X.put("q", -1, (zero, zero, zero, zero))      # (k, s, c, w) of "step N+2"
    X.put("zy", -1, (z4,))                        # Quz0 from M
    X.put("ym", -1, (z4,))                        # carried vector for M's absorb
    X.put("yq", -1, (zero, zero, zero))           # (G0, g2, B00) for Q
    X.put("mq", -1, (zero,))                      # A00 for Q
    for p in range(0, N + 2):
        tq = N + 1 - p
        t = tq - 2                                   # step prepared by M and Y
        kq, sq, cq, wq = X.get("q", p)               # of step tq + 1
        # ================================================================ M
        # absorb c_{tq+1} into S0_{t+1} with the vector carried to t + 1
        (yab,) = X.get("ym", p)
        W = S0_prev + cq * np.outer(yab, yab)        # W_{t+1}
        if rec_ok(t):
            T = W @ F[t]
            Cq = Lzz[t] + F[t].T @ T
            S0 = dt(0.5) * (Cq + Cq.T)
            Quz0 = Luz[t] + f[t] @ T
            A00 = Luu[t] + f[t] @ (W @ f[t])
            X.put("zy", p, (Quz0,))
            X.put("mq", p, (A00,))
            S0_prev = S0
        else:                                        # tail: nothing left for M
            X.put("zy", p, (z4,))
            X.put("mq", p, (zero,))
        # ================================================================ Y
        # (vi) gains of step tq + 1: its s arrived, its y was finalised last phase
        if rec_ok(tq + 1):
            k_out[tq + 1] = kq
            K_out[tq + 1] = -sq * Yr["y1"]
        # (i) finalise y_tq = y'_tq + (c_{tq+1} g_{tq+1,tq}) y_{tq+1} carried to tq
        y = Yr["yp"] + (cq * Yr["g1a"]) * Yr["y1c"]
        # (ii) carry y_tq to tq - 1 and tq - 2, dots with f_{tq-1}, f_{tq-2}
        if rec_ok(tq - 1) and rec_ok(tq):
            yc = F[tq - 1].T @ y
            ga = f[tq - 1] @ y                        # g_{tq,tq-1}
        else:
            yc, ga = z4, zero
        if rec_ok(t) and rec_ok(tq):
            ycc = F[t].T @ yc
            gb = f[t] @ yc                            # g_{tq,tq-2}
        else:
            ycc, gb = z4, zero
        # (iii) y'_{tq-1} = Quz0_{tq-1} + (c_{tq+1} g_{tq+1,tq-1}) y_{tq+1} carried to tq-1
        (Quz0_in,) = X.get("zy", p)                   # Quz0_{tq-1} (M, last phase)
        yp_new = Quz0_in + (cq * Yr["g1b"]) * Yr["y1cc"]
        G0 = f[t] @ yp_new if rec_ok(t) else zero     # f_{tq-2} . y'_{tq-1}
        # (iv) r_{tq-1} = r0_{tq-1} + w_{tq+1} y_{tq+1} carried to tq - 1
        r = r0_next + wq * Yr["y1cc"]
        if rec_ok(t):
            B00 = Lu[t] + f[t] @ r
            r0_next = Lz[t] + F[t].T @ r
        else:
            B00 = zero
        X.put("yq", p, (G0, gb, B00))
        X.put("ym", p, (ycc,))       # y_tq carried to tq-2 = t: absorbed into S0_t next phase
        Yr = dict(y1=y, y1c=yc, y1cc=ycc, g1a=ga, g1b=gb, yp=yp_new)
        # ================================================================ Q
        # on entry: c1/w1 = (c, w) of step tq + 1 (own registers; == cq, wq)
        if rec_ok(tq):
            Quu = Qr["A0p"] + Qr["c1"] * (Qr["g1"] * Qr["g1"])
            Qu = Qr["B0p"] + Qr["w1"] * Qr["g1"]
            k, s, c, w, _ = scalars(Qr["kprev"], Quu, Qu, U[tq], reg, umin,
                                    umax, dt)
        else:
            k = s = c = w = zero
        # off the chain: next step's coefficients from what M and Y published
        # last phase ((A00, G0, g2, B00) of step tq - 1) and c_{tq+1}
        (A00_in,) = X.get("mq", p)
        G0_in, g2_in, B00_in = X.get("yq", p)
        gg = Qr["g1"] * g2_in                 # g_{tq+1,tq} g_{tq+1,tq-1}
        g1n = G0_in + Qr["c1"] * gg
        A0pn = A00_in + Qr["c1"] * (g2_in * g2_in)
        B0pn = B00_in + Qr["w1"] * g2_in
        Qr = dict(kprev=k, c1=c, w1=w, A0p=A0pn, g1=g1n, B0p=B0pn)
        X.put("q", p, (k, s, c, w))
    # epilogue: gains of step 0
    kq, sq, cq, wq = X.get("q", N + 2)
    y0 = Yr["yp"] + (cq * Yr["g1a"]) * Yr["y1c"]
    del y0
    k_out[0] = kq
    K_out[0] = -sq * Yr["y1"]
    return k_out, K_out


def main():
    import oracle as orc
    o = orc.load(np.float64)
    op = orc.make_problem("cartpole", 0.1)
    rng = np.random.RandomState(0)
    N = 100
    umin, umax = np.array([-10.0]), np.array([10.0])
    worst = {}
    for trial in range(int(os.environ.get('TRIALS', '24'))):
        z0 = np.array([0, 0, 0, 0.0]) + 1e-2 * rng.randn(4)
        scale = [0.1, 1.0, 6.0, 12.0][trial % 4]   # the last two saturate
        U = scale * rng.randn(N, 1)
        fw = o.forward(op, z0, U, umin, umax)
        reg = [1.0, 1e-6, 1e2][trial % 3]
        kr, Kr, st = o.backward(fw["F_z"], fw["F_u"], fw["L_z"], fw["L_u"],
                                fw["L_zz"], fw["L_uz"], fw["L_uu"], reg=reg,
                                u_min=umin, u_max=umax, U=U)
        if st != 0:
            continue
        Uc = U[:, 0]
        for dt in (np.float64, np.float32):
            args = (fw["F_z"][:N].astype(dt), fw["F_u"][:N, :, 0].astype(dt),
                    fw["L_zz"][:N].astype(dt), fw["L_uz"][:N, 0].astype(dt),
                    fw["L_uu"][:N, 0, 0].astype(dt), fw["L_z"][:N].astype(dt),
                    fw["L_u"][:N, 0].astype(dt), Uc.astype(dt),
                    fw["L_z"][N].astype(dt), fw["L_zz"][N].astype(dt),
                    dt(reg), dt(umin[0]), dt(umax[0]), dt)
            res = {"plain": plain(*args)}
            for A in (1, 2, 3):
                res["defer%d" % A] = deferred(*args, A=A)
            res["sched2"] = scheduled(*args)
            for name, (k, K) in res.items():
                ek = np.abs(k - kr[:, 0]).max() / max(np.abs(kr).max(), 1e-30)
                eK = np.abs(K - Kr[:, 0]).max() / max(np.abs(Kr).max(), 1e-30)
                key = (name, dt.__name__)
                worst[key] = max(worst.get(key, 0.0), ek, eK)
    for key in sorted(worst):
        print("%-8s %-8s worst rel err vs oracle f64: %.3e" % (key[0], key[1],
                                                              worst[key]))


if __name__ == "__main__":
    main()
