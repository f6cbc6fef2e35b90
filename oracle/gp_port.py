"""numpy restatement of GP moment matching (pddp_amd/models/gp.py), one input
at a time, plain loops.

TEST INFRASTRUCTURE ONLY - the checker, never the product path.

PARITY UNPINNED: the reference (anassinator/pddp) has no Gaussian-process
model (pddp/models/__init__.py:17-20), so nothing of the reference's pins
these formulas.  They restate the published algorithm - Deisenroth &
Rasmussen, "PILCO: A Model-Based and Data-Efficient Approach to Policy
Search" (ICML 2011), eqs. 14-23, and Quinonero-Candela, Girard & Rasmussen,
"Prediction at an Uncertain Input for Gaussian Processes ..." (2003) - and are
cross-checked by sampling (tests/test_gp.py).  Written independently of the
torch code: scalar loops over outputs and training points, numpy.linalg only.
"""
import numpy as np


def kernel(xa, xb, ell, sf2):
    d = (xa - xb) / ell
    return sf2 * np.exp(-0.5 * np.dot(d, d))


def condition(Xt, Y, ell, sf2, sn2, jitter=1e-8):
    """K^-1 and beta = K^-1 y per output.  Xt [M, d], Y [M, E], ell [E, d]."""
    M, E = Xt.shape[0], Y.shape[1]
    Kinv, beta = np.zeros((E, M, M)), np.zeros((E, M))
    for a in range(E):
        K = np.zeros((M, M))
        for i in range(M):
            for j in range(M):
                K[i, j] = kernel(Xt[i], Xt[j], ell[a], sf2[a])
        K += (sn2[a] + jitter) * np.eye(M)
        Kinv[a] = np.linalg.inv(K)
        beta[a] = Kinv[a] @ Y[:, a]
    return Kinv, beta


def moments(Xt, Kinv, beta, ell, sf2, sn2, m, S):
    """Mean mu [E], covariance Sig [E, E] of the increments at x~ ~ N(m, S)
    and V [d, E] = cov[x~, D] (PILCO eqs. 14-23)."""
    M, d = Xt.shape
    E = beta.shape[0]
    nu = Xt - m
    mu = np.zeros(E)
    V = np.zeros((d, E))
    q = np.zeros((E, M))
    for a in range(E):
        L = np.diag(ell[a] ** 2)
        SLi = np.linalg.inv(S + L)
        c = sf2[a] / np.sqrt(np.linalg.det(S @ np.linalg.inv(L) + np.eye(d)))
        for i in range(M):
            q[a, i] = c * np.exp(-0.5 * nu[i] @ SLi @ nu[i])
        mu[a] = beta[a] @ q[a]
        acc = np.zeros(d)
        for i in range(M):
            acc += beta[a, i] * q[a, i] * nu[i]
        V[:, a] = S @ SLi @ acc
    Sig = np.zeros((E, E))
    for a in range(E):
        iLa = np.diag(1.0 / ell[a] ** 2)
        for b in range(E):
            iLb = np.diag(1.0 / ell[b] ** 2)
            R = S @ (iLa + iLb) + np.eye(d)
            RiS = np.linalg.solve(R, S)
            dR = np.linalg.det(R)
            Q = np.zeros((M, M))
            for i in range(M):
                ka = sf2[a] * np.exp(-0.5 * nu[i] @ iLa @ nu[i])
                for j in range(M):
                    kb = sf2[b] * np.exp(-0.5 * nu[j] @ iLb @ nu[j])
                    z = iLa @ nu[i] + iLb @ nu[j]
                    Q[i, j] = ka * kb / np.sqrt(dR) * np.exp(0.5 * z @ RiS @ z)
            Sig[a, b] = beta[a] @ Q @ beta[b] - mu[a] * mu[b]
            if a == b:
                Sig[a, a] += sf2[a] - np.trace(Kinv[a] @ Q) + sn2[a]
    return mu, Sig, V


def augment(mean, covar, ai, ni):
    """Mean and covariance of [x_na, sin a_1, cos a_1, ...] for x ~ N(mean,
    covar), and the cross-covariance cov[x, features] (exact: characteristic
    function of the Gaussian / Stein's lemma)."""
    D = len(mean)
    nn, na = len(ni), len(ai)
    F = nn + 2 * na
    M = np.zeros(F)
    C = np.zeros((F, F))
    X = np.zeros((D, F))
    M[:nn] = mean[ni]
    C[:nn, :nn] = covar[np.ix_(ni, ni)]
    X[:, :nn] = covar[:, ni]
    Es, Ec = np.zeros(na), np.zeros(na)
    for k, a in enumerate(ai):
        damp = np.exp(-0.5 * covar[a, a])
        Es[k], Ec[k] = damp * np.sin(mean[a]), damp * np.cos(mean[a])
        M[nn + 2 * k], M[nn + 2 * k + 1] = Es[k], Ec[k]
        X[:, nn + 2 * k] = covar[:, a] * Ec[k]
        X[:, nn + 2 * k + 1] = -covar[:, a] * Es[k]
        C[:nn, nn + 2 * k] = covar[ni, a] * Ec[k]
        C[:nn, nn + 2 * k + 1] = -covar[ni, a] * Es[k]
        C[nn + 2 * k, :nn] = C[:nn, nn + 2 * k]
        C[nn + 2 * k + 1, :nn] = C[:nn, nn + 2 * k + 1]
    for k, a in enumerate(ai):
        for l, b in enumerate(ai):
            va, vb, c = covar[a, a], covar[b, b], covar[a, b]
            ma, mb = mean[a], mean[b]
            # E[sin a sin b] etc. from E[exp(i (s a + t b))]
            em = np.exp(-0.5 * (va + vb - 2 * c))   # a - b
            ep = np.exp(-0.5 * (va + vb + 2 * c))   # a + b
            Ess = 0.5 * (em * np.cos(ma - mb) - ep * np.cos(ma + mb))
            Ecc = 0.5 * (em * np.cos(ma - mb) + ep * np.cos(ma + mb))
            Esc = 0.5 * (ep * np.sin(ma + mb) + em * np.sin(ma - mb))
            Ecs = 0.5 * (ep * np.sin(ma + mb) - em * np.sin(ma - mb))
            C[nn + 2 * k, nn + 2 * l] = Ess - Es[k] * Es[l]
            C[nn + 2 * k + 1, nn + 2 * l + 1] = Ecc - Ec[k] * Ec[l]
            C[nn + 2 * k, nn + 2 * l + 1] = Esc - Es[k] * Ec[l]
            C[nn + 2 * k + 1, nn + 2 * l] = Ecs - Ec[k] * Es[l]
    return M, C, X


def step(Xt, Kinv, beta, ell, sf2, sn2, mean, covar, u, ai, ni):
    """Next state's mean and covariance: x' = x + D(x~), x~ = [features, u]."""
    Ma, Ca, Xa = augment(mean, covar, ai, ni)
    F, m_ = len(Ma), len(u)
    d = F + m_
    m = np.concatenate([Ma, u])
    S = np.zeros((d, d))
    S[:F, :F] = Ca
    mu, Sig, _ = moments(Xt, Kinv, beta, ell, sf2, sn2, m, S)
    # cov[x, D_a] = cov[x, x~] (S + L_a)^-1 sum_i beta q nu
    Cx = np.zeros((len(mean), len(mean)))
    nu = Xt - m
    for a in range(beta.shape[0]):
        L = np.diag(ell[a] ** 2)
        SLi = np.linalg.inv(S + L)
        c = sf2[a] / np.sqrt(np.linalg.det(S @ np.linalg.inv(L) + np.eye(d)))
        acc = np.zeros(d)
        for i in range(Xt.shape[0]):
            acc += beta[a, i] * c * np.exp(-0.5 * nu[i] @ SLi @ nu[i]) * nu[i]
        Cx[:, a] = np.concatenate([Xa, np.zeros((len(mean), m_))], 1) @ SLi @ acc
    return mean + mu, covar + Sig + Cx + Cx.T
