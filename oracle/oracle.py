"""ctypes binding of oracle/_build/libpddp_oracle.so (numpy in / numpy out).

TEST INFRASTRUCTURE ONLY - the checker, never the product path.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libpddp_oracle.so")

MAX_AUG, MAX_ACTION, MAX_PARAMS = 8, 4, 8

MODEL_IDS = {"cartpole": 1, "double_cartpole": 2, "pendulum": 3,
             "rendezvous": 4}
PROBLEM_NAMES = tuple(MODEL_IDS)
ENC_IGNORE, ENC_DEFAULT = 4, 1


class PddpProblem(ctypes.Structure):
    """Mirror of include/pddp_problem.h."""
    _fields_ = [
        ("model", ctypes.c_int), ("encoding", ctypes.c_int),
        ("state_size", ctypes.c_int), ("action_size", ctypes.c_int),
        ("encoded_size", ctypes.c_int), ("aug_size", ctypes.c_int),
        ("params", ctypes.c_double * MAX_PARAMS),
        ("Q", ctypes.c_double * (MAX_AUG * MAX_AUG)),
        ("Q_term", ctypes.c_double * (MAX_AUG * MAX_AUG)),
        ("R", ctypes.c_double * (MAX_ACTION * MAX_ACTION)),
        ("x_goal", ctypes.c_double * MAX_AUG),
        ("u_goal", ctypes.c_double * MAX_ACTION),
    ]


def make_problem(name, dt, encoding=ENC_IGNORE):
    """Builds the constants of a reference sample problem, float32-rounded
    exactly as the reference's constructors leave them:
    cartpole/model.py:37-54 + cartpole/cost.py:32-58, pendulum/model.py:38-51 +
    pendulum/cost.py:32-60, double_cartpole/model.py:36-66 +
    double_cartpole/cost.py:32-66, rendezvous/model.py:37-48 +
    rendezvous/cost.py:29-43."""
    p = PddpProblem()
    p.model = MODEL_IDS[name]
    p.encoding = int(encoding)
    f = np.float32
    if name == "cartpole":
        D, m, na = 4, 1, 5
        params = [dt, 0.5, 0.5, 0.5, 0.1, 9.82]
        Q = np.zeros((na, na), f)
        Q[0, 0] = 1.0
        Q[0, 3] = Q[3, 0] = 0.5
        Q[3, 3] = Q[4, 4] = f(0.5**2)
        Qt = np.eye(na, dtype=f)
        goal = np.array([0, 0, 0, np.sin(f(np.pi)), np.cos(f(np.pi))], f)
    elif name == "pendulum":
        D, m, na = 2, 1, 3
        params = [dt, 1.0, 1.0, 0.1, 9.80665]
        Q = np.zeros((na, na), f)
        Q[0, 0] = 1.0
        Q[0, 1] = Q[1, 0] = 0.5
        Q[1, 1] = Q[2, 2] = f(0.5**2)
        Qt = (100 * np.eye(na)).astype(f)
        goal = np.array([0, np.sin(f(np.pi)), np.cos(f(np.pi))], f)
    elif name == "double_cartpole":
        D, m, na = 6, 1, 8
        params = [dt, 0.5, 0.5, 0.5, 0.6, 0.6, 0.1, 9.80665]
        C = np.array([[1, -0.6, 0, -0.6, 0], [0, 0, 0.6, 0, 0.6]], f)
        dims = [0, 4, 5, 6, 7]
        Q = np.zeros((na, na), f)
        Q[np.ix_(dims, dims)] = C.T @ C
        Qt = (100 * np.eye(na)).astype(f)
        goal = np.array([0, 0, 0, 0, 0, 1, 0, 1], f)
    elif name == "rendezvous":
        D, m, na = 8, 4, 8
        params = [dt, 1.0, 0.1]
        Q = np.eye(na, dtype=f)
        Q[0, 2] = Q[2, 0] = -1
        Q[1, 3] = Q[3, 1] = -1
        Qt = Q.copy()
        goal = np.zeros(na, f)
    else:
        raise ValueError(name)
    R = (f(0.1) * np.eye(m, dtype=f)).astype(f)
    p.state_size, p.action_size, p.aug_size = D, m, na
    if encoding == ENC_IGNORE:
        p.encoded_size = D
    elif encoding == ENC_DEFAULT:
        p.encoded_size = D + D * (D + 1) // 2
    else:
        raise NotImplementedError(encoding)
    for i, v in enumerate(params):
        p.params[i] = float(f(v))
    for i in range(na):
        p.x_goal[i] = float(goal[i])
        for j in range(na):
            p.Q[i * MAX_AUG + j] = float(Q[i, j])
            p.Q_term[i * MAX_AUG + j] = float(Qt[i, j])
    for i in range(m):
        p.u_goal[i] = 0.0
        for j in range(m):
            p.R[i * MAX_ACTION + j] = float(R[i, j])
    return p


def build(force=False):
    """Compiles the C restatement (gcc). Building the checker is not using
    it."""
    srcs = ("pddp_oracle.c", "pddp_oracle_impl.inc", "pddp_oracle.h")
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, s)) > os.path.getmtime(LIB_PATH)
            for s in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Oracle(object):
    """Thin numpy facade over the C entry points, one per dtype."""

    def __init__(self, lib, dtype):
        self.lib = lib
        self.dtype = np.dtype(dtype)
        self.suffix = "f64" if self.dtype == np.float64 else "f32"

    def _fn(self, name):
        fn = getattr(self.lib, "pddp_oracle_%s_%s" % (name, self.suffix))
        fn.restype = ctypes.c_int
        return fn

    def _a(self, x, shape=None):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def dynamics(self, p, z, u, jac=True):
        n, m = p.encoded_size, p.action_size
        z, u = self._a(z), self._a(u)
        zn = np.empty(n, self.dtype)
        Fz = np.empty((n, n), self.dtype) if jac else None
        Fu = np.empty((n, m), self.dtype) if jac else None
        rc = self._fn("dynamics")(ctypes.byref(p), _ptr(z), _ptr(u), _ptr(zn),
                                  _ptr(Fz), _ptr(Fu))
        assert rc == 0, rc
        return zn, Fz, Fu

    def cost(self, p, z, u, terminal=False):
        n, m = p.encoded_size, p.action_size
        z = self._a(z)
        u = None if terminal else self._a(u)
        l = np.empty(1, self.dtype)
        lz = np.empty(n, self.dtype)
        lzz = np.empty((n, n), self.dtype)
        lu = luz = luu = None
        if not terminal:
            lu = np.empty(m, self.dtype)
            luz = np.empty((m, n), self.dtype)
            luu = np.empty((m, m), self.dtype)
        rc = self._fn("cost")(ctypes.byref(p), _ptr(z), _ptr(u),
                              int(terminal), _ptr(l), _ptr(lz), _ptr(lu),
                              _ptr(lzz), _ptr(luz), _ptr(luu))
        assert rc == 0, rc
        return l[0], lz, lu, lzz, luz, luu

    def forward(self, p, z0, U, u_min=None, u_max=None):
        n, m = p.encoded_size, p.action_size
        U = self._a(U)
        N = U.shape[0]
        z0, u_min, u_max = self._a(z0), self._a(u_min), self._a(u_max)
        d = self.dtype
        out = dict(Z=np.empty((N + 1, n), d), F_z=np.empty((N, n, n), d),
                   F_u=np.empty((N, n, m), d), L=np.empty(N + 1, d),
                   L_z=np.empty((N + 1, n), d), L_u=np.empty((N, m), d),
                   L_zz=np.empty((N + 1, n, n), d),
                   L_uz=np.empty((N, m, n), d), L_uu=np.empty((N, m, m), d))
        rc = self._fn("forward")(
            ctypes.byref(p), _ptr(z0), _ptr(U), N, _ptr(u_min), _ptr(u_max),
            *[_ptr(out[k]) for k in ("Z", "F_z", "F_u", "L", "L_z", "L_u",
                                      "L_zz", "L_uz", "L_uu")])
        assert rc == 0, rc
        return out

    def boxqp(self, x0, Q, c, lower, upper):
        x0, Q, c = self._a(x0), self._a(Q), self._a(c)
        lower, upper = self._a(lower), self._a(upper)
        m = x0.shape[0]
        x = np.empty(m, self.dtype)
        Uf = np.zeros(m * m, self.dtype)
        free = np.zeros(m, np.uint8)
        result = self._fn("boxqp")(m, _ptr(x0), _ptr(Q), _ptr(c), _ptr(lower),
                                   _ptr(upper), _ptr(x), _ptr(Uf), _ptr(free))
        return x, result, Uf, free

    def backward(self, F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu, reg=0.0,
                 V_zz_reg=False, u_min=None, u_max=None, U=None):
        F_z, F_u = self._a(F_z), self._a(F_u)
        N, n, m = F_u.shape
        args = [self._a(x) for x in (L_z, L_u, L_zz, L_uz, L_uu)]
        u_min, u_max, U = self._a(u_min), self._a(u_max), self._a(U)
        k = np.empty((N, m), self.dtype)
        K = np.empty((N, m, n), self.dtype)
        fn = self._fn("backward")
        fn.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 7 + [
            ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 5
        status = fn(n, m, N, _ptr(F_z), _ptr(F_u), *[_ptr(a) for a in args],
                    float(reg), int(V_zz_reg), _ptr(u_min), _ptr(u_max),
                    _ptr(U), _ptr(k), _ptr(K))
        return k, K, status

    def control_law(self, p, Z, U, k, K, alphas, u_min=None, u_max=None):
        n, m = p.encoded_size, p.action_size
        Z, U, k, K = self._a(Z), self._a(U), self._a(k), self._a(K)
        alphas = self._a(alphas)
        u_min, u_max = self._a(u_min), self._a(u_max)
        N, A = U.shape[0], alphas.shape[0]
        Zn = np.empty((N + 1, A, n), self.dtype)
        Un = np.empty((N, A, m), self.dtype)
        rc = self._fn("control_law")(
            ctypes.byref(p), N, A, _ptr(Z), _ptr(U), _ptr(k), _ptr(K),
            _ptr(alphas), _ptr(u_min), _ptr(u_max), _ptr(Zn), _ptr(Un))
        assert rc == 0, rc
        return Zn, Un

    def trajectory_cost(self, p, Z_new, U_new):
        Z_new, U_new = self._a(Z_new), self._a(U_new)
        N, A = U_new.shape[0], U_new.shape[1]
        J = np.empty(A, self.dtype)
        rc = self._fn("trajectory_cost")(ctypes.byref(p), N, A, _ptr(Z_new),
                                         _ptr(U_new), _ptr(J))
        assert rc == 0, rc
        return J

    def fit(self, p, z0, U, alphas, n_iterations=50, tol=5e-6, max_reg=1e10,
            u_min=None, u_max=None, max_trace=4096):
        n, m = p.encoded_size, p.action_size
        U = self._a(U).copy()
        N = U.shape[0]
        z0, alphas = self._a(z0), self._a(alphas)
        u_min, u_max = self._a(u_min), self._a(u_max)
        Z = np.empty((N + 1, n), self.dtype)
        K = np.zeros((N, m, n), self.dtype)
        trace = np.zeros((max_trace, 5), np.float64)
        nt = ctypes.c_int(0)
        fn = self._fn("fit")
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [
            ctypes.c_double] * 2 + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [
                ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
        state = fn(ctypes.addressof(p), _ptr(z0), _ptr(U), N,
                   int(n_iterations), float(tol), float(max_reg), _ptr(u_min),
                   _ptr(u_max), _ptr(alphas), alphas.shape[0], _ptr(Z),
                   _ptr(K), _ptr(trace), max_trace, ctypes.addressof(nt))
        return Z, U, K, state, trace[:nt.value]


_LIB = None


def load(dtype=np.float64):
    """PDDP_ORACLE_LIB=<path> loads another build of the same oracle (the
    sanitizer build: `make -C oracle asan`, see tests/test_host_cpu.py)."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("PDDP_ORACLE_LIB")
        if not path:
            build()
            path = LIB_PATH
        _LIB = ctypes.CDLL(path)
    return Oracle(_LIB, dtype)
