// default_kernels.hip - the sample problems (cartpole, pendulum, double
// cartpole; round 4: rendezvous, see kCarriesCovar) under StateEncoding.DEFAULT = UPPER_TRIANGULAR_CHOLESKY:
// z = mean (D) | triu(upper Cholesky factor U of the covariance, U^T U = C),
// n = D + D (D + 1) / 2 = 14 / 5 / 27.
//
//   dynamics   the example models step the MEAN with the closed-form
//              accelerations and carry the input VARIANCE unchanged,
//              re-encoded as a diagonal covariance:
//              z' = encode(f(mean, u), V = diag(U^T U))
//              (pddp/examples/cartpole/model.py:108,141, pendulum/model.py
//              :84-119, double_cartpole/model.py:100-195), where encode()
//              takes the jittered Cholesky of diag(V)
//              (utils/encoding.py:99-141,536-564): chol'_ii = sqrt(V_i + 1e-12)
//   cost       QRCost on the moment-matched angle augmentation of (mean, C)
//              (pddp/costs/quadratic.py:60-99, examples/*/cost.py,
//              utils/angular.py:47-84,161-248)
//
// Three kernels behind the ordinary problem entry points
// (pddp_nominal_rollout_*, pddp_derivs_*, pddp_line_search_*), selected by
// pddp_problem.encoding:
//   nominal rollout  one lane per trajectory (ilqr.py:457-468)
//   records          one 64-lane workgroup per (trajectory, step): the cost's
//                    value / gradient / Hessian by hyper-dual numbers, one
//                    lane per pair (i <= j) of the n + m inputs, written
//                    straight into the record; the dynamics Jacobian in
//                    closed form: [dmean'/dmean | 0; 0 | dchol'/dchol] with
//                    d sqrt(V_i + j) / dU[k][i] = U[k][i] / sqrt(V_i + j)
//                    (what the reference gets from autograd,
//                    utils/evaluation.py:134-288)
//   line search      one lane per (trajectory, step size): ilqr.py:677-723
//                    _control_law + :764-791 _trajectory_cost
// float and double; the cost is written once over a number type (plain T for
// values, HDual<T> for derivatives).
//
// FULL_COVARIANCE_MATRIX (z = mean | C row-major, n = D + D^2; round 3: the
// double cartpole too, n = 42): C is taken as given - no symmetrisation, so that the partial
// derivatives land on the entries the reference's formulas read - the dynamics
// return diag(diag(C)), and the cost's trace needs no factorisation.
// The same kernels, templated on the encoding, serve VARIANCE_ONLY
// (z = mean | var) and STANDARD_DEVIATION_ONLY (z = mean | std), n = 2 D: the
// covariance is diag(var), the dynamics carry var unchanged (std: sqrt(std^2)),
// the augmented state keeps variances only (angular.py:87-158), so the
// cost's trace term is sum_i Q_ii Var_a[i] (no Cholesky, no jitter).
#include "models.hpp"
#include "problem_args.hpp"

namespace pddp {

PDDP_DEV float exp_(float x) { return expf(x); }
PDDP_DEV double exp_(double x) { return exp(x); }

template <typename T>
struct HDual {  // hyper-dual number: value, d/dx_i, d/dx_j, d2/dx_i dx_j
  T v, a, b, ab;
};
template <typename T> PDDP_DEV HDual<T> operator+(HDual<T> x, HDual<T> y) { return {x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
template <typename T> PDDP_DEV HDual<T> operator-(HDual<T> x, HDual<T> y) { return {x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
template <typename T> PDDP_DEV HDual<T> operator-(HDual<T> x) { return {-x.v, -x.a, -x.b, -x.ab}; }
template <typename T> PDDP_DEV HDual<T> operator*(HDual<T> x, HDual<T> y) {
  return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
          x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
template <typename T> PDDP_DEV HDual<T> operator*(T s, HDual<T> x) { return {s * x.v, s * x.a, s * x.b, s * x.ab}; }
template <typename T> PDDP_DEV HDual<T> operator-(HDual<T> x, T s) { return {x.v - s, x.a, x.b, x.ab}; }
template <typename T> PDDP_DEV HDual<T> exp_(HDual<T> x) {
  const T e = exp_(x.v);
  return {e, e * x.a, e * x.b, e * (x.ab + x.a * x.b)};
}
PDDP_DEV void sincos_lib(float x, float& s, float& c) { sincosf(x, &s, &c); }
PDDP_DEV void sincos_lib(double x, double& s, double& c) { sincos(x, &s, &c); }
template <typename T> PDDP_DEV void sincos_x(T x, T& s, T& c) { sincos_lib(x, s, c); }
template <typename T> PDDP_DEV void sincos_x(HDual<T> x, HDual<T>& s, HDual<T>& c) {
  T sv, cv;
  sincos_lib(x.v, sv, cv);
  s = {sv, cv * x.a, cv * x.b, cv * x.ab - sv * x.a * x.b};
  c = {cv, -sv * x.a, -sv * x.b, -sv * x.ab - cv * x.a * x.b};
}
template <typename T> PDDP_DEV T val(T x) { return x; }
template <typename T> PDDP_DEV T val(HDual<T> x) { return x.v; }
template <typename X, typename T> PDDP_DEV X lift(T v) {
  if constexpr (sizeof(X) == sizeof(T)) return v;
  else return X{v, T(0), T(0), T(0)};
}

// shape of a sample problem under a Gaussian encoding ENC (PDDP_ENC_*)
constexpr int kChol = PDDP_ENC_UPPER_TRIANGULAR_CHOLESKY;
constexpr int kVar = PDDP_ENC_VARIANCE_ONLY;
constexpr int kStd = PDDP_ENC_STANDARD_DEVIATION_ONLY;
constexpr int kFull = PDDP_ENC_FULL_COVARIANCE_MATRIX;
template <int MODEL, int ENC = kChol>
struct DefDims {
  using M = ModelDims<MODEL>;
  static constexpr int D = M::n, m = M::m, na = M::na, nang = M::n_ang;
  static constexpr int nn = na - 2 * nang;      // non-angular rows come first
  static constexpr int NO =
      ENC == kChol ? D * (D + 1) / 2 : (ENC == kFull ? D * D : D);
  static constexpr int n = D + NO;               // encoded size
  static constexpr int non(int r) { return M::col[r]; }
  static constexpr int ang(int a) { return M::col[nn + 2 * a]; }
  // position of U[k][i] (k <= i) among the row-major upper-triangle entries
  static constexpr int tri(int k, int i) {
    return k * D - k * (k - 1) / 2 + (i - k);
  }
};

// covariance of the state from the encoding's second block
template <typename X, int MODEL, int ENC>
PDDP_DEV void covar_of(const X (&oth)[DefDims<MODEL, ENC>::NO],
                       X (&C)[DefDims<MODEL, ENC>::D][DefDims<MODEL, ENC>::D]) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D;
  if constexpr (ENC == kFull) {
    // the matrix as given: its two triangles are separate inputs, and which
    // one a formula reads decides where its partial derivatives land
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c) C[r][c] = oth[r * D + c];
    return;
  }
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = r; c < D; ++c) {
      if constexpr (ENC == kChol) {  // U^T U
        X v = oth[G::tri(0, r)] * oth[G::tri(0, c)];
#pragma unroll
        for (int kx = 1; kx <= r; ++kx)
          v = v + oth[G::tri(kx, r)] * oth[G::tri(kx, c)];
        C[r][c] = v;
        C[c][r] = v;
      } else {
        X v = oth[r] - oth[r];  // zero of the number type
        if (r == c) v = (ENC == kVar) ? oth[r] : oth[r] * oth[r];
        C[r][c] = v;
        C[c][r] = v;
      }
    }
}

// the jittered upper Cholesky only decides which trace the cost sees
// (encoding.py:536-564: jitter 1e-12, 1e-11, ... <= 10, else the diagonal)
template <typename T, int NA>
PDDP_DEV T chol_jitter_of(const T (&C)[NA][NA]) {
  double jit = 1e-12;
  while (jit <= 10.0) {
    T U[NA][NA];
    bool ok = true;
    // (no early exit: a failed pivot ends the reference's attempt, what is
    // computed after it is discarded - the loops unroll, U stays in registers)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int j = i; j < NA; ++j) {
        T s = C[i][j] + (i == j ? (T)jit : T(0));
#pragma unroll
        for (int q = 0; q < i; ++q) s -= U[q][i] * U[q][j];
        if (i == j) {
          if (!(s > T(0))) ok = false;
          U[i][i] = sqrt_(s);
        } else {
          U[i][j] = s / U[i][i];
        }
      }
    }
    if (ok) return (T)jit;
    jit *= 10.0;
  }
  return T(-1);
}

// Rendezvous (8 states, 4 actions, no angles) carries the input's FULL
// covariance through its dynamics (pddp/examples/rendezvous/model.py:94,110:
// encode(mean', C = decode_covar(z))), where the other sample models keep the
// variances only:
//   Cholesky   chol' = the jittered upper Cholesky of U^T U (encoding.py:99-141,
//              536-564), a re-factorisation - the identity up to its 1e-12
//              jitter when U has a positive diagonal, not otherwise
//   full       C' = C, every entry      variance / std   as the other models
template <int MODEL>
constexpr bool kCarriesCovar = (MODEL == PDDP_MODEL_RENDEZVOUS);

template <typename T>
struct FDual {  // forward dual number: value, one tangent
  T v, d;
};
template <typename T> PDDP_DEV FDual<T> operator+(FDual<T> x, FDual<T> y) { return {x.v + y.v, x.d + y.d}; }
template <typename T> PDDP_DEV FDual<T> operator-(FDual<T> x, FDual<T> y) { return {x.v - y.v, x.d - y.d}; }
template <typename T> PDDP_DEV FDual<T> operator*(FDual<T> x, FDual<T> y) { return {x.v * y.v, x.d * y.v + x.v * y.d}; }
template <typename T> PDDP_DEV FDual<T> operator/(FDual<T> x, FDual<T> y) {
  const T q = x.v / y.v;
  return {q, (x.d - q * y.d) / y.v};
}
template <typename T> PDDP_DEV FDual<T> sqrt_x(FDual<T> x) {
  const T r = sqrt_(x.v);
  return {r, x.d / (r + r)};
}
template <typename T> PDDP_DEV T sqrt_x(T x) { return sqrt_(x); }
template <typename T> PDDP_DEV T val(FDual<T> x) { return x.v; }

// R = upper Cholesky factor of U^T U + jitter I, both as the row-major upper
// triangles the encoding stores (utils/encoding.py _cholesky_upper: jitter
// 1e-12, x10 while a pivot fails - per batch there, per row here); X = T or
// FDual<T>.  False when no jitter up to 10 makes it positive-definite (the
// reference raises).
template <typename X, typename T, int D>
PDDP_DEV bool rechol_upper(const X (&U)[D * (D + 1) / 2],
                           X (&R)[D * (D + 1) / 2]) {
  auto tri = [](int k, int i) { return k * D - k * (k - 1) / 2 + (i - k); };
  X C[D * (D + 1) / 2];  // U^T U, upper triangle
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int c = r; c < D; ++c) {
      X v = U[tri(0, r)] * U[tri(0, c)];
#pragma unroll
      for (int k = 1; k <= r; ++k) v = v + U[tri(k, r)] * U[tri(k, c)];
      C[tri(r, c)] = v;
    }
  double jit = 1e-12;
  while (jit <= 10.0) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int j = i; j < D; ++j) {
        X s = C[tri(i, j)];
        if (i == j) s = s + X{(T)jit};
#pragma unroll
        for (int q = 0; q < i; ++q) s = s - R[tri(q, i)] * R[tri(q, j)];
        if (i == j) {
          if (!(val(s) > T(0))) ok = false;
          R[tri(i, i)] = sqrt_x(s);
        } else {
          R[tri(i, j)] = s / R[tri(i, i)];
        }
      }
    }
    if (ok) return true;
    jit *= 10.0;
  }
  return false;
}

// l(z, u) on the augmented Gaussian moments; X = T (value) or HDual<T>.
// mu [D], oth [NO] the encoding's second block, u [m] (clamped).
template <typename X, typename T, int MODEL, int ENC>
PDDP_DEV X qr_cost_default(const ProblemT<T>& P,
                           const X (&mu)[DefDims<MODEL, ENC>::D],
                           const X (&oth)[DefDims<MODEL, ENC>::NO],
                           const X (&u)[DefDims<MODEL, ENC>::m],
                           bool terminal) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D, m = G::m, na = G::na, nn = G::nn, nang = G::nang;
  constexpr int LQ = PDDP_MAX_AUG, LR = PDDP_MAX_ACTION;
  const T* Q = terminal ? P.Qt : P.Q;
  X C[D][D];
  covar_of<X, MODEL, ENC>(oth, C);
  T Cav[na][na];
#pragma unroll
  for (int r = 0; r < na; ++r)
#pragma unroll
    for (int c = 0; c < na; ++c) Cav[r][c] = T(0);
  X Ma[na];
  X tr = lift<X>(T(0)), trd = lift<X>(T(0));  // sum Ca_ij Q_ji; sum Ca_ii Q_ii
  auto put = [&](int r, int c, X v) {  // every entry is written once
    Cav[r][c] = val(v);
    tr = tr + Q[c * LQ + r] * v;
    if (r == c) trd = trd + Q[r * LQ + r] * v;
  };
#pragma unroll
  for (int r = 0; r < nn; ++r) {
    Ma[r] = mu[G::non(r)];
#pragma unroll
    for (int c = 0; c < nn; ++c) put(r, c, C[G::non(r)][G::non(c)]);
  }
#pragma unroll
  for (int a1 = 0; a1 < nang; ++a1) {
    const int i1 = G::ang(a1);
    const X m1 = mu[i1], v1 = C[i1][i1];
    const X damp = exp_(T(-0.5) * v1);
    X s1, c1;
    sincos_x(m1, s1, c1);
    const X Es = damp * s1, Ec = damp * c1;
    const int r = nn + 2 * a1;
    Ma[r] = Es;
    Ma[r + 1] = Ec;
#pragma unroll
    for (int a2 = 0; a2 < nang; ++a2) {
      const int i2 = G::ang(a2);
      const X m2 = mu[i2], v2 = C[i2][i2], cij = C[i1][i2];
      const X lq = T(-0.5) * (v1 + v2), qq = exp_(lq);
      const X ep = exp_(lq + cij) - qq, em = exp_(lq - cij) - qq;
      X sd, cd, ss, cs;
      sincos_x(m1 - m2, sd, cd);
      sincos_x(m1 + m2, ss, cs);
      const int cc = nn + 2 * a2;
      put(r, cc, T(0.5) * (ep * cd - em * cs));          // sin, sin
      put(r + 1, cc + 1, T(0.5) * (ep * cd + em * cs));  // cos, cos
      const X sc = T(0.5) * (ep * sd + em * ss);         // sin_1, cos_2
      put(r, cc + 1, sc);
      put(cc + 1, r, sc);
    }
#pragma unroll
    for (int c = 0; c < nn; ++c) {
      // row = the angle (utils/angular.py:243-245 sums over the row index)
      const X col = C[i1][G::non(c)];
      put(c, r, col * Ec);         // Cov(x, sin)
      put(c, r + 1, -(col * Es));  // Cov(x, cos)
      put(r, c, col * Ec);
      put(r + 1, c, -(col * Es));
    }
  }
  // variance-only encodings: the augmented state keeps diag(Ca) only
  // (full covariance: encode() flattens the augmented matrix as it is)
  const T jit = ENC == kChol ? chol_jitter_of<T, na>(Cav)
                             : (ENC == kFull ? T(0) : T(-1));
  X cost = lift<X>(T(0));
#pragma unroll
  for (int c = 0; c < na; ++c) {
    X row = lift<X>(T(0));
#pragma unroll
    for (int r = 0; r < na; ++r) row = row + Q[r * LQ + c] * (Ma[r] - P.goal[r]);
    cost = cost + row * (Ma[c] - P.goal[c]);
  }
  if (!terminal) {
#pragma unroll
    for (int c = 0; c < m; ++c) {
      X row = lift<X>(T(0));
#pragma unroll
      for (int r = 0; r < m; ++r)
        row = row + P.R[r * LR + c] * (u[r] - P.ugoal[r]);
      cost = cost + row * (u[c] - P.ugoal[c]);
    }
  }
  if (jit >= T(0)) {
    T trq = T(0);
#pragma unroll
    for (int r = 0; r < na; ++r) trq += Q[r * LQ + r];
    cost = cost + tr + lift<X>(jit * trq);  // tr(Q (Ca + jitter I))
  } else {
    cost = cost + trd;  // encode()'s diagonal fall-back
  }
  return cost;
}

// one model step on (mean, other): mean' = f(mean, u); the second block is
// encode(mean', V = decode_var(z)):
//   Cholesky   chol' = diag(sqrt(V_i + 1e-12)), V_i = sum_k U[k][i]^2
//   variance   var' = var            standard deviation   std' = sqrt(std^2)
template <typename T, int MODEL, int ENC>
PDDP_DEV void step_default(const ProblemT<T>& P,
                           T (&mean)[DefDims<MODEL, ENC>::D],
                           T (&oth)[DefDims<MODEL, ENC>::NO],
                           const T (&u)[DefDims<MODEL, ENC>::m]) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D;
  T next[D];
  const Trig<T, MODEL> tr = trig_of<T, MODEL>(mean);
  dynamics<T, MODEL, false>(P, mean, u, tr, next, nullptr, nullptr);
  if constexpr (kCarriesCovar<MODEL> && ENC == kChol) {
    T R[G::NO];
    const bool ok = rechol_upper<T, T, D>(oth, R);
#pragma unroll
    for (int j = 0; j < G::NO; ++j) oth[j] = ok ? R[j] : (T)__builtin_nan("");
  } else if constexpr (kCarriesCovar<MODEL> && ENC == kFull) {
    // C' = C
  } else if constexpr (ENC == kChol) {
    T sd[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      T v = T(0);
#pragma unroll
      for (int k = 0; k <= i; ++k)
        v += oth[G::tri(k, i)] * oth[G::tri(k, i)];  // pow(2).sum(-2)
      sd[i] = sqrt_(v + (T)1e-12);
    }
#pragma unroll
    for (int j = 0; j < G::NO; ++j) oth[j] = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) oth[G::tri(i, i)] = sd[i];
  } else if constexpr (ENC == kStd) {
#pragma unroll
    for (int i = 0; i < D; ++i) oth[i] = sqrt_(oth[i] * oth[i]);
  } else if constexpr (ENC == kFull) {  // diag(decode_var(z)) as a matrix
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (r != c) oth[r * D + c] = T(0);
  }
#pragma unroll
  for (int r = 0; r < D; ++r) mean[r] = next[r];
}

// --------------------------------------------------------------------------
// nominal rollout: one lane per trajectory
// --------------------------------------------------------------------------
template <typename T, int MODEL, int ENC>
__global__ __launch_bounds__(kWave) void rollout_default_kernel(
    ProblemT<T> P, RolloutArgs<T> a) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D, m = G::m, n = G::n, NO = G::NO;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  if (a.mask != nullptr && a.mask[b] == 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  T mean[D], oth[NO];
  T* Zb = a.Z + (size_t)b * (a.N + 1) * n;
  const T* Ub = a.U + (size_t)b * a.N * m;
#pragma unroll
  for (int j = 0; j < D; ++j) mean[j] = a.z0[(size_t)b * n + j];
#pragma unroll
  for (int j = 0; j < NO; ++j) oth[j] = a.z0[(size_t)b * n + D + j];
#pragma unroll
  for (int j = 0; j < D; ++j) Zb[j] = mean[j];
#pragma unroll
  for (int j = 0; j < NO; ++j) Zb[D + j] = oth[j];
  for (int t = 0; t < a.N; ++t) {
    T u[m];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      u[j] = Ub[t * m + j];
      if (bounded) u[j] = clamp1(u[j], a.u_min[j], a.u_max[j]);
    }
    step_default<T, MODEL, ENC>(P, mean, oth, u);
#pragma unroll
    for (int j = 0; j < D; ++j) Zb[(size_t)(t + 1) * n + j] = mean[j];
#pragma unroll
    for (int j = 0; j < NO; ++j) Zb[(size_t)(t + 1) * n + D + j] = oth[j];
  }
}

// --------------------------------------------------------------------------
// derivative records: one workgroup per (trajectory, step)
// --------------------------------------------------------------------------
template <typename T, int MODEL, int ENC>
__global__ __launch_bounds__(kWave) void derivs_default_kernel(
    ProblemT<T> P, DerivArgs<T> a) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D, m = G::m, n = G::n, NO = G::NO;
  constexpr RecLayout lay(n, m);
  constexpr int S = lay.stride;
  __shared__ T sFx[D * D], sFu[D * m], sSd[D];
  const int lane = threadIdx.x;
  const int N = a.N;
  const int bt = blockIdx.x;  // (trajectory, step), step N = terminal
  const int b = bt / (N + 1), t = bt - b * (N + 1);
  if (a.mask != nullptr && a.mask[b] == 0) return;
  const bool terminal = (t == N);
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  const T* z = a.Z + ((size_t)b * (N + 1) + t) * n;
  T* w = a.rec + ((size_t)b * (N + 1) + t) * S;
  T mean[D], oth[NO], un[m], u[m];
#pragma unroll
  for (int j = 0; j < D; ++j) mean[j] = z[j];
#pragma unroll
  for (int j = 0; j < NO; ++j) oth[j] = z[D + j];
#pragma unroll
  for (int r = 0; r < m; ++r) {
    un[r] = terminal ? T(0) : a.U[((size_t)b * N + t) * m + r];
    // derivatives AT the clamped action (ilqr.py:461-462)
    u[r] = (bounded && !terminal) ? clamp1(un[r], a.u_min[r], a.u_max[r])
                                  : un[r];
  }
  // ---- the dynamics' Jacobian blocks, by lane 0 (closed form)
  if (lane == 0) {
    T next[D], Fx[D * D], Fu[D * m];
    if (!terminal) {
      const Trig<T, MODEL> tr = trig_of<T, MODEL>(mean);
      dynamics<T, MODEL, true>(P, mean, u, tr, next, Fx, Fu);
    }
#pragma unroll
    for (int j = 0; j < D * D; ++j) sFx[j] = terminal ? T(0) : Fx[j];
#pragma unroll
    for (int j = 0; j < D * m; ++j) sFu[j] = terminal ? T(0) : Fu[j];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      if constexpr (ENC == kChol) {
        T v = T(0);
#pragma unroll
        for (int k = 0; k <= i; ++k)
          v += oth[G::tri(k, i)] * oth[G::tri(k, i)];
        sSd[i] = sqrt_(v + (T)1e-12);
      } else if constexpr (ENC == kFull) {
        sSd[i] = T(1);
      } else {
        sSd[i] = sqrt_(oth[i] * oth[i]);
      }
    }
  }
  // ---- the cost by hyper-dual numbers: one lane per pair (i <= j)
  const int d = n + (terminal ? 0 : m);
  const int npairs = d * (d + 1) / 2;
  for (int q = lane; q < npairs; q += kWave) {
    int i = 0, rem = q;  // pair q -> (i, j), i <= j, row-major upper triangle
    while (rem >= d - i) { rem -= d - i; ++i; }
    const int j = i + rem;
    using X = HDual<T>;
    auto in = [&](int kx, T v) {
      return X{v, kx == i ? T(1) : T(0), kx == j ? T(1) : T(0), T(0)};
    };
    X mu_[D], oth_[NO], u_[m];
#pragma unroll
    for (int c = 0; c < D; ++c) mu_[c] = in(c, mean[c]);
#pragma unroll
    for (int c = 0; c < NO; ++c) oth_[c] = in(D + c, oth[c]);
#pragma unroll
    for (int r = 0; r < m; ++r) u_[r] = in(n + r, u[r]);
    const X cost = qr_cost_default<X, T, MODEL, ENC>(P, mu_, oth_, u_, terminal);
    if (q == 0) a.L[(size_t)b * (N + 1) + t] = cost.v;
    if (i == j) {
      if (i < n) w[lay.oLz + i] = cost.a;
      else w[lay.oLu + (i - n)] = cost.a;
    }
    if (j < n) {
      w[lay.oLzz + i * n + j] = cost.ab;
      w[lay.oLzz + j * n + i] = cost.ab;
    } else if (i < n) {
      w[lay.oLuz + (j - n) * n + i] = cost.ab;
    } else {
      w[lay.oLuu + (i - n) * m + (j - n)] = cost.ab;
      w[lay.oLuu + (j - n) * m + (i - n)] = cost.ab;
    }
  }
  __syncthreads();
  // ---- F_z = [dmean'/dmean 0; 0 dother'/dother], F_u, the action-side blocks
  // of the terminal record, the un-clamped nominal action, padding
  for (int e = lane; e < n * n; e += kWave) {
    const int r = e / n, c = e - r * n;
    T v = T(0);
    if (!terminal) {
      if (r < D && c < D) {
        v = sFx[r * D + c];
      } else if (r >= D && c >= D) {
        if constexpr (kCarriesCovar<MODEL> && ENC == kChol) {
          continue;  // d chol' / d chol: by dual numbers, below
        } else if constexpr (kCarriesCovar<MODEL> && ENC == kFull) {
          v = (r == c) ? T(1) : T(0);  // C' = C
        } else if constexpr (ENC == kChol) {
          // encoded row r = upper-triangle entry (ri, rj); only the diagonal
          // entries of chol' are non-zero functions of the input:
          // d sqrt(V_i + j) / dU[k][i] = U[k][i] / sqrt(V_i + j)
          int ri = 0, o = r - D;
          while (o >= D - ri) { o -= D - ri; ++ri; }
          const int rj = ri + o;
          int ci = 0, oc = c - D;
          while (oc >= D - ci) { oc -= D - ci; ++ci; }
          const int cj = ci + oc;
          if (ri == rj && cj == ri) v = z[c] / sSd[ri];
        } else if constexpr (ENC == kFull) {
          // C'[i][i] = C[i][i], every other entry of C' is a constant zero
          const int ri = (r - D) / D, rj = (r - D) - ri * D;
          v = (r == c && ri == rj) ? T(1) : T(0);
        } else if constexpr (ENC == kVar) {
          v = (r == c) ? T(1) : T(0);          // var' = var
        } else {
          if (r == c) v = z[c] / sSd[c - D];   // d sqrt(s^2) / ds
        }
      }
    }
    w[lay.oFz + e] = v;
  }
  if constexpr (kCarriesCovar<MODEL> && ENC == kChol) {
    // column D + e of F_z: the re-factorisation in dual numbers seeded at
    // entry e of the input factor (what the reference's autograd returns
    // through torch.linalg.cholesky_ex)
    // (the terminal record's block is zeros, written above)
    for (int e = lane; e < NO && !terminal; e += kWave) {
      using X = FDual<T>;
      X U_[NO], R_[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) U_[c] = X{oth[c], c == e ? T(1) : T(0)};
      const bool ok = rechol_upper<X, T, D>(U_, R_);
#pragma unroll
      for (int r = 0; r < NO; ++r)
        w[lay.oFz + (D + r) * n + (D + e)] = ok ? R_[r].d : (T)__builtin_nan("");
    }
  }
  for (int e = lane; e < n * m; e += kWave) {
    const int r = e / m, c = e - r * m;
    w[lay.oFu + e] = (!terminal && r < D) ? sFu[r * m + c] : T(0);
  }
  if (terminal) {
    for (int e = lane; e < m * n; e += kWave) w[lay.oLuz + e] = T(0);
    for (int e = lane; e < m * m; e += kWave) w[lay.oLuu + e] = T(0);
    if (lane < m) w[lay.oLu + lane] = T(0);
  }
  if (lane < m) w[lay.oU + lane] = un[lane];
  for (int e = lay.oU + m + lane; e < S; e += kWave) w[e] = T(0);
}

// J[b] = sum_t L[b][t] in t order (+ the state reset of pddp_derivs_*)
template <typename T>
__global__ __launch_bounds__(256) void cost_sum_default_kernel(
    int B, int count, const T* L, const uint8_t* mask, T* J, int32_t* state) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (mask != nullptr && mask[b] == 0) return;
  const T* row = L + (size_t)b * count;
  T acc = T(0);
  for (int t = 0; t < count; ++t) acc += row[t];
  J[b] = acc;
  if (state != nullptr) state[b] = PDDP_STATE_UNDEFINED;
}

// --------------------------------------------------------------------------
// line search: one lane per (trajectory, step size)
// --------------------------------------------------------------------------
template <typename T, int MODEL, int ENC>
__global__ __launch_bounds__(kWave) void line_search_default_kernel(
    ProblemT<T> P, LineSearchArgs<T> a) {
  using G = DefDims<MODEL, ENC>;
  constexpr int D = G::D, m = G::m, n = G::n, NO = G::NO;
  constexpr int GS = m + m * n;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.A) return;
  const int b = idx / a.A, ai = idx - b * a.A;
  if (a.active != nullptr && a.active[b] == 0) return;
  if (a.bwd_status != nullptr && a.bwd_status[b] != 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  const int N = a.N;
  const T alpha = a.alphas[ai];
  const T* Zb = a.Z + (size_t)b * (N + 1) * n;
  const T* Ub = a.U + (size_t)b * N * m;
  const T* Gb = a.gains + (size_t)b * N * GS;
  T* Zci = a.Zc + ((size_t)b * (N + 1) * a.A + ai) * n;
  T* Uci = a.Uc + ((size_t)b * N * a.A + ai) * m;
  const size_t zstep = (size_t)a.A * n, ustep = (size_t)a.A * m;

  T mean[D], oth[NO];
#pragma unroll
  for (int j = 0; j < D; ++j) mean[j] = Zb[j];  // Z_new[0] = Z[0] (ilqr.py:690)
#pragma unroll
  for (int j = 0; j < NO; ++j) oth[j] = Zb[D + j];
  T J = T(0);
  for (int t = 0; t < N; ++t) {
    T u[m];
#pragma unroll
    for (int r = 0; r < m; ++r) {
      T du = alpha * Gb[t * GS + r];  // alpha * k[i]              (ilqr.py:708)
      T s = T(0);
      const T* Kr = Gb + t * GS + m + r * n;
      const T* zr = Zb + (size_t)t * n;
#pragma unroll
      for (int c = 0; c < D; ++c) s += (mean[c] - zr[c]) * Kr[c];
#pragma unroll
      for (int c = 0; c < NO; ++c) s += (oth[c] - zr[D + c]) * Kr[D + c];
      du = du + s;  // + dz K^T                                    (ilqr.py:710)
      const T v = Ub[t * m + r] + du;
      u[r] = bounded ? clamp_nan(v, a.u_min[r], a.u_max[r]) : v;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) Zci[(size_t)t * zstep + j] = mean[j];
#pragma unroll
    for (int j = 0; j < NO; ++j) Zci[(size_t)t * zstep + D + j] = oth[j];
#pragma unroll
    for (int j = 0; j < m; ++j) Uci[(size_t)t * ustep + j] = u[j];
    J += qr_cost_default<T, T, MODEL, ENC>(P, mean, oth, u, false);
    step_default<T, MODEL, ENC>(P, mean, oth, u);
  }
#pragma unroll
  for (int j = 0; j < D; ++j) Zci[(size_t)N * zstep + j] = mean[j];
#pragma unroll
  for (int j = 0; j < NO; ++j) Zci[(size_t)N * zstep + D + j] = oth[j];
  T u0[m];
#pragma unroll
  for (int r = 0; r < m; ++r) u0[r] = T(0);
  J += qr_cost_default<T, T, MODEL, ENC>(P, mean, oth, u0, true);
  a.Jc[idx] = J;  // L.sum(0) + l_f                                (ilqr.py:789)
}

// --------------------------------------------------------------------------
// launchers (called from problem_kernels.hip for the Gaussian encodings)
// --------------------------------------------------------------------------
static int check_default(const pddp_problem& p) {
  if (p.encoding != kChol && p.encoding != kVar && p.encoding != kStd &&
      p.encoding != kFull)
    return PDDP_E_UNSUPPORTED;
  // (full covariance: n = D + D^2 = 20 / 6 / 42 - cartpole, pendulum, double
  // cartpole; its sweep is the large generic kernel)

  switch (p.model) {
    case PDDP_MODEL_CARTPOLE:
    case PDDP_MODEL_DOUBLE_CARTPOLE:
    case PDDP_MODEL_PENDULUM:
    case PDDP_MODEL_RENDEZVOUS:  // (carries the full covariance: kCarriesCovar)
      return 0;
  }
  return PDDP_E_UNSUPPORTED;
}

#define PDDP_DEFAULT_ENC(...)                                                \
  switch (p.encoding) {                                                      \
    case kFull: {                                                            \
      constexpr int ENC = kFull; __VA_ARGS__;                                \
    } break;                                                                 \
    case kChol: { constexpr int ENC = kChol; __VA_ARGS__; } break;             \
    case kVar: { constexpr int ENC = kVar; __VA_ARGS__; } break;               \
    default: { constexpr int ENC = kStd; __VA_ARGS__; } break;                 \
  }
#define PDDP_DEFAULT_DISPATCH(...)                                           \
  switch (p.model) {                                                         \
    case PDDP_MODEL_CARTPOLE: {                                              \
      constexpr int MODEL = PDDP_MODEL_CARTPOLE;                             \
      PDDP_DEFAULT_ENC(__VA_ARGS__)                                            \
    } break;                                                                 \
    case PDDP_MODEL_PENDULUM: {                                              \
      constexpr int MODEL = PDDP_MODEL_PENDULUM;                             \
      PDDP_DEFAULT_ENC(__VA_ARGS__)                                            \
    } break;                                                                 \
    case PDDP_MODEL_RENDEZVOUS: {                                            \
      constexpr int MODEL = PDDP_MODEL_RENDEZVOUS;                           \
      PDDP_DEFAULT_ENC(__VA_ARGS__)                                            \
    } break;                                                                 \
    default: {                                                               \
      constexpr int MODEL = PDDP_MODEL_DOUBLE_CARTPOLE;                      \
      PDDP_DEFAULT_ENC(__VA_ARGS__)                                            \
    } break;                                                                 \
  }

template <typename T>
int default_rollout(const pddp_problem& p, RolloutArgs<T> a, hipStream_t st) {
  if (int rc = check_default(p)) return rc;
  const ProblemT<T> P = convert_problem<T>(p);
  const dim3 grid((a.B + kWave - 1) / kWave), block(kWave);
  PDDP_DEFAULT_DISPATCH(PDDP_LAUNCH((rollout_default_kernel<T, MODEL, ENC>),
                                    grid, block, 0, st, P, a))
  return launch_status();
}
template <typename T>
int default_derivs(const pddp_problem& p, DerivArgs<T> a, hipStream_t st) {
  if (int rc = check_default(p)) return rc;
  const ProblemT<T> P = convert_problem<T>(p);
  const dim3 grid(a.B * (a.N + 1)), block(kWave);
  PDDP_DEFAULT_DISPATCH(PDDP_LAUNCH((derivs_default_kernel<T, MODEL, ENC>),
                                    grid, block, 0, st, P, a))
  if (int rc = launch_status()) return rc;
  PDDP_LAUNCH((cost_sum_default_kernel<T>), dim3((a.B + 255) / 256), dim3(256),
              0, st, a.B, a.N + 1, (const T*)a.L, a.mask, a.J, a.state);
  return launch_status();
}
template <typename T>
int default_line_search(const pddp_problem& p, LineSearchArgs<T> a,
                        hipStream_t st) {
  if (int rc = check_default(p)) return rc;
  const ProblemT<T> P = convert_problem<T>(p);
  const int total = a.B * a.A;
  const dim3 grid((total + kWave - 1) / kWave), block(kWave);
  PDDP_DEFAULT_DISPATCH(PDDP_LAUNCH((line_search_default_kernel<T, MODEL, ENC>),
                                    grid, block, 0, st, P, a))
  return launch_status();
}

template int default_rollout<float>(const pddp_problem&, RolloutArgs<float>, hipStream_t);
template int default_rollout<double>(const pddp_problem&, RolloutArgs<double>, hipStream_t);
template int default_derivs<float>(const pddp_problem&, DerivArgs<float>, hipStream_t);
template int default_derivs<double>(const pddp_problem&, DerivArgs<double>, hipStream_t);
template int default_line_search<float>(const pddp_problem&, LineSearchArgs<float>, hipStream_t);
template int default_line_search<double>(const pddp_problem&, LineSearchArgs<double>, hipStream_t);

}  // namespace pddp
