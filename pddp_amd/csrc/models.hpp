// models.hpp - device maths of the reference's sample problems: analytic
// dynamics with Jacobians and the quadratic cost on the angle-augmented state
// with gradient / Hessian.  StateEncoding.IGNORE_UNCERTAINTY (z == mean).
//
// What the reference gets from autograd (utils/evaluation.py:134-288) is
// written out in closed form here.  Restated from:
//   cartpole         pddp/examples/cartpole/model.py:88-141, cost.py:32-87
//   pendulum         pddp/examples/pendulum/model.py:84-119, cost.py:32-88
//   double cartpole  pddp/examples/double_cartpole/model.py:100-195, cost.py:32-96
//   rendezvous       pddp/examples/rendezvous/model.py:79-115, cost.py:29-43
//   QRCost           pddp/costs/quadratic.py:60-99
//   augment_state    pddp/utils/angular.py:251-286
#pragma once

#include <type_traits>
#include "pddp_common.hpp"

namespace pddp {

// Problem constants converted to the arithmetic type (kernel argument).
template <typename T>
struct ProblemT {
  int model, encoding, n, m, na;
  T dt;
  T p[PDDP_MAX_PARAMS - 1];
  T Q[PDDP_MAX_AUG * PDDP_MAX_AUG];
  T Qt[PDDP_MAX_AUG * PDDP_MAX_AUG];
  T R[PDDP_MAX_ACTION * PDDP_MAX_ACTION];
  T goal[PDDP_MAX_AUG];
  T ugoal[PDDP_MAX_ACTION];
};

template <typename T>
inline ProblemT<T> convert_problem(const pddp_problem& s) {
  ProblemT<T> d;
  d.model = s.model;
  d.encoding = s.encoding;
  d.n = s.encoded_size;
  d.m = s.action_size;
  d.na = s.aug_size;
  d.dt = (T)s.params[0];
  for (int i = 0; i < PDDP_MAX_PARAMS - 1; ++i) d.p[i] = (T)s.params[i + 1];
  for (int i = 0; i < PDDP_MAX_AUG * PDDP_MAX_AUG; ++i) {
    d.Q[i] = (T)s.Q[i];
    d.Qt[i] = (T)s.Q_term[i];
  }
  for (int i = 0; i < PDDP_MAX_ACTION * PDDP_MAX_ACTION; ++i) d.R[i] = (T)s.R[i];
  for (int i = 0; i < PDDP_MAX_AUG; ++i) d.goal[i] = (T)s.x_goal[i];
  for (int i = 0; i < PDDP_MAX_ACTION; ++i) d.ugoal[i] = (T)s.u_goal[i];
  return d;
}

// Compile-time shape of each model: state size, action size, augmented size
// and, for each augmented row i, the state column it depends on (col) and
// whether it is a plain copy (0), a sine (1) or a cosine (2).
template <int MODEL>
struct ModelDims;
template <>
struct ModelDims<PDDP_MODEL_CARTPOLE> {
  static constexpr int n = 4, m = 1, na = 5, n_ang = 1;
  static constexpr int col[5] = {0, 1, 3, 2, 2};
  static constexpr int kind[5] = {0, 0, 0, 1, 2};
};
template <>
struct ModelDims<PDDP_MODEL_PENDULUM> {
  static constexpr int n = 2, m = 1, na = 3, n_ang = 1;
  static constexpr int col[3] = {1, 0, 0};
  static constexpr int kind[3] = {0, 1, 2};
};
template <>
struct ModelDims<PDDP_MODEL_DOUBLE_CARTPOLE> {
  static constexpr int n = 6, m = 1, na = 8, n_ang = 2;
  static constexpr int col[8] = {0, 1, 3, 5, 2, 2, 4, 4};
  static constexpr int kind[8] = {0, 0, 0, 0, 1, 2, 1, 2};
};
template <>
struct ModelDims<PDDP_MODEL_RENDEZVOUS> {
  static constexpr int n = 8, m = 4, na = 8, n_ang = 0;
  static constexpr int col[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  static constexpr int kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// sin and cos of one float argument.  The rollouts sit on a chain of N
// dependent steps with one wavefront per SIMD, where every instruction and
// every branch of the step is paid in full; the library's sincosf carries a
// per-lane branch to its Payne-Hanek path, which diverging line-search
// candidates (angles of 1e5 rad and beyond) keep taking.  Here: reduction by
// pi/2 in double precision with a two-constant pi/2 (exact enough for
// |x| < 2^30: the error is ~1e-23 rad), the Cephes minimax polynomials on
// [-pi/4, pi/4] (about 1 ulp), branch-free.  Non-finite arguments give NaN
// like the library; finite ones beyond 2^30 take the library's result,
// computed under one wave-uniform test and selected per lane (a trajectory's
// result must not depend on its neighbours in the wavefront).
// polynomials on the reduced argument r in [-pi/4, pi/4], quadrant q
PDDP_DEV void sincos_poly(float r, int q, float& s, float& c) {
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
  const bool swap = (q & 1) != 0;
  const unsigned ss = ((unsigned)q & 2u) << 30;
  const unsigned cs = (((unsigned)q + 1u) & 2u) << 30;
  s = __uint_as_float(__float_as_uint(swap ? cr : sr) ^ ss);
  c = __uint_as_float(__float_as_uint(swap ? sr : cr) ^ cs);
}
// the branch-free part: ~1 ulp for |x| < 2^30, NaN for non-finite arguments;
// beyond 2^30 (where neighbouring floats are 64 rad and more apart) a finite
// value in [-1, 1] without meaning.  (A float Cody-Waite reduction for
// |x| < 4096 with this double-precision one kept for larger arguments under a
// wave-uniform test was measured: the line search got 3 us SLOWER - its
// diverging candidates take the second path in most wavefronts.)
PDDP_DEV void sincos_core(float x, float& s, float& c) {
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * 0.63661977236758138243);  // 2 / pi
  double rd = __builtin_fma(kd, -1.57079632679489655800e+00, xd);
  rd = __builtin_fma(kd, -6.12323399573676603587e-17, rd);
  // x = +-inf: kd = +-inf and rd = inf - inf = NaN; x = NaN: rd = NaN.  The
  // polynomials, the swap and the sign flips keep a NaN a NaN - no test needed
  sincos_poly((float)rd, (int)kd, s, c);
}
constexpr float kTrigCoreLimit = 1073741824.0f;  // 2^30
PDDP_DEV void sincos_(float x, float& s, float& c) {
  sincos_core(x, s, c);
  // one compare on the rollouts' chain: an infinite argument takes the
  // library's path as well (NaN from both); a state stays infinite for one
  // step only - its sine is NaN and so is everything after it
  const bool big = fabsf(x) >= kTrigCoreLimit;
  if (__builtin_expect(__any(big), 0)) {
    float sl, cl;
    sincosf(x, &sl, &cl);
    s = big ? sl : s;
    c = big ? cl : c;
  }
}
PDDP_DEV void sincos_(double x, double& s, double& c) { sincos(x, &s, &c); }

// 1 / a: float through v_rcp_f32 and one Newton step (<= 1 ulp, 3
// instructions instead of the 11 of an IEEE division); double divides.
PDDP_DEV float inv_(float a) {
  const float r = __builtin_amdgcn_rcpf(a);
  return __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r);
}
PDDP_DEV double inv_(double a) { return 1.0 / a; }

// sin / cos of a state's angles, evaluated once per state and shared by the
// dynamics and the cost (both need them; each costs a range reduction).
template <typename T, int MODEL>
struct Trig {
  T s[ModelDims<MODEL>::n_ang > 0 ? ModelDims<MODEL>::n_ang : 1];
  T c[ModelDims<MODEL>::n_ang > 0 ? ModelDims<MODEL>::n_ang : 1];
};
template <typename T, int MODEL>
PDDP_DEV Trig<T, MODEL> trig_of(const T* z) {
  Trig<T, MODEL> tr;
  tr.s[0] = T(0);
  tr.c[0] = T(1);
  if constexpr (MODEL == PDDP_MODEL_CARTPOLE) sincos_(z[2], tr.s[0], tr.c[0]);
  if constexpr (MODEL == PDDP_MODEL_PENDULUM) sincos_(z[0], tr.s[0], tr.c[0]);
  if constexpr (MODEL == PDDP_MODEL_DOUBLE_CARTPOLE) {
    sincos_(z[2], tr.s[0], tr.c[0]);
    sincos_(z[4], tr.s[1], tr.c[1]);
  }
  return tr;
}

// 3x3 solve with partial pivoting, NR right-hand sides (B[3][NR]).
template <typename T, int NR>
PDDP_DEV void solve3(const T (&A)[3][3], T (&B)[3][NR]) {
  T a[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = A[i][j];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // pick the pivot row by swapping values (no dynamic register indexing)
#pragma unroll
    for (int r = c + 1; r < 3; ++r) {
      const bool sw = abs_(a[r][c]) > abs_(a[c][c]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const T x = a[c][j], y = a[r][j];
        a[c][j] = sw ? y : x;
        a[r][j] = sw ? x : y;
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const T x = B[c][j], y = B[r][j];
        B[c][j] = sw ? y : x;
        B[r][j] = sw ? x : y;
      }
    }
#pragma unroll
    for (int r = c + 1; r < 3; ++r) {
      const T f = a[r][c] / a[c][c];
#pragma unroll
      for (int j = c; j < 3; ++j) a[r][j] -= f * a[c][j];
#pragma unroll
      for (int j = 0; j < NR; ++j) B[r][j] -= f * B[c][j];
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int r = 2; r >= 0; --r) {
      T s = B[r][j];
#pragma unroll
      for (int c = r + 1; c < 3; ++c) s -= a[r][c] * B[c][j];
      B[r][j] = s / a[r][r];
    }
}

// z_next = model(z, u); if JAC also F_z [n][n] and F_u [n][m] (row-major).
template <typename T, int MODEL, bool JAC>
PDDP_DEV void dynamics(const ProblemT<T>& P, const T* z, const T* u,
                       const Trig<T, MODEL>& tr, T* zn, T* Fz, T* Fu) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  const T dt = P.dt;
  if constexpr (JAC) {
#pragma unroll
    for (int i = 0; i < n * n; ++i) Fz[i] = T(0);
#pragma unroll
    for (int i = 0; i < n * m; ++i) Fu[i] = T(0);
  }

  if constexpr (MODEL == PDDP_MODEL_CARTPOLE) {
    const T mc = P.p[0], mp = P.p[1], l = P.p[2], mu = P.p[3], g = P.p[4];
    const T x = z[0], xd = z[1], th = z[2], thd = z[3], F = u[0];
    const T s = tr.s[0], c = tr.c[0];
    T a0 = mp * l * thd * thd * s;
    T a1 = g * s;
    T a2 = F - mu * xd;
    T a3 = T(4) * (mc + mp) - T(3) * mp * c * c;
    T num_t = a0 * c + T(2) * ((mc + mp) * a1 + a2 * c);
    // one reciprocal of a3 serves both accelerations (and the Jacobians)
    T ia3 = inv_(a3);
    const T il = T(1) / l;  // loop-invariant: hoisted out of the rollouts
    const T thdd = (T(-3) * num_t) * (ia3 * il);
    T num_x = T(2) * a0 + T(3) * mp * a1 * c + T(4) * a2;
    const T xdd = num_x * ia3;
    const T nxd = xd + xdd * dt;
    const T nthd = thd + thdd * dt;
    zn[0] = x + nxd * dt;
    zn[1] = nxd;
    zn[2] = th + nthd * dt;
    zn[3] = nthd;
    if constexpr (JAC) {
      T da0_th = mp * l * thd * thd * c;
      T da0_thd = T(2) * mp * l * thd * s;
      const T da1_th = g * c;
      T da3_th = T(6) * mp * c * s;
      T dnt_th = da0_th * c - a0 * s + T(2) * ((mc + mp) * da1_th - a2 * s);
      T dnt_thd = da0_thd * c;
      const T dnt_xd = T(-2) * mu * c;
      const T dnt_F = T(2) * c;
      T dnx_th = T(2) * da0_th + T(3) * mp * (da1_th * c - a1 * s);
      const T dnx_thd = T(2) * da0_thd;
      const T dnx_xd = T(-4) * mu;
      const T dnx_F = T(4);
      const T kt = T(-3) * il;
      const T dthdd_xd = kt * dnt_xd * ia3;
      const T dthdd_th = kt * (dnt_th * a3 - num_t * da3_th) * ia3 * ia3;
      const T dthdd_thd = kt * dnt_thd * ia3;
      const T dthdd_F = kt * dnt_F * ia3;
      T dthdd_thp = dthdd_th, dthdd_xdp = dthdd_xd, dthdd_thdp = dthdd_thd;
      const T dxdd_xd = dnx_xd * ia3;
      const T dxdd_th = (dnx_th * a3 - num_x * da3_th) * ia3 * ia3;
      const T dxdd_thd = dnx_thd * ia3;
      const T dxdd_F = dnx_F * ia3;
      Fz[1 * n + 1] = T(1) + dxdd_xd * dt;
      Fz[1 * n + 2] = dxdd_th * dt;
      Fz[1 * n + 3] = dxdd_thd * dt;
      Fz[3 * n + 1] = dthdd_xdp * dt;
      Fz[3 * n + 2] = dthdd_thp * dt;
      Fz[3 * n + 3] = T(1) + dthdd_thdp * dt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Fz[0 * n + j] = Fz[1 * n + j] * dt;
        Fz[2 * n + j] = Fz[3 * n + j] * dt;
      }
      Fz[0 * n + 0] += T(1);
      Fz[2 * n + 2] += T(1);
      Fu[1] = dxdd_F * dt;
      Fu[3] = dthdd_F * dt;
      Fu[0] = Fu[1] * dt;
      Fu[2] = Fu[3] * dt;
    }
  } else if constexpr (MODEL == PDDP_MODEL_PENDULUM) {
    const T mm = P.p[0], l = P.p[1], mu = P.p[2], g = P.p[3];
    const T th = z[0], thd = z[1], tq = u[0];
    const T temp = mm * l;
    const T s = tr.s[0], c = tr.c[0];
    T acc = tq - mu * thd - T(0.5) * temp * g * s;
    const T sc = T(3) / (temp * l);
    acc = T(3) * acc / (temp * l);
    zn[0] = th + thd * dt;
    zn[1] = thd + acc * dt;
    if constexpr (JAC) {
      Fz[0] = T(1);
      Fz[1] = dt;
      Fz[2] = sc * (T(-0.5) * temp * g * c) * dt;
      Fz[3] = T(1) + sc * (-mu) * dt;
      Fu[0] = T(0);
      Fu[1] = sc * dt;
    }
  } else if constexpr (MODEL == PDDP_MODEL_DOUBLE_CARTPOLE) {
    const T mc = P.p[0], mp1 = P.p[1], mp2 = P.p[2], l1 = P.p[3], l2 = P.p[4],
            mu = P.p[5], g = P.p[6];
    const T x = z[0], xd = z[1], t1 = z[2], t1d = z[3], t2 = z[4], t2d = z[5],
            F = u[0];
    const T s1 = tr.s[0], c1 = tr.c[0], s2 = tr.s[1], c2 = tr.c[1];
    T sd, cd;
    sincos_(t1 - t2, sd, cd);
    const T a0 = mp2 + T(2) * mc;
    const T a1 = mc * l2;
    const T a2 = l1 * t1d * t1d;
    const T a3 = a1 * t2d * t2d;
    const T A[3][3] = {
        {T(2) * (mp1 + mp2 + mc), -a0 * l1 * c1, -a1 * c2},
        {T(-3) * a0 * c1, (T(2) * a0 + T(2) * mc) * l1, T(3) * a1 * cd},
        {T(-3) * c2, T(3) * l1 * cd, T(2) * l2}};
    T sol[3][1] = {{T(2) * F - T(2) * mu * xd - a0 * a2 * s1 - a3 * s2},
                   {T(3) * a0 * g * s1 - T(3) * a3 * sd},
                   {T(3) * a2 * sd + T(3) * g * s2}};
    solve3<T, 1>(A, sol);
    const T nxd = xd + sol[0][0] * dt;
    const T nt1d = t1d + sol[1][0] * dt;
    const T nt2d = t2d + sol[2][0] * dt;
    zn[0] = x + nxd * dt;
    zn[1] = nxd;
    zn[2] = t1 + nt1d * dt;
    zn[3] = nt1d;
    zn[4] = t2 + nt2d * dt;
    zn[5] = nt2d;
    if constexpr (JAC) {
      // d sol / dq = A^-1 (db/dq - dA/dq sol), q = xd, t1, t1d, t2, t2d, F
      const T da2 = T(2) * l1 * t1d, da3 = T(2) * a1 * t2d;
      const T q0 = sol[0][0], q1 = sol[1][0], q2 = sol[2][0];
      T R[3][6];
      R[0][0] = T(-2) * mu; R[1][0] = T(0); R[2][0] = T(0);
      R[0][1] = -a0 * a2 * c1 - (a0 * l1 * s1 * q1);
      R[1][1] = T(3) * a0 * g * c1 - T(3) * a3 * cd -
                (T(3) * a0 * s1 * q0 - T(3) * a1 * sd * q2);
      R[2][1] = T(3) * a2 * cd - (T(-3) * l1 * sd * q1);
      R[0][2] = -a0 * da2 * s1; R[1][2] = T(0); R[2][2] = T(3) * da2 * sd;
      R[0][3] = -a3 * c2 - (a1 * s2 * q2);
      R[1][3] = T(3) * a3 * cd - (T(3) * a1 * sd * q2);
      R[2][3] = T(-3) * a2 * cd + T(3) * g * c2 -
                (T(3) * s2 * q0 + T(3) * l1 * sd * q1);
      R[0][4] = -da3 * s2; R[1][4] = T(-3) * da3 * sd; R[2][4] = T(0);
      R[0][5] = T(2); R[1][5] = T(0); R[2][5] = T(0);
      solve3<T, 6>(A, R);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int vrow = 2 * r + 1, prow = 2 * r;
#pragma unroll
        for (int q = 0; q < 5; ++q) Fz[vrow * n + (q + 1)] = R[r][q] * dt;
        Fz[vrow * n + vrow] += T(1);
#pragma unroll
        for (int j = 0; j < 6; ++j) Fz[prow * n + j] = Fz[vrow * n + j] * dt;
        Fz[prow * n + prow] += T(1);
        Fu[vrow] = R[r][5] * dt;
        Fu[prow] = Fu[vrow] * dt;
      }
    }
  } else {  // PDDP_MODEL_RENDEZVOUS
    const T mass = P.p[0], alpha = P.p[1];
    const T fr = T(1) - alpha * dt / mass;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      zn[i] = z[i] + z[4 + i] * dt;
      T acc = z[4 + i] * fr;
      acc += u[i] * dt / mass;
      zn[4 + i] = z[4 + i] + acc * dt;
      if constexpr (JAC) {
        Fz[i * n + i] = T(1);
        Fz[i * n + 4 + i] = dt;
        Fz[(4 + i) * n + 4 + i] = T(1) + fr * dt;
        Fu[(4 + i) * m + i] = dt / mass * dt;
      }
    }
  }
}

// Augmented state a = augment_state(z), and d a_i / d z_{col_i}.
template <typename T, int MODEL>
PDDP_DEV void augment(const T* z, const Trig<T, MODEL>& tr, T* a, T* d) {
  using D = ModelDims<MODEL>;
  int q = 0;  // angle counter: rows come as (sin a_q, cos a_q) pairs
#pragma unroll
  for (int i = 0; i < D::na; ++i) {
    if (D::kind[i] == 0) {
      a[i] = z[D::col[i]];
      d[i] = T(1);
    } else if (D::kind[i] == 1) {
      a[i] = tr.s[q];
      d[i] = tr.c[q];
    } else {
      a[i] = tr.c[q];
      d[i] = -tr.s[q];
      ++q;
    }
  }
}

// Which augmented-state rows / columns of a cost matrix hold a non-zero entry
// (bit i: row i or column i does).  CartpoleCost's stage Q (reference
// examples/cartpole/cost.py:46-51) lives on {x, sin, cos} = 0b11001: of the 25
// products of dx^T Q dx, 16 are against structural zeros.
inline unsigned live_mask(const double* Q, int na) {
  unsigned mask = 0;
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < na; ++j)
      if (Q[i * PDDP_MAX_AUG + j] != 0.0) mask |= (1u << i) | (1u << j);
  return mask;
}
template <int MODEL>
constexpr unsigned kFullMask = (1u << ModelDims<MODEL>::na) - 1u;

// Cost value only (line search, ilqr.py:764-791). u == nullptr <=> terminal.
// QM: the live rows / columns of the matrix (live_mask); the terms dropped are
// products with an exact zero, which leave every finite partial sum unchanged
// (x + 0 * y == x), so the value is bitwise that of the full double loop.  A
// non-finite entry of a dropped row would have turned the sum into NaN through
// 0 * inf: the rollouts keep that - such a state is non-finite in a live row
// one step later at the latest (positions integrate the velocities, sin / cos
// of a non-finite angle are NaN) and the terminal cost is always evaluated in
// full, so J is NaN in both forms.
template <typename T, int MODEL, unsigned QM = kFullMask<MODEL>>
PDDP_DEV T cost_value(const ProblemT<T>& P, const T* z, const T* u,
                      const Trig<T, MODEL>& tr, bool terminal) {
  using D = ModelDims<MODEL>;
  constexpr int na = D::na, m = D::m;
  const T* Q = terminal ? P.Qt : P.Q;
  T a[na], d[na], dx[na];
  augment<T, MODEL>(z, tr, a, d);
#pragma unroll
  for (int i = 0; i < na; ++i) dx[i] = a[i] - P.goal[i];
  T cost = T(0);
#pragma unroll
  for (int j = 0; j < na; ++j) {
    if (!((QM >> j) & 1u)) continue;
    T dq = T(0);
#pragma unroll
    for (int i = 0; i < na; ++i)
      if ((QM >> i) & 1u) dq += dx[i] * Q[i * PDDP_MAX_AUG + j];
    cost += dq * dx[j];
  }
  if (!terminal) {
    T du[m];
#pragma unroll
    for (int i = 0; i < m; ++i) du[i] = u[i] - P.ugoal[i];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      T dr = T(0);
#pragma unroll
      for (int i = 0; i < m; ++i) dr += du[i] * P.R[i * PDDP_MAX_ACTION + j];
      cost += dr * du[j];
    }
  }
  return cost;
}

// Cost with gradient and Hessian w.r.t. (z, u) (ilqr.py:464-465,471-473).
// l_uz is identically zero for QRCost and is not returned.  QM as in
// cost_value: rows / columns outside it only ever contribute exact zeros.
template <typename T, int MODEL, unsigned QM = kFullMask<MODEL>>
PDDP_DEV T cost_derivs(const ProblemT<T>& P, const T* z, const T* u,
                       const Trig<T, MODEL>& tr, bool terminal, T* l_z,
                       T* l_zz, T* l_u, T* l_uu) {
  using D = ModelDims<MODEL>;
  constexpr int na = D::na, n = D::n, m = D::m;
  const T* Q = terminal ? P.Qt : P.Q;
  T a[na], d[na], dx[na], g[na];
  augment<T, MODEL>(z, tr, a, d);
#pragma unroll
  for (int i = 0; i < na; ++i) dx[i] = a[i] - P.goal[i];
  T cost = T(0);
#pragma unroll
  for (int j = 0; j < na; ++j) {
    if (!((QM >> j) & 1u)) continue;
    T dq = T(0);
#pragma unroll
    for (int i = 0; i < na; ++i)
      if ((QM >> i) & 1u) dq += dx[i] * Q[i * PDDP_MAX_AUG + j];
    cost += dq * dx[j];
  }
#pragma unroll
  for (int i = 0; i < na; ++i) {
    T s = T(0);
    if ((QM >> i) & 1u) {
#pragma unroll
      for (int j = 0; j < na; ++j)
        if ((QM >> j) & 1u)
          s += (Q[i * PDDP_MAX_AUG + j] + Q[j * PDDP_MAX_AUG + i]) * dx[j];
    }
    g[i] = s;
  }
#pragma unroll
  for (int c = 0; c < n; ++c) l_z[c] = T(0);
#pragma unroll
  for (int i = 0; i < na; ++i)
    if ((QM >> i) & 1u) l_z[D::col[i]] += d[i] * g[i];
#pragma unroll
  for (int i = 0; i < n * n; ++i) l_zz[i] = T(0);
#pragma unroll
  for (int i = 0; i < na; ++i) {
    if (!((QM >> i) & 1u)) continue;
#pragma unroll
    for (int k = 0; k < na; ++k)
      if ((QM >> k) & 1u)
        l_zz[D::col[i] * n + D::col[k]] +=
            d[i] *
            ((Q[i * PDDP_MAX_AUG + k] + Q[k * PDDP_MAX_AUG + i]) * d[k]);
  }
  // second derivative of the augmentation: d2 sin = -sin, d2 cos = -cos
#pragma unroll
  for (int i = 0; i < na; ++i)
    if (D::kind[i] != 0 && ((QM >> i) & 1u))
      l_zz[D::col[i] * n + D::col[i]] += g[i] * (-a[i]);
  if (!terminal) {
    T du[m];
#pragma unroll
    for (int i = 0; i < m; ++i) du[i] = u[i] - P.ugoal[i];
#pragma unroll
    for (int j = 0; j < m; ++j) {
      T dr = T(0);
#pragma unroll
      for (int i = 0; i < m; ++i) dr += du[i] * P.R[i * PDDP_MAX_ACTION + j];
      cost += dr * du[j];
    }
#pragma unroll
    for (int i = 0; i < m; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < m; ++j) {
        const T rij = P.R[i * PDDP_MAX_ACTION + j] + P.R[j * PDDP_MAX_ACTION + i];
        s += rij * du[j];
        l_uu[i * m + j] = rij;
      }
      l_u[i] = s;
    }
  }
  return cost;
}

// One derivative record (layout: pddp_hip.h) of state z under the un-clamped
// nominal action un; returns the stage / terminal cost.  The dynamics see the
// clamped action; the record keeps the un-clamped nominal u for the BoxQP
// bounds (ilqr.py:457-473, 602-603).  QM: live rows of the STAGE cost matrix
// (a terminal row is evaluated in full whatever QM says when `terminal` is a
// run-time flag shared by both; pass the full mask then).
template <typename T, int MODEL, unsigned QM = kFullMask<MODEL>>
PDDP_DEV T record_of(const ProblemT<T>& P, const T* z, const T* un,
                     bool terminal, bool bounded, const T* u_min,
                     const T* u_max, T* w) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr RecLayout lay(n, m);
  constexpr int S = lay.stride;
  T u[m], zn[n];
  T Fz[n * n], Fu[n * m], lz[n], lzz[n * n], lu[m], luu[m * m];
#pragma unroll
  for (int j = 0; j < m; ++j) {
    u[j] = bounded ? clamp1(un[j], u_min[j], u_max[j]) : un[j];
    lu[j] = T(0);
  }
#pragma unroll
  for (int j = 0; j < m * m; ++j) luu[j] = T(0);
  Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
  if (!terminal) {
    dynamics<T, MODEL, true>(P, z, u, tr, zn, Fz, Fu);
  } else {
#pragma unroll
    for (int j = 0; j < n * n; ++j) Fz[j] = T(0);
#pragma unroll
    for (int j = 0; j < n * m; ++j) Fu[j] = T(0);
  }
  const T l =
      cost_derivs<T, MODEL, QM>(P, z, u, tr, terminal, lz, lzz, lu, luu);
#pragma unroll
  for (int j = 0; j < n * n; ++j) w[lay.oFz + j] = Fz[j];
#pragma unroll
  for (int j = 0; j < n * n; ++j) w[lay.oLzz + j] = lzz[j];
#pragma unroll
  for (int j = 0; j < n * m; ++j) w[lay.oFu + j] = Fu[j];
#pragma unroll
  for (int j = 0; j < m * n; ++j) w[lay.oLuz + j] = T(0);
#pragma unroll
  for (int j = 0; j < n; ++j) w[lay.oLz + j] = lz[j];
#pragma unroll
  for (int j = 0; j < m * m; ++j) w[lay.oLuu + j] = luu[j];
#pragma unroll
  for (int j = 0; j < m; ++j) w[lay.oLu + j] = lu[j];
#pragma unroll
  for (int j = 0; j < m; ++j) w[lay.oU + j] = un[j];
#pragma unroll
  for (int j = lay.oU + m; j < S; ++j) w[j] = T(0);
  return l;
}

}  // namespace pddp
