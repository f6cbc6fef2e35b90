// gains.hpp - small dense algebra on the action space (m <= 4), in registers.
//
// Everything here runs redundantly in every lane of the wavefront that owns a
// trajectory (uniform control flow), on register arrays with compile-time
// extents, so nothing spills to scratch.  The free/clamped structure of BoxQP
// is handled by MASKING instead of compaction: a clamped dimension is replaced
// by an identity row/column, which leaves the arithmetic on the free block
// bit-identical to factorising the compacted matrix.
//
// Reference behaviour restated (paths relative to the reference repo):
//   boxqp            pddp/utils/constraint.py:150-266
//   eig-clamp        pddp/controllers/ilqr.py:631-634
//   potrf / potrs    pddp/controllers/ilqr.py:595-597,616,661
#pragma once

#include "pddp_common.hpp"

namespace pddp {

// Upper Cholesky A = U^T U of the free block (mask bit set = free).
// Returns true on failure (pivot not > 0 or not finite).
template <typename T, int M>
PDDP_DEV bool chol_upper_masked(const T (&A)[M * M], unsigned free_bits,
                                T (&U)[M * M]) {
  bool fail = false;
#pragma unroll
  for (int i = 0; i < M * M; ++i) U[i] = T(0);
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const bool fj = (free_bits >> j) & 1u;
    T d = fj ? A[j * M + j] : T(1);
#pragma unroll
    for (int k = 0; k < j; ++k) d -= U[k * M + j] * U[k * M + j];
    fail = fail || !(d > T(0)) || !is_finite(d);
    const T ujj = sqrt_(d);
    U[j * M + j] = ujj;
#pragma unroll
    for (int c = j + 1; c < M; ++c) {
      const bool fc = (free_bits >> c) & 1u;
      T s = (fj && fc) ? A[j * M + c] : T(0);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= U[k * M + j] * U[k * M + c];
      U[j * M + c] = s / ujj;
    }
  }
  return fail;
}

// Solves (U^T U) x = b in place; masked entries of b must be 0 on entry and
// come out 0.
template <typename T, int M>
PDDP_DEV void chol_solve(const T (&U)[M * M], T (&b)[M]) {
#pragma unroll
  for (int i = 0; i < M; ++i) {
    T s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= U[k * M + i] * b[k];
    b[i] = s / U[i * M + i];
  }
#pragma unroll
  for (int i = M - 1; i >= 0; --i) {
    T s = b[i];
#pragma unroll
    for (int k = i + 1; k < M; ++k) s -= U[i * M + k] * b[k];
    b[i] = s / U[i * M + i];
  }
}

// Symmetric eigendecomposition A = E diag(e) E^T by cyclic Jacobi.
template <typename T, int M>
PDDP_DEV void jacobi_eig(const T (&A)[M * M], T (&e)[M], T (&E)[M * M]) {
  if constexpr (M == 1) {
    e[0] = A[0];
    E[0] = T(1);
  } else {
    T a[M * M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        a[i * M + j] = A[i * M + j];
        E[i * M + j] = (i == j) ? T(1) : T(0);
      }
    for (int sweep = 0; sweep < 64; ++sweep) {
      T off = T(0), diag = T(0);
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          if (i != j) off += a[i * M + j] * a[i * M + j];
          else diag += a[i * M + j] * a[i * M + j];
        }
      if (!(off > T(0)) || off <= T(1e-60) * diag) break;
#pragma unroll
      for (int p = 0; p < M - 1; ++p)
#pragma unroll
        for (int q = p + 1; q < M; ++q) {
          const T apq = a[p * M + q];
          const bool skip = (apq == T(0));
          const T theta =
              (a[q * M + q] - a[p * M + p]) / (T(2) * (skip ? T(1) : apq));
          const T r = sqrt_(theta * theta + T(1));
          T t = (theta >= T(0)) ? T(1) / (theta + r) : T(-1) / (-theta + r);
          T c = T(1) / sqrt_(t * t + T(1));
          T s = t * c;
          if (skip) { c = T(1); s = T(0); }
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const T akp = a[k * M + p], akq = a[k * M + q];
            a[k * M + p] = skip ? akp : c * akp - s * akq;
            a[k * M + q] = skip ? akq : s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const T apk = a[p * M + k], aqk = a[q * M + k];
            a[p * M + k] = skip ? apk : c * apk - s * aqk;
            a[q * M + k] = skip ? aqk : s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < M; ++k) {
            const T ekp = E[k * M + p], ekq = E[k * M + q];
            E[k * M + p] = skip ? ekp : c * ekp - s * ekq;
            E[k * M + q] = skip ? ekq : s * ekp + c * ekq;
          }
        }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) e[i] = a[i * M + i];
  }
}

template <typename T, int M>
PDDP_DEV T qp_objective(const T (&Q)[M * M], const T (&c)[M], const T (&x)[M]) {
  // 0.5 * x.matmul(Q).matmul(x) + x.matmul(c)   (constraint.py:182,251)
  T quad = T(0), lin = T(0);
#pragma unroll
  for (int j = 0; j < M; ++j) {
    T xq = T(0);
#pragma unroll
    for (int i = 0; i < M; ++i) xq += x[i] * Q[i * M + j];
    quad += xq * x[j];
    lin += x[j] * c[j];
  }
  return T(0.5) * quad + lin;
}

// Projected-Newton box QP. Returns the reference's `result` code; free_bits is
// the (possibly stale, constraint.py:191-193 vs :200-204) free set, U the
// masked Cholesky factor that goes with it.
template <typename T, int M>
PDDP_DEV int boxqp(const T (&x0)[M], const T (&Q)[M * M], const T (&c)[M],
                   const T (&lower)[M], const T (&upper)[M], T (&x)[M],
                   T (&U)[M * M], unsigned& free_bits) {
  const T min_grad = T(1e-8), tol = T(1e-8), armijo = T(0.1);
  const double step_dec = 0.6, min_step = 1e-22;
  int result = 0;
  unsigned clamped = 0u;
  free_bits = (1u << M) - 1u;
  T g[M], xc[M], search[M];
  T old_f = T(0);
#pragma unroll
  for (int i = 0; i < M; ++i) {
    T v = clamp1(x0[i], lower[i], upper[i]);
    v = ((v - v != T(0)) && (v == v)) ? T(0) : v;  // x[isinf(x)] = 0
    x[i] = v;
  }
#pragma unroll
  for (int i = 0; i < M * M; ++i) U[i] = T(0);
  T f = qp_objective<T, M>(Q, c, x);

  for (int it = 0; it < 100; ++it) {
    if (it > 0 && (old_f - f) < tol * abs_(old_f)) {
      result = 4;
      break;
    }
    old_f = f;
    unsigned new_clamped = 0u;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j) s += Q[i * M + j] * x[j];
      g[i] = s + c[i];
      const bool cl = ((x[i] == lower[i]) && (g[i] > T(0))) ||
                      ((x[i] == upper[i]) && (g[i] < T(0)));
      new_clamped |= (cl ? 1u : 0u) << i;
    }
    const bool changed = (new_clamped != clamped);
    clamped = new_clamped;
    free_bits = ~clamped & ((1u << M) - 1u);
    if (free_bits == 0u) {
      result = 6;
      break;
    }
    if (it == 0 || changed) {
      if (chol_upper_masked<T, M>(Q, free_bits, U)) {
        result = -1;
        break;
      }
    }
    T gn = T(0);
#pragma unroll
    for (int i = 0; i < M; ++i)
      if ((free_bits >> i) & 1u) gn += g[i] * g[i];
    gn = sqrt_(gn);
    if (gn < min_grad) {
      result = 5;
      break;
    }
    T rhs[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      T s = T(0);
#pragma unroll
      for (int j = 0; j < M; ++j)
        s += Q[i * M + j] * (x[j] * (((clamped >> j) & 1u) ? T(1) : T(0)));
      rhs[i] = ((free_bits >> i) & 1u) ? (s + c[i]) : T(0);
    }
    chol_solve<T, M>(U, rhs);
    T sdotg = T(0);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      search[i] = ((free_bits >> i) & 1u) ? (-rhs[i] - x[i]) : T(0);
      sdotg += search[i] * g[i];
    }
    double step = 1.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
      xc[i] = clamp1(x[i] + T(step) * search[i], lower[i], upper[i]);
    T fc = qp_objective<T, M>(Q, c, xc);
    bool ls_fail = false;
    while ((fc - old_f) / (T(step) * sdotg) < armijo) {
      step *= step_dec;
#pragma unroll
      for (int i = 0; i < M; ++i)
        xc[i] = clamp1(x[i] + T(step) * search[i], lower[i], upper[i]);
      fc = qp_objective<T, M>(Q, c, xc);
      if (step < min_step) {
        ls_fail = true;
        break;
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) x[i] = xc[i];
    f = fc;
    if (ls_fail) {
      result = 2;
      break;
    }
  }
  return result;
}

}  // namespace pddp
