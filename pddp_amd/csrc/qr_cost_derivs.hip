// qr_cost_derivs.hip - value, gradient and Hessian of the quadratic cost on the
// angle-augmented Gaussian state under the DEFAULT encoding
// (z = mean | triu(upper Cholesky of the covariance)):
//   l(z, u) = (Ma - g)^T Q (Ma - g) + tr(Q Ca'') + (u - u_g)^T R (u - u_g)
// with (Ma, Ca) the moment-matched augmentation [non-angular..., sin, cos ...]
// of (mean, U^T U) and Ca'' its jittered re-encoding
// (pddp/costs/quadratic.py:60-99, pddp/examples/cartpole/cost.py:60-87,
// pddp/utils/angular.py:47-84,161-248, pddp/utils/encoding.py:99-141,536-564).
//
// The reference gets L_z, L_u, L_zz, L_uz, L_uu from autograd's double backward
// over (n + m) replicated inputs per time step (utils/evaluation.py:238-288):
// ~2000 tiny launches per step, 5 s per derivative rollout at B = 4096, N =
// 100.  Here the cost is written once over a scalar type and evaluated on
// hyper-dual numbers (v, d/dx_i, d/dx_j, d2/dx_i dx_j): one lane per pair
// (i <= j) of the n + m inputs, 120 pairs for cartpole, every (trajectory, time
// step) in one launch.  The terminal state (no action, Q_term) is step N.
#include "pddp_common.hpp"

namespace pddp {

PDDP_DEV float qr_exp(float x) { return expf(x); }
PDDP_DEV double qr_exp(double x) { return exp(x); }
PDDP_DEV void qr_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
PDDP_DEV void qr_sincos(double x, double& s, double& c) { sincos(x, &s, &c); }

// pddp_qr_cost / pddp_qr_cost_f64 with the scalar type as a parameter
template <typename T>
struct QrCostV {
  int32_t B, N, D, m;
  int32_t n_ang, ang[2], n_non, non[8];
  const T* Z;
  const T* U;
  const T* u_min;
  const T* u_max;
  const T* Q;
  const T* Q_term;
  const T* R;
  const T* x_goal;
  const T* u_goal;
  T* L;
  T* L_z;
  T* L_u;
  T* L_zz;
  T* L_uz;
  T* L_uu;
};
static_assert(sizeof(QrCostV<float>) == sizeof(pddp_qr_cost) &&
              sizeof(QrCostV<double>) == sizeof(pddp_qr_cost_f64), "");

template <typename T>
struct HDT {  // hyper-dual number
  typedef T S;
  T v, a, b, ab;
};
template <typename T> PDDP_DEV HDT<T> hd_(typename HDT<T>::S v) { return HDT<T>{v, 0.f, 0.f, 0.f}; }
template <typename T> PDDP_DEV HDT<T> operator+(HDT<T> x, HDT<T> y) { return HDT<T>{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
template <typename T> PDDP_DEV HDT<T> operator-(HDT<T> x, HDT<T> y) { return HDT<T>{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
template <typename T> PDDP_DEV HDT<T> operator-(HDT<T> x) { return HDT<T>{-x.v, -x.a, -x.b, -x.ab}; }
template <typename T> PDDP_DEV HDT<T> operator*(HDT<T> x, HDT<T> y) {
  return HDT<T>{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
            x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
template <typename T> PDDP_DEV HDT<T> operator*(typename HDT<T>::S s, HDT<T> x) { return HDT<T>{s * x.v, s * x.a, s * x.b, s * x.ab}; }
template <typename T> PDDP_DEV HDT<T> operator+(HDT<T> x, typename HDT<T>::S s) { return HDT<T>{x.v + s, x.a, x.b, x.ab}; }
template <typename T> PDDP_DEV HDT<T> operator-(HDT<T> x, typename HDT<T>::S s) { return HDT<T>{x.v - s, x.a, x.b, x.ab}; }
template <typename T> PDDP_DEV HDT<T> exp_(HDT<T> x) {
  const T e = qr_exp(x.v);
  return HDT<T>{e, e * x.a, e * x.b, e * (x.ab + x.a * x.b)};
}
template <typename T> PDDP_DEV void sincos_(HDT<T> x, HDT<T>& s, HDT<T>& c) {
  T sv, cv;
  qr_sincos(x.v, sv, cv);
  s = HDT<T>{sv, cv * x.a, cv * x.b, cv * x.ab - sv * x.a * x.b};
  c = HDT<T>{cv, -sv * x.a, -sv * x.b, -sv * x.ab - cv * x.a * x.b};
}

constexpr int kQrMaxAng = 2, kQrMaxM = 2;

// the jittered upper Cholesky only decides which trace the cost sees
// (encoding.py:536-564: jitter 1e-12, 1e-11, ... <= 10, else the diagonal)
template <typename T, int NA>
PDDP_DEV T chol_jitter_of(const T (&C)[NA][NA]) {
  double jit = 1e-12;
  while (jit <= 10.0) {
    T U[NA][NA];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int j = i; j < NA; ++j) {
        T s = C[i][j] + (i == j ? (T)jit : 0.f);
#pragma unroll
        for (int q = 0; q < i; ++q) s -= U[q][i] * U[q][j];
        if (i == j) {
          // (a failed pivot ends the reference's attempt; what follows here
          // is computed on garbage and discarded)
          if (!(s > 0.f)) ok = false;
          U[i][i] = sqrt_(s);
        } else {
          U[i][j] = s / U[i][i];
        }
      }
    }
    if (ok) return (T)jit;
    jit *= 10.0;
  }
  return -1.f;
}

// D state dimensions of which NANG are angles: every index below is a compile-
// time constant (the state is permuted to non-angular | angular as it is
// loaded), so the hyper-dual working set - U, U^T U, the augmented moments -
// lives in registers.  (Indexed through the runtime `non` / `ang` lists it
// lived in scratch memory: 149 GB of HBM traffic and 49 ms per launch for the
// double cartpole, D = 6; 10 ms for cartpole.)
template <typename T, int D, int NANG>
__global__ __launch_bounds__(64) void qr_cost_derivs_kernel(QrCostV<T> s) {
  using HD = HDT<T>;
  auto hd = [](T v) { return HD{v, 0.f, 0.f, 0.f}; };
  constexpr int n = D + D * (D + 1) / 2;
  constexpr int NN = D - NANG;       // non-angular dimensions
  constexpr int NA = NN + 2 * NANG;  // augmented dimensions
  const int lane = threadIdx.x;
  const int N = s.N, m = s.m;
  const int bt = blockIdx.x;  // (trajectory, step), step N = terminal
  const int b = bt / (N + 1), t = bt - b * (N + 1);
  const bool terminal = (t == N);
  const int d = n + (terminal ? 0 : m);
  const int npairs = d * (d + 1) / 2;
  const T* z = s.Z + ((size_t)b * (N + 1) + t) * n;
  const T* Q = terminal ? s.Q_term : s.Q;
  T u[kQrMaxM];
  for (int r = 0; r < m; ++r) {
    T v = terminal ? 0.f : s.U[((size_t)b * N + t) * m + r];
    if (!terminal && s.u_min != nullptr && s.u_max != nullptr)
      v = clamp1(v, s.u_min[r], s.u_max[r]);  // ilqr.py:461-462
    u[r] = v;
  }
  // state dimension held at permuted position a
  int perm[D];
#pragma unroll
  for (int a = 0; a < D; ++a) perm[a] = a < NN ? s.non[a] : s.ang[a - NN];
  // the inputs' values, permuted: mean, and row k of the Cholesky factor at
  // the permuted columns (zero below the diagonal) with its index in z
  T zmu[D], zU[D][D];
  int oU[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    zmu[a] = z[perm[a]];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int c = perm[a];
      // row-major upper triangle: rows 0 .. k - 1 hold D, D - 1, .. entries
      const int o = D + k * D - (k * (k - 1)) / 2 + (c - k);
      const bool up = c >= k;
      oU[k][a] = up ? o : -1;
      zU[k][a] = up ? z[up ? o : 0] : 0.f;
    }
  }
  T Qr[NA][NA], goal[NA];
#pragma unroll
  for (int r = 0; r < NA; ++r) {
    goal[r] = s.x_goal[r];
#pragma unroll
    for (int c = 0; c < NA; ++c) Qr[r][c] = Q[r * NA + c];
  }

  T jit = 0.f;
  bool have_jit = false;
  for (int q = lane; q < npairs; q += 64) {
    int i = 0, rem = q;  // pair q -> (i, j), i <= j, row-major upper triangle
    while (rem >= d - i) { rem -= d - i; ++i; }
    const int j = i + rem;
    auto in = [&](int kx, T v) {
      return HD{v, kx == i ? 1.f : 0.f, kx == j ? 1.f : 0.f, 0.f};
    };
    HD mu[D];
#pragma unroll
    for (int a = 0; a < D; ++a) mu[a] = in(perm[a], zmu[a]);
    HD C[D][D];  // (U^T U) in permuted order; terms in ascending row order
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = a; c < D; ++c) C[a][c] = hd(0.f);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      HD row[D];
#pragma unroll
      for (int a = 0; a < D; ++a) row[a] = in(oU[k][a], zU[k][a]);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = a; c < D; ++c) C[a][c] = C[a][c] + row[a] * row[c];
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c < a; ++c) C[a][c] = C[c][a];

    T Cav[NA][NA];
    HD Ma[NA];
    HD tr = hd(0.f), trd = hd(0.f);  // sum Ca_ij Q_ji; sum Ca_ii Q_ii
    auto put = [&](int r, int c, HD v) {  // every entry is written once
      Cav[r][c] = v.v;
      tr = tr + Qr[c][r] * v;
      if (r == c) trd = trd + Qr[r][r] * v;
    };
#pragma unroll
    for (int r = 0; r < NN; ++r) {
      Ma[r] = mu[r];
#pragma unroll
      for (int c = 0; c < NN; ++c) put(r, c, C[r][c]);
    }
#pragma unroll
    for (int a1 = 0; a1 < NANG; ++a1) {
      const int i1 = NN + a1;
      const HD m1 = mu[i1], v1 = C[i1][i1];
      const HD damp = exp_(-0.5f * v1);
      HD s1, c1;
      sincos_(m1, s1, c1);
      const HD Es = damp * s1, Ec = damp * c1;
      const int r = NN + 2 * a1;
      Ma[r] = Es;
      Ma[r + 1] = Ec;
#pragma unroll
      for (int a2 = 0; a2 < NANG; ++a2) {
        const int i2 = NN + a2;
        const HD m2 = mu[i2], v2 = C[i2][i2], cij = C[i1][i2];
        const HD lq = -0.5f * (v1 + v2), qq = exp_(lq);
        const HD ep = exp_(lq + cij) - qq, em = exp_(lq - cij) - qq;
        HD sd, cd, ss, cs;
        sincos_(m1 - m2, sd, cd);
        sincos_(m1 + m2, ss, cs);
        const int cc = NN + 2 * a2;
        put(r, cc, 0.5f * (ep * cd - em * cs));          // sin, sin
        put(r + 1, cc + 1, 0.5f * (ep * cd + em * cs));  // cos, cos
        const HD sc = 0.5f * (ep * sd + em * ss);        // sin_1, cos_2
        put(r, cc + 1, sc);
        put(cc + 1, r, sc);
      }
#pragma unroll
      for (int c = 0; c < NN; ++c) {
        const HD col = C[c][i1];
        put(c, r, col * Ec);        // Cov(x, sin)
        put(c, r + 1, -(col * Es));  // Cov(x, cos)
        put(r, c, col * Ec);
        put(r + 1, c, -(col * Es));
      }
    }
    if (!have_jit) {  // values only: the same for every pair of this step
      jit = chol_jitter_of<T, NA>(Cav);
      have_jit = true;
    }
    HD cost = hd(0.f);
#pragma unroll
    for (int c = 0; c < NA; ++c) {
      HD row = hd(0.f);
#pragma unroll
      for (int r = 0; r < NA; ++r) row = row + Qr[r][c] * (Ma[r] - goal[r]);
      cost = cost + row * (Ma[c] - goal[c]);
    }
    if (!terminal) {
      for (int c = 0; c < m; ++c) {
        HD row = hd(0.f);
        for (int r = 0; r < m; ++r)
          row = row + s.R[r * m + c] * (in(n + r, u[r]) - s.u_goal[r]);
        cost = cost + row * (in(n + c, u[c]) - s.u_goal[c]);
      }
    }
    if (jit >= 0.f) {
      T trq = 0.f;
#pragma unroll
      for (int r = 0; r < NA; ++r) trq += Qr[r][r];
      cost = cost + tr + jit * trq;  // tr(Q (Ca + jitter I))
    } else {
      cost = cost + trd;             // encode()'s diagonal fall-back
    }

    // ---- scatter: value, gradient (diagonal pairs), Hessian (both triangles)
    const size_t st = (size_t)b * (N + 1) + t;  // state index incl. terminal
    const size_t su = (size_t)b * N + t;        // step index (t < N)
    if (q == 0) s.L[st] = cost.v;
    if (i == j) {
      if (i < n) s.L_z[st * n + i] = cost.a;
      else s.L_u[su * m + (i - n)] = cost.a;
    }
    if (j < n) {
      s.L_zz[(st * n + i) * n + j] = cost.ab;
      s.L_zz[(st * n + j) * n + i] = cost.ab;
    } else if (i < n) {
      s.L_uz[(su * m + (j - n)) * n + i] = cost.ab;
    } else {
      s.L_uu[(su * m + (i - n)) * m + (j - n)] = cost.ab;
      s.L_uu[(su * m + (j - n)) * m + (i - n)] = cost.ab;
    }
  }
}

template <typename T, int D>
static int launch_qr_cost(const QrCostV<T>& s, hipStream_t st) {
  const dim3 grid(s.B * (s.N + 1)), block(64);
  switch (s.n_ang) {
    case 0: PDDP_LAUNCH((qr_cost_derivs_kernel<T, D, 0>), grid, block, 0, st, s); break;
    case 1: PDDP_LAUNCH((qr_cost_derivs_kernel<T, D, 1>), grid, block, 0, st, s); break;
    case 2: PDDP_LAUNCH((qr_cost_derivs_kernel<T, D, 2>), grid, block, 0, st, s); break;
    default: return PDDP_E_UNSUPPORTED;
  }
  return launch_status();
}


template <typename T>
static int qr_cost_derivs(const QrCostV<T>* s, void* stream) {
  if (s == nullptr) return PDDP_E_BADARG;
  if (s->B <= 0 || s->N <= 0 || !s->Z || !s->U || !s->Q || !s->Q_term ||
      !s->R || !s->x_goal || !s->u_goal || !s->L || !s->L_z || !s->L_u ||
      !s->L_zz || !s->L_uz || !s->L_uu)
    return PDDP_E_BADARG;
  if (s->m < 1 || s->m > kQrMaxM || s->n_ang < 0 ||
      s->n_ang > kQrMaxAng || s->n_non < 0 || s->n_non + s->n_ang != s->D)
    return PDDP_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (s->D) {
    case 2: return launch_qr_cost<T, 2>(*s, st);
    case 4: return launch_qr_cost<T, 4>(*s, st);
    case 6: return launch_qr_cost<T, 6>(*s, st);
  }
  return PDDP_E_UNSUPPORTED;
}

}  // namespace pddp

extern "C" int pddp_qr_cost_derivs_f32(const pddp_qr_cost* s, void* stream) {
  return pddp::qr_cost_derivs(reinterpret_cast<const pddp::QrCostV<float>*>(s), stream);
}
extern "C" int pddp_qr_cost_derivs_f64(const pddp_qr_cost_f64* s, void* stream) {
  return pddp::qr_cost_derivs(reinterpret_cast<const pddp::QrCostV<double>*>(s), stream);
}
