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

struct HD {  // hyper-dual number
  float v, a, b, ab;
};
PDDP_DEV HD hd(float v) { return HD{v, 0.f, 0.f, 0.f}; }
PDDP_DEV HD operator+(HD x, HD y) { return HD{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
PDDP_DEV HD operator-(HD x, HD y) { return HD{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
PDDP_DEV HD operator-(HD x) { return HD{-x.v, -x.a, -x.b, -x.ab}; }
PDDP_DEV HD operator*(HD x, HD y) {
  return HD{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b,
            x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
PDDP_DEV HD operator*(float s, HD x) { return HD{s * x.v, s * x.a, s * x.b, s * x.ab}; }
PDDP_DEV HD operator+(HD x, float s) { return HD{x.v + s, x.a, x.b, x.ab}; }
PDDP_DEV HD operator-(HD x, float s) { return HD{x.v - s, x.a, x.b, x.ab}; }
PDDP_DEV HD exp_(HD x) {
  const float e = expf(x.v);
  return HD{e, e * x.a, e * x.b, e * (x.ab + x.a * x.b)};
}
PDDP_DEV void sincos_(HD x, HD& s, HD& c) {
  float sv, cv;
  sincosf(x.v, &sv, &cv);
  s = HD{sv, cv * x.a, cv * x.b, cv * x.ab - sv * x.a * x.b};
  c = HD{cv, -sv * x.a, -sv * x.b, -sv * x.ab - cv * x.a * x.b};
}

constexpr int kQrMaxAng = 2, kQrMaxM = 2;

// the jittered upper Cholesky only decides which trace the cost sees
// (encoding.py:536-564: jitter 1e-12, 1e-11, ... <= 10, else the diagonal)
template <int NA>
PDDP_DEV float chol_jitter_of(const float (&C)[NA][NA], int na) {
  double jit = 1e-12;
  while (jit <= 10.0) {
    float U[NA][NA];
    bool ok = true;
    for (int i = 0; i < na && ok; ++i)
      for (int j = i; j < na; ++j) {
        float s = C[i][j] + (i == j ? (float)jit : 0.f);
        for (int q = 0; q < i; ++q) s -= U[q][i] * U[q][j];
        if (i == j) {
          if (!(s > 0.f)) { ok = false; break; }
          U[i][i] = sqrtf(s);
        } else {
          U[i][j] = s / U[i][i];
        }
      }
    if (ok) return (float)jit;
    jit *= 10.0;
  }
  return -1.f;
}

template <int D>
__global__ __launch_bounds__(64) void qr_cost_derivs_kernel(pddp_qr_cost s) {
  constexpr int n = D + D * (D + 1) / 2;
  constexpr int NA = D + kQrMaxAng;
  const int lane = threadIdx.x;
  const int N = s.N, m = s.m;
  const int bt = blockIdx.x;  // (trajectory, step), step N = terminal
  const int b = bt / (N + 1), t = bt - b * (N + 1);
  const bool terminal = (t == N);
  const int d = n + (terminal ? 0 : m);
  const int npairs = d * (d + 1) / 2;
  const int nn = s.n_non, nang = s.n_ang, na = nn + 2 * nang;
  const float* z = s.Z + ((size_t)b * (N + 1) + t) * n;
  const float* Q = terminal ? s.Q_term : s.Q;
  float u[kQrMaxM];
  for (int r = 0; r < m; ++r) {
    float v = terminal ? 0.f : s.U[((size_t)b * N + t) * m + r];
    if (!terminal && s.u_min != nullptr && s.u_max != nullptr)
      v = clamp1(v, s.u_min[r], s.u_max[r]);  // ilqr.py:461-462
    u[r] = v;
  }

  for (int q = lane; q < npairs; q += 64) {
    int i = 0, rem = q;  // pair q -> (i, j), i <= j, row-major upper triangle
    while (rem >= d - i) { rem -= d - i; ++i; }
    const int j = i + rem;
    auto in = [&](int kx, float v) {
      return HD{v, kx == i ? 1.f : 0.f, kx == j ? 1.f : 0.f, 0.f};
    };
    HD mu[D], U[D][D];
    for (int c = 0; c < D; ++c) mu[c] = in(c, z[c]);
    {
      int o = D;
      for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
          if (c >= r) { U[r][c] = in(o, z[o]); ++o; }
          else U[r][c] = hd(0.f);
        }
    }
    HD C[D][D];  // U^T U
    for (int r = 0; r < D; ++r)
      for (int c = r; c < D; ++c) {
        HD v = hd(0.f);
        for (int kx = 0; kx <= r; ++kx) v = v + U[kx][r] * U[kx][c];
        C[r][c] = v;
        C[c][r] = v;
      }
    float Cav[NA][NA];
    for (int r = 0; r < NA; ++r)
      for (int c = 0; c < NA; ++c) Cav[r][c] = 0.f;
    HD Ma[NA];
    HD tr = hd(0.f), trd = hd(0.f);  // sum Ca_ij Q_ji; sum Ca_ii Q_ii
    auto put = [&](int r, int c, HD v) {  // every entry is written once
      Cav[r][c] = v.v;
      tr = tr + Q[c * na + r] * v;
      if (r == c) trd = trd + Q[r * na + r] * v;
    };
    for (int r = 0; r < nn; ++r) {
      Ma[r] = mu[s.non[r]];
      for (int c = 0; c < nn; ++c) put(r, c, C[s.non[r]][s.non[c]]);
    }
    for (int a1 = 0; a1 < nang; ++a1) {
      const int i1 = s.ang[a1];
      const HD m1 = mu[i1], v1 = C[i1][i1];
      const HD damp = exp_(-0.5f * v1);
      HD s1, c1;
      sincos_(m1, s1, c1);
      const HD Es = damp * s1, Ec = damp * c1;
      const int r = nn + 2 * a1;
      Ma[r] = Es;
      Ma[r + 1] = Ec;
      for (int a2 = 0; a2 < nang; ++a2) {
        const int i2 = s.ang[a2];
        const HD m2 = mu[i2], v2 = C[i2][i2], cij = C[i1][i2];
        const HD lq = -0.5f * (v1 + v2), qq = exp_(lq);
        const HD ep = exp_(lq + cij) - qq, em = exp_(lq - cij) - qq;
        HD sd, cd, ss, cs;
        sincos_(m1 - m2, sd, cd);
        sincos_(m1 + m2, ss, cs);
        const int cc = nn + 2 * a2;
        put(r, cc, 0.5f * (ep * cd - em * cs));          // sin, sin
        put(r + 1, cc + 1, 0.5f * (ep * cd + em * cs));  // cos, cos
        const HD sc = 0.5f * (ep * sd + em * ss);        // sin_1, cos_2
        put(r, cc + 1, sc);
        put(cc + 1, r, sc);
      }
      for (int c = 0; c < nn; ++c) {
        const HD col = C[s.non[c]][i1];
        put(c, r, col * Ec);        // Cov(x, sin)
        put(c, r + 1, -(col * Es));  // Cov(x, cos)
        put(r, c, col * Ec);
        put(r + 1, c, -(col * Es));
      }
    }
    const float jit = chol_jitter_of<NA>(Cav, na);
    HD cost = hd(0.f);
    for (int c = 0; c < na; ++c) {
      HD row = hd(0.f);
      for (int r = 0; r < na; ++r) row = row + Q[r * na + c] * (Ma[r] - s.x_goal[r]);
      cost = cost + row * (Ma[c] - s.x_goal[c]);
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
      float trq = 0.f;
      for (int r = 0; r < na; ++r) trq += Q[r * na + r];
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

}  // namespace pddp

extern "C" int pddp_qr_cost_derivs_f32(const pddp_qr_cost* s, void* stream) {
  if (s == nullptr) return PDDP_E_BADARG;
  if (s->B <= 0 || s->N <= 0 || !s->Z || !s->U || !s->Q || !s->Q_term ||
      !s->R || !s->x_goal || !s->u_goal || !s->L || !s->L_z || !s->L_u ||
      !s->L_zz || !s->L_uz || !s->L_uu)
    return PDDP_E_BADARG;
  if (s->m < 1 || s->m > pddp::kQrMaxM || s->n_ang < 0 ||
      s->n_ang > pddp::kQrMaxAng || s->n_non < 0 || s->n_non + s->n_ang != s->D)
    return PDDP_E_UNSUPPORTED;
  const dim3 grid(s->B * (s->N + 1)), block(64);
  hipStream_t st = (hipStream_t)stream;
  switch (s->D) {
    case 2: PDDP_LAUNCH(pddp::qr_cost_derivs_kernel<2>, grid, block, 0, st, *s); break;
    case 4: PDDP_LAUNCH(pddp::qr_cost_derivs_kernel<4>, grid, block, 0, st, *s); break;
    case 6: PDDP_LAUNCH(pddp::qr_cost_derivs_kernel<6>, grid, block, 0, st, *s); break;
    default: return PDDP_E_UNSUPPORTED;
  }
  return pddp::launch_status();
}
