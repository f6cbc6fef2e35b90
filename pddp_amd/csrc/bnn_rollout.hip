// bnn_rollout.hip - one time step of the moment-matched line-search rollout
// under a BNN dynamics model, everything except the network itself
// (csrc/bnn_mlp.hip): what pddp/controllers/ilqr.py:677-723 (_control_law),
// :764-791 (_trajectory_cost), pddp/models/bnn/modules.py:287-386
// (BNNDynamicsModel.forward), pddp/utils/encoding.py:99-141 (encode, DEFAULT =
// mean | upper Cholesky) and pddp/utils/angular.py:47-84,161-248 (moment-matched
// angle augmentation for the cost) do per candidate and time step - in the
// framework that is ~150 small launches per step; here it is one.
//
// One wavefront per candidate (b, alpha); lanes own particles.  Step t:
//   X_t = X_{t-1} + net_out * dX_std + dX_mean          (t > 0; modules.py:262)
//   z_t = encode(mean_p X_t, cov_p X_t)   (t > 0; z_0 is the nominal's start)
//   u_t = clamp(U_t + alpha k_t + K_t (z_t - Z_t)),  J += l(z_t, u_t)   (t < N)
//   F   = ((augment(X_t) | u_t) - X_mean) * X_std_inv   -> the network's input
// and at t = N the terminal cost.  With `infer_noise_variables` the reference
// re-whitens the previous output particles with the Cholesky factor it then
// re-colours them with (modules.py:333-348): the particle cloud is simply
// carried from step to step, which is what X_t above does.
//
// The per-candidate algebra (4x4 ... 8x8 Cholesky with the reference's jitter
// escalation, augmentation moments, quadratic cost) runs on lane 0 out of LDS:
// it is ~1 kFLOP per step against the network's 8.6 MFLOP, and there are tens
// of thousands of candidates to fill the machine.
#include "pddp_common.hpp"

namespace pddp {

constexpr int kBnnMaxD = 8, kBnnMaxAng = 2, kBnnMaxM = 2;
constexpr int kBnnMaxNa = kBnnMaxD + kBnnMaxAng;            // augmented size
constexpr int kBnnMaxN = kBnnMaxD + kBnnMaxD * (kBnnMaxD + 1) / 2;  // 44

PDDP_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// upper Cholesky of the d x d matrix C + jitter I (row-major, ld = kBnnMaxNa),
// false when a pivot is not positive (encoding.py:536-564 tries jitter = 1e-12,
// 1e-11, ... <= 10)
PDDP_DEV bool chol_upper(const float* C, int d, float jitter, float* U) {
  for (int i = 0; i < d; ++i) {
    for (int j = i; j < d; ++j) {
      float s = C[i * kBnnMaxNa + j] + (i == j ? jitter : 0.f);
      for (int k = 0; k < i; ++k) s -= U[k * kBnnMaxNa + i] * U[k * kBnnMaxNa + j];
      if (i == j) {
        if (!(s > 0.f)) return false;
        U[i * kBnnMaxNa + i] = sqrtf(s);
      } else {
        U[i * kBnnMaxNa + j] = s / U[i * kBnnMaxNa + i];
      }
    }
    for (int j = 0; j < i; ++j) U[i * kBnnMaxNa + j] = 0.f;
  }
  return true;
}
// with the escalation; returns the jitter used, -1 when none up to 10 works
PDDP_DEV float chol_upper_jittered(const float* C, int d, float* U) {
  double jit = 1e-12;  // python float in the reference
  while (true) {
    if (chol_upper(C, d, (float)jit, U)) return (float)jit;
    jit *= 10.0;
    if (jit > 10.0) return -1.f;
  }
}

__global__ __launch_bounds__(64) void bnn_moment_step_kernel(pddp_bnn_step s) {
  __shared__ float Ms[kBnnMaxD], Cs[kBnnMaxNa * kBnnMaxNa], Us[kBnnMaxNa * kBnnMaxNa];
  __shared__ float zs[kBnnMaxN], us[kBnnMaxM];
  __shared__ float Ma[kBnnMaxNa], Ca[kBnnMaxNa * kBnnMaxNa], Ua[kBnnMaxNa * kBnnMaxNa];

  const int c = blockIdx.x;  // candidate = b * A + ai
  const int lane = threadIdx.x;
  const int b = c / s.A, ai = c - b * s.A;
  if (s.active != nullptr && s.active[b] == 0) return;
  if (s.bwd_status != nullptr && s.bwd_status[b] != 0) return;
  const int D = s.D, P = s.P, m = s.m, N = s.N, t = s.t;
  const int n = D + D * (D + 1) / 2;
  const int na = s.n_non + 2 * s.n_ang;
  const bool terminal = (t == N);

  // ---- particles of this step (two per lane: P <= 128)
  float x[2][kBnnMaxD];
  bool has[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = lane + 64 * q;
    has[q] = p < P;
    const size_t row = (size_t)c * P + (has[q] ? p : 0);
    for (int d = 0; d < D; ++d) {
      float v = s.Xp[row * D + d];
      if (t > 0)  // X + dx, dx = out[:D] * dX_std + dX_mean     (modules.py:262)
        v = v + (s.net_out[row * s.out_dim + d] * s.dX_std[d] + s.dX_mean[d]);
      x[q][d] = v;
      if (t > 0 && has[q]) s.Xp[row * D + d] = v;
    }
  }

  // ---- z_t
  if (t == 0) {
    if (lane < n) zs[lane] = s.Z[((size_t)b * (N + 1)) * n + lane];
    __syncthreads();
    if (lane == 0) {  // mean and covariance factor back out of z_0 (cost)
      for (int d = 0; d < D; ++d) Ms[d] = zs[d];
      int o = D;
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
          Us[i * kBnnMaxNa + j] = j >= i ? zs[o++] : 0.f;
    }
  } else {
    // moments over the particles (modules.py:372-386): mean, then the
    // unbiased covariance of the deviations
    for (int d = 0; d < D; ++d) {
      float v = (has[0] ? x[0][d] : 0.f) + (has[1] ? x[1][d] : 0.f);
      v = wave_sum(v) / (float)P;
      if (lane == 0) Ms[d] = v;
      x[0][d] -= v;  // deviations from here on (features add the mean back)
      x[1][d] -= v;
    }
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) {
        float v = (has[0] ? x[0][i] * x[0][j] : 0.f) +
                  (has[1] ? x[1][i] * x[1][j] : 0.f);
        v = wave_sum(v) / (float)(P - 1);
        if (lane == 0) {
          Cs[i * kBnnMaxNa + j] = v;
          Cs[j * kBnnMaxNa + i] = v;
        }
      }
    __syncthreads();
    if (lane == 0) {
      // encode (encoding.py:99-141): jittered upper Cholesky; not positive
      // definite even with jitter 10 -> the diagonal of standard deviations
      if (chol_upper_jittered(Cs, D, Us) < 0.f) {
        for (int i = 0; i < D; ++i)
          for (int j = 0; j < D; ++j)
            Us[i * kBnnMaxNa + j] =
                i == j ? sqrtf(Cs[i * kBnnMaxNa + i]) : 0.f;
      }
      for (int d = 0; d < D; ++d) zs[d] = Ms[d];
      int o = D;
      for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) zs[o++] = Us[i * kBnnMaxNa + j];
    }
    __syncthreads();
    for (int d = 0; d < D; ++d) {  // particles again (Ms is visible now)
      x[0][d] += Ms[d];
      x[1][d] += Ms[d];
    }
  }
  __syncthreads();
  if (lane < n)
    s.Zc[(((size_t)b * (N + 1) + t) * s.A + ai) * n + lane] = zs[lane];

  // ---- control law, cost (lane 0)
  if (lane == 0) {
    if (!terminal) {
      const int GS = m + m * n;
      const float* g = s.gains + ((size_t)b * N + t) * GS;
      const float* zn = s.Z + ((size_t)b * (N + 1) + t) * n;
      for (int r = 0; r < m; ++r) {
        float du = s.alphas[ai] * g[r];                     // ilqr.py:708
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += (zs[k] - zn[k]) * g[m + r * n + k];
        du = du + acc;                                      // ilqr.py:710
        float v = s.U[((size_t)b * N + t) * m + r] + du;
        if (s.u_min != nullptr && s.u_max != nullptr)
          v = clamp1(v, s.u_min[r], s.u_max[r]);
        us[r] = v;
        s.Uc[(((size_t)b * N + t) * s.A + ai) * m + r] = v;
      }
    }
    // cost on the angle-augmented moments (examples/*/cost.py, quadratic.py:
    // 60-99, angular.py:161-248).  Covariance as the cost sees it: C = U^T U.
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        float v = 0.f;
        for (int k = 0; k < D; ++k) v += Us[k * kBnnMaxNa + i] * Us[k * kBnnMaxNa + j];
        Cs[i * kBnnMaxNa + j] = v;
      }
    const int nn = s.n_non, nang = s.n_ang;
    for (int i = 0; i < na * kBnnMaxNa; ++i) Ca[i] = 0.f;
    for (int i = 0; i < nn; ++i) {
      Ma[i] = Ms[s.non[i]];
      for (int j = 0; j < nn; ++j)
        Ca[i * kBnnMaxNa + j] = Cs[s.non[i] * kBnnMaxNa + s.non[j]];
    }
    for (int a1 = 0; a1 < nang; ++a1) {
      const int i1 = s.ang[a1];
      const float m1 = Ms[i1], v1 = Cs[i1 * kBnnMaxNa + i1];
      const float damp = expf(-0.5f * v1);
      const float Es = damp * sinf(m1), Ec = damp * cosf(m1);
      Ma[nn + 2 * a1] = Es;
      Ma[nn + 2 * a1 + 1] = Ec;
      for (int a2 = 0; a2 < nang; ++a2) {
        const int i2 = s.ang[a2];
        const float m2 = Ms[i2], v2 = Cs[i2 * kBnnMaxNa + i2];
        const float cij = Cs[i1 * kBnnMaxNa + i2];
        const float lq = -0.5f * (v1 + v2), q = expf(lq);
        const float ep = expf(lq + cij) - q, em = expf(lq - cij) - q;
        const float dm = m1 - m2, sm = m1 + m2;
        const int r = nn + 2 * a1, cc = nn + 2 * a2;
        Ca[r * kBnnMaxNa + cc] = 0.5f * (ep * cosf(dm) - em * cosf(sm));           // sin, sin
        Ca[(r + 1) * kBnnMaxNa + cc + 1] = 0.5f * (ep * cosf(dm) + em * cosf(sm)); // cos, cos
        Ca[r * kBnnMaxNa + cc + 1] = 0.5f * (ep * sinf(dm) + em * sinf(sm));       // sin, cos
        // (cos_i, sin_j) = (sin_j, cos_i): written when the roles swap
        Ca[(cc + 1) * kBnnMaxNa + r] = Ca[r * kBnnMaxNa + cc + 1];
      }
      for (int i = 0; i < nn; ++i) {
        const float col = Cs[s.non[i] * kBnnMaxNa + i1];
        const int r = nn + 2 * a1;
        Ca[i * kBnnMaxNa + r] = col * Ec;        // Cov(x, sin)
        Ca[i * kBnnMaxNa + r + 1] = -col * Es;   // Cov(x, cos)
        Ca[r * kBnnMaxNa + i] = col * Ec;
        Ca[(r + 1) * kBnnMaxNa + i] = -col * Es;
      }
    }
    // the cost re-encodes the augmented covariance (a jittered Cholesky) and
    // decodes it again: C'' = Ca + jitter I for the first jitter that works
    float jit = chol_upper_jittered(Ca, na, Ua);
    const float* Q = terminal ? s.Q_term : s.Q;
    float cost = 0.f;
    for (int i = 0; i < na; ++i) {
      float row = 0.f;
      for (int j = 0; j < na; ++j) row += (Ma[j] - s.x_goal[j]) * Q[j * na + i];
      cost += row * (Ma[i] - s.x_goal[i]);
    }
    if (!terminal) {
      for (int i = 0; i < m; ++i) {
        float row = 0.f;
        for (int j = 0; j < m; ++j) row += (us[j] - s.u_goal[j]) * s.R[j * m + i];
        cost += row * (us[i] - s.u_goal[i]);
      }
    }
    float tr = 0.f;
    if (jit >= 0.f) {
      for (int i = 0; i < na; ++i)
        for (int j = 0; j < na; ++j) {
          float cij = 0.f;  // (Ua^T Ua)[i][j]
          for (int k = 0; k < na; ++k) cij += Ua[k * kBnnMaxNa + i] * Ua[k * kBnnMaxNa + j];
          tr += cij * Q[j * na + i];
        }
    } else {  // diagonal fallback of encode(): variances only
      for (int i = 0; i < na; ++i) tr += Ca[i * kBnnMaxNa + i] * Q[i * na + i];
    }
    cost += tr;
    const float J = (t == 0 ? 0.f : s.J[c]) + cost;
    s.J[c] = J;
    if (terminal) s.Jc[c] = J;
  }
  if (terminal) return;
  __syncthreads();

  // ---- the network's input rows for this candidate's particles
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (!has[q]) continue;
    const size_t row = (size_t)c * P + lane + 64 * q;
    float* f = s.F + row * s.in_dim;
    int o = 0;
    for (int i = 0; i < s.n_non; ++i, ++o)
      f[o] = (x[q][s.non[i]] - s.X_mean[o]) * s.X_std_inv[o];
    for (int a1 = 0; a1 < s.n_ang; ++a1) {
      float sn, cs;
      sincosf(x[q][s.ang[a1]], &sn, &cs);
      f[o] = (sn - s.X_mean[o]) * s.X_std_inv[o]; ++o;
      f[o] = (cs - s.X_mean[o]) * s.X_std_inv[o]; ++o;
    }
    for (int r = 0; r < m; ++r, ++o)
      f[o] = (us[r] - s.X_mean[o]) * s.X_std_inv[o];
  }
}

}  // namespace pddp

extern "C" int pddp_bnn_moment_step_f32(const pddp_bnn_step* s, void* stream) {
  if (s == nullptr) return PDDP_E_BADARG;
  if (s->B <= 0 || s->A <= 0 || s->P <= 1 || s->N <= 0 || s->t < 0 ||
      s->t > s->N || !s->Z || !s->U || !s->gains || !s->alphas || !s->Q ||
      !s->Q_term || !s->R || !s->x_goal || !s->u_goal || !s->X_mean ||
      !s->X_std_inv || !s->dX_mean || !s->dX_std || !s->Xp || !s->F || !s->Zc ||
      !s->Uc || !s->J || !s->Jc || (s->t > 0 && !s->net_out))
    return PDDP_E_BADARG;
  if (s->D < 1 || s->D > pddp::kBnnMaxD || s->m < 1 || s->m > pddp::kBnnMaxM ||
      s->P > 128 || s->n_ang < 0 || s->n_ang > pddp::kBnnMaxAng ||
      s->n_non < 0 || s->n_non + s->n_ang != s->D ||
      s->in_dim != s->n_non + 2 * s->n_ang + s->m || s->out_dim < s->D)
    return PDDP_E_UNSUPPORTED;
  PDDP_LAUNCH(pddp::bnn_moment_step_kernel, dim3(s->B * s->A), dim3(64),
                     0, (hipStream_t)stream, *s);
  return pddp::launch_status();
}
