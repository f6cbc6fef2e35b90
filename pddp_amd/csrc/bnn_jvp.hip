// bnn_jvp.hip - Jacobians of one moment-matched BNN dynamics step in forward
// mode, everything except the network itself (csrc/bnn_mlp.hip, JVP mode).
//
// What the reference obtains per time step from autograd on n replicated
// inputs (pddp/controllers/ilqr.py:457-468 -> pddp/utils/evaluation.py:203-235
// batch_eval_dynamics through pddp/models/bnn/modules.py:287-386):
//   F_z = d z' / d z,  F_u = d z' / d u,   z = (mean | triu(upper Cholesky))
// of   z' = encode(mean_p(out), cov_p(out)),   out_p = X_p + dX_std * net(
//          ((augment(X_p) | u) - X_mean) * X_std_inv)[:D] + dX_mean,
//      X_p = mean + eps_p U      with eps DETACHED: `infer_noise_variables`
//          re-whitens the previous call's output particles,
//          eps = (out_prev - mean) U^-1 (modules.py:333-348), a constant of the
//          differentiation.
// Forward mode: one tangent direction per input coordinate, 15 for cartpole
// (D = 4: 4 means, 10 Cholesky entries, 1 action), pushed through the network
// together with the primal row as a group of 16 rows (bnn_mlp.hip JVP mode):
//   jvp_features:  (X_p, z_t, u_t) -> eps_p, the primal input row and the 15
//                  tangent rows  d feat / d (mean_d | U_ab | u);
//   jvp_moments :  network output rows -> out_p (the particle cloud carried to
//                  step t + 1), M, C, U' = chol(C), and per direction k
//                  dM_k = mean_p dout_pk,
//                  dC_k = (S_k + S_k^T) / (P - 1),  S_k = sum_p dout_pk (out_p - M)^T,
//                  dU'_k = Phi(U'^-T dC_k U'^-1) U'   (Phi: upper triangle, half
//                  the diagonal - the differential of the Cholesky factor),
//                  column k of (F_z | F_u) = (dM_k | triu(dU'_k)).
// Only the D + m directions (mean_d | u) go through the network: X = mean +
// eps U makes the tangent of X for a Cholesky direction U_ab the PARTICLE'S
// scalar eps[a] times its tangent for mean_b, and everything downstream of X is
// linear in the tangent, so d out / d U_ab = eps[a] d out / d mean_b per
// particle.  Network rows per (state, particle): 8 = input + D + m <= 7
// tangents (instead of 16 / 32: the Cholesky directions were 10 of cartpole's
// 15, 21 of the double cartpole's 28).  jvp_moments still has one lane per
// direction of z (G = 16 or 32 lanes per trajectory), each reading its base
// direction's row and scaling.
#include "pddp_common.hpp"
#include "models.hpp"  // sincos_

namespace pddp {

// G = rows per (state, particle): the input row + up to G - 1 tangent rows.
// G = 16: D <= 4 (cartpole: 4 + 10 + 1 directions); G = 32: D <= 6 (double
// cartpole: 6 + 21 + 1).

constexpr int kNetRows = 8;  // network rows per (state, particle)

// EXACT: D == kJvpMaxD, known at compile time - every small matrix below is
// then indexed statically and lives in registers (under a runtime D they sat in
// scratch memory: 160 .. 1024 bytes per lane, and the kernels waited for it).
PDDP_DEV float jvp_exp(float x) { return expf(x); }
PDDP_DEV double jvp_exp(double x) { return exp(x); }
PDDP_DEV float jvp_log(float x) { return logf(x); }
PDDP_DEV double jvp_log(double x) { return log(x); }

// pddp_bnn_jvp / pddp_bnn_jvp_f64 (include/pddp_hip.h) with the scalar type as
// a parameter: the two structs differ in the pointee type only
template <typename T>
struct BnnJvpV {
  int32_t B, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8];
  int32_t in_dim, out_dim;
  const T* Z;
  const T* U;
  const T* u_min;
  const T* u_max;
  const T* X_mean;
  const T* X_std_inv;
  const T* dX_mean;
  const T* dX_std;
  const T* net_out;
  const T* Xp;
  T* Xp_next;
  T* eps;
  T* F;
  T* Z_next;
  T* F_z;
  T* F_u;
  const T* eps_out;
  int32_t independent_noise;
  const int32_t* slot;
};
static_assert(sizeof(BnnJvpV<float>) == sizeof(pddp_bnn_jvp) &&
              sizeof(BnnJvpV<double>) == sizeof(pddp_bnn_jvp_f64), "");

template <typename T, int kJvpMaxD, bool EXACT>
__global__ __launch_bounds__(64) void bnn_jvp_features_kernel(BnnJvpV<T> s) {
  constexpr int kJvpRows = kNetRows;
  const int lane = threadIdx.x;
  const int k = lane & (kJvpRows - 1);  // row of the group
  const int bp = blockIdx.x * (64 / kJvpRows) + lane / kJvpRows;  // (b, p)
  if (bp >= s.B * s.P) return;
  const int b = bp / s.P;
  // (slot: skipped trajectories leave; the others' network rows are packed)
  const int sb = s.slot != nullptr ? s.slot[b] : b;
  if (sb < 0) return;
  const int sbp = sb * s.P + (bp - b * s.P);
  const int D = EXACT ? kJvpMaxD : s.D;
  const int m = s.m, n = D + D * (D + 1) / 2;
  const T* z = s.Z + ((size_t)b * (s.N + 1) + s.t) * n;
  const T* xp = s.Xp + (size_t)bp * D;
  // eps = (X - mean) U^-1: forward substitution against the upper factor
  T U[kJvpMaxD][kJvpMaxD], eps[kJvpMaxD], x[kJvpMaxD];
  {
    int o = D;
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) U[i][j] = j >= i ? z[o++] : 0.f;
  }
  for (int j = 0; j < D; ++j) {
    x[j] = xp[j];
    T v = x[j] - z[j];
    for (int i = 0; i < j; ++i) v -= eps[i] * U[i][j];
    eps[j] = v / U[j][j];
  }
  if (k == 0)
    for (int j = 0; j < D; ++j) s.eps[(size_t)bp * D + j] = eps[j];
  // row k >= 1: direction mean_{k-1} (k - 1 < D) or u_{k-1-D}.  By state
  // dimension c (a static index) to the feature slot the `non` / `ang` lists
  // give it.
  T* f = s.F + ((size_t)sbp * kJvpRows + k) * s.in_dim;
#pragma unroll
  for (int c = 0; c < kJvpMaxD; ++c) {
    if (c >= D) break;
    const T dXc = (k - 1 == c) ? 1.f : 0.f;
    int o = -1, oa = -1;
    for (int i = 0; i < s.n_non; ++i) o = s.non[i] == c ? i : o;
    for (int a = 0; a < s.n_ang; ++a) oa = s.ang[a] == c ? s.n_non + 2 * a : oa;
    if (o >= 0)
      f[o] = k == 0 ? (x[c] - s.X_mean[o]) * s.X_std_inv[o]
                    : dXc * s.X_std_inv[o];
    if (oa >= 0) {
      T sn, cs;
      sincos_(x[c], sn, cs);
      f[oa] = k == 0 ? (sn - s.X_mean[oa]) * s.X_std_inv[oa]
                     : (cs * dXc) * s.X_std_inv[oa];
      f[oa + 1] = k == 0 ? (cs - s.X_mean[oa + 1]) * s.X_std_inv[oa + 1]
                         : (-sn * dXc) * s.X_std_inv[oa + 1];
    }
  }
  int o = s.n_non + 2 * s.n_ang;
  for (int r = 0; r < m; ++r, ++o) {
    // derivatives AT the clamped action (ilqr.py:461-462: the clamp is not
    // differentiated through)
    T u = s.U[((size_t)b * s.N + s.t) * m + r];
    if (s.u_min != nullptr && s.u_max != nullptr)
      u = clamp1(u, s.u_min[r], s.u_max[r]);
    f[o] = k == 0 ? (u - s.X_mean[o]) * s.X_std_inv[o]
                  : ((k - 1 == D + r) ? s.X_std_inv[o] : 0.f);
  }
}

template <typename T, int kJvpRows, int kJvpMaxD, bool EXACT>
__global__ __launch_bounds__(64) void bnn_jvp_moments_kernel(BnnJvpV<T> s) {
  // one wavefront per trajectory: lane = (particle slice, direction k); the
  // NS = 64 / G slices split the particle loop and meet in xor butterflies
  constexpr int NS = 64 / kJvpRows;
  const int lane = threadIdx.x;
  const int k = lane & (kJvpRows - 1);
  const int slice = lane / kJvpRows;
  const int b = blockIdx.x;
  if (b >= s.B) return;
  const int sb = s.slot != nullptr ? s.slot[b] : b;
  if (sb < 0) return;
  const int D = EXACT ? kJvpMaxD : s.D;
  const int P = s.P, m = s.m, n = D + D * (D + 1) / 2;
  const int OUT = s.out_dim;
  const T* Y = s.net_out + (size_t)sb * P * kNetRows * OUT;
  auto across_slices = [&](T v) {
#pragma unroll
    for (int o = kJvpRows; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
  };
  // direction k - 1 of (z | u) -> network row (1 + base direction) and, for a
  // Cholesky direction U_ab, the index a of the particle's eps that scales it
  int yrow = 0, ea = -1;
  {
    const int d = k - 1;
    if (d >= 0 && d < D) yrow = 1 + d;
    else if (d >= n && d < n + m) yrow = 1 + D + (d - n);
    else if (d >= D && d < n) {
      int o = D;
      for (int aa = 0; aa < D; ++aa)
        for (int bb = aa; bb < D; ++bb, ++o)
          if (o == d) { yrow = 1 + bb; ea = aa; }
    }
  }
  const T* Xin = s.Xp + (size_t)b * P * D;
  T sd[kJvpMaxD], mu[kJvpMaxD];
  for (int d = 0; d < D; ++d) { sd[d] = s.dX_std[d]; mu[d] = s.dX_mean[d]; }

  // ---- primal moments (every lane of the group, same arithmetic)
  T M[kJvpMaxD];
  for (int d = 0; d < D; ++d) M[d] = 0.f;
  // use_predicted_std (modules.py:242-260): + exp(log_std + log dX_std) eps
  const bool ups = s.eps_out != nullptr;
  T lsd[kJvpMaxD];
  for (int d = 0; d < D; ++d) lsd[d] = ups ? jvp_log(sd[d]) : 0.f;
  auto primal_out = [&](int p, int d, T& stdeps) {
    const T* y = Y + (size_t)p * kNetRows * OUT;
    T dx = y[d] * sd[d] + mu[d];
    stdeps = 0.f;
    if (ups) {
      stdeps = jvp_exp(y[D + d] + lsd[d]) * s.eps_out[p * D + d];
      dx = dx + stdeps;
    }
    return Xin[p * D + d] + dx;
  };
  for (int p = slice; p < P; p += NS)
    for (int d = 0; d < D; ++d) {
      T unused;
      M[d] += primal_out(p, d, unused);
    }
  for (int d = 0; d < D; ++d) M[d] = across_slices(M[d]) / (T)P;

  // ---- covariance, and this lane's tangent sums
  T C[kJvpMaxD][kJvpMaxD], S[kJvpMaxD][kJvpMaxD], dM[kJvpMaxD];
  for (int i = 0; i < D; ++i) {
    dM[i] = 0.f;
    for (int j = 0; j < D; ++j) { C[i][j] = 0.f; S[i][j] = 0.f; }
  }
  for (int p = slice; p < P; p += NS) {
    T dev[kJvpMaxD], dout[kJvpMaxD];
    // tangent of X for the base direction: e_b (mean_b / U_ab), 0 (u); scale
    const T sc = ea < 0 ? 1.f : s.eps[((size_t)b * P + p) * D + ea];
    for (int d = 0; d < D; ++d) {
      T stdeps;
      const T out = primal_out(p, d, stdeps);
      dev[d] = out - M[d];
      const T dXd = (yrow >= 1 && yrow <= D && yrow - 1 == d) ? 1.f : 0.f;
      const T* yt = Y + ((size_t)p * kNetRows + yrow) * OUT;
      T dnet = yt[d] * sd[d];
      // d exp(log_std + c) eps = std eps d log_std
      if (ups && s.independent_noise == 0) dnet = dnet + stdeps * yt[D + d];
      dout[d] = sc * (dXd + dnet);
      if (k == 0 && s.Xp_next != nullptr)
        s.Xp_next[((size_t)b * P + p) * D + d] = out;
    }
    for (int i = 0; i < D; ++i) {
      dM[i] += dout[i];
      for (int j = 0; j < D; ++j) {
        C[i][j] += dev[i] * dev[j];
        S[i][j] += dout[i] * dev[j];
      }
    }
  }
  for (int i = 0; i < D; ++i) {
    dM[i] = across_slices(dM[i]) / (T)P;
    for (int j = 0; j < D; ++j) {
      C[i][j] = across_slices(C[i][j]) / (T)(P - 1);
      S[i][j] = across_slices(S[i][j]);
    }
  }
  if (slice != 0) return;  // (every slice holds the totals; one writes)

  // ---- U' = chol(C + jitter I), upper (encoding.py:536-564)
  T Uc[kJvpMaxD][kJvpMaxD];
  bool ok = false;
  {
    double jit = 1e-12;
    while (!ok && jit <= 10.0) {
      ok = true;
      // (no early exit: a failed pivot ends the reference's attempt; what is
      // computed after it here is garbage that the next attempt overwrites -
      // the loops stay fully unrolled and Uc in registers)
#pragma unroll
      for (int i = 0; i < kJvpMaxD; ++i) {
        if (i >= D) break;
#pragma unroll
        for (int j = i; j < kJvpMaxD; ++j) {
          if (j >= D) break;
          T v = C[i][j] + (i == j ? (T)jit : 0.f);
#pragma unroll
          for (int q = 0; q < i; ++q) v -= Uc[q][i] * Uc[q][j];
          if (i == j) {
            if (!(v > 0.f)) ok = false;
            Uc[i][i] = sqrt_(v);
          } else {
            Uc[i][j] = v / Uc[i][i];
          }
        }
      }
      jit *= 10.0;
    }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < i; ++j) Uc[i][j] = 0.f;
  }

  T dU[kJvpMaxD][kJvpMaxD];
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) dU[i][j] = 0.f;
  if (ok) {
    // dC = (S + S^T) / (P - 1);  W = U'^-T dC U'^-1;  dU' = Phi(W) U'
    T dC[kJvpMaxD][kJvpMaxD], T1[kJvpMaxD][kJvpMaxD], W[kJvpMaxD][kJvpMaxD];
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) dC[i][j] = (S[i][j] + S[j][i]) / (T)(P - 1);
    // T1 = U'^-T dC: solve U'^T T1 = dC (U'^T lower): forward substitution
    for (int c = 0; c < D; ++c)
      for (int i = 0; i < D; ++i) {
        T v = dC[i][c];
        for (int q = 0; q < i; ++q) v -= Uc[q][i] * T1[q][c];
        T1[i][c] = v / Uc[i][i];
      }
    // W = T1 U'^-1: solve W U' = T1 row by row, forward in the column index
    for (int r = 0; r < D; ++r)
      for (int j = 0; j < D; ++j) {
        T v = T1[r][j];
        for (int q = 0; q < j; ++q) v -= W[r][q] * Uc[q][j];
        W[r][j] = v / Uc[j][j];
      }
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) {
        T v = 0.f;
        for (int q = i; q <= j; ++q)
          v += (q == i ? 0.5f * W[i][i] : W[i][q]) * Uc[q][j];
        dU[i][j] = v;
      }
  } else {
    // encode()'s fall-back: the diagonal of standard deviations
    // (modules.py:380-386 -> encoding.py:99-141)
    for (int i = 0; i < D; ++i) {
      const T sdev = sqrt_(C[i][i]);
      Uc[i][i] = sdev;
      dU[i][i] = (S[i][i] + S[i][i]) / (T)(P - 1) / (2.f * sdev);
    }
  }

  // ---- outputs: the next encoded state (lane 0) and column k - 1 of (F_z|F_u)
  if (k == 0) {
    if (s.Z_next != nullptr) {
      T* zn = s.Z_next + (size_t)b * n;
      for (int d = 0; d < D; ++d) zn[d] = M[d];
      int o = D;
      for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) zn[o++] = Uc[i][j];
    }
    return;
  }
  const int col = k - 1;
  if (col >= n + m) return;
  T* dst;
  int ld;
  if (col < n) {
    dst = s.F_z + ((size_t)b * s.N + s.t) * n * n + col;
    ld = n;
  } else {
    dst = s.F_u + ((size_t)b * s.N + s.t) * n * m + (col - n);
    ld = m;
  }
  for (int d = 0; d < D; ++d) dst[(size_t)d * ld] = dM[d];
  int o = D;
  for (int i = 0; i < D; ++i)
    for (int j = i; j < D; ++j, ++o) dst[(size_t)o * ld] = dU[i][j];
}

}  // namespace pddp

extern "C" {

/* rows per (state, particle) group for a problem: 16 or 32 (0: unsupported) */
int pddp_bnn_jvp_group(int D, int m) {
  const int dirs = D + D * (D + 1) / 2 + m;
  if (D + m > 7) return 0;  // network rows per particle: input + D + m <= 8
  if (D <= 4 && dirs <= 15) return 16;
  if (D <= 6 && dirs <= 31) return 32;
  return 0;
}

}  // extern "C"

namespace pddp {

template <typename T>
static int bnn_jvp_check(const BnnJvpV<T>* s) {
  if (s == nullptr) return PDDP_E_BADARG;
  if (s->B <= 0 || s->P <= 1 || s->N <= 0 || s->t < 0 || s->t >= s->N ||
      !s->Z || !s->U || !s->X_mean || !s->X_std_inv || !s->dX_mean ||
      !s->dX_std || !s->Xp || !s->eps || !s->F)
    return PDDP_E_BADARG;
  const int n = s->D + s->D * (s->D + 1) / 2;
  if (s->D < 1 || s->D > 6 || s->m < 1 || n + s->m > 31 ||
      s->D + s->m > kNetRows - 1 || s->n_ang < 0 || s->n_ang > 2 ||
      s->n_non < 0 || s->n_non + s->n_ang != s->D ||
      s->in_dim != s->n_non + 2 * s->n_ang + s->m || s->out_dim < s->D ||
      (s->eps_out != nullptr && s->out_dim < 2 * s->D))
    return PDDP_E_UNSUPPORTED;
  return 0;
}

template <typename T>
static int bnn_jvp_features(const BnnJvpV<T>* s, void* stream) {
  if (int rc = bnn_jvp_check(s)) return rc;
  const dim3 grid((s->B * s->P + 7) / 8), block(64);
  hipStream_t st = (hipStream_t)stream;
  switch (s->D) {
    case 2: PDDP_LAUNCH((bnn_jvp_features_kernel<T, 2, true>), grid, block, 0, st, *s); break;
    case 4: PDDP_LAUNCH((bnn_jvp_features_kernel<T, 4, true>), grid, block, 0, st, *s); break;
    case 6: PDDP_LAUNCH((bnn_jvp_features_kernel<T, 6, true>), grid, block, 0, st, *s); break;
    default:
      if (s->D <= 4)
        PDDP_LAUNCH((bnn_jvp_features_kernel<T, 4, false>), grid, block, 0, st, *s);
      else
        PDDP_LAUNCH((bnn_jvp_features_kernel<T, 6, false>), grid, block, 0, st, *s);
  }
  return launch_status();
}

template <typename T>
static int bnn_jvp_moments(const BnnJvpV<T>* s, void* stream) {
  if (int rc = bnn_jvp_check(s)) return rc;
  if (!s->net_out || !s->F_z || !s->F_u) return PDDP_E_BADARG;
  const dim3 grid(s->B), block(64);
  hipStream_t st = (hipStream_t)stream;
  const bool g16 = pddp_bnn_jvp_group(s->D, s->m) == 16;
  if (g16 && s->D == 2)
    PDDP_LAUNCH((bnn_jvp_moments_kernel<T, 16, 2, true>), grid, block, 0, st, *s);
  else if (g16 && s->D == 4)
    PDDP_LAUNCH((bnn_jvp_moments_kernel<T, 16, 4, true>), grid, block, 0, st, *s);
  else if (g16)
    PDDP_LAUNCH((bnn_jvp_moments_kernel<T, 16, 4, false>), grid, block, 0, st, *s);
  else if (s->D == 6)
    PDDP_LAUNCH((bnn_jvp_moments_kernel<T, 32, 6, true>), grid, block, 0, st, *s);
  else
    PDDP_LAUNCH((bnn_jvp_moments_kernel<T, 32, 6, false>), grid, block, 0, st, *s);
  return launch_status();
}

}  // namespace pddp

extern "C" {

int pddp_bnn_jvp_features_f32(const pddp_bnn_jvp* s, void* stream) {
  return pddp::bnn_jvp_features(reinterpret_cast<const pddp::BnnJvpV<float>*>(s), stream);
}
int pddp_bnn_jvp_moments_f32(const pddp_bnn_jvp* s, void* stream) {
  return pddp::bnn_jvp_moments(reinterpret_cast<const pddp::BnnJvpV<float>*>(s), stream);
}
int pddp_bnn_jvp_features_f64(const pddp_bnn_jvp_f64* s, void* stream) {
  return pddp::bnn_jvp_features(reinterpret_cast<const pddp::BnnJvpV<double>*>(s), stream);
}
int pddp_bnn_jvp_moments_f64(const pddp_bnn_jvp_f64* s, void* stream) {
  return pddp::bnn_jvp_moments(reinterpret_cast<const pddp::BnnJvpV<double>*>(s), stream);
}

}  // extern "C"
