// bnn_rollout.hip - one time step of the moment-matched line-search rollout
// under a BNN dynamics model, everything except the network itself
// (csrc/bnn_mlp.hip): what pddp/controllers/ilqr.py:677-723 (_control_law),
// :764-791 (_trajectory_cost), pddp/models/bnn/modules.py:287-386
// (BNNDynamicsModel.forward), pddp/utils/encoding.py:99-141 (encode, DEFAULT =
// mean | upper Cholesky) and pddp/utils/angular.py:47-84,161-248 (moment-matched
// angle augmentation for the cost) do per candidate and time step - in the
// framework that is ~150 small launches per step; here it is one.
//
// Sixteen lanes per candidate (b, alpha), four candidates per wavefront; lanes
// own particles (up to eight each).  Step t:
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
// escalation, augmentation moments, quadratic cost) runs on the first lane of
// each group out of LDS: ~1 kFLOP per step against the network's 8.6 MFLOP, but
// a serial instruction stream that a wavefront issues whether one lane or four
// are active - hence four candidates per wavefront (round 1 had one: 64-lane
// reductions, the same serial stream for a quarter of the work).
#include "pddp_common.hpp"
#include "models.hpp"  // sincos_, sincos_core: branch-free, ~1 ulp

namespace pddp {

constexpr int kBnnMaxD = 8, kBnnMaxAng = 2, kBnnMaxM = 2;
constexpr int kBnnMaxNa = kBnnMaxD + kBnnMaxAng;            // augmented size
constexpr int kBnnMaxN = kBnnMaxD + kBnnMaxD * (kBnnMaxD + 1) / 2;  // 44

constexpr int kBnnGroup = 16;                  // lanes per candidate
constexpr int kBnnPerWave = 64 / kBnnGroup;    // candidates per wavefront
constexpr int kBnnMaxPPL = 128 / kBnnGroup;    // particles per lane (P <= 128)

PDDP_DEV float bnn_exp(float x) { return expf(x); }
PDDP_DEV double bnn_exp(double x) { return exp(x); }
PDDP_DEV float bnn_log(float x) { return logf(x); }
PDDP_DEV double bnn_log(double x) { return log(x); }
PDDP_DEV void bnn_sincos_core(float x, float& s, float& c) { sincos_core(x, s, c); }
PDDP_DEV void bnn_sincos_core(double x, double& s, double& c) { sincos(x, &s, &c); }

// pddp_bnn_step / pddp_bnn_step_f64 (include/pddp_hip.h) with the scalar type
// as a parameter: the two structs differ in the pointee type only
template <typename T>
struct BnnStepV {
  int32_t B, A, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8];
  int32_t in_dim, out_dim;
  const T* Z;
  const T* U;
  const T* gains;
  const T* alphas;
  const T* u_min;
  const T* u_max;
  const uint8_t* active;
  const int32_t* bwd_status;
  const T* Q;
  const T* Q_term;
  const T* R;
  const T* x_goal;
  const T* u_goal;
  const T* X_mean;
  const T* X_std_inv;
  const T* dX_mean;
  const T* dX_std;
  const T* net_out;
  T* Xp;
  T* F;
  T* Zc;
  T* Uc;
  T* J;
  T* Jc;
  const T* eps_out;
  const int32_t* slot;
};
static_assert(sizeof(BnnStepV<float>) == sizeof(pddp_bnn_step) &&
              sizeof(BnnStepV<double>) == sizeof(pddp_bnn_step_f64), "");

template <typename T>
PDDP_DEV T group_sum(T v) {  // over the 16 lanes of a candidate
#pragma unroll
  for (int o = kBnnGroup / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// upper Cholesky of the d x d matrix C + jitter I (row-major, ld = kBnnMaxNa),
// false when a pivot is not positive (encoding.py:536-564 tries jitter = 1e-12,
// 1e-11, ... <= 10)
template <typename T>
PDDP_DEV bool chol_upper(const T* C, int d, T jitter, T* U) {
  for (int i = 0; i < d; ++i) {
    for (int j = i; j < d; ++j) {
      T s = C[i * kBnnMaxNa + j] + (i == j ? jitter : 0.f);
      for (int k = 0; k < i; ++k) s -= U[k * kBnnMaxNa + i] * U[k * kBnnMaxNa + j];
      if (i == j) {
        if (!(s > 0.f)) return false;
        U[i * kBnnMaxNa + i] = sqrt_(s);
      } else {
        U[i * kBnnMaxNa + j] = s / U[i * kBnnMaxNa + i];
      }
    }
    for (int j = 0; j < i; ++j) U[i * kBnnMaxNa + j] = 0.f;
  }
  return true;
}
// with the escalation; returns the jitter used, -1 when none up to 10 works
template <typename T>
PDDP_DEV T chol_upper_jittered(const T* C, int d, T* U) {
  double jit = 1e-12;  // python T in the reference
  while (true) {
    if (chol_upper(C, d, (T)jit, U)) return (T)jit;
    jit *= 10.0;
    if (jit > 10.0) return -1.f;
  }
}

// DT: the state dimension as a compile-time constant (0: taken from the
// argument).  With it the particle registers x[.][d] are indexed statically;
// under a runtime D they lived in scratch memory (272 B per lane, 136 scratch
// instructions) and the kernel waited for them.
template <typename T, int DT>
__global__ __launch_bounds__(64) void bnn_moment_step_kernel(BnnStepV<T> s) {
  constexpr int DX = DT > 0 ? DT : kBnnMaxD;
  // per candidate: Ms, Cs, Us, zs, us, Ma, Ca, Ua; the stride is odd in banks
  constexpr int kM2 = kBnnMaxNa * kBnnMaxNa;
  constexpr int kStride = kBnnMaxD + 4 * kM2 + kBnnMaxN + kBnnMaxM + kBnnMaxNa + 1;
  static_assert(kStride % 2 == 1, "groups must start in different banks");
  __shared__ T sm[kBnnPerWave][kStride];

  const int lane = threadIdx.x & (kBnnGroup - 1);  // lane of the group
  const int grp = threadIdx.x / kBnnGroup;
  const int c = blockIdx.x * kBnnPerWave + grp;    // candidate = b * A + ai
  if (c >= s.B * s.A) return;
  T* Ms = sm[grp];
  T* Cs = Ms + kBnnMaxD;
  T* Us = Cs + kM2;
  T* zs = Us + kM2;
  T* us = zs + kBnnMaxN;
  T* Ma = us + kBnnMaxM;
  T* Ca = Ma + kBnnMaxNa;
  T* Ua = Ca + kM2;
  const int b = c / s.A, ai = c - b * s.A;
  // (a group that leaves early takes no part in the barriers below: one
  // wavefront per workgroup, a barrier only orders its LDS traffic)
  if (s.active != nullptr && s.active[b] == 0) return;
  if (s.bwd_status != nullptr && s.bwd_status[b] != 0) return;
  // rows of the network's input / output: the candidate's own, or - with
  // `slot` - those of its trajectory's rank among the live ones, so that the
  // network runs on the live candidates' rows only
  const int cc = s.slot != nullptr ? s.slot[b] * s.A + ai : c;
  const int D = DT > 0 ? DT : s.D;
  const int P = s.P, m = s.m, N = s.N, t = s.t;
  const int n = D + D * (D + 1) / 2;
  const int na = s.n_non + 2 * s.n_ang;
  const bool terminal = (t == N);

  // ---- particles of this step (up to eight per lane: P <= 128)
  T x[kBnnMaxPPL][DX];
  bool has[kBnnMaxPPL];
#pragma unroll
  for (int q = 0; q < kBnnMaxPPL; ++q) {
    const int p = lane + kBnnGroup * q;
    has[q] = p < P;
    const size_t row = (size_t)c * P + (has[q] ? p : 0);
    const size_t nrow = (size_t)cc * P + (has[q] ? p : 0);  // (network rows)
    // rows of D (and out_dim = D or 2 D) floats: 8-byte vector accesses when D
    // is a compile-time even number (rows are then 8-byte aligned)
    constexpr bool kVec = DT > 0 && DT % 2 == 0;
    T xin[DX], net[DX], nls[DX];
    if constexpr (kVec) {
      typedef T tx2 __attribute__((ext_vector_type(2)));
      const tx2* xr = reinterpret_cast<const tx2*>(s.Xp + row * D);
      const tx2* nr = reinterpret_cast<const tx2*>(
          s.net_out + nrow * s.out_dim);  // (dereferenced for t > 0 only)
#pragma unroll
      for (int d2 = 0; d2 < DX / 2; ++d2) {
        const tx2 a = xr[d2];
        xin[2 * d2] = a[0];
        xin[2 * d2 + 1] = a[1];
        if (t > 0) {
          const tx2 b = nr[d2];
          net[2 * d2] = b[0];
          net[2 * d2 + 1] = b[1];
          if (s.eps_out != nullptr) {
            const tx2 e = nr[DX / 2 + d2];
            nls[2 * d2] = e[0];
            nls[2 * d2 + 1] = e[1];
          }
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < DX; ++d) {
        if (d >= D) break;
        xin[d] = s.Xp[row * D + d];
        if (t > 0) {
          net[d] = s.net_out[nrow * s.out_dim + d];
          if (s.eps_out != nullptr) nls[d] = s.net_out[nrow * s.out_dim + D + d];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      if (d >= D) break;
      T v = xin[d];
      if (t > 0) {  // X + dx, dx = out[:D] * dX_std + dX_mean   (modules.py:262)
        T dx = net[d] * s.dX_std[d] + s.dX_mean[d];
        if (s.eps_out != nullptr)  // + exp(log_std + log dX_std) eps (:242-260)
          dx = dx + bnn_exp(nls[d] + bnn_log(s.dX_std[d])) *
                        s.eps_out[(has[q] ? p : 0) * D + d];
        v = v + dx;
      }
      x[q][d] = v;
    }
    if (t > 0 && has[q]) {
      if constexpr (kVec) {
        typedef T tx2 __attribute__((ext_vector_type(2)));
        tx2* xw = reinterpret_cast<tx2*>(s.Xp + row * D);
#pragma unroll
        for (int d2 = 0; d2 < DX / 2; ++d2)
          xw[d2] = tx2{x[q][2 * d2], x[q][2 * d2 + 1]};
      } else {
#pragma unroll
        for (int d = 0; d < DX; ++d) {
          if (d >= D) break;
          s.Xp[row * D + d] = x[q][d];
        }
      }
    }
  }

  // ---- z_t
  if (t == 0) {
    for (int k = lane; k < n; k += kBnnGroup)
      zs[k] = s.Z[((size_t)b * (N + 1)) * n + k];
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
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      if (d >= D) break;
      T v = 0.f;
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q) v += has[q] ? x[q][d] : 0.f;
      v = group_sum(v) / (T)P;
      if (lane == 0) Ms[d] = v;
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q)
        x[q][d] -= v;  // deviations from here on (features add the mean back)
    }
#pragma unroll
    for (int i = 0; i < DX; ++i)
#pragma unroll
      for (int j = i; j < DX; ++j) {
        if (j >= D) break;
        T v = 0.f;
#pragma unroll
        for (int q = 0; q < kBnnMaxPPL; ++q)
          v += has[q] ? x[q][i] * x[q][j] : 0.f;
        v = group_sum(v) / (T)(P - 1);
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
                i == j ? sqrt_(Cs[i * kBnnMaxNa + i]) : 0.f;
      }
      for (int d = 0; d < D; ++d) zs[d] = Ms[d];
      int o = D;
      for (int i = 0; i < D; ++i)
        for (int j = i; j < D; ++j) zs[o++] = Us[i * kBnnMaxNa + j];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DX; ++d) {  // particles again (Ms is visible now)
      if (d >= D) break;
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q) x[q][d] += Ms[d];
    }
  }
  __syncthreads();
  for (int k = lane; k < n; k += kBnnGroup)
    s.Zc[(((size_t)b * (N + 1) + t) * s.A + ai) * n + k] = zs[k];

  // ---- control law, cost (lane 0)
  if (lane == 0) {
    if (!terminal) {
      const int GS = m + m * n;
      const T* g = s.gains + ((size_t)b * N + t) * GS;
      const T* zn = s.Z + ((size_t)b * (N + 1) + t) * n;
      for (int r = 0; r < m; ++r) {
        T du = s.alphas[ai] * g[r];                     // ilqr.py:708
        T acc = 0.f;
        for (int k = 0; k < n; ++k) acc += (zs[k] - zn[k]) * g[m + r * n + k];
        du = du + acc;                                      // ilqr.py:710
        T v = s.U[((size_t)b * N + t) * m + r] + du;
        if (s.u_min != nullptr && s.u_max != nullptr)
          v = clamp1(v, s.u_min[r], s.u_max[r]);
        us[r] = v;
        s.Uc[(((size_t)b * N + t) * s.A + ai) * m + r] = v;
      }
    }
    // cost on the angle-augmented moments (examples/*/cost.py, quadratic.py:
    // 60-99, angular.py:161-248).  Covariance as the cost sees it: C = U^T U.
    // (U is upper triangular: rows k <= min(i, j) only - the skipped terms
    // are exact zeros; the matrix is symmetric)
    for (int i = 0; i < D; ++i)
      for (int j = i; j < D; ++j) {
        T v = 0.f;
        for (int k = 0; k <= i; ++k) v += Us[k * kBnnMaxNa + i] * Us[k * kBnnMaxNa + j];
        Cs[i * kBnnMaxNa + j] = v;
        Cs[j * kBnnMaxNa + i] = v;
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
      const T m1 = Ms[i1], v1 = Cs[i1 * kBnnMaxNa + i1];
      const T damp = bnn_exp(-0.5f * v1);
      T sin1, cos1;
      sincos_(m1, sin1, cos1);
      const T Es = damp * sin1, Ec = damp * cos1;
      Ma[nn + 2 * a1] = Es;
      Ma[nn + 2 * a1 + 1] = Ec;
      for (int a2 = 0; a2 < nang; ++a2) {
        const int i2 = s.ang[a2];
        const T m2 = Ms[i2], v2 = Cs[i2 * kBnnMaxNa + i2];
        const T cij = Cs[i1 * kBnnMaxNa + i2];
        const T lq = -0.5f * (v1 + v2), q = bnn_exp(lq);
        const T ep = bnn_exp(lq + cij) - q, em = bnn_exp(lq - cij) - q;
        T sd, cd, ss, cs;
        sincos_(m1 - m2, sd, cd);
        sincos_(m1 + m2, ss, cs);
        const int r = nn + 2 * a1, cc = nn + 2 * a2;
        Ca[r * kBnnMaxNa + cc] = 0.5f * (ep * cd - em * cs);            // sin, sin
        Ca[(r + 1) * kBnnMaxNa + cc + 1] = 0.5f * (ep * cd + em * cs);  // cos, cos
        Ca[r * kBnnMaxNa + cc + 1] = 0.5f * (ep * sd + em * ss);        // sin, cos
        // (cos_i, sin_j) = (sin_j, cos_i): written when the roles swap
        Ca[(cc + 1) * kBnnMaxNa + r] = Ca[r * kBnnMaxNa + cc + 1];
      }
      for (int i = 0; i < nn; ++i) {
        const T col = Cs[s.non[i] * kBnnMaxNa + i1];
        const int r = nn + 2 * a1;
        Ca[i * kBnnMaxNa + r] = col * Ec;        // Cov(x, sin)
        Ca[i * kBnnMaxNa + r + 1] = -col * Es;   // Cov(x, cos)
        Ca[r * kBnnMaxNa + i] = col * Ec;
        Ca[(r + 1) * kBnnMaxNa + i] = -col * Es;
      }
    }
    // the cost re-encodes the augmented covariance (a jittered Cholesky) and
    // decodes it again: C'' = Ca + jitter I for the first jitter that works
    T jit = chol_upper_jittered(Ca, na, Ua);
    const T* Q = terminal ? s.Q_term : s.Q;
    T cost = 0.f;
    for (int i = 0; i < na; ++i) {
      T row = 0.f;
      for (int j = 0; j < na; ++j) row += (Ma[j] - s.x_goal[j]) * Q[j * na + i];
      cost += row * (Ma[i] - s.x_goal[i]);
    }
    if (!terminal) {
      for (int i = 0; i < m; ++i) {
        T row = 0.f;
        for (int j = 0; j < m; ++j) row += (us[j] - s.u_goal[j]) * s.R[j * m + i];
        cost += row * (us[i] - s.u_goal[i]);
      }
    }
    T tr = 0.f;
    if (jit >= 0.f) {
      // tr(Q Ua^T Ua) with Ua^T Ua = Ca + jitter I: the factor only decides
      // which jitter the cost sees (as csrc/qr_cost_derivs.hip takes it);
      // rebuilding the product from Ua was na^3 serial multiply-adds, the
      // largest single piece of this kernel
      for (int i = 0; i < na; ++i) {
        for (int j = 0; j < na; ++j) tr += Ca[i * kBnnMaxNa + j] * Q[j * na + i];
        tr += jit * Q[i * na + i];
      }
    } else {  // diagonal fallback of encode(): variances only
      for (int i = 0; i < na; ++i) tr += Ca[i * kBnnMaxNa + i] * Q[i * na + i];
    }
    cost += tr;
    const T J = (t == 0 ? 0.f : s.J[c]) + cost;
    s.J[c] = J;
    if (terminal) s.Jc[c] = J;
  }
  if (terminal) return;
  __syncthreads();

  // ---- the network's input rows for this candidate's particles: by state
  // dimension (a static register index) to the feature slot the `non` / `ang`
  // lists give it - the other way round is a dynamic index into x, i.e.
  // scratch memory
  T* frow = s.F + ((size_t)cc * P + lane) * s.in_dim;
  const size_t qstep = (size_t)kBnnGroup * s.in_dim;
#pragma unroll
  for (int d = 0; d < DX; ++d) {
    if (d >= D) break;
    int o = -1, oa = -1;
    for (int i = 0; i < s.n_non; ++i) o = s.non[i] == d ? i : o;
    for (int a1 = 0; a1 < s.n_ang; ++a1)
      oa = s.ang[a1] == d ? s.n_non + 2 * a1 : oa;
    if (o >= 0) {
      const T mu = s.X_mean[o], si = s.X_std_inv[o];
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q)
        if (has[q]) frow[q * qstep + o] = (x[q][d] - mu) * si;
    }
    if (oa >= 0) {
      const T mu0 = s.X_mean[oa], si0 = s.X_std_inv[oa];
      const T mu1 = s.X_mean[oa + 1], si1 = s.X_std_inv[oa + 1];
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q) {
        T sn, cs;  // (a particle beyond 2^30 rad: see sincos_core)
        bnn_sincos_core(x[q][d], sn, cs);
        if (has[q]) {
          frow[q * qstep + oa] = (sn - mu0) * si0;
          frow[q * qstep + oa + 1] = (cs - mu1) * si1;
        }
      }
    }
  }
  {
    const int o0 = s.n_non + 2 * s.n_ang;
    for (int r = 0; r < m; ++r) {
      const T v = (us[r] - s.X_mean[o0 + r]) * s.X_std_inv[o0 + r];
#pragma unroll
      for (int q = 0; q < kBnnMaxPPL; ++q)
        if (has[q]) frow[q * qstep + o0 + r] = v;
    }
  }
}

}  // namespace pddp

namespace pddp {

template <typename T>
static int bnn_moment_step(const BnnStepV<T>* s, void* stream) {
  if (s == nullptr) return PDDP_E_BADARG;
  if (s->B <= 0 || s->A <= 0 || s->P <= 1 || s->N <= 0 || s->t < 0 ||
      s->t > s->N || !s->Z || !s->U || !s->gains || !s->alphas || !s->Q ||
      !s->Q_term || !s->R || !s->x_goal || !s->u_goal || !s->X_mean ||
      !s->X_std_inv || !s->dX_mean || !s->dX_std || !s->Xp || !s->F || !s->Zc ||
      !s->Uc || !s->J || !s->Jc || (s->t > 0 && !s->net_out))
    return PDDP_E_BADARG;
  if (s->D < 1 || s->D > kBnnMaxD || s->m < 1 || s->m > kBnnMaxM ||
      s->P > 128 || s->n_ang < 0 || s->n_ang > kBnnMaxAng ||
      s->n_non < 0 || s->n_non + s->n_ang != s->D ||
      s->in_dim != s->n_non + 2 * s->n_ang + s->m || s->out_dim < s->D ||
      (s->eps_out != nullptr && s->out_dim < 2 * s->D))
    return PDDP_E_UNSUPPORTED;
  const int groups = s->B * s->A;
  const dim3 grid((groups + kBnnPerWave - 1) / kBnnPerWave);
  hipStream_t st = (hipStream_t)stream;
  switch (s->D) {
    case 2: PDDP_LAUNCH((bnn_moment_step_kernel<T, 2>), grid, dim3(64), 0, st, *s); break;
    case 4: PDDP_LAUNCH((bnn_moment_step_kernel<T, 4>), grid, dim3(64), 0, st, *s); break;
    case 6: PDDP_LAUNCH((bnn_moment_step_kernel<T, 6>), grid, dim3(64), 0, st, *s); break;
    default: PDDP_LAUNCH((bnn_moment_step_kernel<T, 0>), grid, dim3(64), 0, st, *s); break;
  }
  return launch_status();
}

}  // namespace pddp

extern "C" int pddp_bnn_moment_step_f32(const pddp_bnn_step* s, void* stream) {
  return pddp::bnn_moment_step(
      reinterpret_cast<const pddp::BnnStepV<float>*>(s), stream);
}
extern "C" int pddp_bnn_moment_step_f64(const pddp_bnn_step_f64* s, void* stream) {
  return pddp::bnn_moment_step(
      reinterpret_cast<const pddp::BnnStepV<double>*>(s), stream);
}
