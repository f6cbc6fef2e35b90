// gp_step.hip - one moment-matched step of the GP dynamics plugin
// (pddp_amd/models/gp.py: squared-exponential ARD GPs per state increment,
// exact first and second moment of the posterior at a Gaussian input;
// Deisenroth & Rasmussen 2011 eqs. 14-23) and, optionally, its Jacobian with
// respect to the encoded state and the action - the records of the GP
// workload's derivative rollout (BASELINE configs[3]).  PARITY UNPINNED: the
// reference has no GP (pddp/models/__init__.py:17-20); the checker is the
// torch module itself (autograd for the Jacobian) and oracle/gp_port.py.
//
// One workgroup (four wavefronts) per row (= one trajectory at one time step):
//   A0  wave 0: decode the encoded state, moment-matched trigonometric
//       features, m [d], S [d d], cov[x, features] - written to LDS element by
//       element; with JAC lane k carries the tangent of input k (dual numbers)
//   A1  lanes 0 .. E + E(E+1)/2: G_s = (S + diag delta_s)^-1 and log det, one
//       9 x 9 Cholesky per lane, in registers.  Every inverse the moments need
//       is of this form: (S + L_a)^-1, and R^-1 S = lam - lam G lam with
//       lam = (L_a^-1 + L_b^-1)^-1, G = (S + lam)^-1 (Woodbury), det R = det(S +
//       lam) / prod lam.  Meanwhile waves 1-3: nu_i = x_i - m, log k_a(x_i, m)
//   A2  per output a (a wavefront each, lane = training point): q_a, mu_a,
//       W_a = (S + L_a)^-1 sum beta q nu
//   B   per pair a <= b (a wavefront each, lane = i, loop over j):
//       F_ab = sum_ij (beta_ai beta_bj - [a=b] Kinv_a,ij) exp(n2_ij) - M^2
//       exponentials per pair, the bulk of the work.  With JAC the same loop
//       accumulates Y1 = sum c_ij y_ij and Y2 = sum c_ij y_ij y_ij^T (y = L_a^-1
//       nu_i + L_b^-1 nu_j) through row sums, column sums and v_i = sum_j c_ij
//       z_bj: ALL derivatives of F_ab with respect to m and S are contractions
//       of Y1, Y2 (d n2 / dT = y y^T / 2, d n2 / dm = y - lam^-1 (T y)), so the
//       Jacobian costs no second M^2 loop
//   C   (JAC) gradients of mu_a, Sigma_ab with respect to (m, S) in LDS
//       (reverse mode over the small algebra: dG = -G dS G); tangents of W per
//       (input k, output a) by a loop over the training points
//   A3  lane k: next mean, covariance, encoding (upper Cholesky) in dual
//       numbers -> column k of the Jacobian
// Deterministic: fixed-order wave reductions, no atomics.
#include "pddp_common.hpp"
#include "models.hpp"  // sincos_: one range reduction for both, ~1 ulp

namespace pddp {
namespace gp {

// Debug build (-DPDDP_GP_MARKS): wavefront 0 of every row leaves s_memtime at
// the phase boundaries in a global buffer (pddp_debug_gp_marks)
#ifdef PDDP_GP_MARKS
__device__ long long g_gp_marks[8 * 8];
#define PDDP_GP_MARK(k) \
  do { if (tid == 0 && blockIdx.x < 8) g_gp_marks[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PDDP_GP_MARK(k) do { } while (0)
#endif

constexpr int kThreads = 256;
constexpr int kMaxAng = 4, kMaxNon = 8;

template <typename T>
struct Args {
  int R, M, m_act, n_ang, n_non, encoding, n;
  // rows of group r / rows_per_mask are skipped where row_mask[group] == 0
  // (nullable: every row)
  const uint8_t* row_mask;
  int rows_per_mask;
  int ang[kMaxAng], non[kMaxNon];
  const T* Xt;    // [M d]
  const T* XtP;   // [MQ/2][PS]: (x_2j[p], x_2j+1[p]) pairs, MQ = M rounded up to 4, PS = 2 d rounded up to 4
  const T* beta;  // [E M]
  const T* betaP; // [E][MQ/2][2]
  const T* Kinv;  // [E M M]
  const T* iL;    // [E d]  1 / lengthscale^2
  const T* sf2;   // [E]
  const T* sn2;   // [E]
  const T* z;     // [R n]
  const T* u;     // [R m]
  T* z_next;      // [R n]
  T* Fz;          // [R n n] or null
  T* Fu;          // [R n m] or null
};

// The line search as a device rollout (pddp_gp_rollout_*, ROLL kernels): row
// r = (trajectory b, step size a) of time step t takes its state from the
// candidates' own array, forms its action by the control law (ilqr.py:708-716:
// u = clamp(U + alpha k + K (z - Z))), adds the stage cost of (z, u) - the QR
// cost on the angle-augmented Gaussian state (costs/quadratic.py:60-99), whose
// moments ARE the kernel's feature moments - to Jc, steps, and writes the next
// state into the candidates' array: N launches with nothing between them, and
// one more (terminal) for the terminal cost.
constexpr int kMaxAct = 4;
template <typename T>
struct Roll {
  int B, N, A, t, terminal, na;
  const T* Z;       // [B][N+1][n] nominal
  const T* U;       // [B][N][m]
  const T* gains;   // [B][N][m + m n]: k | K
  const T* alphas;  // [A]
  const T* u_min;   // [m] nullable (with u_max)
  const T* u_max;
  const uint8_t* active;   // [B] nullable
  const int32_t* status;   // [B] nullable: rows of a failed sweep are skipped
  T* Zc;            // [B][N+1][A][n]
  T* Uc;            // [B][N][A][m]
  T* Jc;            // [B][A]
  const T* Q;       // [na][na]
  const T* Qt;      // [na][na] terminal
  const T* Rm;      // [m][m]
  const T* xg;      // [na]
  const T* ug;      // [m]
};

// ---- dual numbers (one tangent) ----------------------------------------------
template <typename T>
struct Dual {
  T p, t;
};
template <typename T> PDDP_DEV Dual<T> operator+(Dual<T> a, Dual<T> b) { return {a.p + b.p, a.t + b.t}; }
template <typename T> PDDP_DEV Dual<T> operator-(Dual<T> a, Dual<T> b) { return {a.p - b.p, a.t - b.t}; }
template <typename T> PDDP_DEV Dual<T> operator-(Dual<T> a) { return {-a.p, -a.t}; }
template <typename T> PDDP_DEV Dual<T> operator*(Dual<T> a, Dual<T> b) { return {a.p * b.p, a.p * b.t + a.t * b.p}; }
template <typename T> PDDP_DEV Dual<T> operator*(T a, Dual<T> b) { return {a * b.p, a * b.t}; }
template <typename T> PDDP_DEV Dual<T> operator/(Dual<T> a, Dual<T> b) {
  const T r = (T)1 / b.p, q = a.p * r;
  return {q, (a.t - q * b.t) * r};
}
using pddp::cos_;
using pddp::sin_;
PDDP_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PDDP_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
using pddp::sqrt_;
PDDP_DEV float exp_(float x) { return expf(x); }
PDDP_DEV double exp_(double x) { return exp(x); }
PDDP_DEV float log_(float x) { return logf(x); }
PDDP_DEV double log_(double x) { return log(x); }
template <typename T> PDDP_DEV Dual<T> exp_(Dual<T> a) { const T e = exp_(a.p); return {e, e * a.t}; }
template <typename T> PDDP_DEV Dual<T> sin_(Dual<T> a) {
  T sv, cv;
  pddp::sincos_(a.p, sv, cv);
  return {sv, cv * a.t};
}
template <typename T> PDDP_DEV Dual<T> cos_(Dual<T> a) {
  T sv, cv;
  pddp::sincos_(a.p, sv, cv);
  return {cv, -sv * a.t};
}
// sine and cosine of one argument together
PDDP_DEV void sincos2(float x, float& s, float& c) { pddp::sincos_(x, s, c); }
PDDP_DEV void sincos2(double x, double& s, double& c) { pddp::sincos_(x, s, c); }
template <typename T> PDDP_DEV void sincos2(Dual<T> a, Dual<T>& s, Dual<T>& c) {
  T sv, cv;
  pddp::sincos_(a.p, sv, cv);
  s = {sv, cv * a.t};
  c = {cv, -sv * a.t};
}
template <typename T> PDDP_DEV Dual<T> sqrt_(Dual<T> a) { const T s = sqrt_(a.p); return {s, a.t / ((T)2 * s)}; }
template <typename T> PDDP_DEV T prim(T a) { return a; }
template <typename T> PDDP_DEV T prim(Dual<T> a) { return a.p; }
template <typename T> PDDP_DEV T tang(T) { return (T)0; }
template <typename T> PDDP_DEV T tang(Dual<T> a) { return a.t; }
template <typename X, typename T> PDDP_DEV X lift(T p, T t);
template <> PDDP_DEV float lift<float, float>(float p, float) { return p; }
template <> PDDP_DEV double lift<double, double>(double p, double) { return p; }
template <> PDDP_DEV Dual<float> lift<Dual<float>, float>(float p, float t) { return {p, t}; }
template <> PDDP_DEV Dual<double> lift<Dual<double>, double>(double p, double t) { return {p, t}; }

// Sum over the wavefront, in every lane.  float: six DPP additions (within the
// quads, the rows of 16, then row to row) and a read of lane 63; double: the
// butterfly through ds_bpermute
template <int CTRL, int ROWS>
PDDP_DEV float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true));
}
PDDP_DEV float wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1 0 3 2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2 3 0 1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane has its row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
PDDP_DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// exp of an argument already multiplied by log2(e): one v_exp_f32 (a result
// below the normal range is flushed to zero - such a term weighs nothing)
PDDP_DEV float exp2_scaled(float x) { return __builtin_amdgcn_exp2f(x); }
PDDP_DEV double exp2_scaled(double x) { return exp(x); }
template <typename T> PDDP_DEV T exp_scale();
template <> PDDP_DEV float exp_scale<float>() { return 1.4426950408889634f; }
template <> PDDP_DEV double exp_scale<double>() { return 1.0; }

// ---- LDS layout (in T) ---------------------------------------------------------
template <int E, int D>
struct Lds {
  static constexpr int NP = E * (E + 1) / 2, NS = E + NP, DD = D * D;
  // nu in pairs of training points: (j, p) at (j >> 1) PS + 2 p + (j & 1) - a
  // 16-byte read gives (p, p + 1) of two points, the operands of v_pk_fma_f32
  static constexpr int PS = (2 * D + 3) & ~3;
  int m, S, Sx, Cxf, G, ld, nu, lk, be, mu, W, h, c, F, Sig, ub, sj;  // always
  int g, Y1, Y2, dm, dS, dSx, dCxf, gmu, GSmu, gmS, GSS, dW, dO;  // JAC
  int total;
  // `gstore`: the per-point g_i = G_a nu_i of the Jacobian form (E M D words,
  // 216 of the 336 bytes a training point costs it in float) are kept from
  // A2 for C's tangents of W_a.  False - the launcher's choice when the
  // layout would not fit 160 KB otherwise (round 5) - they are formed again
  // where C needs them (81 FMAs per point and task: C costs twice as much, the
  // double cartpole's limit goes from 318 to 890 training points, f64 74 to
  // 208)
  bool gstore;
  PDDP_HD Lds(int M, int K, bool jac, bool gstore_ = true) : gstore(gstore_) {
    int o = 0;
    auto take = [&](int k) { const int r = o; o += (k + 3) & ~3; return r; };
    m = take(D); S = take(DD); Sx = take(E * E); Cxf = take(E * D);
    G = take(NS * DD); ld = take(NS); nu = take(((M + 1) >> 1) * PS); lk = take(E * M);
    be = take(E * M); mu = take(E); W = take(E * D); h = take(E * D); c = take(E);
    F = take(NP); Sig = take(NP); ub = take(4 * 2 * (((M + 3) >> 2) << 1)); sj = take(4 * 2 * (((M + 3) >> 2) << 1));
    g = Y1 = Y2 = dm = dS = dSx = dCxf = gmu = GSmu = gmS = GSS = dW = dO = 0;
    if (jac) {
      g = gstore ? take(E * M * D) : 0; Y1 = take(NP * D); Y2 = take(NP * DD);
      dm = take(K * D); dS = take(K * DD); dSx = take(K * E * E); dCxf = take(K * E * D);
      gmu = take(E * D); GSmu = take(E * DD); gmS = take(NP * D); GSS = take(NP * DD);
      dW = take(K * E * D); dO = take(K * NS);
    }
    total = o;
  }
};

// the Jacobian form's layout: with g_i kept when that fits a workgroup's LDS
constexpr long long kGpLdsMax = 160 * 1024;
template <int E, int D>
PDDP_HD Lds<E, D> lds_of(int M, int K, bool jac, int element_size) {
  const Lds<E, D> full(M, K, jac, true);
  if (!jac || (long long)full.total * element_size <= kGpLdsMax) return full;
  return Lds<E, D>(M, K, jac, false);
}

PDDP_DEV void pair_of(int item, int E, int& a, int& b) {
  a = 0;
  while (item >= E - a) item -= E - a, ++a;
  b = a + item;
}

// Encoded input `idx` of this row as the scalar type X: lane k's tangent seed
template <typename X, typename T>
PDDP_DEV X seed(T v, int idx, int k) {
  return lift<X, T>(v, idx == k ? (T)1 : (T)0);
}

// The decoded state covariance, element (i, j), on demand (utils/encoding.py
// decode_covar): nothing of the front end is held in arrays
template <typename X, typename T, int E>
PDDP_DEV X sx_of(const T* z, int enc, int i, int j, int k) {
  if (enc == 1) {  // UPPER_TRIANGULAR_CHOLESKY: Sx = U^T U, U row-major triu
    X s = lift<X, T>((T)0, (T)0);
    const int lim = i < j ? i : j;
    for (int r = 0; r <= lim; ++r) {
      const int base = E + r * E - r * (r - 1) / 2;  // offset of U[r][r]
      s = s + seed<X, T>(z[base + (i - r)], base + (i - r), k) *
                  seed<X, T>(z[base + (j - r)], base + (j - r), k);
    }
    return s;
  }
  if (i != j) return lift<X, T>((T)0, (T)0);
  if (enc == 2) return seed<X, T>(z[E + i], E + i, k);  // VARIANCE_ONLY
  if (enc == 3) {                                      // STANDARD_DEVIATION_ONLY
    const X s = seed<X, T>(z[E + i], E + i, k);
    return s * s;
  }
  return lift<X, T>((T)1e-6, (T)0);  // IGNORE_UNCERTAINTY (encoding.py:209-212)
}

// In-register inverse and log-determinant of S + diag(delta) (SPD, D x D)
template <typename T, int D>
PDDP_DEV void spd_inverse(const T* S, const T (&delta)[D], T* G, T& logdet) {
  T a[D * (D + 1) / 2];  // lower triangle, row-major: (i, j <= i) at i(i+1)/2 + j
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i * (i + 1) / 2 + j] = S[i * D + j] + (i == j ? delta[i] : (T)0);
  T ld = 0;
#pragma unroll
  for (int j = 0; j < D; ++j) {  // Cholesky, column by column
    T d = a[j * (j + 1) / 2 + j];
#pragma unroll
    for (int r = 0; r < j; ++r) d -= a[j * (j + 1) / 2 + r] * a[j * (j + 1) / 2 + r];
    ld += log_(d);
    const T l = sqrt_(d), il = (T)1 / l;
    a[j * (j + 1) / 2 + j] = il;  // (the diagonal holds 1 / L_jj from here on)
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T s = a[i * (i + 1) / 2 + j];
#pragma unroll
      for (int r = 0; r < j; ++r) s -= a[i * (i + 1) / 2 + r] * a[j * (j + 1) / 2 + r];
      a[i * (i + 1) / 2 + j] = s * il;
    }
  }
  logdet = ld;
  // L^-1 in place (lower): column by column
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      T s = a[i * (i + 1) / 2 + j] * a[j * (j + 1) / 2 + j];  // L_ij * inv_jj
#pragma unroll
      for (int r = j + 1; r < i; ++r) s += a[i * (i + 1) / 2 + r] * a[r * (r + 1) / 2 + j];
      a[i * (i + 1) / 2 + j] = -s * a[i * (i + 1) / 2 + i];
    }
  }
  // G = L^-T L^-1
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      T s = 0;
#pragma unroll
      for (int r = p; r < D; ++r) s += a[r * (r + 1) / 2 + p] * a[r * (r + 1) / 2 + q];
      G[p * D + q] = s;
      G[q * D + p] = s;
    }
}

template <typename T, int E, int D, bool JAC, bool ROLL = false>
PDDP_DEV void gp_step_body(const Args<T>& A, const Roll<T>& RL = Roll<T>()) {
  static_assert(!(JAC && ROLL), "");
  using X = typename std::conditional<JAC, Dual<T>, T>::type;
  constexpr int NP = E * (E + 1) / 2, NS = E + NP, DD = D * D, PS = Lds<E, D>::PS;
  extern __shared__ __align__(32) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int M = A.M, n = A.n, K = n + A.m_act;
  const Lds<E, D> o = lds_of<E, D>(M, K, JAC, (int)sizeof(T));
  auto nu_at = [&](int i, int p) -> T& { return sm[o.nu + (i >> 1) * PS + 2 * p + (i & 1)]; };
  const int row = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if constexpr (!ROLL) {
    if (A.row_mask != nullptr && A.row_mask[row / A.rows_per_mask] == 0) return;
  }
  const T* z = A.z + (size_t)row * n;
  const T* u = A.u + (size_t)row * A.m_act;
  T* z_out = A.z_next + (size_t)row * n;
  // (ROLL: the action, formed here and kept in four scalars - an array indexed
  // at run time would live in scratch)
  static_assert(kMaxAct == 4, "act() below selects among four scalars");
  [[maybe_unused]] T ua0 = 0, ua1 = 0, ua2 = 0, ua3 = 0;
  [[maybe_unused]] auto act = [&](int r) -> T { return r == 0 ? ua0 : r == 1 ? ua1 : r == 2 ? ua2 : ua3; };
  if constexpr (ROLL) {
    const int b = row / RL.A, ai = row - b * RL.A, t = RL.t, m_ = A.m_act;
    if (RL.active != nullptr && RL.active[b] == 0) return;
    if (RL.status != nullptr && RL.status[b] != 0) return;
    const T* zn = RL.Z + ((size_t)b * (RL.N + 1) + t) * n;  // nominal state
    T* zc = RL.Zc + (((size_t)b * (RL.N + 1) + t) * RL.A + ai) * n;
    // Z_new[0] = Z[0] (ilqr.py:690): the candidates start on the nominal
    z = t == 0 ? zn : zc;
    if (t == 0 && tid < n) zc[tid] = zn[tid];
    z_out = zc + (size_t)RL.A * n;
    // (the terminal launch has no action: its feature slot reads zeros)
    if (!RL.terminal) {
      // every thread forms the action itself (n FMAs on broadcast loads)
      const T* g = RL.gains + ((size_t)b * RL.N + t) * (m_ + m_ * n);
      const T alpha = RL.alphas[ai];
      for (int r = 0; r < m_; ++r) {
        T sK = 0;
        for (int c = 0; c < n; ++c) sK = fma_(z[c] - zn[c], g[m_ + r * n + c], sK);
        T v = RL.U[((size_t)b * RL.N + t) * m_ + r] + fma_(alpha, g[r], sK);
        if (RL.u_min != nullptr) v = clamp_nan(v, RL.u_min[r], RL.u_max[r]);
        ua0 = r == 0 ? v : ua0;
        ua1 = r == 1 ? v : ua1;
        ua2 = r == 2 ? v : ua2;
        ua3 = r == 3 ? v : ua3;
        if (tid == 0) RL.Uc[(((size_t)b * RL.N + t) * RL.A + ai) * m_ + r] = v;
      }
    }
  }
  const int nn = A.n_non, nang = A.n_ang, na = nn + 2 * nang, enc = A.encoding;

  // ---- front end, element by element, generic in the scalar type ---------------
  auto sx = [&](int i, int j, int k) { return sx_of<X, T, E>(z, enc, i, j, k); };
  auto mxv = [&](int i, int k) { return seed<X, T>(z[i], i, k); };
  // E[sin], E[cos] of angle q
  auto ang_mean = [&](int q, int k, X& es, X& ec) {
    const int ai = A.ang[q];
    const X damp = exp_((T)-0.5 * sx(ai, ai, k)), mu_ = mxv(ai, k);
    X sv, cv;
    sincos2(mu_, sv, cv);
    es = damp * sv;
    ec = damp * cv;
  };
  // feature mean p (p < na), action appended behind
  auto m_of = [&](int p, int k) -> X {
    if (p < nn) return mxv(A.non[p], k);
    if (p < na) {
      X es, ec;
      ang_mean((p - nn) >> 1, k, es, ec);
      return ((p - nn) & 1) ? ec : es;
    }
    if constexpr (ROLL) return seed<X, T>(act(p - na), n + (p - na), k);
    else return seed<X, T>(u[p - na], n + (p - na), k);
  };
  // feature covariance (p, q), utils/angular.py augment_moments
  auto S_of = [&](int p, int q, int k) -> X {
    const X zero = lift<X, T>((T)0, (T)0);
    if (p >= na || q >= na) return zero;
    if (p > q) { const int t_ = p; p = q; q = t_; }
    if (q < nn) return sx(A.non[p], A.non[q], k);
    if (p < nn) {  // x, sin / cos: C[angle, x] E[cos], -C[angle, x] E[sin]
      const int qa = (q - nn) >> 1;
      X es, ec;
      ang_mean(qa, k, es, ec);
      const X cv = sx(A.ang[qa], A.non[p], k);
      return ((q - nn) & 1) ? -(cv * es) : cv * ec;
    }
    const int ka = (p - nn) >> 1, la = (q - nn) >> 1;
    const int ia = A.ang[ka], ja = A.ang[la];
    const X vi = sx(ia, ia, k), vj = sx(ja, ja, k), ci = sx(ia, ja, k);
    const X lq = (T)-0.5 * (vi + vj), qq = exp_(lq);
    const X ep = exp_(lq + ci) - qq, em = exp_(lq - ci) - qq;
    const X mi = mxv(ia, k), mj = mxv(ja, k);
    const bool ps = !((p - nn) & 1), qs = !((q - nn) & 1);  // sin rows
    X sd, cd, ss_, cs_;  // of the difference and of the sum, one evaluation each
    sincos2(mi - mj, sd, cd);
    sincos2(mi + mj, ss_, cs_);
    if (ps && qs) return (T)0.5 * (ep * cd - em * cs_);
    if (!ps && !qs) return (T)0.5 * (ep * cd + em * cs_);
    if (ps) return (T)0.5 * (ep * sd + em * ss_);  // sin_k, cos_l
    return (T)0.5 * (em * ss_ - ep * sd);          // cos_k, sin_l: sin(m_l - m_k) = -sin(m_k - m_l)
  };
  // cov[x_r, feature q] (Stein's lemma)
  auto cxf = [&](int r, int q, int k) -> X {
    if (q < nn) return sx(r, A.non[q], k);
    if (q >= na) return lift<X, T>((T)0, (T)0);
    const int qa = (q - nn) >> 1;
    X es, ec;
    ang_mean(qa, k, es, ec);
    const X cv = sx(r, A.ang[qa], k);
    return ((q - nn) & 1) ? -(cv * es) : cv * ec;
  };

  // ---- A0: one task per (entry, input k) over all threads ------------------------------
  PDDP_GP_MARK(0);
  {
    constexpr int nS = D * (D + 1) / 2, nX = E * (E + 1) / 2;
    const int n_entries = D + nS + E * D + nX, Kq = JAC ? K : 1;
    for (int task = tid; task < n_entries * Kq; task += kThreads) {
      // (input fastest: the lanes of a wavefront share the entry - one code path)
      int e = JAC ? task / Kq : task;
      const int k = JAC ? task - e * Kq : -1;
      const bool first = !JAC || k == 0;
      if (e < D) {
        const X v = m_of(e, k);
        if (first) sm[o.m + e] = prim(v);
        if (JAC) sm[o.dm + k * D + e] = tang(v);
        continue;
      }
      e -= D;
      if (e < nS) {
        int p = 0;
        while (e >= D - p) e -= D - p, ++p;
        const int q = p + e;
        const X v = S_of(p, q, k);
        if (first) sm[o.S + p * D + q] = sm[o.S + q * D + p] = prim(v);
        if (JAC) sm[o.dS + k * DD + p * D + q] = sm[o.dS + k * DD + q * D + p] = tang(v);
        continue;
      }
      e -= nS;
      if (e < E * D) {
        const X v = cxf(e / D, e % D, k);
        if (first) sm[o.Cxf + e] = prim(v);
        if (JAC) sm[o.dCxf + k * E * D + e] = tang(v);
        continue;
      }
      e -= E * D;
      int a = 0;
      while (e >= E - a) e -= E - a, ++a;
      const int b = a + e;
      const X v = sx(a, b, k);
      if (first) sm[o.Sx + a * E + b] = sm[o.Sx + b * E + a] = prim(v);
      if (JAC) sm[o.dSx + (k * E + a) * E + b] = sm[o.dSx + (k * E + b) * E + a] = tang(v);
    }
  }
  __syncthreads();

  if constexpr (ROLL) {
    // stage / terminal cost of (z, u): E[(x~ - g)^T Q (x~ - g)] + (u - ug)^T R
    // (u - ug) on the augmented state = the first na features, whose mean and
    // covariance the front end just left in LDS.  The covariance enters with
    // the encoding (quadratic.py:92 and the variance-only encodings: mean only
    // / diagonal only).  Wave 3, lane (i, j): one product each
    if (wave == 3) {
      const int na_ = RL.na, i = lane / 8, j = lane & 7;
      T term = 0;
      if (i < na_ && j < na_) {
        const T* Qm = RL.terminal ? RL.Qt : RL.Q;
        // (mean only: the augmented STATE, sin / cos of the mean itself -
        // utils/angular.py augment_state - not the moment-matched features,
        // which carry the 1e-6 placeholder variance's damping)
        auto feat = [&](int p) -> T {
          if (enc != 4) return sm[o.m + p];
          if (p < nn) return z[A.non[p]];
          T sv, cv;
          sincos2(z[A.ang[(p - nn) >> 1]], sv, cv);
          return ((p - nn) & 1) ? cv : sv;
        };
        const T di = feat(i) - RL.xg[i], dj = feat(j) - RL.xg[j];
        T second = di * dj;
        if (enc == 1 || ((enc == 2 || enc == 3) && i == j))
          second += sm[o.S + i * D + j];
        term = second * Qm[i * na_ + j];
      }
      if (!RL.terminal && lane >= 56) {  // (lanes 56..: i = 7 >= na or spare)
        const int r = lane - 56, m_ = A.m_act;
        if (r < m_) {
          T acc = 0;
          for (int c = 0; c < m_; ++c)
            acc += (act(c) - RL.ug[c]) * RL.Rm[c * m_ + r];
          term += acc * (act(r) - RL.ug[r]);
        }
      }
      const T cost = wave_sum(term);
      if (lane == 0) {
        T* Jp = RL.Jc + row;
        *Jp = (RL.t == 0 && !RL.terminal ? (T)0 : *Jp) + cost;
      }
    }
    if (RL.terminal) return;
  }

  // ---- A1 ------------------------------------------------------------------------
  PDDP_GP_MARK(1);
  if (wave == 0) {
    if (lane < NS) {
      T delta[D];
      if (lane < E) {
#pragma unroll
        for (int p = 0; p < D; ++p) delta[p] = (T)1 / A.iL[lane * D + p];
      } else {
        int a, b;
        pair_of(lane - E, E, a, b);
#pragma unroll
        for (int p = 0; p < D; ++p) delta[p] = (T)1 / (A.iL[a * D + p] + A.iL[b * D + p]);
      }
      T ld;
      spd_inverse<T, D>(sm + o.S, delta, sm + o.G + lane * DD, ld);
      sm[o.ld + lane] = ld;
    }
  } else {
    for (int e = tid - 64; e < M * D; e += kThreads - 64) {
      const int i = e / D, p = e - i * D;
      nu_at(i, p) = A.Xt[e] - sm[o.m + p];
    }
    if ((M & 1) && tid - 64 < D) nu_at(M, tid - 64) = 0;  // the pair partner of the last point
  }
  __syncthreads();
  for (int e = tid; e < E * M; e += kThreads) {  // log k_a(x_i, m)
    const int a = e / M, i = e - a * M;
    T s = 0;
#pragma unroll
    for (int p = 0; p < D; ++p) s += nu_at(i, p) * nu_at(i, p) * A.iL[a * D + p];
    sm[o.lk + e] = log_(A.sf2[a]) - (T)0.5 * s;
  }

  // ---- A2: the mean and the input-output covariance --------------------------------
  PDDP_GP_MARK(2);
  for (int a = wave; a < E; a += 4) {
    const T* Ga = sm + o.G + a * DD;
    T s0 = 0, s1[D], gg[JAC ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) s1[p] = 0;
    if (JAC) {
#pragma unroll
      for (int e = 0; e < D * (D + 1) / 2; ++e) gg[e] = 0;
    }
    for (int i = lane; i < M; i += 64) {
      T nu[D], g[D], quad = 0;
#pragma unroll
      for (int p = 0; p < D; ++p) nu[p] = nu_at(i, p);
#pragma unroll
      for (int p = 0; p < D; ++p) {
        T s = 0;
#pragma unroll
        for (int q = 0; q < D; ++q) s += Ga[p * D + q] * nu[q];
        g[p] = s;
        quad += s * nu[p];
      }
      const T be = A.beta[a * M + i] * exp_((T)-0.5 * quad);
      sm[o.be + a * M + i] = be;
      s0 += be;
#pragma unroll
      for (int p = 0; p < D; ++p) s1[p] += be * nu[p];
      if (JAC) {
#pragma unroll
        for (int p = 0; p < D; ++p) {
          if (o.gstore) sm[o.g + (a * M + i) * D + p] = g[p];
#pragma unroll
          for (int q = 0; q <= p; ++q) gg[p * (p + 1) / 2 + q] += be * g[p] * g[q];
        }
      }
    }
    s0 = wave_sum(s0);
#pragma unroll
    for (int p = 0; p < D; ++p) s1[p] = wave_sum(s1[p]);
    T sl = 0;  // sum log ell^2
#pragma unroll
    for (int p = 0; p < D; ++p) sl -= log_(A.iL[a * D + p]);
    const T c = A.sf2[a] * exp_((T)-0.5 * (sm[o.ld + a] - sl));
    const T mu = c * s0;
    if (lane == 0) sm[o.mu + a] = mu, sm[o.c + a] = c;
    if (lane < D) {
      T w = 0;
#pragma unroll
      for (int q = 0; q < D; ++q) w += Ga[lane * D + q] * (c * s1[q]);
      sm[o.W + a * D + lane] = w;
      T hv = 0;  // (s1 is in every lane after the reduction; select without indexing)
#pragma unroll
      for (int q = 0; q < D; ++q) hv = (q == lane) ? c * s1[q] : hv;
      sm[o.h + a * D + lane] = hv;
      if (JAC) sm[o.gmu + a * D + lane] = w;  // d mu_a / d m = W_a
    }
    if (JAC) {  // d mu_a / d S = -1/2 mu_a A_a + 1/2 c sum beta e g g^T
#pragma unroll
      for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          const T v = wave_sum(gg[p * (p + 1) / 2 + q]);
          if (lane == 0) {
            const T r = (T)0.5 * (c * v - mu * Ga[p * D + q]);
            sm[o.GSmu + a * DD + p * D + q] = r;
            sm[o.GSmu + a * DD + q * D + p] = r;
          }
        }
    }
  }
  __syncthreads();

  // ---- B: the M^2 sums of every pair ---------------------------------------------
  PDDP_GP_MARK(3);
  for (int item = wave; item < NP; item += 4) {
    int a, b;
    pair_of(item, E, a, b);
    const T* Gs = sm + o.G + (E + item) * DD;
    T iLa[D], iLb[D], lam[D];
#pragma unroll
    for (int p = 0; p < D; ++p) {
      iLa[p] = A.iL[a * D + p];
      iLb[p] = A.iL[b * D + p];
      lam[p] = (T)1 / (iLa[p] + iLb[p]);
    }
    const int MP = (M + 1) >> 1;          // pairs of training points
    const int MQ = ((M + 3) >> 2) << 2;    // points, padded to pairs of pairs
    T* ub = sm + o.ub + wave * MQ;  // this wavefront's scratch: u_b of the points, in pairs
    const T kx = exp_scale<T>();       // exponents in units of ln 2 for float
    T* sj = sm + o.sj + wave * MQ;
    // T x = lam x - lam G (lam x)
    auto t_apply = [&](const T (&x)[D], T (&y)[D]) {
      T lx[D];
#pragma unroll
      for (int p = 0; p < D; ++p) lx[p] = lam[p] * x[p];
#pragma unroll
      for (int p = 0; p < D; ++p) {
        T s = 0;
#pragma unroll
        for (int q = 0; q < D; ++q) s += Gs[p * D + q] * lx[q];
        y[p] = lx[p] - lam[p] * s;
      }
    };
    // u_b[j] = log k_b(x_j, m) + 1/2 z_bj^T T z_bj
    for (int j = lane; j < M; j += 64) {
      T zb[D], tz[D], s = 0;
#pragma unroll
      for (int p = 0; p < D; ++p) zb[p] = iLb[p] * nu_at(j, p);
      t_apply(zb, tz);
#pragma unroll
      for (int p = 0; p < D; ++p) s += zb[p] * tz[p];
      ub[j] = kx * (sm[o.lk + b * M + j] + (T)0.5 * s);
      sj[j] = 0;
    }
    if (lane < MQ - M) ub[M + lane] = (T)-1e30, sj[M + lane] = 0;  // phantom points: weigh nothing
    // (same-wavefront LDS traffic is in order; the compiler needs telling)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T Fa = 0, y1[D], y2[JAC ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) y1[p] = 0;
    if (JAC) {
#pragma unroll
      for (int e = 0; e < D * (D + 1) / 2; ++e) y2[e] = 0;
    }
    const int tiles = (M + 63) >> 6;
    for (int it = 0; it < tiles; ++it) {
      const int i = it * 64 + lane;
      const bool live = i < M;
      const int ii = live ? i : 0;
      T za[D], tza[D], ua = 0;
#pragma unroll
      for (int p = 0; p < D; ++p) za[p] = iLa[p] * nu_at(ii, p);
      t_apply(za, tza);
#pragma unroll
      for (int p = 0; p < D; ++p) ua += za[p] * tza[p];
      ua = kx * (sm[o.lk + a * M + ii] + (T)0.5 * ua);
      // z_bj = L_b^-1 nu_j: the factor goes to this lane's side of the product
      T tzb[D];
#pragma unroll
      for (int p = 0; p < D; ++p) tzb[p] = kx * tza[p] * iLb[p];
      const T bai = live ? A.beta[a * M + ii] : (T)0;  // (a dead lane weighs nothing)
      const T* krow = A.Kinv + (size_t)a * M * M + ii;  // (symmetric: row j, coalesced)
      using V4 = T __attribute__((ext_vector_type(4)));
      using V2 = T __attribute__((ext_vector_type(2)));
      V2 r2 = {0, 0}, vn2[D], tzb2[D];
#pragma unroll
      for (int p = 0; p < D; ++p) vn2[p] = V2{0, 0}, tzb2[p] = V2{tzb[p], tzb[p]};
      // The training inputs are the same for every row and lane: they come
      // through the SCALAR cache (constant address space: s_load_dwordx16 into
      // SGPR pairs that v_pk_fma_f32 takes as an operand), not through LDS -
      // six 16-byte LDS reads per step were what the loop waited for.  nu_j =
      // x_j - m: the m part is this lane's constant, folded into u_a
      T uam = ua;
#pragma unroll
      for (int p = 0; p < D; ++p) uam -= tzb[p] * sm[o.m + p];
      const V2 ua2 = {uam, uam}, bai2 = {bai, bai};
      typedef const __attribute__((address_space(4))) V4* CV4;
      typedef const __attribute__((address_space(4))) V2* CV2;
      const CV4 xp = (CV4)(uintptr_t)A.XtP;
      const CV2 bp = (CV2)(uintptr_t)A.betaP + __builtin_amdgcn_readfirstlane(b * (MQ >> 1));
      // two training points j per step, as the two halves of packed operations
      // a step's operands (scalar registers: the training inputs and weights of
      // two points; u_b from LDS) are requested for TWO steps before the first
      // is computed: one scalar-load latency per two steps
      struct Step {
        V2 n2[PS / 2], uq, bq;
      };
      auto fetch = [&](int jp) {
        Step st;
        st.uq = *reinterpret_cast<const V2*>(ub + 2 * jp);
        st.bq = bp[jp];
#pragma unroll
        for (int q = 0; q < PS / 4; ++q) {
          const V4 t4 = xp[jp * (PS / 4) + q];
          st.n2[2 * q] = V2{t4.x, t4.y}, st.n2[2 * q + 1] = V2{t4.z, t4.w};
        }
        return st;
      };
      auto body = [&](int jp, const Step& st, V2 kv) {
        V2 e = ua2 + st.uq;
#pragma unroll
        for (int p = 0; p < D; ++p) e = tzb2[p] * st.n2[p] + e;
        const V2 w = bai2 * st.bq - kv;
        const V2 ex = {exp2_scaled(e.x), exp2_scaled(e.y)};
        if (!JAC) {
          r2 = w * ex + r2;
        } else {
          const V2 c = w * ex;
          r2 += c;
#pragma unroll
          for (int p = 0; p < D; ++p) vn2[p] = c * st.n2[p] + vn2[p];
          const T c0 = wave_sum(c.x), c1 = wave_sum(c.y);  // column sums of this tile
          if (lane == 0) sj[2 * jp] += c0, sj[2 * jp + 1] += c1;
        }
      };
      // (the pair arrays are padded to an even number of pairs: step MP of an
      // odd MP reads zeros - weight 0 - and u_b = -1e30)
      const int MP2 = (MP + 1) & ~1;
      for (int j0 = 0; j0 < MP2; j0 += 2) {
        V2 kv0 = {0, 0}, kv1 = {0, 0};
        if (a == b) {  // K_a^-1[i][j]: from HBM / L2, four requests in flight
          const int j = 2 * j0;
          kv0.x = (live && j < M) ? krow[(size_t)j * M] : (T)0;
          kv0.y = (live && j + 1 < M) ? krow[(size_t)(j + 1) * M] : (T)0;
          kv1.x = (live && j + 2 < M) ? krow[(size_t)(j + 2) * M] : (T)0;
          kv1.y = (live && j + 3 < M) ? krow[(size_t)(j + 3) * M] : (T)0;
        }
        const Step s0 = fetch(j0), s1 = fetch(j0 + 1);
        body(j0, s0, kv0);
        body(j0 + 1, s1, kv1);
      }
      const T r = r2.x + r2.y;
      T vn[D];  // sum_j c_ij nu_j = sum_j c_ij x_j - m sum_j c_ij
#pragma unroll
      for (int p = 0; p < D; ++p) vn[p] = vn2[p].x + vn2[p].y - sm[o.m + p] * r;
      Fa += r;
      if (JAC) {
#pragma unroll
        for (int p = 0; p < D; ++p) vn[p] *= iLb[p];  // v_i = sum_j c_ij z_bj
#pragma unroll
        for (int p = 0; p < D; ++p) {
          y1[p] += r * za[p] + vn[p];
#pragma unroll
          for (int q = 0; q <= p; ++q)
            y2[p * (p + 1) / 2 + q] += r * za[p] * za[q] + za[p] * vn[q] + vn[p] * za[q];
        }
      }
    }
    Fa = wave_sum(Fa);
    if (lane == 0) sm[o.F + item] = Fa;
    if (JAC) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int j = lane; j < M; j += 64) {  // sum_j s_j z_bj z_bj^T
        const T s = sj[j];
        T zb[D];
#pragma unroll
        for (int p = 0; p < D; ++p) zb[p] = iLb[p] * nu_at(j, p);
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
          for (int q = 0; q <= p; ++q) y2[p * (p + 1) / 2 + q] += s * zb[p] * zb[q];
      }
#pragma unroll
      for (int p = 0; p < D; ++p) {
        const T t1 = wave_sum(y1[p]);
        if (lane == 0) sm[o.Y1 + item * D + p] = t1;
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          const T t2 = wave_sum(y2[p * (p + 1) / 2 + q]);
          if (lane == 0) sm[o.Y2 + item * DD + p * D + q] = sm[o.Y2 + item * DD + q * D + p] = t2;
        }
      }
    }
  }
  __syncthreads();

  // ---- C: Sigma_ab and (JAC) the gradients with respect to (m, S) -------------------
  PDDP_GP_MARK(4);
  // kappa_ab = det(R)^-1/2 = exp(-1/2 (log det(S + lam) - sum log lam))
  auto kappa_of = [&](int item, int a, int b) {
    T sl = 0;
    for (int p = 0; p < D; ++p) sl -= log_(A.iL[a * D + p] + A.iL[b * D + p]);
    return exp_((T)-0.5 * (sm[o.ld + E + item] - sl));
  };
  if (tid < NP) {
    int a, b;
    pair_of(tid, E, a, b);
    T s = kappa_of(tid, a, b) * sm[o.F + tid] - sm[o.mu + a] * sm[o.mu + b];
    if (a == b) s += A.sf2[a] + A.sn2[a];
    sm[o.Sig + tid] = s;
  }
  if (JAC) {
    // thread (item, p): row p of H = 1/2 G (lam Y2 lam) G and of the gradients
    for (int e = tid; e < NP * D; e += kThreads) {
      const int item = e / D, p = e - item * D;
      int a, b;
      pair_of(item, E, a, b);
      const T* Gs = sm + o.G + (E + item) * DD;
      const T* Y2 = sm + o.Y2 + item * DD;
      const T kap = kappa_of(item, a, b), Fv = sm[o.F + item];
      const T mua = sm[o.mu + a], mub = sm[o.mu + b];
      T lam[D], tmp[D];
#pragma unroll
      for (int q = 0; q < D; ++q) lam[q] = (T)1 / (A.iL[a * D + q] + A.iL[b * D + q]);
#pragma unroll
      for (int q = 0; q < D; ++q) {  // tmp = (G lam Y2 lam)[p][:]
        T s = 0;
#pragma unroll
        for (int r = 0; r < D; ++r) s += Gs[p * D + r] * lam[r] * Y2[r * D + q];
        tmp[q] = s * lam[q];
      }
      T gm = 0;  // (G (lam Y1))[p]
#pragma unroll
      for (int r = 0; r < D; ++r) gm += Gs[p * D + r] * lam[r] * sm[o.Y1 + item * D + r];
      sm[o.gmS + item * D + p] = kap * gm - mub * sm[o.gmu + a * D + p] - mua * sm[o.gmu + b * D + p];
#pragma unroll
      for (int q = 0; q < D; ++q) {
        T hv = 0;
#pragma unroll
        for (int r = 0; r < D; ++r) hv += tmp[r] * Gs[r * D + q];
        sm[o.GSS + item * DD + p * D + q] =
            kap * ((T)0.5 * hv - (T)0.5 * Fv * Gs[p * D + q]) -
            mub * sm[o.GSmu + a * DD + p * D + q] - mua * sm[o.GSmu + b * DD + p * D + q];
      }
    }
    // tangent of W_a along input k:
    //   dW = A [-dS W - 1/2 tr(A dS) h + c (sum_i beta e_i (g_i.dm + 1/2 g_i^T dS g_i) nu_i) - mu dm]
    for (int task = tid; task < K * E; task += kThreads) {
      const int k = task / E, a = task - k * E;
      const T* Ga = sm + o.G + a * DD;
      const T* dS = sm + o.dS + k * DD;
      const T* dm = sm + o.dm + k * D;
      T ds[D * (D + 1) / 2], dmv[D], acc[D];
      T tr = 0;
#pragma unroll
      for (int p = 0; p < D; ++p) {
        dmv[p] = dm[p];
        acc[p] = 0;
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          ds[p * (p + 1) / 2 + q] = dS[p * D + q];
          tr += (p == q ? (T)1 : (T)2) * Ga[p * D + q] * dS[p * D + q];
        }
      }
      for (int i = 0; i < M; ++i) {
        T gv[D], q2 = 0, gd = 0;
        if (o.gstore) {
          const T* g = sm + o.g + (a * M + i) * D;
#pragma unroll
          for (int p = 0; p < D; ++p) gv[p] = g[p];
        } else {  // g_i = G_a nu_i, formed again (the same sums as in A2)
          T nu[D];
#pragma unroll
          for (int p = 0; p < D; ++p) nu[p] = nu_at(i, p);
#pragma unroll
          for (int p = 0; p < D; ++p) {
            T s_ = 0;
#pragma unroll
            for (int q = 0; q < D; ++q) s_ += Ga[p * D + q] * nu[q];
            gv[p] = s_;
          }
        }
#pragma unroll
        for (int p = 0; p < D; ++p) gd += gv[p] * dmv[p];
#pragma unroll
        for (int p = 0; p < D; ++p) {
          T s = (T)0.5 * ds[p * (p + 1) / 2 + p] * gv[p];
#pragma unroll
          for (int q = 0; q < p; ++q) s += ds[p * (p + 1) / 2 + q] * gv[q];
          q2 += s * gv[p];  // 1/2 g^T dS g
        }
        const T coef = sm[o.be + a * M + i] * (gd + q2);
#pragma unroll
        for (int p = 0; p < D; ++p) acc[p] += coef * nu_at(i, p);
      }
      const T c = sm[o.c + a], mu = sm[o.mu + a];
      T vec[D];
#pragma unroll
      for (int p = 0; p < D; ++p) {
        T s = 0;
#pragma unroll
        for (int q = 0; q < D; ++q) s += dS[p * D + q] * sm[o.W + a * D + q];
        vec[p] = -s - (T)0.5 * tr * sm[o.h + a * D + p] + c * acc[p] - mu * dmv[p];
      }
#pragma unroll
      for (int p = 0; p < D; ++p) {
        T s = 0;
#pragma unroll
        for (int q = 0; q < D; ++q) s += Ga[p * D + q] * vec[q];
        sm[o.dW + (k * E + a) * D + p] = s;
      }
    }
  }
  if (JAC) {
    // tangent of mu_a (o < E) and Sigma_ab (o >= E) along input k: gradient . (dm_k, dS_k)
    __syncthreads();
    for (int task = tid; task < K * NS; task += kThreads) {
      const int oo = task / K, k = task - oo * K;  // (k fastest: the gradient is a broadcast)
      const T* gm = oo < E ? sm + o.gmu + oo * D : sm + o.gmS + (oo - E) * D;
      const T* GS = oo < E ? sm + o.GSmu + oo * DD : sm + o.GSS + (oo - E) * DD;
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int p = 0; p < D; ++p) s0 += gm[p] * sm[o.dm + k * D + p];
#pragma unroll 9
      for (int e = 0; e < DD; ++e) s1 += GS[e] * sm[o.dS + k * DD + e];
      sm[o.dO + k * NS + oo] = s0 + s1;
    }
  }
  __syncthreads();

  // ---- A3: next mean, covariance and encoding; lane k carries input k ---------------
  PDDP_GP_MARK(5);
  if (wave == 0 && (JAC ? lane < K : lane == 0)) {
    const int k = lane;
    X Mn[E], Cn[E * (E + 1) / 2];  // upper triangle, row-major: (r, c >= r)
    auto up = [&](int r, int c) { return r * E - r * (r - 1) / 2 + (c - r); };
#pragma unroll
    for (int a = 0; a < E; ++a)
      Mn[a] = mxv(a, k) + lift<X, T>(sm[o.mu + a], JAC ? sm[o.dO + k * NS + a] : (T)0);
    {
      int item = 0;
#pragma unroll
      for (int a = 0; a < E; ++a)
#pragma unroll
        for (int b = a; b < E; ++b, ++item)
          Cn[up(a, b)] = lift<X, T>(sm[o.Sx + a * E + b], JAC ? sm[o.dSx + (k * E + a) * E + b] : (T)0) +
                         lift<X, T>(sm[o.Sig + item], JAC ? sm[o.dO + k * NS + E + item] : (T)0);
    }
#pragma unroll
    for (int r = 0; r < E; ++r)
#pragma unroll
      for (int a = 0; a < E; ++a) {  // C[r][a] = sum_q cov[x_r, f_q] W_a[q]
        X s = lift<X, T>((T)0, (T)0);
#pragma unroll
        for (int q = 0; q < na; ++q)
          s = s + lift<X, T>(sm[o.Cxf + r * D + q], JAC ? sm[o.dCxf + (k * E + r) * D + q] : (T)0) *
                      lift<X, T>(sm[o.W + a * D + q], JAC ? sm[o.dW + (k * E + a) * D + q] : (T)0);
        // C + C^T on the upper triangle: (r, a) and (a, r) both land on (min, max)
        const int lo = r < a ? r : a, hi = r < a ? a : r;
        Cn[up(lo, hi)] = Cn[up(lo, hi)] + (r == a ? s + s : s);
      }
    T* out = z_out;
    auto emit = [&](int idx, X v) {
      if (lane == 0) out[idx] = prim(v);
      if (JAC) {
        if (k < n) A.Fz[((size_t)row * n + idx) * n + k] = tang(v);
        else A.Fu[((size_t)row * n + idx) * A.m_act + (k - n)] = tang(v);
      }
    };
#pragma unroll
    for (int a = 0; a < E; ++a) emit(a, Mn[a]);
    if (enc == 1) {
      // upper Cholesky U^T U = Cn + jitter (utils/encoding.py _cholesky_upper:
      // 1e-12, x10 while a pivot fails - per row here, per batch there)
      T jitter = (T)1e-12;
      X U[E * (E + 1) / 2];
      for (int attempt = 0; attempt < 14; ++attempt) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          X d = Cn[up(r, r)] + lift<X, T>(jitter, (T)0);
#pragma unroll
          for (int t_ = 0; t_ < r; ++t_) d = d - U[up(t_, r)] * U[up(t_, r)];
          ok = ok && prim(d) > (T)0;
          const X l = sqrt_(d);
          U[up(r, r)] = l;
#pragma unroll
          for (int c = r + 1; c < E; ++c) {
            X s = Cn[up(r, c)];
#pragma unroll
            for (int t_ = 0; t_ < r; ++t_) s = s - U[up(t_, r)] * U[up(t_, c)];
            U[up(r, c)] = s / l;
          }
        }
        if (ok) break;
        jitter *= (T)10;
      }
#pragma unroll
      for (int e = 0; e < E * (E + 1) / 2; ++e) emit(E + e, U[e]);
    } else if (enc == 2 || enc == 3) {
#pragma unroll
      for (int a = 0; a < E; ++a) {
        X v = Cn[up(a, a)];
        if (!(prim(v) > (T)1e-12)) v = lift<X, T>((T)1e-12, (T)0);  // clamp_min
        emit(E + a, enc == 3 ? sqrt_(v) : v);
      }
    }
  }
  PDDP_GP_MARK(6);
}

template <typename T, int E, int D, bool JAC>
__global__ __launch_bounds__(kThreads) void gp_step_kernel(const Args<T> A) {
  gp_step_body<T, E, D, JAC>(A);
}
#ifndef PDDP_GP_FWD_WAVES
#define PDDP_GP_FWD_WAVES 3
#endif
// The line search's kernel (float, no Jacobian) held to the registers of
// PDDP_GP_FWD_WAVES workgroups per CU: the serial front and back end of a row
// (A0, A1, A3: most of the workgroup waits) are filled by other rows' M^2 loops
template <int E, int D>
__global__ __launch_bounds__(kThreads)
__attribute__((amdgpu_waves_per_eu(PDDP_GP_FWD_WAVES, PDDP_GP_FWD_WAVES)))
void gp_step_fwd_f32_kernel(const Args<float> A) {
  gp_step_body<float, E, D, false>(A);
}
template <typename T, int E, int D>
__global__ __launch_bounds__(kThreads) void gp_roll_kernel(const Args<T> A, const Roll<T> RL) {
  gp_step_body<T, E, D, false, true>(A, RL);
}
template <int E, int D>
__global__ __launch_bounds__(kThreads)
__attribute__((amdgpu_waves_per_eu(PDDP_GP_FWD_WAVES, PDDP_GP_FWD_WAVES)))
void gp_roll_f32_kernel(const Args<float> A, const Roll<float> RL) {
  gp_step_body<float, E, D, false, true>(A, RL);
}
template <typename T, int E, int D>
struct RollKernels {
  static auto pick() { return gp_roll_kernel<T, E, D>; }
};
template <int E, int D>
struct RollKernels<float, E, D> {
  static auto pick() { return gp_roll_f32_kernel<E, D>; }
};

template <typename T, int E, int D>
struct Kernels {
  static auto pick(bool jac) { return jac ? gp_step_kernel<T, E, D, true> : gp_step_kernel<T, E, D, false>; }
};
template <int E, int D>
struct Kernels<float, E, D> {
  static auto pick(bool jac) {
    return jac ? gp_step_kernel<float, E, D, true> : gp_step_fwd_f32_kernel<E, D>;
  }
};

template <typename T, int E, int D>
int launch(const Args<T>& a, bool jac, hipStream_t st) {
  const int K = a.n + a.m_act;
  if (K > 64) return PDDP_E_UNSUPPORTED;
  const Lds<E, D> o = lds_of<E, D>(a.M, K, jac, (int)sizeof(T));
  const size_t bytes = (size_t)o.total * sizeof(T);
  if (bytes > 160 * 1024) return PDDP_E_UNSUPPORTED;
  auto kern = Kernels<T, E, D>::pick(jac);
  if (bytes > 64 * 1024) {
    const hipError_t e =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(a.R), dim3(kThreads), bytes, st, a);
  return (int)hipGetLastError();
}

template <typename T, int E, int D>
int launch_roll(Args<T> a, Roll<T> r, hipStream_t st) {
  if (a.n + a.m_act > 64 || r.na > 8 || r.na != a.n_non + 2 * a.n_ang || a.m_act > kMaxAct)
    return PDDP_E_UNSUPPORTED;
  const Lds<E, D> o(a.M, a.n + a.m_act, false);
  const size_t bytes = (size_t)o.total * sizeof(T);
  if (bytes > 160 * 1024) return PDDP_E_UNSUPPORTED;
  auto kern = RollKernels<T, E, D>::pick();
  if (bytes > 64 * 1024) {
    const hipError_t e =
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  a.R = r.B * r.A;
  // N steps and the terminal cost: N + 1 launches, nothing between them
  for (int t = 0; t <= r.N; ++t) {
    r.t = t;
    r.terminal = t == r.N ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(a.R), dim3(kThreads), bytes, st, a, r);
  }
  return (int)hipGetLastError();
}

template <typename T>
int fill_model(const pddp_gp_model* g, Args<T>& a) {
  if (g == nullptr) return PDDP_E_BADARG;
  if (!g->Xt || !g->Xt_pairs || !g->beta || !g->beta_pairs || !g->Kinv || !g->inv_ell2 || !g->sf2 || !g->sn2)
    return PDDP_E_BADARG;
  if (g->n_ang < 0 || g->n_ang > kMaxAng || g->n_non < 0 || g->n_non > kMaxNon) return PDDP_E_BADARG;
  if (g->n_ang + g->n_non != g->state_size || g->M < 1 || g->action_size < 1) return PDDP_E_BADARG;
  const int E = g->state_size;
  a.M = g->M;
  a.m_act = g->action_size;
  a.n_ang = g->n_ang;
  a.n_non = g->n_non;
  a.encoding = g->encoding;
  switch (g->encoding) {
    case 1: a.n = E + E * (E + 1) / 2; break;
    case 2:
    case 3: a.n = 2 * E; break;
    case 4: a.n = E; break;
    default: return PDDP_E_UNSUPPORTED;  // FULL_COVARIANCE_MATRIX: the torch path
  }
  for (int i = 0; i < kMaxAng; ++i) a.ang[i] = i < g->n_ang ? g->ang[i] : 0;
  for (int i = 0; i < kMaxNon; ++i) a.non[i] = i < g->n_non ? g->non[i] : 0;
  for (int i = 0; i < g->n_ang; ++i)
    if (g->ang[i] < 0 || g->ang[i] >= E) return PDDP_E_BADARG;
  for (int i = 0; i < g->n_non; ++i)
    if (g->non[i] < 0 || g->non[i] >= E) return PDDP_E_BADARG;
  a.Xt = (const T*)g->Xt;
  a.XtP = (const T*)g->Xt_pairs;
  a.betaP = (const T*)g->beta_pairs;
  a.beta = (const T*)g->beta;
  a.Kinv = (const T*)g->Kinv;
  a.iL = (const T*)g->inv_ell2;
  a.sf2 = (const T*)g->sf2;
  a.sn2 = (const T*)g->sn2;
  a.z = nullptr; a.u = nullptr; a.z_next = nullptr; a.Fz = nullptr; a.Fu = nullptr;
  a.row_mask = nullptr; a.rows_per_mask = 1;
  return 0;
}

template <typename T>
int rollout(const pddp_gp_model* g, const pddp_gp_rollout* q, void* stream) {
  if (q == nullptr || q->B <= 0 || q->N <= 0 || q->A <= 0 || !q->Z || !q->U || !q->gains ||
      !q->alphas || !q->Zc || !q->Uc || !q->Jc || !q->Q || !q->Q_term || !q->R || !q->x_goal ||
      !q->u_goal || ((q->u_min == nullptr) != (q->u_max == nullptr)))
    return PDDP_E_BADARG;
  Args<T> a;
  if (int rc = fill_model<T>(g, a)) return rc;
  Roll<T> r;
  r.B = q->B; r.N = q->N; r.A = q->A; r.t = 0; r.terminal = 0;
  r.na = g->n_non + 2 * g->n_ang;
  r.Z = (const T*)q->Z; r.U = (const T*)q->U; r.gains = (const T*)q->gains;
  r.alphas = (const T*)q->alphas; r.u_min = (const T*)q->u_min; r.u_max = (const T*)q->u_max;
  r.active = q->active; r.status = q->bwd_status;
  r.Zc = (T*)q->Zc; r.Uc = (T*)q->Uc; r.Jc = (T*)q->Jc;
  r.Q = (const T*)q->Q; r.Qt = (const T*)q->Q_term; r.Rm = (const T*)q->R;
  r.xg = (const T*)q->x_goal; r.ug = (const T*)q->u_goal;
  const int E = g->state_size, D = g->n_non + 2 * g->n_ang + g->action_size;
  hipStream_t st = (hipStream_t)stream;
  if (E == 2 && D == 4) return launch_roll<T, 2, 4>(a, r, st);
  if (E == 4 && D == 6) return launch_roll<T, 4, 6>(a, r, st);
  if (E == 6 && D == 9) return launch_roll<T, 6, 9>(a, r, st);
  return PDDP_E_UNSUPPORTED;
}

template <typename T>
int step(const pddp_gp_model* g, int R, const T* z, const T* u, T* z_next, T* Fz, T* Fu, void* stream,
         const uint8_t* row_mask = nullptr, int rows_per_mask = 1) {
  if (g == nullptr || R < 0 || z == nullptr || u == nullptr || z_next == nullptr) return PDDP_E_BADARG;
  if (!g->Xt || !g->Xt_pairs || !g->beta || !g->beta_pairs || !g->Kinv || !g->inv_ell2 || !g->sf2 || !g->sn2)
    return PDDP_E_BADARG;
  if ((Fz == nullptr) != (Fu == nullptr)) return PDDP_E_BADARG;
  if (row_mask != nullptr && rows_per_mask < 1) return PDDP_E_BADARG;
  if (R == 0) return 0;
  if (g->n_ang < 0 || g->n_ang > kMaxAng || g->n_non < 0 || g->n_non > kMaxNon) return PDDP_E_BADARG;
  if (g->n_ang + g->n_non != g->state_size || g->M < 1 || g->action_size < 1) return PDDP_E_BADARG;
  const int E = g->state_size, D = g->n_non + 2 * g->n_ang + g->action_size;
  Args<T> a;
  a.R = R;
  a.M = g->M;
  a.m_act = g->action_size;
  a.n_ang = g->n_ang;
  a.n_non = g->n_non;
  a.encoding = g->encoding;
  switch (g->encoding) {
    case 1: a.n = E + E * (E + 1) / 2; break;
    case 2:
    case 3: a.n = 2 * E; break;
    case 4: a.n = E; break;
    default: return PDDP_E_UNSUPPORTED;  // FULL_COVARIANCE_MATRIX: the torch path
  }
  for (int i = 0; i < kMaxAng; ++i) a.ang[i] = i < g->n_ang ? g->ang[i] : 0;
  for (int i = 0; i < kMaxNon; ++i) a.non[i] = i < g->n_non ? g->non[i] : 0;
  for (int i = 0; i < g->n_ang; ++i)
    if (g->ang[i] < 0 || g->ang[i] >= E) return PDDP_E_BADARG;
  for (int i = 0; i < g->n_non; ++i)
    if (g->non[i] < 0 || g->non[i] >= E) return PDDP_E_BADARG;
  a.Xt = (const T*)g->Xt;
  a.XtP = (const T*)g->Xt_pairs;
  a.betaP = (const T*)g->beta_pairs;
  a.beta = (const T*)g->beta;
  a.Kinv = (const T*)g->Kinv;
  a.iL = (const T*)g->inv_ell2;
  a.sf2 = (const T*)g->sf2;
  a.sn2 = (const T*)g->sn2;
  a.z = z;
  a.u = u;
  a.z_next = z_next;
  a.Fz = Fz;
  a.Fu = Fu;
  a.row_mask = row_mask;
  a.rows_per_mask = rows_per_mask;
  const bool jac = Fz != nullptr;
  hipStream_t st = (hipStream_t)stream;
  // the systems of the reference's examples: pendulum, cartpole, double cartpole
  if (E == 2 && D == 4) return launch<T, 2, 4>(a, jac, st);
  if (E == 4 && D == 6) return launch<T, 4, 6>(a, jac, st);
  if (E == 6 && D == 9) return launch<T, 6, 9>(a, jac, st);
  return PDDP_E_UNSUPPORTED;
}

}  // namespace gp
}  // namespace pddp

extern "C" {
#ifdef PDDP_GP_MARKS
int pddp_debug_gp_marks(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::gp::g_gp_marks), sizeof(long long) * 64);
}
#endif
long long pddp_gp_step_lds_bytes(int state_size, int d, int M, int inputs, int jacobian,
                                 int element_size) {
  using namespace pddp::gp;
  long long words = -1;
  if (state_size == 2 && d == 4) words = lds_of<2, 4>(M, inputs, jacobian != 0, element_size).total;
  if (state_size == 4 && d == 6) words = lds_of<4, 6>(M, inputs, jacobian != 0, element_size).total;
  if (state_size == 6 && d == 9) words = lds_of<6, 9>(M, inputs, jacobian != 0, element_size).total;
  return words < 0 ? -1 : words * element_size;
}
int pddp_gp_step_f32(const pddp_gp_model* g, int R, const float* z, const float* u, float* z_next,
                     float* Fz, float* Fu, void* stream) {
  return pddp::gp::step<float>(g, R, z, u, z_next, Fz, Fu, stream);
}
int pddp_gp_step_f64(const pddp_gp_model* g, int R, const double* z, const double* u, double* z_next,
                     double* Fz, double* Fu, void* stream) {
  return pddp::gp::step<double>(g, R, z, u, z_next, Fz, Fu, stream);
}
int pddp_gp_step_masked_f32(const pddp_gp_model* g, int R, const float* z, const float* u,
                            float* z_next, float* Fz, float* Fu, const uint8_t* row_mask,
                            int rows_per_mask, void* stream) {
  return pddp::gp::step<float>(g, R, z, u, z_next, Fz, Fu, stream, row_mask, rows_per_mask);
}
int pddp_gp_step_masked_f64(const pddp_gp_model* g, int R, const double* z, const double* u,
                            double* z_next, double* Fz, double* Fu, const uint8_t* row_mask,
                            int rows_per_mask, void* stream) {
  return pddp::gp::step<double>(g, R, z, u, z_next, Fz, Fu, stream, row_mask, rows_per_mask);
}
int pddp_gp_rollout_f32(const pddp_gp_model* g, const pddp_gp_rollout* r, void* stream) {
  return pddp::gp::rollout<float>(g, r, stream);
}
int pddp_gp_rollout_f64(const pddp_gp_model* g, const pddp_gp_rollout* r, void* stream) {
  return pddp::gp::rollout<double>(g, r, stream);
}
}
