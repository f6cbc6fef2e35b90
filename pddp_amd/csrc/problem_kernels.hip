// problem_kernels.hip - kernels that evaluate the sample problems' dynamics and
// cost: nominal rollout, derivative records, and the batched line search.
//
//   nominal rollout   pddp/controllers/ilqr.py:457-468   (sequential in t)
//   derivative records ilqr.py:464-473 via analytic Jacobians / Hessians
//                      (parallel over trajectory AND time step)
//   line search       ilqr.py:677-723 _control_law + :764-791 _trajectory_cost
#include <type_traits>
#include "models.hpp"
#include "problem_args.hpp"
#include "accept.hpp"
#include "riccati_n4.hpp"  // DPP helpers of the 16-lane groups
#include "line_search_lds.hpp"

namespace pddp {

// --------------------------------------------------------------------------
// nominal rollout: one lane per trajectory
// --------------------------------------------------------------------------

template <typename T, int MODEL>
__global__ __launch_bounds__(kWave) void nominal_rollout_kernel(
    ProblemT<T> P, RolloutArgs<T> a) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  if (a.mask != nullptr && a.mask[b] == 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  T z[n], zn[n], u[m], umin[m], umax[m];
#pragma unroll
  for (int r = 0; r < m; ++r) {
    umin[r] = bounded ? a.u_min[r] : T(0);
    umax[r] = bounded ? a.u_max[r] : T(0);
  }
  T* Zb = a.Z + (size_t)b * (a.N + 1) * n;
  const T* Ub = a.U + (size_t)b * a.N * m;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    z[j] = a.z0[(size_t)b * n + j];
    Zb[j] = z[j];
  }
  for (int t = 0; t < a.N; ++t) {
#pragma unroll
    for (int j = 0; j < m; ++j) {
      u[j] = Ub[t * m + j];
      if (bounded) u[j] = clamp1(u[j], umin[j], umax[j]);
    }
    const Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
    dynamics<T, MODEL, false>(P, z, u, tr, zn, nullptr, nullptr);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      z[j] = zn[j];
      Zb[(t + 1) * n + j] = z[j];
    }
  }
}

// --------------------------------------------------------------------------
// derivative records: one workgroup per trajectory, one lane per time step;
// records are staged through LDS so the HBM writes are fully coalesced.
// --------------------------------------------------------------------------
// (group_copy, store4: line_search_lds.hpp)
// (record_of: models.hpp)

constexpr int kDerivThreads = 64;

template <typename T, int MODEL>
__global__ __launch_bounds__(kDerivThreads) void derivs_kernel(
    ProblemT<T> P, DerivArgs<T> a) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr RecLayout lay(n, m);
  constexpr int S = lay.stride;
  constexpr int LD = kDerivThreads + 1;  // +1: conflict-free transposed reads
  __shared__ T stage[S * LD];
  __shared__ T Lsum[kDerivThreads];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (a.mask != nullptr && a.mask[b] == 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  const int N = a.N;
  const T* Zb = a.Z + (size_t)b * (N + 1) * n;
  const T* Ub = a.U + (size_t)b * N * m;
  T* rec_b = a.rec + (size_t)b * (N + 1) * S;
  T Jacc = T(0);  // only meaningful in lane 0

  for (int t0 = 0; t0 <= N; t0 += kDerivThreads) {
    const int t = t0 + tid;
    T l = T(0);
    if (t <= N) {
      T z[n], un[m], w[S];
#pragma unroll
      for (int j = 0; j < n; ++j) z[j] = Zb[t * n + j];
      const bool terminal = (t == N);
#pragma unroll
      for (int j = 0; j < m; ++j) un[j] = terminal ? T(0) : Ub[t * m + j];
      l = record_of<T, MODEL>(P, z, un, terminal, bounded, a.u_min, a.u_max, w);
      T* col = stage + tid;
#pragma unroll
      for (int j = 0; j < S; ++j) col[j * LD] = w[j];
      a.L[(size_t)b * (N + 1) + t] = l;
    }
    Lsum[tid] = l;
    __syncthreads();
    // coalesced write-out of this chunk's records
    const int nrec = min(kDerivThreads, N + 1 - t0);
    T* dst = rec_b + (size_t)t0 * S;
    for (int o = tid; o < nrec * S; o += kDerivThreads) {
      const int r = o / S, w = o - r * S;
      dst[o] = stage[w * LD + r];
    }
    if (tid == 0)
      for (int r = 0; r < nrec; ++r) Jacc += Lsum[r];  // L.sum(), in t order
    __syncthreads();
  }
  if (tid == 0) {
    a.J[b] = Jacc;
    if (a.state != nullptr) a.state[b] = PDDP_STATE_UNDEFINED;
  }
}

// --------------------------------------------------------------------------
// line search: one lane per (trajectory, alpha) candidate
// --------------------------------------------------------------------------

template <typename T, int MODEL>
__global__ __launch_bounds__(kWave) void line_search_kernel(
    ProblemT<T> P, LineSearchArgs<T> a) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr int GS = m + m * n;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = a.B * a.A;
  if (idx >= total) return;
  const int b = idx / a.A, ai = idx - b * a.A;
  if (a.active != nullptr && a.active[b] == 0) return;
  if (a.bwd_status != nullptr && a.bwd_status[b] != 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  T umin[m], umax[m];
#pragma unroll
  for (int r = 0; r < m; ++r) {
    umin[r] = bounded ? a.u_min[r] : T(0);
    umax[r] = bounded ? a.u_max[r] : T(0);
  }
  const int N = a.N;
  const T alpha = a.alphas[ai];
  const T* Zb = a.Z + (size_t)b * (N + 1) * n;
  const T* Ub = a.U + (size_t)b * N * m;
  const T* Gb = a.gains + (size_t)b * N * GS;

  T z[n], zn[n], un[m];
  T zr[n], ur[m], gr[GS];  // this step's nominal z, u and gains
#pragma unroll
  for (int j = 0; j < n; ++j) {
    zr[j] = Zb[j];
    z[j] = zr[j];  // Z_new[0] = Z[0]                             (ilqr.py:690)
  }
#pragma unroll
  for (int j = 0; j < m; ++j) ur[j] = Ub[j];
#pragma unroll
  for (int j = 0; j < GS; ++j) gr[j] = Gb[j];

  // time-major output [b][t][alpha][.]: at every step the A lanes of a
  // trajectory write one contiguous A*n-word segment (see the note at
  // LineSearchArgs)
  T* Zci = a.Zc + ((size_t)b * (N + 1) * a.A + ai) * n;
  T* Uci = a.Uc + ((size_t)b * N * a.A + ai) * m;
  const size_t zstep = (size_t)a.A * n, ustep = (size_t)a.A * m;
  T J = T(0);
  for (int t = 0; t < N; ++t) {
    // prefetch the next step's nominal data before the dependent chain
    T zr2[n], ur2[m], gr2[GS];
    const int tn = (t + 1 < N) ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < n; ++j) zr2[j] = Zb[tn * n + j];
#pragma unroll
    for (int j = 0; j < m; ++j) ur2[j] = Ub[tn * m + j];
#pragma unroll
    for (int j = 0; j < GS; ++j) gr2[j] = Gb[tn * GS + j];

#pragma unroll
    for (int r = 0; r < m; ++r) {
      T du = alpha * gr[r];  // alpha * k[i]                      (ilqr.py:708)
      T s = T(0);
#pragma unroll
      for (int c = 0; c < n; ++c) s += (z[c] - zr[c]) * gr[m + r * n + c];
      du = du + s;  // + dz K^T                                   (ilqr.py:710)
      T v = ur[r] + du;
      un[r] = bounded ? clamp_nan(v, umin[r], umax[r]) : v;
    }
#pragma unroll
    for (int j = 0; j < n; ++j) Zci[(size_t)t * zstep + j] = z[j];
#pragma unroll
    for (int j = 0; j < m; ++j) Uci[(size_t)t * ustep + j] = un[j];
    const Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
    J += cost_value<T, MODEL>(P, z, un, tr, false);
    dynamics<T, MODEL, false>(P, z, un, tr, zn, nullptr, nullptr);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      z[j] = zn[j];
      zr[j] = zr2[j];
    }
#pragma unroll
    for (int j = 0; j < m; ++j) ur[j] = ur2[j];
#pragma unroll
    for (int j = 0; j < GS; ++j) gr[j] = gr2[j];
  }
#pragma unroll
  for (int j = 0; j < n; ++j) Zci[(size_t)N * zstep + j] = z[j];
  const T lf = cost_value<T, MODEL>(P, z, nullptr, trig_of<T, MODEL>(z), true);
  a.Jc[idx] = J + lf;  // L.sum(0) + l_f                           (ilqr.py:789)
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
// The one sparse stage-cost pattern instantiated per model: the shipped
// example cost's (cartpole: {x, sin, cos}); a cost matrix with entries outside
// it runs the full form.  Models without a sparse instantiation: the full mask
// (the dispatch below then folds to one launch).
template <int MODEL>
constexpr unsigned kSparseMask =
    MODEL == PDDP_MODEL_CARTPOLE ? 0b11001u : kFullMask<MODEL>;
template <int MODEL, unsigned QM>
static bool stage_cost_on(const pddp_problem& p) {
  if (QM == kFullMask<MODEL>) return false;
  return (live_mask(p.Q, ModelDims<MODEL>::na) & ~QM) == 0;
}
template <typename T, int MODEL>
static int launch_rollout(const pddp_problem& p, RolloutArgs<T> a,
                          hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  const int blocks = (a.B + kWave - 1) / kWave;
  PDDP_LAUNCH((nominal_rollout_kernel<T, MODEL>), dim3(blocks),
                     dim3(kWave), 0, st, P, a);
  return launch_status();
}
template <typename T, int MODEL>
static int launch_derivs(const pddp_problem& p, DerivArgs<T> a, hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  PDDP_LAUNCH((derivs_kernel<T, MODEL>), dim3(a.B), dim3(kDerivThreads),
                     0, st, P, a);
  return launch_status();
}
template <typename T, int MODEL>
static int launch_line_search(const pddp_problem& p, LineSearchArgs<T> a,
                              hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  using D = ModelDims<MODEL>;
  const size_t per = (size_t)(a.N + 1) * D::n + (size_t)a.N * D::m +
                     (size_t)a.N * (D::m + D::m * D::n);
  const size_t lds = 4 * per * sizeof(T);
  if (a.A <= 16 && lds <= 64 * 1024) {  // nominal data staged in LDS
    if (4 * lds <= 64 * 1024) {
      if (stage_cost_on<MODEL, kSparseMask<MODEL>>(p))
        PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, false, 4, 1,
                                            kSparseMask<MODEL>>),
                           dim3((a.B + 15) / 16), dim3(kWave * 4), 4 * lds, st,
                           P, a, AcceptArgs<T>{}, (T*)nullptr, (T*)nullptr);
      else
        PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, false, 4>),
                           dim3((a.B + 15) / 16), dim3(kWave * 4), 4 * lds, st,
                           P, a, AcceptArgs<T>{}, (T*)nullptr, (T*)nullptr);
    } else
      PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, false, 1>),
                         dim3((a.B + 3) / 4), dim3(kWave), lds, st, P, a,
                         AcceptArgs<T>{}, (T*)nullptr, (T*)nullptr);
    return launch_status();
  }
  const int total = a.B * a.A;
  const int blocks = (total + kWave - 1) / kWave;
  PDDP_LAUNCH((line_search_kernel<T, MODEL>), dim3(blocks), dim3(kWave),
                     0, st, P, a);
  return launch_status();
}

// fused line search + accept + derivative records; PDDP_E_UNSUPPORTED when the
// LDS kernel does not apply (more than 16 step sizes, nominal data > 64 KB)
template <typename T>
struct SearchAcceptArgs {
  LineSearchArgs<T> ls;
  AcceptArgs<T> ac;
  T* rec;
  T* L;
};
// 0 auto (by batch), 1 the paired form always, 2 the dense form where built
inline int& search_form_choice() {
  static int choice = 0;
  return choice;
}
template <typename T, int MODEL>
static int launch_search_accept(const pddp_problem& p, SearchAcceptArgs<T> a,
                                hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  using D = ModelDims<MODEL>;
  const size_t per = (size_t)(a.ls.N + 1) * D::n + (size_t)a.ls.N * D::m +
                     (size_t)a.ls.N * (D::m + D::m * D::n);
  const size_t lds = 4 * per * sizeof(T);
  if (a.ls.A > 16 || lds > 64 * 1024) return PDDP_E_UNSUPPORTED;
  a.ac.n = D::n;
  a.ac.m = D::m;
  if constexpr (sizeof(T) == 4 && D::n <= 4) {
    // the dense form (see the kernel): from the batch on that the paired
    // form's two workgroups per CU no longer hold at once
    const int mode = search_form_choice();
    const size_t lds_dense =
        16 * (size_t)a.ls.N * (D::m + D::m * D::n) * sizeof(T);
    // (measured, tools/dbg/search_form_scan.py: 100.5 -> 90.7 us at 12288
    // trajectories, 117.8 -> 105.5 at 16384, 206.7 -> 192.1 at 32768; slower
    // at 8192 and below - 50.5 -> 60.3 - and at 65536 - 406 -> 483)
    const bool dense =
        mode == 2 || (mode == 0 && a.ls.B > 8192 && a.ls.B <= 49152);
    if (dense && 4 * lds <= 64 * 1024 && lds_dense <= 40 * 1024 &&
        a.ls.N + 1 <= 128) {
      if (stage_cost_on<MODEL, kSparseMask<MODEL>>(p))
        PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, true, 4, 1,
                                            kSparseMask<MODEL>, true>),
                    dim3((a.ls.B + 15) / 16), dim3(kWave * 4), lds_dense, st,
                    P, a.ls, a.ac, a.rec, a.L);
      else
        PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, true, 4, 1,
                                            kFullMask<MODEL>, true>),
                    dim3((a.ls.B + 15) / 16), dim3(kWave * 4), lds_dense, st,
                    P, a.ls, a.ac, a.rec, a.L);
      return launch_status();
    }
  }
  if (4 * lds <= 64 * 1024) {
    // four rollout waves (one per SIMD of a CU) + their four helpers
    if (stage_cost_on<MODEL, kSparseMask<MODEL>>(p))
      PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, true, 4, 2,
                                          kSparseMask<MODEL>>),
                         dim3((a.ls.B + 15) / 16), dim3(kWave * 8), 4 * lds, st,
                         P, a.ls, a.ac, a.rec, a.L);
    else
      PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, true, 4, 2>),
                         dim3((a.ls.B + 15) / 16), dim3(kWave * 8), 4 * lds, st,
                         P, a.ls, a.ac, a.rec, a.L);
  } else
    PDDP_LAUNCH((line_search_lds_kernel<T, MODEL, true, 1>),
                       dim3((a.ls.B + 3) / 4), dim3(kWave), lds, st, P, a.ls,
                       a.ac, a.rec, a.L);
  return launch_status();
}

// default_kernels.hip: the same three operations under the DEFAULT (upper-
// triangular Cholesky) encoding, selected by pddp_problem.encoding
template <typename T>
int default_rollout(const pddp_problem& p, RolloutArgs<T> a, hipStream_t st);
template <typename T>
int default_derivs(const pddp_problem& p, DerivArgs<T> a, hipStream_t st);
template <typename T>
int default_line_search(const pddp_problem& p, LineSearchArgs<T> a,
                        hipStream_t st);
static bool is_default_encoding(const pddp_problem* p) {
  return p != nullptr && (p->encoding == PDDP_ENC_UPPER_TRIANGULAR_CHOLESKY ||
                          p->encoding == PDDP_ENC_VARIANCE_ONLY ||
                          p->encoding == PDDP_ENC_STANDARD_DEVIATION_ONLY ||
                          p->encoding == PDDP_ENC_FULL_COVARIANCE_MATRIX);
}

static int check_problem(const pddp_problem* p) {
  if (p == nullptr) return PDDP_E_BADARG;
  if (p->encoding != PDDP_ENC_IGNORE_UNCERTAINTY) return PDDP_E_UNSUPPORTED;
  switch (p->model) {
    case PDDP_MODEL_CARTPOLE:
    case PDDP_MODEL_DOUBLE_CARTPOLE:
    case PDDP_MODEL_PENDULUM:
    case PDDP_MODEL_RENDEZVOUS:
      return 0;
  }
  return PDDP_E_UNSUPPORTED;
}

#define PDDP_DISPATCH_MODEL(FN, T, p, args, st)                                \
  switch ((p)->model) {                                                        \
    case PDDP_MODEL_CARTPOLE:                                                  \
      return FN<T, PDDP_MODEL_CARTPOLE>(*(p), args, st);                       \
    case PDDP_MODEL_DOUBLE_CARTPOLE:                                           \
      return FN<T, PDDP_MODEL_DOUBLE_CARTPOLE>(*(p), args, st);                \
    case PDDP_MODEL_PENDULUM:                                                  \
      return FN<T, PDDP_MODEL_PENDULUM>(*(p), args, st);                       \
    default:                                                                   \
      return FN<T, PDDP_MODEL_RENDEZVOUS>(*(p), args, st);                     \
  }

template <typename T>
static int nominal_rollout_impl(const pddp_problem* p, int B, int N,
                                const T* z0, const T* U, const T* u_min,
                                const T* u_max, const uint8_t* mask, T* Z,
                                void* stream) {
  if (B <= 0 || N <= 0 || !z0 || !U || !Z) return PDDP_E_BADARG;
  RolloutArgs<T> a{B, N, z0, U, u_min, u_max, mask, Z};
  if (is_default_encoding(p))
    return default_rollout<T>(*p, a, (hipStream_t)stream);
  if (int rc = check_problem(p)) return rc;
  PDDP_DISPATCH_MODEL(launch_rollout, T, p, a, (hipStream_t)stream)
}

template <typename T>
static int derivs_impl(const pddp_problem* p, int B, int N, const T* Z,
                       const T* U, const T* u_min, const T* u_max,
                       const uint8_t* mask, T* rec, T* L, T* J, int32_t* state,
                       void* stream) {
  if (B <= 0 || N <= 0 || !Z || !U || !rec || !L || !J) return PDDP_E_BADARG;
  DerivArgs<T> a{B, N, Z, U, u_min, u_max, mask, rec, L, J, state};
  if (is_default_encoding(p))
    return default_derivs<T>(*p, a, (hipStream_t)stream);
  if (int rc = check_problem(p)) return rc;
  PDDP_DISPATCH_MODEL(launch_derivs, T, p, a, (hipStream_t)stream)
}

template <typename T>
static int line_search_impl(const pddp_problem* p, int B, int N, int A,
                            const T* Z, const T* U, const T* gains,
                            const T* alphas, const T* u_min, const T* u_max,
                            const uint8_t* active, const int32_t* bwd_status,
                            T* Zc, T* Uc, T* Jc, void* stream) {
  if (B <= 0 || N <= 0 || A <= 0 || !Z || !U || !gains || !alphas || !Zc ||
      !Uc || !Jc)
    return PDDP_E_BADARG;
  LineSearchArgs<T> a{B, N, A, Z, U, gains, alphas, u_min, u_max, active,
                      bwd_status, Zc, Uc, Jc};
  if (is_default_encoding(p))
    return default_line_search<T>(*p, a, (hipStream_t)stream);
  if (int rc = check_problem(p)) return rc;
  PDDP_DISPATCH_MODEL(launch_line_search, T, p, a, (hipStream_t)stream)
}

// 0 auto (by the size of the candidates), 1 keep them, 2 drop them
inline int& search_candidates_choice() {
  static int choice = 0;
  return choice;
}

template <typename T>
static int search_accept_impl(const pddp_problem* p, int B, int N, int A, T* Z,
                              T* U, const T* gains, const T* alphas,
                              const T* u_min, const T* u_max, uint8_t* active,
                              const int32_t* bwd_status, T* Zc, T* Uc, T* Jc,
                              double tol, double max_reg, int n_iterations,
                              T* gains_acc, T* J_opt, double* mu,
                              double* delta, int32_t* state, int32_t* iter,
                              uint8_t* fresh, int32_t* n_live, T* rec, T* L,
                              void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (B <= 0 || N <= 0 || A <= 0 || !Z || !U || !gains || !alphas || !active ||
      !bwd_status || !Zc || !Uc || !Jc || !gains_acc || !J_opt || !mu ||
      !delta || !state || !iter || !fresh || (L != nullptr && !rec))
    return PDDP_E_BADARG;
  SearchAcceptArgs<T> a;
  a.ls = LineSearchArgs<T>{B, N, A, Z, U, gains, alphas, u_min, u_max, active,
                           bwd_status, Zc, Uc, Jc};
  a.ac = AcceptArgs<T>{B, N, 0, 0, A, Zc, Uc, Jc, gains, bwd_status, tol,
                       max_reg, n_iterations, Z, U, gains_acc, J_opt, mu, delta,
                       state, iter, active, fresh, n_live};
  a.rec = rec;
  a.L = L;
  // keep the candidates while they fit the Infinity Cache next to the rest of
  // the round's traffic (256 MB; a launch's candidates: 82 MB at B = 4096,
  // 164 MB at 8192: 50 us; 328 MB at 16384: 163 us kept, 97 us dropped)
  const double cand_bytes =
      (double)B * A * ((double)(N + 1) * p->encoded_size +
                       (double)N * p->action_size) * sizeof(T);
  const int mode = search_candidates_choice();
  a.ls.drop_candidates =
      (L == nullptr && rec != nullptr &&
       (mode == 2 || (mode == 0 && cand_bytes > 200e6))) ? 1 : 0;
  PDDP_DISPATCH_MODEL(launch_search_accept, T, p, a, (hipStream_t)stream)
}

}  // namespace pddp

extern "C" {

#ifdef PDDP_WG_TIMELINE
int pddp_debug_search_timeline(long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::g_search_timeline),
                      sizeof(long long) * 1024 * 12);
  return 0;
}
#endif

int pddp_search_form(int mode) {
  const int prev = pddp::search_form_choice();
  if (mode >= 0 && mode <= 2) pddp::search_form_choice() = mode;
  return prev;
}
int pddp_search_candidates(int mode) {
  const int prev = pddp::search_candidates_choice();
  if (mode >= 0 && mode <= 2) pddp::search_candidates_choice() = mode;
  return prev;
}

int pddp_search_accept_f32(const pddp_problem* p, int B, int N, int A, float* Z,
                           float* U, const float* gains, const float* alphas,
                           const float* u_min, const float* u_max,
                           uint8_t* active, const int32_t* bwd_status,
                           float* Zc, float* Uc, float* Jc, double tol,
                           double max_reg, int n_iterations, float* gains_acc,
                           float* J_opt, double* mu, double* delta,
                           int32_t* state, int32_t* iter, uint8_t* fresh,
                           int32_t* n_live, float* rec, float* L,
                           void* stream) {
  return pddp::search_accept_impl<float>(
      p, B, N, A, Z, U, gains, alphas, u_min, u_max, active, bwd_status, Zc, Uc,
      Jc, tol, max_reg, n_iterations, gains_acc, J_opt, mu, delta, state, iter,
      fresh, n_live, rec, L, stream);
}
int pddp_search_accept_f64(const pddp_problem* p, int B, int N, int A,
                           double* Z, double* U, const double* gains,
                           const double* alphas, const double* u_min,
                           const double* u_max, uint8_t* active,
                           const int32_t* bwd_status, double* Zc, double* Uc,
                           double* Jc, double tol, double max_reg,
                           int n_iterations, double* gains_acc, double* J_opt,
                           double* mu, double* delta, int32_t* state,
                           int32_t* iter, uint8_t* fresh, int32_t* n_live,
                           double* rec, double* L, void* stream) {
  return pddp::search_accept_impl<double>(
      p, B, N, A, Z, U, gains, alphas, u_min, u_max, active, bwd_status, Zc, Uc,
      Jc, tol, max_reg, n_iterations, gains_acc, J_opt, mu, delta, state, iter,
      fresh, n_live, rec, L, stream);
}

int pddp_nominal_rollout_f32(const pddp_problem* p, int B, int N,
                             const float* z0, const float* U,
                             const float* u_min, const float* u_max,
                             const uint8_t* mask, float* Z, void* stream) {
  return pddp::nominal_rollout_impl<float>(p, B, N, z0, U, u_min, u_max, mask,
                                           Z, stream);
}
int pddp_nominal_rollout_f64(const pddp_problem* p, int B, int N,
                             const double* z0, const double* U,
                             const double* u_min, const double* u_max,
                             const uint8_t* mask, double* Z, void* stream) {
  return pddp::nominal_rollout_impl<double>(p, B, N, z0, U, u_min, u_max, mask,
                                            Z, stream);
}
int pddp_derivs_f32(const pddp_problem* p, int B, int N, const float* Z,
                    const float* U, const float* u_min, const float* u_max,
                    const uint8_t* mask, float* rec, float* L, float* J,
                    int32_t* state, void* stream) {
  return pddp::derivs_impl<float>(p, B, N, Z, U, u_min, u_max, mask, rec, L, J,
                                  state, stream);
}
int pddp_derivs_f64(const pddp_problem* p, int B, int N, const double* Z,
                    const double* U, const double* u_min, const double* u_max,
                    const uint8_t* mask, double* rec, double* L, double* J,
                    int32_t* state, void* stream) {
  return pddp::derivs_impl<double>(p, B, N, Z, U, u_min, u_max, mask, rec, L,
                                   J, state, stream);
}
int pddp_line_search_f32(const pddp_problem* p, int B, int N, int A,
                         const float* Z, const float* U, const float* gains,
                         const float* alphas, const float* u_min,
                         const float* u_max, const uint8_t* active,
                         const int32_t* bwd_status, float* Zc, float* Uc,
                         float* Jc, void* stream) {
  return pddp::line_search_impl<float>(p, B, N, A, Z, U, gains, alphas, u_min,
                                       u_max, active, bwd_status, Zc, Uc, Jc,
                                       stream);
}
int pddp_line_search_f64(const pddp_problem* p, int B, int N, int A,
                         const double* Z, const double* U, const double* gains,
                         const double* alphas, const double* u_min,
                         const double* u_max, const uint8_t* active,
                         const int32_t* bwd_status, double* Zc, double* Uc,
                         double* Jc, void* stream) {
  return pddp::line_search_impl<double>(p, B, N, A, Z, U, gains, alphas, u_min,
                                        u_max, active, bwd_status, Zc, Uc, Jc,
                                        stream);
}

}  // extern "C"
