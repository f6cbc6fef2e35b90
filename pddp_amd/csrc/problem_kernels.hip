// problem_kernels.hip - kernels that evaluate the sample problems' dynamics and
// cost: nominal rollout, derivative records, and the batched line search.
//
//   nominal rollout   pddp/controllers/ilqr.py:457-468   (sequential in t)
//   derivative records ilqr.py:464-473 via analytic Jacobians / Hessians
//                      (parallel over trajectory AND time step)
//   line search       ilqr.py:677-723 _control_law + :764-791 _trajectory_cost
#include "models.hpp"

namespace pddp {

// --------------------------------------------------------------------------
// nominal rollout: one lane per trajectory
// --------------------------------------------------------------------------
template <typename T>
struct RolloutArgs {
  int B, N;
  const T* z0;
  const T* U;
  const T* u_min;
  const T* u_max;
  const uint8_t* mask;
  T* Z;
};

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
template <typename T>
struct DerivArgs {
  int B, N;
  const T* Z;
  const T* U;
  const T* u_min;
  const T* u_max;
  const uint8_t* mask;
  T* rec;
  T* L;
  T* J;
  int32_t* state;
};

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
      T z[n], u[m], un[m], zn[n];
      T Fz[n * n], Fu[n * m], lz[n], lzz[n * n], lu[m], luu[m * m];
#pragma unroll
      for (int j = 0; j < n; ++j) z[j] = Zb[t * n + j];
      const bool terminal = (t == N);
#pragma unroll
      for (int j = 0; j < m; ++j) {
        un[j] = terminal ? T(0) : Ub[t * m + j];
        u[j] = bounded ? clamp1(un[j], a.u_min[j], a.u_max[j]) : un[j];
        lu[j] = T(0);
      }
#pragma unroll
      for (int j = 0; j < m * m; ++j) luu[j] = T(0);
      const Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
      l = cost_derivs<T, MODEL>(P, z, u, tr, terminal, lz, lzz, lu, luu);
      if (!terminal) {
        dynamics<T, MODEL, true>(P, z, u, tr, zn, Fz, Fu);
      } else {
#pragma unroll
        for (int j = 0; j < n * n; ++j) Fz[j] = T(0);
#pragma unroll
        for (int j = 0; j < n * m; ++j) Fu[j] = T(0);
      }
      T* col = stage + tid;
#pragma unroll
      for (int j = 0; j < n * n; ++j) col[(lay.oFz + j) * LD] = Fz[j];
#pragma unroll
      for (int j = 0; j < n * n; ++j) col[(lay.oLzz + j) * LD] = lzz[j];
#pragma unroll
      for (int j = 0; j < n * m; ++j) col[(lay.oFu + j) * LD] = Fu[j];
#pragma unroll
      for (int j = 0; j < m * n; ++j) col[(lay.oLuz + j) * LD] = T(0);
#pragma unroll
      for (int j = 0; j < n; ++j) col[(lay.oLz + j) * LD] = lz[j];
#pragma unroll
      for (int j = 0; j < m * m; ++j) col[(lay.oLuu + j) * LD] = luu[j];
#pragma unroll
      for (int j = 0; j < m; ++j) col[(lay.oLu + j) * LD] = lu[j];
#pragma unroll
      for (int j = 0; j < m; ++j) col[(lay.oU + j) * LD] = un[j];
#pragma unroll
      for (int j = lay.oU + m; j < S; ++j) col[j * LD] = T(0);
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
template <typename T>
struct LineSearchArgs {
  int B, N, A;
  const T* Z;
  const T* U;
  const T* gains;
  const T* alphas;
  const T* u_min;
  const T* u_max;
  const uint8_t* active;
  const int32_t* bwd_status;
  // Candidates are laid out time-major, Zc [B][N+1][A][n], Uc [B][N][A][m]:
  // the A lanes of a trajectory then write ONE contiguous segment per step
  // (160 B for cartpole) instead of A scattered 16-B pieces of A different
  // rows.  Measured on gfx950 (rocprofv3 WRITE_SIZE): candidate-major cost
  // 152 MB of HBM writes per launch for 82 MB of data and a third of the
  // kernel's time; the accept kernel's strided read of the one winning row is
  // 12x smaller than what this saves.
  T* Zc;
  T* Uc;
  T* Jc;
};

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

// Same line search with the trajectory's nominal data staged in LDS: 16 lanes
// per trajectory (one per alpha, A <= 16), four trajectories per wavefront.
// Z, U and the gains of a trajectory (4 KB for cartpole at N = 100) are copied
// into LDS once, coalesced, and every step then reads them as LDS broadcasts:
// the dependent chain never waits on a global load.
template <typename T, int MODEL>
__global__ __launch_bounds__(kWave) void line_search_lds_kernel(
    ProblemT<T> P, LineSearchArgs<T> a) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr int GS = m + m * n;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x;
  const int grp = lane >> 4, ai = lane & 15;
  const int N = a.N;
  const int per = (N + 1) * n + N * m + N * GS;  // scalars per trajectory
  const int b0 = blockIdx.x * 4;

  // cooperative, coalesced staging of up to four trajectories
  for (int g = 0; g < 4; ++g) {
    const int bg = b0 + g;
    if (bg >= a.B) break;
    if (a.active != nullptr && a.active[bg] == 0) continue;
    T* dst = smem + (size_t)g * per;
    const T* zs = a.Z + (size_t)bg * (N + 1) * n;
    const T* us = a.U + (size_t)bg * N * m;
    const T* gs = a.gains + (size_t)bg * N * GS;
    for (int o = lane; o < (N + 1) * n; o += kWave) dst[o] = zs[o];
    for (int o = lane; o < N * m; o += kWave) dst[(N + 1) * n + o] = us[o];
    for (int o = lane; o < N * GS; o += kWave)
      dst[(N + 1) * n + N * m + o] = gs[o];
  }
  __syncthreads();

  const int b = b0 + grp;
  if (b >= a.B || ai >= a.A) return;
  if (a.active != nullptr && a.active[b] == 0) return;
  if (a.bwd_status != nullptr && a.bwd_status[b] != 0) return;
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  T umin[m], umax[m];  // hoisted: a load in the loop sits on the chain
#pragma unroll
  for (int r = 0; r < m; ++r) {
    umin[r] = bounded ? a.u_min[r] : T(0);
    umax[r] = bounded ? a.u_max[r] : T(0);
  }
  const T alpha = a.alphas[ai];
  const T* Zs = smem + (size_t)grp * per;
  const T* Us = Zs + (N + 1) * n;
  const T* Gs = Us + N * m;
  const int idx = b * a.A + ai;
  T* Zci = a.Zc + ((size_t)b * (N + 1) * a.A + ai) * n;
  T* Uci = a.Uc + ((size_t)b * N * a.A + ai) * m;
  const size_t zstep = (size_t)a.A * n, ustep = (size_t)a.A * m;

  T z[n], zn[n], un[m];
#pragma unroll
  for (int j = 0; j < n; ++j) z[j] = Zs[j];  // Z_new[0] = Z[0]    (ilqr.py:690)
  T J = T(0);
  for (int t = 0; t < N; ++t) {
    const T* zr = Zs + t * n;
    const T* gr = Gs + t * GS;
#pragma unroll
    for (int r = 0; r < m; ++r) {
      T du = alpha * gr[r];  // alpha * k[i]                      (ilqr.py:708)
      T s = T(0);
#pragma unroll
      for (int c = 0; c < n; ++c) s += (z[c] - zr[c]) * gr[m + r * n + c];
      du = du + s;  // + dz K^T                                   (ilqr.py:710)
      const T v = Us[t * m + r] + du;
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
    for (int j = 0; j < n; ++j) z[j] = zn[j];
  }
#pragma unroll
  for (int j = 0; j < n; ++j) Zci[(size_t)N * zstep + j] = z[j];
  const T lf = cost_value<T, MODEL>(P, z, nullptr, trig_of<T, MODEL>(z), true);
  a.Jc[idx] = J + lf;  // L.sum(0) + l_f                           (ilqr.py:789)
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
template <typename T, int MODEL>
static int launch_rollout(const pddp_problem& p, RolloutArgs<T> a,
                          hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  const int blocks = (a.B + kWave - 1) / kWave;
  hipLaunchKernelGGL((nominal_rollout_kernel<T, MODEL>), dim3(blocks),
                     dim3(kWave), 0, st, P, a);
  return launch_status();
}
template <typename T, int MODEL>
static int launch_derivs(const pddp_problem& p, DerivArgs<T> a, hipStream_t st) {
  const ProblemT<T> P = convert_problem<T>(p);
  hipLaunchKernelGGL((derivs_kernel<T, MODEL>), dim3(a.B), dim3(kDerivThreads),
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
    hipLaunchKernelGGL((line_search_lds_kernel<T, MODEL>),
                       dim3((a.B + 3) / 4), dim3(kWave), lds, st, P, a);
    return launch_status();
  }
  const int total = a.B * a.A;
  const int blocks = (total + kWave - 1) / kWave;
  hipLaunchKernelGGL((line_search_kernel<T, MODEL>), dim3(blocks), dim3(kWave),
                     0, st, P, a);
  return launch_status();
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
  if (int rc = check_problem(p)) return rc;
  if (B <= 0 || N <= 0 || !z0 || !U || !Z) return PDDP_E_BADARG;
  RolloutArgs<T> a{B, N, z0, U, u_min, u_max, mask, Z};
  PDDP_DISPATCH_MODEL(launch_rollout, T, p, a, (hipStream_t)stream)
}

template <typename T>
static int derivs_impl(const pddp_problem* p, int B, int N, const T* Z,
                       const T* U, const T* u_min, const T* u_max,
                       const uint8_t* mask, T* rec, T* L, T* J, int32_t* state,
                       void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (B <= 0 || N <= 0 || !Z || !U || !rec || !L || !J) return PDDP_E_BADARG;
  DerivArgs<T> a{B, N, Z, U, u_min, u_max, mask, rec, L, J, state};
  PDDP_DISPATCH_MODEL(launch_derivs, T, p, a, (hipStream_t)stream)
}

template <typename T>
static int line_search_impl(const pddp_problem* p, int B, int N, int A,
                            const T* Z, const T* U, const T* gains,
                            const T* alphas, const T* u_min, const T* u_max,
                            const uint8_t* active, const int32_t* bwd_status,
                            T* Zc, T* Uc, T* Jc, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (B <= 0 || N <= 0 || A <= 0 || !Z || !U || !gains || !alphas || !Zc ||
      !Uc || !Jc)
    return PDDP_E_BADARG;
  LineSearchArgs<T> a{B, N, A, Z, U, gains, alphas, u_min, u_max, active,
                      bwd_status, Zc, Uc, Jc};
  PDDP_DISPATCH_MODEL(launch_line_search, T, p, a, (hipStream_t)stream)
}

}  // namespace pddp

extern "C" {

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
