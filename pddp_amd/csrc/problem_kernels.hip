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
// dst[0..count) = src[0..count) by the 16 lanes of a trajectory group (lane
// index l16).  Loads are issued eight at a time before the first store: a
// plain load-store loop pays the full memory latency once per iteration.
template <typename T>
PDDP_DEV void group_copy(T* dst, const T* src, int count, int l16) {
  constexpr int kDeep = 8;
  for (int o0 = l16; o0 < count; o0 += 16 * kDeep) {
    T tmp[kDeep];
#pragma unroll
    for (int r = 0; r < kDeep; ++r) {
      const int o = o0 + 16 * r;
      tmp[r] = src[o < count ? o : 0];
    }
#pragma unroll
    for (int r = 0; r < kDeep; ++r) {
      const int o = o0 + 16 * r;
      if (o < count) dst[o] = tmp[r];
    }
  }
}

// 16-B aligned store of four consecutive words (two 16-B stores for double)
PDDP_DEV void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
PDDP_DEV void store4(double* p, double a, double b, double c, double d) {
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
  *reinterpret_cast<double2*>(p + 2) = make_double2(c, d);
}

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

// u = clamp(U + alpha k + K (z - Z))  (ilqr.py:708-712, utils/constraint.py
// clamp) with the multiply-adds written out: the rollout and the tail of the
// fused launch (which re-evaluates the winner's actions instead of gathering
// them from Uc) must round alike, and the compiler contracts a plain
// expression differently from one inlined copy to the next.
template <typename T, int n, int m>
PDDP_DEV void control_law(const T* z, const T* zr, const T* gr, const T* us,
                          T alpha, const T* umin, const T* umax, T* un) {
#pragma unroll
  for (int r = 0; r < m; ++r) {
    T s = T(0);
#pragma unroll
    for (int c = 0; c < n; ++c)
      s = n4::fma_(z[c] - zr[c], gr[m + r * n + c], s);  // dz K^T  (ilqr.py:710)
    const T du = n4::fma_(alpha, gr[r], s);  // alpha * k[i] +       (ilqr.py:708)
    const T v = us[r] + du;
    // unbounded: umin / umax are -inf / +inf
    un[r] = clamp_nan(v, umin[r], umax[r]);
  }
}

// Same line search with the trajectory's nominal data staged in LDS: 16 lanes
// per trajectory (one per alpha, A <= 16), four trajectories per wavefront.
// Z, U and the gains of a trajectory (4 KB for cartpole at N = 100) are copied
// into LDS once, coalesced, and every step then reads them as LDS broadcasts:
// the dependent chain never waits on a global load.
//
// FUSED: the same wavefront goes on with what the round does next for its four
// trajectories - argmin over the candidates (DPP butterflies), the accept /
// regularisation state machine (accept.hpp), the copy of the winning
// candidate into the nominal and, where the fit continues, the derivative
// records of the new nominal - so that a round is three launches (records of
// fresh nominals only at the start, sweep, this) instead of five, and the
// winner's rows are read back while they are still in L2.
// WPB wavefronts per workgroup, each with its own four trajectories and LDS
// slice (no interaction after the staging barrier).  WPB = 4 makes a workgroup
// one wave per SIMD of a CU whatever the dispatcher did before - with
// one-wave workgroups the placement of 1024 of them on 1024 SIMDs depended on
// the previous kernel's shape (measured: +13 us after a 128-thread kernel).
//
// H = 2 (FUSED only): a HELPER wavefront per rollout wavefront.  The rollouts
// are a dependent chain that one wave per SIMD runs as fast as it can be run;
// the tail is the opposite - every (trajectory, step) record of the accepted
// nominals is ~770 independent instructions - and a wave that has a SIMD to
// itself issues at half the SIMD's rate.  The helper (same four trajectories,
// same LDS slice) helps with the staging, sleeps at a barrier through the
// rollouts, and takes every other row of the tail.
// QM: live rows / columns of the stage cost matrix (models.hpp live_mask).
// DENSE (FUSED, H = 1; round 4): for the batches that do not fit a CU's two
// resident workgroups of the paired form (from 8193 trajectories on, where the
// launch ran one and a half rounds of workgroups: DESIGN.md 3.4).  Only the
// gains are staged in LDS (2 KB per trajectory instead of 4: the nominal's
// states and actions are read from global memory - with three or four rollout
// wavefronts on a SIMD their latency is covered), there is no helper wavefront
// (the tail's short form takes eight rows per lane instead of four), and four
// workgroups of four wavefronts share a CU.
template <typename T, int MODEL, bool FUSED, int WPB, int H = 1,
          unsigned QM = kFullMask<MODEL>, bool DENSE = false>
// (f32, n <= 4: at most 128 VGPRs, so that two workgroups of eight waves share
// a CU at the batches that have more than 256 workgroups)
__global__ __launch_bounds__(kWave * WPB * H) __attribute__((
    amdgpu_waves_per_eu((WPB * H >= 8 || DENSE) && sizeof(T) == 4 &&
                                ModelDims<MODEL>::n <= 4
                            ? 4
                            : 1))) void
line_search_lds_kernel(
    ProblemT<T> P, LineSearchArgs<T> a, AcceptArgs<T> c, T* rec, T* Lout) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr int GS = m + m * n;
  static_assert(H == 1 || (H == 2 && FUSED), "");
  static_assert(!DENSE || (FUSED && H == 1), "");
  constexpr int kTailRows = DENSE ? 8 : 4;  // rows per lane of the short tail
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int sh_dec[WPB][4][2];  // H = 2: {amin_out, fresh} per trajectory
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_all = threadIdx.x >> 6;
  const int wave = H == 1 ? wave_all : wave_all % WPB;
  const int hid = H == 1 ? 0 : wave_all / WPB;  // 0 rollout wave, 1 helper
  const int grp = lane >> 4, ai = lane & 15;
  const int N = a.N;
  // scalars per trajectory in LDS (DENSE: the gains only)
  const int per = DENSE ? N * GS : (N + 1) * n + N * m + N * GS;
  T* smem = reinterpret_cast<T*>(smem_raw) + (size_t)wave * 4 * per;
  const int b0 = (blockIdx.x * WPB + wave) * 4;

  // cooperative, coalesced staging of up to four trajectories
  for (int g = 0; g < 4; ++g) {
    const int bg = b0 + g;
    if (bg >= a.B) break;
    if (H == 2 && (g & 1) != hid) continue;  // the pair splits the copies
    if (a.active != nullptr && a.active[bg] == 0) continue;
    T* dst = smem + (size_t)g * per;
    const T* zs = a.Z + (size_t)bg * (N + 1) * n;
    const T* us = a.U + (size_t)bg * N * m;
    const T* gs = a.gains + (size_t)bg * N * GS;
    if constexpr (DENSE) {
      for (int o = lane; o < N * GS; o += kWave) dst[o] = gs[o];
    } else {
      for (int o = lane; o < (N + 1) * n; o += kWave) dst[o] = zs[o];
      for (int o = lane; o < N * m; o += kWave) dst[(N + 1) * n + o] = us[o];
      for (int o = lane; o < N * GS; o += kWave)
        dst[(N + 1) * n + N * m + o] = gs[o];
    }
  }
  __syncthreads();

  const int b = b0 + grp;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const bool attempted =
      exists && (a.active == nullptr || a.active[bc] != 0);
  const bool run = attempted && ai < a.A &&
                   (a.bwd_status == nullptr || a.bwd_status[bc] == 0);
  if constexpr (!FUSED) {
    if (!run) return;
  }
  const bool bounded = a.u_min != nullptr && a.u_max != nullptr;
  T umin[m], umax[m];  // hoisted: a load in the loop sits on the chain
#pragma unroll
  for (int r = 0; r < m; ++r) {
    umin[r] = bounded ? a.u_min[r] : -(T)__builtin_inff();
    umax[r] = bounded ? a.u_max[r] : (T)__builtin_inff();
  }
  // the nominal's rows: LDS copies (DENSE: states and actions in place)
  const T* Zs = DENSE ? a.Z + (size_t)bc * (N + 1) * n
                      : smem + (size_t)grp * per;
  const T* Us = DENSE ? a.U + (size_t)bc * N * m : Zs + (N + 1) * n;
  const T* Gs = DENSE ? smem + (size_t)grp * per : Us + N * m;
  const size_t zstep_c = (size_t)a.A * n, ustep_c = (size_t)a.A * m;
  // the state machine's inputs, requested now: their latency hides behind
  // the rollout
  AcceptIn<T> acc_in = {};
  if constexpr (FUSED) {
    if (attempted && ai == 0) acc_in = accept_load(c, b);
  }
  T Jmine = T(0);
  // (candidates dropped: see LineSearchArgs::drop_candidates)
  // (where the tail's short form applies: it reads the winner's compact rows)
  // (n <= 4: a second inlined copy of the larger models' step spills)
  const bool nocand = FUSED && n <= 4 && Lout == nullptr && rec != nullptr &&
                      a.drop_candidates != 0 && N + 1 <= 16 * H * kTailRows;
  // One rollout of this lane's candidate (ilqr.py:677-723 + :764-791): states
  // to Zci (stride zstep per step), actions to Uci (stride ustep), the cost
  // returned.  A stride of zero makes the target a one-row scratch - the
  // stores stay in the instruction stream (no exec mask on the chain), their
  // line stays in L2.
  auto rollout = [&](T alpha, T* Zci, size_t zstep, T* Uci, size_t ustep) {
    T z[n], zn[n], un[m];
#pragma unroll
    for (int j = 0; j < n; ++j) z[j] = Zs[j];  // Z_new[0] = Z[0]  (ilqr.py:690)
    T J = T(0);
    auto step = [&](const int t) {
      // the step's nominal row, requested from LDS first; the sines and
      // cosines of the state need none of it and run while it arrives (the
      // compiler, left alone, starts with the control law and stalls on it)
      T zr[n], gr[GS], us[m];
#pragma unroll
      for (int j = 0; j < n; ++j) zr[j] = Zs[t * n + j];
#pragma unroll
      for (int j = 0; j < GS; ++j) gr[j] = Gs[t * GS + j];
#pragma unroll
      for (int j = 0; j < m; ++j) us[j] = Us[t * m + j];
      const Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
      __builtin_amdgcn_sched_barrier(0);
      control_law<T, n, m>(z, zr, gr, us, alpha, umin, umax, un);
#pragma unroll
      for (int j = 0; j < n; ++j) Zci[(size_t)t * zstep + j] = z[j];
#pragma unroll
      for (int j = 0; j < m; ++j) Uci[(size_t)t * ustep + j] = un[j];
      J += cost_value<T, MODEL, QM>(P, z, un, tr, false);
      dynamics<T, MODEL, false>(P, z, un, tr, zn, nullptr, nullptr);
#pragma unroll
      for (int j = 0; j < n; ++j) z[j] = zn[j];
    };
    // four steps per trip, by hand (the wave-uniform test inside sincos_ is a
    // convergent operation: the compiler does not unroll such a loop with a
    // run-time trip count): a quarter of the loop branches and address
    // updates - 42.3 -> 37.5 us for the rollouts at B = 4096
    int t = 0;
    if constexpr (n <= 4) {  // (the larger models' steps spill when copied)
      for (; t + 3 < N; t += 4) {
        step(t);
        step(t + 1);
        step(t + 2);
        step(t + 3);
      }
    }
#pragma unroll 1
    for (; t < N; ++t) step(t);
#pragma unroll
    for (int j = 0; j < n; ++j) Zci[(size_t)N * zstep + j] = z[j];
    const T lf =
        cost_value<T, MODEL>(P, z, nullptr, trig_of<T, MODEL>(z), true);
    return J + lf;  // L.sum(0) + l_f                              (ilqr.py:789)
  };
  if (run && hid == 0) {
    const T alpha = a.alphas[ai];
    const int idx = b * a.A + ai;
    // FUSED without records, `rec` given as scratch: the FULL STEP (candidate
    // 0, the winner of 19 accepted attempts in 20 - tools/ls_tail_profile.py)
    // writes its states to rec[b][N + 1][n], rows next to one another, instead
    // of Zc[b][.][0][.]: the tail's copy of the winner into the nominal then
    // reads whole sectors instead of 16 bytes out of every 160-byte step of Zc
    // (a per-lane stride in the address update: no instruction more)
    const bool compact0 = FUSED && Lout == nullptr && rec != nullptr && ai == 0;
    T* Zci = compact0 ? rec + (size_t)b * (N + 1) * n
                      : a.Zc + ((size_t)b * (N + 1) * a.A + ai) * n;
    T* Uci = a.Uc + ((size_t)b * N * a.A + ai) * m;
    // candidates dropped: every other step size overwrites ONE row of its own
    const size_t zstep = compact0 ? (size_t)n : (nocand ? 0 : zstep_c);
    const size_t ustep = nocand ? 0 : ustep_c;
    Jmine = rollout(alpha, Zci, zstep, Uci, ustep);
    a.Jc[idx] = Jmine;
  }

  if constexpr (FUSED) {
    // ---- argmin with torch's semantics: the first NaN wins, else the first
    // minimum (ilqr.py:161); every lane of the group gets the same answer
    constexpr int kNone = 99;
    const T kInf = (T)__builtin_inff();
    const int nan_first =
        n4::group_min((run && Jmine != Jmine) ? ai : kNone);
    T Jf = (run && Jmine == Jmine) ? Jmine : kInf;
    {
      T y = n4::dpp<0x128>(Jf); Jf = y < Jf ? y : Jf;
      y = n4::dpp<0x12C>(Jf); Jf = y < Jf ? y : Jf;
      y = n4::dpp<(2 | (3 << 2) | (0 << 4) | (1 << 6))>(Jf); Jf = y < Jf ? y : Jf;
      y = n4::dpp<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(Jf); Jf = y < Jf ? y : Jf;
    }
    const int min_first = n4::group_min((run && Jmine == Jf) ? ai : kNone);
    const int amin =
        nan_first != kNone ? nan_first : (min_first != kNone ? min_first : 0);
    const T J_new = nan_first != kNone ? (T)__builtin_nanf("") : Jf;

    // ---- accept / reject, mu schedule, masks: lane 0 of the group
    int amin_out = -1, fresh_i = 0;
    if (attempted && ai == 0 && hid == 0) {
      bool fr;
      amin_out = accept_decide(c, b, acc_in, amin, J_new, fr);
      fresh_i = fr ? 1 : 0;
    }
    amin_out = __shfl(amin_out, lane & 48);
    fresh_i = __shfl(fresh_i, lane & 48);
    // candidates dropped and the winner is not the full step: its lane rolls
    // it out once more, into the compact rows the tail reads (the same code on
    // the same inputs: its states to rounding - two inlined copies of the step
    // are not contracted alike - and its cost, Jc, from the first time)
    if constexpr (n <= 4) {
      if (nocand && hid == 0 && __any(amin_out > 0)) {
        if (amin_out > 0 && ai == amin_out)
          rollout(a.alphas[ai], rec + (size_t)b * (N + 1) * n, (size_t)n,
                  a.Uc + ((size_t)b * N * a.A + ai) * m, 0);
      }
    }
    if constexpr (H == 1) {
      if (!__any(amin_out >= 0)) return;
      // the candidate rows written above are read back below, by this same
      // wavefront: workgroup scope (an agent-scope fence writes back the
      // XCD's whole L2 on gfx950 - measured: +40 us per launch)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    } else {
      // hand the decisions (and, through the workgroup-scope fence, the
      // candidate rows) to the helper wave
      if (hid == 0 && ai == 0) {
        sh_dec[wave][grp][0] = amin_out;
        sh_dec[wave][grp][1] = fresh_i;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __syncthreads();
      amin_out = sh_dec[wave][grp][0];
      fresh_i = sh_dec[wave][grp][1];
    }
    if (amin_out >= 0) {
      // nominal <- winning candidate; self._K <- K            (ilqr.py:167-169)
      constexpr RecLayout lay(n, m);
      constexpr int S = lay.stride;
      const T* srcz = a.Zc + ((size_t)b * (N + 1) * a.A + amin_out) * n;
      const T* srcu = a.Uc + ((size_t)b * N * a.A + amin_out) * m;
      T* Zb = c.Z + (size_t)b * (N + 1) * n;
      T* Ub = c.U + (size_t)b * N * m;
      T* rec_b = rec + (size_t)b * (N + 1) * S;
      T* Ls = smem + (size_t)grp * per;  // the staged nominal is dead: L[t]
      // rows t = ai, ai + 16, ... of the winner; the next row is requested
      // before this row's record is evaluated (a rolled loop: the record
      // code is ~900 instructions, unrolled copies would not fit the I-cache)
      const T* G = c.gains + (size_t)b * N * GS;
      T* Ga = c.gains_acc + (size_t)b * N * GS;
      T zc[n], uc[m];
      const int t_first = ai + 16 * hid;  // rows t_first, t_first + 16 H, ...
      constexpr int KR = kTailRows;  // rows per lane the short form below
                                     // covers (`nocand` above knows it)
      if (n <= 6 && Lout == nullptr && N + 1 <= 16 * H * KR) {
        // No records to write (the next sweep evaluates them): the tail is
        // the winner's rows - all of this lane's requested at once, one
        // memory latency instead of one per row - and the gains, which are
        // still in the LDS copy staged for the rollouts.  What is left of the
        // tail (~9 us of a launch that accepts everything) is this gather:
        // 16- and 4-byte rows out of the 160- / 40-byte steps of Zc / Uc, long
        // evicted from L2 - a 64-byte sector from memory for each.  (Measured
        // and not kept: the full step - the winner 19 times in 20 - rolled out
        // once more by the helper wavefront into adjacent rows: the second
        // rollout on the SIMD slows the first by a quarter; the full step's
        // lane writing its rows over the nominal's in LDS: +3 us per launch,
        // as much as it saves.)
        // The winner's ACTIONS are not gathered (a 64-byte sector for four
        // bytes): they are its control law at the gathered states, evaluated
        // again from the nominal row in LDS - the same operations in the
        // same order as in the rollout (control_law), bit for bit.
        T zz[KR][n], uu[KR][m];
        // (compact0; with the candidates dropped every winner's rows are there)
        const bool from_rec = rec != nullptr && (amin_out == 0 || nocand);
        const T* cz = from_rec ? rec + (size_t)b * (N + 1) * n : srcz;
        const size_t czs = from_rec ? (size_t)n : zstep_c;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const int t = t_first + 16 * H * k;
          const int tz = t <= N ? t : N;
#pragma unroll
          for (int j = 0; j < n; ++j) zz[k][j] = cz[(size_t)tz * czs + j];
        }
        const T* Gl = Gs;  // (the staged gains)
        for (int o = ai + 16 * hid; o < N * GS; o += 16 * H) Ga[o] = Gl[o];
        const T alpha_w = a.alphas[amin_out];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const int t = t_first + 16 * H * k;
          const int tu = t < N ? t : 0;
          T zr[n], gr[GS], us[m];
#pragma unroll
          for (int j = 0; j < n; ++j) zr[j] = Zs[tu * n + j];
#pragma unroll
          for (int j = 0; j < GS; ++j) gr[j] = Gs[tu * GS + j];
#pragma unroll
          for (int j = 0; j < m; ++j) us[j] = Us[tu * m + j];
          control_law<T, n, m>(zz[k], zr, gr, us, alpha_w, umin, umax, uu[k]);
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const int t = t_first + 16 * H * k;
          if (t <= N) {
#pragma unroll
            for (int j = 0; j < n; ++j) Zb[t * n + j] = zz[k][j];
          }
          if (t < N) {
#pragma unroll
            for (int j = 0; j < m; ++j) Ub[t * m + j] = uu[k][j];
          }
        }
      } else {
      {
        const int tz = t_first <= N ? t_first : N;
        const int tu = t_first < N ? t_first : 0;
#pragma unroll
        for (int j = 0; j < n; ++j) zc[j] = srcz[(size_t)tz * zstep_c + j];
#pragma unroll
        for (int j = 0; j < m; ++j) uc[j] = srcu[(size_t)tu * ustep_c + j];
      }
#pragma unroll 1
      for (int t = t_first; t <= N; t += 16 * H) {
        T zn_[n], un_[m];
        {
          const int t2 = t + 16 * H;
          const int tz = t2 <= N ? t2 : N, tu = t2 < N ? t2 : 0;
#pragma unroll
          for (int j = 0; j < n; ++j) zn_[j] = srcz[(size_t)tz * zstep_c + j];
#pragma unroll
          for (int j = 0; j < m; ++j) un_[j] = srcu[(size_t)tu * ustep_c + j];
        }
        const bool terminal = (t == N);
        T un[m];
#pragma unroll
        for (int j = 0; j < m; ++j) un[j] = terminal ? T(0) : uc[j];
#pragma unroll
        for (int j = 0; j < n; ++j) Zb[t * n + j] = zc[j];
        if (!terminal) {
#pragma unroll
          for (int j = 0; j < m; ++j) Ub[t * m + j] = un[j];
        }
        if (fresh_i && Lout != nullptr) {
          // derivative record of the new nominal (the next round's sweep;
          // Lout == nullptr: that sweep evaluates them itself)
          T w[S];
          const T l = record_of<T, MODEL>(P, zc, un, terminal, bounded, a.u_min,
                                         a.u_max, w);
          T* dst = rec_b + (size_t)t * S;
#pragma unroll
          for (int j = 0; j < S; j += 4)
            store4(dst + j, w[j], w[j + 1], w[j + 2], w[j + 3]);
          Lout[(size_t)b * (N + 1) + t] = l;
          Ls[t] = l;
        }
#pragma unroll
        for (int j = 0; j < n; ++j) zc[j] = zn_[j];
#pragma unroll
        for (int j = 0; j < m; ++j) uc[j] = un_[j];
      }
      if constexpr (H == 1) {
        group_copy(Ga, G, N * GS, ai);
      } else {  // each wave of the pair copies half of the gains
        const int half = (N * GS + 1) / 2;
        const int off = hid * half;
        const int cnt = hid == 0 ? half : N * GS - half;
        group_copy(Ga + off, G + off, cnt, ai);
      }
      }  // (the form that may write records)
      if constexpr (H == 1) {
        if (fresh_i && Lout != nullptr) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (ai == 0) {
            T Jacc = T(0);
            for (int t = 0; t <= N; ++t) Jacc += Ls[t];  // L.sum(), in t order
            c.J_opt[b] = Jacc;
            c.fresh[b] = 0;  // its records are up to date
          }
        }
      }
    }
    if constexpr (H == 2) {
      __syncthreads();  // both waves' stage costs Ls[t] are in LDS
      if (amin_out >= 0 && fresh_i && Lout != nullptr && hid == 0 && ai == 0) {
        const T* Ls = smem + (size_t)grp * per;
        T Jacc = T(0);
        for (int t = 0; t <= N; ++t) Jacc += Ls[t];  // L.sum(), in t order
        c.J_opt[b] = Jacc;
        c.fresh[b] = 0;  // its records are up to date
      }
    }
  }
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
