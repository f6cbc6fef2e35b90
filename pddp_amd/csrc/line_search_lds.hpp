// line_search_lds.hpp - the batched line search with the trajectory's nominal
// rows and gains in LDS (ilqr.py:677-723 _control_law + :764-791
// _trajectory_cost), and - FUSED - what the round does next for the same
// trajectories: argmin, accept / regularisation schedule (ilqr.py:140-181,
// 364-390), the copy of the winner into the nominal.  A device function, so
// that the stand-alone launches (problem_kernels.hip) and the one-launch round
// of the cartpole f32 path (round_n4.hip: the sweep from the nominal first, in
// the same workgroup) run the same code.
#pragma once

#include <type_traits>
#include "models.hpp"
#include "problem_args.hpp"
#include "accept.hpp"
#include "riccati_n4.hpp"  // DPP helpers of the 16-lane groups

namespace pddp {

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
#ifdef PDDP_WG_TIMELINE
// (riccati_n4_elem.hpp: the same marks for the search launch - 0 entry, 1
// after the staging barrier, 2 after the rollouts, 3 end; wavefront 0 in slots
// 0-3, its helper in 4-7)
__device__ long long g_search_timeline[1024][12];
#define PDDP_TLS(I)                                                           \
  do {                                                                        \
    if ((threadIdx.x & 255) == 0 && blockIdx.x < 1024)                        \
      g_search_timeline[blockIdx.x][(I) + 4 * (threadIdx.x >> 8)] =           \
          wall_clock64();                                                     \
  } while (0)
#define PDDP_TLS_AT(SLOT)                                                     \
  do {                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024)                                \
      g_search_timeline[blockIdx.x][SLOT] = wall_clock64();                   \
  } while (0)
#else
#define PDDP_TLS(I)
#define PDDP_TLS_AT(SLOT)
#endif

// PRE (round_n4.hip: the sweep and this in ONE launch): the nominal's rows and
// the gains are in LDS already - `pre` points at this lane's trajectory's - the
// sweep's status and the nominal's cost come in registers, and the caller has
// passed the barrier behind which all of that is visible.
template <typename T>
struct PreStaged {
  const T* Zs;
  const T* Us;
  const T* Gs;
  int status;
  T J_opt;
  // nullable: [17][n + m] in LDS, the nominal's rows t = N - 16 .. N at index
  // N - t - the winner's go there too (round_n4.hip, several rounds per
  // launch: the next round's first records are made from them)
  T* carry_rows = nullptr;
};
template <typename T, int MODEL, bool FUSED, int WPB, int H, unsigned QM,
          bool DENSE, bool PRE = false>
PDDP_DEV void line_search_lds_body(const ProblemT<T> P,
                                   const LineSearchArgs<T> a,
                                   const AcceptArgs<T> c, T* rec, T* Lout,
                                   unsigned char* smem_raw,
                                   const PreStaged<T> pre,
                                   const unsigned tid = threadIdx.x) {
  using D = ModelDims<MODEL>;
  constexpr int n = D::n, m = D::m;
  constexpr int GS = m + m * n;
  static_assert(H == 1 || (H == 2 && FUSED), "");
  static_assert(!DENSE || (FUSED && H == 1), "");
  static_assert(!PRE || (FUSED && H == 2 && !DENSE), "");
  // rows of the staged gains / actions: [k, K] of a step next to one another
  // and the actions in a table of their own - or, PRE (round_n4.hip), one row
  // [k, K, u] per step (riccati_n4_elem.hpp lays it out: 8-byte reads)
  constexpr int GST = PRE ? GS + m : GS;
  constexpr int UST = PRE ? GS + m : m;
  constexpr int kTailRows = DENSE ? 8 : 4;  // rows per lane of the short tail
  __shared__ int sh_dec[WPB][4][2];  // H = 2: {amin_out, fresh} per trajectory
  const int lane = tid & (kWave - 1);
  const int wave_all = tid >> 6;
  const int wave = H == 1 ? wave_all : wave_all % WPB;
  const int hid = H == 1 ? 0 : wave_all / WPB;  // 0 rollout wave, 1 helper
  const int grp = lane >> 4, ai = lane & 15;
  const int N = a.N;
  // scalars per trajectory in LDS (DENSE: the gains only)
  const int per = DENSE ? N * GS : (N + 1) * n + N * m + N * GS;
  T* smem = reinterpret_cast<T*>(smem_raw) + (size_t)wave * 4 * per;
  const int b0 = (blockIdx.x * WPB + wave) * 4;
  PDDP_TLS(0);

  // cooperative, coalesced staging of up to four trajectories.  (5.4 us of the
  // launch at B = 4096, tools/wg_timeline.py - and not the loop's doing:
  // eight or sixteen requests in flight per lane change nothing, round 5; all
  // 256 workgroups ask for their 64 KB in the launch's first microsecond, 16
  // MB at what the memory system delivers in a burst.  round_n4.hip stages
  // the same words during the sweep's last block instead.)
  for (int g = 0; g < (PRE ? 0 : 4); ++g) {
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
  if constexpr (!PRE) __syncthreads();
  PDDP_TLS(1);

  const int b = b0 + grp;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const bool attempted =
      exists && (a.active == nullptr || a.active[bc] != 0);
  const bool run =
      attempted && ai < a.A &&
      (PRE ? pre.status == 0
           : (a.bwd_status == nullptr || a.bwd_status[bc] == 0));
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
  const T* Zs = PRE ? pre.Zs
                    : DENSE ? a.Z + (size_t)bc * (N + 1) * n
                            : smem + (size_t)grp * per;
  const T* Us = PRE ? pre.Us
                    : DENSE ? a.U + (size_t)bc * N * m : Zs + (N + 1) * n;
  const T* Gs = PRE ? pre.Gs : DENSE ? smem + (size_t)grp * per : Us + N * m;
  const size_t zstep_c = (size_t)a.A * n, ustep_c = (size_t)a.A * m;
  // the state machine's inputs, requested now: their latency hides behind
  // the rollout
  AcceptIn<T> acc_in = {};
  if constexpr (FUSED) {
    if (attempted && ai == 0) {
      acc_in = accept_load(c, b);
      if constexpr (PRE) {
        acc_in.bstat = pre.status;
        acc_in.J_opt = pre.J_opt;
      }
    }
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
  // (the candidates' actions are only stored where something reads them: the
  // tail's short form re-evaluates the winner's)
  const bool store_uc = !(FUSED && n <= 6 && Lout == nullptr &&
                          rec != nullptr && N + 1 <= 16 * H * kTailRows);
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
      if constexpr (PRE && n == 4) {
        // (the round kernel's staged rows are 16-byte aligned: one read)
        typedef T V4_ __attribute__((ext_vector_type(4)));
        const V4_ v = *reinterpret_cast<const V4_*>(
            __builtin_assume_aligned(Zs + t * 4, 16));
        zr[0] = v[0]; zr[1] = v[1]; zr[2] = v[2]; zr[3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < n; ++j) zr[j] = Zs[t * n + j];
      }
      if constexpr (PRE && GST == 6) {
        // (k K0 | K1 K2 | K3 u: three 8-byte reads)
        typedef T V2_ __attribute__((ext_vector_type(2)));
        const V2_* g2 = reinterpret_cast<const V2_*>(
            __builtin_assume_aligned(Gs + t * 6, 8));
        const V2_ a0 = g2[0], a1 = g2[1], a2 = g2[2];
        gr[0] = a0[0]; gr[1] = a0[1]; gr[2] = a1[0]; gr[3] = a1[1];
        gr[4] = a2[0]; us[0] = a2[1];
      } else {
#pragma unroll
        for (int j = 0; j < GS; ++j) gr[j] = Gs[t * GST + j];
#pragma unroll
        for (int j = 0; j < m; ++j) us[j] = Us[t * UST + j];
      }
      const Trig<T, MODEL> tr = trig_of<T, MODEL>(z);
      __builtin_amdgcn_sched_barrier(0);
      // (PRE: the row was requested a sincos ago - ONE s_waitcnt lgkmcnt(0)
      // for all of it instead of one per operand of the control law: 51.0 ->
      // 50.7 us per round, tools/dbg/ab_round.py)
      if constexpr (PRE) __builtin_amdgcn_s_waitcnt(0xC07F);
      control_law<T, n, m>(z, zr, gr, us, alpha, umin, umax, un);
#pragma unroll
      for (int j = 0; j < n; ++j) Zci[(size_t)t * zstep + j] = z[j];
      // (PRE - the round kernel: the candidates' actions are not stored.
      // Its tail is the short form, which re-evaluates the winner's actions
      // from the nominal row in LDS, bit for bit, and nothing else reads Uc:
      // one store instruction per step and 18 MB per round less - 52.4 ->
      // 51.8 us per round, tools/dbg/ab_round.py)
      if constexpr (!PRE) {
        if (store_uc) {
#pragma unroll
          for (int j = 0; j < m; ++j) Uci[(size_t)t * ustep + j] = un[j];
        }
      }
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
    // (only where the tail's short form - the reader of those rows - applies:
    // the long form gathers every winner from Zc.  Round 5: it was taken for
    // N + 1 > 64 H as well, and an accepted full step then copied rows of Zc
    // nobody had written into the nominal - cartpole f32 horizons beyond
    // ~101, found by test_one_launch_round_equals_two_launches[16-127])
    const bool compact0 = FUSED && Lout == nullptr && rec != nullptr &&
                          ai == 0 && n <= 6 && N + 1 <= 16 * H * kTailRows;
    T* Zci = compact0 ? rec + (size_t)b * (N + 1) * n
                      : a.Zc + ((size_t)b * (N + 1) * a.A + ai) * n;
    T* Uci = a.Uc + ((size_t)b * N * a.A + ai) * m;
    // candidates dropped: every other step size overwrites ONE row of its own
    const size_t zstep = compact0 ? (size_t)n : (nocand ? 0 : zstep_c);
    const size_t ustep = nocand ? 0 : ustep_c;
    Jmine = rollout(alpha, Zci, zstep, Uci, ustep);
    a.Jc[idx] = Jmine;
  }
  PDDP_TLS(2);

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
    PDDP_TLS_AT(8);
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
    PDDP_TLS_AT(9);
    if (amin_out >= 0) {
      // nominal <- winning candidate; self._K <- K            (ilqr.py:167-169)
      constexpr RecLayout lay(n, m);
      constexpr int S = lay.stride;
      const T* srcz = a.Zc + ((size_t)b * (N + 1) * a.A + amin_out) * n;
      const T* srcu = a.Uc + ((size_t)b * N * a.A + amin_out) * m;
      T* Zb = c.Z + (size_t)b * (N + 1) * n;
      T* Ub = c.U + (size_t)b * N * m;
      T* rec_b = rec + (size_t)b * (N + 1) * S;
      // the staged nominal is dead: L[t]  (PRE: no records, never used)
      T* Ls = PRE ? nullptr : smem + (size_t)grp * per;
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
        if constexpr (PRE) {
          // (rows of GST words in LDS, of GS in gains_acc)
          for (int t = ai + 16 * hid; t < N; t += 16 * H) {
#pragma unroll
            for (int j = 0; j < GS; ++j) Ga[t * GS + j] = Gl[t * GST + j];
          }
        } else {
          // eight words per lane requested from LDS before the first is
          // stored (a plain loop waits out an LDS latency per word: 16 trips
          // at N = 100)
          constexpr int kDeep = 8;
          int o = ai + 16 * hid;
          for (; o + 16 * H * (kDeep - 1) < N * GS; o += 16 * H * kDeep) {
            T tmp[kDeep];
#pragma unroll
            for (int r = 0; r < kDeep; ++r) tmp[r] = Gl[o + 16 * H * r];
#pragma unroll
            for (int r = 0; r < kDeep; ++r) Ga[o + 16 * H * r] = tmp[r];
          }
          for (; o < N * GS; o += 16 * H) Ga[o] = Gl[o];
        }
        PDDP_TLS_AT(10);
        const T alpha_w = a.alphas[amin_out];
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const int t = t_first + 16 * H * k;
          const int tu = t < N ? t : 0;
          T zr[n], gr[GS], us[m];
#pragma unroll
          for (int j = 0; j < n; ++j) zr[j] = Zs[tu * n + j];
#pragma unroll
          for (int j = 0; j < GS; ++j) gr[j] = Gs[tu * GST + j];
#pragma unroll
          for (int j = 0; j < m; ++j) us[j] = Us[tu * UST + j];
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
          if constexpr (PRE) {
            if (pre.carry_rows != nullptr && t <= N && N - t <= 16) {
              T* cr = pre.carry_rows + (N - t) * (n + m);
#pragma unroll
              for (int j = 0; j < n; ++j) cr[j] = zz[k][j];
              if (t < N) {
#pragma unroll
                for (int j = 0; j < m; ++j) cr[n + j] = uu[k][j];
              }
            }
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
      if (!PRE && amin_out >= 0 && fresh_i && Lout != nullptr && hid == 0 &&
          ai == 0) {
        const T* Ls = smem + (size_t)grp * per;
        T Jacc = T(0);
        for (int t = 0; t <= N; ++t) Jacc += Ls[t];  // L.sum(), in t order
        c.J_opt[b] = Jacc;
        c.fresh[b] = 0;  // its records are up to date
      }
    }
  }
  PDDP_TLS(3);
}


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
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  line_search_lds_body<T, MODEL, FUSED, WPB, H, QM, DENSE>(
      P, a, c, rec, Lout, smem_raw, PreStaged<T>{});
}

}  // namespace pddp
