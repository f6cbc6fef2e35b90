// round_n4.hip - ONE launch for a whole round of the cartpole f32 path
// (BASELINE.json configs[1]): the backward sweep from the nominal
// (riccati_n4_elem.hpp; ilqr.py:489-674 with the records of :393-486 evaluated
// in place) and then, in the SAME wavefronts for the same four trajectories,
// the batched line search, argmin, accept / regularisation schedule and the
// copy of the winner into the nominal (line_search_lds.hpp; ilqr.py:677-791,
// :140-181, :364-390).
//
// Why (tools/wg_timeline.py, profiles/r05_wg_timeline_*.txt):
// as two launches a round at B = 4096 was 29.3 us of sweep workgroups, 2.6 us
// of idle chip between the launches, 36.8 us of search workgroups and 2.6 us
// again before the next round - and inside the search launch every workgroup
// spent 5.4 us copying the nominal and the gains from global memory into LDS
// before its first step.  Both launches have the same shape (16 trajectories
// per workgroup, four chain wavefronts and four partner wavefronts, one
// workgroup per CU), so the workgroup simply goes on: the gains never leave
// LDS between the phases (they are still written to HBM, by the partner
// wavefront, for the API), the partner stages the nominal's rows while the
// sweep runs its last block, the nominal's cost and the sweep's status cross in
// LDS / registers, and a workgroup whose sweep ends early starts its search
// early - the round is the slowest workgroup's SUM, not the sum of the two
// slowest phases.
//
// Same code as the two launches (device functions shared with them): the
// sweep's outputs are theirs bit for bit, decisions and masks identical, the
// search's values to rounding - the same closed forms inlined into another
// kernel are contracted into FMAs differently (tests/test_gpu_parity.py::
// test_one_launch_round_equals_two_launches); R rounds per launch are R
// one-round launches bit for bit (test_rounds_in_one_launch_equal_single_
// rounds).  The search's step here differs from the stand-alone launch's in
// three places, each an A/B-measured cut: the nominal row is
// read with one 16-byte and three 8-byte LDS reads, one LDS wait per step, and
// between the rounds of a launch the nominal's last rows stay in LDS.
#include "riccati_n4_elem.hpp"
#include "line_search_lds.hpp"

namespace pddp {

template <unsigned QM, bool MULTI>
__global__ __launch_bounds__(2 * n4e::kWaves * kWave) void round_n4_kernel(
    RiccatiArgs<float> a, n4d::GenArgs<float> gen, ProblemT<float> prob,
    LineSearchArgs<float> ls, AcceptArgs<float> ac, float* scratch,
    int rounds, long long* phase_ticks, int use_carry) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // (bench.py's roofline leg: what share of the launch is sweep - rocprofv3
  // sees one kernel.  Wavefront 0 of the workgroup reads the chip's 100 MHz
  // clock around its phases and adds the differences up; NULL: nothing)
  const bool timed = phase_ticks != nullptr && threadIdx.x == 0;
  long long t_sweep = 0, t_search = 0;
  // `rounds` attempts of every trajectory, back to back: trajectories are
  // independent (ilqr.py:298-314 is a loop over ONE trajectory's attempts), a
  // workgroup owns its sixteen for the whole launch, and everything a round
  // hands to the next - nominal, regularisation state, masks, costs - was
  // written by this workgroup: a workgroup-scope fence and a barrier, no
  // launch boundary.  (A trajectory that has left the fit is skipped, as by
  // the next launch.)
  for (int r = 0;; ++r) {
    // (a zero the compiler cannot see through, added to the horizon: address
    // arithmetic and everything else that depends on it stays INSIDE the
    // round - hoisted out of this loop, both phases' invariants live across
    // both phases: 255 VGPRs and 53 spilled against 102)
    int z = 0;
    unsigned tid = threadIdx.x;
    if constexpr (MULTI) {
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      asm volatile("" : "+v"(tid));  // (likewise: what the lane id feeds)
    }
    RiccatiArgs<float> a_r = a;
    a_r.N += z;
    LineSearchArgs<float> ls_r = ls;
    ls_r.N += z;
    AcceptArgs<float> ac_r = ac;
    ac_r.N += z;
    n4e::RoundOut ro;
    const long long t0 = timed ? wall_clock64() : 0;
    // (a pair without a live trajectory leaves here, both wavefronts alike -
    // it has none in any later round either; s_barrier does not wait for
    // wavefronts that have ended)
    // (several rounds per launch: the nominal's last rows ride in LDS)
    const int carry = MULTI && use_carry ? (r == 0 ? 1 : 0) : -1;
    if (!n4e::elem_sweep_body<float, QM, true, true>(a_r, gen, prob, smem_raw,
                                                     ro, tid, carry))
      break;
    const long long t1 = timed ? wall_clock64() : 0;
    const PreStaged<float> pre{ro.Zs, ro.Us,      ro.Gs,
                               ro.status, ro.J_opt, ro.carry_rows};
    line_search_lds_body<float, PDDP_MODEL_CARTPOLE, true, n4e::kWaves, 2, QM,
                         false, true>(prob, ls_r, ac_r, scratch, nullptr,
                                      smem_raw, pre, tid);
    if (timed) {
      t_sweep += t1 - t0;
      t_search += wall_clock64() - t1;
    }
    if (!MULTI || r + 1 >= rounds) break;
    // the round's writes (global: nominal, mu, delta, J_opt, masks; LDS: read
    // to the end by the tail) before the next round's reads and LDS writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
  }
  if (timed) {
    phase_ticks[2 * blockIdx.x] += t_sweep;
    phase_ticks[2 * blockIdx.x + 1] += t_search;
  }
}

static int launch_round_n4(const pddp_problem& p, const RiccatiArgs<float>& a,
                           const n4d::GenArgs<float>& gen,
                           const LineSearchArgs<float>& ls,
                           const AcceptArgs<float>& ac, float* scratch,
                           int rounds, long long* phase_ticks,
                           hipStream_t st) {
  if (p.model != PDDP_MODEL_CARTPOLE ||
      p.encoding != PDDP_ENC_IGNORE_UNCERTAINTY || a.u_min == nullptr ||
      a.u_max == nullptr || a.branch != PDDP_BRANCH_EIG || a.N < 1 ||
      a.N + 1 > 128 || ls.A > 16 || rounds < 1)
    return PDDP_E_UNSUPPORTED;
  constexpr int kPer = n4e::kWaves * n4e::kTrajW;  // trajectories / workgroup
  const dim3 grid((a.B + kPer - 1) / kPer);
  // one workgroup per CU (142 KB of LDS at N = 100): beyond 256 workgroups
  // the two launches, whose forms for large batches share a CU, are the
  // faster round
  if (grid.x > 256u) return PDDP_E_UNSUPPORTED;
  // (several rounds per launch: the carried rows, where they fit - N <= 123)
  const size_t lds0 = (size_t)n4e::kWaves * sizeof(float) *
                      (n4e::kPairLdsOvl + n4e::round_gains_floats(a.N));
  const size_t lds1 = lds0 + (size_t)n4e::kWaves * sizeof(float) * n4e::kCarryF;
  const int use_carry = rounds > 1 && lds1 <= 159 * 1024;
  const size_t lds = use_carry ? lds1 : lds0;
  if (lds > 159 * 1024) return PDDP_E_UNSUPPORTED;
  const ProblemT<float> P = convert_problem<float>(p);
  constexpr unsigned kSparse = 0b11001u;  // CartpoleCost: {x, sin, cos}
  constexpr unsigned kFull = kFullMask<PDDP_MODEL_CARTPOLE>;
  const bool sparse =
      (live_mask(p.Q, ModelDims<PDDP_MODEL_CARTPOLE>::na) & ~kSparse) == 0;
#define PDDP_ROUND_GO(QMV)                                                    \
  do {                                                                        \
    auto kern = rounds > 1 ? round_n4_kernel<QMV, true>                       \
                           : round_n4_kernel<QMV, false>;                     \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, dim3(2 * n4e::kWaves * kWave), lds, st, a, gen,   \
                P, ls, ac, scratch, rounds, phase_ticks, use_carry);          \
  } while (0)
  if (sparse) PDDP_ROUND_GO(kSparse); else PDDP_ROUND_GO(kFull);
#undef PDDP_ROUND_GO
  return launch_status();
}

}  // namespace pddp

extern "C" int pddp_round_nominal_f32(
    const pddp_problem* problem, int B, int N, int A, float* Z, float* U,
    const float* alphas, const float* u_min, const float* u_max, int branch,
    uint8_t* active, uint8_t* fresh, float* gains, int32_t* bwd_status,
    float* L, float* J_opt, float* Zc, float* Uc, float* Jc, double tol,
    double max_reg, int n_iterations, float* gains_acc, double* mu,
    double* delta, int32_t* state, int32_t* iter, int32_t* n_live,
    float* scratch, int rounds, long long* phase_ticks, void* stream) {
  if (problem == nullptr || B <= 0 || N <= 0 || A <= 0 || !Z || !U || !alphas ||
      !active || !fresh || !gains || !bwd_status || !L || !J_opt || !Zc ||
      !Uc || !Jc || !gains_acc || !mu || !delta || !state || !iter || !scratch)
    return PDDP_E_BADARG;
  if (branch != PDDP_BRANCH_EIG && branch != PDDP_BRANCH_CHOLESKY)
    return PDDP_E_BADARG;
  pddp::RiccatiArgs<float> a;
  a.B = B; a.N = N; a.n = 4;
  a.rec = nullptr;
  a.u_min = u_min; a.u_max = u_max;
  a.reg = mu;
  a.branch = branch;
  a.active = active;
  a.gains = gains;
  a.status = bwd_status;
  const pddp::n4d::GenArgs<float> gen = {Z, U, L, J_opt, fresh};
  const pddp::LineSearchArgs<float> ls{B, N, A, Z, U, gains, alphas, u_min,
                                       u_max, active, bwd_status, Zc, Uc, Jc};
  const pddp::AcceptArgs<float> ac{
      B, N, 4, 1, A, Zc, Uc, Jc, gains, bwd_status, tol, max_reg, n_iterations,
      Z, U, gains_acc, J_opt, mu, delta, state, iter, active, fresh, n_live};
  return pddp::launch_round_n4(*problem, a, gen, ls, ac, scratch, rounds,
                               phase_ticks, (hipStream_t)stream);
}

#ifdef PDDP_WG_TIMELINE
// (this translation unit's copies of the marks: tools/wg_timeline.py)
extern "C" int pddp_debug_round_timeline(long long* sweep, long long* search) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(sweep, HIP_SYMBOL(pddp::n4e::g_elem_timeline),
                            sizeof(long long) * 1024 * 12);
  (void)hipMemcpyFromSymbol(search, HIP_SYMBOL(pddp::g_search_timeline),
                            sizeof(long long) * 1024 * 12);
  return 0;
}
#endif
