// accept.hpp - the per-trajectory controller state machine
//   accept / reject + mu schedule   pddp/controllers/ilqr.py:102-181,364-390
//   fit() loop bookkeeping          pddp/controllers/ilqr.py:298-314
// shared by the stand-alone accept kernel (controller.hip) and the fused
// line search + accept + derivative kernel (problem_kernels.hip).
#pragma once

#include "pddp_common.hpp"

namespace pddp {

template <typename T>
struct AcceptArgs {
  int B, N, n, m, A;
  const T* Zc;
  const T* Uc;
  const T* Jc;
  const T* gains;
  const int32_t* bwd_status;
  double tol, max_reg;
  int n_iterations;
  T* Z;
  T* U;
  T* gains_acc;
  T* J_opt;
  double* mu;
  double* delta;
  int32_t* state;
  int32_t* iter;
  uint8_t* active;
  uint8_t* fresh;
  int32_t* n_live;
};

constexpr double kMuMin = 1e-6;   // ilqr.py:94
constexpr double kDelta0 = 2.0;   // ilqr.py:95

constexpr int kAcceptThreads = 64;  // one wavefront: all 4096 blocks of the
                                    // bench batch are resident at once
constexpr int kMaxAlphas = 16;
constexpr int kLiveShards = 256;  // PDDP_LIVE_SHARDS of include/pddp_hip.h


// What the state machine reads of trajectory b; loaded side by side up front
// (the decision is a chain of memory latencies otherwise).
template <typename T>
struct AcceptIn {
  int bstat, iter;
  double mu, delta;
  T J_opt;
};
template <typename T>
PDDP_DEV AcceptIn<T> accept_load(const AcceptArgs<T>& a, int b) {
  AcceptIn<T> in;
  in.bstat = a.bwd_status[b];
  in.mu = a.mu[b];
  in.delta = a.delta[b];
  in.J_opt = a.J_opt[b];
  in.iter = a.iter[b];
  return in;
}

// argmin over the candidate costs with torch's semantics (first minimum, a
// NaN wins; ilqr.py:161), sequential form
template <typename T>
PDDP_DEV int argmin_first(const T* J, int A, T& Jm_out) {
  int amin = 0;
  T Jm = J[0];
#pragma unroll
  for (int i = 1; i < kMaxAlphas; ++i) {
    const bool take = (i < A) && (Jm == Jm) && (J[i] < Jm || J[i] != J[i]);
    amin = take ? i : amin;
    Jm = take ? J[i] : Jm;
  }
  Jm_out = Jm;
  return amin;
}

// One attempted trajectory: decides, writes mu / delta / state / J_opt / iter
// and the masks of the next round.  Returns the accepted candidate, -1 when
// the nominal stays; `fresh_out` = the nominal changed and the fit goes on.
template <typename T>
PDDP_DEV int accept_decide(const AcceptArgs<T>& a, int b, const AcceptIn<T>& in,
                           int amin, T J_new, bool& fresh_out) {
  int amin_out = -1;
  double mu = in.mu, delta = in.delta;
  int st;
  bool increase = false;
  if (in.bstat != 0) {
    increase = true;  // RuntimeError path                      (ilqr.py:140-145)
    st = PDDP_STATE_NOT_PD;
  } else {
    const T J_opt = in.J_opt;
    if (J_new < J_opt) {  // ilqr.py:166
      amin_out = amin;
      delta = (delta < 1.0 ? delta : 1.0) / kDelta0;  // _decrease_reg :369-374
      mu *= delta;
      if (mu <= kMuMin) mu = 0.0;
      const T rel = abs_(J_opt - J_new) / J_opt;
      st = (rel < (T)a.tol) ? PDDP_STATE_CONVERGED : PDDP_STATE_ACCEPTED;
      a.J_opt[b] = J_new;
    } else {
      increase = true;
      st = PDDP_STATE_REJECTED;
    }
  }
  if (increase) {  // _increase_reg                              (ilqr.py:376-390)
    delta = (delta > 1.0 ? delta : 1.0) * kDelta0;
    mu = (kMuMin > mu * delta) ? kMuMin : mu * delta;
    if (mu >= a.max_reg) st = PDDP_STATE_MAX_REG;
  }
  a.mu[b] = mu;
  a.delta[b] = delta;
  a.state[b] = st;
  // masks of the next round (fit loop, ilqr.py:298-314)
  uint8_t act = 0, fr = 0;
  if (st == PDDP_STATE_NOT_PD || st == PDDP_STATE_REJECTED) {
    act = 1;
  } else if (st == PDDP_STATE_ACCEPTED) {
    if (in.iter < a.n_iterations) {
      a.iter[b] = in.iter + 1;
      act = 1;
      fr = 1;
    }
  }
  a.active[b] = act;
  a.fresh[b] = fr;
  // sharded counter: 4096 adds on ONE word serialise at ~12 ns each (= 50 us)
  if (act && a.n_live != nullptr)
    atomicAdd(a.n_live + (b & (kLiveShards - 1)), 1);
  fresh_out = fr != 0;
  return amin_out;
}

}  // namespace pddp
