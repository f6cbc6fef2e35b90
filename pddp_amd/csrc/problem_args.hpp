// problem_args.hpp - kernel argument blocks of the problem kernels (shared by
// problem_kernels.hip: IGNORE_UNCERTAINTY, and default_kernels.hip: the DEFAULT
// / upper-triangular Cholesky encoding).
#pragma once

#include "pddp_common.hpp"

namespace pddp {

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
  // Fused launch without records (pddp_search_accept_*, L == NULL, rec given
  // as scratch) only: 1 = the candidates are NOT kept - a step size other than
  // the full step writes nothing but its cost, the full step its compact rows
  // in `rec`; a winner other than the full step (1 accepted attempt in 20) is
  // rolled out a second time.  B A N (n + m) words of candidates are 82 MB at
  // B = 4096 - absorbed by the 256 MB Infinity Cache - and 1.3 GB at B = 65536,
  // where writing them out at ~2 TB/s WAS the launch (654 of its 654 us).
  int drop_candidates = 0;
};

}  // namespace pddp
