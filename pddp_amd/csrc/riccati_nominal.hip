// riccati_nominal.hip - the sweeps that evaluate their derivative records
// themselves, from the nominal trajectory (pddp_sweep_nominal_*): the n = 4
// sweep of riccati_n4_elem.hpp (cartpole; f32 with the generator on partner
// wavefronts, f64 inline) and the 16 x 16 matrix-core sweep of
// riccati_mfma16_nominal.hpp (pendulum, double cartpole).  A translation unit
// of its own: no SLP pairing (the FMAs take DPP operands, riccati_quad.hip)
// and matrix-instruction results in ordinary VGPRs.
#include "riccati_n4_elem.hpp"
#include "riccati_mfma16_nominal.hpp"

extern "C" int pddp_sweep_nominal_f32(const pddp_problem* problem, int B, int N,
                                      const float* Z, const float* U,
                                      const float* u_min, const float* u_max,
                                      const double* reg, int branch,
                                      const uint8_t* active, uint8_t* fresh,
                                      float* gains, int32_t* status, float* L,
                                      float* J_opt, void* stream) {
  if (problem == nullptr || B <= 0 || N <= 0 || Z == nullptr || U == nullptr ||
      reg == nullptr || gains == nullptr || status == nullptr ||
      L == nullptr || J_opt == nullptr)
    return PDDP_E_BADARG;
  if (branch != PDDP_BRANCH_EIG && branch != PDDP_BRANCH_CHOLESKY)
    return PDDP_E_BADARG;
  pddp::RiccatiArgs<float> a;
  a.B = B; a.N = N; a.n = 4;
  a.rec = nullptr;
  a.u_min = u_min; a.u_max = u_max;
  a.reg = reg;
  a.branch = branch;
  a.active = active;
  a.gains = gains;
  a.status = status;
  const pddp::n4d::GenArgs<float> gen = {Z, U, L, J_opt, fresh};
  if (problem->model != PDDP_MODEL_CARTPOLE) {
    // pendulum, double cartpole: the 16 x 16 matrix-core sweep with its
    // records generated in the wavefront (riccati_mfma16_nominal.hpp)
    a.n = problem->state_size;
    return pddp::launch_m16_nominal<float>(*problem, a, gen,
                                           (hipStream_t)stream);
  }
  // riccati_n4_elem.hpp: its record generator inline, on wavefronts of its
  // own, or (auto) by batch
  const int choice = pddp::nominal_kernel_choice();
  return pddp::launch_n4_elem(*problem, a, gen, (hipStream_t)stream,
                              choice == 3 ? 0 : choice == 4 ? 1 : -1);
}

extern "C" int pddp_sweep_nominal_f64(const pddp_problem* problem, int B, int N,
                                      const double* Z, const double* U,
                                      const double* u_min, const double* u_max,
                                      const double* reg, int branch,
                                      const uint8_t* active, uint8_t* fresh,
                                      double* gains, int32_t* status, double* L,
                                      double* J_opt, void* stream) {
  if (problem == nullptr || B <= 0 || N <= 0 || Z == nullptr || U == nullptr ||
      reg == nullptr || gains == nullptr || status == nullptr ||
      L == nullptr || J_opt == nullptr)
    return PDDP_E_BADARG;
  if (branch != PDDP_BRANCH_EIG && branch != PDDP_BRANCH_CHOLESKY)
    return PDDP_E_BADARG;
  pddp::RiccatiArgs<double> a;
  a.B = B; a.N = N; a.n = problem->state_size;
  a.rec = nullptr;
  a.u_min = u_min; a.u_max = u_max;
  a.reg = reg;
  a.branch = branch;
  a.active = active;
  a.gains = gains;
  a.status = status;
  const pddp::n4d::GenArgs<double> gen = {Z, U, L, J_opt, fresh};
  if (problem->model == PDDP_MODEL_CARTPOLE) {
    // riccati_n4_elem.hpp in float64 (round 5): the inline form
    a.n = 4;
    return pddp::launch_n4_elem_f64(*problem, a, gen, (hipStream_t)stream);
  }
  return pddp::launch_m16_nominal<double>(*problem, a, gen,
                                          (hipStream_t)stream);
}

namespace pddp {
namespace n4e {
// elem_gains on `count` independent scalar problems, four per wavefront as in
// the sweep (pddp_boxqp_m1_lean_f32: the unit-test entry of the benched
// sweep's BoxQP).  x = k, free_mask = (K is not zeroed), status = PDDP_BWD_*.
__global__ __launch_bounds__(kWave) void elem_gains_kernel(
    int count, const float* x0, const float* Quu, const float* Qu,
    const float* reg, const float* lo, const float* hi, float* x,
    uint8_t* free_mask, int32_t* status_out, float* coeffs) {
  const int lane = threadIdx.x;
  const int p = blockIdx.x * kTrajW + (lane >> 4);
  const bool exists = p < count;
  const int pc = exists ? p : count - 1;
  int status = PDDP_BWD_OK;
  unsigned long long alive_m = __ballot(exists);
  const ElemGains<float> g = elem_gains<float>(
      x0[pc], Quu[pc], Qu[pc], reg[pc], lo[pc], hi[pc], lane, status, alive_m);
  if (exists && (lane & 15) == 0) {
    x[p] = g.kt;
    free_mask[p] = g.sK != 0.0f ? 1 : 0;
    status_out[p] = status;
    if (coeffs != nullptr) {
      coeffs[3 * p] = g.sK;
      coeffs[3 * p + 1] = g.c;
      coeffs[3 * p + 2] = g.wv;
    }
  }
}

}  // namespace n4e
}  // namespace pddp

extern "C" int pddp_boxqp_m1_lean_f32(int count, const float* x0,
                                      const float* Quu, const float* Qu,
                                      const float* reg, const float* lower,
                                      const float* upper, float* x,
                                      uint8_t* free_mask, int32_t* status,
                                      float* coeffs, void* stream) {
  if (count <= 0 || !x0 || !Quu || !Qu || !reg || !lower || !upper || !x ||
      !free_mask || !status)
    return PDDP_E_BADARG;
  PDDP_LAUNCH(pddp::n4e::elem_gains_kernel,
              dim3((count + pddp::n4e::kTrajW - 1) / pddp::n4e::kTrajW),
              dim3(pddp::kWave), 0, (hipStream_t)stream, count, x0, Quu, Qu,
              reg, lower, upper, x, free_mask, status, coeffs);
  return pddp::launch_status();
}

extern "C" int pddp_sweep_nominal_kernel(int which) {
  const int prev = pddp::nominal_kernel_choice();
  if (which == 0 || which == 3 || which == 4)
    pddp::nominal_kernel_choice() = which;
  return prev;
}

#ifdef PDDP_ELEM_MARKS
extern "C" int pddp_debug_elem_marks(long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::n4e::g_elem_marks), 64);
  long long z[8] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(pddp::n4e::g_elem_marks), z, 64);
  return 0;
}
#endif
#ifdef PDDP_WG_TIMELINE
extern "C" int pddp_debug_elem_timeline(long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::n4e::g_elem_timeline),
                      sizeof(long long) * 1024 * 12);
  return 0;
}
#endif
