// riccati_quad.hip - the quad-mapped n = 4 sweep (riccati_n4_quad.hpp) in its
// own translation unit: it is compiled with -fno-slp-vectorize.  The SLP
// vectoriser pairs the kernel's FMAs into v_pk_fma_f32, which cannot take a
// DPP operand; every quad broadcast then costs a v_mov_b32_dpp of its own (58
// per step).  Unpaired, the broadcast folds into v_fmac_f32_dpp.
#include "riccati_n4_quad.hpp"

namespace pddp {

int launch_n4_quad_f32(const RiccatiArgs<float>& a, hipStream_t st,
                       bool fast_math, bool loop_always) {
  return launch_n4_quad<float>(a, st, fast_math, loop_always);
}
int launch_n4_quad_f64(const RiccatiArgs<double>& a, hipStream_t st,
                       bool fast_math, bool loop_always) {
  return launch_n4_quad<double>(a, st, fast_math, loop_always);
}

}  // namespace pddp

#ifdef PDDP_QP_STATS
extern "C" int pddp_debug_quad_stats(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::n4q::g_quad_stats), 64);
  if (reset) {
    unsigned long long z[8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(pddp::n4q::g_quad_stats), z, 64);
  }
  return 0;
}
#endif
