// riccati_defer.hip - the deferred rank-one n = 4 sweep (riccati_n4_defer.hpp)
// in its own translation unit: no SLP pairing (its FMAs take DPP operands,
// riccati_quad.hip) and matrix-instruction results in ordinary VGPRs
// (-amdgpu-mfma-vgpr-form: the 4x4x1 products feed vector code directly).
#include "riccati_n4_defer.hpp"

namespace pddp {

int launch_n4_defer_f32(const RiccatiArgs<float>& a, hipStream_t st,
                        bool fast_math) {
  return launch_n4_defer<float>(a, st, fast_math);
}
int launch_n4_defer_f64(const RiccatiArgs<double>& a, hipStream_t st,
                        bool fast_math) {
  return launch_n4_defer<double>(a, st, fast_math);
}

}  // namespace pddp

#ifdef PDDP_QP_STATS
extern "C" int pddp_debug_defer_stats(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::n4d::g_defer_stats), 64);
  if (reset) {
    unsigned long long z[8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(pddp::n4d::g_defer_stats), z, 64);
  }
  return 0;
}
extern "C" int pddp_debug_defer_seg(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::n4d::g_defer_seg), 256);
  if (reset) {
    unsigned long long z[32] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(pddp::n4d::g_defer_seg), z, 256);
  }
  return 0;
}
#endif
