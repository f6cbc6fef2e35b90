// pddp_common.hpp - shared device/host helpers of libpddp_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/pddp_hip.h"

#define PDDP_DEV __device__ __forceinline__
#define PDDP_HD __host__ __device__ __forceinline__

namespace pddp {

constexpr int kWave = 64;  // CDNA wavefront

// Record layout of include/pddp_hip.h, computable on host and device.
struct RecLayout {
  int n = 0, m = 0;
  int oFz = 0, oLzz = 0, oFu = 0, oLuz = 0, oLz = 0, oLuu = 0, oLu = 0, oU = 0;
  int stride = 0, gstride = 0;
  PDDP_HD constexpr RecLayout(int n_, int m_) : n(n_), m(m_) {
    oFz = 0;
    oLzz = oFz + n * n;
    oFu = oLzz + n * n;
    oLuz = oFu + n * m;
    oLz = oLuz + m * n;
    oLuu = oLz + n;
    oLu = oLuu + m * m;
    oU = oLu + m;
    const int total = oU + m;
    stride = (total + 3) & ~3;
    gstride = m + m * n;
  }
};

template <typename T>
PDDP_DEV T clamp1(T v, T lo, T hi) {
  // utils/constraint.py:146-147: torch.min(torch.max(u, lo), hi); NaN stays NaN
  T t = v > lo ? v : lo;
  t = (v != v) ? v : t;
  T r = t < hi ? t : hi;
  return (t != t) ? t : r;
}

// the same clamp in 3 instructions for float (v_med3 drops a NaN, put it back)
PDDP_DEV float clamp_nan(float v, float lo, float hi) {
  const float r = __builtin_amdgcn_fmed3f(v, lo, hi);
  return (v != v) ? v : r;
}
PDDP_DEV double clamp_nan(double v, double lo, double hi) {
  double t = v > lo ? v : lo;
  t = (v != v) ? v : t;
  const double r = t < hi ? t : hi;
  return (t != t) ? t : r;
}

template <typename T>
PDDP_DEV bool is_finite(T v) {
  return __builtin_isfinite(v);  // one v_cmp_class
}

PDDP_DEV float sqrt_(float x) { return __fsqrt_rn(x); }
PDDP_DEV double sqrt_(double x) { return __dsqrt_rn(x); }
PDDP_DEV float sin_(float x) { return sinf(x); }
PDDP_DEV double sin_(double x) { return sin(x); }
PDDP_DEV float cos_(float x) { return cosf(x); }
PDDP_DEV double cos_(double x) { return cos(x); }
PDDP_DEV float abs_(float x) { return fabsf(x); }
PDDP_DEV double abs_(double x) { return fabs(x); }

// Optional HIP events attached to the NEXT sweep dispatch itself
// (hipExtLaunchKernelGGL): they time the kernel from its own start to its own
// end - what rocprofv3 --kernel-trace reports - without the stream's dispatch
// gaps that a pair of hipEventRecord calls around the launch includes.
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};
inline LaunchEvents& launch_events() {
  static thread_local LaunchEvents ev;
  return ev;
}
#define PDDP_LAUNCH(kernel, grid, block, shmem, st, ...)                      \
  do {                                                                        \
    ::pddp::LaunchEvents& ev_ = ::pddp::launch_events();                      \
    if (ev_.start != nullptr) {                                               \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, ev_.start,        \
                            ev_.stop, 0, __VA_ARGS__);                        \
      ev_ = ::pddp::LaunchEvents();                                           \
    } else {                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);        \
    }                                                                         \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace pddp
