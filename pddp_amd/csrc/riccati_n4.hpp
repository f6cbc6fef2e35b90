// riccati_n4.hpp - specialised backward sweep for n = 4, m = 1 (cartpole).
// Placeholder dispatch: routed to the generic kernel until the specialised
// kernel lands.
#pragma once

#include "riccati_generic.hpp"

namespace pddp {

template <typename T>
static int launch_n4(const RiccatiArgs<T>& a, hipStream_t st) {
  hipLaunchKernelGGL((riccati_generic_kernel<T, 8, 1>), dim3(a.B), dim3(kWave),
                     0, st, a);
  return launch_status();
}

}  // namespace pddp
