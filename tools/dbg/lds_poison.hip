// Debug aid: leaves a chosen bit pattern in the LDS of every CU, so that a
// kernel launched next which reads LDS it has not written shows it (NaNs for
// pattern 0x7ff80000, which is a NaN as f32 and - doubled - as f64).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/dbg/lds_poison.hip -o tools/dbg/lds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void poison_kernel(unsigned pattern, int words, unsigned* out) {
  extern __shared__ unsigned s[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) s[i] = pattern;
  __syncthreads();
  // (keeps the stores alive, and the workgroup resident for a while so that
  // the next one goes to another CU)
  unsigned acc = 0;
  for (int r = 0; r < 64; ++r) acc += s[(threadIdx.x * 33 + r * 1024) % words];
  if (acc == 0x12345u) out[0] = acc;
}
extern "C" int lds_poison(unsigned pattern, unsigned* scratch, void* stream) {
  const int bytes = 160 * 1024;
  hipError_t e = hipFuncSetAttribute((const void*)poison_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(1024), bytes, (hipStream_t)stream, pattern,
                     bytes / 4, scratch);
  return (int)hipGetLastError();
}
