// What does v_mfma_f64_16x16x4_f64 cost on gfx950?  One wavefront per SIMD
// (blockDim 256, one workgroup per CU), NCH independent accumulator chains,
// wall-clock cycles (s_memtime at 100 MHz scaled by the measured launch) and
// the whole-chip TFLOP/s it amounts to: the denominator of the f64 network
// kernel's roofline (csrc/bnn_mlp_f64.hip; bench.py MFMA_F64_PEAK_TFLOPS).
// hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f64_rate_probe.hip -o /tmp/mfp && /tmp/mfp
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  f64x4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f64x4{0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
static void run(int waves_per_simd) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double* out;
  hipMalloc(&out, sizeof(double) * cus * 256 * waves_per_simd);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NCH><<<cus * waves_per_simd, 256>>>(out, 10);
  hipEventRecord(e0);
  probe<NCH><<<cus * waves_per_simd, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 16 * NCH;  // instructions per wavefront
  const double flop = n * 2048.0 * cus * 4 * waves_per_simd;
  printf("chains %d, %d wavefront(s) per SIMD: %.3f ms, %.1f ns per instruction "
         "and SIMD, %.1f TFLOP/s\n", NCH, waves_per_simd, ms,
         ms * 1e6 / (n * waves_per_simd), flop / (ms * 1e-3) * 1e-12);
  hipFree(out);
}

int main() {
  run<1>(1);
  run<2>(1);
  run<4>(1);
  run<4>(2);
  return 0;
}
