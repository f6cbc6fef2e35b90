// Does a wavefront's (or a SIMD's) f32 vector work overlap with its f32 MFMA
// work on gfx950?  Three loops of the same length per wavefront:
//   mode 0: 64 dependent v_mfma_f32_32x32x2_f32 per iteration
//   mode 1: 64 x K independent v_fma_f32 per iteration (K = 8)
//   mode 2: both, interleaved (K vector FMAs after every MFMA)
//   mode 3: two wavefronts per SIMD, one runs mode 0, the other mode 1
// Prints cycles per iteration (s_memtime) of wavefront 0.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 8;

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  // mode 3: waves 0..3 (one per SIMD) do MFMA, waves 4..7 do vector FMAs
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float v[K];
  for (int k = 0; k < K; ++k) v[k] = threadIdx.x * 1e-3f + k;
  const float a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      if (do_mfma) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if (do_valu) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      }
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int k = 0; k < K; ++k) r += v[k];
  for (int k = 0; k < 16; ++k) r += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x + 128] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / iters;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = (t1 - t0) / iters;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * (256 * 512 + 128));
  hipMemset(out, 0, sizeof(float) * (256 * 512 + 128));
  hipMalloc(&cyc, 16);
  const int iters = 200;
  long long h[2];
  const char* names[4] = {"MFMA only (4 waves/CU)", "vector FMA only (4 waves/CU)",
                          "MFMA + K vector FMAs each, same wave",
                          "MFMA waves and vector waves share the SIMDs"};
  for (int mode = 0; mode < 4; ++mode) {
    const int threads = mode == 3 ? 512 : 256;
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) probe<0><<<256, threads>>>(out, cyc, iters);
      if (mode == 1) probe<1><<<256, threads>>>(out, cyc, iters);
      if (mode == 2) probe<2><<<256, threads>>>(out, cyc, iters);
      if (mode == 3) probe<3><<<256, threads>>>(out, cyc, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("mode %d %-48s: %lld clock64 ticks per iteration (64 MFMA and / or %d FMA)"
           "%s\n", mode, names[mode], h[0], 64 * K,
           mode == 3 ? "" : "");
    if (mode == 3) printf("       (the vector wave of the pair: %lld)\n", h[1]);
  }
  return 0;
}
