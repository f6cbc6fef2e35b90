// Calibrates what bounds a latency-bound step on gfx950:
//   (a) dependent VALU chain, cycles per operation (one wave per SIMD)
//   (b) dependent chain with DPP moves (row_ror) in it
//   (c) two-wave LDS exchange: ds_write, lgkmcnt(0), s_barrier, ds_read, wait
//   (d) s_barrier alone
// hipcc --offload-arch=gfx950 tools/probe/sync_latency_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2000
__global__ void chain(float* out, float a, float b) {
  float x = threadIdx.x * 1e-3f;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x = __builtin_fmaf(x, a, b);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (ITERS * 32);
  out[1 + blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void chain_dpp(float* out, float a, float b) {
  float x = threadIdx.x * 1e-3f;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
      x = __builtin_fmaf(y, a, b);
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (ITERS * 16);
  out[1 + blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void chain_salu(float* out, float a, float b) {
  // v_cmp -> s_and -> v_cndmask dependent round trip
  float x = threadIdx.x * 1e-3f;
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      bool c = (x > a) & (x < b);
      x = c ? x + a : x - b;
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (ITERS * 16);
  out[1 + blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int MODE>
__global__ void exchange(float* out) {
  __shared__ float buf[2][2][64];
  const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float x = lane;
  buf[0][0][lane] = 0; buf[0][1][lane] = 0; buf[1][0][lane] = 0; buf[1][1][lane] = 0;
  __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {
      buf[i & 1][role][lane] = x;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      x = buf[i & 1][1 - role][lane] + 1.0f;
    } else if (MODE == 1) {
      asm volatile("s_barrier" ::: "memory");
    } else {
      buf[i & 1][role][lane] = x;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      x = buf[i & 1][role][lane] + 1.0f;   // own data: LDS round trip only
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / ITERS;
  out[1 + blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
  float* d;
  hipMalloc(&d, (1 + 1024 * 128) * 4);
  float h;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double per) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-40s %7.1f clock64 ticks, %7.1f ns per item\n", name, h, ms * 1e6 / per);
  };
  run("fma chain (1 wave/SIMD), per op", [&] { hipLaunchKernelGGL(chain, dim3(1024), dim3(64), 0, 0, d, 1.0001f, 0.5f); }, ITERS * 32.0);
  run("dpp+fma chain, per pair", [&] { hipLaunchKernelGGL(chain_dpp, dim3(1024), dim3(64), 0, 0, d, 1.0001f, 0.5f); }, ITERS * 16.0);
  run("cmp/s_and/cndmask/add chain, per round", [&] { hipLaunchKernelGGL(chain_salu, dim3(1024), dim3(64), 0, 0, d, 0.3f, 0.7f); }, ITERS * 16.0);
  run("2-wave LDS exchange + barrier, per iter", [&] { hipLaunchKernelGGL(exchange<0>, dim3(1024), dim3(128), 0, 0, d); }, (double)ITERS);
  run("s_barrier alone (2 waves), per iter", [&] { hipLaunchKernelGGL(exchange<1>, dim3(1024), dim3(128), 0, 0, d); }, (double)ITERS);
  run("LDS write+read own data, per iter", [&] { hipLaunchKernelGGL(exchange<2>, dim3(1024), dim3(128), 0, 0, d); }, (double)ITERS);
  return 0;
}
