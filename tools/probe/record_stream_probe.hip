// What HBM bandwidth does the backward sweep's ACCESS PATTERN allow?  Every
// wavefront streams one trajectory's records backwards in time, 1808 bytes per
// step (n = 14), through a four-slot LDS-DMA ring - and does nothing else.
//   layout 0: rec[b][t][S]  (the product's: a trajectory is contiguous)
//   layout 1: rec[t][b][S]  (a time step's records of all trajectories adjacent)
// hipcc --offload-arch=gfx950 tools/probe/record_stream_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
template <int LAYOUT, int RING>
__global__ __launch_bounds__(256) void stream(const float* rec, float* out, int B, int N, int S) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  float* ring = smem + wave * RING * 512;
  const int chunks = S / 4;
  const uint32_t q0 = (uint32_t)(lane % chunks) * 16u, q1 = (uint32_t)((lane + 64) % chunks) * 16u;
  auto dma = [&](int slot, int t) {
    const int tt = t < 0 ? 0 : t;
    const size_t rix = LAYOUT == 0 ? (size_t)b * (N + 1) + tt : (size_t)tt * B + b;
    const char* base = reinterpret_cast<const char*>(rec + rix * S);
    const uint32_t l = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(ring + slot * 512));
    lds_dma16(base, q0, l);
    lds_dma16(base, q1, l + 1024);
  };
  for (int s = 0; s < RING; ++s) dma(s, N - 1 - s);
  float acc = 0.f;
  int slot = 0;
  for (int t = N - 1; t >= 0; --t) {
    wait_vmcnt<(RING - 1) * 2>();
    acc += ring[slot * 512 + lane];
    dma(slot, t - RING);
    slot = slot + 1 == RING ? 0 : slot + 1;
  }
  wait_vmcnt<0>();
  if (acc == 123.456f) out[b] = acc;
}
// the n = 4 sweep's pattern: a wavefront streams the 192-byte records of FOUR
// trajectories (one full-wave DMA per step, 768 bytes used), B / 4 wavefronts
template <int RING>
__global__ __launch_bounds__(64) void stream_n4(const float* rec, float* out, int B, int N) {
  __shared__ __attribute__((aligned(16))) float ring[RING][256];
  const int lane = threadIdx.x;
  const int b0 = blockIdx.x * 4;
  int q = lane < 48 ? lane : lane - 48;
  const int tg = q / 12, c = q - tg * 12;
  const uint32_t off = (uint32_t)(tg * (N + 1) * 48 * 4 + c * 16);
  const char* base = reinterpret_cast<const char*>(rec + (size_t)b0 * (N + 1) * 48);
  auto dma = [&](int slot, int t) {
    const int tt = t < 0 ? 0 : t;
    const uint32_t l = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&ring[slot][0]));
    lds_dma16(base, off + (uint32_t)tt * 192u, l);
  };
  for (int s = 0; s < RING; ++s) dma(s, N - 1 - s);
  float acc = 0.f;
  int slot = 0;
  for (int t = N - 1; t >= 0; --t) {
    wait_vmcnt<RING - 1>();
    acc += ring[slot][lane];
    dma(slot, t - RING);
    slot = slot + 1 == RING ? 0 : slot + 1;
  }
  wait_vmcnt<0>();
  if (acc == 123.456f) out[b0] = acc;
}
// reference: plain coalesced streaming read of the same buffer
__global__ __launch_bounds__(256) void plain_read(const float4* src, float* out, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
// the ceiling proper (round 3): EIGHT independent 16-byte loads in flight per
// lane, non-temporal, block-contiguous tiles (MI355X_MICROARCH.md, HBM: a
// swept read wants >= 8 loads per lane outstanding; one grid-stride load per
// iteration - plain_read above - measures latency, not bandwidth)
typedef float f4v __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(256) void deep_read(const f4v* src, float* out, size_t n4) {
  float acc = 0.f;
  const size_t tile = (size_t)256 * K;
  for (size_t base = (size_t)blockIdx.x * tile; base + tile <= n4; base += (size_t)gridDim.x * tile) {
    f4v v[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
      v[k] = __builtin_nontemporal_load(src + base + (size_t)k * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const int B = 4096, N = 100, S = 452;
  const size_t words = (size_t)B * (N + 1) * S;
  float *rec, *out, *flush;
  hipMalloc(&rec, words * 4); hipMalloc(&out, B * 4); hipMalloc(&flush, 512u << 20);
  hipMemset(rec, 0, words * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f, sum = 0;
    for (int i = 0; i < 6; ++i) {
      hipMemsetAsync(flush, i, 512u << 20);  // evict the 256 MB infinity cache
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)B * N * S * 4;
    printf("%-44s avg %7.1f us  min %7.1f us  -> %.2f TB/s\n", name, sum / 5 * 1e3, best * 1e3,
           bytes / (sum / 5 * 1e-3) * 1e-12);
  };
  run("rec[b][t][S], ring 4 (the sweep's pattern)", [&] { hipLaunchKernelGGL((stream<0, 4>), dim3(B / 4), dim3(256), 4 * 4 * 2048, 0, rec, out, B, N, S); });
  run("rec[b][t][S], ring 8", [&] { hipLaunchKernelGGL((stream<0, 8>), dim3(B / 4), dim3(256), 4 * 8 * 2048, 0, rec, out, B, N, S); });
  run("rec[t][b][S], ring 4", [&] { hipLaunchKernelGGL((stream<1, 4>), dim3(B / 4), dim3(256), 4 * 4 * 2048, 0, rec, out, B, N, S); });
  run("rec[t][b][S], ring 8", [&] { hipLaunchKernelGGL((stream<1, 8>), dim3(B / 4), dim3(256), 4 * 8 * 2048, 0, rec, out, B, N, S); });
  {
    const double bytes4 = (double)B * N * 48 * 4;
    for (int pass = 0; pass < 2; ++pass) {
      float best = 1e9f, sum = 0;
      for (int i = 0; i < 6; ++i) {
        hipMemsetAsync(flush, i, 512u << 20);
        hipEventRecord(e0);
        if (pass == 0)
          hipLaunchKernelGGL((stream_n4<8>), dim3(B / 4), dim3(64), 0, 0, rec, out, B, N);
        else
          hipLaunchKernelGGL(plain_read, dim3(256 * 8), dim3(256), 0, 0, (const float4*)rec, out,
                             (size_t)(bytes4 / 16));
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i) { sum += ms; best = ms < best ? ms : best; }
      }
      printf("%-44s avg %7.1f us  min %7.1f us  -> %.2f TB/s\n",
             pass == 0 ? "n = 4 pattern: 4 x 192 B per wave-step, ring 8"
                       : "plain read of the same 78.6 MB",
             sum / 5 * 1e3, best * 1e3, bytes4 / (sum / 5 * 1e-3) * 1e-12);
    }
  }
  {
    float best = 1e9f, sum = 0;
    const size_t n4 = (size_t)B * N * S / 4;
    for (int i = 0; i < 6; ++i) {
      hipMemsetAsync(flush, i, 512u << 20);
      hipEventRecord(e0);
      hipLaunchKernelGGL(plain_read, dim3(256 * 16), dim3(256), 0, 0, (const float4*)rec, out, n4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-44s avg %7.1f us  min %7.1f us  -> %.2f TB/s\n", "plain coalesced float4 read (reference)",
           sum / 5 * 1e3, best * 1e3, (double)n4 * 16 / (sum / 5 * 1e-3) * 1e-12);
  }
  for (int grid : {256 * 4, 256 * 8, 256 * 16}) {
    float best = 1e9f, sum = 0;
    const size_t n4 = (size_t)B * N * S / 4;
    for (int i = 0; i < 6; ++i) {
      hipMemsetAsync(flush, i, 512u << 20);
      hipEventRecord(e0);
      hipLaunchKernelGGL((deep_read<8>), dim3(grid), dim3(256), 0, 0, (const f4v*)rec, out, n4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("8 x 16 B loads in flight per lane, nt, %5d blocks  avg %7.1f us  min %7.1f us  -> %.2f TB/s\n",
           grid, sum / 5 * 1e3, best * 1e3, (double)n4 * 16 / (sum / 5 * 1e-3) * 1e-12);
  }
  return 0;
}
