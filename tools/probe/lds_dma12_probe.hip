// Probes where global_load_lds_dwordx3 puts each lane's 12 bytes in LDS.
// hipcc --offload-arch=gfx950 tools/probe/lds_dma12_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned buf[512];
  for (int q = threadIdx.x; q < 512; q += 64) buf[q] = 0xdeadbeefu;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)(src + threadIdx.x * 3),
      (__attribute__((address_space(3))) void*)buf, 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int q = threadIdx.x; q < 512; q += 64) out[q] = buf[q];
}
int main() {
  std::vector<unsigned> h(192);
  for (int i = 0; i < 192; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 192 * 4);
  hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), 192 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 272; ++i) {
    if (r[i] == 0xdeadbeefu) printf("  . ");
    else printf("%3u ", r[i]);
    if (i % 16 == 15) printf("\n");
  }
  return 0;
}
