// Which SIMD does wavefront k of a workgroup land on?  (HW_REG_HW_ID: SIMD_ID
// in bits 5:4, WAVE_ID 3:0, CU_ID 11:8.)  One workgroup of 3 / 5 / 9 waves per
// launch, then a grid of them.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/wave_simd_probe.hip -o /tmp/wsp && /tmp/wsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d;
  (void)hipMalloc(&d, 4096 * 16 * 4);
  for (int waves : {3, 5, 9}) {
    for (int grid : {1, 512}) {
      (void)hipMemset(d, 0xff, 4096 * 16 * 4);
      probe<<<grid, waves * 64>>>(d);
      (void)hipDeviceSynchronize();
      std::vector<unsigned> h(grid * 16);
      (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
      for (int b : {0, grid - 1}) {
        printf("waves %d grid %d block %d: simd of wave k =", waves, grid, b);
        for (int k = 0; k < waves; ++k) printf(" %u", (h[b * 16 + k] >> 4) & 3);
        printf("   (cu %u)\n", (h[b * 16] >> 8) & 15);
        if (grid == 1) break;
      }
    }
  }
  return 0;
}
