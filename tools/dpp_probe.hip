// dpp_probe.hip - prints which source lane each DPP control used by
// riccati_n4.hpp reads from (run once on gfx950 to pin the lane algebra).
//   hipcc --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ int dpp(int v) {
  return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, true);
}

__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = dpp<0x124>(lane);              // row_ror:4
  out[1 * 64 + lane] = dpp<0x128>(lane);              // row_ror:8
  out[2 * 64 + lane] = dpp<0x12C>(lane);              // row_ror:12
  out[3 * 64 + lane] = dpp<(1 | 2 << 2 | 3 << 4 | 0 << 6)>(lane);  // quad_perm [1,2,3,0]
  out[4 * 64 + lane] = dpp<(2 | 3 << 2 | 0 << 4 | 1 << 6)>(lane);  // quad_perm [2,3,0,1]
  out[5 * 64 + lane] = dpp<(3 | 0 << 2 | 1 << 4 | 2 << 6)>(lane);  // quad_perm [3,0,1,2]
  const int l = lane & 15, i = l >> 2, j = l & 3;
  out[6 * 64 + lane] = __builtin_amdgcn_ds_bpermute(((lane & 48) | (j * 4 + i)) * 4, lane);
}

int main() {
  int* d;
  hipMalloc(&d, 7 * 64 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[7 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[7] = {"row_ror:4", "row_ror:8", "row_ror:12", "quad[1,2,3,0]",
                          "quad[2,3,0,1]", "quad[3,0,1,2]", "bpermute-transpose"};
  for (int r = 0; r < 7; ++r) {
    printf("%-20s", names[r]);
    for (int l = 16; l < 32; ++l) printf(" %2d", h[r * 64 + l]);
    printf("\n");
  }
  return 0;
}
