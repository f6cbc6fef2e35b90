// What does the irreducible dependent work of ONE step of the n = 4, m = 1
// backward sweep cost a wavefront that has its SIMD to itself?  (VERDICT round
// 3, task 2: "measure the floor of the sweep's chain".)
//
// The sweep of BASELINE.json configs[1] is N = 100 dependent steps; at B = 4096
// every design puts at most one busy wavefront on a SIMD, so a step costs
// (instructions on its busiest wavefront) x (issue interval of a lone
// wavefront) + what it waits for.  This probe runs the step of
// csrc/riccati_n4_elem.hpp (16 lanes per trajectory, plain recursion) 100 times
// with EVERY operand pre-staged in registers - no LDS reads, no record
// generator, no gains out, no barrier - in three cuts:
//
//   core        the 27 hand-scheduled products / reductions + the two
//               transposes (ds_bpermute) + the rank-one value update, the gain
//               taken as s = 1 / Quu (no BoxQP): the matrix part's floor
//   core+qp     + QpLean1 (the lean closed-form BoxQP) + (c, w): the floor of
//               the whole dependent chain V -> Quu -> BoxQP -> c -> V'
//   (round 4 also timed the retired four-role kernel's chain of eight
//    dependent v_mfma_f32_4x4x1: 431 cycles per step, profiles/r04_riccati_floor.txt)
//   qp only     QpLean1 + (c, w) alone, its inputs two FMAs away from its
//               outputs (the scalar chain that crosses a step in the deferred
//               form)
//
// 256 workgroups x 4 wavefronts (one per SIMD), clock64 around the 100 steps,
// mean over the wavefronts; events over 200 launches give the wall time.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fno-slp-vectorize \
//     -I pddp_amd/csrc tools/probe/riccati_floor_probe.hip -o /tmp/rfp && /tmp/rfp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "riccati_n4_elem.hpp"

using namespace pddp;
using n4e::f32x4;

template <int CUT>
__global__ __launch_bounds__(256) void floor_probe(float* out, long long* cyc,
                                                   int steps, float seed) {
  const int lane = threadIdx.x & 63;
  const int l = lane & 15, i = l >> 2, j = l & 3;
  const int tr_addr = ((lane & 48) | (j * 4 + i)) * 4;
  // a contracting step: F = 0.9 I + small, so that V stays bounded
  const float e = 1e-3f * (1.0f + seed * lane);
  f32x4 Fs = {0.9f, e, -e, 0.5f * e}, Fq = {0.9f, -e, e, 0.25f * e};
  const float fr = 0.1f + e, fc = 0.1f - e, Lzz = (i == j) ? 1.0f : 0.0f;
  const float Lzc = 0.01f, Luzr = 0.0f, Luu = 0.2f, Lu = 0.05f;
  const float lo = -10.0f, hi = 10.0f, reg = 1.0f;
  float V = (i == j) ? 1.0f : 0.0f, vc = 0.1f, kprev = 0.0f;
  float acc = 0.0f;
  __syncthreads();
  const long long t0 = clock64();
  if constexpr (CUT == 0 || CUT == 1) {
    for (int t = 0; t < steps; ++t) {
      const n4e::StepCore q =
          n4e::step_core(V, vc, fr, fc, Fs, Fq, Lzz, Lzc, Luzr, Luu, Lu);
      const float QzzT = n4::bperm(tr_addr, q.Qzz);
      const float Quzc = n4::bperm(tr_addr, q.Quzr);
      __builtin_amdgcn_sched_barrier(0);
      const float qp_Q =
          n4d::bsel(n4d::splat(n4d::sgn(q.Quu)), 1e-12f, q.Quu) + reg;
      float kt, sK;
      if constexpr (CUT == 1) {
        n4e::QpLean1 ql;
        ql.solve(kprev, qp_Q, q.Qu, lo, hi);
        kt = ql.x;
        sK = __int_as_float(n4d::splat(ql.free_w) & __float_as_int(ql.inv));
      } else {
        sK = __builtin_amdgcn_rcpf(qp_Q);
        kt = -(q.Qu * sK);
      }
      float c, wv;
      n4q::rank_one_coeffs(kt, sK, q.Quu, q.Qu, c, wv);
      kprev = kt;
      V = n4::fma_(0.5f, n4::opaque(q.Qzz + QzzT),
                   n4::mul_nc(c, n4::mul_nc(q.Quzr, Quzc)));
      vc = n4::fma_(wv, Quzc, q.Qzc);
    }
    acc += V + vc;
  } else {
    float Quu = 0.3f, Qu = 0.02f, c = -0.1f, wv = 0.01f;
    for (int t = 0; t < steps; ++t) {
      Quu = n4::fma_(c, 0.01f, 0.3f);
      Qu = n4::fma_(wv, 0.1f, 0.02f);
      const float qp_Q = n4d::bsel(n4d::splat(n4d::sgn(Quu)), 1e-12f, Quu) + reg;
      n4e::QpLean1 ql;
      ql.solve(kprev, qp_Q, Qu, lo, hi);
      const float kt = ql.x;
      const float sK =
          __int_as_float(n4d::splat(ql.free_w) & __float_as_int(ql.inv));
      n4q::rank_one_coeffs(kt, sK, Quu, Qu, c, wv);
      kprev = kt;
    }
    acc += c + wv;
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CUT>
static void run(const char* name) {
  const int blocks = 256, threads = 256, steps = 100;
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * threads * blocks);
  (void)hipMalloc(&cyc, sizeof(long long) * blocks * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 200; ++w)  // (the shader clock takes a few ms to ramp)
    floor_probe<CUT><<<blocks, threads>>>(out, cyc, steps, 0.001f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 200; ++w)
    floor_probe<CUT><<<blocks, threads>>>(out, cyc, steps, 0.001f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(),
                  hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= (double)h.size();
  printf("%-10s %7.1f cycles per step  (launch of 100 steps: %.2f us)\n", name,
         mean / steps, ms * 1e3 / 200);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0>("core");
  run<1>("core+qp");
  run<3>("qp only");
  return 0;
}
