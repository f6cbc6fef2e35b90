// How fast does ONE wavefront that has its SIMD to itself issue vector
// instructions, as a function of how many INDEPENDENT dependency chains its
// instruction stream interleaves?  (The line search, the n = 4 sweep's roles and
// the record evaluation are all single wavefronts per SIMD at B = 4096.)
//   chains = 1: every instruction depends on the one before it
//   chains = 2, 4, 8: round-robin over that many independent accumulators
// and the same with W wavefronts on the SIMD (blockDim = 256 * W: waves w, w+4,
// ... share a SIMD), f32 FMA / f64 FMA / v_rcp_f32 / v_mul + v_add pairs.
// hipcc --offload-arch=gfx950 -O2 tools/probe/issue_rate_probe.hip -o /tmp/irp && /tmp/irp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// 64 instructions per asm block, CH chains
template <int OP, int CH>
__device__ __forceinline__ void body(float (&a)[8], double (&d)[8], float m, double md) {
  if (OP == 0) {  // v_fma_f32
    if (CH == 1) asm volatile(REP64("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(a[0]) : "v"(m));
    if (CH == 2) asm volatile(REP16(REP4("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2\n\t") ) : "+v"(a[0]), "+v"(a[1]) : "v"(m));
    if (CH == 4) asm volatile(REP16("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4\n\t")
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(m));
    if (CH == 8) asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                                       "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8\n\t"))
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));
  }
  if (OP == 1) {  // v_fma_f64
    if (CH == 1) asm volatile(REP64("v_fma_f64 %0, %0, %1, %1\n\t") : "+v"(d[0]) : "v"(md));
    if (CH == 2) asm volatile(REP16(REP4("v_fma_f64 %0, %0, %2, %2\n\tv_fma_f64 %1, %1, %2, %2\n\t")) : "+v"(d[0]), "+v"(d[1]) : "v"(md));
    if (CH == 4) asm volatile(REP16("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4\n\t")
                              : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "v"(md));
  }
  if (OP == 2) {  // v_rcp_f32
    if (CH == 1) asm volatile(REP64("v_rcp_f32 %0, %0\n\t") : "+v"(a[0]));
    if (CH == 2) asm volatile(REP16(REP4("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\t")) : "+v"(a[0]), "+v"(a[1]));
    if (CH == 4) asm volatile(REP16("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t")
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  }
  if (OP == 3) {  // v_mul_f32 with VOP2 encoding (e32)
    if (CH == 1) asm volatile(REP64("v_mul_f32_e32 %0, %1, %0\n\t") : "+v"(a[0]) : "v"(m));
    if (CH == 2) asm volatile(REP16(REP4("v_mul_f32_e32 %0, %2, %0\n\tv_mul_f32_e32 %1, %2, %1\n\t")) : "+v"(a[0]), "+v"(a[1]) : "v"(m));
    if (CH == 4) asm volatile(REP16("v_mul_f32_e32 %0, %4, %0\n\tv_mul_f32_e32 %1, %4, %1\n\tv_mul_f32_e32 %2, %4, %2\n\tv_mul_f32_e32 %3, %4, %3\n\t")
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(m));
  }
  if (OP == 4) {  // one chain, each instruction followed by an independent s_nop-free SALU op
    if (CH == 1) asm volatile(REP64("v_fma_f32 %0, %0, %1, %1\n\ts_add_u32 s20, s20, 1\n\t") : "+v"(a[0]) : "v"(m) : "s20", "scc");
  }
  if (OP == 5) {  // v_pk_fma_f32: two f32 FMAs per lane per instruction
    if (CH == 1) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %1\n\t") : "+v"(d[0]) : "v"(md));
    if (CH == 2) asm volatile(REP16(REP4("v_pk_fma_f32 %0, %0, %2, %2\n\tv_pk_fma_f32 %1, %1, %2, %2\n\t")) : "+v"(d[0]), "+v"(d[1]) : "v"(md));
  }
  if (OP == 6) {  // DPP-modified move + add (the quad kernels' cross-lane steps)
    if (CH == 1) asm volatile(REP64("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(a[0]));
    if (CH == 2) asm volatile(REP16(REP4("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a[0]), "+v"(a[1]));
  }
}

template <int OP, int CH>
__global__ void probe(float* out, long long* cyc, int iters, float m) {
  float a[8];
  double d[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + threadIdx.x * 1e-3f + i, d[i] = 1.0 + threadIdx.x * 1e-3 + i;
  const double md = m;
  body<OP, CH>(a, d, m, md);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) body<OP, CH>(a, d, m, md);
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int CH>
static void run(const char* name, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 200;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<OP, CH><<<blocks, threads>>>(out, cyc, iters, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP, CH><<<blocks, threads>>>(out, cyc, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += v;
  mean /= h.size();
  const double n_instr = 64.0 * iters * (OP == 4 ? 1 : 1);
  // clock64 = s_memtime (shader clock domain on gfx9); the event time gives ns
  printf("%-14s chains %d  waves/SIMD %d : %6.2f clk/instr  (%.2f ns/instr by events)\n", name, CH,
         waves_per_simd, mean / n_instr, ms * 1e6 / (64.0 * (iters)) );
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 1>("v_fma_f32", w);
    run<0, 2>("v_fma_f32", w);
    run<0, 4>("v_fma_f32", w);
    run<0, 8>("v_fma_f32", w);
    run<3, 1>("v_mul_f32_e32", w);
    run<3, 2>("v_mul_f32_e32", w);
    run<3, 4>("v_mul_f32_e32", w);
    run<1, 1>("v_fma_f64", w);
    run<1, 2>("v_fma_f64", w);
    run<1, 4>("v_fma_f64", w);
    run<2, 1>("v_rcp_f32", w);
    run<2, 2>("v_rcp_f32", w);
    run<2, 4>("v_rcp_f32", w);
    run<4, 1>("fma+s_add", w);
    run<5, 1>("v_pk_fma_f32", w);
    run<5, 2>("v_pk_fma_f32", w);
    run<6, 1>("v_add_dpp", w);
    run<6, 2>("v_add_dpp", w);
  }
  return 0;
}
