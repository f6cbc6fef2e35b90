// bnn_mlp.hip - the Bayesian network of the learned dynamics model, fused:
//   y = W3 relu(M2 * (W2 relu(M1 * (W1 x + b1)) + b2)) + b3
// for R = (states x particles) rows, dropout masks M1, M2 fixed per
// (particle, unit) and shared by every state (pddp/models/bnn/modules.py:
// 462-483 mask cache, :550-583 CDropout.forward, :774-789 BSequential,
// :792-864 bayesian_model: fc -> dropout -> ReLU, ..., fc_out).
//
// This is the one dense contraction of the path: 8.6 MFLOP
// per state and time step at P = 100, H = 200, and the only place the matrix
// cores apply.  Done layer by layer with library GEMMs, the H-wide activations
// of R = 4 million rows make three round trips through HBM per time step; here
// they never leave the CU.
//
// Weights-stationary mapping.  A workgroup is 8 wavefronts; wavefront j < NB
// (NB = ceil(H / 32)) owns 32 hidden units of layer 2 and keeps ITS rows of
// W2 in registers for the whole kernel: lane (i = l & 31, h = l >> 5) holds
// W2[32 j + i][2 s + h], s < H / 2, exactly the A operand of
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bitwise an fmaf chain; the
// exact-f32 matrix rate of gfx950 is 1/16 of bf16).  The
// workgroup then streams 32-row tiles:
//   A  all 512 lanes: layer 1 (in_dim <= 8: plain FMAs) for the tile, written
//      to LDS transposed, in the order the B operand is read back (one
//      ds_read_b128 feeds four MFMAs);
//   B  wavefront j: h2^T[32 units][32 rows] = W2_j . h1^T, 100 MFMAs on one
//      accumulator tile; bias, mask, ReLU on the accumulator registers; the
//      result has the data row on the lane and the unit in the register, which
//      IS the B operand of the next MFMA (k-slot h of step i <-> unit
//      32 j + (i & 3) + 8 (i >> 2) + 4 h), so layer 3 takes it with W3
//      permuted to match: 16 more MFMAs, no LDS, no shuffles;
//   C  the NB partial outputs are summed from LDS, b3 added, rows stored.
#include "pddp_common.hpp"

namespace pddp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BnnMlpArgs {
  int R, P, in_dim, H, out_dim;
  const float* X;
  const float* W1;
  const float* b1;
  const float* MT1;  // [H][P] mask of layer 1, transposed; nullable (= 1)
  const float* W2;
  const float* b2;
  const float* MT2;
  const float* W3;
  const float* b3;
  float* Y;
};

constexpr int kMlpThreads = 512;
constexpr int kMlpTile = 32;     // rows per tile
constexpr int kMlpW1Stride = 16; // W1 row (<= 15 inputs) | b1, in LDS
constexpr int kMlpMaxOut = 16;

// unit index held by accumulator register r of lane-half h in block j
PDDP_DEV int unit_of(int j, int r, int h) {
  return 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h;
}

template <int H>
__global__ __launch_bounds__(kMlpThreads) void bnn_mlp_kernel(BnnMlpArgs a) {
  static_assert(H % 8 == 0 && H <= 224, "H: multiple of 8, at most 224");
  constexpr int KS = H / 2;          // MFMA steps of layer 2
  constexpr int NB = (H + 31) / 32;  // 32-unit blocks = working wavefronts
  __shared__ __attribute__((aligned(16))) float h1t[KS * 64];
  __shared__ __attribute__((aligned(16))) float w1b[H * kMlpW1Stride];
  __shared__ float part[NB * kMlpMaxOut * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int P = a.P, IN = a.in_dim, OUT = a.out_dim, R = a.R;

  for (int o = tid; o < H * kMlpW1Stride; o += kMlpThreads) {
    const int k = o / kMlpW1Stride, c = o - k * kMlpW1Stride;
    w1b[o] = c < IN ? a.W1[k * IN + c]
                    : (c == kMlpW1Stride - 1 ? a.b1[k] : 0.f);
  }

  // ---- this wavefront's share of W2, W3, b2: registers for the whole kernel
  const int j = wave;  // block of layer-2 units (wavefronts >= NB only help
                       // with layers 1 and the output)
  float a2[KS];
  float a3[16], b2r[16];
  {
    const int u = 32 * j + li;  // A operand: row i = li is unit u
#pragma unroll
    for (int s = 0; s < KS; ++s)
      a2[s] = (j < NB && u < H) ? a.W2[(size_t)u * H + 2 * s + lh] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = unit_of(j, r, lh);
      const bool ok = j < NB && n < H;
      // layer 3, step r: A[i = li = output][k-slot lh] = W3[li][n]
      a3[r] = (ok && li < OUT) ? a.W3[(size_t)li * H + n] : 0.f;
      b2r[r] = ok ? a.b2[n] : 0.f;
    }
  }
  __syncthreads();

  const int ntiles = (R + kMlpTile - 1) / kMlpTile;
  const int row_a = tid & 31, grp_a = tid >> 5;  // phase A: row, unit group
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * kMlpTile;
    // ---- A: layer 1 for 32 rows x H units, transposed into LDS
    {
      const int row = row0 + row_a;
      const bool live = row < R;
      const int p = live ? row % P : 0;
      float x[kMlpW1Stride];
#pragma unroll
      for (int c = 0; c < kMlpW1Stride; ++c)
        x[c] = (live && c < IN) ? a.X[(size_t)row * IN + c] : 0.f;
      x[kMlpW1Stride - 1] = 1.f;  // multiplies the bias slot
      for (int k = grp_a; k < H; k += 16) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w1b + k * kMlpW1Stride);
        float acc = 0.f;
#pragma unroll
        for (int c4 = kMlpW1Stride / 4 - 1; c4 >= 0; --c4) {
          const f32x4 w = wr[c4];
          // bias first (slot 15), then inputs in ascending order within a
          // quad: the accumulation order of a plain dot product is not
          // reproduced bit for bit (torch's addmm order is unspecified too)
          acc = __builtin_fmaf(x[4 * c4 + 3], w[3], acc);
          acc = __builtin_fmaf(x[4 * c4 + 2], w[2], acc);
          acc = __builtin_fmaf(x[4 * c4 + 1], w[1], acc);
          acc = __builtin_fmaf(x[4 * c4 + 0], w[0], acc);
        }
        const float m = a.MT1 != nullptr ? a.MT1[(size_t)k * P + p] : 1.f;
        const float v = fmaxf(acc * m, 0.f);
        const int s = k >> 1, h = k & 1;
        h1t[(((s >> 2) * 32 + row_a) * 2 + h) * 4 + (s & 3)] = v;
      }
    }
    __syncthreads();

    // ---- B: layer 2 on the matrix cores, layer 3 on its accumulators
    if (j < NB) {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const f32x4* bsrc = reinterpret_cast<const f32x4*>(h1t) + (li * 2 + lh);
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        const f32x4 b4 = bsrc[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 0], b4[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 1], b4[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 2], b4[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 3], b4[3], acc, 0, 0, 0);
      }
      // accumulator register r of this lane: unit unit_of(j, r, lh), data
      // row li - bias, mask, ReLU in place
      const int row = row0 + li;
      const int p = row < R ? row % P : 0;
      f32x16 out = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = unit_of(j, r, lh);
        const float m = (a.MT2 != nullptr && n < H) ? a.MT2[(size_t)n * P + p] : 1.f;
        const float h2 = fmaxf((acc[r] + b2r[r]) * m, 0.f);
        out = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[r], h2, out, 0, 0, 0);
      }
      // out: register r of lane-half lh = output unit (r & 3) + 8 (r >> 2)
      // + 4 lh of data row li (partial sum over this block's units)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < OUT) part[(j * kMlpMaxOut + o) * 32 + li] = out[r];
      }
    }
    __syncthreads();

    // ---- C: sum the blocks' partial outputs, add b3, store
    if (tid < 32 * OUT) {
      const int row = row0 + (tid & 31), o = tid >> 5;
      float y = a.b3[o];
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) y += part[(jj * kMlpMaxOut + o) * 32 + (tid & 31)];
      if (row < R) a.Y[(size_t)row * OUT + o] = y;
    }
    // (the next tile's phase A touches only h1t, its phase B waits at the
    // barrier above before writing `part` again)
  }
}

template <int H>
static int launch_bnn_mlp(const BnnMlpArgs& a, hipStream_t st) {
  static int cus = 0;  // queried once: hipGetDeviceProperties costs ms
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipGetDeviceProperties(&prop, dev) == hipSuccess)
              ? prop.multiProcessorCount
              : 256;
  }
  const int ntiles = (a.R + kMlpTile - 1) / kMlpTile;
  const int grid = ntiles < cus ? ntiles : cus;  // persistent: one per CU
  hipLaunchKernelGGL((bnn_mlp_kernel<H>), dim3(grid), dim3(kMlpThreads), 0, st,
                     a);
  return launch_status();
}

}  // namespace pddp

extern "C" {

int pddp_bnn_mlp_f32(int R, int P, int in_dim, int H, int out_dim,
                     const float* X, const float* W1, const float* b1,
                     const float* MT1, const float* W2, const float* b2,
                     const float* MT2, const float* W3, const float* b3,
                     float* Y, void* stream) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !W2 || !b2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if (in_dim >= pddp::kMlpW1Stride || out_dim > pddp::kMlpMaxOut)
    return PDDP_E_UNSUPPORTED;
  const pddp::BnnMlpArgs a{R, P, in_dim, H, out_dim, X, W1, b1, MT1, W2,
                           b2, MT2, W3, b3, Y};
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 64: return pddp::launch_bnn_mlp<64>(a, st);
    case 128: return pddp::launch_bnn_mlp<128>(a, st);
    case 200: return pddp::launch_bnn_mlp<200>(a, st);
  }
  return PDDP_E_UNSUPPORTED;
}

}  // extern "C"
