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
//
// JVP mode (the derivative rollout, ilqr.py:457-468 through
// utils/evaluation.py:203-235): rows come in groups of 8, 16 or 32 = one (state,
// particle) input and 7 / 15 / 31 tangent directions of it.  A tangent row goes through
// the same weights without biases, and through the ReLUs linearised at its
// group's primal row: d relu(m h) = m dh [m h > 0].  Groups are aligned to the
// 16-lane DPP rows of both the producer and the accumulator layout (data row =
// lane & 31), so the primal's pre-activation of the same unit arrives by
// `row_newbcast` moves (two lane reads for 32-row groups).  Forward mode
// replaces autograd's replicate-the-input pass; the caller (bnn_jvp.hip) sends
// 8-row groups: the input and the D + m mean / action directions - the
// Cholesky directions are per-particle multiples of the mean ones.
#include "pddp_common.hpp"

namespace pddp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BnnMlpArgs {
  int R, P, in_dim, H, out_dim;
  const float* X;
  const float* W1;
  const float* b1;
  const float* MT1;  // layer-1 mask, parity-split [P][2][H/2]: M1[p][2 s + h]
                     // at [p][h][s] - the order a producer lane consumes it
  const float* W2;
  const float* b2;
  const float* MT2;  // layer-2 mask [P][H] (as the framework holds it)
  const float* W3;
  const float* b3;
  float* Y;
};

constexpr int kMlpThreads = 512;
constexpr int kMlpTile = 32;     // rows per tile
constexpr int kMlpW1Max = 16;    // W1 row (<= 15 inputs) | b1, in LDS
constexpr int kMlpMaxOut = 16;

// unit index held by accumulator register r of lane-half h in block j
PDDP_DEV int unit_of(int j, int r, int h) {
  return 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h;
}

template <int H, int W1S>
constexpr size_t bnn_mlp_lds_floats() {
  return 2 * (H / 2) * 64 + H * W1S + 2 * ((H + 31) / 32) * kMlpMaxOut * 32;
}

// kMlpW1Stride: LDS stride of a W1 row | b1 (8: in_dim <= 7, 16: <= 15) - the
// producer wavefront's work is proportional to it
// value of the group's first row for this lane's unit set.  G = 16: lane 0 of
// this lane's 16-lane row (gfx90a+ DPP row_newbcast).  G = 32 (a whole tile is
// one group): lane 0 for the lanes of half 0, lane 32 for half 1.
template <int G>
PDDP_DEV float row_first(float v) {
  if constexpr (G == 8) {
    // two groups per 16-lane DPP row: lanes 0-7 take lane 0, lanes 8-15 lane 8
    const float lo = __int_as_float(__builtin_amdgcn_update_dpp(
        0, __float_as_int(v), 0x150, 0xf, 0xf, true));
    const float hi = __int_as_float(__builtin_amdgcn_update_dpp(
        0, __float_as_int(v), 0x158, 0xf, 0xf, true));
    return (threadIdx.x & 8) ? hi : lo;
  } else if constexpr (G == 16) {
    return __int_as_float(__builtin_amdgcn_update_dpp(
        0, __float_as_int(v), 0x150, 0xf, 0xf, true));
  } else {
    const int lo = __builtin_amdgcn_readlane(__float_as_int(v), 0);
    const int hi = __builtin_amdgcn_readlane(__float_as_int(v), 32);
    return __int_as_float((threadIdx.x & 32) ? hi : lo);
  }
}

// kJvpGroup = rows per (state, particle) in JVP mode (0: plain inference)
template <int H, int kMlpW1Stride, int kJvpGroup = 0>
__global__ __launch_bounds__(kMlpThreads) void bnn_mlp_kernel(BnnMlpArgs a) {
  constexpr bool JVP = kJvpGroup != 0;
  static_assert(kJvpGroup == 0 || kJvpGroup == 8 || kJvpGroup == 16 ||
                    kJvpGroup == 32, "");
  static_assert(H % 8 == 0 && H <= 224, "H: multiple of 8, at most 224");
  constexpr int KS = H / 2;          // MFMA steps of layer 2
  constexpr int NB = (H + 31) / 32;  // 32-unit blocks = consumer wavefronts
  static_assert(NB < kMlpThreads / 64, "one wavefront is the producer");
  // LDS (dynamic): two h1^T buffers, W1 | b1, two buffers of partial outputs
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* h1t = lds;                       // [2][KS * 64]
  float* w1b = h1t + 2 * KS * 64;         // [H][16]
  float* part = w1b + H * kMlpW1Stride;   // [2][NB * 16 * 32]
  constexpr int kH1 = KS * 64, kPart = NB * kMlpMaxOut * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int P = a.P, IN = a.in_dim, OUT = a.out_dim, R = a.R;

  for (int o = tid; o < H * kMlpW1Stride; o += kMlpThreads) {
    const int k = o / kMlpW1Stride, c = o - k * kMlpW1Stride;
    w1b[o] = c < IN ? a.W1[k * IN + c]
                    : (c == kMlpW1Stride - 1 ? a.b1[k] : 0.f);
  }
  __syncthreads();

  const int ntiles = (R + kMlpTile - 1) / kMlpTile;
  // tiles of this workgroup: blockIdx.x + i * gridDim.x, i < my
  const int my = blockIdx.x < ntiles
                     ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

  if (wave >= NB) {
    // =====================================================================
    // producer wavefront(s): layer 1 of tile i + 1 into h1t[(i + 1) & 1] and
    // the output rows of tile i - 1 from part[(i - 1) & 1], while the
    // consumers run tile i on the matrix cores.  Lane (row li, half lh)
    // computes the units k = 2 s + lh, four s at a time (one ds_write_b128).
    // =====================================================================
    const bool first_prod = wave == NB;  // extra wavefronts just keep step
    auto layer1 = [&](int i) {           // tile index in this workgroup's list
      if (!first_prod || i >= my) return;
      const int row = (blockIdx.x + i * gridDim.x) * kMlpTile + li;
      const bool live = row < R;
      const int p = live ? (JVP ? (row / (JVP ? kJvpGroup : 1)) % P : row % P) : 0;
      float x[kMlpW1Stride];
#pragma unroll
      for (int c = 0; c < kMlpW1Stride; ++c)
        x[c] = (live && c < IN) ? a.X[(size_t)row * IN + c] : 0.f;
      // multiplies the bias slot (tangent rows carry no bias)
      x[kMlpW1Stride - 1] =
          (JVP && (li & ((JVP ? kJvpGroup : 1) - 1)) != 0) ? 0.f : 1.f;
      f32x4* dst = reinterpret_cast<f32x4*>(h1t + (i & 1) * kH1) + (li * 2 + lh);
      // this lane's KS mask values, contiguous in the parity-split layout:
      // all requested up front (KS / 4 independent 16-B loads)
      f32x4 mk[KS / 4];
      {
        const f32x4* msrc =
            reinterpret_cast<const f32x4*>(a.MT1 + ((size_t)p * 2 + lh) * KS);
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) mk[q] = msrc[q];
      }
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 2 * (4 * q + e) + lh;
          const f32x4* wr =
              reinterpret_cast<const f32x4*>(w1b + k * kMlpW1Stride);
          float acc = 0.f;
#pragma unroll
          for (int c4 = kMlpW1Stride / 4 - 1; c4 >= 0; --c4) {
            const f32x4 w = wr[c4];
            // bias first (last slot), then the inputs: not the accumulation
            // order of a library GEMM bit for bit (that order is unspecified)
            acc = __builtin_fmaf(x[4 * c4 + 3], w[3], acc);
            acc = __builtin_fmaf(x[4 * c4 + 2], w[2], acc);
            acc = __builtin_fmaf(x[4 * c4 + 1], w[1], acc);
            acc = __builtin_fmaf(x[4 * c4 + 0], w[0], acc);
          }
          if constexpr (JVP) {
            // linearised at the group's primal row (for which this IS relu)
            v[e] = (row_first<JVP ? kJvpGroup : 16>(acc) * mk[q][e] > 0.f)
                       ? acc * mk[q][e] : 0.f;
          } else {
            v[e] = fmaxf(acc * mk[q][e], 0.f);
          }
        }
        dst[q * 64] = v;
      }
    };
    auto store_out = [&](int i) {  // sum the blocks' partial outputs, + b3
      if (!first_prod || i < 0) return;
      const int row0 = (blockIdx.x + i * gridDim.x) * kMlpTile;
      const float* pr = part + (i & 1) * kPart;
      for (int t = lane; t < 32 * OUT; t += 64) {
        const int rr = row0 + (t & 31), o = t >> 5;
        float y = (JVP && ((t & 31) & ((JVP ? kJvpGroup : 1) - 1)) != 0)
                      ? 0.f : a.b3[o];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj)
          y += pr[(jj * kMlpMaxOut + o) * 32 + (t & 31)];
        if (rr < R) a.Y[(size_t)rr * OUT + o] = y;
      }
    };
    layer1(0);
    __syncthreads();
    for (int i = 0; i < my; ++i) {
      store_out(i - 1);
      layer1(i + 1);
      __syncthreads();
    }
    store_out(my - 1);
    return;
  }

  // =======================================================================
  // consumer wavefront j: its 32 units of layer 2 - rows of W2 in registers
  // for the whole kernel - and its share of layer 3
  // =======================================================================
  const int j = wave;
  float a2[KS];
  float a3[16], b2r[16];
  {
    const int u = 32 * j + li;  // A operand: row i = li is unit u
    const bool uok = u < H;
    const float* w2row = a.W2 + (size_t)(uok ? u : 0) * H + lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float v = w2row[2 * s];  // (clamped address + select: no branch
      a2[s] = uok ? v : 0.f;         // per element)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = unit_of(j, r, lh);
      const int nc = n < H ? n : 0;
      // layer 3, step r: A[i = li = output][k-slot lh] = W3[li][n]
      const float w3 = a.W3[(size_t)(li < OUT ? li : 0) * H + nc];
      const float bb = a.b2[nc];
      a3[r] = (n < H && li < OUT) ? w3 : 0.f;
      b2r[r] = n < H ? bb : 0.f;
    }
  }
  __syncthreads();  // pairs with the producer's first barrier: h1t[0] ready

  for (int i = 0; i < my; ++i) {
    const int row0 = (blockIdx.x + i * gridDim.x) * kMlpTile;
    // mask of layer 2, requested before the MFMAs so that its latency is
    // theirs: registers 4 g .. 4 g + 3 are the units 32 j + 8 g + 4 lh +
    // (0..3), one 16-B load of the mask row each (clamped inside the row for
    // the padded units of the last block, whose weights are zero)
    const int row = row0 + li;
    const int p = row < R ? (JVP ? (row / (JVP ? kJvpGroup : 1)) % P : row % P) : 0;
    f32x4 m2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = 32 * j + 8 * g + 4 * lh;
      m2[g] = *reinterpret_cast<const f32x4*>(
          a.MT2 + (size_t)p * H + (n0 + 4 <= H ? n0 : 0));
    }
    // ---- layer 2 on the matrix cores
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x4* bsrc =
        reinterpret_cast<const f32x4*>(h1t + (i & 1) * kH1) + (li * 2 + lh);
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) {
      const f32x4 b4 = bsrc[q * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 0], b4[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 1], b4[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 2], b4[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 3], b4[3], acc, 0, 0, 0);
    }
    // accumulator register r of this lane: unit unit_of(j, r, lh), data row
    // li - bias, mask, ReLU in place; layer 3 straight from the accumulators
    f32x16 out = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float h2;
      if constexpr (JVP) {
        const float pre =
            acc[r] + ((li & ((JVP ? kJvpGroup : 1) - 1)) != 0 ? 0.f : b2r[r]);
        const float mm = m2[r >> 2][r & 3];
        h2 = (row_first<JVP ? kJvpGroup : 16>(pre) * mm > 0.f) ? pre * mm : 0.f;
      } else {
        h2 = fmaxf((acc[r] + b2r[r]) * m2[r >> 2][r & 3], 0.f);
      }
      out = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[r], h2, out, 0, 0, 0);
    }
    // out: register r of lane-half lh = output unit (r & 3) + 8 (r >> 2)
    // + 4 lh of data row li (partial sum over this block's units)
    float* pw = part + (i & 1) * kPart;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (o < OUT) pw[(j * kMlpMaxOut + o) * 32 + li] = out[r];
    }
    __syncthreads();
  }
}

template <int H, int W1S, int JVP = 0>
static int launch_bnn_mlp_w(const BnnMlpArgs& a, hipStream_t st) {
  // per device (a process may drive several GPUs): CU count queried once -
  // hipGetDeviceProperties costs ms - and the > 64 KB dynamic-LDS opt-in,
  // which is a per-device function attribute
  constexpr int kMaxDev = 16;
  static int cus_of[kMaxDev] = {};
  static bool attr_set[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev)
    return PDDP_E_UNSUPPORTED;
  if (cus_of[dev] == 0) {
    hipDeviceProp_t prop;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess
                      ? prop.multiProcessorCount
                      : 256;
  }
  const int cus = cus_of[dev];
  const int ntiles = (a.R + kMlpTile - 1) / kMlpTile;
  const int grid = ntiles < cus ? ntiles : cus;  // persistent: one per CU
  constexpr size_t lds = sizeof(float) * bnn_mlp_lds_floats<H, W1S>();
  if (!attr_set[dev]) {  // more than 64 KB of dynamic LDS needs the opt-in
    const hipError_t e = hipFuncSetAttribute(
        (const void*)bnn_mlp_kernel<H, W1S, JVP>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  PDDP_LAUNCH((bnn_mlp_kernel<H, W1S, JVP>), dim3(grid),
                     dim3(kMlpThreads), lds, st, a);
  return launch_status();
}

template <int H, int JVP = 0>
static int launch_bnn_mlp(const BnnMlpArgs& a, hipStream_t st) {
  return a.in_dim < 8 ? launch_bnn_mlp_w<H, 8, JVP>(a, st)
                      : launch_bnn_mlp_w<H, 16, JVP>(a, st);
}

}  // namespace pddp

extern "C" {

int pddp_bnn_mlp_f32(int R, int P, int in_dim, int H, int out_dim,
                     const float* X, const float* W1, const float* b1,
                     const float* MT1, const float* W2, const float* b2,
                     const float* MT2, const float* W3, const float* b3,
                     float* Y, void* stream) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !MT1 || !W2 || !b2 || !MT2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if (in_dim >= pddp::kMlpW1Max || out_dim > pddp::kMlpMaxOut)
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

int pddp_bnn_mlp_jvp_f32(int R, int P, int group, int in_dim, int H,
                         int out_dim, const float* X, const float* W1, const float* b1,
                         const float* MT1, const float* W2, const float* b2,
                         const float* MT2, const float* W3, const float* b3,
                         float* Y, void* stream) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !MT1 || !W2 || !b2 || !MT2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if ((group != 8 && group != 16 && group != 32) || R % group != 0)
    return PDDP_E_BADARG;
  if (in_dim >= pddp::kMlpW1Max || out_dim > pddp::kMlpMaxOut)
    return PDDP_E_UNSUPPORTED;
  const pddp::BnnMlpArgs a{R, P, in_dim, H, out_dim, X, W1, b1, MT1, W2,
                           b2, MT2, W3, b3, Y};
  hipStream_t st = (hipStream_t)stream;
  if (group == 8) {
    switch (H) {
      case 64: return pddp::launch_bnn_mlp<64, 8>(a, st);
      case 128: return pddp::launch_bnn_mlp<128, 8>(a, st);
      case 200: return pddp::launch_bnn_mlp<200, 8>(a, st);
    }
  } else if (group == 16) {
    switch (H) {
      case 64: return pddp::launch_bnn_mlp<64, 16>(a, st);
      case 128: return pddp::launch_bnn_mlp<128, 16>(a, st);
      case 200: return pddp::launch_bnn_mlp<200, 16>(a, st);
    }
  } else {
    switch (H) {
      case 64: return pddp::launch_bnn_mlp<64, 32>(a, st);
      case 128: return pddp::launch_bnn_mlp<128, 32>(a, st);
      case 200: return pddp::launch_bnn_mlp<200, 32>(a, st);
    }
  }
  return PDDP_E_UNSUPPORTED;
}

}  // extern "C"
