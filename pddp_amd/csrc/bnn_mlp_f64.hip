// bnn_mlp_f64.hip - the Bayesian network of the learned dynamics model in
// double precision (the reference runs in the dtype of its inputs:
// pddp/models/bnn/modules.py:287-386, :774-864), fused like bnn_mlp.hip:
//   y = W3 relu(M2 * (W2 relu(M1 * (W1 x + b1)) + b2)) + b3
// on v_mfma_f64_16x16x4_f64 (16 units x 16 rows x 4 k per instruction, 64
// cycles).  Not a widened copy of the f32 kernel: the f64 instruction is 16 x
// 16, W2 takes twice the registers, and what pays here is another deal.
//
// Weights-stationary, one workgroup of FOUR wavefronts per CU - one per SIMD,
// so that each may hold 512 registers: wavefront w owns the 16-unit blocks w,
// w + 4, w + 8, (w + 12) of both hidden layers and keeps ITS rows of W2 in
// registers for the whole kernel (H = 200: 13 blocks, 4 / 3 / 3 / 3; 416
// registers on the first wavefront).  Tiles of 16 rows:
//   1  layer 1 of the own blocks: 4 matrix instructions per block (K = 16:
//      inputs | zeros | bias slot; the A operand W1 | b1 from LDS), mask and
//      ReLU on the accumulators, to LDS in the order layer 2 reads them;
//      barrier
//   2  layer 2 of the own blocks: 4 NB instructions per block, the B operand
//      (one 32-byte LDS read per four instructions) shared by the wavefront's
//      blocks; mask and ReLU on the accumulators - which, with the k-slots
//      dealt as below, ARE the B operand of layer 3 for the block's 16 units:
//      4 more instructions per block leave this wavefront's partial outputs;
//      they go to LDS; barrier
//   3  sum of the four partial outputs (fixed order), + b3, rows stored.
// k-slots: instruction s of a group of four takes k = 16 S + kk + 4 s in slot
// kk (kk = lane >> 4).  The f64 C/D layout holds row g + 4 r in register r of
// lane group g (riccati_mfma16.hpp), so register q of lane (row j, g) of a
// block's accumulator is unit 16 ub + g + 4 q = the B operand of instruction
// s = q, slot kk = g: the hidden activations of layer 2 never leave the
// registers, and layer 1's go to LDS at position 4 g + q of their block - one
// 32-byte write, and one 32-byte read per (S, lane) in layer 2.
//
// JVP mode (the derivative rollout; see bnn_mlp.hip): rows in groups of 8 or
// 16 = one input row and its tangent rows; tangents pass without biases and
// through the ReLUs linearised at the group's first row.  Groups are whole
// inside a 16-row tile; rows k >= live of a group are neither read nor
// written.
#include <type_traits>

#include "pddp_common.hpp"

namespace pddp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct BnnMlpArgs64 {
  int R, P, in_dim, H, out_dim, live;
  const double* X;
  const double* W1;
  const double* b1;
  const double* MT1;  // layer-1 mask [P][H]
  const double* W2;
  const double* b2;
  const double* MT2;  // layer-2 mask [P][H]
  const double* W3;
  const double* b3;
  double* Y;
  const int32_t* live_rows;  // nullable: rows beyond *live_rows are left out
};

constexpr int kMlp64Threads = 256;
constexpr int kMlp64Tile = 16;    // rows per tile
constexpr int kMlp64W1Max = 16;   // W1 row (<= 15 inputs) | b1
constexpr int kMlp64MaxOut = 16;
constexpr int kMlp64W1Stride = 18;  // doubles: 144 B, odd in 16-byte units

template <int H>
struct Mlp64Shape {
  static constexpr int NB = (H + 15) / 16;   // 16-unit blocks
  static constexpr int NBW = (NB + 3) / 4;   // blocks of the first wavefront
  static constexpr int KP = 16 * NB + 2;     // row stride in LDS: odd in 16 B
  static constexpr size_t lds_doubles =
      (size_t)kMlp64Tile * KP            // h1
      + (size_t)16 * NB * kMlp64W1Stride  // W1 | b1
      + (size_t)16 * KP                  // W3
      + (size_t)4 * 64 * 4;              // partial outputs
};

PDDP_DEV f64x4 lds_read4(const double* p) {  // 16-byte aligned
  const f64x2 a = *reinterpret_cast<const f64x2*>(p);
  const f64x2 b = *reinterpret_cast<const f64x2*>(p + 2);
  return f64x4{a[0], a[1], b[0], b[1]};
}
PDDP_DEV void lds_write4(double* p, const f64x4& v) {
  *reinterpret_cast<f64x2*>(p) = f64x2{v[0], v[1]};
  *reinterpret_cast<f64x2*>(p + 2) = f64x2{v[2], v[3]};
}

// G: rows per (state, particle) in JVP mode (8 or 16; 0: plain inference)
template <int H, int G>
__global__ __launch_bounds__(kMlp64Threads) void bnn_mlp_f64_kernel(BnnMlpArgs64 a) {
  using S_ = Mlp64Shape<H>;
  constexpr int NB = S_::NB, NBW = S_::NBW, KP = S_::KP;
  static_assert(H % 4 == 0 && H <= 208, "H: multiple of 4, at most 208");
  static_assert(G == 0 || G == 8 || G == 16, "");
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  double* h1 = lds64;                              // [16][KP]
  double* w1s = h1 + kMlp64Tile * KP;              // [16 NB][18]
  double* w3s = w1s + 16 * NB * kMlp64W1Stride;    // [16][KP]
  double* part = w3s + 16 * KP;                    // [4][64][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int P = a.P, IN = a.in_dim, OUT = a.out_dim;
  int R = a.R;
  if (a.live_rows != nullptr) {
    const int lr = *a.live_rows;
    R = lr < R ? (lr > 0 ? lr : 0) : R;
  }
  const int ntiles = (R + kMlp64Tile - 1) / kMlp64Tile;
  const int my = (int)blockIdx.x < ntiles
                     ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x
                     : 0;
  if (my == 0) return;

  // position of k-slot (kk, s) inside its 16-block: k = kk + 4 s sits at 4 kk + s
  // ---- W1 | b1 and W3 to LDS, in operand order
  for (int idx = tid; idx < 16 * NB * 16; idx += kMlp64Threads) {
    const int u = idx >> 4, pos = idx & 15;
    const int c = (pos >> 2) + 4 * (pos & 3);  // input slot at position pos
    double v = 0.0;
    if (u < H) {
      if (c < IN) v = a.W1[(size_t)u * IN + c];
      else if (c == 15) v = a.b1[u];
    }
    w1s[u * kMlp64W1Stride + pos] = v;
  }
  for (int idx = tid; idx < 16 * 16 * NB; idx += kMlp64Threads) {
    const int o = idx / (16 * NB), r = idx - o * (16 * NB);
    const int S = r >> 4, pos = r & 15;
    const int k = 16 * S + (pos >> 2) + 4 * (pos & 3);
    w3s[o * KP + 16 * S + pos] = (o < OUT && k < H) ? a.W3[(size_t)o * H + k] : 0.0;
  }

  // rows of a tile as this lane sees them (lane j = tile row j)
  const int kin = G != 0 ? j % (G != 0 ? G : 1) : 0;  // row within its group
  const bool tangent = G != 0 && kin != 0;
  const bool in_use = G == 0 || kin < a.live;
  // first lane of this lane's group within the 64-bit lane mask
  const int first_lane = G != 0 ? ((lane & 0x30) | (j - kin)) : lane;

  const int nown = (NB - wave + 3) / 4;  // blocks of this wavefront (wave-uniform)

  auto run = [&](auto NOWN_) {
    constexpr int NOWN = decltype(NOWN_)::value;
    if constexpr (NOWN > 0) {
      // ---- this wavefront's rows of W2: lane (i = j, kk = g) holds
      // W2[16 ub + i][16 S + kk + 4 s]
      double a2[NOWN][NB][4];
      f64x4 binit[NOWN];
#pragma unroll
      for (int i = 0; i < NOWN; ++i) {
        const int ub = wave + 4 * i;
        const int u = 16 * ub + j;
        const bool uok = u < H;
        const double* w2row = a.W2 + (size_t)(uok ? u : 0) * H;
#pragma unroll
        for (int S = 0; S < NB; ++S)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (16 * S + 4 * s >= H) continue;  // (no lane has a k < H there)
            const int k = 16 * S + g + 4 * s;
            const double v = w2row[k < H ? k : 0];
            a2[i][S][s] = (uok && k < H) ? v : 0.0;
          }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 16 * ub + g + 4 * q;
          const double bb = a.b2[n < H ? n : 0];
          binit[i][q] = (n < H && !tangent) ? bb : 0.0;
        }
      }
      // mask, ReLU (or its linearisation at the group's first row)
      auto epilogue = [&](const f64x4& pre, const double (&m)[4]) {
        f64x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (G != 0) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(pre[q] > 0.0);
            const bool pos = (bal >> first_lane) & 1ull;
            v[q] = pos ? pre[q] * m[q] : 0.0;
          } else {
            v[q] = fmax(pre[q] * m[q], 0.0);
          }
        }
        return v;
      };
      auto masks_of = [&](const double* M, int p, int ub, double (&m)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 16 * ub + g + 4 * q;
          m[q] = M[(size_t)p * H + (n < H ? n : 0)];
        }
      };
      __syncthreads();  // W1, W3 staged

      for (int it = 0; it < my; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        const int mrow = tile * kMlp64Tile + j;
        const bool live = mrow < R && in_use;
        const int group = G != 0 ? mrow / (G != 0 ? G : 1) : mrow;
        const int p = live ? group % P : 0;
        // ---- layer 1
        double xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int c = g + 4 * s;
          const double v = a.X[(size_t)(live ? mrow : 0) * IN + (c < IN ? c : 0)];
          xb[s] = c < IN ? (live ? v : 0.0)
                         : (c == 15 ? (tangent ? 0.0 : 1.0) : 0.0);
        }
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
          const int ub = wave + 4 * i;
          double m1[4];
          masks_of(a.MT1, p, ub, m1);
          const f64x4 wa = lds_read4(w1s + (16 * ub + j) * kMlp64W1Stride + 4 * g);
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s], xb[s], acc, 0, 0, 0);
          lds_write4(h1 + j * KP + 16 * ub + 4 * g, epilogue(acc, m1));
        }
        __syncthreads();  // h1 complete
        // ---- layer 2
        f64x4 acc[NOWN];
#pragma unroll
        for (int i = 0; i < NOWN; ++i) acc[i] = binit[i];
#pragma unroll
        for (int S = 0; S < NB; ++S) {
          const f64x4 b = lds_read4(h1 + j * KP + 16 * S + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (16 * S + 4 * s >= H) continue;  // H = 200: 50 k-steps, not 52
#pragma unroll
            for (int i = 0; i < NOWN; ++i)
              acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i][S][s], b[s],
                                                            acc[i], 0, 0, 0);
          }
        }
        // ---- layer 3 of the own blocks' units
        f64x4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
          const int ub = wave + 4 * i;
          double m2[4];
          masks_of(a.MT2, p, ub, m2);
          const f64x4 h2 = epilogue(acc[i], m2);
          const f64x4 w3a = lds_read4(w3s + j * KP + 16 * ub + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            o = __builtin_amdgcn_mfma_f64_16x16x4f64(w3a[s], h2[s], o, 0, 0, 0);
        }
        lds_write4(part + (wave * 64 + lane) * 4, o);
        __syncthreads();  // partial outputs complete (and h1 free)
        // ---- sum, bias, store: thread (row = tid & 15, output = tid >> 4);
        // output oo of row r is register oo >> 2 of lane (r, oo & 3)
        {
          const int row = tid & 15, oo = tid >> 4;
          const int src = (((oo & 3) << 4) | row) * 4 + (oo >> 2);
          double y = part[src];
          y += part[64 * 4 + src];
          y += part[2 * 64 * 4 + src];
          y += part[3 * 64 * 4 + src];
          const int mr = tile * kMlp64Tile + row;
          const int kr = G != 0 ? row % (G != 0 ? G : 1) : 0;
          const bool lv = mr < R && (G == 0 || kr < a.live);
          if (oo < OUT && lv) {
            if (kr == 0) y += a.b3[oo];  // tangents: no bias
            a.Y[(size_t)mr * OUT + oo] = y;
          }
        }
      }
    }
  };
  if (nown == NBW) {
    run(std::integral_constant<int, NBW>());
  } else {
    run(std::integral_constant<int, NBW - 1>());
  }
}

template <int H, int G>
static int launch_bnn_mlp_f64(const BnnMlpArgs64& a, hipStream_t st) {
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
  const int ntiles = (a.R + kMlp64Tile - 1) / kMlp64Tile;
  const int grid = ntiles < cus ? ntiles : cus;  // persistent: one per CU
  constexpr size_t lds = sizeof(double) * Mlp64Shape<H>::lds_doubles;
  static_assert(lds <= 160 * 1024, "a workgroup's LDS");
  if (!attr_set[dev]) {  // more than 64 KB of dynamic LDS needs the opt-in
    const hipError_t e = hipFuncSetAttribute(
        (const void*)bnn_mlp_f64_kernel<H, G>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  PDDP_LAUNCH((bnn_mlp_f64_kernel<H, G>), dim3(grid), dim3(kMlp64Threads), lds,
              st, a);
  return launch_status();
}

template <int G>
static int launch_bnn_mlp_f64_h(const BnnMlpArgs64& a, hipStream_t st) {
  switch (a.H) {
    case 64: return launch_bnn_mlp_f64<64, G>(a, st);
    case 128: return launch_bnn_mlp_f64<128, G>(a, st);
    case 200: return launch_bnn_mlp_f64<200, G>(a, st);
  }
  return PDDP_E_UNSUPPORTED;
}

static int bnn_mlp_f64_impl(int R, int P, int group, int live, int in_dim, int H,
                            int out_dim, const double* X, const double* W1,
                            const double* b1, const double* MT1, const double* W2,
                            const double* b2, const double* MT2, const double* W3,
                            const double* b3, double* Y, const int32_t* live_rows,
                            void* stream) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !MT1 || !W2 || !b2 || !MT2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if (group != 0 &&
      ((group != 8 && group != 16 && group != 32) || R % group != 0 || live < 1 ||
       live > group))
    return PDDP_E_BADARG;
  if (in_dim >= kMlp64W1Max || out_dim > kMlp64MaxOut || group == 32)
    return PDDP_E_UNSUPPORTED;
  const BnnMlpArgs64 a{R, P, in_dim, H, out_dim, live, X, W1, b1, MT1,
                       W2, b2, MT2, W3, b3, Y, live_rows};
  hipStream_t st = (hipStream_t)stream;
  switch (group) {
    case 0: return launch_bnn_mlp_f64_h<0>(a, st);
    case 8: return launch_bnn_mlp_f64_h<8>(a, st);
    default: return launch_bnn_mlp_f64_h<16>(a, st);
  }
}

}  // namespace pddp

extern "C" {

int pddp_bnn_mlp_f64(int R, int P, int in_dim, int H, int out_dim,
                     const double* X, const double* W1, const double* b1,
                     const double* MT1, const double* W2, const double* b2,
                     const double* MT2, const double* W3, const double* b3,
                     double* Y, void* stream) {
  return pddp::bnn_mlp_f64_impl(R, P, 0, 1, in_dim, H, out_dim, X, W1, b1, MT1,
                                W2, b2, MT2, W3, b3, Y, nullptr, stream);
}

int pddp_bnn_mlp_rows_f64(int R, int P, int in_dim, int H, int out_dim,
                          const double* X, const double* W1, const double* b1,
                          const double* MT1, const double* W2, const double* b2,
                          const double* MT2, const double* W3, const double* b3,
                          double* Y, const int32_t* live_rows, void* stream) {
  return pddp::bnn_mlp_f64_impl(R, P, 0, 1, in_dim, H, out_dim, X, W1, b1, MT1,
                                W2, b2, MT2, W3, b3, Y, live_rows, stream);
}

int pddp_bnn_mlp_jvp_rows_f64(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const double* X,
                              const double* W1, const double* b1,
                              const double* MT1, const double* W2,
                              const double* b2, const double* MT2,
                              const double* W3, const double* b3, double* Y,
                              const int32_t* live_rows, void* stream) {
  if (group == 0) return PDDP_E_BADARG;
  return pddp::bnn_mlp_f64_impl(R, P, group, live, in_dim, H, out_dim, X, W1, b1,
                                MT1, W2, b2, MT2, W3, b3, Y, live_rows, stream);
}

}  // extern "C"
