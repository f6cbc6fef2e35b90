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
// w + 4, w + 8 of both hidden layers and keeps ITS rows of W2 in registers for
// the whole kernel (H = 200: 300 registers).  H = 200 is 12 such blocks and a
// thirteenth with eight real units: that one is SHARED - its rows of W2 sit in
// LDS and wavefront w takes the k-steps S = w, w + 4, .. of its contraction
// (190 matrix instructions per SIMD and tile instead of 232 / 174 / 174 / 174
// with the block owned by one wavefront).  Tiles of 16 rows, two barriers:
//   1  layer 1 of the own blocks (two instructions per block for up to seven
//      inputs; H = 200: wavefronts 1 - 3 also take one of
//      wavefront 0's blocks each, 3 the shared block's - wavefront 0 is the
//      finisher of the previous tile meanwhile): 4
//      matrix instructions per block (K = 16: inputs | zeros | bias slot; the
//      A operand W1 | b1 from LDS), mask and ReLU on the accumulators, to LDS
//      in the order layer 2 reads them; barrier
//   2  layer 2 of the own blocks: 50 instructions per block, the B operand
//      (one 32-byte LDS read per four instructions, requested a k-step ahead)
//      shared by the wavefront's blocks; mask and ReLU on the accumulators -
//      which, with the k-slots dealt as below, ARE the B operand of layer 3
//      for the block's 16 units: 4 more instructions per block leave this
//      wavefront's partial outputs; they and its partial sums of the shared
//      block go to LDS; barrier
//   3  the finisher, wavefront 0, while the others are in layer 1 of the next
//      tile: the shared block's sum, mask, ReLU and layer 3, the sum of the
//      partial outputs (fixed order), + b3, rows stored.
// Requests are pure loads made where they cost nothing: the inputs and layer-1
// masks of tile i + 1 inside the layer-2 loop of tile i (their address
// arithmetic issues in the shadow of the matrix instructions), the layer-2
// masks of tile i + 1 right after their registers' last use; W2 is loaded
// with nothing between the load and the matrix instruction (see w2_of).
// Layout.  The f64 C/D layout holds row g + 4 r in register r of lane group g
// (riccati_mfma16.hpp).  Row i of every A operand is therefore unit (output)
// 4 (i mod 4) + i / 4 of its block, which makes register r of lane (row j, g)
// unit 16 ub + 4 g + r: a lane's four units - and, with k = 16 S + 4 kk + s in
// slot kk = lane >> 4 of instruction s, its four k, inputs and outputs - are
// consecutive: one 32-byte access each for a mask segment, a weight
// quadruple, an h1 segment, the outputs.  The accumulator of a block's layer 2
// (after mask and ReLU) IS the B operand of layer 3 for its 16 units: the
// hidden activations of layer 2 never leave the registers.  (The half k-step
// of H = 200 alone takes k = 16 S + kk + 4 s: two of its four instructions
// cover k = 192 .. 199, the other two are left out.)
// Measured (MI355X, 4.1 M rows, H = 200): 6.6 ms = 53 TFLOP/s = 0.67 of the
// f64 matrix peak (78.6; tools/probe/mfma_f64_rate_probe.hip reaches 69 - 73.5
// with nothing but matrix instructions), 3.6x the library GEMMs; the matrix
// pipe is busy 74 % of a wavefront's cycles (profiles/r04_mlp64_pmc.txt,
// phase by phase: tools/mlp64_marks.py).
//
// JVP mode (the derivative rollout; see bnn_mlp.hip): rows in groups of 8 or
// 16 = one input row and its tangent rows; tangents pass without biases and
// through the ReLUs linearised at the group's first row.  Groups are whole
// inside a 16-row tile; rows k >= live of a group are neither read nor
// written.
#include <atomic>
#include <type_traits>

#include "pddp_common.hpp"

namespace pddp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Debug build (-DPDDP_MLP64_MARKS, tools/mlp64_marks.py): the four wavefronts
// of workgroup 0 leave s_memtime at the phase boundaries of their sixth tile
#ifdef PDDP_MLP64_MARKS
__device__ long long g_mlp64_marks[4 * 8];
#define PDDP_MLP64_MARK(k)                                                   \
  do {                                                                        \
    if (blockIdx.x == 0 && it == 5 && lane == 0)                              \
      g_mlp64_marks[wave * 8 + (k)] = __builtin_readcyclecounter();          \
  } while (0)
#else
#define PDDP_MLP64_MARK(k) do { } while (0)
#endif

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
      + (size_t)8 * 64 * 4               // partial outputs | the shared block's partial sums
      + (size_t)16 * KP                  // the shared block's rows of W2
      + (size_t)16 * NB + 16;            // b2 (operand order), b3
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
// SMALL: in_dim <= 7 (cartpole 6, pendulum 4) - layer 1 is K = 8, inputs |
// bias in slot 7, on TWO instructions per block, their slots dealt as c = kk +
// 4 s (as the half k-step's), two input loads per lane instead of four.  (A
// template parameter: as a run-time branch the second code path cost 16
// spilled registers and 0.8 ms.)
template <int H, int G, bool SMALL>
__global__ __launch_bounds__(kMlp64Threads) void bnn_mlp_f64_kernel(BnnMlpArgs64 a) {
  using S_ = Mlp64Shape<H>;
  constexpr int NB = S_::NB, NBW = S_::NBW, KP = S_::KP;
  static_assert(H % 4 == 0 && H <= 208, "H: multiple of 4, at most 208");
  static_assert(G == 0 || G == 8 || G == 16, "");
  extern __shared__ __attribute__((aligned(16))) double lds64[];
  double* h1 = lds64;                              // [16][KP]
  constexpr int kH1 = kMlp64Tile * KP;
  double* w1s = h1 + kH1;                          // [16 NB][18]
  double* w3s = w1s + 16 * NB * kMlp64W1Stride;    // [16][KP]
  double* part = w3s + 16 * KP;                    // [4 + 4][64][4]
  constexpr int kPart = 8 * 64 * 4;
  double* w2c = part + kPart;                      // [16][KP]: W2 rows of the last block
  double* b2s = w2c + 16 * KP;                     // [16 NB] at position 4 g + q of its block
  double* b3s = b2s + 16 * NB;                     // [16]

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

  // ---- W1 | b1 and W3 to LDS as the A operand reads them: row i of a
  // 16-row panel is unit (output) rowperm(i) of its block
  auto rowperm = [](int i) { return 4 * (i & 3) + (i >> 2); };
  for (int idx = tid; idx < 16 * NB * 16; idx += kMlp64Threads) {
    const int r = idx >> 4, c = idx & 15;  // panel row, input slot
    const int u = (r & ~15) + rowperm(r & 15);
    double v = 0.0;
    if (u < H) {
      if (c < IN) v = a.W1[(size_t)u * IN + c];
      else if (c == 15) v = a.b1[u];
    }
    w1s[r * kMlp64W1Stride + c] = v;
  }
  for (int idx = tid; idx < 16 * 16 * NB; idx += kMlp64Threads) {
    const int i = idx / (16 * NB), k = idx - i * (16 * NB);
    const int o = rowperm(i);
    w3s[i * KP + k] = (o < OUT && k < H) ? a.W3[(size_t)o * H + k] : 0.0;
  }

  for (int idx = tid; idx < 16 * 16 * NB; idx += kMlp64Threads) {
    // the last block's rows of W2 as the A operand reads them
    const int i = idx / (16 * NB), k = idx - i * (16 * NB);
    const int u = 16 * (NB - 1) + rowperm(i);
    w2c[i * KP + k] = (u < H && k < H) ? a.W2[(size_t)u * H + k] : 0.0;
  }
  for (int idx = tid; idx < 16 * NB; idx += kMlp64Threads)
    b2s[idx] = idx < H ? a.b2[idx] : 0.0;
  if (tid < 16) b3s[tid] = tid < OUT ? a.b3[tid] : 0.0;

  // rows of a tile as this lane sees them (lane j = tile row j)
  const int kin = G != 0 ? j % (G != 0 ? G : 1) : 0;  // row within its group
  const bool tangent = G != 0 && kin != 0;
  const bool in_use = G == 0 || kin < a.live;
  // first lane of this lane's group within the 64-bit lane mask
  const int first_lane = G != 0 ? ((lane & 0x30) | (j - kin)) : lane;

  // COOP (H = 200: 13 blocks): the last block is dealt out by k - wavefront w
  // takes the k-steps S = w, w + 4, .. of its contraction - instead of making
  // one wavefront's fourth block: 190 matrix instructions per SIMD and tile,
  // not 232 / 174 / 174 / 174.  The four partial accumulators meet in LDS and
  // the finisher (wavefront 0, below) completes the block.
  // H % 16 == 8 (H = 200): the last k-step holds eight real k.  It alone deals
  // its k-slots as k = 16 S + kk + 4 s, so that instructions s = 0, 1 cover
  // them and s = 2, 3 are left out (50 k-steps a block, not 52); everywhere
  // else k = 16 S + 4 kk + s - four consecutive k per lane, one 32-byte read.
  static_assert(H % 16 == 0 || H % 16 == 8, "");
  constexpr bool HALF = (H % 16) == 8;
  auto half_step = [](int S) { return HALF && S == NB - 1; };
  auto skipped = [&](int S, int s) { return half_step(S) && s >= 2; };
  auto k_of = [&](int S, int s) {  // of this lane's slot kk = g
    return half_step(S) ? 16 * S + g + 4 * s : 16 * S + 4 * g + s;
  };
  // the B operand of k-step S: the activations at this lane's k-slots
  auto b_of = [&](const double* hrow, int S) {
    if (half_step(S)) {
      return f64x4{hrow[16 * S + g], hrow[16 * S + g + 4], 0.0, 0.0};
    }
    return lds_read4(hrow + 16 * S + 4 * g);
  };
  constexpr bool COOP = (NB % 4) == 1 && NB > 4;
  constexpr int CB = NB - 1;                 // the shared block
  constexpr int NOWN_MAX = COOP ? (NB - 1) / 4 : NBW;
  const int nown = COOP ? NOWN_MAX : (NB - wave + 3) / 4;  // (wave-uniform)

  auto run = [&](auto NOWN_) {
    constexpr int NOWN = decltype(NOWN_)::value;
    if constexpr (NOWN > 0) {
      // ---- this wavefront's rows of W2: lane (i = j, kk = g) holds
      // W2[16 ub + i][16 S + kk + 4 s]
      double a2[NOWN][NB][4];

      // Straight from memory, nothing in between: a value that a vector
      // instruction touches (a select for padding) must live in an ordinary
      // register, and of those there are 256 - the compiler then parks the
      // weights in accumulation registers and copies each one back before its
      // matrix instruction (two v_accvgpr_read per instruction, which on this
      // chip take matrix time: 110 cycles per instruction instead of 64).  A
      // loaded value that only matrix instructions read is allocated to an
      // accumulation register directly.  No padding is needed: a block owned
      // whole has 16 real units (a partial last block is the shared one, from
      // LDS), and H % 4 == 0 keeps every k of a kept k-step below H.
      static_assert(H % 16 == 0 || COOP, "a partial last block must be the shared one");
      auto w2_of = [&](int ub, int S, int s) {
        return a.W2[(size_t)(16 * ub + 4 * (j & 3) + (j >> 2)) * H + k_of(S, s)];
      };
      // b2 of a block's units as the accumulator holds them (zero on a
      // tangent row: a property of the lane)
      auto bias2_of = [&](int ub) {
        const f64x4 v = lds_read4(b2s + 16 * ub + 4 * g);
        return tangent ? f64x4{0.0, 0.0, 0.0, 0.0} : v;
      };
#pragma unroll
      for (int i = 0; i < NOWN; ++i) {
        const int ub = wave + 4 * i;
#pragma unroll
        for (int S = 0; S < NB; ++S)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (skipped(S, s)) continue;
            a2[i][S][s] = w2_of(ub, S, s);
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
      // (units 16 ub + 4 g .. + 3: one 32-byte load; H % 4 == 0 keeps it
      // aligned and whole inside the row, the padding units' address is
      // clamped - their weights are zero)
      auto masks_of = [&](const double* M, int p, int ub, double (&m)[4]) {
        const int n0 = 16 * ub + 4 * g;
        const f64x4 v = *reinterpret_cast<const f64x4*>(
            M + (size_t)p * H + (n0 < H ? n0 : 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = v[q];
      };
      // what a tile's rows are to this lane; the inputs and the layer-1 masks
      // of a tile are requested a tile ahead
      struct Row { int mrow, p, live; };  // (no padding: a copy stays in registers)
      auto row_of = [&](int it) {
        Row r;
        r.mrow = (blockIdx.x + it * gridDim.x) * kMlp64Tile + j;
        r.live = r.mrow < R && in_use;
        const int group = G != 0 ? r.mrow / (G != 0 ? G : 1) : r.mrow;
        r.p = r.live ? group % P : 0;
        return r;
      };
      // The inputs are REQUESTED only: nothing looks at a loaded value before
      // layer 1 of its tile.  (With the padding selects at the request the
      // compiler put each load under its own exec mask with an s_waitcnt
      // behind it: two round trips to HBM per tile, 3 - 4 k cycles.)  Slot c =
      // g + 4 s of the lane takes x[c] for c < in_dim, the bias slot's one,
      // zeros between - selected at the use.
      // COOP: wavefront 0 - the finisher - takes no part in layer 1; its blocks
      // 0, 4, 8 go to wavefronts 1, 2, 3 (and the shared block to 3): between
      // the two barriers of a tile the finisher's 1.6 k cycles and the others'
      // layer 1 then run side by side instead of one after the other
      static_assert(!COOP || NOWN_MAX == 3, "wavefront 0's blocks go to 1, 2, 3");
      const bool l1_own = !COOP || wave != 0;   // (wave-uniform)
      const bool l1_extra = COOP && wave != 0;
      // (mx: the layer-1 mask of the extra block on wavefronts 1 - 3, the
      // layer-2 mask of the shared block on wavefront 0 - one array, so that
      // the allocator sees one set of registers)
      double xraw[SMALL ? 2 : 4], m1[NOWN][4], m1c[4], mx[4];
      int xlive = 0;
      auto request_inputs = [&](const Row& r) {
        if (!l1_own) return;
        const double* xrow = a.X + (size_t)(r.live ? r.mrow : 0) * IN;
        if constexpr (SMALL) {
          xraw[0] = xrow[g < IN ? g : 0];
          xraw[1] = xrow[g + 4 < IN ? g + 4 : 0];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int c = 4 * g + s;
            xraw[s] = xrow[c < IN ? c : 0];
          }
        }
        xlive = r.live;
#pragma unroll
        for (int i = 0; i < NOWN; ++i) masks_of(a.MT1, r.p, wave + 4 * i, m1[i]);
        if (l1_extra) masks_of(a.MT1, r.p, 4 * (wave - 1), mx);
        if (COOP && wave == 3) masks_of(a.MT1, r.p, CB, m1c);
      };
      auto layer1_of = [&](double* dst, int ub, const double (&m)[4]) {
        const double* wrow = w1s + (16 * ub + j) * kMlp64W1Stride;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        if constexpr (SMALL) {
          // (the staged row holds zeros from in_dim to 14 and b1 at 15)
          const double w0 = wrow[g], w1 = wrow[g == 3 ? 15 : g + 4];
          const double x0 = (g < IN && xlive) ? xraw[0] : 0.0;
          const double x1 = g == 3 ? (tangent ? 0.0 : 1.0)
                                   : ((g + 4 < IN && xlive) ? xraw[1] : 0.0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, x0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, x1, acc, 0, 0, 0);
        } else {
          const f64x4 wa = lds_read4(wrow + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int c = 4 * g + s;
            const double xb = c < IN ? (xlive ? xraw[s] : 0.0)
                                     : ((c == 15 && !tangent) ? 1.0 : 0.0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s], xb, acc, 0, 0, 0);
          }
        }
        lds_write4(dst + j * KP + 16 * ub + 4 * g, epilogue(acc, m));
      };
      auto layer1 = [&](int it) {  // of tile `it`, from xb / m1 (requested)
        double* dst = h1;
        (void)it;
        if (!l1_own) return;
#pragma unroll
        for (int i = 0; i < NOWN; ++i) layer1_of(dst, wave + 4 * i, m1[i]);
        if (l1_extra) layer1_of(dst, 4 * (wave - 1), mx);
        if (COOP && wave == 3) layer1_of(dst, CB, m1c);
      };
      // Two barriers per tile: layer 1 | barrier | layer 2, 3 | barrier | the
      // finisher (wavefront 0) beside the others' layer 1 of the next tile.
      // Inputs and layer-1 masks are requested a tile ahead, the layer-2 masks
      // ahead of the matrix instructions that cover their latency.
      Row row = row_of(0);     // tile `it`
      request_inputs(row);
      __syncthreads();         // W1, W3 staged
      Row nxt = row;

      // the layer-2 masks, too, are requested a tile ahead - right after their
      // last use: requested at the top of their own tile, ahead of the matrix
      // instructions, the compiler moved the loads down to their use and the
      // epilogue waited out a round trip to L2 per block (3.9 k of a tile's
      // 20.7 k cycles, tools/mlp64_marks.py)
      double m2[NOWN][4];
#pragma unroll
      for (int i = 0; i < NOWN; ++i) masks_of(a.MT2, row.p, wave + 4 * i, m2[i]);
      if (COOP && wave == 0) masks_of(a.MT2, row.p, CB, mx);
      for (int it = 0; it < my; ++it) {
        PDDP_MLP64_MARK(0);
        layer1(it);
        PDDP_MLP64_MARK(1);
        __syncthreads();
        PDDP_MLP64_MARK(2);
        const double* h1r = h1;
        double* partw = part;
        // ---- layer 2
        f64x4 acc[NOWN];
        f64x4 accc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NOWN; ++i) acc[i] = bias2_of(wave + 4 * i);
        // (the B operand of k-step S + 1 and the shared block's A operand are
        // requested before the matrix instructions of k-step S: issued where
        // they are used, every k-step waited out an LDS round trip - the
        // wavefront issues in order and has no other to hide behind)
        f64x4 b = b_of(h1r + j * KP, 0);
#pragma unroll
        for (int S = 0; S < NB; ++S) {
          f64x4 bn = b, wc = b;
          if (S + 1 < NB) bn = b_of(h1r + j * KP, S + 1);
          // (the next tile's inputs and layer-1 masks: requested here, where
          // the instructions that form their addresses issue in the shadow of
          // the matrix instructions; their registers are free since layer 1)
          if (S == 1 && it + 1 < my) {
            nxt = row_of(it + 1);
            request_inputs(nxt);
          }
          const bool mine = COOP && (S & 3) == wave;  // (wave-uniform)
          if (mine) wc = b_of(w2c + j * KP, S);  // (same slots as the B operand)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (skipped(S, s)) continue;  // H = 200: 50 k-steps, not 52
#pragma unroll
            for (int i = 0; i < NOWN; ++i)
              acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i][S][s], b[s],
                                                            acc[i], 0, 0, 0);
          }
          if (mine) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if (skipped(S, s)) continue;
              accc = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[s], b[s], accc, 0, 0, 0);
            }
          }
          b = bn;
        }
        PDDP_MLP64_MARK(3);
        // ---- layer 3 of the own blocks' units
        f64x4 o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
          const int ub = wave + 4 * i;
          const f64x4 h2 = epilogue(acc[i], m2[i]);
          const f64x4 w3a = lds_read4(w3s + j * KP + 16 * ub + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            o = __builtin_amdgcn_mfma_f64_16x16x4f64(w3a[s], h2[s], o, 0, 0, 0);
        }
        lds_write4(partw + (wave * 64 + lane) * 4, o);
        if constexpr (COOP) lds_write4(partw + ((4 + wave) * 64 + lane) * 4, accc);
        // ---- layer 1 of the next tile, requests for the one after
        const Row cur = row;
        if (it + 1 < my) {
          row = nxt;
#pragma unroll
          for (int i = 0; i < NOWN; ++i)
            masks_of(a.MT2, nxt.p, wave + 4 * i, m2[i]);
        }
        PDDP_MLP64_MARK(4);
        __syncthreads();  // partial sums of tile it, h1 of tile it + 1
        PDDP_MLP64_MARK(5);
        // ---- the finisher (wavefront 0): the shared block's sum, its mask,
        // ReLU and layer 3; the sum of the partial outputs in a fixed order,
        // + b3; rows stored from the accumulator layout - output g + 4 q of
        // row j
        if (wave == 0) {
          f64x4 y = lds_read4(partw + lane * 4);
#pragma unroll
          for (int w = 1; w < 4; ++w) y += lds_read4(partw + (w * 64 + lane) * 4);
          if constexpr (COOP) {
            f64x4 pre = bias2_of(CB);
#pragma unroll
            for (int w = 0; w < 4; ++w)
              pre += lds_read4(partw + ((4 + w) * 64 + lane) * 4);
            const f64x4 h2 = epilogue(pre, mx);
            const f64x4 w3a = lds_read4(w3s + j * KP + 16 * CB + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s)
              y = __builtin_amdgcn_mfma_f64_16x16x4f64(w3a[s], h2[s], y, 0, 0, 0);
          }
          if (cur.live) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int oo = 4 * g + q;
              if (oo < OUT)
                a.Y[(size_t)cur.mrow * OUT + oo] = y[q] + (tangent ? 0.0 : b3s[oo]);
            }
          }
        }
        if (COOP && wave == 0 && it + 1 < my) masks_of(a.MT2, row.p, CB, mx);
        PDDP_MLP64_MARK(6);
      }
    }
  };
  if (nown == NOWN_MAX) {
    run(std::integral_constant<int, NOWN_MAX>());
  } else {
    run(std::integral_constant<int, NOWN_MAX - 1>());
  }
}

template <int H, int G, bool SMALL>
static int launch_bnn_mlp_f64_s(const BnnMlpArgs64& a, hipStream_t st) {
  constexpr int kMaxDev = 16;
  // (first use from several host threads: both caches are idempotent - every
  // writer stores the same value - and atomic, so a racing reader sees either
  // "unset" and repeats the query / the opt-in, or the value)
  static std::atomic<int> cus_of[kMaxDev] = {};
  static std::atomic<bool> attr_set[kMaxDev] = {};
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
        (const void*)bnn_mlp_f64_kernel<H, G, SMALL>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  PDDP_LAUNCH((bnn_mlp_f64_kernel<H, G, SMALL>), dim3(grid), dim3(kMlp64Threads), lds,
              st, a);
  return launch_status();
}

template <int H, int G>
static int launch_bnn_mlp_f64(const BnnMlpArgs64& a, hipStream_t st) {
  return a.in_dim <= 7 ? launch_bnn_mlp_f64_s<H, G, true>(a, st)
                       : launch_bnn_mlp_f64_s<H, G, false>(a, st);
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
  // (the masks are read 32 bytes at a time)
  if ((((uintptr_t)MT1 | (uintptr_t)MT2) & 31) != 0) return PDDP_E_BADARG;
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

#ifdef PDDP_MLP64_MARKS
int pddp_debug_mlp64_marks(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::g_mlp64_marks),
                                  sizeof(long long) * 32);
}
#endif

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
