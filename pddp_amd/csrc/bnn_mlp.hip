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
// workgroup then streams 32-row tiles, three stages in flight:
//   A  wavefront j, tile i + 1: layer 1 of ITS 32 units on the matrix cores
//      too (K = in_dim | bias slot: 4 or 8 MFMAs, W1 | b1 rows in registers) -
//      as plain FMAs it was a fifth of the kernel: on gfx950 the f32 MFMA and
//      the f32 VALU do not overlap, every vector instruction costs the SIMD
//      ~10 cycles of matrix time (measured by ablation) - mask and ReLU on the
//      accumulators, to LDS as held: four ds_write_b128;
//   B  wavefront j, tile i: h2^T[32 units][32 rows] = W2_j . h1^T, 100 MFMAs
//      on one accumulator tile that starts at the bias; the contraction runs
//      over the units in the order stage A left them (MFMA step 4 q + e, k-slot
//      h <-> unit 8 q + 4 h + e: one ds_read_b128 feeds four MFMAs); mask and
//      ReLU on the accumulator registers, which go to LDS as they are;
//   C  wavefront NB (the finisher), tile i - 1: layer 3 for ALL blocks on
//      v_mfma_f32_16x16x4_f32 (16 outputs x 16 rows x 4 units: half the
//      matrix-pipe time of the 32x32 form for out_dim <= 16), B operand = one
//      ds_read_b32 per instruction from B's registers-as-written (unit
//      4 s + kk of block j, row 16 hf + n sits at word n * 4 + kk of a
//      256-word group: conflict-free), + b3, rows stored.
// With H = 200 the seven blocks sit 2 / 2 / 2 / 1 on the four SIMDs and the
// finisher shares the fourth: 208 MFMAs of 64 cycles per SIMD and tile, against
// 232 when every block ran its own layer 3 (round 1, where a producer wavefront
// computing all of layer 1 was what the tile waited for - 18 % of the kernel).
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
#include <cstdlib>

#include "pddp_common.hpp"

namespace pddp {

// Debug build (-DPDDP_MLP_MARKS, tools/mlp_marks.py): the eight wavefronts of
// workgroup 0 leave s_memtime at the phase boundaries of their sixth tile
#ifdef PDDP_MLP_MARKS
__device__ long long g_mlp_marks[8 * 8];
#define PDDP_MLP_MARK(k)                                                       \
  do {                                                                          \
    if (blockIdx.x == 0 && i == 5 && lane == 0) {                               \
      g_mlp_marks[wave * 8 + (k)] = __builtin_readcyclecounter();              \
      if ((k) == 0 || (k) == 4) /* the chip-wide 100 MHz clock: words 5, 6 */   \
        g_mlp_marks[wave * 8 + 5 + (k) / 4] = wall_clock64();                   \
    }                                                                           \
  } while (0)
#else
#define PDDP_MLP_MARK(k) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// x = hi + mid + lo, three bf16 (24 significant bits): the parts of the
// bf16-split twin of layer 2 (PREC = 3 below).  Both subtractions are exact.
PDDP_DEV void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  const float r2 = r1 - (float)mid;
  lo = (__bf16)r2;
}

struct BnnMlpArgs {
  int R, P, in_dim, H, out_dim;
  const float* X;
  const float* W1;
  const float* b1;
  const float* MT1;  // layer-1 mask [P][H] (as the framework holds it)
  const float* W2;
  const float* b2;
  const float* MT2;  // layer-2 mask [P][H] (as the framework holds it)
  const float* W3;
  const float* b3;
  float* Y;
  const int32_t* live_rows;  // nullable: rows beyond *live_rows are left out
};

constexpr int kMlpThreads = 512;
constexpr int kMlpTile = 32;     // rows per tile
constexpr int kMlpW1Max = 16;    // W1 row (<= 15 inputs) | b1
constexpr int kMlpMaxOut = 16;

// unit index held by accumulator register r of lane-half h in block j
PDDP_DEV int unit_of(int j, int r, int h) {
  return 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// BAL (H = 200): the six blocks that share a SIMD with another block hand the
// last one or two 8-unit chunks of their layer-2 contraction to the finisher
// wavefront (whose SIMD holds one block only): 196 MFMAs per SIMD and tile
// instead of 208.  The finisher leaves partial accumulators in LDS; the owners
// add them one tile later, before their (deferred) mask / ReLU epilogue - no
// extra barrier.
//
// BAL = 2 (round 5): the LAST block holds 8 real units of 32.  Its layer 2
// runs on v_mfma_f32_16x16x4_f32 - 16 units (8 real) x 16 rows x 4 k per
// instruction of 32 cycles: 2 x 52 of them per tile = 52 MFMA-equivalents
// instead of 100 - with the B operand read as the other blocks read it (one
// ds_read_b128 of the h1^T buffer = four steps).  What that frees goes into the
// deal: blocks 0, 1, 4, 5 (two SIMDs with two full blocks each) hand their last
// THREE chunks away, blocks 0, 1 to the finisher as before and 4, 5 to the
// small block's wavefront; blocks 2, 3 and the small one keep theirs: 184 /
// 184 / 184 / 178 MFMA-equivalents per SIMD and tile instead of 196.  The
// finisher skips the 24 units of the last block that do not exist.
constexpr int kMlpGivers = 6;
constexpr int mlp_givers(int bal) { return bal == 2 ? 4 : (bal == 1 ? 6 : 0); }
// chunks a giver of BAL = 2 hands over: three level the SIMDs by the
// instruction count (184 / 184 / 184 / 178 MFMA-equivalents) and three is the
// measured optimum - inference at 4.1 M rows: 3.36 / 3.30 / 3.20 / 3.30 / 3.39 /
// 3.48 ms per launch for 1 .. 6 (-DPDDP_MLP_GIVE2=n builds, round 5)
#ifndef PDDP_MLP_GIVE2
#define PDDP_MLP_GIVE2 3
#endif
constexpr int kMlpGive2 = PDDP_MLP_GIVE2;
template <int H, int W1S, int BAL = 0, int PREC = 0>
constexpr size_t bnn_mlp_lds_floats() {
  // two h1^T buffers, two buffers of h2 (1024 words per block), BAL: two
  // buffers of six partial accumulator tiles.  PREC = 3: h1 as three bf16
  // planes [part][k-step of 16 units][half 2][row 32][8] = 256 words per
  // k-step and part
  return (PREC == 3 ? 2 * 3 * ((H + 15) / 16) * 256 : 2 * (H / 2) * 64) +
         2 * ((H + 31) / 32) * 1024 + 2 * mlp_givers(BAL) * 1024;
}

// kMlpW1Stride: inputs | zeros | bias slot of layer 1 (8: in_dim <= 7, 16: <=
// 15) = twice its MFMA steps.
// group_first_positive: is `v` of the first row of this lane's group positive?
// Groups are L consecutive lanes of a 32-lane half, the first at lane 0 (L need
// not divide 32: lanes past the last whole group get `false`).  One vector
// compare; its 64-bit lane mask keeps the bits of the groups' first lanes and
// a scalar multiply smears each over its L lanes - scalar instructions, which
// do not take matrix time (a vector instruction does: see above).
template <int L>
PDDP_DEV bool group_first_positive(float v) {
  constexpr unsigned first_lanes = [] {
    unsigned m = 0;
    for (int g = 0; g < 32 / L; ++g) m |= 1u << (g * L);
    return m;
  }();
  constexpr unsigned smear = L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
  const unsigned long long b = __builtin_amdgcn_ballot_w64(v > 0.f);
  const unsigned lo = ((unsigned)b & first_lanes) * smear;
  const unsigned hi = ((unsigned)(b >> 32) & first_lanes) * smear;
  return __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
}

// The same for the small block of BAL = 2, whose accumulators hold the rows
// 16 hf + (lane & 15) in two register sets (v0: hf = 0, v1: hf = 1) and a
// different unit in each 16-lane slice (only slices 0 and 1 hold real units):
// the 32-bit row mask of a slice is put together from the two ballots,
// smeared, and taken apart again - scalar instructions.
template <int L>
PDDP_DEV void group_first_positive_halves(float v0, float v1, bool& c0, bool& c1) {
  constexpr unsigned first_lanes = [] {
    unsigned m = 0;
    for (int g = 0; g < 32 / L; ++g) m |= 1u << (g * L);
    return m;
  }();
  constexpr unsigned smear = L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
  const unsigned b0 = (unsigned)__builtin_amdgcn_ballot_w64(v0 > 0.f);
  const unsigned b1 = (unsigned)__builtin_amdgcn_ballot_w64(v1 > 0.f);
  const unsigned r0 = (b0 & 0xFFFFu) | (b1 << 16);      // slice 0, rows 0 .. 31
  const unsigned r1 = (b0 >> 16) | (b1 & 0xFFFF0000u);  // slice 1
  const unsigned s0 = (r0 & first_lanes) * smear, s1 = (r1 & first_lanes) * smear;
  c0 = __builtin_amdgcn_inverse_ballot_w64(
      (unsigned long long)((s0 & 0xFFFFu) | (s1 << 16)));
  c1 = __builtin_amdgcn_inverse_ballot_w64(
      (unsigned long long)((s0 >> 16) | (s1 & 0xFFFF0000u)));
}

// kJvpGroup = rows per (state, particle) in memory in JVP mode (0: plain
// inference); kJvpLive <= kJvpGroup of them are in use (the input row and the
// tangent rows that exist: 1 + D + m), the rest is padding that is neither
// read nor written.  A tile takes 32 / kJvpLive whole groups - with cartpole's
// 6 live rows of 8 that is 5 groups per tile instead of 4: a fifth fewer
// tiles.
// PREC = 3: the BF16-SPLIT TWIN of layer 2 (opt-in, pddp_bnn_mlp_precision):
// W2 and the layer-1 activations as three bf16 parts each, a product to f32
// accuracy from six v_mfma_f32_32x32x16_bf16 (hi.hi, hi.mid, mid.hi, mid.mid,
// hi.lo, lo.hi; 32 cycles for 16 k against 64 cycles for 2 k of the exact-f32
// instruction): 78 matrix instructions of 32 cycles per tile and block instead
// of 100 of 64.  The accumulator layout of the two instructions is the same,
// so mask, ReLU, the h2 buffers and layer 3 (exact f32, as layer 1) are
// untouched.  Not bit-exact f32: products are exact, the sums are rounded in
// another order (and the lowest product terms, 2^-24 and below, are dropped).
template <int H, int kMlpW1Stride, int kJvpGroup = 0, int kJvpLive = kJvpGroup,
          int BAL = 0, int PREC = 0>
__global__ __launch_bounds__(kMlpThreads) void bnn_mlp_kernel(BnnMlpArgs a) {
  static_assert(!BAL || H == 200, "the balanced roles are laid out for 7 blocks");
  static_assert(BAL != 2 || H % 32 == 8, "the small block holds 8 units");
  static_assert(PREC == 0 || (PREC == 3 && !BAL), "");
  constexpr int KS16 = (H + 15) / 16;  // PREC = 3: k-steps of layer 2
  constexpr bool JVP = kJvpGroup != 0;
  constexpr int G = JVP ? kJvpGroup : 1;  // rows per (state, particle)
  constexpr int LIVE = JVP ? kJvpLive : 1;
  constexpr int GPT = JVP ? kMlpTile / LIVE : kMlpTile;  // groups per tile
  constexpr int TROWS = GPT * G;                         // memory rows per tile
  static_assert(kJvpGroup == 0 || kJvpGroup == 8 || kJvpGroup == 16 ||
                    kJvpGroup == 32, "");
  static_assert(kJvpLive >= 0 && kJvpLive <= kJvpGroup, "");
  static_assert(H % 8 == 0 && H <= 224, "H: multiple of 8, at most 224");
  constexpr int KS = H / 2;          // MFMA steps of layer 2
  constexpr int NQ = KS / 4;         // 8-unit chunks of layer 1
  constexpr int NB = (H + 31) / 32;  // 32-unit blocks = consumer wavefronts
  constexpr int kWavesMlp = kMlpThreads / 64;
  static_assert(NB < kWavesMlp, "one wavefront is the finisher");
  static_assert(kWavesMlp == 8, "layer-1 chunks are dealt out modulo 8");
  // LDS (dynamic): two h1^T buffers, two h2 buffers
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* h1t = lds;                       // [2][KS * 64]
  constexpr int kH1 = PREC == 3 ? 3 * KS16 * 256 : KS * 64;  // words per buffer
  float* h2b = h1t + 2 * kH1;             // [2][NB * 1024]
  constexpr int kH2 = NB * 1024;
  float* part = h2b + 2 * kH2;            // BAL: [2][kMlpGivers * 1024]
  constexpr int kPart = mlp_givers(BAL) * 1024;
  constexpr bool SB = BAL == 2;  // the small last block on 16 x 16 x 4 tiles
  // BAL: the chunks q >= q_own(wave) of a block's contraction are the
  // finisher's; block 3 shares its SIMD with the finisher and keeps all of its
  // own, blocks 0 .. 2 give two chunks, blocks 4 .. 6 one
  // BAL = 2: blocks 0, 1 (to the finisher) and 4, 5 (to the small block's
  // wavefront) give three chunks each
  auto q_own = [](int w) {
    if (BAL == 2) return (w == 0 || w == 1 || w == 4 || w == 5) ? NQ - kMlpGive2 : NQ;
    return !BAL ? NQ : (w == 3 ? NQ : (w < 3 ? NQ - 2 : NQ - 1));
  };
  auto is_giver = [](int w) {
    return BAL == 2 ? (w == 0 || w == 1 || w == 4 || w == 5) : (BAL == 1 && w != 3);
  };
  auto giver_index = [](int w) {  // slot of a giver's partial tile
    return BAL == 2 ? (w < 2 ? w : w - 2) : (w < 3 ? w : w - 1);
  };
  // BAL = 2, the two takers: the last three chunks of two blocks' contraction
  // for one tile - A: lane (i = li, h = lh) holds W2[32 jb + i][8 q + 4 h + e]
  // like the owner itself would; the partial accumulator tiles go to LDS in
  // the owner's layout
  auto taker_weights = [&](int jb, float (&w)[4 * kMlpGive2]) {
    const int li_ = threadIdx.x & 31, lh_ = (threadIdx.x & 63) >> 5;
    const int u = 32 * jb + li_;
    const bool uok = u < H;
    const float* w2row = a.W2 + (size_t)(uok ? u : 0) * H + 4 * lh_;
#pragma unroll
    for (int c = 0; c < kMlpGive2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = w2row[8 * (NQ - kMlpGive2 + c) + e];
        w[4 * c + e] = uok ? v : 0.f;
      }
  };
  auto taker_partials = [&](int i, const float (&w0)[4 * kMlpGive2],
                            const float (&w1)[4 * kMlpGive2],
                            int slot0) {
    const int ln = threadIdx.x & 63;
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(h1t + (i & 1) * kH1) +
                        ((ln & 31) * 2 + (ln >> 5));
    f32x4 b[kMlpGive2];
#pragma unroll
    for (int c = 0; c < kMlpGive2; ++c) b[c] = bsrc[(NQ - kMlpGive2 + c) * 64];
    f32x4* pw = reinterpret_cast<f32x4*>(part + (i & 1) * kPart) + ln;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < kMlpGive2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(
              gi == 0 ? w0[4 * c + e] : w1[4 * c + e], b[c][e], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pw[((slot0 + gi) * 4 + g) * 64] = f32x4{acc[4 * g], acc[4 * g + 1],
                                                acc[4 * g + 2], acc[4 * g + 3]};
    }
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int P = a.P, IN = a.in_dim, OUT = a.out_dim;
  int R = a.R;
  if (a.live_rows != nullptr) {  // (one scalar load: the count is device data)
    const int lr = *a.live_rows;
    R = lr < R ? (lr > 0 ? lr : 0) : R;
  }

  const int ntiles = (R + TROWS - 1) / TROWS;
  // tile row rt (MFMA column) of tile `tile` -> memory row; rows of padding
  // lanes (past the last whole group) and past R are dead
  struct RowOf { int mrow, group; bool live, tangent; };
  auto row_of = [&](int tile, int rt) {
    RowOf r;
    if constexpr (!JVP) {
      r.mrow = tile * kMlpTile + rt;
      r.group = r.mrow;
      r.tangent = false;
      r.live = r.mrow < R;
    } else {
      const int g = rt / LIVE, k = rt - g * LIVE;
      r.group = tile * GPT + g;
      r.mrow = r.group * G + k;
      r.tangent = k != 0 || g >= GPT;
      r.live = g < GPT && r.mrow < R;
    }
    return r;
  };
  // tiles of this workgroup: blockIdx.x + i * gridDim.x, i < my (my >= 1: the
  // grid is never larger than the number of tiles)
  const int my = blockIdx.x < ntiles
                     ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  if (my == 0) return;  // (live_rows: fewer tiles than the launch was sized for)
  // LDS only: the tile barrier must not wait for the finisher's row stores
  auto tile_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // barriers of a workgroup: one after the first tile's layer 1, then one per
  // iteration; BAL runs two more iterations (epilogues and layer 3 trail by
  // one tile each)
  const int iters = my + (BAL ? 2 : 0);
  if (wave > NB) {
    // spare wavefronts (H < 200) only keep step
    tile_barrier();
    for (int i = 0; i < iters; ++i) tile_barrier();
    return;
  }

  if (wave == NB) {
    // =====================================================================
    // finisher: layer 3 of tile i - 1 for all blocks while the consumers run
    // tile i.  A operand: lane (o = l & 15, kk = l >> 4) holds
    // W3[o][32 j + 4 s + kk]; accumulators: register q of lane (n, kk) =
    // output 4 kk + q of data row 16 hf + n.
    // =====================================================================
    const int n16 = lane & 15, kk = lane >> 4;
    float a3[NB][8];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int u = 32 * j + 4 * s + kk;
        const float w3 = a.W3[(size_t)(n16 < OUT ? n16 : 0) * H + (u < H ? u : 0)];
        a3[j][s] = (u < H && n16 < OUT) ? w3 : 0.f;
      }
    f32x4 bias;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias[q] = 4 * kk + q < OUT ? a.b3[4 * kk + q] : 0.f;
    auto layer3 = [&](int t) {
      const float* hb = h2b + (t & 1) * kH2 + (n16 * 4 + kk);
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          // (BAL = 2: the small block writes its 8 real units only)
          if (SB && j == NB - 1 && 4 * s >= H - 32 * (NB - 1)) continue;
          // unit 4 s + kk of the block = register (s >> 1) * 4 + kk of the
          // consumer's lane half s & 1
          const int off = ((j * 4 + (s >> 1)) * 64 + (s & 1) * 32) * 4;
          const float b0 = hb[off], b1 = hb[off + 64];
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[j][s], b0, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[j][s], b1, o1, 0, 0, 0);
        }
      }
      const int tile = blockIdx.x + t * gridDim.x;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const RowOf ro = row_of(tile, 16 * hf + n16);
        const int rr = ro.mrow;
        f32x4 y = hf == 0 ? o0 : o1;
        if (!ro.tangent) y += bias;  // tangents: no bias
        if (ro.live) {
          if ((OUT & 3) == 0) {
            if (4 * kk < OUT)
              *reinterpret_cast<f32x4*>(a.Y + (size_t)rr * OUT + 4 * kk) = y;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (4 * kk + q < OUT) a.Y[(size_t)rr * OUT + 4 * kk + q] = y[q];
          }
        }
      }
    };
    if constexpr (BAL == 2) {
      float w0[4 * kMlpGive2], w1[4 * kMlpGive2];
      taker_weights(0, w0);
      taker_weights(1, w1);
      tile_barrier();
      for (int i = 0; i < iters; ++i) {
        PDDP_MLP_MARK(0);
        if (i >= 2) layer3(i - 2);
        PDDP_MLP_MARK(1);
        if (i < my) taker_partials(i, w0, w1, 0);
        PDDP_MLP_MARK(2);
        PDDP_MLP_MARK(3);
        tile_barrier();
        PDDP_MLP_MARK(4);
      }
      return;
    }
    if constexpr (BAL == 1) {
      // the givers' last chunks: lane (i = li, h = lh) holds
      // W2[32 jb + i][8 q + 4 h + e] like the owner itself would
      constexpr int kGiver[kMlpGivers] = {0, 1, 2, 4, 5, 6};
      float a2h[kMlpGivers][8];
#pragma unroll
      for (int gi = 0; gi < kMlpGivers; ++gi) {
        const int jb = kGiver[gi];
        const int u = 32 * jb + li;
        const bool uok = u < H;
        const float* w2row = a.W2 + (size_t)(uok ? u : 0) * H + 4 * lh;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = w2row[8 * (NQ - 2 + c) + e];
            a2h[gi][4 * c + e] = uok ? v : 0.f;
          }
      }
      auto partials = [&](int i) {  // of tile i, for the owners' next iteration
        const f32x4* bsrc =
            reinterpret_cast<const f32x4*>(h1t + (i & 1) * kH1) + (li * 2 + lh);
        const f32x4 b23 = bsrc[(NQ - 2) * 64], b24 = bsrc[(NQ - 1) * 64];
        f32x4* pw = reinterpret_cast<f32x4*>(part + (i & 1) * kPart) + lane;
#pragma unroll
        for (int gi = 0; gi < kMlpGivers; ++gi) {
          f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          if (gi < 3) {  // blocks 0 .. 2 give chunk NQ - 2 as well
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2h[gi][e], b23[e], acc, 0, 0, 0);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2h[gi][4 + e], b24[e], acc, 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            pw[(gi * 4 + g) * 64] = f32x4{acc[4 * g], acc[4 * g + 1],
                                          acc[4 * g + 2], acc[4 * g + 3]};
        }
      };
      tile_barrier();
      for (int i = 0; i < iters; ++i) {
        PDDP_MLP_MARK(0);
        // h2 of tile i - 2 was completed by the owners in iteration i - 1
        if (i >= 2) layer3(i - 2);
        PDDP_MLP_MARK(1);
        if (i < my) partials(i);
        PDDP_MLP_MARK(2);
        PDDP_MLP_MARK(3);
        tile_barrier();
        PDDP_MLP_MARK(4);
      }
      return;
    }
    tile_barrier();
    for (int i = 0; i < my; ++i) {
      if (i > 0) layer3(i - 1);
      tile_barrier();
    }
    layer3(my - 1);
    return;
  }

  // =======================================================================
  // consumer wavefront j: its 32 units of both hidden layers - rows of W2
  // (and of W1 | b1) in registers for the whole kernel.  Unit order of the
  // layer-2 contraction: MFMA step 4 q + e, k-slot h  <->  unit 8 q + 4 h + e,
  // so that the accumulator registers of layer 1 (unit_of) are written to LDS
  // as they are and come back as B operands four steps per ds_read_b128.
  // =======================================================================
  const int j = wave;
  constexpr int KS1 = kMlpW1Stride / 2;  // MFMA steps of layer 1
  float a2[PREC == 3 ? 1 : KS];
  // PREC = 3: lane (i = li, h = lh) holds W2[32 j + i][16 s + 8 h + e], e < 8,
  // of k-step s as three bf16 parts: the A operand of the 32x32x16 instruction
  bf16x8 a2h[PREC == 3 ? KS16 : 1], a2m[PREC == 3 ? KS16 : 1],
      a2l[PREC == 3 ? KS16 : 1];
  float a1[KS1];
  float b2r[16];
  {
    const int u = 32 * j + li;  // A operand: row i = li is unit u
    const bool uok = u < H;
    const float* w2row = a.W2 + (size_t)(uok ? u : 0) * H + 4 * lh;
    if constexpr (PREC == 3) {
#pragma unroll
      for (int s2 = 0; s2 < KS16; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * s2 + 8 * lh + e;
          const float v = a.W2[(size_t)(uok ? u : 0) * H + (k < H ? k : 0)];
          __bf16 hi, mid, lo;
          split3((uok && k < H) ? v : 0.f, hi, mid, lo);
          a2h[s2][e] = hi; a2m[s2][e] = mid; a2l[s2][e] = lo;
        }
    } else if (!(SB && j == NB - 1)) {  // (the small block: its own operand)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = w2row[8 * q + e];  // (clamped address + select: no
        a2[4 * q + e] = uok ? v : 0.f;     // branch per element)
      }
    }
    // layer 1: W1 | 0 | b1; MFMA step s, k-slot lh takes input slot
    // c = W1S - 1 - (2 s + lh): the bias first, then the inputs from the last
    // to the first (the order of round 1's FMA chain)
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const int c = kMlpW1Stride - 1 - (2 * s + lh);
      const float w = a.W1[(size_t)(uok ? u : 0) * IN + (c < IN ? c : 0)];
      const float bb = a.b1[uok ? u : 0];
      a1[s] = !uok ? 0.f : (c < IN ? w : (c == kMlpW1Stride - 1 ? bb : 0.f));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = unit_of(j, r, lh);
      const float bb = a.b2[n < H ? n : 0];
      b2r[r] = n < H ? bb : 0.f;
    }
  }
  // what the layer-2 accumulator starts from: the bias, or zero on a tangent
  // row (a property of the lane, not of the tile) - the C operand of the
  // tile's first MFMA, no copy
  f32x16 binit;
  {
    const bool tangent_lane = row_of(0, li).tangent;
#pragma unroll
    for (int r = 0; r < 16; ++r) binit[r] = tangent_lane ? 0.f : b2r[r];
  }
  // layer 1 of tile i for this block: inputs requested by l1_load (early),
  // four MFMAs, mask and ReLU on the accumulators, four ds_write_b128
  float xin[KS1];
  f32x4 m1[4];
  // The mask row [p][.] of this lane's data row, as a BYTE OFFSET that steps
  // from one tile of the workgroup to its next (groups advance by a constant,
  // so p = group % P advances by that constant modulo P: an add and an
  // unsigned min instead of a division per tile and mask - every vector
  // instruction here costs matrix time).  Two running offsets: the layer-1
  // masks run a tile ahead of the layer-2 masks.  Lanes without a live row
  // step through valid rows like the others (their products are dead).
  const unsigned rowsB = (unsigned)P * (unsigned)(H * 4);
  const unsigned stepB =
      (unsigned)(((JVP ? GPT : kMlpTile) * (int)gridDim.x) % P) * (unsigned)(H * 4);
  unsigned pm1 = (unsigned)(row_of(blockIdx.x, li).group % P) * (unsigned)(H * 4);
  const unsigned pm_tile0 = pm1;
  auto masks_load = [&](const float* M, unsigned pm, f32x4 (&m)[4]) {
    // registers 4 g .. 4 g + 3 are the units 32 j + 8 g + 4 lh + (0..3): one
    // 16-B load of the mask row each, 32-bit offsets from the (uniform) base;
    // the padded units of the last block (zero weights) read chunk 0 again
    const unsigned base = pm + (unsigned)(32 * j + 4 * lh) * 4u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool ok = 32 * j + 8 * g + 8 <= H;  // (wave-uniform)
      m[g] = *reinterpret_cast<const f32x4*>(
          reinterpret_cast<const char*>(M) + (base + (ok ? 32u * g : 0u)));
    }
  };
  auto masks_at = [&](const float* M, unsigned& pm, f32x4 (&m)[4]) {
    masks_load(M, pm, m);
    pm += stepB;
    const unsigned wrapped = pm - rowsB;  // (huge unless pm >= rowsB)
    pm = wrapped < pm ? wrapped : pm;
  };
  auto l1_load = [&](int i) {
    const RowOf ro = row_of(blockIdx.x + i * gridDim.x, li);
    const int row = ro.mrow;
    const bool live = ro.live;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const int c = kMlpW1Stride - 1 - (2 * s + lh);
      const float v = a.X[(size_t)(live ? row : 0) * IN + (c < IN ? c : 0)];
      // the last slot multiplies the bias (tangent rows carry none)
      xin[s] = c < IN ? (live ? v : 0.f)
                      : (c == kMlpW1Stride - 1 ? (ro.tangent ? 0.f : 1.f)
                                               : 0.f);
    }
    masks_at(a.MT1, pm1, m1);
  };
  auto layer1 = [&](int i) {
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS1; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], xin[s], acc, 0, 0, 0);
    f32x4* dst = reinterpret_cast<f32x4*>(h1t + (i & 1) * kH1) +
                 (4 * j * 64 + li * 2 + lh);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // (the products two at a time: v_pk_mul_f32 - the same IEEE products)
      const f32x2 t01 = f32x2{acc[4 * g], acc[4 * g + 1]} * f32x2{m1[g][0], m1[g][1]};
      const f32x2 t23 = f32x2{acc[4 * g + 2], acc[4 * g + 3]} * f32x2{m1[g][2], m1[g][3]};
      const float t[4] = {t01[0], t01[1], t23[0], t23[1]};
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (JVP) {
          // linearised at the group's primal row (for which this IS relu;
          // the mask is that of the group's particle, >= 0: where it is zero
          // the product is)
          v[e] = group_first_positive<JVP ? LIVE : 16>(acc[4 * g + e]) ? t[e] : 0.f;
        } else {
          v[e] = fmaxf(t[e], 0.f);
        }
      }
      if constexpr (PREC == 3) {
        // unit 32 j + 8 g + 4 lh + e = element 4 lh + e of half g & 1 of
        // k-step 2 j + (g >> 1): 8 bytes per part
        const int s2 = 2 * j + (g >> 1);
        if (s2 < KS16) {  // (wave-uniform)
          bf16x4 ph, pm, pl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            __bf16 hi, mid, lo;
            split3(v[e], hi, mid, lo);
            ph[e] = hi; pm[e] = mid; pl[e] = lo;
          }
          // (plane layout [k-step][half][row][8]: the 8-byte writes of 64
          // lanes are two-way on the banks, [row][half] was four-way; the
          // reads stay one contiguous kilobyte per wavefront)
          char* base = reinterpret_cast<char*>(h1t + (i & 1) * kH1) +
                       s2 * 1024 + (g & 1) * 512 + li * 16 + lh * 8;
          *reinterpret_cast<bf16x4*>(base) = ph;
          *reinterpret_cast<bf16x4*>(base + KS16 * 1024) = pm;
          *reinterpret_cast<bf16x4*>(base + 2 * KS16 * 1024) = pl;
        }
      } else {
        if (4 * j + g < NQ) dst[g * 64] = v;  // (wave-uniform)
      }
    }
  };
  l1_load(0);
  layer1(0);
  tile_barrier();  // h1t[0] ready

  // mask, ReLU of an accumulator tile and its way to LDS as held (the
  // finisher's reads know the order)
  auto epilogue = [&](int i, const f32x16& acc, const f32x4 (&m2)[4]) {
    f32x4* hw = reinterpret_cast<f32x4*>(h2b + (i & 1) * kH2) + (j * 4 * 64 + lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x2 t01 = f32x2{acc[4 * g], acc[4 * g + 1]} * f32x2{m2[g][0], m2[g][1]};
      const f32x2 t23 = f32x2{acc[4 * g + 2], acc[4 * g + 3]} * f32x2{m2[g][2], m2[g][3]};
      const float t[4] = {t01[0], t01[1], t23[0], t23[1]};
      f32x4 h2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (JVP) {
          h2[e] = group_first_positive<JVP ? LIVE : 16>(acc[4 * g + e]) ? t[e] : 0.f;
        } else {
          h2[e] = fmaxf(t[e], 0.f);
        }
      }
      hw[g * 64] = h2;
    }
  };
  // layer 2 on the matrix cores; the accumulator starts at the bias
  const int qown = q_own(wave);
  auto layer2 = [&](int i) {
    f32x16 acc;
    if constexpr (PREC == 3) {
      // B operand: lane (row li, half lh) holds h1[row][16 s + 8 lh + e]
      const char* bp = reinterpret_cast<const char*>(h1t + (i & 1) * kH1) +
                       lh * 512 + li * 16;
      acc = binit;
#pragma unroll
      for (int s2 = 0; s2 < KS16; ++s2) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + s2 * 1024);
        const bf16x8 bm =
            *reinterpret_cast<const bf16x8*>(bp + (KS16 + s2) * 1024);
        const bf16x8 bl =
            *reinterpret_cast<const bf16x8*>(bp + (2 * KS16 + s2) * 1024);
        // the small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2l[s2], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2h[s2], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2m[s2], bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2m[s2], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2h[s2], bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2h[s2], bh, acc, 0, 0, 0);
      }
    } else {
      const f32x4* bsrc =
          reinterpret_cast<const f32x4*>(h1t + (i & 1) * kH1) + (li * 2 + lh);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        // (wave-uniform: the chunks given)
        if (BAL && q >= NQ - (BAL == 2 ? kMlpGive2 : 2) && q >= qown) continue;
        const f32x4 b4 = bsrc[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(
            a2[4 * q + 0], b4[0], q == 0 ? binit : acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 1], b4[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 2], b4[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[4 * q + 3], b4[3], acc, 0, 0, 0);
      }
    }
    return acc;
  };

  if constexpr (SB) {
    if (j == NB - 1) {
      // ===================================================================
      // the small block (BAL = 2): units 32 jb .. 32 jb + 7 on 16 x 16 x 4
      // tiles.  A: lane (o = l & 15, kk = l >> 4) holds W2[32 jb + o][k] for
      // the four k of a step; B: lane (n = l & 15, kk) reads the SAME
      // ds_read_b128 chunks as the other blocks - chunk q = 2 t + (kk >> 1),
      // lane half kk & 1, row 16 hf + n: step (t, e) contracts the units
      // 8 (2 t + dq) + 4 lh + e, dq, lh in {0, 1}; D: register r of lane
      // (n, kk) = unit 32 jb + 4 kk + r of row 16 hf + n, which is the
      // finisher's word order - one ds_write_b128 per row half.
      // ===================================================================
      constexpr int jb = NB - 1;
      constexpr int NT = (NQ + 1) / 2;
      const int n16 = lane & 15, kk = lane >> 4, dq = kk >> 1, l2 = kk & 1;
      float a6[NT][4];
      {
        const int u = 32 * jb + n16;
        const bool uok = u < H;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int q = 2 * t + dq;
          const bool qok = q < NQ;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = a.W2[(size_t)(uok ? u : 0) * H + 8 * (qok ? q : 0) +
                                 4 * l2 + e];
            a6[t][e] = (uok && qok) ? v : 0.f;
          }
        }
      }
      f32x4 binit6[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bool tangent_row = row_of(0, 16 * hf + n16).tangent;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 32 * jb + 4 * kk + r;
          const float bb = a.b2[u < H ? u : 0];
          binit6[hf][r] = (u < H && !tangent_row) ? bb : 0.f;
        }
      }
      float w4[4 * kMlpGive2], w5[4 * kMlpGive2];
      taker_weights(4, w4);
      taker_weights(5, w5);
      unsigned pq[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        pq[hf] = (unsigned)(row_of(blockIdx.x, 16 * hf + n16).group % P) *
                 (unsigned)(H * 4);
      auto masks6 = [&](f32x4 (&m)[2]) {
        // units 32 jb + 4 kk + (0..3) for kk < 2 (the other slices read the
        // same words; their values are not written)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          m[hf] = *reinterpret_cast<const f32x4*>(
              reinterpret_cast<const char*>(a.MT2) +
              (pq[hf] + (unsigned)(32 * jb + 4 * l2) * 4u));
          pq[hf] += stepB;
          const unsigned wrapped = pq[hf] - rowsB;
          pq[hf] = wrapped < pq[hf] ? wrapped : pq[hf];
        }
      };
      auto layer2_6 = [&](int i, f32x4 (&acc)[2]) {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(h1t + (i & 1) * kH1);
        acc[0] = binit6[0];
        acc[1] = binit6[1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          // (odd NQ: the last step's second chunk does not exist - A is zero
          // there, B reads the first again)
          const int q = (2 * t + 1 < NQ) ? 2 * t + dq : 2 * t;
          const f32x4 b0 = bsrc[q * 64 + n16 * 2 + l2];
          const f32x4 b1 = bsrc[q * 64 + (16 + n16) * 2 + l2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a6[t][e], b0[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a6[t][e], b1[e], acc[1], 0, 0, 0);
          }
        }
      };
      auto epilogue6 = [&](int i, const f32x4 (&acc)[2], const f32x4 (&m)[2]) {
        f32x4 h[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x2 t01 = f32x2{acc[hf][0], acc[hf][1]} * f32x2{m[hf][0], m[hf][1]};
          const f32x2 t23 = f32x2{acc[hf][2], acc[hf][3]} * f32x2{m[hf][2], m[hf][3]};
          h[hf] = f32x4{t01[0], t01[1], t23[0], t23[1]};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (JVP) {
            bool c0, c1;
            group_first_positive_halves<JVP ? LIVE : 16>(acc[0][r], acc[1][r], c0, c1);
            h[0][r] = c0 ? h[0][r] : 0.f;
            h[1][r] = c1 ? h[1][r] : 0.f;
          } else {
            h[0][r] = fmaxf(h[0][r], 0.f);
            h[1][r] = fmaxf(h[1][r], 0.f);
          }
        }
        if (kk < 2) {
          f32x4* hw = reinterpret_cast<f32x4*>(h2b + (i & 1) * kH2) +
                      ((jb * 4) * 64 + kk * 32 + n16);
          hw[0] = h[0];
          hw[16] = h[1];
        }
      };
      f32x4 acc6[2] = {binit6[0], binit6[1]}, m6[2] = {binit6[0], binit6[0]};
      for (int i = 0; i < iters; ++i) {
        PDDP_MLP_MARK(0);
        if (i + 1 < my) l1_load(i + 1);
        if (i >= 1 && i <= my) epilogue6(i - 1, acc6, m6);
        PDDP_MLP_MARK(1);
        if (i < my) {
          masks6(m6);
          layer2_6(i, acc6);
          taker_partials(i, w4, w5, 2);
        }
        PDDP_MLP_MARK(2);
        if (i + 1 < my) layer1(i + 1);
        PDDP_MLP_MARK(3);
        tile_barrier();
        PDDP_MLP_MARK(4);
      }
      return;
    }
  }

  if constexpr (BAL) {
    // iteration i: the epilogue of tile i - 1 (with the finisher's partial
    // sums of its last chunks), layer 2 of tile i, layer 1 of tile i + 1
    f32x16 acc = binit;
    f32x4 m2[4];
    unsigned pm2 = pm_tile0;
    for (int i = 0; i < iters; ++i) {
      PDDP_MLP_MARK(0);
      if (i + 1 < my) l1_load(i + 1);
      if (i >= 1 && i <= my) {
        if (is_giver(wave)) {
          const f32x4* pr = reinterpret_cast<const f32x4*>(
                                part + ((i - 1) & 1) * kPart) +
                            (giver_index(wave) * 4 * 64 + lane);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = pr[g * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] += v[e];
          }
        }
        epilogue(i - 1, acc, m2);
      }
      PDDP_MLP_MARK(1);
      if (i < my) {
        // mask of layer 2, requested before the MFMAs so that its latency
        // is theirs (used by the epilogue, one iteration later)
        masks_at(a.MT2, pm2, m2);
        acc = layer2(i);
      }
      PDDP_MLP_MARK(2);
      if (i + 1 < my) layer1(i + 1);
      PDDP_MLP_MARK(3);
      tile_barrier();
      PDDP_MLP_MARK(4);
    }
    return;
  }

  // the mask of layer 2 of tile i is requested right after the epilogue of
  // tile i - 1 has used the registers: a whole iteration ahead of its use (the
  // bf16-split layer 2 is too short to cover the latency of a request made
  // just before it - 10 % of that kernel)
  // (likewise the inputs and the layer-1 mask of tile i + 2, as soon as layer
  // 1 of tile i + 1 has consumed the registers)
  f32x4 m2[4];
  unsigned pm2 = pm_tile0;
  masks_at(a.MT2, pm2, m2);
  if (my > 1) l1_load(1);
  for (int i = 0; i < my; ++i) {
    const bool nxt = i + 1 < my;
    const f32x16 acc = layer2(i);
    epilogue(i, acc, m2);
    if (nxt) masks_at(a.MT2, pm2, m2);
    if (nxt) layer1(i + 1);
    if (i + 2 < my) l1_load(i + 2);
    tile_barrier();
  }
}

template <int H, int W1S, int JVP, int LIVE, int BAL, int PREC = 0>
static int launch_bnn_mlp_b(const BnnMlpArgs& a, hipStream_t st) {
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
  // memory rows per tile: 32, or (32 / LIVE) whole groups of JVP rows
  constexpr int trows = JVP == 0 ? kMlpTile : (kMlpTile / LIVE) * JVP;
  const int ntiles = (a.R + trows - 1) / trows;
  const int grid = ntiles < cus ? ntiles : cus;  // persistent: one per CU
  constexpr size_t lds =
      sizeof(float) * bnn_mlp_lds_floats<H, W1S, BAL, PREC>();
  static_assert(lds <= 160 * 1024, "a workgroup's LDS");
  if (!attr_set[dev]) {  // more than 64 KB of dynamic LDS needs the opt-in
    const hipError_t e = hipFuncSetAttribute(
        (const void*)bnn_mlp_kernel<H, W1S, JVP, LIVE, BAL, PREC>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  PDDP_LAUNCH((bnn_mlp_kernel<H, W1S, JVP, LIVE, BAL, PREC>), dim3(grid),
                     dim3(kMlpThreads), lds, st, a);
  return launch_status();
}

// 0: exact f32 (default); 3: the bf16-split twin of layer 2 (H = 200 only).
// Process-wide; PDDP_MLP_BF16X3=1 in the environment makes 3 the default.
static int& mlp_precision() {
  static int prec = [] {
    const char* e = getenv("PDDP_MLP_BF16X3");
    return (e != nullptr && e[0] == '1') ? 3 : 0;
  }();
  return prec;
}

// -1: the default deal per mode (launch_bnn_mlp_w); 0 / 1 / 2: that deal for
// inference and forward mode alike.  Process-wide; PDDP_MLP_BALANCED in the
// environment sets the initial value.
static int& mlp_deal() {
  static int deal = [] {
    const char* e = getenv("PDDP_MLP_BALANCED");
    return e == nullptr ? -1 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2));
  }();
  return deal;
}

template <int H, int W1S, int JVP = 0, int LIVE = JVP>
static int launch_bnn_mlp_w(const BnnMlpArgs& a, hipStream_t st) {
  if constexpr (H == 200) {
    if (mlp_precision() == 3)
      return launch_bnn_mlp_b<H, W1S, JVP, LIVE, 0, 3>(a, st);
    // the balanced role layouts.  Default: 2 (the small last block on 16 x 16
    // tiles, three chunks handed over) for inference, 1 (round 2's deal) in
    // forward mode - another deal is another summation order, to rounding the
    // same numbers, but forward mode linearises the ReLUs at the primal row and
    // a pre-activation within rounding of zero then takes the other sign: one
    // Jacobian block of the real-size fixture moves by 1e-3 of its scale under
    // deal 2 (test_bnn_hip_kernels_vs_reference_real_size holds the exact
    // kernel to 1e-5 there), so forward mode keeps the order the fixtures were
    // pinned with.  PDDP_MLP_BALANCED=0 / 1 / 2 forces one deal for both.
    const int forced = mlp_deal();
    const int balanced = forced >= 0 ? forced : (JVP == 0 ? 2 : 1);
    if (balanced == 2) return launch_bnn_mlp_b<H, W1S, JVP, LIVE, 2>(a, st);
    if (balanced == 1) return launch_bnn_mlp_b<H, W1S, JVP, LIVE, 1>(a, st);
  }
  return launch_bnn_mlp_b<H, W1S, JVP, LIVE, 0>(a, st);
}

template <int H, int JVP = 0, int LIVE = JVP>
static int launch_bnn_mlp(const BnnMlpArgs& a, hipStream_t st) {
  return a.in_dim < 8 ? launch_bnn_mlp_w<H, 8, JVP, LIVE>(a, st)
                      : launch_bnn_mlp_w<H, 16, JVP, LIVE>(a, st);
}

}  // namespace pddp

extern "C" {

#ifdef PDDP_MLP_MARKS
int pddp_debug_mlp_marks(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pddp::g_mlp_marks),
                                  sizeof(long long) * 64);
}
#endif

int pddp_bnn_mlp_precision(int mode) {
  if (mode != 0 && mode != 3 && mode != -1) return PDDP_E_BADARG;
  const int prev = pddp::mlp_precision();
  if (mode >= 0) pddp::mlp_precision() = mode;
  return prev;
}

int pddp_bnn_mlp_deal(int deal) {
  if (deal < -2 || deal > 2) return PDDP_E_BADARG;
  const int prev = pddp::mlp_deal();
  if (deal >= -1) pddp::mlp_deal() = deal;
  return prev;
}

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
  return pddp_bnn_mlp_rows_f32(R, P, in_dim, H, out_dim, X, W1, b1, MT1, W2, b2,
                               MT2, W3, b3, Y, nullptr, stream);
}

int pddp_bnn_mlp_rows_f32(int R, int P, int in_dim, int H, int out_dim,
                          const float* X, const float* W1, const float* b1,
                          const float* MT1, const float* W2, const float* b2,
                          const float* MT2, const float* W3, const float* b3,
                          float* Y, const int32_t* live_rows, void* stream) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !MT1 || !W2 || !b2 || !MT2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if (in_dim >= pddp::kMlpW1Max || out_dim > pddp::kMlpMaxOut)
    return PDDP_E_UNSUPPORTED;
  const pddp::BnnMlpArgs a{R, P, in_dim, H, out_dim, X, W1, b1, MT1, W2,
                           b2, MT2, W3, b3, Y, live_rows};
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 64: return pddp::launch_bnn_mlp<64>(a, st);
    case 128: return pddp::launch_bnn_mlp<128>(a, st);
    case 200: return pddp::launch_bnn_mlp<200>(a, st);
  }
  return PDDP_E_UNSUPPORTED;
}

static int bnn_mlp_jvp_impl(int R, int P, int group, int live, int in_dim,
                            int H, int out_dim, const float* X, const float* W1,
                            const float* b1, const float* MT1, const float* W2,
                            const float* b2, const float* MT2, const float* W3,
                            const float* b3, float* Y, void* stream,
                            const int32_t* live_rows = nullptr) {
  if (R <= 0 || P <= 0 || in_dim <= 0 || H <= 0 || out_dim <= 0 || !X || !W1 ||
      !b1 || !MT1 || !W2 || !b2 || !MT2 || !W3 || !b3 || !Y)
    return PDDP_E_BADARG;
  if ((group != 8 && group != 16 && group != 32) || R % group != 0 ||
      live < 1 || live > group)
    return PDDP_E_BADARG;
  if (in_dim >= pddp::kMlpW1Max || out_dim > pddp::kMlpMaxOut)
    return PDDP_E_UNSUPPORTED;
  const pddp::BnnMlpArgs a{R, P, in_dim, H, out_dim, X, W1, b1, MT1, W2,
                           b2, MT2, W3, b3, Y, live_rows};
  hipStream_t st = (hipStream_t)stream;
#define PDDP_MLP_H(G, L)                                             \
  switch (H) {                                                       \
    case 64: return pddp::launch_bnn_mlp<64, G, L>(a, st);           \
    case 128: return pddp::launch_bnn_mlp<128, G, L>(a, st);         \
    case 200: return pddp::launch_bnn_mlp<200, G, L>(a, st);         \
  }                                                                  \
  return PDDP_E_UNSUPPORTED
  if (group == 8) {
    // packed forms: 8 groups of 4 live rows or 5 groups of 6 per tile
    // (pendulum: 1 + 2 + 1 rows, cartpole: 1 + 4 + 1); anything else runs 4
    // groups of 8
    if (live <= 4) { PDDP_MLP_H(8, 4); }
    if (live <= 6) { PDDP_MLP_H(8, 6); }
    PDDP_MLP_H(8, 8);
  } else if (group == 16) {
    PDDP_MLP_H(16, 16);
  } else {
    PDDP_MLP_H(32, 32);
  }
#undef PDDP_MLP_H
}

int pddp_bnn_mlp_jvp_f32(int R, int P, int group, int in_dim, int H,
                         int out_dim, const float* X, const float* W1, const float* b1,
                         const float* MT1, const float* W2, const float* b2,
                         const float* MT2, const float* W3, const float* b3,
                         float* Y, void* stream) {
  return bnn_mlp_jvp_impl(R, P, group, group, in_dim, H, out_dim, X, W1, b1, MT1,
                          W2, b2, MT2, W3, b3, Y, stream);
}

int pddp_bnn_mlp_jvp_live_f32(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const float* X,
                              const float* W1, const float* b1, const float* MT1,
                              const float* W2, const float* b2, const float* MT2,
                              const float* W3, const float* b3, float* Y,
                              void* stream) {
  return bnn_mlp_jvp_impl(R, P, group, live, in_dim, H, out_dim, X, W1, b1, MT1,
                          W2, b2, MT2, W3, b3, Y, stream);
}

int pddp_bnn_mlp_jvp_rows_f32(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const float* X,
                              const float* W1, const float* b1, const float* MT1,
                              const float* W2, const float* b2, const float* MT2,
                              const float* W3, const float* b3, float* Y,
                              const int32_t* live_rows, void* stream) {
  return bnn_mlp_jvp_impl(R, P, group, live, in_dim, H, out_dim, X, W1, b1, MT1,
                          W2, b2, MT2, W3, b3, Y, stream, live_rows);
}

}  // extern "C"
