// riccati_mfma32.hpp - riccati_mfma16.hpp's scheme on 32x32 tiles
// (v_mfma_f32_32x32x2_f32) for 15 <= n <= 30, m = 1, fp32, both gain branches:
// the DEFAULT (Cholesky) encoding of the double cartpole is n = 27
// (BASELINE.json configs[3]: 0.96 GB of records per GPU and sweep).
//
// One sweeping wavefront per trajectory (two per workgroup, plus a producer
// wavefront that streams their records), two 32x32x32 products per step (16 MFMA
// instructions each) on the augmented F~ = [F_z | F_u | 0], L~ = [[L_zz, L_uz^T,
// L_z], [L_uz, L_uu, L_u]] with V_z riding in column 31 of X = V F~.  The
// accumulator layout (column on the lane, row (r & 3) + 8 (r >> 2) + 4 h in
// register r of lane half h) of the symmetric V' is the A operand of the next
// product when the k-slot h of instruction r is taken to be that row; X's
// registers are then the B operand of Q~ = L~ + F~^T X, and the sixteen F~
// words a lane gathers serve both products (see riccati_mfma16.hpp).
#pragma once

#include <type_traits>

#include "riccati_mfma16.hpp"

namespace pddp {
namespace m32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kWaves = 2;  // sweeping wavefronts (trajectories) per workgroup
constexpr int kRing = 2;   // record slots per sweeping wavefront
// + one producer wavefront per workgroup: it issues the record DMAs of both
// trajectories (an LDS-DMA instruction parks its wavefront for ~90 cycles -
// eight per step were 720 of the sweeping wave's ~7600) and meets them at one
// barrier per step
constexpr int kThreads = kWave * (kWaves + 1);
constexpr int kTileLd = 36;  // row stride of the transpose tile (16-B aligned,
                             // spreads the b128 writes over the banks)
constexpr int kTile = 32 * kTileLd;

PDDP_DEV int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <bool BOUNDED, bool FAST, int NDMA, bool CHOL>
__global__ __launch_bounds__(kThreads) void riccati_mfma32_kernel(
    RiccatiArgs<float> a) {
  using T = float;
  constexpr int kSlotWords = NDMA * kWave * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  constexpr int per_wave = kRing * kSlotWords + kTile;
  float* ls_tail = smem + kWaves * per_wave;
  if constexpr (BOUNDED) {
    for (int q = threadIdx.x; q < n4::kLsSteps; q += kThreads)
      ls_tail[q] = (T)n4::kLs.v[q];
  }
  const T lstep0 = (T)n4::kLs.v[lane & 15];
  const int n = a.n, N = a.N;
  const RecLayout lay(n, 1);
  const int S = lay.stride;
  const int chunks = S / 4;
  const int nd_live = (chunks + kWave - 1) / kWave;  // DMA instructions per record
  auto live_of = [&](int bb) {
    return bb < a.B && (a.active == nullptr || a.active[bb] != 0);
  };
  auto step_barrier = [] { asm volatile("s_barrier" ::: "memory"); };

  if (wave == kWaves) {
    // =====================================================================
    // producer: record t of both trajectories has landed before barrier
    // N - 1 - t; record t - 1 is requested right after it, into the slot the
    // sweeping waves have just finished with (they passed the barrier).
    // =====================================================================
    uint32_t qoff[NDMA];
#pragma unroll
    for (int r = 0; r < NDMA; ++r) qoff[r] = (uint32_t)(lane + kWave * r) * 16u;
    // the last instruction carries the record's tail: its lanes past the
    // record stay out, so that the slot's padding keeps its zeros
    const uint32_t qoff_last = (uint32_t)(lane + kWave * (nd_live - 1)) * 16u;
    const bool in_tail = lane + kWave * (nd_live - 1) < chunks;
    auto dma = [&](int w, int slot, int t) {
      const int bb = blockIdx.x * kWaves + w;
      if (!live_of(bb) || t < 0) return;  // (wave-uniform)
      const char* rec_b = reinterpret_cast<const char*>(
          a.rec + (size_t)bb * (size_t)(N + 1) * S);
      const uint32_t base = (uint32_t)t * (uint32_t)(S * sizeof(T));
      const uint32_t lbase = __builtin_amdgcn_readfirstlane(
          n4::lds_addr(smem + w * per_wave + slot * kSlotWords));
#pragma unroll
      for (int r = 0; r < NDMA - 1; ++r) {
        if (r < nd_live - 1)  // (instructions without a chunk are not issued)
          n4::lds_dma16(rec_b, base + qoff[r], lbase + r * kWave * 16);
      }
      if (in_tail)
        n4::lds_dma16(rec_b, base + qoff_last,
                      lbase + (uint32_t)(nd_live - 1) * kWave * 16);
    };
    __syncthreads();  // the sweeping waves have zeroed their slots' padding
#pragma unroll
    for (int w = 0; w < kWaves; ++w) dma(w, 0, N - 1);
    for (int t = N - 1; t >= 0; --t) {
      n4::wait_vmcnt<0>();  // record t (the only requests in flight)
      step_barrier();
      const int slot_next = (N - t) & 1;  // record t - 1 -> the other slot
#pragma unroll
      for (int w = 0; w < kWaves; ++w) dma(w, slot_next, t - 1);
    }
    return;
  }

  float* ring = smem + wave * per_wave;
  float* tile = ring + kRing * kSlotWords;
  // zero the padding of both slots once: operand entries outside the matrices
  // read it, and the record DMA never touches it
  for (int sl = 0; sl < kRing; ++sl)
    for (int wd = S + lane; wd < kSlotWords; wd += kWave)
      ring[sl * kSlotWords + wd] = T(0);
  __syncthreads();

  const int b = blockIdx.x * kWaves + wave;
  if (!live_of(b)) {  // keep step with the workgroup's barriers
    for (int t = N - 1; t >= 0; --t) step_barrier();
    return;
  }
  const int h = lane >> 5, j = lane & 31;
  const T reg = (T)a.reg[b];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) { umin = a.u_min[0]; umax = a.u_max[0]; }

  // ---- word offsets of this lane's operands inside a record.  Entries
  // outside the matrices read a word of the slot's padding, which is zeroed
  // once and which the record DMA never touches (its lanes past the record
  // are masked off) - no per-entry masks: 64 boolean lane masks would not fit
  // the SGPR file (the first version spilled them: 117 v_readlane + their
  // hazard nops per step; the second multiplied by 0 / 1 floats: 32 more
  // vector instructions and 32 more registers per step).
  int oF[16], oL[16];
  T mK[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = row_of(r, h);
    int f = (k < n) ? (j < n ? lay.oFz + k * n + j
                             : (j == n ? lay.oFu + k : -1))
                    : -1;
    int o = -1;
    if (k < n) {
      if (j < n) o = lay.oLzz + k * n + j;
      else if (j == n) o = lay.oLuz + k;  // L_uz^T
      else if (j == 31) o = lay.oLz + k;
    } else if (k == n) {
      if (j < n) o = lay.oLuz + j;
      else if (j == n) o = lay.oLuu;
      else if (j == 31) o = lay.oLu;
    }
    mK[r] = k < n ? T(1) : T(0);
    oF[r] = f < 0 ? S : f;  // word S: the first word of the zeroed padding
    oL[r] = o < 0 ? S : o;
  }
  // Cholesky branch (ilqr.py:587-625): + reg f^T F~ on row n (mfma16.hpp)
  int oFf[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    oFf[r] = lay.oFu + (row_of(r, h) < n ? row_of(r, h) : 0);

  // ---- terminal value function in the accumulator layout (ilqr.py:581-583)
  T V[16], Vz[16];
  {
    const T* term = a.rec + ((size_t)b * (size_t)(N + 1) + N) * S;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = row_of(r, h);
      V[r] = (k < n && j < n) ? term[lay.oLzz + k * n + j] : T(0);
      Vz[r] = (k < n) ? term[lay.oLz + k] : T(0);
    }
  }

  // row n of Q~ = (Q_uz | Q_uu | .. | Q_u at column 31): register rn of the
  // lanes of half hn
  const int hn = (n >> 2) & 1, rn = (n & 3) + 4 * (n >> 3);
  T* gains_b = a.gains + (size_t)b * (size_t)N * lay.gstride;
  T kprev = T(0);
  int status = PDDP_BWD_OK;
  static_assert(kRing == 2, "the step is instantiated once per ring slot");
  // one step of the sweep on the record in ring slot SLOT - a compile-time
  // constant, so that the 32 operand gathers are ds_read with an immediate
  // slot offset on per-lane addresses computed once (round 1: an address
  // computation per gather and step)
  auto step = [&](auto slot_c, int t) {
    constexpr int slot = decltype(slot_c)::value;
    step_barrier();  // record t has landed (the producer waited for it)
    const T* R = ring + slot * kSlotWords;
    T Fa[16];
    f32x16 Q;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Fa[r] = R[oF[r]];
      Q[r] = R[oL[r]];
    }
    const T Un = BOUNDED ? R[lay.oU] : T(0);
    T ffrow = T(0);  // (f^T F~)[j]: f^T F_z for j < n, f.f at j = n
    if constexpr (CHOL) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ffrow += (R[oFf[r]] * mK[r]) * Fa[r];
      ffrow += __shfl_xor(ffrow, 32);
    }

    // ---- X = V F~ ; X[:, 31] = V_z
    f32x16 X = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; ++r)
      X = __builtin_amdgcn_mfma_f32_32x32x2f32(V[r], Fa[r], X, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) X[r] = (j == 31) ? Vz[r] : X[r];
    // ---- Q~ = L~ + F~^T X
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Q = __builtin_amdgcn_mfma_f32_32x32x2f32(Fa[r], X[r], Q, 0, 0, 0);

    T rowv = Q[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) rowv = (r == rn) ? Q[r] : rowv;
    const T Quu = __int_as_float(
        __builtin_amdgcn_readlane(__float_as_int(rowv), hn * 32 + n));
    const T Qu = __int_as_float(
        __builtin_amdgcn_readlane(__float_as_int(rowv), hn * 32 + 31));
    const T rowg = CHOL ? rowv + reg * ffrow : rowv;  // (Q_uz_reg | Q_uu_reg)
    const T Quug = CHOL ? __int_as_float(__builtin_amdgcn_readlane(
                              __float_as_int(rowg), hn * 32 + n))
                        : Quu;
    // transpose tile: T[col][row] = Q~[row][col]; registers 4q .. 4q + 3 are
    // the consecutive rows 8q + 4h + (0..3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      m16::f32x4 v = {Q[4 * q], Q[4 * q + 1], Q[4 * q + 2], Q[4 * q + 3]};
      *reinterpret_cast<m16::f32x4*>(tile + j * kTileLd + 8 * q + 4 * h) = v;
    }
    if constexpr (CHOL) {  // row 31 (free: n + 1 <= 31) carries Q_uz_reg
      if (h == hn) tile[j * kTileLd + 31] = rowg;
    }

    // ---- gains (every lane the same scalars)                 (ilqr.py:629-657)
    int st = PDDP_BWD_OK;
    T qp_Q;
    if constexpr (CHOL) {
      qp_Q = Quug;  // Cholesky of Q_uu_reg                        (ilqr.py:595)
      if (!BOUNDED && (!(Quug > T(0)) || !is_finite(Quug))) st = PDDP_BWD_NOT_PD;
    } else {
      if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
      const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
      qp_Q = e + reg;                             // ilqr.py:634
    }
    T kt, sE;
    int stt = st;
    if constexpr (BOUNDED) {
      n4::QpClosed<T, FAST> qc;
      qc.solve(kprev, qp_Q, Qu, umin - Un, umax - Un);
      kt = qc.x;
      bool Kz = !qc.free_, fail = qc.fail;
      if (__builtin_amdgcn_ballot_w64(qc.slow) != 0) {
        const n4::SlowQpOut<T> o = n4::boxqp1_outlined<T, FAST>(
            kprev, qp_Q, Qu, umin - Un, umax - Un, lstep0, ls_tail, lane);
        kt = o.x;
        Kz = (o.result_free & 1) == 0;
        fail = o.result_free < 2;
      }
      // (a NaN Q_uu fails `eig` before the BoxQP is reached, ilqr.py:631)
      if (fail && st == PDDP_BWD_OK) stt = PDDP_BWD_BOXQP_FAILED;
      if constexpr (FAST) sE = Kz ? T(0) : qc.inv;
      else sE = Kz ? T(0) : n4::div_<false>(n4::div_<false>(T(1), qc.U), qc.U);
    } else {
      sE = n4::div_<FAST>(T(1), qp_Q);  // (E / e) E^T             (ilqr.py:636)
      kt = -(sE * Qu);
      // NaN in k or K raises (ilqr.py:639-640)
      const bool nanK = (h == hn) && (j < n) && (sE * rowg != sE * rowg);
      if (!CHOL && (kt != kt || __builtin_amdgcn_ballot_w64(nanK) != 0))
        stt = PDDP_BWD_NAN;
    }
    if (status == PDDP_BWD_OK && stt != PDDP_BWD_OK) status = stt;
    kprev = kt;
    const T c = sE * (sE * Quu - T(2));
    const T w = kt - sE * (Qu + Quu * kt);
    const T c2 = sE * sE * Quu;        // Cholesky branch: K = -sE Q_uz_reg
    const T wz = sE * (Qu + Quu * kt);

    // ---- k, K of step t: lanes of row n hold Q_uz[j] (j < n), lane j = n: k
    {
      const T val = (j < n) ? -(sE * rowg) : kt;
      T* dst = gains_b + (size_t)t * lay.gstride + (j < n ? 1 + j : 0);
      if (h == hn && j <= n) *dst = val;
    }

    // ---- V' = sym(Q_zz) + c Q_uz^T Q_uz,  V_z' = Q_z + Q_uz^T w
    const T Quz_j = tile[j * kTileLd + n];  // Q~[n][j]
    const T Qg_j = CHOL ? tile[j * kTileLd + 31] : T(0);  // Q_uz_reg[j]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = row_of(r, h);
      const T QT = tile[k * kTileLd + j];     // Q~[j][k]
      const T Quz_k = tile[k * kTileLd + n];  // Q~[n][k]
      const T sym = T(0.5) * (Q[r] + QT);
      if constexpr (CHOL) {
        // V' = sym + K^T Quu K + K^T Quz + Quz^T K,  K = -sE Q_uz_reg
        const T Qg_k = tile[k * kTileLd + 31];
        // (no masks on V', V_z': their entries outside the n x n block meet
        // zero rows / columns of F~ in both products of the next step, and
        // V_z' is read from lane j = 31 only - they are finite whenever the
        // step itself is)
        V[r] = sym + c2 * (Qg_k * Qg_j) - sE * (Qg_k * Quz_j + Quz_k * Qg_j);
        Vz[r] = Q[r] + Quz_k * kt - Qg_k * wz;
      } else {
        V[r] = __builtin_fmaf(c * Quz_k, Quz_j, sym);
        Vz[r] = __builtin_fmaf(Quz_k, w, Q[r]);  // (lanes j = 31)
      }
    }
  };
  for (int t = N - 1; t >= 0; t -= 2) {
    step(std::integral_constant<int, 0>{}, t);
    if (t >= 1) step(std::integral_constant<int, 1>{}, t - 1);
  }
  n4::wait_vmcnt<0>();
  if (lane == 0) a.status[b] = status;
}

}  // namespace m32

// 15 <= n <= 30, m = 1, fp32; PDDP_E_UNSUPPORTED otherwise
static int launch_mfma32(const RiccatiArgs<float>& a, hipStream_t st,
                         bool fast_math) {
  if (a.n < 15 || a.n > 30) return PDDP_E_UNSUPPORTED;
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  const RecLayout lay(a.n, 1);
  const int chunks = lay.stride / 4;
  // (strictly fewer chunks than DMA lanes: the slot keeps zeroed padding)
  const int ndma = chunks / kWave + 1 <= 4 ? 4 : 8;
  if (chunks >= ndma * kWave) return PDDP_E_UNSUPPORTED;
  const size_t lds = sizeof(float) * ((size_t)m32::kWaves *
                                          (m32::kRing * ndma * kWave * 4 +
                                           m32::kTile) +
                                      n4::kLsSteps);
  const dim3 grid((a.B + m32::kWaves - 1) / m32::kWaves),
      block(m32::kThreads);
  const bool bounded = a.u_min != nullptr;
#define PDDP_M32(Bd, F, ND)                                                    \
  do {                                                                         \
    auto kern = chol ? m32::riccati_mfma32_kernel<Bd, F, ND, true>             \
                     : m32::riccati_mfma32_kernel<Bd, F, ND, false>;           \
    const hipError_t e_ = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,         \
        (int)lds);                                                             \
    if (e_ != hipSuccess) return (int)e_;                                      \
    PDDP_LAUNCH(kern, grid, block, lds, st, a);                                \
  } while (0)
#define PDDP_M32_ND(Bd, F)                                                     \
  do {                                                                         \
    if (ndma == 4) PDDP_M32(Bd, F, 4); else PDDP_M32(Bd, F, 8);                \
  } while (0)
  if (bounded) { if (fast_math) PDDP_M32_ND(true, true); else PDDP_M32_ND(true, false); }
  else { if (fast_math) PDDP_M32_ND(false, true); else PDDP_M32_ND(false, false); }
#undef PDDP_M32_ND
#undef PDDP_M32
  return launch_status();
}

}  // namespace pddp
