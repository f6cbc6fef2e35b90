// riccati_mfma16.hpp - backward Riccati sweep for n + 1 <= 15, m = 1 on the
// matrix cores: fp32 (v_mfma_f32_16x16x4_f32: exact f32, bitwise an fmaf
// chain) and fp64 (v_mfma_f64_16x16x4_f64), both gain branches (eig clamp ilqr.py:629-672, V_zz-regularised
// Cholesky :587-625), bounded or not.  The DEFAULT
// (Cholesky) encoding of cartpole is n = 14: BASELINE.json configs[2]'s sweep,
// 744 MB per launch at B = 4096, N = 100.
//
// One wavefront per trajectory; the whole step is TWO 16x16x16 products on
// augmented matrices and stays in registers:
//   F~ = [F_z | F_u | 0]            (rows k < n, column n = F_u)
//   X  = V F~                       A operand = V, B operand = F~
//   X[:, 15] := V_z                 (column 15 is free: n + 1 <= 15)
//   Q~ = L~ + F~^T X                A operand = F~^T, B operand = X
// with L~ = [[L_zz, L_uz^T, L_z], [L_uz, L_uu, L_u]]:  Q~ holds Q_zz (raw), Q_uz
// (row n), Q_uu ([n][n]) and in column 15 Q_z, Q_u (ilqr.py:489-526) - every
// second- AND first-order term of the step from eight MFMA instructions.
// Why nothing moves between lanes:
//   * the C/D layout (column on the lane, rows 4g + r in register r of lane
//     group g) of V' is, V' being symmetric, exactly the A operand of the next
//     X = V F~ when k-slab r of lane group g is taken to be k = 4g + r;
//   * with the same k assignment the D registers of X are the B operand of
//     the second product, and the four F~ words a lane reads from the record
//     (F~[4g + r][j], j = lane & 15) serve as B operand of the first product
//     AND as A operand (F~^T[i][k], i = lane & 15) of the second.
// One LDS round trip per step remains: the transpose for 0.5 (Q + Q^T) and the
// broadcast of Q_uz for the rank-one value update
//   V' = sym(Q_zz) + c Q_uz^T Q_uz,  V_z' = Q_z + Q_uz^T w
// (c, w from the scalar BoxQP; riccati_n4_pipe.hpp).  Records stream HBM -> LDS
// by full-wave LDS-DMA, R slots ahead.
// fp64: the f64 instruction's C/D layout puts row g + 4 r (not 4 g + r) in
// register r of lane group g; with the k-slab of MFMA r taken as k = g + 4 r
// everything above holds unchanged (`Tile<T>::row`).
#pragma once

#include "riccati_n4.hpp"

namespace pddp {
namespace m16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kWaves = 4;  // independent trajectories (wavefronts) per workgroup
constexpr int kRing = 3;   // record slots in flight per wavefront
constexpr int kMaxDma = 4; // LDS-DMA instructions per record (64 x 16 B each)

// the 16x16x4 instruction of each type: accumulator vector, the matrix row
// (= k-slab) held by register r of lane group g, and where row n sits
template <typename T> struct Tile;
template <> struct Tile<float> {
  using Acc = f32x4;
  PDDP_DEV static int row(int g, int r) { return 4 * g + r; }
  PDDP_DEV static int group_of(int k) { return k >> 2; }
  PDDP_DEV static int reg_of(int k) { return k & 3; }
  PDDP_DEV static Acc mma(float x, float y, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
  }
  PDDP_DEV static float read_lane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
  }
};
template <> struct Tile<double> {
  using Acc = f64x4;
  PDDP_DEV static int row(int g, int r) { return g + 4 * r; }
  PDDP_DEV static int group_of(int k) { return k & 3; }
  PDDP_DEV static int reg_of(int k) { return k >> 2; }
  PDDP_DEV static Acc mma(double x, double y, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0);
  }
  PDDP_DEV static double read_lane(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
};

// NDMA: LDS-DMA instructions per record, chosen so that a slot (NDMA * 64
// chunks of 16 bytes) has strictly more chunks than the record: the slot's
// padding is zeroed once, the last DMA instruction's lanes past the record
// stay out, and operand entries outside the matrices read the first padding
// word - no per-entry zero selects; the slot stride is a compile-time
// constant, so that a step instantiated per ring slot gathers its operands
// with immediate offsets on addresses computed once (round 3: 141 -> 109
// vector instructions per step; at B = 4096 this kernel's four wavefronts per
// SIMD are bound by instruction issue, not by HBM).
template <typename T, bool BOUNDED, bool FAST, bool CHOL, int NDMA>
__global__ __launch_bounds__(kWave * kWaves) void riccati_mfma16_kernel(
    RiccatiArgs<T> a) {
  constexpr int slot_words = NDMA * kWave * (16 / (int)sizeof(T));
  using TL = Tile<T>;
  using Acc = typename TL::Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  // per wave: kRing record slots, the 16x16 transpose tile; shared: step sizes
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int per_wave = kRing * slot_words + 256;
  T* ring = smem + wave * per_wave;
  T* tile = ring + kRing * slot_words;
  T* ls_tail = smem + kWaves * per_wave;
  if constexpr (BOUNDED) {
    for (int q = threadIdx.x; q < n4::kLsSteps; q += kWave * kWaves)
      ls_tail[q] = (T)n4::kLs.v[q];
  }
  const T lstep0 = (T)n4::kLs.v[lane & 15];
  __syncthreads();

  const int b = blockIdx.x * kWaves + wave;
  if (b >= a.B) return;
  if (a.active != nullptr && a.active[b] == 0) return;
  const int n = a.n, N = a.N;
  const RecLayout lay(n, 1);
  const int S = lay.stride;
  const int g = lane >> 4, j = lane & 15;
  // this wavefront's slots: zero the padding behind the record, once
  for (int sl = 0; sl < kRing; ++sl)
    for (int wd = S + lane; wd < slot_words; wd += kWave)
      ring[sl * slot_words + wd] = T(0);
  const T reg = (T)a.reg[b];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) { umin = a.u_min[0]; umax = a.u_max[0]; }

  // ---- word offsets of this lane's operands inside a record (-1: zero)
  int oF[4], oL[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = TL::row(g, r);  // F~ row / L~ row
    oF[r] = (k < n) ? (j < n ? lay.oFz + k * n + j
                             : (j == n ? lay.oFu + k : S))
                    : S;  // (word S: the first word of the zeroed padding)
    int o = S;
    if (k < n) {
      if (j < n) o = lay.oLzz + k * n + j;
      else if (j == n) o = lay.oLuz + k;  // L_uz^T
      else if (j == 15) o = lay.oLz + k;
    } else if (k == n) {
      if (j < n) o = lay.oLuz + j;
      else if (j == n) o = lay.oLuu;
      else if (j == 15) o = lay.oLu;
    }
    oL[r] = o;
  }
  // Cholesky branch (ilqr.py:587-625): Q_uu, Q_uz once more with V + reg I,
  // i.e. + reg f^T F~ (row n of F~^T F~): this lane's share needs f[4g + r]
  int oFf[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    oFf[r] = (TL::row(g, r) < n) ? lay.oFu + TL::row(g, r) : S;

  // ---- record DMA: chunk q of 16 bytes -> lane q % 64 of instruction q / 64
  const char* rec_b = reinterpret_cast<const char*>(
      a.rec + (size_t)b * (size_t)(N + 1) * S);
  const int chunks = S * (int)sizeof(T) / 16;  // S is a multiple of 4 words
  // (padding lanes past the record re-load an early chunk into the slot's pad)
  // (the launcher picked NDMA with (NDMA - 1) * 64 < chunks < NDMA * 64: the
  // last instruction's lanes past the record stay out of it)
  uint32_t qoff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) qoff[i] = (uint32_t)(lane + i * kWave) * 16u;
  const bool in_tail = lane + (NDMA - 1) * kWave < chunks;
  auto dma = [&](int slot, int t) {
    const int tt = t < 0 ? 0 : t;
    const uint32_t base = (uint32_t)tt * (uint32_t)(S * sizeof(T));
    const uint32_t lbase =
        __builtin_amdgcn_readfirstlane(n4::lds_addr(ring + slot * slot_words));
#pragma unroll
    for (int i = 0; i < NDMA - 1; ++i)
      n4::lds_dma16(rec_b, base + qoff[i], lbase + i * kWave * 16);
    if (in_tail)
      n4::lds_dma16(rec_b, base + qoff[NDMA - 1],
                    lbase + (NDMA - 1) * kWave * 16);
  };

  // ---- terminal value function in the accumulator layout (ilqr.py:581-583)
  T V[4], Vz[4];
  {
    const T* term = a.rec + ((size_t)b * (size_t)(N + 1) + N) * S;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = TL::row(g, r);
      V[r] = (k < n && j < n) ? term[lay.oLzz + k * n + j] : T(0);
      Vz[r] = (k < n) ? term[lay.oLz + k] : T(0);
    }
  }
  for (int s = 0; s < kRing; ++s) dma(s, N - 1 - s);
  n4::wait_vmcnt<0>();  // (also the terminal loads above)

  T* gains_b = a.gains + (size_t)b * (size_t)N * lay.gstride;
  T kprev = T(0);
  int status = PDDP_BWD_OK;
  // gather addresses of slot 0, computed once: the step is instantiated per
  // ring slot and reaches the others through the read's immediate offset
  const T* pF[4];
  const T* pL[4];
  const T* pFf[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pF[r] = ring + oF[r];
    pL[r] = ring + oL[r];
    pFf[r] = ring + oFf[r];
  }
  const T* pU = ring + lay.oU;
  // (V_z travels as column 15 of X's initial value: zero in the other lanes)
#pragma unroll
  for (int r = 0; r < 4; ++r) Vz[r] = (j == 15) ? Vz[r] : T(0);
  auto step = [&](const int t, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int so = slot * slot_words;
    // record t has landed once at most (kRing - 1) younger {DMA x NDMA, store}
    // groups are outstanding
    n4::wait_vmcnt<(kRing - 1) * (NDMA + 1)>();
    T Fa[4], La[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Fa[r] = pF[r][so];
      La[r] = pL[r][so];
    }
    const T Un = BOUNDED ? pU[so] : T(0);
    T ffrow = T(0);  // (f^T F~)[j]: f^T F_z for j < n, f.f at j = n
    if constexpr (CHOL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ffrow += pFf[r][so] * Fa[r];
      ffrow += __shfl_xor(ffrow, 16);
      ffrow += __shfl_xor(ffrow, 32);
    }

    // ---- X = V F~ ; X[:, 15] = V_z (column 15 of F~ is zero)
    Acc X = {Vz[0], Vz[1], Vz[2], Vz[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) X = TL::mma(V[r], Fa[r], X);
    // ---- Q~ = L~ + F~^T X
    Acc Q = {La[0], La[1], La[2], La[3]};
    Q = TL::mma(Fa[0], X[0], Q);
    Q = TL::mma(Fa[1], X[1], Q);
    Q = TL::mma(Fa[2], X[2], Q);
    Q = TL::mma(Fa[3], X[3], Q);

    // row n (register rn of lane group gn) of Q~ is (Q_uz | Q_uu | Q_u at
    // column 15)
    const int gn = TL::group_of(n), rn = TL::reg_of(n);
    const T rowv = rn == 0 ? Q[0] : (rn == 1 ? Q[1] : (rn == 2 ? Q[2] : Q[3]));
    const T Quu = TL::read_lane(rowv, gn * 16 + n);
    const T Qu = TL::read_lane(rowv, gn * 16 + 15);
    // the regularised row (Q_uz_reg | Q_uu_reg) of the Cholesky branch
    const T rowg = CHOL ? rowv + reg * ffrow : rowv;
    const T Quug = CHOL ? TL::read_lane(rowg, gn * 16 + n) : Quu;
    // transpose tile: T[col][row] = Q~[row][col]; row 15 (free: n + 1 <= 15)
    // carries Q_uz_reg
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(tile + j * 16 + 4 * g) = Q;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[j * 16 + TL::row(g, r)] = Q[r];
    }
    if constexpr (CHOL) {
      if (g == gn) tile[j * 16 + 15] = rowg;
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
    bool Kz = false;
    int stt = st;
    if constexpr (BOUNDED) {
      n4::QpClosed<T, FAST> qc;
      qc.solve(kprev, qp_Q, Qu, umin - Un, umax - Un);
      kt = qc.x;
      Kz = !qc.free_;
      bool fail = qc.fail;
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
      const bool nanK = (g == gn) && (j < n) && (sE * rowg != sE * rowg);
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
      if (g == gn && j <= n) *dst = val;
    }
    // this slot is consumed: refill it, kRing steps further down the sweep
    dma(slot, t - kRing);

    // ---- V' = sym(Q_zz) + c Q_uz^T Q_uz,  V_z' = Q_z + Q_uz^T w
    const T Quz_j = tile[j * 16 + n];  // Q~[n][j]
    const T Qg_j = CHOL ? tile[j * 16 + 15] : T(0);  // Q_uz_reg[j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = TL::row(g, r);
      const T QT = tile[k * 16 + j];      // Q~[j][k]
      const T Quz_k = tile[k * 16 + n];   // Q~[n][k]
      const T sym = T(0.5) * (Q[r] + QT);
      if constexpr (CHOL) {
        // V' = sym + K^T Quu K + K^T Quz + Quz^T K,  K = -sE Q_uz_reg
        const T Qg_k = tile[k * 16 + 15];
        const T v = sym + c2 * (Qg_k * Qg_j) - sE * (Qg_k * Quz_j + Quz_k * Qg_j);
        // (no masks on V', V_z': their entries outside the n x n block only
        // ever meet zero rows / columns of F~ in the next step's products)
        V[r] = v;
        Vz[r] = (j == 15) ? Q[r] + Quz_k * kt - Qg_k * wz : T(0);
      } else {
        V[r] = n4::fma_(c * Quz_k, Quz_j, sym);
        Vz[r] = (j == 15) ? n4::fma_(Quz_k, w, Q[r]) : T(0);
      }
    }
  };
  // record tau lives in slot (N - 1 - tau) % kRing
  int t = N - 1;
  static_assert(kRing == 3 || kRing == 4, "");
  for (; t >= kRing - 1; t -= kRing) {
    step(t, std::integral_constant<int, 0>{});
    step(t - 1, std::integral_constant<int, 1>{});
    step(t - 2, std::integral_constant<int, 2>{});
    if constexpr (kRing == 4) step(t - 3, std::integral_constant<int, 3>{});
  }
  if (t >= 0) step(t, std::integral_constant<int, 0>{});
  if (t >= 1) step(t - 1, std::integral_constant<int, 1>{});
  if constexpr (kRing == 4) {
    if (t >= 2) step(t - 2, std::integral_constant<int, 2>{});
  }
  n4::wait_vmcnt<0>();
  if (j == 0 && g == 0) a.status[b] = status;
}

}  // namespace m16

// n + 1 <= 15, m = 1; PDDP_E_UNSUPPORTED otherwise (fp64: IEEE division only)
template <typename T>
static int launch_mfma16(const RiccatiArgs<T>& a, hipStream_t st,
                         bool fast_math) {
  if (a.n + 1 > 15) return PDDP_E_UNSUPPORTED;
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  const RecLayout lay(a.n, 1);
  const int chunks = lay.stride * (int)sizeof(T) / 16;
  // strictly more DMA lanes than chunks: the slot keeps a zeroed padding
  const int ndma = chunks / kWave + 1;
  if (ndma > m16::kMaxDma) return PDDP_E_UNSUPPORTED;
  const int slot_words = ndma * kWave * (16 / (int)sizeof(T));
  const size_t lds = sizeof(T) * ((size_t)m16::kWaves *
                                      (m16::kRing * slot_words + 256) +
                                  n4::kLsSteps);
  const dim3 grid((a.B + m16::kWaves - 1) / m16::kWaves),
      block(kWave * m16::kWaves);
  const bool bounded = a.u_min != nullptr;
#define PDDP_M16N(Bd, F, C, ND)                                                \
  PDDP_LAUNCH((m16::riccati_mfma16_kernel<T, Bd, F, C, ND>), grid, block, lds, \
              st, a)
#define PDDP_M16C(Bd, F, C)                                                    \
  do {                                                                         \
    switch (ndma) {                                                            \
      case 1: PDDP_M16N(Bd, F, C, 1); break;                                   \
      case 2: PDDP_M16N(Bd, F, C, 2); break;                                   \
      case 3: PDDP_M16N(Bd, F, C, 3); break;                                   \
      default: PDDP_M16N(Bd, F, C, 4); break;                                  \
    }                                                                          \
  } while (0)
#define PDDP_M16(Bd, F)                                                        \
  do {                                                                         \
    if (chol) PDDP_M16C(Bd, F, true);                                          \
    else PDDP_M16C(Bd, F, false);                                              \
  } while (0)
  if constexpr (sizeof(T) == 4) {
    if (bounded) { if (fast_math) PDDP_M16(true, true); else PDDP_M16(true, false); }
    else { if (fast_math) PDDP_M16(false, true); else PDDP_M16(false, false); }
  } else {
    if (bounded) PDDP_M16(true, false); else PDDP_M16(false, false);
  }
#undef PDDP_M16C
#undef PDDP_M16N
#undef PDDP_M16
  return launch_status();
}

}  // namespace pddp
