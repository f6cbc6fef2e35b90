// riccati_mfma16_nominal.hpp - the backward sweep FROM THE NOMINAL for the
// sample problems whose records go through the 16 x 16 matrix-core sweep
// (riccati_mfma16.hpp: n + 1 <= 15, m = 1): pendulum (n = 2) and double
// cartpole (n = 6) under IGNORE_UNCERTAINTY, fp32 and fp64, all four gain
// branches.  What pddp_sweep_nominal_f32 does for cartpole on
// riccati_n4_elem.hpp (task 6): the derivative records
// (ilqr.py:464-473) are evaluated inside the sweep's wavefront, into LDS, and
// never written - the round is two launches (this, the fused search + accept)
// instead of three, and B (N + 1) S words of records are neither written nor
// read back.
//
// One wavefront per trajectory, as in riccati_mfma16.hpp, and the same step
// (two 16 x 16 x 16 products on the augmented matrices, one LDS round trip for
// the transpose).  The record ring holds a BLOCK of RB steps: every RB steps
// the wavefront turns into RB generators - lane l evaluates the record of step
// t_hi - l (models.hpp record_of: the model's closed-form Jacobian and the
// cost's gradient / Hessian, the code pddp_derivs_* runs) and stores it into
// slot (t mod RB); a record costs a lane 300 .. 1500 instructions, 5 .. 25 per
// step of the sweep.  Stage costs go to LDS and from there to L [B][N + 1];
// J_opt = their sum in t order (ilqr.py:289) where `fresh` is set.
#pragma once

#include "models.hpp"
#include "riccati_mfma16.hpp"
#include "riccati_n4_elem.hpp"  // GenArgs

namespace pddp {
namespace m16n {

using m16::Tile;
using m16::kWaves;

template <typename T, int MODEL, bool BOUNDED, bool FAST, bool CHOL, int RB>
__global__ __launch_bounds__(kWave * kWaves) void riccati_mfma16_nominal_kernel(
    RiccatiArgs<T> a, n4d::GenArgs<T> gen, ProblemT<T> P) {
  using D = ModelDims<MODEL>;
  using TL = Tile<T>;
  using Acc = typename TL::Acc;
  constexpr int n = D::n;
  static_assert(D::m == 1 && n + 1 <= 15, "the 16 x 16 sweep's shapes");
  static_assert(RB <= kWave, "one generator lane per step of a block");
  constexpr RecLayout lay(n, 1);
  constexpr int S = lay.stride;
  constexpr int SW = S + 4;  // slot: the record + four zero words (word S is
                             // what operand entries outside the matrices read)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int N = a.N;
  const int LW = (N + 1 + 3) & ~3;  // stage costs of the trajectory
  // per wave: RB slots + one for the terminal record, the transpose tile, L
  const int per_wave = (RB + 1) * SW + 256 + LW;
  T* ring = smem + wave * per_wave;
  T* tile = ring + (RB + 1) * SW;
  T* Lsh = tile + 256;
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
  const int g = lane >> 4, j = lane & 15;
  for (int sl = lane; sl < RB + 1; sl += kWave) {  // the slots' zero words, once
#pragma unroll
    for (int q = 0; q < 4; ++q) ring[sl * SW + S + q] = T(0);
  }
  const T reg = (T)a.reg[b];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) { umin = a.u_min[0]; umax = a.u_max[0]; }

  // ---- the generators: lane l < RB evaluates the record of step t_hi - l
  const T* Zb = gen.Z + (size_t)b * (size_t)(N + 1) * n;
  const T* Ub = gen.U + (size_t)b * (size_t)N;
  auto put_record = [&](int t, int slot, bool terminal) {
    T z[n], un[1], w[S];
#pragma unroll
    for (int q = 0; q < n; ++q) z[q] = Zb[(size_t)t * n + q];
    un[0] = terminal ? T(0) : Ub[t];
    const T l = record_of<T, MODEL>(P, z, un, terminal, BOUNDED, a.u_min,
                                    a.u_max, w);
    T* dst = ring + slot * SW;
    using V4 = T __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < S; q += 4)
      *reinterpret_cast<V4*>(dst + q) = V4{w[q], w[q + 1], w[q + 2], w[q + 3]};
    Lsh[t] = l;
  };
  // (same-wavefront LDS traffic is in order; the compiler needs telling)
  auto lds_fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto gen_block = [&](int t_hi) {
    const int t = t_hi - lane;
    if (lane < RB && t >= 0) put_record(t, t % RB, false);
    lds_fence();
  };
  if (lane == 0) put_record(N, RB, true);  // terminal: slot RB
  lds_fence();

  // ---- word offsets of this lane's operands inside a record (S: zero)
  int oF[4], oL[4], oFf[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = TL::row(g, r);
    oF[r] = (k < n) ? (j < n ? lay.oFz + k * n + j
                             : (j == n ? lay.oFu + k : S))
                    : S;
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
    oFf[r] = (k < n) ? lay.oFu + k : S;
  }

  // ---- terminal value function in the accumulator layout (ilqr.py:581-583)
  T V[4], Vz[4];
  {
    const T* term = ring + RB * SW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = TL::row(g, r);
      V[r] = (k < n && j < n) ? term[lay.oLzz + k * n + j] : T(0);
      // (V_z travels as column 15 of X's initial value: zero elsewhere)
      Vz[r] = (k < n && j == 15) ? term[lay.oLz + k] : T(0);
    }
  }

  T* gains_b = a.gains + (size_t)b * (size_t)N * lay.gstride;
  T kprev = T(0);
  int status = PDDP_BWD_OK;
  const int gn = TL::group_of(n), rn = TL::reg_of(n);
  // one step of the sweep on the record in ring slot `slot`
  // (riccati_mfma16.hpp's, the operands gathered at a run-time slot offset)
  auto step = [&](const int t, const int slot) {
    const T* R = ring + slot * SW;
    T Fa[4], La[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Fa[r] = R[oF[r]];
      La[r] = R[oL[r]];
    }
    const T Un = BOUNDED ? R[lay.oU] : T(0);
    T ffrow = T(0);  // (f^T F~)[j]: f^T F_z for j < n, f.f at j = n
    if constexpr (CHOL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ffrow += R[oFf[r]] * Fa[r];
      ffrow += __shfl_xor(ffrow, 16);
      ffrow += __shfl_xor(ffrow, 32);
    }
    // ---- X = V F~ ; X[:, 15] = V_z ;  Q~ = L~ + F~^T X
    Acc X = {Vz[0], Vz[1], Vz[2], Vz[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) X = TL::mma(V[r], Fa[r], X);
    Acc Q = {La[0], La[1], La[2], La[3]};
    Q = TL::mma(Fa[0], X[0], Q);
    Q = TL::mma(Fa[1], X[1], Q);
    Q = TL::mma(Fa[2], X[2], Q);
    Q = TL::mma(Fa[3], X[3], Q);

    const T rowv = rn == 0 ? Q[0] : (rn == 1 ? Q[1] : (rn == 2 ? Q[2] : Q[3]));
    const T Quu = TL::read_lane(rowv, gn * 16 + n);
    const T Qu = TL::read_lane(rowv, gn * 16 + 15);
    const T rowg = CHOL ? rowv + reg * ffrow : rowv;
    const T Quug = CHOL ? TL::read_lane(rowg, gn * 16 + n) : Quu;
    // transpose tile: T[col][row] = Q~[row][col]; row 15 carries Q_uz_reg
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[j * 16 + TL::row(g, r)] = Q[r];
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
    {  // k, K of step t
      const T val = (j < n) ? -(sE * rowg) : kt;
      T* dst = gains_b + (size_t)t * lay.gstride + (j < n ? 1 + j : 0);
      if (g == gn && j <= n) *dst = val;
    }
    // ---- V' = sym(Q_zz) + c Q_uz^T Q_uz,  V_z' = Q_z + Q_uz^T w
    lds_fence();
    const T Quz_j = tile[j * 16 + n];  // Q~[n][j]
    const T Qg_j = CHOL ? tile[j * 16 + 15] : T(0);  // Q_uz_reg[j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = TL::row(g, r);
      const T QT = tile[k * 16 + j];      // Q~[j][k]
      const T Quz_k = tile[k * 16 + n];   // Q~[n][k]
      const T sym = T(0.5) * (Q[r] + QT);
      if constexpr (CHOL) {
        const T Qg_k = tile[k * 16 + 15];
        V[r] = sym + c2 * (Qg_k * Qg_j) - sE * (Qg_k * Quz_j + Quz_k * Qg_j);
        Vz[r] = (j == 15) ? Q[r] + Quz_k * kt - Qg_k * wz : T(0);
      } else {
        V[r] = n4::fma_(c * Quz_k, Quz_j, sym);
        Vz[r] = (j == 15) ? n4::fma_(Quz_k, w, Q[r]) : T(0);
      }
    }
    lds_fence();  // (the tile is written again by the next step)
  };
  for (int t = N - 1; t >= 0; --t) {
    if ((N - 1 - t) % RB == 0) gen_block(t);
    step(t, t % RB);
  }
  if (j == 0 && g == 0) a.status[b] = status;
  // ---- stage costs out; J_opt of a new nominal: their sum in t order
  lds_fence();
  T* Lb = gen.L + (size_t)b * (size_t)(N + 1);
  for (int t = lane; t <= N; t += kWave) Lb[t] = Lsh[t];
  if (lane == 0 && (gen.fresh == nullptr || gen.fresh[b] != 0)) {
    T acc = T(0);
    for (int t = 0; t <= N; ++t) acc += Lsh[t];
    gen.J_opt[b] = acc;
    if (gen.fresh != nullptr) gen.fresh[b] = 0;
  }
}

}  // namespace m16n

// pendulum / double cartpole under IGNORE_UNCERTAINTY; PDDP_E_UNSUPPORTED
// otherwise.  (Cartpole in fp64 was tried on this kernel: 482 us per round
// against 200 on records - four lanes in sixteen of its 16 x 16 tiles are
// matrix - and is left out.)
template <typename T, int MODEL>
static int launch_m16_nominal_model(const pddp_problem& p,
                                    const RiccatiArgs<T>& a,
                                    const n4d::GenArgs<T>& gen, hipStream_t st) {
  using D = ModelDims<MODEL>;
  const ProblemT<T> P = convert_problem<T>(p);
  constexpr RecLayout lay(D::n, 1);
  // steps per block: a lane per step, and the block's records next to the
  // tile in at most ~36 KB of LDS per wavefront
  constexpr int RB = (lay.stride + 4) * (int)sizeof(T) * 64 <= 36 * 1024
                         ? 64
                         : ((lay.stride + 4) * (int)sizeof(T) * 32 <= 36 * 1024 ? 32 : 16);
  const int LW = (a.N + 1 + 3) & ~3;
  const size_t lds =
      sizeof(T) * ((size_t)m16::kWaves * ((RB + 1) * (lay.stride + 4) + 256 + LW) +
                   n4::kLsSteps);
  if (lds > 160 * 1024) return PDDP_E_UNSUPPORTED;
  const dim3 grid((a.B + m16::kWaves - 1) / m16::kWaves),
      block(kWave * m16::kWaves);
  const bool bounded = a.u_min != nullptr;
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  constexpr bool FAST = sizeof(T) == 4;
#define PDDP_M16N_GO(Bd, C)                                                    \
  do {                                                                         \
    auto kern = m16n::riccati_mfma16_nominal_kernel<T, MODEL, Bd, FAST, C, RB>; \
    const hipError_t e_ = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,         \
        (int)lds);                                                             \
    if (e_ != hipSuccess) return (int)e_;                                      \
    PDDP_LAUNCH(kern, grid, block, lds, st, a, gen, P);                        \
  } while (0)
  if (bounded) { if (chol) PDDP_M16N_GO(true, true); else PDDP_M16N_GO(true, false); }
  else { if (chol) PDDP_M16N_GO(false, true); else PDDP_M16N_GO(false, false); }
#undef PDDP_M16N_GO
  return launch_status();
}

template <typename T>
static int launch_m16_nominal(const pddp_problem& p, const RiccatiArgs<T>& a,
                              const n4d::GenArgs<T>& gen, hipStream_t st) {
  if (p.encoding != PDDP_ENC_IGNORE_UNCERTAINTY || a.N < 1 ||
      ((a.u_min == nullptr) != (a.u_max == nullptr)))
    return PDDP_E_UNSUPPORTED;
  switch (p.model) {
    case PDDP_MODEL_PENDULUM:
      return launch_m16_nominal_model<T, PDDP_MODEL_PENDULUM>(p, a, gen, st);
    case PDDP_MODEL_DOUBLE_CARTPOLE:
      return launch_m16_nominal_model<T, PDDP_MODEL_DOUBLE_CARTPOLE>(p, a, gen, st);
  }
  return PDDP_E_UNSUPPORTED;
}

}  // namespace pddp
