// riccati_n4_quad.hpp - backward Riccati sweep for n = 4, m = 1 with FOUR lanes
// per trajectory (a DPP quad), sixteen trajectories per wavefront.
//
// Why: the 16-lanes-per-trajectory kernels (riccati_n4*.hpp) spend most of a
// step's issue slots moving data between lanes (row rotations, butterflies,
// ds_bpermute transposes, LDS exchanges between roles): ~116 VALU per step for
// 4 trajectories, 2 x ~110 in the two-wavefront forms.  A lone wave issues
// about one instruction every 4.3 cycles, and at large batches the same count
// caps the chip.  Here lane q of a quad owns COLUMN q of every
// 4x4 matrix (V being symmetric that is also row q); every product is an FMA
// whose moving operand is a quad_perm broadcast (DPP modifier, no LDS, no
// butterflies), the scalars of the action dimension (Q_uu, Q_u) are two-step
// quad butterflies, the BoxQP runs replicated in the four lanes:
//
//     T[:, q]    = V F[:, q]                    16 FMA   (V[k][l] = lane l's Vc[k])
//     Qzz[:, q]  = Lzz[:, q] + F^T T[:, q]      16 FMA   (F[k][i] = lane i's Fc[k])
//     Qzz[q, :]  = Lzz[q, :] + F[:, q]^T T      16 FMA   (mirror, for 0.5 (Q + Q^T))
//     Quz[q], Qz[q], w[q] = (V f)[q]             4 FMA each
//     Quu, Qu                                   quad sums
//     V'[:, q]   = sym(Qzz)[:, q] + rank-one / K terms
//
// ~170 instructions per step for 16 trajectories (10.7 per trajectory-step
// against ~29 / ~55).  Lane q of a quad computes element (i, q) and its mirror
// (q, i) from the same products in the same order as lane i does, so V stays
// exactly symmetric without any transpose.
//
// Records stream HBM -> LDS by full-wave 16-byte LDS-DMA instructions, R
// steps ahead: 4 records (48 chunks) per instruction, the 16 spare lanes
// re-load chunks into padding; a group of 4 records is placed at a stride of
// 1040 bytes so that the column gathers F[k][q] of 64 lanes fall into 32
// different banks.
//
// Restates pddp/controllers/ilqr.py:489-526 (Q) and :529-674 (backward) for
// m = 1, all four gain branches; BoxQP = QpClosed (riccati_n4.hpp) with the
// reference's loop (utils/constraint.py:150-266) out of line (n4::BoxQp1 on
// the quad: four back-tracking candidates per round).  Summation order differs from the reference's (results agree to
// rounding); the Cholesky branches form the second Q() of ilqr.py:590-592 as
// f^T (V + reg I) F = f^T V F + reg f^T F.
#pragma once

#include <cstdlib>

#include "riccati_n4.hpp"

#ifdef PDDP_Q4_NOSLOW  // timing experiment: never take the loop fall-back
#define PDDP_Q4_SLOWTEST(x) (__any(x) && false)
#else
#define PDDP_Q4_SLOWTEST(x) __any(x)
#endif

namespace pddp {

namespace n4q {

using n4::div_;
using n4::dpp;
using n4::fma_;
using n4::mul_nc;
using n4::opaque;
using n4::sqrtx;

constexpr int kRec = n4::kRec;    // 48 scalars per record
constexpr int kGain = n4::kGain;  // k, K[0..3]

// value of lane I of the quad (quad_perm broadcast)
template <int I, typename T>
PDDP_DEV T qb(T v) {
  return dpp<(I | (I << 2) | (I << 4) | (I << 6))>(v);
}
// ---- float fast path: FMAs with the quad broadcast as DPP operand.  The
// compiler folds a DPP move into v_mul / v_add but not into v_fmac (its
// accumulator is tied), so every broadcast FMA costs a v_mov_b32_dpp of its
// own - 52 per step.  Hand-written v_fmac_f32_dpp (dst += dpp(src0) * src1):
// four per statement so that the scheduler can still interleave them with the
// BoxQP chain.  `NOP`: "s_nop 1" first where a DPP source may have been written
// by one of the two preceding VALU instructions (the compiler does not see
// into the statement).
#define PDDP_DPPQ(I) " quad_perm:[" #I "," #I "," #I "," #I \
                     "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// a_k = dpp_0(v_k) * f   (k = 0..3)
PDDP_DEV void dpp_mul4_src(float& a0, float& a1, float& a2, float& a3, float v0,
                           float v1, float v2, float v3, float f) {
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, %4, %8" PDDP_DPPQ(0)
      "v_mul_f32_dpp %1, %5, %8" PDDP_DPPQ(0)
      "v_mul_f32_dpp %2, %6, %8" PDDP_DPPQ(0)
      "v_mul_f32_dpp %3, %7, %8" PDDP_DPPQ(0)
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(f));
}
// a_k += dpp_L(v_k) * f   (k = 0..3): one lane L, four sources
#define PDDP_FMAC4_SRC(L)                                                     \
  PDDP_DEV void dpp_fmac4_src##L(float& a0, float& a1, float& a2, float& a3,  \
                                 float v0, float v1, float v2, float v3,      \
                                 float f) {                                   \
    asm("v_fmac_f32_dpp %0, %4, %8" PDDP_DPPQ(L)                              \
        "v_fmac_f32_dpp %1, %5, %8" PDDP_DPPQ(L)                              \
        "v_fmac_f32_dpp %2, %6, %8" PDDP_DPPQ(L)                              \
        "v_fmac_f32_dpp %3, %7, %8" PDDP_DPPQ(L)                              \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                              \
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(f));                        \
  }
PDDP_FMAC4_SRC(1)
PDDP_FMAC4_SRC(2)
PDDP_FMAC4_SRC(3)
#undef PDDP_FMAC4_SRC
// a_i += dpp_i(v) * f   (i = 0..3): one source, four lanes.  NOP as above.
template <bool NOP>
PDDP_DEV void dpp_fmac4_lanes(float& a0, float& a1, float& a2, float& a3,
                              float v, float f) {
  if constexpr (NOP) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %4, %5" PDDP_DPPQ(0)
        "v_fmac_f32_dpp %1, %4, %5" PDDP_DPPQ(1)
        "v_fmac_f32_dpp %2, %4, %5" PDDP_DPPQ(2)
        "v_fmac_f32_dpp %3, %4, %5" PDDP_DPPQ(3)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(v), "v"(f));
  } else {
    asm("v_fmac_f32_dpp %0, %4, %5" PDDP_DPPQ(0)
        "v_fmac_f32_dpp %1, %4, %5" PDDP_DPPQ(1)
        "v_fmac_f32_dpp %2, %4, %5" PDDP_DPPQ(2)
        "v_fmac_f32_dpp %3, %4, %5" PDDP_DPPQ(3)
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(v), "v"(f));
  }
}
// a += dpp_0(v) f0 + dpp_1(v) f1 + dpp_2(v) f2 + dpp_3(v) f3
PDDP_DEV void dpp_dot4(float& a, float v, float f0, float f1, float f2,
                       float f3) {
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %1, %2" PDDP_DPPQ(0)
      "v_fmac_f32_dpp %0, %1, %3" PDDP_DPPQ(1)
      "v_fmac_f32_dpp %0, %1, %4" PDDP_DPPQ(2)
      "v_fmac_f32_dpp %0, %1, %5" PDDP_DPPQ(3)
      : "+v"(a)
      : "v"(v), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
}
#undef PDDP_DPPQ

// sum over the quad, bit-identical in its four lanes
template <typename T>
PDDP_DEV T quad_sum(T x) {
#pragma clang fp contract(off)
  x = opaque(x);
  const T y = x + dpp<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(x);
  return y + dpp<(2 | (3 << 2) | (0 << 4) | (1 << 6))>(y);
}

// The reference's projected-Newton loop for ONE scalar problem
// (utils/constraint.py:150-266: exit codes, the possibly stale `free` flag),
// solved by the whole wavefront: every argument is wave-uniform, lane l
// evaluates back-tracking candidate n = l (then l + 64) of the scan (:248-259)
// and the first passing n is a ballot + count-trailing-zeros.  The fall-back
// of QpClosed in the quad kernel, where one slow trajectory would otherwise
// hold up the sixteen of its wavefront for a sequential scan.  Same arithmetic
// per candidate as n4::BoxQp1.
template <typename T, bool FAST>
__device__ __noinline__ n4::SlowQpOut<T> boxqp1_wave(T x0, T Q, T c, T lo,
                                                     T hi, const T* ls_tail,
                                                     int lane) {
  constexpr T kMinGrad = T(1e-8), kTol = T(1e-8), kArmijo = T(0.1);
  constexpr int kFail = n4::kLs.n_fail;
  static_assert(kFail < 2 * kWave && n4::kLsSteps <= 2 * kWave, "");
  auto obj = [&](T v) { return T(0.5) * ((v * Q) * v) + v * c; };
  T x = clamp1(x0, lo, hi);
  x = ((x - x != T(0)) && (x == x)) ? T(0) : x;  // x[isinf(x)] = 0   (:179)
  T f = obj(x), old_f = T(0);
  bool free_ = true;
  int result = 0;
  const T U = sqrtx<FAST>(Q);
  const T newton = -div_<FAST>(div_<FAST>(c, U), U);  // -potrs(g, U)
  const bool not_pd = !(Q > T(0)) || !is_finite(Q);
  const int n1 = lane + kWave;
  const T st0 = ls_tail[lane];
  const T st1 = ls_tail[n1 < n4::kLsSteps ? n1 : n4::kLsSteps - 1];
  for (int it = 0; it < 100; ++it) {
    if (it > 0 && (old_f - f) < kTol * abs_(old_f)) {  // (:191-193)
      result = 4;
      break;
    }
    old_f = f;
    const T g = Q * x + c;
    const bool ncl = ((x == lo) && (g > T(0))) || ((x == hi) && (g < T(0)));
    free_ = !ncl;                                    // (:200-204)
    if (ncl) { result = 6; break; }                  // (:207-209)
    if (it == 0 && not_pd) { result = -1; break; }   // (:212-228)
    if (abs_(g) < kMinGrad) { result = 5; break; }   // (:231-234)
    const T search = newton - x;                     // (:237-239)
    const T sdotg = search * g;
    T xn = n4::clampq<FAST>(x + st0 * search, lo, hi);
    T fn = obj(xn);
    bool ok = !(div_<FAST>(fn - old_f, st0 * sdotg) < kArmijo) || lane >= kFail;
    unsigned long long bal = __ballot(ok);
    int n = 0;
    if (bal == 0) {  // candidates 64 .. : n >= kFail always passes
      xn = n4::clampq<FAST>(x + st1 * search, lo, hi);
      fn = obj(xn);
      ok = !(div_<FAST>(fn - old_f, st1 * sdotg) < kArmijo) || n1 >= kFail;
      bal = __ballot(ok);
      n = kWave;
    }
    const int first = __builtin_ctzll(bal);
    n += first;
    x = __shfl(xn, first);
    f = __shfl(fn, first);
    if (n >= kFail) { result = 2; break; }           // step < min_step
  }
  n4::SlowQpOut<T> o;
  o.x = x;
  o.U = U;
  o.result_free = result * 2 + (free_ ? 1 : 0);
  return o;
}

#ifdef PDDP_QP_STATS
__device__ unsigned long long g_quad_stats[8];
#endif

template <typename T>
struct QuadGeom {
  static constexpr int CB = 16;                          // bytes per DMA chunk
  static constexpr int CH = kRec * (int)sizeof(T) / CB;  // chunks per record
  static constexpr int RPI = 48 / CH;        // records per DMA instruction
  static constexpr int NI = 16 / RPI;        // DMA instructions per step
  // a DMA instruction writes 1024 B; groups sit 1040 B apart (bank skew)
  static constexpr int GSB = 1040;
  static constexpr int GS = GSB / (int)sizeof(T);   // group stride, scalars
  static constexpr int SLOT = NI * GS;               // scalars per ring slot
};

// LOOP: every BoxQP through the reference's loop (boxqp1_wave) - the exact
// A/B twin of the closed form, and the test vehicle of the fall-back.
template <typename T, bool CHOL, bool BOUNDED, bool FAST, int R, int WPB,
          bool LOOP = false>
__global__ __launch_bounds__(kWave * WPB) void riccati_n4_quad_kernel(
    RiccatiArgs<T> a) {
  using G = QuadGeom<T>;
  constexpr int NI = G::NI, RPI = G::RPI, CH = G::CH, CB = G::CB;
  constexpr int kTraj = 16;  // trajectories per wavefront
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ T ls_tail[n4::kLsSteps];  // T(0.6^n): the back-tracking steps
  const int lane = threadIdx.x & (kWave - 1);
  const int wave =
      WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  T* ring = reinterpret_cast<T*>(smem_raw) + (size_t)wave * R * G::SLOT;
  if constexpr (BOUNDED) {
    for (int i = threadIdx.x; i < n4::kLsSteps; i += kWave * WPB)
      ls_tail[i] = (T)n4::kLs.v[i];
    __syncthreads();
  }

  const int q = lane & 3, tr = lane >> 2;
  const int N = a.N;
  const int b0 = (blockIdx.x * WPB + wave) * kTraj;
  if (b0 >= a.B) return;  // a whole wave past the batch
  const int b = b0 + tr;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return;

  const T reg = (T)a.reg[bc];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) {
    umin = a.u_min[0];
    umax = a.u_max[0];
  }

  // ---- DMA source addressing: instruction I loads records I*RPI .. of the
  // wave, chunk c of them in lane c; lanes >= 48 re-load chunk c - 48 (into
  // the group's padding).  Wave-uniform 64-bit base + 32-bit lane offset.
  const char* rec_w =
      reinterpret_cast<const char*>(a.rec + (size_t)b0 * (size_t)(N + 1) * kRec);
  uint32_t src_off[NI];
#pragma unroll
  for (int I = 0; I < NI; ++I) {
    const int c = lane < 48 ? lane : lane - 48;
    const int ri = I * RPI + c / CH, part = c - (c / CH) * CH;
    int tb = b0 + ri;
    tb = tb < a.B ? tb : a.B - 1;
    src_off[I] =
        (uint32_t)((tb - b0) * (N + 1) * kRec * (int)sizeof(T) + part * CB);
  }
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(n4::lds_addr(ring));
  auto dma = [&](int slot, int t) {
    const int tt = t < 0 ? 0 : t;  // tail: harmless reload keeps vmcnt exact
    const uint32_t toff = (uint32_t)tt * (uint32_t)(kRec * sizeof(T));
#pragma unroll
    for (int I = 0; I < NI; ++I)
      n4::lds_dma16(rec_w, src_off[I] + toff,
                    ring_lds + (uint32_t)((slot * G::SLOT + I * G::GS) *
                                          (int)sizeof(T)));
  };

  // ---- this lane's record inside a slot (scalars)
  const int rbase = (tr / RPI) * G::GS + (tr % RPI) * kRec;
  const int oq = rbase + q;       // + 4k: F[k][q];  + 16 + 4i: Lzz[i][q]
  const int or4 = rbase + 4 * q;  // + 16: Lzz[q][0..3]

  // ---- terminal value function: column q of V = L_zz[N], V_z[q] = L_z[N][q]
  const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
  T Vc0 = term[16 + 0 + q], Vc1 = term[16 + 4 + q], Vc2 = term[16 + 8 + q],
    Vc3 = term[16 + 12 + q];
  T vz = term[40 + q];

#pragma unroll
  for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
  n4::wait_vmcnt<0>();

  T kprev = T(0);
  int status = PDDP_BWD_OK;
  char* gains_w =
      reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
  // byte offset of K[q] of step t (k sits one scalar before K[0])
  uint32_t gout_off = (uint32_t)(
      ((bc - b0) * N * kGain + (N - 1) * kGain + 1 + q) * (int)sizeof(T));

  struct Words {
    T F0, F1, F2, F3;      // F[k][q]
    T Lc0, Lc1, Lc2, Lc3;  // Lzz[i][q]
    T Lr0, Lr1, Lr2, Lr3;  // Lzz[q][i]
    T f0, f1, f2, f3, fq;  // F_u, F_u[q]
    T Luz, Lz, Luu, Lu, Un;
  };
  auto gather = [&](int slot) {
    const T* rc = ring + slot * G::SLOT;
    Words w;
    w.F0 = rc[oq]; w.F1 = rc[oq + 4]; w.F2 = rc[oq + 8]; w.F3 = rc[oq + 12];
    w.Lc0 = rc[oq + 16]; w.Lc1 = rc[oq + 20]; w.Lc2 = rc[oq + 24];
    w.Lc3 = rc[oq + 28];
    w.Lr0 = rc[or4 + 16]; w.Lr1 = rc[or4 + 17]; w.Lr2 = rc[or4 + 18];
    w.Lr3 = rc[or4 + 19];
    w.f0 = rc[rbase + 32]; w.f1 = rc[rbase + 33]; w.f2 = rc[rbase + 34];
    w.f3 = rc[rbase + 35];
    w.fq = rc[oq + 32];
    w.Luz = rc[oq + 36];
    w.Lz = rc[oq + 40];
    w.Luu = rc[rbase + 44]; w.Lu = rc[rbase + 45];
    w.Un = BOUNDED ? rc[rbase + 46] : T(0);
    return w;
  };

  int t = N - 1;
  auto step = [&](const Words& w, const int s) {
    const bool alive = counted & (status == PDDP_BWD_OK);
    const T kprev_in = kprev;  // warm start of this step's BoxQP
    // ---- T[:, q] = V F[:, q]: T[k][q] = sum_l V[k][l] F[l][q], V[k][l] =
    // (column l)[k] = lane l's Vc_k
    T T0, T1, T2, T3;
    if constexpr (sizeof(T) == 4) {
      dpp_mul4_src(T0, T1, T2, T3, Vc0, Vc1, Vc2, Vc3, w.F0);
      dpp_fmac4_src1(T0, T1, T2, T3, Vc0, Vc1, Vc2, Vc3, w.F1);
      dpp_fmac4_src2(T0, T1, T2, T3, Vc0, Vc1, Vc2, Vc3, w.F2);
      dpp_fmac4_src3(T0, T1, T2, T3, Vc0, Vc1, Vc2, Vc3, w.F3);
    } else {
      T0 = qb<0>(Vc0) * w.F0; T1 = qb<0>(Vc1) * w.F0;
      T2 = qb<0>(Vc2) * w.F0; T3 = qb<0>(Vc3) * w.F0;
      T0 = fma_(qb<1>(Vc0), w.F1, T0); T1 = fma_(qb<1>(Vc1), w.F1, T1);
      T2 = fma_(qb<1>(Vc2), w.F1, T2); T3 = fma_(qb<1>(Vc3), w.F1, T3);
      T0 = fma_(qb<2>(Vc0), w.F2, T0); T1 = fma_(qb<2>(Vc1), w.F2, T1);
      T2 = fma_(qb<2>(Vc2), w.F2, T2); T3 = fma_(qb<2>(Vc3), w.F2, T3);
      T0 = fma_(qb<3>(Vc0), w.F3, T0); T1 = fma_(qb<3>(Vc1), w.F3, T1);
      T2 = fma_(qb<3>(Vc2), w.F3, T2); T3 = fma_(qb<3>(Vc3), w.F3, T3);
    }

    // ---- the scalars of the action dimension first (they head the BoxQP
    // chain): w[q] = (V f)[q] = sum_l V[l][q] f[l] (symmetry), Q_uu, Q_u
    T wq = Vc0 * w.f0;
    wq = fma_(Vc1, w.f1, wq);
    wq = fma_(Vc2, w.f2, wq);
    wq = fma_(Vc3, w.f3, wq);
    const T Quu = w.Luu + quad_sum(w.fq * wq);
    const T Qu = w.Lu + quad_sum(w.fq * vz);
    // Q_uz[q] = L_uz[q] + sum_k f[k] T[k][q]
    T Quz = fma_(w.f0, T0, w.Luz);
    Quz = fma_(w.f1, T1, Quz);
    Quz = fma_(w.f2, T2, Quz);
    Quz = fma_(w.f3, T3, Quz);
    T Quug = Quu, Quzg = Quz;
    if constexpr (CHOL) {
      // second Q() with V + reg I (ilqr.py:590-592):
      // f^T (V + reg I) f = f^T V f + reg f^T f, likewise for Q_uz
      const T ff = quad_sum(w.fq * w.fq);
      T fF = w.f0 * w.F0;
      fF = fma_(w.f1, w.F1, fF);
      fF = fma_(w.f2, w.F2, fF);
      fF = fma_(w.f3, w.F3, fF);
      Quug = fma_(reg, ff, Quu);
      Quzg = fma_(reg, fF, Quz);
    }

    // ---- gains, part 1: closed-form BoxQP / unconstrained solve (replicated
    // in the quad; independent of the products below)
    T kt = T(0), Uch = T(1), inv = T(0);
    bool Kzero = false, by_inv = false, fail = false;
    int st = PDDP_BWD_OK;
    n4::QpClosed<T, FAST> qc;
    T qp_Q = T(1);
    if constexpr (BOUNDED) {
      if constexpr (!CHOL) {
        if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
        const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
        qp_Q = e + reg;                             // ilqr.py:634
      } else {
        qp_Q = Quug;
      }
      qc.solve(kprev, qp_Q, Qu, umin - w.Un, umax - w.Un);
    } else if constexpr (!CHOL) {
      if (!is_finite(Quu)) st = PDDP_BWD_NAN;
      T e = (Quu < T(0)) ? T(1e-12) : Quu;
      e += reg;
      inv = div_<FAST>(T(1), e);
      kt = -(inv * Qu);
      by_inv = true;
      if (kt != kt) st = PDDP_BWD_NAN;
    } else {
      if (!(Quug > T(0)) || !is_finite(Quug)) st = PDDP_BWD_NOT_PD;
      Uch = sqrtx<FAST>(Quug);
      kt = -div_<FAST>(div_<FAST>(Qu, Uch), Uch);
    }

    // ---- Qzz[:, q] (column) and Qzz[q, :] (row, the mirror): lane q's row
    // element i and lane i's column element q are the same products in the
    // same order, so 0.5 (col + row) is exactly symmetric
    T C0 = w.Lc0, C1 = w.Lc1, C2 = w.Lc2, C3 = w.Lc3;
    T R0 = w.Lr0, R1 = w.Lr1, R2 = w.Lr2, R3 = w.Lr3;
    if constexpr (sizeof(T) == 4) {
      // C_i += F[k][i] T[k][q]   (F_k of lane i; loaded from LDS: no hazard)
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F0, T0);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F1, T1);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F2, T2);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F3, T3);
      // R_i += T[k][i] F[k][q]   (T_k of lane i)
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T0, w.F0);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T1, w.F1);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T2, w.F2);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T3, w.F3);
    } else {
      C0 = fma_(qb<0>(w.F0), T0, C0); C1 = fma_(qb<1>(w.F0), T0, C1);
      C2 = fma_(qb<2>(w.F0), T0, C2); C3 = fma_(qb<3>(w.F0), T0, C3);
      C0 = fma_(qb<0>(w.F1), T1, C0); C1 = fma_(qb<1>(w.F1), T1, C1);
      C2 = fma_(qb<2>(w.F1), T1, C2); C3 = fma_(qb<3>(w.F1), T1, C3);
      C0 = fma_(qb<0>(w.F2), T2, C0); C1 = fma_(qb<1>(w.F2), T2, C1);
      C2 = fma_(qb<2>(w.F2), T2, C2); C3 = fma_(qb<3>(w.F2), T2, C3);
      C0 = fma_(qb<0>(w.F3), T3, C0); C1 = fma_(qb<1>(w.F3), T3, C1);
      C2 = fma_(qb<2>(w.F3), T3, C2); C3 = fma_(qb<3>(w.F3), T3, C3);
      R0 = fma_(w.F0, qb<0>(T0), R0); R1 = fma_(w.F0, qb<1>(T0), R1);
      R2 = fma_(w.F0, qb<2>(T0), R2); R3 = fma_(w.F0, qb<3>(T0), R3);
      R0 = fma_(w.F1, qb<0>(T1), R0); R1 = fma_(w.F1, qb<1>(T1), R1);
      R2 = fma_(w.F1, qb<2>(T1), R2); R3 = fma_(w.F1, qb<3>(T1), R3);
      R0 = fma_(w.F2, qb<0>(T2), R0); R1 = fma_(w.F2, qb<1>(T2), R1);
      R2 = fma_(w.F2, qb<2>(T2), R2); R3 = fma_(w.F2, qb<3>(T2), R3);
      R0 = fma_(w.F3, qb<0>(T3), R0); R1 = fma_(w.F3, qb<1>(T3), R1);
      R2 = fma_(w.F3, qb<2>(T3), R2); R3 = fma_(w.F3, qb<3>(T3), R3);
    }
    // 0.5 (Q + Q^T), column q (the halving is exact: folded in below)
    const T S0 = C0 + R0, S1 = C1 + R1, S2 = C2 + R2, S3 = C3 + R3;
    // Q_z[q] = L_z[q] + sum_k F[k][q] V_z[k]
    T Qz = w.Lz;
    if constexpr (sizeof(T) == 4) {
      dpp_dot4(Qz, vz, w.F0, w.F1, w.F2, w.F3);
    } else {
      Qz = fma_(w.F0, qb<0>(vz), Qz);
      Qz = fma_(w.F1, qb<1>(vz), Qz);
      Qz = fma_(w.F2, qb<2>(vz), Qz);
      Qz = fma_(w.F3, qb<3>(vz), Qz);
    }

    if constexpr (BOUNDED) {
      kt = qc.x;
      Uch = qc.U;
      inv = qc.inv;
      Kzero = !qc.free_;
      fail = qc.fail;
    }
    // the rest of the step given the minimiser: K, status, stores, value update
    auto tail = [&]() {
      int stt = st != PDDP_BWD_OK ? st : (fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
      T Kq;
      if (by_inv) {
        Kq = -(inv * Quzg);
      } else if constexpr (BOUNDED && FAST) {
        Kq = Kzero ? T(0) : -(Quzg * inv);
      } else {
        Kq = Kzero ? T(0) : -div_<FAST>(div_<FAST>(Quzg, Uch), Uch);
      }
      if constexpr (!CHOL && !BOUNDED) {
        // NaN anywhere in K raises too (ilqr.py:639-640)
        const bool any4 = quad_sum((Kq != Kq) ? T(1) : T(0)) != T(0);
        if (any4 && stt == PDDP_BWD_OK) stt = PDDP_BWD_NAN;
      }
      status = (alive & (stt != PDDP_BWD_OK)) ? stt : status;

      // ---- store K[q] (every lane) and k (lane 0 of the quad); trajectories
      // past the batch store nothing, failed ones store junk
      {
        T* dst = reinterpret_cast<T*>(gains_w + gout_off);
        if (exists) {
          *dst = Kq;
          if (q == 0) dst[-1] = kt;
        }
      }
      kprev = kt;

      // ---- value update with the un-regularised Q_uu, Q_uz
      // (ilqr.py:619-625, 664-672): V_z' = Q_z + K (Q_u + Q_uu k) + Q_uz k
      vz = fma_(Quz, kt, fma_(Kq, fma_(Quu, kt, Qu), Qz));
      // V'[i][q] = 0.5 (Qzz + Qzz^T)[i][q] + Q_uu K_i K_q + K_i Quz_q + Quz_i K_q
      // every product rounded on its own (mul_nc), sums of commuting pairs:
      // lane q's element i and lane i's element q are bit-identical
      // Q_uu K_i K_q must not depend on which of the two lanes forms it:
      // (Q_uu K_q) K_i on lane q vs (Q_uu K_i) K_q on lane i differ in the
      // last bit, so the product K_i K_q is formed first (it commutes)
      const T KK0 = mul_nc(qb<0>(Kq), Kq), KK1 = mul_nc(qb<1>(Kq), Kq),
              KK2 = mul_nc(qb<2>(Kq), Kq), KK3 = mul_nc(qb<3>(Kq), Kq);
      const T X0 = mul_nc(qb<0>(Kq), Quz) + mul_nc(qb<0>(Quz), Kq);
      const T X1 = mul_nc(qb<1>(Kq), Quz) + mul_nc(qb<1>(Quz), Kq);
      const T X2 = mul_nc(qb<2>(Kq), Quz) + mul_nc(qb<2>(Quz), Kq);
      const T X3 = mul_nc(qb<3>(Kq), Quz) + mul_nc(qb<3>(Quz), Kq);
      Vc0 = fma_(T(0.5), S0, fma_(Quu, KK0, X0));
      Vc1 = fma_(T(0.5), S1, fma_(Quu, KK1, X1));
      Vc2 = fma_(T(0.5), S2, fma_(Quu, KK2, X2));
      Vc3 = fma_(T(0.5), S3, fma_(Quu, KK3, X3));
    };
#ifdef PDDP_QP_STATS
    if constexpr (BOUNDED) {
      if (alive && q == 0) {
        atomicAdd(&g_quad_stats[0], 1ull);
        if (qc.slow) atomicAdd(&g_quad_stats[1], 1ull);
        for (int bit = 0; bit < 6; ++bit)
          if (qc.dbg & (1 << bit)) atomicAdd(&g_quad_stats[2 + bit], 1ull);
      }
    }
#endif
    // The step is finished on the closed form's answer first; the (rare) test
    // whether some trajectory of the wave needs the reference's loop comes
    // AFTER it, off the dependent chain: the tail only rewrites state from
    // this step's products (V', V_z', k, the stored gains), so it is simply
    // run again with the loop's result.
    tail();
    if constexpr (BOUNDED) {
      const bool slow = (LOOP || qc.slow) && alive;
      if (__builtin_expect(PDDP_Q4_SLOWTEST(slow), 0)) {
        // one slow trajectory at a time on the whole wavefront (boxqp1_wave)
        unsigned long long todo = __ballot(slow && q == 0);
        const T lo_b = umin - w.Un, hi_b = umax - w.Un;
        while (todo != 0) {
          const int src = __builtin_ctzll(todo);
          todo &= todo - 1;
          const n4::SlowQpOut<T> o = boxqp1_wave<T, FAST>(
              __shfl(kprev_in, src), __shfl(qp_Q, src), __shfl(Qu, src),
              __shfl(lo_b, src), __shfl(hi_b, src), ls_tail, lane);
          const bool mine = (lane >> 2) == (src >> 2);
          kt = mine ? o.x : kt;
          Uch = mine ? o.U : Uch;
          Kzero = mine ? ((o.result_free & 1) == 0) : Kzero;
          fail = mine ? (o.result_free < 2) : fail;
        }
        tail();
      }
    }
    gout_off -= (uint32_t)(kGain * sizeof(T));
    dma(s, t - R);
  };

  // Two word sets alternate (one in use, one being gathered for the next
  // step).  DMA(t-1) has landed once at most (R-2) younger {2 stores, NI DMAs}
  // groups are outstanding.
  Words wa = gather(0), wb = wa;
  while (t >= 0) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      if (t < 0) break;
      n4::wait_vmcnt<(R - 2) * (2 + NI)>();
      wb = gather((s + 1) % R);  // (at t == 0: a stale slot, unused)
      step(wa, s);
      --t;
      if (t < 0) break;
      n4::wait_vmcnt<(R - 2) * (2 + NI)>();
      wa = gather((s + 2) % R);
      step(wb, s + 1);
      --t;
    }
  }
  n4::wait_vmcnt<0>();
  if (counted && q == 0) a.status[bc] = status;
}

// c and w of the rank-one form of the value update (ilqr.py:664-672 with
// K = -s Quz):  V' = sym(Qzz) + c Quz Quz^T,  V_z' = Qz + w Quz
template <typename T>
PDDP_DEV void rank_one_coeffs(T k, T s, T Quu, T Qu, T& c, T& w) {
  c = n4::mul_nc(s, n4::fma_(s, Quu, T(-2)));
  w = n4::fma_(-s, n4::fma_(Quu, k, Qu), k);
}

}  // namespace n4q

// 16 trajectories per wavefront; WPB wavefronts per workgroup (independent
// after launch): 1 up to 4096 trajectories (256 waves: one per CU), 4 above
// (one per SIMD of a CU by construction).
template <typename T>
static int launch_n4_quad(const RiccatiArgs<T>& a, hipStream_t st,
                          bool fast_math, bool loop_always = false) {
  constexpr int R = 8;
  using G = n4q::QuadGeom<T>;
  const bool bounded = a.u_min != nullptr;
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  const int waves = (a.B + 15) / 16;
  const size_t lds1 = (size_t)R * G::SLOT * sizeof(T);
  // wavefronts per workgroup: as many as keep at least one workgroup on every
  // CU (256) and fit the CU's 160 KB of LDS - 512 waves as 128 workgroups of
  // four left half the chip idle (B = 8192: 81 against 66 us; B = 12288 runs
  // 89 us as 192 workgroups of four, 94 us as 384 of two)
  int wpb = waves >= 768 ? 4 : (waves >= 512 ? 2 : 1);
  while (wpb > 1 && (size_t)wpb * lds1 > 150 * 1024) wpb /= 2;
  if (const char* e = getenv("PDDP_QUAD_WPB")) {  // (A/B measurements)
    const int w = atoi(e);
    if (w == 1 || w == 2 || w == 4) wpb = w;
  }
  if (loop_always) {  // IEEE, bounded, one wave per workgroup
    if (!bounded) return PDDP_E_UNSUPPORTED;
    auto k0 = n4q::riccati_n4_quad_kernel<T, false, true, false, R, 1, true>;
    auto k1 = n4q::riccati_n4_quad_kernel<T, true, true, false, R, 1, true>;
    auto kern = chol ? k1 : k0;
    const hipError_t e = hipFuncSetAttribute(
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds1);
    if (e != hipSuccess) return (int)e;
    PDDP_LAUNCH(kern, dim3(waves), dim3(kWave), lds1, st, a);
    return launch_status();
  }
#define PDDP_Q4_GO(C, Bd, F, W)                                               \
  do {                                                                        \
    auto kern = n4q::riccati_n4_quad_kernel<T, C, Bd, F, R, W>;               \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)(W * lds1));                                                     \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, dim3((waves + W - 1) / W), dim3(kWave * W), W * lds1,   \
                st, a);                                                       \
  } while (0)
#define PDDP_Q4_LAUNCH(C, Bd, F)                                              \
  do {                                                                        \
    if (wpb == 4) PDDP_Q4_GO(C, Bd, F, 4);                                    \
    else if (wpb == 2) PDDP_Q4_GO(C, Bd, F, 2);                               \
    else PDDP_Q4_GO(C, Bd, F, 1);                                             \
  } while (0)
#define PDDP_Q4_BRANCH(F)                                                     \
  do {                                                                        \
    if (chol) {                                                               \
      if (bounded) PDDP_Q4_LAUNCH(true, true, F);                             \
      else PDDP_Q4_LAUNCH(true, false, F);                                    \
    } else {                                                                  \
      if (bounded) PDDP_Q4_LAUNCH(false, true, F);                            \
      else PDDP_Q4_LAUNCH(false, false, F);                                   \
    }                                                                         \
  } while (0)
  if (fast_math && sizeof(T) == 4) PDDP_Q4_BRANCH(true);
  else PDDP_Q4_BRANCH(false);
#undef PDDP_Q4_BRANCH
#undef PDDP_Q4_LAUNCH
#undef PDDP_Q4_GO
  return launch_status();
}

}  // namespace pddp
