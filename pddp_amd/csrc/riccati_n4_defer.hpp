// riccati_n4_defer.hpp - the n = 4, m = 1 bounded eig-clamp sweep (branch B,
// ilqr.py:629-672) with the rank-one part of the value update DEFERRED, so
// that the only dependent chain that crosses a step is the scalar BoxQP.
//
// The reference's step is linear in (V_{t+1}, V_z,t+1) up to the BoxQP, and
// the BoxQP's result enters the next value function through two scalars:
//
//     V_t   = sym(Qzz_t) + c_t Quz_t^T Quz_t       c_t = s_t (s_t Quu_t - 2)
//     V_z,t = Qz_t + w_t Quz_t                     w_t = k_t - s_t (Qu_t + Quu_t k_t)
//
// (K_t = -s_t Quz_t, s_t = 0 on a clamped step; riccati_n4_pipe.hpp).  Write
//
//     V_{t+1}   = W_{t+1} + c_{t+1} y_{t+1} y_{t+1}^T + c_{t+2} y_{t+2} y_{t+2}^T
//     V_z,t+1   = r_{t+1} + w_{t+1} y_{t+1}         + w_{t+2} y_{t+2}
//
// with y_j = Quz_j carried to time t+1 by F^T: the 4x4 products run on W
// alone and take a rank-one term in only two steps after it was born
// (W_t = S0_t + c_{t+2} y y^T), the young terms reach the action scalars
// through dot products g_{j,t} = f_t . y_j:
//
//     Quu_t = A00_t + c_{t+2} g_{t+2,t}^2 + c_{t+1} g_{t+1,t}^2
//     Qu_t  = B00_t + w_{t+2} g_{t+2,t}   + w_{t+1} g_{t+1,t}
//     Quz_t = Quz0_t + c_{t+2} g_{t+2,t} (F_t^T y_{t+2}) + c_{t+1} g_{t+1,t} (F_t^T y_{t+1})
//
// and the youngest dot product is itself affine in the scalar that is still
// missing: g_{t+1,t} = G0_t + c_{t+2} g_{t+2,t+1} g_{t+2,t}.  Four wavefronts
// of a workgroup (sixteen trajectories, the quad mapping of
// riccati_n4_quad.hpp: lane q of a quad = column q), one phase per step, one
// s_barrier per phase; in the phase in which
//
//   Q  solves the BoxQP of step tq (closed form, riccati_n4.hpp QpClosed; the
//      reference's loop as fall-back) from coefficients it completed one phase
//      earlier - its step is two FMAs and the BoxQP, it waits for nobody,
//   Y  finalises the vector y_tq (c_{tq+1} arrived), carries it to tq - 1 and
//      tq - 2, forms the dot products and the vector part (r, B00, G0) of step
//      tq - 2, stores the gains of step tq + 1,
//   M  takes c_{tq+1} into W and forms the 4x4 products of step tq - 2 on the
//      4x4x1 matrix instruction (sixteen independent 4x4 blocks per wavefront:
//      one block per trajectory, operands and result in the quad layout),
//   P  streams the records (four LDS-DMA instructions per step).
//
// Every role reads what the others published in the PREVIOUS phase (two
// parities of each exchange buffer).  tools/defer_proto.py is the numpy
// restatement of this schedule, checked against the oracle: same results as
// the plain recursion to rounding (fp64 ~5e-12 on the oracle's gains; the
// products run on W, which lacks the last two - negative - rank-one terms, so
// on a diverging value function the cancellation is somewhat worse than the
// plain form's).
#pragma once

#include "models.hpp"
#include "riccati_n4_qpipe.hpp"

namespace pddp {

namespace n4d {

using n4::fma_;
using n4::mul_nc;
using n4q::kGain;
using n4q::kRec;
using n4q::qb;
using n4q::quad_sum;

constexpr int kThreads = 4 * kWave;
constexpr int kTraj = 16;

#ifdef PDDP_QP_STATS
// cycles each role waits at the phase barrier ([role]) and in total
// ([4 + role]); tools/defer_wait.py
__device__ unsigned long long g_defer_stats[8];
#define PDDP_DW_DECL unsigned long long wait_acc = 0; const long long t_begin = clock64(); PDDP_DW_SEGDECL
#define PDDP_DW_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t0_ = clock64(); seg_acc[7] += (unsigned long long)(t0_ - seg_last); asm volatile("s_barrier" ::: "memory"); seg_last = clock64(); wait_acc += (unsigned long long)(seg_last - t0_); } while (0)
#define PDDP_DW_END(ROLE) do { if (lane == 0) { atomicAdd(&g_defer_stats[ROLE], wait_acc); atomicAdd(&g_defer_stats[4 + ROLE], (unsigned long long)(clock64() - t_begin)); for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_defer_seg[ROLE][i_], seg_acc[i_]); } } while (0)
// segment profile of a phase: cycles from the previous stamp (or the phase
// barrier) to stamp I; the s_memtime read drains lgkmcnt, so a segment that
// follows LDS reads shows their full latency
__device__ unsigned long long g_defer_seg[4][8];
// time marks of workgroup 0: [wave M / generator][begin, first phase, last
// phase done, end]
__device__ long long g_defer_marks[2][4];
#define PDDP_DW_MARK(W, I) do { if (blockIdx.x == 0 && lane == 0) g_defer_marks[W][I] = clock64(); } while (0)
#define PDDP_DW_COUNT(I)
#define PDDP_DW_SEGDECL unsigned long long seg_acc[8] = {}; long long seg_last = clock64();
#define PDDP_DW_STAMP(I) do { const long long n_ = clock64(); seg_acc[I] += (unsigned long long)(n_ - seg_last); seg_last = n_; } while (0)
#else
#define PDDP_DW_SEGDECL
#ifdef PDDP_QP_MARKS  // the time marks alone: nothing inside the phases
__device__ long long g_defer_marks[2][4];
// [phases of role Q that left the lean BoxQP, those that ran the loop]
__device__ unsigned long long g_defer_odd[2];
#define PDDP_DW_COUNT(I) do { if (lane == 0) atomicAdd(&g_defer_odd[I], 1ull); } while (0)
#define PDDP_DW_MARK(W, I) do { if (blockIdx.x == 0 && lane == 0) g_defer_marks[W][I] = clock64(); } while (0)
#else
#define PDDP_DW_MARK(W, I)
#endif
#define PDDP_DW_STAMP(I)
#ifndef PDDP_DW_COUNT
#define PDDP_DW_COUNT(I)
#endif
#define PDDP_DW_DECL
#define PDDP_DW_BARRIER() n4::lds_publish_barrier()
#define PDDP_DW_END(ROLE)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc_i[lane q] += a[lane i] * b[lane q] within every quad: one
// v_mfma_f32_4x4x1_16b_f32 (sixteen 4x4 outer products, block = quad);
// double: four FMAs on quad broadcasts.
template <typename T>
struct Acc4 {
  T v0, v1, v2, v3;
};
PDDP_DEV void opa(Acc4<float>& c, float a, float b) {
  f32x4 x = {c.v0, c.v1, c.v2, c.v3};
  x = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, x, 0, 0, 0);
  c.v0 = x[0]; c.v1 = x[1]; c.v2 = x[2]; c.v3 = x[3];
}
PDDP_DEV void opa(Acc4<double>& c, double a, double b) {
  c.v0 = fma_(qb<0>(a), b, c.v0);
  c.v1 = fma_(qb<1>(a), b, c.v1);
  c.v2 = fma_(qb<2>(a), b, c.v2);
  c.v3 = fma_(qb<3>(a), b, c.v3);
}

// sum_k dpp_k(v) f_k + a   (v's element k sits in lane k of the quad)
PDDP_DEV float bdot4(float a, float v, float f0, float f1, float f2, float f3) {
  n4q::dpp_dot4(a, v, f0, f1, f2, f3);
  return a;
}
PDDP_DEV double bdot4(double a, double v, double f0, double f1, double f2,
                      double f3) {
  a = fma_(qb<0>(v), f0, a);
  a = fma_(qb<1>(v), f1, a);
  a = fma_(qb<2>(v), f2, a);
  return fma_(qb<3>(v), f3, a);
}


// ---- flags carried in the SIGN BIT of a 32-bit word (role Q's lean BoxQP).
// A compare that goes through an SGPR pair (v_cmp -> v_cndmask) costs a
// dependent chain ~40 cycles per trip; a subtraction leaves
// the same predicate in the sign bit of a VGPR, where v_and / v_or / v_bfi
// combine it at 4 cycles each.  The two instructions the optimiser would turn
// back into compare + select are issued by hand.
PDDP_DEV int sgn(float x) { return __float_as_int(x); }
PDDP_DEV int splat(int w) {  // 0 / -1 from the sign bit
  int r;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(w));
  return r;
}
PDDP_DEV float bsel(int mask, float a, float b) {  // mask ? a : b, bitwise
  float r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}
PDDP_DEV int bseli(int mask, int a, int b) {
  int r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}

// The closed form of the scalar BoxQP (riccati_n4.hpp QpClosed: the paths the
// reference's loop, utils/constraint.py:150-266, takes on all but ~0.01 % of
// the steps) without a single compare on its value chain.  Same case analysis
// as QpClosed::solve; what differs is rounding only: the objective is
// evaluated as v (Q/2 v + c) (three instructions less per evaluation; it feeds
// the convergence and Armijo tests, which are decided far from their
// thresholds wherever the closed form applies), the tolerance products are
// fused into the subtraction whose sign is tested, Q is taken as positive
// definite and finite (the caller routes every other value - and every `slow`
// case - through QpClosed and the reference's loop), and x0 as finite (it is
// the previous step's minimiser, inside finite bounds).
struct QpLean {
  float x, inv;
  int free_w, slow_w;  // flags in the sign bit
  PDDP_DEV void solve(float x0, float Q, float c, float lo, float hi) {
    const float d_lo = lo - x0, d_hi = x0 - hi;  // sign: x0 > lo, x0 < hi
    const float xs = __builtin_amdgcn_fmed3f(x0, lo, hi);
    const float hQ = 0.5f * Q;
    inv = __builtin_amdgcn_rcpf(Q);
    // ---- iteration 0                                        (:191-239)
    const float g0 = fma_(Q, xs, c);
    // clamped: (x == lo & g > 0) | (x == hi & g < 0), with x == lo <=> x0 <= lo
    // <=> sign(lo - x0) clear: one three-input bit operation on the sign bits
    // of (x0 - hi, lo - x0, g).  (g > 0 is read as "sign clear": a gradient
    // that is exactly zero ON a bound counts as clamped - the reference frees
    // that coordinate; x is the same either way.)
    const int ncl0 = (sgn(d_hi) & ~sgn(d_lo) & ~sgn(g0)) | (~sgn(d_hi) & sgn(g0));
    const int small0 = sgn(__builtin_fabsf(g0) - 1e-8f);
    const int done0 = ncl0 | small0;
    const float newton = -(c * inv);
    const float s0 = newton - xs;
    const float xa = xs + s0;
    const float x1 = __builtin_amdgcn_fmed3f(xa, lo, hi);
    const float d1_lo = lo - xa, d1_hi = xa - hi;  // x1 == lo <=> xa <= lo
    const float f0 = xs * fma_(hQ, xs, c), f1 = x1 * fma_(hQ, x1, c);
    const float num = f1 - f0;
    // ---- iteration 1: exit tests, one more full step
    const int conv = sgn(fma_(-1e-8f, __builtin_fabsf(f0), -num));
    const float g1 = fma_(Q, x1, c);
    const int ncl1 = (sgn(d1_hi) & ~sgn(d1_lo) & ~sgn(g1)) | (~sgn(d1_hi) & sgn(g1));
    const int small1 = sgn(__builtin_fabsf(g1) - 1e-8f);
    const int stop1 = conv | ncl1 | small1;
    const float x2 = __builtin_amdgcn_fmed3f(x1 + (newton - x1), lo, hi);
    x = bsel(splat(done0), xs, bsel(splat(stop1), x1, x2));
    // free = (done0 & !ncl0) | (!done0 & (conv | !ncl1)), done0 = ncl0 | small0
    free_w = ~ncl0 & (small0 | conv | ~ncl1);
    // ---- does the reference's loop leave these paths?  (QpClosed: pass0,
    // guard, live1 & on_bound1)
    const float sdotg = s0 * g0;
    const int npass = sgn(fma_(0.1f, sdotg, -num));  // !(num <= 0.1 sdotg)
    const int pass0 = sgn(sdotg) & ~npass;
    const int onb1 = ~(sgn(d1_lo) & sgn(d1_hi));
    const float lhs = __builtin_fabsf(x1 - xs) * __builtin_fabsf(sdotg);
    const float rhs = (3.0f * __builtin_fabsf(num)) * __builtin_fabsf(s0);
    const int guard = onb1 & sgn(sdotg) & sgn(num) & ~sgn(rhs - lhs);
    slow_w = ~done0 & (~(pass0 | guard) | (~stop1 & onb1));
  }
};

// What the sweep needs to evaluate its records itself (generator wavefronts,
// NP > 0 below): the nominal trajectory instead of `a.rec`, and where the
// stage costs and their sum go.
template <typename T>
struct GenArgs {
  const T* Z;      // [B][N + 1][4]
  const T* U;      // [B][N]  (un-clamped nominal actions)
  T* L;            // [B][N + 1] stage / terminal cost of the nominal
  T* J_opt;        // [B]: sum of L in t order, where `fresh` is set
  uint8_t* fresh;  // [B] nullable: "the nominal changed"; cleared
};

// Role P streams the records from `a.rec` by LDS-DMA (above).  (Round 3's
// form with a generator wavefront instead of role P - the sweep from the
// nominal - is superseded by riccati_n4_elem.hpp and gone; GenArgs stays: it is
// that kernel's argument block.)
template <typename T, bool FAST, int R>
PDDP_DEV void defer_body(const RiccatiArgs<T>& a) {
  using G = n4q::QuadGeom<T>;
  constexpr int kThreads = 4 * kWave;
  constexpr int NI = G::NI, RPI = G::RPI, CH = G::CH, CB = G::CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // exchange buffers, two parities each (written in phase p, read in p + 1)
  __shared__ __attribute__((aligned(16))) T xq[2][kTraj][4];   // Q: k, s, c, w
  __shared__ __attribute__((aligned(16))) T xin[2][kTraj][4];  // M: A00 | Y: G0, g2, B00
  __shared__ __attribute__((aligned(16))) T xz[2][kTraj][4];   // M: Quz0[q]
  __shared__ __attribute__((aligned(16))) T xy[2][kTraj][4];   // Y: carried y[q]
  __shared__ T ls_tail[n4::kLsSteps];
  T* ring = reinterpret_cast<T*>(smem_raw);

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  for (int i = threadIdx.x; i < n4::kLsSteps; i += kThreads)
    ls_tail[i] = (T)n4::kLs.v[i];
  for (int i = threadIdx.x; i < 2 * kTraj * 4; i += kThreads) {
    (&xq[0][0][0])[i] = T(0);
    (&xin[0][0][0])[i] = T(0);
    (&xz[0][0][0])[i] = T(0);
    (&xy[0][0][0])[i] = T(0);
  }

  const int q = lane & 3, tr = lane >> 2;
  const int N = a.N;
  const int b0 = blockIdx.x * kTraj;
  const int b = b0 + tr;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  // (identical in the four waves: they own the same sixteen trajectories)
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return;
  const int rbase = (tr / RPI) * G::GS + (tr % RPI) * kRec;
  const int oq = rbase + q, or4 = rbase + 4 * q;
  const int P = N + 2;  // phases: Q solves step tq = N + 1 - p in phase p
  PDDP_DW_DECL
  if (role == 0) PDDP_DW_MARK(0, 0);

  if (role == 3) {
    // =================================================================== P
    const char* rec_w = reinterpret_cast<const char*>(
        a.rec + (size_t)b0 * (size_t)(N + 1) * kRec);
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
    // record tau lives in slot (N - 1 - tau) % R
#pragma unroll
    for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
    n4::wait_vmcnt<0>();
    __syncthreads();  // ring filled, exchange buffers zeroed
    int p = 0;
    while (p < P) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (p >= P) break;
        // phase p reads record N - p (Q: U of the next step) and gathers
        // record N - 2 - p (M, Y: next phase's operands): the slot of record
        // N + 1 - p is dead from phase p on
        if (p >= 2) dma((s + R - 2) % R, N + 1 - p - R);
        // record N - 3 - p (gathered during phase p + 1) was requested in
        // phase p + 4 - R: at most R - 4 younger groups may be outstanding
        n4::wait_vmcnt<(R - 4) * NI>();
        PDDP_DW_BARRIER();
        ++p;
      }
    }
    n4::wait_vmcnt<0>();
    PDDP_DW_END(3);
    return;
  }

  if (role == 1) {
    // =================================================================== Q
    const T reg = (T)a.reg[bc];
    const T umin = a.u_min[0], umax = a.u_max[0];
    // (k, c, w) of step tq + 1; the coefficients of step tq
    T kprev = T(0), c1 = T(0), w1 = T(0);
    T A0p = T(0), B0p = T(0), g1 = T(0), g1sq = T(0);
    T lo_b = T(0), hi_b = T(0);  // bounds of step tq's BoxQP: u_min/max - U
    int status = PDDP_BWD_OK;
    __syncthreads();
    int p = 0;
    // The exact path of one step: QpClosed, the reference's loop behind it.
    // Lanes outside `take` keep what they hold.
    auto exact = [&](bool take, bool alive, T Quu, T Qu, T qp_Q, T& kt, T& sK,
                     T& c, T& w) {
      int st = PDDP_BWD_OK;
      if (!is_finite(Quu)) st = PDDP_BWD_NAN;       // eig raises (ilqr.py:631)
      n4::QpClosed<T, FAST> qc;
      qc.solve(kprev, qp_Q, Qu, lo_b, hi_b);
      T kx = qc.x;
      bool Kzero = !qc.free_, fail = qc.fail;
      const bool slow = qc.slow && alive && take;
      if (__builtin_expect(__any(slow), 0)) {
        PDDP_DW_COUNT(1);
        // rare: the reference's loop, one slow trajectory at a time on the
        // whole wavefront
        unsigned long long todo = __ballot(slow && q == 0);
        while (todo != 0) {
          const int src = __builtin_ctzll(todo);
          todo &= todo - 1;
          const n4::SlowQpOut<T> o = n4q::boxqp1_wave<T, FAST>(
              __shfl(kprev, src), __shfl(qp_Q, src), __shfl(Qu, src),
              __shfl(lo_b, src), __shfl(hi_b, src), ls_tail, lane);
          const bool mine = (lane >> 2) == (src >> 2);
          kx = mine ? o.x : kx;
          Kzero = mine ? ((o.result_free & 1) == 0) : Kzero;
          fail = mine ? (o.result_free < 2) : fail;
        }
      }
      // K = -s Quz: 1 / Q through v_rcp (FAST) or an IEEE division
      T sx;
      if constexpr (FAST) sx = qc.inv;
      else sx = T(1) / qp_Q;
      sx = Kzero ? T(0) : sx;
      const int stt = st != PDDP_BWD_OK ? st : (fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
      T cx, wx;
      n4q::rank_one_coeffs(kx, sx, Quu, Qu, cx, wx);
      kt = take ? kx : kt; sK = take ? sx : sK;
      c = take ? cx : c; w = take ? wx : w;
      status = (take & alive & (stt != PDDP_BWD_OK)) ? stt : status;
    };
    auto phase = [&](const int s) {
      const int tq = N + 1 - p;
      // what M and Y published last phase: the coefficients of step tq - 1;
      // U of step tq - 1 (record N - p: slot (p - 1) % R)
      const T* pi = &xin[(p + 1) & 1][tr][0];
      const T A00 = pi[0], G0 = pi[1], g2 = pi[2], B00 = pi[3];
      const T Unext = ring[((s + R - 1) % R) * G::SLOT + rbase + 46];
      T kt = T(0), sK = T(0), c = T(0), w = T(0);
      T* pq = &xq[p & 1][tr][0];
      if (tq <= N - 1) {
        PDDP_DW_STAMP(0);
        const bool alive = counted & (status == PDDP_BWD_OK);
        const T Quu = fma_(c1, g1sq, A0p);
        const T Qu = fma_(w1, g1, B0p);
        if constexpr (FAST && sizeof(T) == 4) {
          // e = Quu < 0 ? 1e-12 : Quu (ilqr.py:633), + reg (:634)
          const T qp_Q = bsel(splat(sgn(Quu)), 1e-12f, Quu) + reg;
          QpLean ql;
          ql.solve(kprev, qp_Q, Qu, lo_b, hi_b);
          kt = ql.x;
          sK = __int_as_float(splat(ql.free_w) & __float_as_int(ql.inv));
          n4q::rank_one_coeffs(kt, sK, Quu, Qu, c, w);
          // every lane of the quad holds the same four words
          *reinterpret_cast<f32x4*>(pq) = f32x4{kt, sK, c, w};
          PDDP_DW_STAMP(1);
          // off the chain (role M and Y read after the barrier): anything
          // the lean form does not cover goes through the exact path
          const bool odd = !is_finite(Quu) |
                           !__builtin_amdgcn_classf(qp_Q, 0x180) |
                           (ql.slow_w < 0);
          if (__builtin_expect(__any(odd & alive), 0)) {
            PDDP_DW_COUNT(0);
            exact(odd & alive, alive, Quu, Qu, qp_Q, kt, sK, c, w);
            *reinterpret_cast<f32x4*>(pq) = f32x4{kt, sK, c, w};
          }
        } else {
          const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
          const T qp_Q = e + reg;                     // ilqr.py:634
          exact(true, alive, Quu, Qu, qp_Q, kt, sK, c, w);
          if (q == 0) { pq[0] = kt; pq[1] = sK; pq[2] = c; pq[3] = w; }
          PDDP_DW_STAMP(1);
        }
      } else if (q == 0) {
        pq[0] = T(0); pq[1] = T(0); pq[2] = T(0); pq[3] = T(0);
      }
      // off the chain: the coefficients of step tq - 1 given (c, w) of step
      // tq + 1;  g_{tq,tq-1} = G0 + c_{tq+1} g_{tq+1,tq} g_{tq+1,tq-1}
      const T g1n = fma_(c1, mul_nc(g1, g2), G0);
      A0p = fma_(c1, mul_nc(g2, g2), A00);
      B0p = fma_(w1, g2, B00);
      g1 = g1n;
      g1sq = mul_nc(g1n, g1n);
      kprev = kt; c1 = c; w1 = w;
      lo_b = umin - Unext;
      hi_b = umax - Unext;
      PDDP_DW_STAMP(2);
      PDDP_DW_BARRIER();
    };
    while (p < P) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (p >= P) break;
        phase(s);
        ++p;
      }
    }
    if (counted && q == 0) a.status[bc] = status;
    PDDP_DW_END(1);
    return;
  }

  if (role == 2) {
    // =================================================================== Y
    // vector tq + 1 (finalised last phase): y1, carried to tq (y1c) and to
    // tq - 1 (y1cc), its dot products g1a = f_tq . y1, g1b = f_{tq-1} . y1c;
    // yp = y'_tq (everything of y_tq but the c_{tq+1} term); r0n = r0_{tq-1}
    T y1 = T(0), y1c = T(0), y1cc = T(0), g1a = T(0), g1b = T(0), yp = T(0);
    T r0n = T(0);
    {
      const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
      r0n = term[40 + q];  // r0_N = L_z[N]
    }
    char* gains_w =
        reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
    struct Ops {  // record words of one step, this lane's column
      T F0, F1, F2, F3, fq, Lz, Lu;
    };
    auto gather = [&](int slot) {
      const T* rc = ring + slot * G::SLOT;
      Ops o;
      o.F0 = rc[oq]; o.F1 = rc[oq + 4]; o.F2 = rc[oq + 8]; o.F3 = rc[oq + 12];
      o.fq = rc[oq + 32];
      o.Lz = rc[oq + 40];
      o.Lu = rc[rbase + 45];
      return o;
    };
    __syncthreads();
    int p = 0;
    // on entry to phase p: o_cur = record t = N - 1 - p (the package step),
    // o_prev = record tq - 1 = N - p (in phase 0 it does not exist: okA false)
    Ops o_cur = gather(0), o_prev = o_cur;
    auto phase = [&](const int s) {
      const int tq = N + 1 - p, t = tq - 2;
      const T* pq = &xq[(p + 1) & 1][tr][0];  // step tq + 1
      const T kq = pq[0], sq = pq[1], cq = pq[2], wq = pq[3];
      const T Quz0 = xz[(p + 1) & 1][tr][q];  // Quz0_{tq-1}[q] (M, last phase)
      asm volatile("" ::: "memory");  // (nothing else queues before these)
      const Ops oa = o_prev, ob = o_cur;
      const bool okA = (tq <= N - 1) & (tq >= 1);  // records tq, tq - 1 exist
      const bool okB = (tq <= N - 1) & (t >= 0);   // ... and t
      const bool okT = (t >= 0);
      PDDP_DW_STAMP(0);
      // (i) y_tq = y'_tq + (c_{tq+1} g_{tq+1,tq}) y_{tq+1} carried to tq
      const T y = fma_(mul_nc(cq, g1a), y1c, yp);
      // (ii) carry it to tq - 1 and tq - 2; dot products with f
      T yc = bdot4(T(0), y, oa.F0, oa.F1, oa.F2, oa.F3);
      T ga = quad_sum(mul_nc(oa.fq, y));
      yc = okA ? yc : T(0);
      ga = okA ? ga : T(0);
      T ycc = bdot4(T(0), yc, ob.F0, ob.F1, ob.F2, ob.F3);
      T gb = quad_sum(mul_nc(ob.fq, yc));
      ycc = okB ? ycc : T(0);
      gb = okB ? gb : T(0);
      xy[p & 1][tr][q] = ycc;  // y_tq at time t: taken into S0_t next phase
      PDDP_DW_STAMP(1);
      // (iii) y'_{tq-1} = Quz0_{tq-1} + (c_{tq+1} g_{tq+1,tq-1}) y_{tq+1} at tq - 1
      const T ypn = fma_(mul_nc(cq, g1b), y1cc, Quz0);
      T G0 = quad_sum(mul_nc(ob.fq, ypn));
      G0 = okT ? G0 : T(0);
      // (iv) r_{tq-1} = r0_{tq-1} + w_{tq+1} y_{tq+1} at tq - 1; step t's part
      const T r = fma_(wq, y1cc, r0n);
      T B00 = ob.Lu + quad_sum(mul_nc(ob.fq, r));
      B00 = okT ? B00 : T(0);
      // (v) publish
      if (q == 0) {
        T* po = &xin[p & 1][tr][0];
        po[1] = G0; po[2] = gb; po[3] = B00;
      }
      asm volatile("" ::: "memory");  // (published; the rest is off the path)
      PDDP_DW_STAMP(2);
      const T r0 = bdot4(ob.Lz, r, ob.F0, ob.F1, ob.F2, ob.F3);
      r0n = okT ? r0 : r0n;
      // (vi) gains of step tq + 1: its s arrived, its y was finalised last phase
      if (tq + 1 <= N - 1 && exists) {
        T* dst = reinterpret_cast<T*>(
            gains_w + (size_t)(((bc - b0) * N + (tq + 1)) * kGain + 1 + q) *
                          sizeof(T));
        *dst = -(sq * y1);
        if (q == 0) dst[-1] = kq;
      }
      y1 = y; y1c = yc; y1cc = ycc; g1a = ga; g1b = gb; yp = ypn;
      o_prev = ob;
      o_cur = gather((s + 1) % R);  // record t - 1, for the next phase
      PDDP_DW_STAMP(3);
      PDDP_DW_BARRIER();
    };
    while (p < P) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (p >= P) break;
        phase(s);
        ++p;
      }
    }
    // gains of step 0: its BoxQP result was published by the last barrier
    if (exists) {
      const T* pq = &xq[(P + 1) & 1][tr][0];
      T* dst = reinterpret_cast<T*>(
          gains_w + (size_t)(((bc - b0) * N + 0) * kGain + 1 + q) * sizeof(T));
      *dst = -(pq[1] * y1);
      if (q == 0) dst[-1] = pq[0];
    }
    PDDP_DW_END(2);
    return;
  }

  // ===================================================================== M
  struct Words {            // record t: the products' operands
    T F0, F1, F2, F3;       // F[k][q]
    T Lc0, Lc1, Lc2, Lc3;   // Lzz[i][q]
    T Lr0, Lr1, Lr2, Lr3;   // Lzz[q][i]
    T f0, f1, f2, f3, fq;   // F_u, F_u[q]
    T Luz, Luu;
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
    w.Luu = rc[rbase + 44];
    return w;
  };
  // S0_{t+1}, column q ("S0_N": the terminal value function, ilqr.py:581-583)
  Acc4<T> S0 = {T(0), T(0), T(0), T(0)};
  {
    const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
    S0.v0 = term[16 + 0 + q]; S0.v1 = term[16 + 4 + q];
    S0.v2 = term[16 + 8 + q]; S0.v3 = term[16 + 12 + q];
  }
  __syncthreads();
  int p = 0;
  PDDP_DW_MARK(0, 1);
  Words wn = gather(0);  // record N - 1
  auto phase = [&](const int s) {
    const int t = N - 1 - p;
    // c of step t + 3 and its vector carried to t + 1
    const T cq = xq[(p + 1) & 1][tr][2];
    const T yv = xy[(p + 1) & 1][tr][q];
    asm volatile("" ::: "memory");  // (nothing else queues before these)
    const Words w = wn;
    if (t >= 0) {
      PDDP_DW_STAMP(0);
      // W_{t+1} = S0_{t+1} + c y y^T
      Acc4<T> W = S0;
      opa(W, mul_nc(cq, yv), yv);
      // T = W F:  T_i[q] += W[i][k] F[k][q], W[i][k] = lane i's W_k
      Acc4<T> Tm = {T(0), T(0), T(0), T(0)};
      opa(Tm, W.v0, w.F0);
      opa(Tm, W.v1, w.F1);
      opa(Tm, W.v2, w.F2);
      opa(Tm, W.v3, w.F3);
      PDDP_DW_STAMP(1);
      // A00 = Luu + f^T W f (h = W f through the column: W symmetric to rounding)
      T h = mul_nc(W.v0, w.f0);
      h = fma_(W.v1, w.f1, h);
      h = fma_(W.v2, w.f2, h);
      h = fma_(W.v3, w.f3, h);
      const T A00 = w.Luu + quad_sum(mul_nc(w.fq, h));
      // Quz0[q] = Luz[q] + sum_k f[k] T[k][q]
      T Quz0 = fma_(w.f0, Tm.v0, w.Luz);
      Quz0 = fma_(w.f1, Tm.v1, Quz0);
      Quz0 = fma_(w.f2, Tm.v2, Quz0);
      Quz0 = fma_(w.f3, Tm.v3, Quz0);
      xz[p & 1][tr][q] = Quz0;
      if (q == 0) xin[p & 1][tr][0] = A00;
      asm volatile("" ::: "memory");  // (published; S0 is this wave's own)
      PDDP_DW_STAMP(2);
      // S0_t = 0.5 (Lzz + F^T T) + 0.5 (Lzz + F^T T)^T: column part C and its
      // mirror R accumulate the same products in the same order as the partner
      // lane's mirror / column, so that C + R is symmetric to the last bit
      const T h0 = T(0.5) * w.F0, h1 = T(0.5) * w.F1, h2 = T(0.5) * w.F2,
              h3 = T(0.5) * w.F3;
      Acc4<T> C = {T(0.5) * w.Lc0, T(0.5) * w.Lc1, T(0.5) * w.Lc2,
                   T(0.5) * w.Lc3};
      Acc4<T> Rm = {T(0.5) * w.Lr0, T(0.5) * w.Lr1, T(0.5) * w.Lr2,
                    T(0.5) * w.Lr3};
      opa(C, h0, Tm.v0);
      opa(Rm, Tm.v0, h0);
      opa(C, h1, Tm.v1);
      opa(Rm, Tm.v1, h1);
      opa(C, h2, Tm.v2);
      opa(Rm, Tm.v2, h2);
      opa(C, h3, Tm.v3);
      opa(Rm, Tm.v3, h3);
      S0.v0 = C.v0 + Rm.v0; S0.v1 = C.v1 + Rm.v1;
      S0.v2 = C.v2 + Rm.v2; S0.v3 = C.v3 + Rm.v3;
      PDDP_DW_STAMP(3);
    }
    wn = gather((s + 1) % R);  // record t - 1, for the next phase
    PDDP_DW_BARRIER();
  };
  while (p < P) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      if (p >= P) break;
      phase(s);
      ++p;
    }
  }
  PDDP_DW_MARK(0, 2);
  PDDP_DW_END(0);
}

template <typename T, bool FAST, int R>
__global__ __launch_bounds__(kThreads) void riccati_n4_defer_kernel(
    RiccatiArgs<T> a) {
  defer_body<T, FAST, R>(a);
}

}  // namespace n4d

// bounded eig-clamp branch only; 16 trajectories per workgroup of four waves
template <typename T>
static int launch_n4_defer(const RiccatiArgs<T>& a, hipStream_t st,
                           bool fast_math) {
  constexpr int R = 8;
  using G = n4q::QuadGeom<T>;
  if (a.u_min == nullptr || a.branch != PDDP_BRANCH_EIG)
    return PDDP_E_UNSUPPORTED;
  const size_t lds = (size_t)R * G::SLOT * sizeof(T);
  const dim3 grid((a.B + 15) / 16), block(n4d::kThreads);
#define PDDP_DF_GO(F)                                                         \
  do {                                                                        \
    auto kern = n4d::riccati_n4_defer_kernel<T, F, R>;                        \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, block, lds, st, a);                               \
  } while (0)
  if (fast_math && sizeof(T) == 4) PDDP_DF_GO(true);
  else PDDP_DF_GO(false);
#undef PDDP_DF_GO
  return launch_status();
}

}  // namespace pddp
