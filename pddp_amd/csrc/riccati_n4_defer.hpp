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

// NP = 0: role P streams the records from `a.rec` by LDS-DMA (above).
// NP > 0: nothing is read from `a.rec`.  NP GENERATOR wavefronts evaluate the
// records of the nominal (gen.Z, gen.U) with the sample problem's closed
// forms (models.hpp record_of - the code of derivs_kernel and of the fused
// line search's tail) straight into the ring: one lane per (trajectory, step),
// a wavefront = a BLOCK of four consecutive steps of the sixteen trajectories.
// A block takes a generator 4 NP phases (its evaluation is cut into segments
// by the phase barrier: the sync points of models.hpp, which also pin the
// segment's results so that the compiler keeps the arithmetic between the
// barriers it was written between); block j = records N-1-4j .. N-4-4j is
// first read in phase 4j - 1, written in phase 4j - 2, and its slots are dead
// from phase 4j + 5 - R on: R >= 4 NP + 6 ring slots.  Built: one generator,
// twelve slots (two generators and sixteen slots were measured too: the fifth
// wavefront costs every phase barrier more than the shorter segments save).
// Role Q evaluates the terminal record before the first phase; generator 0
// sums the stage costs (J_opt) in the phases after its last block.  The 79 MB
// of records per launch at B = 4096 are then neither written by the line
// search's tail nor read here.
template <typename T, bool FAST, int R, int NP, unsigned QM>
PDDP_DEV void defer_body(const RiccatiArgs<T>& a, const GenArgs<T>& gen,
                         const ProblemT<T>& prob) {
  using G = n4q::QuadGeom<T>;
  constexpr bool GEN = NP > 0;
  constexpr int kThreads = (3 + (GEN ? NP : 1)) * kWave;
  constexpr int NI = G::NI, RPI = G::RPI, CH = G::CH, CB = G::CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // GEN: terminal L_zz (16) and L_z (4) of every trajectory
  __shared__ __attribute__((aligned(16))) T term_sh[GEN ? kTraj : 1][20];
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
  // GEN: the first operands of this wavefront's role - the generator's first
  // block, role Q's terminal state - requested before anything waits on
  // memory (`active` below): one memory latency at the start, not two (in
  // the fit loop the nominal was written by the launch before: not in L2)
  T pre_z[4] = {T(0), T(0), T(0), T(0)}, pre_u = T(0);
  if constexpr (GEN) {
    const int gt = lane & 15;
    const int gb = b0 + gt < a.B ? b0 + gt : a.B - 1;
    int row = -1;
    if (role >= 3) {
      row = N - 1 - 4 * (role - 3) - (lane >> 4);
      row = row < 0 ? 0 : row;
    } else if (role == 1 && lane < kTraj) {
      row = N;
    }
    if (row >= 0) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(
          gen.Z + ((size_t)gb * (size_t)(N + 1) + row) * 4);
      pre_z[0] = v[0]; pre_z[1] = v[1]; pre_z[2] = v[2]; pre_z[3] = v[3];
      if (row < N) pre_u = gen.U[(size_t)gb * (size_t)N + row];
    }
  }
  // (identical in the four waves: they own the same sixteen trajectories)
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return;
  const int rbase = (tr / RPI) * G::GS + (tr % RPI) * kRec;
  const int oq = rbase + q, or4 = rbase + 4 * q;
  const int P = N + 2;  // phases: Q solves step tq = N + 1 - p in phase p
  PDDP_DW_DECL
  if (role == 0) PDDP_DW_MARK(0, 0);

  // GEN: stage costs of the sixteen trajectories, [kTraj][N + 1], behind the ring
  T* Lsh = ring + R * G::SLOT;
  if constexpr (GEN) {
    if (role >= 3) {
      // =========================================================== generator
      constexpr int MODEL = PDDP_MODEL_CARTPOLE;
      const int g = role - 3;
      const int gt = lane & 15, sb = lane >> 4;
      const bool gex = b0 + gt < a.B;
      const int gb = gex ? b0 + gt : a.B - 1;
      const int grbase = (gt / RPI) * G::GS + (gt % RPI) * kRec;
      const T* Zg = gen.Z + (size_t)gb * (size_t)(N + 1) * 4;
      const T* Ug = gen.U + (size_t)gb * (size_t)N;
      const int nblk = (N + 3) / 4;
      // (asked now: needed after the last block only)
      const bool sums = gex && (a.active == nullptr || a.active[gb] != 0) &&
                        (gen.fresh == nullptr || gen.fresh[gb] != 0);
      // A pass = one block = four phases: block j is evaluated in the phases
      // 4j - 5 .. 4j - 2 and first read in phase 4j - 1.  Block 0 is
      // evaluated before the first phase (its sync points do nothing), block
      // 1 starts before it too: the first sync point of its pass is the
      // workgroup's start barrier, the others are phase barriers - of which
      // every wavefront of the workgroup executes exactly P.  The three
      // kinds of pass are three instantiations (MODE): a state machine in
      // the barrier cost the pass a hundred scalar instructions.
      static_assert(NP == 1, "one generator (see above for two)");
      constexpr int SP = 4;
      // the sync points of record_of (models.hpp) this generator stops at:
      // SP - 1 (the pass ends with one more, after the LDS writes)
#ifndef PDDP_GEN_STOPS
#define PDDP_GEN_STOPS 0b00011010u
#endif
      constexpr unsigned kStops = PDDP_GEN_STOPS;
      static_assert(__builtin_popcount(kStops) == SP - 1, "");
      int p = 0;
      T* dst = ring;  // this pass's slot, this lane's record (set by pass)
      // operands of the block evaluated next, requested one pass ahead
      T zq[4], uq;
      auto request = [&](int j) {
        int tau = N - 1 - 4 * j - sb;
        tau = tau < 0 ? 0 : tau;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Zg + 4 * tau);
        zq[0] = v[0]; zq[1] = v[1]; zq[2] = v[2]; zq[3] = v[3];
        uq = Ug[tau];
      };
      // MODE 0: before the first phase; 1: the pass that crosses the start
      // barrier; 2: steady state
      auto pass = [&](auto mode, int j) {
        constexpr int MODE = decltype(mode)::value;
        int nsync = 0;
        auto barrier = [&]() {
          if constexpr (MODE == 2) {
            if (p < P) {  // (always: block j's pass ends with phase 4j - 2)
              PDDP_DW_STAMP(nsync & 3);  // (stats build: segment lengths)
              PDDP_DW_BARRIER();
              ++p;
            }
          } else if constexpr (MODE == 1) {
            if (nsync == 0) {
              __syncthreads();
            } else if (p < P) {
              PDDP_DW_BARRIER();
              ++p;
            }
          }
          ++nsync;
        };
        auto sync = [&](auto point, auto&... vals) {
          constexpr int K = decltype(point)::value;
          if constexpr (K == 3) {
            // F_z, F_u: into the ring as soon as they exist - left to itself
            // the compiler sinks the whole evaluation to the stores at the
            // end of the pass, into one segment as long as two phases
            const T f[] = {vals...};
            static_assert(sizeof...(vals) == 20, "");
#pragma unroll
            for (int k = 0; k < 16; k += 4)
              *reinterpret_cast<f32x4*>(dst + k) =
                  f32x4{f[k], f[k + 1], f[k + 2], f[k + 3]};
            *reinterpret_cast<f32x4*>(dst + 32) =
                f32x4{f[16], f[17], f[18], f[19]};
            asm volatile("" ::: "memory");
          }
          if constexpr ((kStops >> K) & 1u) {
            if constexpr (K != 3) (pin_value(vals), ...);
            barrier();
          }
        };
        const T z[4] = {zq[0], zq[1], zq[2], zq[3]};
        const T u = uq;
        if (j + 1 < nblk) request(j + 1);
        const int tau = N - 1 - 4 * j - sb;
        T w[kRec];
        // the block's slots: record tau lives in slot (N - 1 - tau) % R
        dst = ring + ((4 * j + sb) % R) * G::SLOT + grbase;
        const T l = record_of<T, MODEL, QM>(prob, z, &u, false, true, a.u_min,
                                            a.u_max, w, sync);
        // (words 0..15 and 32..35, F_z and F_u, went out at sync point 3;
        // 36..39, L_uz, are zero in every record: written the first time a
        // slot is used only)
#pragma unroll
        for (int k = 16; k < kRec; k += 4)
          if (k != 32 && k != 36)
            *reinterpret_cast<f32x4*>(dst + k) =
                f32x4{w[k], w[k + 1], w[k + 2], w[k + 3]};
        if (MODE != 2 || j < R / 4)
          *reinterpret_cast<f32x4*>(dst + 36) = f32x4{T(0), T(0), T(0), T(0)};
        if (tau >= 0) Lsh[gt * (N + 1) + tau] = l;  // (to `L` at the end)
        barrier();  // SP-th: the block is visible from the next phase on
      };
      PDDP_DW_MARK(1, 0);
      zq[0] = pre_z[0]; zq[1] = pre_z[1]; zq[2] = pre_z[2]; zq[3] = pre_z[3];
      uq = pre_u;  // (block 0: requested at the top of the kernel)
      pass(std::integral_constant<int, 0>{}, 0);
      PDDP_DW_MARK(1, 1);
      if (nblk > 1) pass(std::integral_constant<int, 1>{}, 1);
      else __syncthreads();
#pragma unroll 1
      for (int j = 2; j < nblk; ++j) pass(std::integral_constant<int, 2>{}, j);
      // every stage cost is in LDS now (the last block's barrier is behind
      // us): J_opt = L.sum() in t order, in the phases this wavefront would
      // otherwise idle through.  The LDS reads sixteen at a time - one read
      // per add would expose the LDS latency a hundred times.
      if (g == 0 && lane < kTraj && sums) {
        T Jacc = T(0);
        const T* Lt = Lsh + gt * (N + 1);
        int t = 0;
        for (; t + 16 <= N + 1; t += 16) {
          T v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = Lt[t + i];
#pragma unroll
          for (int i = 0; i < 16; ++i) Jacc += v[i];
        }
        for (; t <= N; ++t) Jacc += Lt[t];
        gen.J_opt[gb] = Jacc;
        if (gen.fresh != nullptr) gen.fresh[gb] = 0;
      }
      // the stage costs to `L`, whole rows (one word at a time from the
      // passes they were 4-byte writes into 400 000 different sectors, still
      // draining when the kernel was over: 2 us)
      {
        // (L is [B][N + 1]: the sixteen trajectories' rows are one run)
        const int cnt = ((a.B - b0 < kTraj) ? a.B - b0 : kTraj) * (N + 1);
        T* Lw = gen.L + (size_t)b0 * (size_t)(N + 1);
        for (int i0 = lane; i0 < cnt; i0 += 8 * kWave) {
          T v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kWave;
            v[k] = Lsh[i < cnt ? i : 0];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * kWave;
            if (i < cnt) Lw[i] = v[k];
          }
        }
      }
      while (p < P) {
        PDDP_DW_BARRIER();
        ++p;
      }
      PDDP_DW_MARK(1, 2);
      PDDP_DW_MARK(1, 3);
      PDDP_DW_END(3);
      return;
    }
  }
  if (!GEN && role == 3) {
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
    if constexpr (GEN) {
      // the terminal record (L_zz, L_z of z_N: what roles M and Y start
      // from), while the generators evaluate the first blocks
      if (lane < kTraj) {
        constexpr int MODEL = PDDP_MODEL_CARTPOLE;
        const T zN[4] = {pre_z[0], pre_z[1], pre_z[2], pre_z[3]};
        T lz[4], lzz[16], lu[1], luu[1];
        const T l = cost_derivs<T, MODEL>(prob, zN, nullptr,
                                          trig_of<T, MODEL>(zN), true, lz, lzz,
                                          lu, luu);
#pragma unroll
        for (int i = 0; i < 16; ++i) term_sh[lane][i] = lzz[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) term_sh[lane][16 + i] = lz[i];
        Lsh[lane * (N + 1) + N] = l;
      }
    }
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
      const int stt = fail ? (int)PDDP_BWD_BOXQP_FAILED : st;
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
    if constexpr (!GEN) {
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
    if constexpr (GEN) r0n = term_sh[tr][16 + q];
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
  if constexpr (!GEN) {
    const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
    S0.v0 = term[16 + 0 + q]; S0.v1 = term[16 + 4 + q];
    S0.v2 = term[16 + 8 + q]; S0.v3 = term[16 + 12 + q];
  }
  __syncthreads();
  if constexpr (GEN) {
    S0.v0 = term_sh[tr][0 + q]; S0.v1 = term_sh[tr][4 + q];
    S0.v2 = term_sh[tr][8 + q]; S0.v3 = term_sh[tr][12 + q];
  }
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
  defer_body<T, FAST, R, 0, 0u>(a, GenArgs<T>{}, ProblemT<T>{});
}

constexpr int kGenWaves = 1;   // generator wavefronts
constexpr int kGenRing = 12;   // ring slots = three blocks (>= 4 NP + 6)
template <unsigned QM>
__global__ __launch_bounds__((3 + kGenWaves) * kWave) void riccati_n4_gen_kernel(
    RiccatiArgs<float> a, GenArgs<float> gen, ProblemT<float> P) {
  defer_body<float, true, kGenRing, kGenWaves, QM>(a, gen, P);
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

// The same sweep from the nominal trajectory of a cartpole problem (f32,
// bounded, eig-clamp branch): records evaluated in the workgroup, none read.
static int launch_n4_gen(const pddp_problem& p, const RiccatiArgs<float>& a,
                         const n4d::GenArgs<float>& gen, hipStream_t st) {
  using G = n4q::QuadGeom<float>;
  if (p.model != PDDP_MODEL_CARTPOLE ||
      p.encoding != PDDP_ENC_IGNORE_UNCERTAINTY || a.u_min == nullptr ||
      a.u_max == nullptr || a.branch != PDDP_BRANCH_EIG || a.N < 8)
    return PDDP_E_UNSUPPORTED;
  const ProblemT<float> P = convert_problem<float>(p);
  const size_t lds = ((size_t)n4d::kGenRing * G::SLOT +
                      (size_t)n4d::kTraj * (a.N + 1)) * sizeof(float);
  if (lds > 150 * 1024) return PDDP_E_UNSUPPORTED;
  const dim3 grid((a.B + 15) / 16), block((3 + n4d::kGenWaves) * kWave);
  constexpr unsigned kSparse = 0b11001u;  // CartpoleCost: {x, sin, cos}
  const bool sparse =
      (live_mask(p.Q, ModelDims<PDDP_MODEL_CARTPOLE>::na) & ~kSparse) == 0;
#define PDDP_GEN_GO(QMV)                                                      \
  do {                                                                        \
    auto kern = n4d::riccati_n4_gen_kernel<QMV>;                              \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, block, lds, st, a, gen, P);                       \
  } while (0)
  if (sparse) PDDP_GEN_GO(kSparse);
  else PDDP_GEN_GO(kFullMask<PDDP_MODEL_CARTPOLE>);
#undef PDDP_GEN_GO
  return launch_status();
}

}  // namespace pddp
