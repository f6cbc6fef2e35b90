// riccati_n4.hpp - backward Riccati sweep specialised for n = 4, m = 1
// (cartpole; BASELINE.json configs[1]).
//
// Mapping: 16 lanes per trajectory, lane (i, j) = (row, column) of the 4x4
// matrices, FOUR trajectories per 64-lane wavefront, one wavefront per
// workgroup -> B/4 workgroups (1024 at B = 4096: one wave per SIMD on all 256
// CUs).  V_zz lives in ONE register per lane, V_z in one more.  The sweep is a
// chain of N dependent steps, so what bounds it is the per-step instruction
// count of one wave, not bandwidth: every product is a single FMA whose moving
// operand arrives through a DPP modifier (row rotation / quad permutation -
// no LDS traffic, no shuffles on the critical path), and every reduction is a
// two-step butterfly whose result is bit-identical in all participating lanes
// (a + b == b + a), which keeps replicated quantities consistent.
//
// Records stream HBM -> LDS by `global_load_lds` DMA (no VGPR round trip) into
// a ring of R slots per wave, R steps ahead of the dependent chain; each lane
// then gathers the ~17 record words it needs (skewed copies of F_z for the
// rotation-based products, row/column forms of the vectors) with ds_read.
//
// Restates pddp/controllers/ilqr.py:489-526 (Q) and :529-674 (backward) for
// m = 1, all four gain branches.  Summation order differs from the reference's
// left-to-right dot products (butterflies), i.e. results agree to rounding.
#pragma once

#include "gains.hpp"
#include "riccati_generic.hpp"

namespace pddp {

namespace n4 {

constexpr int kRec = 48;   // scalars per record (RecLayout(4,1).stride)
constexpr int kGain = 5;   // k, K[0..3]
constexpr int kRing = 8;   // record slots in flight per wave

template <int CTRL>
PDDP_DEV float dpp(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
PDDP_DEV int dppi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
PDDP_DEV double dpp(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL,
                                             0xf, 0xf, true);
  const int hi =
      __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) |
                              (long long)(unsigned int)lo);
}

// value of lane (i + D, j): DPP row_ror reads lane (l - n) mod 16 (probed on
// gfx950, tools/dpp_probe.hip), so "+4D lanes" is a rotation by 16 - 4D.
template <int D, typename T>
PDDP_DEV T from_row_plus(T v) {
  static_assert(D >= 1 && D <= 3, "");
  return dpp<0x120 + (16 - 4 * D)>(v);
}
// value of lane (i, j + D): quad_perm [D, D+1, D+2, D+3] (mod 4)
template <int D, typename T>
PDDP_DEV T from_col_plus(T v) {
  static_assert(D >= 1 && D <= 3, "");
  constexpr int P = ((0 + D) & 3) | (((1 + D) & 3) << 2) |
                    (((2 + D) & 3) << 4) | (((3 + D) & 3) << 6);
  return dpp<P>(v);
}
// A product that is never fused into a following add.  Replicated
// quantities must come out bit-identical in every lane that holds a copy, and
// x + y == y + x only holds when BOTH terms are already rounded: a contracted
// fma(a, b, y) on one lane against fma(c, d, x) on its partner would differ in
// the last bit and let the lanes of one trajectory disagree on branches.
// (The empty asm makes the rounded product opaque to the optimiser: LLVM
// otherwise distributes the lane permutation over the multiply,
// dpp(a * b) -> dpp(a) * dpp(b), and fuses THAT product into the add.)
PDDP_DEV float opaque(float x) {
  asm("" : "+v"(x));  // not volatile: free to move, never looked through
  return x;
}
PDDP_DEV double opaque(double x) {
  asm("" : "+v"(x));
  return x;
}
template <typename T>
PDDP_DEV T mul_nc(T a, T b) {
#pragma clang fp contract(off)
  return opaque(a * b);
}
PDDP_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PDDP_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// sum over the four rows (same column) of a * b; identical bits in all rows
template <typename T>
PDDP_DEV T dot_rows(T a, T b) {
#pragma clang fp contract(off)
  const T x = opaque(a * b);
  const T y = x + from_row_plus<2>(x);
  return y + from_row_plus<1>(y);
}
// sum over the four columns of a row (a quad) of a * b; identical in the quad
template <typename T>
PDDP_DEV T dot_cols(T a, T b) {
#pragma clang fp contract(off)
  const T x = opaque(a * b);
  const T y = x + from_col_plus<2>(x);
  return y + dpp<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(y);
}
// sum over all 16 lanes of a trajectory's group (identical in all of them)
template <typename T>
PDDP_DEV T group_sum(T x) {
#pragma clang fp contract(off)
  x = opaque(x);
  T y = x + from_row_plus<2>(x);
  y = y + from_row_plus<1>(y);
  y = y + from_col_plus<2>(y);
  return y + dpp<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(y);
}
template <typename T>
PDDP_DEV T sum_cols(T x) {
#pragma clang fp contract(off)
  const T y = x + from_col_plus<2>(x);
  return y + dpp<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(y);
}
PDDP_DEV float bperm(int addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}
PDDP_DEV double bperm(int addr, double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) |
                              (long long)(unsigned int)lo);
}

template <bool FAST>
PDDP_DEV float div_(float a, float b) {
  if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
  else return a / b;
}
template <bool FAST>
PDDP_DEV double div_(double a, double b) { return a / b; }
template <bool FAST>
PDDP_DEV float sqrtx(float a) {
  if constexpr (FAST) return __builtin_amdgcn_sqrtf(a);
  else return sqrt_(a);
}
template <bool FAST>
PDDP_DEV double sqrtx(double a) { return sqrt_(a); }

template <int N>
PDDP_DEV void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One full-wave 16-byte LDS DMA: lane l's 16 bytes at sbase + voff land at LDS
// byte `lds` + 16 l.  Hand-issued: with __builtin_amdgcn_global_load_lds the
// compiler's waitcnt pass drains vmcnt(0) before any LDS read it cannot prove
// disjoint from the DMA target (seen: once per ring revolution, +4 us per
// sweep); the record ring is ordered by the hand-counted wait_vmcnt<> below.
// m0 is written behind the compiler's back - nothing else in these kernels
// uses it.
PDDP_DEV void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               ::"v"(voff), "s"(sbase), "s"(lds)
               : "memory");
}
template <typename P>
PDDP_DEV uint32_t lds_addr(P* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
}

// Step sizes of the reference's backtracking line search: python
// `step *= step_dec` in double, rounded to T when it multiplies a tensor
// (utils/constraint.py:248-259).  kLsFail = first n with step < min_step.
constexpr int kLsSteps = 112;
struct LsTable {
  double v[kLsSteps];
  int n_fail;
  constexpr LsTable() : v(), n_fail(kLsSteps) {
    double s = 1.0;
    for (int n = 0; n < kLsSteps; ++n) {
      v[n] = s;
      if (s < 1e-22 && n < n_fail) n_fail = n;
      s *= 0.6;
    }
  }
};
__device__ constexpr LsTable kLs{};

// all LDS traffic of this wave done (its words are visible), then meet the
// other wavefronts of the workgroup.  No vmcnt wait: global traffic stays in
// flight.
PDDP_DEV void lds_publish_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
PDDP_DEV void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// clamp for the hot loop: v_med3 when the operands are known finite
template <bool FAST>
PDDP_DEV float clampq(float v, float lo, float hi) {
  if constexpr (FAST) return __builtin_amdgcn_fmed3f(v, lo, hi);
  else return clamp1(v, lo, hi);
}
template <bool FAST>
PDDP_DEV double clampq(double v, double lo, double hi) {
  return clamp1(v, lo, hi);
}

// group-wide integer min (identical in all 16 lanes)
PDDP_DEV int group_min(int m) {
  m = min(m, dppi<0x128>(m));
  m = min(m, dppi<0x12C>(m));
  m = min(m, dppi<(2 | (3 << 2) | (0 << 4) | (1 << 6))>(m));
  return min(m, dppi<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(m));
}

// the same two reductions over a group of GW lanes (16: a DPP row; 4: a quad)
template <int GW>
PDDP_DEV int gmin(int m) {
  if constexpr (GW == 16) {
    return group_min(m);
  } else {
    static_assert(GW == 4, "");
    m = min(m, dppi<(2 | (3 << 2) | (0 << 4) | (1 << 6))>(m));
    return min(m, dppi<(1 | (0 << 2) | (3 << 4) | (2 << 6))>(m));
  }
}
template <int GW, typename T>
PDDP_DEV T gsum(T x) {
  if constexpr (GW == 16) {
    return group_sum(x);
  } else {
    static_assert(GW == 4, "");
    return sum_cols(opaque(x));
  }
}

// Scalar BoxQP (m = 1): the reference's projected-Newton loop
// (utils/constraint.py:150-266) with its exit codes and its possibly stale
// `free` flag.  Every lane of the 16-lane group holds the same scalars.
//
// Two things keep it short on a wavefront that carries four trajectories:
//  * the back-tracking line search (:248-259), a sequential scan for the
//    first step 0.6^n that passes the Armijo test, evaluates 16 candidate n
//    per round, one per lane; the first passing n is a DPP integer-min
//    butterfly - the same n the sequential scan stops at;
//  * iterations are PREDICATED, not branched: a trajectory that has left the
//    loop keeps executing with its commits masked (`live`), so the whole
//    wavefront has one loop back-edge instead of six divergent exits.
// For one action dimension Q never changes, so the factorisation U = sqrt(Q)
// (:212-228) happens exactly once, and the Newton point -potrs(c, U) is
// loop-invariant.
// `lstep0` = T(0.6^l) for this lane (l = lane % 16); later rounds read LDS.
template <typename T, bool FAST, int GW = 16>
struct BoxQp1 {
  static constexpr T kMinGrad = T(1e-8), kTol = T(1e-8), kArmijo = T(0.1);
  T Q, c, lo, hi;           // the problem
  T x, f, old_f, U, newton;  // iterate, objective, factor, Newton point
  bool free_, not_pd;
  int result;
  // state handed from an iteration's head to its tail
  bool live, found;
  T search, sdotg, xc, fc;

  PDDP_DEV T obj(T v) const { return T(0.5) * ((v * Q) * v) + v * c; }

  // Branch-free first half of iteration `it`: exit tests, Newton direction and
  // the full-step candidate (the n = 0 the line search tests first, which
  // passes whenever the Newton point lies inside the box).
  PDDP_DEV void head(int it) {
    live = (result == 0);
    if (it > 0) {  // convergence on the objective decrease       (:191-193)
      const bool conv = (old_f - f) < kTol * abs_(old_f);
      result = (live && conv) ? 4 : result;
      live = live && !conv;
    }
    old_f = live ? f : old_f;
    const T g = Q * x + c;
    const bool ncl = ((x == lo) && (g > T(0))) || ((x == hi) && (g < T(0)));
    free_ = live ? !ncl : free_;                      // (:200-204)
    result = (live && ncl) ? 6 : result;              // all clamped (:207-209)
    live = live && !ncl;
    if (it == 0) {                                    // factorise (:212-228)
      result = (live && not_pd) ? -1 : result;
      live = live && !not_pd;
    }
    const bool gsmall = abs_(g) < kMinGrad;           // (:231-234)
    result = (live && gsmall) ? 5 : result;
    live = live && !gsmall;
    search = newton - x;                              // (:237-239)
    sdotg = search * g;
    xc = clampq<FAST>(x + search, lo, hi);
    fc = obj(xc);
    found = !live || !(div_<FAST>(fc - old_f, sdotg) < kArmijo);
  }

  // Second half: trajectories whose full step failed the Armijo test scan the
  // step sizes 0.6^n, 16 candidates per round (candidate n = nb + l in lane
  // l, restarting from n = 0 so that the reference's n is found); commit.
  PDDP_DEV void tail(T lstep0, const T* ls_tail, int l) {
    int nsel = 0;
    if (__any(!found)) {
      for (int nb = 0;; nb += GW) {
        const int n = nb + l;
        const T st = (nb == 0) ? lstep0 : ls_tail[n < kLsSteps ? n : kLsSteps - 1];
        const T xn = clampq<FAST>(x + st * search, lo, hi);
        const T fn = obj(xn);
        const bool ok = !(div_<FAST>(fn - old_f, st * sdotg) < kArmijo) ||
                        (n >= kLs.n_fail);
        const int m = gmin<GW>(ok ? n : 0x7fffffff);
        const bool hit = (m != 0x7fffffff) && !found;
        // broadcast the winner's (xn, fn): exactly one lane contributes, the
        // others add zeros, so the butterfly sums are exact
        const T xw = gsum<GW>((n == m) ? xn : T(0));
        const T fw = gsum<GW>((n == m) ? fn : T(0));
        xc = hit ? xw : xc;
        fc = hit ? fw : fc;
        nsel = hit ? m : nsel;
        found = found || (m != 0x7fffffff);
        if (!__any(!found)) break;
      }
    }
    x = live ? xc : x;
    f = live ? fc : f;
    result = (live && nsel >= kLs.n_fail) ? 2 : result;  // step < min_step
  }

  // Setup + head of iteration 0: straight-line code the scheduler can
  // interleave with the independent 4x4 products of the same step.
  PDDP_DEV void begin(T x0, T Q_, T c_, T lo_, T hi_) {
    Q = Q_; c = c_; lo = lo_; hi = hi_;
    x = clamp1(x0, lo, hi);
    x = ((x - x != T(0)) && (x == x)) ? T(0) : x;  // x[isinf(x)] = 0 (:179)
    f = obj(x);
    old_f = T(0);
    free_ = true;
    result = 0;
    U = sqrtx<FAST>(Q);
    newton = -div_<FAST>(div_<FAST>(c, U), U);  // -potrs(g_clamped, U)
    not_pd = !(Q > T(0)) || !is_finite(Q);
    head(0);
  }

  // Closed form of the loop for the paths a scalar QP almost always takes.
  // Call after begin().  Iteration 0's head has run exactly as the reference
  // runs it; what can follow for one action dimension is short:
  //  * the full Newton step passes the Armijo test (always, when the Newton
  //    point is inside the box: the ratio is 1/2), or it is cut short by a
  //    bound and every back-tracked candidate down to the first passing step
  //    size still clamps to that same bound (guard below), so whatever n the
  //    scan stops at, the new iterate is the n = 0 candidate (xc, fc);
  //  * iteration 1's exit tests run exactly (objective decrease -> 4 with the
  //    stale `free`, clamped at the bound -> 6, small gradient -> 5);
  //  * still live and strictly inside the box: the iterate is within rounding
  //    of the Newton point; one more full step lands on it and later
  //    iterations of the reference only move it by rounding errors before
  //    they exit with a result >= 1 and `free` set.
  // Returns true when none of this applies (cancellation in the Armijo
  // ratio, non-finite data, an iterate on a bound with the gradient pointing
  // inward); the caller then runs the loop itself: begin() + finish().
  PDDP_DEV bool closed_form() {
    const T ratio0 = div_<FAST>(fc - old_f, sdotg);
    const bool on_bound = (xc == lo) || (xc == hi);
    // all candidates x + 0.6^k search, k <= n, overshoot the bound by 2x:
    // 0.6^n > 6 ratio0 for the first n with ratio0 / 0.6^n >= 0.1
    const bool guard = on_bound && (ratio0 >= T(1e-18)) &&
                       (abs_(xc - x) <= (T(3) * ratio0) * abs_(search));
    bool slow = live && !(found || guard);
    x = live ? xc : x;
    f = live ? fc : f;
    head(1);
    const bool inside = (x > lo) && (x < hi) && (xc > lo) && (xc < hi);
    slow = slow || (live && !inside);
    x = live ? xc : x;
    result = live ? 5 : result;
    return slow;
  }

  PDDP_DEV int finish(T lstep0, const T* ls_tail, int lane) {
    const int l = lane & (GW - 1);
    tail(lstep0, ls_tail, l);
    for (int it = 1; it < 100; ++it) {
      if (!__any(result == 0)) break;
      head(it);
      tail(lstep0, ls_tail, l);
    }
    return result;
  }
};

template <typename T, bool FAST, int GW = 16>
PDDP_DEV int boxqp1(T x0, T Q, T c, T lo, T hi, T lstep0, const T* ls_tail,
                    int lane, T& x_out, T& U_out, bool& free_out) {
  BoxQp1<T, FAST, GW> qp;
  qp.begin(x0, Q, c, lo, hi);
  const int res = qp.finish(lstep0, ls_tail, lane);
  x_out = qp.x;
  U_out = qp.U;
  free_out = qp.free_;
  return res;
}

// The closed form of BoxQp1 (see BoxQp1::closed_form) as straight-line code
// for the sweep kernel, which needs the minimiser, the `free` flag, failure
// (result < 1: only "not positive definite" can occur for m = 1) and whether
// the loop has to run instead - not the result code itself.
// FAST: potrs through one reciprocal of Q instead of two divisions by
// sqrt(Q), and the Armijo ratio tests cross-multiplied instead of divided.
template <typename T, bool FAST>
struct QpClosed {
  T x, U, inv;  // minimiser; sqrt(Q) (IEEE) or 1 / Q (FAST)
  bool free_, fail, slow;
#ifdef PDDP_QP_STATS
  int dbg;  // why: bit 0 done0, 1 Armijo / guard failed, 2 live on a bound, ...
#endif

  // All flag logic below is written with the eager `&` / `|` on bools: with
  // `&&` / `||` the compiler builds exec-mask branches around the compares.
  // `slow` is conservative (it may ask for the loop where the closed form
  // would have been right) and cheap: anything that is not a descent step
  // with a comfortably passing Armijo test goes to the loop.
  PDDP_DEV void solve(T x0, T Q, T c, T lo, T hi) {
    constexpr T kMinGrad = T(1e-8), kTol = T(1e-8), kArmijo = T(0.1);
    auto obj = [&](T v) { return T(0.5) * ((v * Q) * v) + v * c; };
    T xs = clampq<FAST>(x0, lo, hi);
    xs = __builtin_isinf(xs) ? T(0) : xs;                    // (:179)
    const T f0 = obj(xs);
    // ---- iteration 0                                        (:191-239)
    const T g0 = Q * xs + c;
    const bool ncl0 = ((xs == lo) & (g0 > T(0))) | ((xs == hi) & (g0 < T(0)));
    const bool not_pd = !(Q > T(0)) | !is_finite(Q);
    const bool done0 = ncl0 | not_pd | (abs_(g0) < kMinGrad);
    fail = !ncl0 & not_pd;
    T newton;
    if constexpr (FAST) {
      inv = div_<true>(T(1), Q);
      U = T(0);
      newton = -(c * inv);
    } else {
      U = sqrt_(Q);
      inv = T(0);
      newton = -((c / U) / U);
    }
    const T s0 = newton - xs;
    const T sdotg = s0 * g0;
    const T x1 = clampq<FAST>(xs + s0, lo, hi);
    const T f1 = obj(x1);
    const T num = f1 - f0;
    // Armijo at the full step, (f1 - f0) / sdotg >= 0.1, cross-multiplied for
    // the descent case sdotg < 0 (anything else, NaN included: the loop)
    const bool pass0 = (sdotg < T(0)) & (num <= kArmijo * sdotg);
    // cut short by a bound: every back-tracked candidate down to the first
    // passing step size 0.6^n > 6 (num / sdotg) still overshoots the bound
    const bool on_bound1 = (x1 == lo) | (x1 == hi);
    const bool guard = on_bound1 & (sdotg < T(0)) & (num < T(0)) &
                       (abs_(x1 - xs) * abs_(sdotg) <=
                        (T(3) * abs_(num)) * abs_(s0));
    // ---- iteration 1: exit tests, one more full step
    const bool conv = (f0 - f1) < kTol * abs_(f0);
    const T g1 = Q * x1 + c;
    const bool ncl1 = ((x1 == lo) & (g1 > T(0))) | ((x1 == hi) & (g1 < T(0)));
    const bool live1 = !(conv | ncl1 | (abs_(g1) < kMinGrad));
    const T x2 = clampq<FAST>(x1 + (newton - x1), lo, hi);
    slow = !done0 & (!(pass0 | guard) | (live1 & on_bound1));
#ifdef PDDP_QP_STATS
    dbg = (done0 ? 1 : 0) | ((!done0 & !(pass0 | guard)) ? 2 : 0) |
          ((!done0 & live1 & on_bound1) ? 4 : 0) | (on_bound1 ? 8 : 0) |
          ((!done0 & !pass0 & guard) ? 16 : 0) | (!(sdotg < T(0)) ? 32 : 0);
#endif
    x = done0 ? xs : (live1 ? x2 : x1);
    free_ = (done0 & !ncl0) | (!done0 & (conv | !ncl1));
  }
};

#ifdef PDDP_QP_STATS
__device__ unsigned long long g_qp_stats[4];  // wave-steps: slow, total; traj-steps slow, total
#endif
// Out-of-line copy of the loop for the closed-form kernels: called on well
// under 0.1 % of the steps, so it must not cost the hot path registers,
// code size or scheduling freedom.
template <typename T>
struct SlowQpOut {
  T x, U;
  int result_free;  // result * 2 + free
};
template <typename T, bool FAST, int GW = 16>
__device__ __noinline__ SlowQpOut<T> boxqp1_outlined(T x0, T Q, T c, T lo,
                                                     T hi, T lstep0,
                                                     const T* ls_tail,
                                                     int lane) {
  SlowQpOut<T> o;
  bool fr;
  const int res = boxqp1<T, FAST, GW>(x0, Q, c, lo, hi, lstep0, ls_tail, lane,
                                      o.x, o.U, fr);
  o.result_free = res * 2 + (fr ? 1 : 0);
  return o;
}

// WPB wavefronts per workgroup, independent of each other after the one
// barrier that publishes the step-size table: WPB = 4 puts one wave on every
// SIMD of a CU by construction (one-wave workgroups left that to the
// dispatcher, whose placement depends on the previous kernel's shape).
template <typename T, bool CHOL, bool BOUNDED, bool FAST, int G,
          bool QPCF = false, int WPB = 1>
__global__ __launch_bounds__(kWave * WPB) void riccati_n4_kernel(
    RiccatiArgs<T> a) {
  // QPCF: BoxQP through BoxQp1::closed_form(), the loop only as fall-back
  // G = trajectories per wavefront (4: all lanes busy; 2: half the lanes idle
  // but twice the wavefronts and less BoxQP divergence per wavefront)
  // Record DMAs are FULL-wave 16-byte instructions: the four records of a
  // wave are 48 (f32) / 96 (f64) chunks; the lanes past them re-load an earlier
  // chunk into the slot's padding (a slot is NI KiB).  No lane-dependent
  // branch around a DMA: the compiler merges such divergent calls into one
  // whose LDS base is a per-lane value, i.e. wrong data (seen with f64).
  // (global_load_lds_dwordx3 is no way out: its 12 bytes land at lane * 16.)
  static_assert(G == 4, "four trajectories per wavefront");
  constexpr int CB = 16;                            // bytes per chunk
  constexpr int CH = kRec * (int)sizeof(T) / CB;    // chunks per record
  constexpr int NI = (4 * CH + kWave - 1) / kWave;  // DMA instructions / step
  constexpr int kSlot = NI * kWave * CB / (int)sizeof(T);  // scalars per slot
  // ring depth: 8 slots; 6 where four waves' rings of 2 KiB slots (f64) would
  // not fit the 64 KiB of static LDS
  constexpr int R = (WPB * kRing * kSlot * (int)sizeof(T) > 60 * 1024) ? 6 : kRing;
  __shared__ __attribute__((aligned(16))) T ring_all[WPB][R][kSlot];
  __shared__ T ls_tail[kLsSteps];  // T(0.6^n), read only past n = 31

  const int lane = threadIdx.x & (kWave - 1);
  const int wave =
      WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  T (*ring)[kSlot] = ring_all[wave];
  T lstep[1] = {T(0)};
  if constexpr (BOUNDED) {
    for (int q = threadIdx.x; q < kLsSteps; q += kWave * WPB)
      ls_tail[q] = (T)kLs.v[q];
    lstep[0] = (T)kLs.v[lane & 15];
    __syncthreads();
  }
  const int grp = lane >> 4, l = lane & 15, i = l >> 2, j = l & 3;
  const int N = a.N;
  const int b0 = (blockIdx.x * WPB + wave) * G;
  if (b0 >= a.B) return;  // a whole wave past the batch (WPB > 1)
  const int b = b0 + grp;
  const bool exists = (grp < G) && (b < a.B);
  const int bc = exists ? b : a.B - 1;
  // `counted` trajectories write a status at the end; a trajectory is alive
  // while it is counted and its status is still OK (one compare per step, no
  // loop-carried flag)
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return;

  const T reg = (T)a.reg[bc];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) {
    umin = a.u_min[0];
    umax = a.u_max[0];
  }

  // ---- DMA source addressing: chunk q of the wave's 4 records -> lane
  // (wave-uniform 64-bit base in SGPRs + a 32-bit lane offset: one VALU add
  // per step instead of 64-bit address arithmetic)
  const char* rec_w =
      reinterpret_cast<const char*>(a.rec + (size_t)b0 * (size_t)(N + 1) * kRec);
  uint32_t src_off[NI];
#pragma unroll
  for (int r = 0; r < NI; ++r) {
    int q = lane + kWave * r;
    q = q < 4 * CH ? q : q - 4 * CH;  // padding lanes: any valid chunk
    const int tg = q / CH, c = q - tg * CH;
    int tb = b0 + tg;
    tb = tb < a.B ? tb : a.B - 1;
    src_off[r] = (uint32_t)((tb - b0) * (N + 1) * kRec * (int)sizeof(T) + c * CB);
  }
  auto dma = [&](int slot, int t) {
    const int tt = t < 0 ? 0 : t;  // tail: harmless reload keeps vmcnt exact
    const uint32_t toff = (uint32_t)tt * (uint32_t)(kRec * sizeof(T));
#pragma unroll
    for (int r = 0; r < NI; ++r)
      lds_dma16(rec_w, src_off[r] + toff,
                __builtin_amdgcn_readfirstlane(lds_addr(&ring[slot][0])) +
                    r * kWave * CB);
  };

  // ---- LDS gather offsets (words inside this group's record)
  const int gb = grp * kRec;
  int oFs[4], oFq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    oFs[d] = gb + ((i + d) & 3) * 4 + i;  // F_z[(i+d)%4][i]
    oFq[d] = gb + ((j + d) & 3) * 4 + j;  // F_z[(j+d)%4][j]
  }
  const int oFt = gb + j * 4 + i;         // F_z[j][i]
  const int oLzz = gb + 16 + i * 4 + j;   // L_zz[i][j]
  const int oFur = gb + 32 + i, oFuc = gb + 32 + j;
  const int oLuzr = gb + 36 + i;
  const int oLzr = gb + 40 + i;
  const int oLuu = gb + 44, oLu = gb + 45, oU = gb + 46;
  const int tr_addr = ((lane & 48) | (j * 4 + i)) * 4;  // lane (j, i)

  // ---- terminal value function: V = L_zz[N], V_z = L_z[N] (column form)
  const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
  T V = term[16 + i * 4 + j];
  T Vzc = term[40 + j];

  // ---- prologue: fill the ring
#pragma unroll
  for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
  wait_vmcnt<0>();

  T kprev = T(0);
  int status = PDDP_BWD_OK;
  char* gains_w = reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
  // byte offset of this lane's k / K word of step t, walked down by one
  // gain record per step
  uint32_t gout_off = (uint32_t)(
      ((bc - b0) * N * kGain + (N - 1) * kGain + ((l < 4) ? 1 + l : 0)) *
      (int)sizeof(T));

  // The words a lane needs from one record, gathered from LDS one step ahead
  // of their use so that the ds_read latency overlaps the previous step.
  struct Words {
    T Fs0, Fs1, Fs2, Fs3, Fq0, Fq1, Fq2, Fq3, Ft, Lzz, fr, fc, Luzr, Lzr, Luu,
        Lu, Un;
  };
  auto gather = [&](int slot) {
    const T* rc = &ring[slot][0];
    Words w;
    w.Fs0 = rc[oFs[0]]; w.Fs1 = rc[oFs[1]]; w.Fs2 = rc[oFs[2]]; w.Fs3 = rc[oFs[3]];
    w.Fq0 = rc[oFq[0]]; w.Fq1 = rc[oFq[1]]; w.Fq2 = rc[oFq[2]]; w.Fq3 = rc[oFq[3]];
    w.Ft = rc[oFt]; w.Lzz = rc[oLzz];
    w.fr = rc[oFur]; w.fc = rc[oFuc];
    w.Luzr = rc[oLuzr]; w.Lzr = rc[oLzr];
    w.Luu = rc[oLuu]; w.Lu = rc[oLu];
    w.Un = BOUNDED ? rc[oU] : T(0);
    return w;
  };
  int t = N - 1;
  // one step of the sweep on the words `w` of record t, ring slot s
  auto step = [&](const Words& w, const int s) {
    {
      const T Fs0 = w.Fs0, Fs1 = w.Fs1, Fs2 = w.Fs2, Fs3 = w.Fs3;
      const T Fq0 = w.Fq0, Fq1 = w.Fq1, Fq2 = w.Fq2, Fq3 = w.Fq3;
      const T Ft = w.Ft, Lzz = w.Lzz, fr = w.fr, fc = w.fc;
      const T Luzr = w.Luzr, Lzr = w.Lzr, Luu = w.Luu, Lu = w.Lu, Un = w.Un;

      const bool alive = counted & (status == PDDP_BWD_OK);
      // ---- scalars that feed the gain computation first: (f^T V)[j]
      // (column form), Q_uu, Q_u (and their V + reg I twins)
      const T bTc = dot_rows(fr, V);
      const T Quu = Luu + dot_cols(bTc, fc);
      const T Qu = Lu + dot_cols(fc, Vzc);
      T Quug = Quu;
      T Vr = V;
      if constexpr (CHOL) {
        // second Q() with V + reg I                           (ilqr.py:590-592)
        Vr = (i == j) ? V + reg : V;
        const T bTrc = dot_rows(fr, Vr);
        Quug = Luu + dot_cols(bTrc, fc);
      }

      // ---- gains, part 1 (every lane of the group computes the same
      // scalars).  For the bounded branches this is the straight-line head of
      // the BoxQP; it is independent of the 4x4 products below, so the
      // scheduler interleaves the two dependency chains.
      T kt = T(0);
      T Uch = T(1);
      bool Kzero = false, by_inv = false;
      T inv = T(0);
      int st = PDDP_BWD_OK;
      // dead groups get a trivial QP so that their BoxQP loop exits at once
      const T Quu_s = alive ? Quu : T(1);
      const T Quug_s = alive ? Quug : T(1);
      const T Qu_s = alive ? Qu : T(0);
      const T x0_s = alive ? kprev : T(0);
      const T Un_s = alive ? Un : T(0);
      BoxQp1<T, FAST> qp;
      QpClosed<T, FAST> qc;
      T qp_Q = T(1);
      if constexpr (QPCF && BOUNDED) {
        // closed form: straight-line, no sanitising needed (no loop to leave)
        if constexpr (!CHOL) {
          if (!is_finite(Quu)) st = PDDP_BWD_NAN;  // eig raises (ilqr.py:631)
          const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
          qp_Q = e + reg;                             // ilqr.py:634, (E e) E^T
        } else {
          qp_Q = Quug;
        }
        qc.solve(kprev, qp_Q, Qu, umin - Un, umax - Un);
      } else if constexpr (!CHOL) {
        if (!is_finite(Quu_s)) st = PDDP_BWD_NAN;  // eig raises (ilqr.py:631)
        T e = (Quu_s < T(0)) ? T(1e-12) : Quu_s;   // ilqr.py:633
        e += reg;                                  // ilqr.py:634
        if constexpr (!BOUNDED) {
          inv = div_<FAST>(T(1), e) * T(1);        // (E / e) E^T
          kt = -(inv * Qu_s);
          by_inv = true;
          if (kt != kt) st = PDDP_BWD_NAN;
        } else {
          const T Qg = (T(1) * e) * T(1);          // (E * e) E^T
          qp.begin(x0_s, Qg, Qu_s, umin - Un_s, umax - Un_s);
        }
      } else {
        if constexpr (!BOUNDED) {
          if (!(Quug_s > T(0)) || !is_finite(Quug_s)) st = PDDP_BWD_NOT_PD;
          Uch = sqrtx<FAST>(Quug_s);
          kt = -div_<FAST>(div_<FAST>(Qu_s, Uch), Uch);
        } else {
          qp.begin(x0_s, Quug_s, Qu_s, umin - Un_s, umax - Un_s);
        }
      }

      // ---- the 4x4 products
      // A = F^T V : A[i][j] = sum_k F[k][i] V[k][j], k = (i + d) % 4
      T A = Fs0 * V;
      A += Fs1 * from_row_plus<1>(V);
      A += Fs2 * from_row_plus<2>(V);
      A += Fs3 * from_row_plus<3>(V);
      // Q_zz (raw) = L_zz + A F : sum_k A[i][k] F[k][j], k = (j + d) % 4
      T Qzz = Lzz + A * Fq0;
      Qzz += from_col_plus<1>(A) * Fq1;
      Qzz += from_col_plus<2>(A) * Fq2;
      Qzz += from_col_plus<3>(A) * Fq3;
      // Q_uz (row form) = L_uz + A f ; Q_z (row form)
      const T Quzr = Luzr + dot_cols(A, fc);
      const T Qzr = Lzr + dot_cols(Ft, Vzc);
      T Quzgr = Quzr;  // operand of the K solve
      if constexpr (CHOL) {
        T Ar = Fs0 * Vr;
        Ar += Fs1 * from_row_plus<1>(Vr);
        Ar += Fs2 * from_row_plus<2>(Vr);
        Ar += Fs3 * from_row_plus<3>(Vr);
        Quzgr = Luzr + dot_cols(Ar, fc);
      }

      // transposes (lane (i,j) <- lane (j,i)); latency hidden by the BoxQP
      const T QzzT = bperm(tr_addr, Qzz);
      const T Quzc = bperm(tr_addr, Quzr);
      const T Qzc = bperm(tr_addr, Qzr);
      T Quzgc = Quzc;
      if constexpr (CHOL) Quzgc = bperm(tr_addr, Quzgr);

      // ---- gains, part 2 (loop variants): line-search scan and further
      // BoxQP iterations
      bool fail = false;
      if constexpr (BOUNDED && QPCF) {
        kt = qc.x;
        Uch = qc.U;
        Kzero = !qc.free_;
        fail = qc.fail;
      } else if constexpr (BOUNDED) {
        const int res = qp.finish(lstep[0], ls_tail, lane);
        kt = qp.x;
        Uch = qp.U;
        Kzero = !qp.free_;
        fail = res < 1;
      }
      const T Qzzs = mul_nc(T(0.5), Qzz + QzzT);
      // the rest of the step given the minimiser: K, status, stores, value
      // update.  Instantiated twice in the closed-form kernels (after the
      // closed form and after the rare loop call) so that the two paths only
      // merge at the step boundary.
      auto tail = [&]() {
        int stt = st != PDDP_BWD_OK ? st : (fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
        // K in row and column form, same arithmetic on transposed copies
        T Kr, Kc;
        if (by_inv) {
          Kr = -(inv * Quzgr);
          Kc = -(inv * Quzgc);
        } else if constexpr (BOUNDED && QPCF && FAST) {
          Kr = Kzero ? T(0) : -(Quzgr * qc.inv);
          Kc = Kzero ? T(0) : -(Quzgc * qc.inv);
        } else {
          Kr = Kzero ? T(0) : -div_<FAST>(div_<FAST>(Quzgr, Uch), Uch);
          Kc = Kzero ? T(0) : -div_<FAST>(div_<FAST>(Quzgc, Uch), Uch);
        }
        if constexpr (!CHOL && !BOUNDED) {
          // NaN anywhere in K raises too (ilqr.py:639-640)
          const bool nanK = (Kc != Kc);
          const bool any4 = sum_cols(nanK ? T(1) : T(0)) != T(0);
          if (any4 && stt == PDDP_BWD_OK) stt = PDDP_BWD_NAN;
        }
        status = (alive & (stt != PDDP_BWD_OK)) ? stt : status;

        // ---- store k, K (lanes l < 5 of each group; dead groups write junk)
        {
          const T val = (l < 4) ? Kc : kt;
          T* dst = reinterpret_cast<T*>(gains_w + gout_off);
          if (exists && l < 5) *dst = val;
        }
        kprev = kt;

        // ---- value update with the un-regularised Q_uu, Q_uz
        // (ilqr.py:664-672); on the diagonal QzzT == Qzz, 0.5 (q + q) == q
        {
          T v = Qzc + Kc * Qu;
          v += (Kc * Quu) * kt;
          v += Quzc * kt;
          Vzc = v;
        }
        {
          // lane (i,j) forms V'[i][j] AND V'[j][i] from the row / column
          // copies with mirrored operation trees, so that its partner lane
          // (j,i) computes bit-identical values and 0.5 (a + b) is exactly
          // symmetric
          const T va = fma_(mul_nc(Kr, Quu), Kc, Qzzs) +
                       fma_(Kr, Quzc, mul_nc(Quzr, Kc));
          const T vb = fma_(mul_nc(Kc, Quu), Kr, Qzzs) +
                       fma_(Kc, Quzr, mul_nc(Quzc, Kr));
          V = T(0.5) * (va + vb);
        }

      };
      if constexpr (BOUNDED && QPCF) {
#ifdef PDDP_QP_STATS
        if (lane == 0) {
          atomicAdd(&g_qp_stats[1], 1ull);
          if (__any(qc.slow && alive)) atomicAdd(&g_qp_stats[0], 1ull);
        }
#endif
        if (__any(qc.slow && alive)) {  // rare: the reference's loop as written
          // dead groups get a trivial QP so that their loop exits at once
          const SlowQpOut<T> o = boxqp1_outlined<T, FAST>(
              alive ? kprev : T(0), alive ? qp_Q : T(1), alive ? Qu : T(0),
              umin - (alive ? Un : T(0)), umax - (alive ? Un : T(0)), lstep[0],
              ls_tail, lane);
          kt = o.x;
          Uch = o.U;
          Kzero = (o.result_free & 1) == 0;
          fail = o.result_free < 2;
          tail();
        } else {
          tail();
        }
      } else {
        tail();
      }

      // refill this slot with the record R steps further down the sweep
      gout_off -= (uint32_t)(kGain * sizeof(T));
      dma(s, t - R);
    }
  };

  // Two word sets alternate (one in use, one being gathered for the next
  // step), so no register copies are needed between steps.  DMA(t-1) has
  // landed once at most (R-2) younger {store, DMA} pairs are outstanding.
  Words wa = gather(0), wb = wa;
  while (t >= 0) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      if (t < 0) break;
      wait_vmcnt<(R - 2) * (1 + NI)>();
      wb = gather((s + 1) % R);  // (at t == 0 this reads a stale slot, unused)
      step(wa, s);
      --t;
      if (t < 0) break;
      wait_vmcnt<(R - 2) * (1 + NI)>();
      wa = gather((s + 2) % R);
      step(wb, s + 1);
      --t;
    }
  }
  wait_vmcnt<0>();
  if (counted && l == 0) a.status[bc] = status;
}

// Stand-alone batched scalar BoxQP (utils/constraint.py:150-266 for m = 1):
// one 16-lane group per problem, the code path of the sweep kernel.
template <typename T, bool FAST>
__global__ __launch_bounds__(kWave) void boxqp1_kernel(
    int count, const T* x0, const T* Q, const T* c, const T* lo, const T* hi,
    T* x, int32_t* result, uint8_t* free_mask) {
  __shared__ T ls_tail[kLsSteps];
  const int lane = threadIdx.x;
  for (int q = lane; q < kLsSteps; q += kWave) ls_tail[q] = (T)kLs.v[q];
  T lstep[1] = {(T)kLs.v[lane & 15]};
  __syncthreads();
  const int p = blockIdx.x * 4 + (lane >> 4);
  const int pc = p < count ? p : count - 1;
  T xo, U;
  bool fr;
  const int res = boxqp1<T, FAST>(x0[pc], Q[pc], c[pc], lo[pc], hi[pc],
                                  lstep[0], ls_tail, lane, xo, U, fr);
  if (p < count && (lane & 15) == 0) {
    x[p] = xo;
    result[p] = res;
    free_mask[p] = fr ? 1 : 0;
  }
}

}  // namespace n4

template <typename T>
static int launch_n4(const RiccatiArgs<T>& a, hipStream_t st, bool fast_math) {
  // four trajectories per one-wave workgroup; bounded problems solve the
  // BoxQP in closed form (QpClosed), the reference's loop as fall-back
  const bool bounded = a.u_min != nullptr;
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  constexpr int G = 4;
  const dim3 grid((a.B + G - 1) / G), block(kWave);
#define PDDP_N4_LAUNCH(C, Bd, F)                                             \
  PDDP_LAUNCH((n4::riccati_n4_kernel<T, C, Bd, F, 4, Bd, 1>), grid, block, 0, \
              st, a)
  if (fast_math) {
    if (chol) { if (bounded) PDDP_N4_LAUNCH(true, true, true); else PDDP_N4_LAUNCH(true, false, true); }
    else { if (bounded) PDDP_N4_LAUNCH(false, true, true); else PDDP_N4_LAUNCH(false, false, true); }
  } else {
    if (chol) { if (bounded) PDDP_N4_LAUNCH(true, true, false); else PDDP_N4_LAUNCH(true, false, false); }
    else { if (bounded) PDDP_N4_LAUNCH(false, true, false); else PDDP_N4_LAUNCH(false, false, false); }
  }
#undef PDDP_N4_LAUNCH
  return launch_status();
}

}  // namespace pddp
