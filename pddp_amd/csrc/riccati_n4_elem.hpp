// riccati_n4_elem.hpp - the n = 4, m = 1 bounded eig-clamp sweep (branch B,
// ilqr.py:629-672) from the nominal trajectory, ONE wavefront per four
// trajectories and nothing shared between wavefronts: no roles, no phase
// barrier, no exchange buffers.
//
// Why: at B = 4096 every n = 4 sweep is a chain of N
// dependent steps on wavefronts that have a SIMD to themselves, and a lone
// wavefront issues one instruction every ~5 cycles whatever it depends on -
// the time of a step is the instruction count of its busiest wavefront plus
// what it waits for.  The four-role kernel of round 3 (retired, docs/history)
// cut the chain that crosses a step down to the scalar BoxQP, but paid an LDS
// exchange and an s_barrier per step: ~745 cycles for ~100 instructions per
// role.  This kernel goes the other way: the plain recursion in the
// lane-per-matrix-element mapping of riccati_n4.hpp (16 lanes per
// trajectory, V one register per lane, every product a v_fmac with a DPP
// operand), with everything that made that kernel 177 instructions per step
// removed -
//   * the records are not streamed from HBM (no DMA ring, no vmcnt
//     bookkeeping): every 16 steps the wavefront evaluates the next 16
//     records of its four trajectories itself, one lane per (trajectory,
//     step) - models.hpp record_of, the code of derivs_kernel - and lays them
//     out in its own LDS slice in the form its lanes read them: the skewed
//     copy of F_z as a 4x4 table whose rows are ds_read_b128 operands (2
//     reads instead of 9), {f, L_z, L_uz} interleaved per index (2 reads
//     instead of 5);
//   * the BoxQP is QpLean1 (below: every predicate in the sign bit of a
//     VGPR, no compare -> SGPR -> select round trips), QpClosed and the
//     reference's loop behind it for the rows it does not cover;
//   * the value update is the rank-one form V' = sym(Qzz) + c Quz Quz^T,
//     V_z' = Qz + w Quz (two FMAs) instead of the mirrored K-trees;
//   * two transposes (ds_bpermute) instead of three, issued before the BoxQP.
//
// Restates pddp/controllers/ilqr.py:489-526 (Q), :529-674 (backward, branch
// B) with the records of :393-486 evaluated in place.  Summation order
// differs from the reference's dot products (DPP butterflies / rotations):
// results agree to rounding.
#pragma once

#include <type_traits>
#include "models.hpp"
#include "riccati_n4_quad.hpp"  // boxqp1_wave, rank_one_coeffs

namespace pddp {

namespace n4d {
// ---- flags carried in the SIGN BIT of a 32-bit word (the lean BoxQP).
// A compare that goes through an SGPR pair (v_cmp -> v_cndmask) costs a
// dependent chain ~40 cycles per trip; a subtraction leaves the same
// predicate in the sign bit of a VGPR, where v_and / v_or / v_bfi combine it
// at 4 cycles each.  The two instructions the optimiser would turn back into
// compare + select are issued by hand.
typedef float f32x4 __attribute__((ext_vector_type(4)));
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

// What a sweep needs to evaluate its records itself: the nominal trajectory
// instead of `a.rec`, and where the stage costs and their sum go.
template <typename T>
struct GenArgs {
  const T* Z;      // [B][N + 1][n]
  const T* U;      // [B][N]  (un-clamped nominal actions)
  T* L;            // [B][N + 1] stage / terminal cost of the nominal
  T* J_opt;        // [B]: sum of L in t order, where `fresh` is set
  uint8_t* fresh;  // [B] nullable: "the nominal changed"; cleared
};
}  // namespace n4d

namespace n4e {

using n4::bperm;
using n4::dpp;
using n4::fma_;
using n4::group_sum;
using n4::kGain;
using n4::mul_nc;
using n4d::bsel;
using n4d::f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
using n4d::GenArgs;
using n4d::sgn;
using n4d::splat;

constexpr int kWaves = 4;    // wavefronts per workgroup (independent)
constexpr int kTrajW = 4;    // trajectories per wavefront
constexpr int kBlk = 16;     // steps per block = lanes of a trajectory's row
// image of one (trajectory, step) in LDS, floats:
//   [ 0, 16)  S[r][d] = F_z[(r + d) % 4][r]      (row r: one b128 operand)
//   [16, 24)  {F_u[x], L_uz[x]}, x = 0..3          (read by row index)
//   [24, 32)  {F_u[x], L_z[x]},  x = 0..3          (read by column index)
//   [32, 48)  L_zz row-major
//   [48, 52)  {L_uu, L_u, u_min - U, u_max - U}  (U: the un-clamped nominal
//             action; the BoxQP's bounds, constraint.py via ilqr.py:602-603)
// (every word of every read is used: a dead component of a wide read is a
// register the allocator hands out again while the read is still in flight -
// and the next write to it waits for ALL of the step's reads)
constexpr int kImg = 52;
// a trajectory's 16 images; + 16 floats: the four rows' reads of one step fall
// into four different quarters of the 64 banks
constexpr int kRowStride = kBlk * kImg + 16;
constexpr int kImgBuf = kTrajW * kRowStride;  // a block's images, four rows
// The gains of a block go out 16 steps at a time: lane (row, l) leaves its
// word of step s - K[l] for l < 4, k otherwise - in word l of the image of
// (row, s), which is dead by then (its words were read into registers a step
// earlier, and LDS executes a wavefront's operations in order): one ds_write
// per step, no exec mask, no address arithmetic on the chain, no LDS of its own.
constexpr int kTermRow = 24;  // terminal L_zz (16), L_z (4), cost, pad per row
constexpr int kTermSh = kTrajW * kTermRow;
// floats per wavefront (inline generator; the terminal rows alias the images,
// which are written after they are read: 13.25 KB - three workgroups per CU) /
// per (sweep, generator) pair
constexpr int kPairLdsInl = kImgBuf;
constexpr int kPairLdsOvl = 2 * kImgBuf + kTermSh;

// back-tracking step sizes of the reference's BoxQP loop as floats in global
// memory (the loop runs on ~0.02 % of the steps: no LDS copy, no barrier)
struct LsTableF {
  float v[n4::kLsSteps];
  constexpr LsTableF() : v() {
    for (int n = 0; n < n4::kLsSteps; ++n) v[n] = (float)n4::kLs.v[n];
  }
};
__device__ constexpr LsTableF kLsF{};
template <typename T>
PDDP_DEV const T* ls_table();
template <>
PDDP_DEV const float* ls_table<float>() { return kLsF.v; }
template <>
PDDP_DEV const double* ls_table<double>() { return n4::kLs.v; }
template <typename T>
struct Vec {
  typedef T v4 __attribute__((ext_vector_type(4)));
  typedef T v2 __attribute__((ext_vector_type(2)));
};

// The scalar BoxQP of constraint.py:150-266 for m = 1 as straight-line code on
// sign bits: the clamped warm start, one projected Newton step, and every
// DECISION the reference's loop takes on the way - the exit tests of both
// passes (clamped gradient / small gradient / relative improvement) and the
// stale `free` flag it returns with.  Two things the loop does are not
// evaluated, because for one action they cannot change its answer:
//  * the second full Newton step from a live interior x1 (constraint.py:237-259,
//    second pass): x2 = clamp(x1 + (newton - x1)) is the rounded Newton point
//    again - |x2 - x1| <= 2 ulp, both are roundings of the same number;
//  * the Armijo back-tracking (constraint.py:241-252).  With s = newton - xs
//    and theta the fraction of s the clamp lets through, f(x1) - f(xs) =
//    s g (theta - theta^2 / 2): the test `<= 0.1 s g` passes at the full step
//    for theta >= 0.1056, and below that the first step size 0.6^k that
//    passes still overshoots the bound (0.6^k > theta), so the accepted
//    candidate is clamp(xs + 0.6^k s) = the bound = x1 again.  In exact
//    arithmetic the loop returns x1 whatever the back-tracking does; it leaves
//    x1 only where f(x1) - f(xs) is rounding noise (0.02 % of the steps of the
//    benched workload in float32, none in float64).  Earlier builds detected
//    those steps and ran the loop for them (14 instructions per step on the
//    chain); without that the lean form agrees BETTER with the float64 oracle
//    (x: 0 instead of 11 of 59,980 differ by more than 1e-6; status and the
//    round's parity statistics unchanged - tests/test_gpu_parity.py::
//    test_lean_boxqp_of_the_benched_sweep_vs_oracle) and a round
//    is 2.9 us shorter.  The loop itself still runs behind QpClosed: on the
//    records-path variants (exact=True), in float64, and for the irregular
//    rows of the class test below.
struct QpLean1 {
  float x, inv;
  int free_w;  // flag in the sign bit
  PDDP_DEV void solve(float x0, float Q, float c, float lo, float hi) {
    const float d_lo = lo - x0, d_hi = x0 - hi;  // sign: x0 > lo, x0 < hi
    const float xs = __builtin_amdgcn_fmed3f(x0, lo, hi);
    const float hQ = 0.5f * Q;
    inv = __builtin_amdgcn_rcpf(Q);
    // ---- iteration 0                                        (:191-239)
    const float g0 = fma_(Q, xs, c);
    const int ncl0 = (sgn(d_hi) & ~sgn(d_lo) & ~sgn(g0)) | (~sgn(d_hi) & sgn(g0));
    const int small0 = sgn(__builtin_fabsf(g0) - 1e-8f);
    const int done0 = ncl0 | small0;
    const float s0 = fma_(c, -inv, -xs);  // newton - xs
    const float xa = xs + s0;
    const float x1 = __builtin_amdgcn_fmed3f(xa, lo, hi);
    const float d1_lo = lo - xa, d1_hi = xa - hi;  // x1 == lo <=> xa <= lo
    const float f0 = xs * fma_(hQ, xs, c);
    const float num = fma_(x1, fma_(hQ, x1, c), -f0);  // f1 - f0
    // ---- iteration 1: exit tests
    const int conv = sgn(fma_(-1e-8f, __builtin_fabsf(f0), -num));
    const float g1 = fma_(Q, x1, c);
    const int ncl1 = (sgn(d1_hi) & ~sgn(d1_lo) & ~sgn(g1)) | (~sgn(d1_hi) & sgn(g1));
    x = bsel(splat(done0), xs, x1);
    // free = (done0 & !ncl0) | (!done0 & (conv | !ncl1)), done0 = ncl0 | small0
    free_w = ~ncl0 & (small0 | conv | ~ncl1);
  }
};

// The products and reductions of one step, hand-scheduled: 27 instructions,
// every DPP read at least two instructions behind the write of its source (the
// compiler has to pad a dependent DPP chain with s_nop - nine per step when
// these were seven separate statements; it also folds a DPP move into v_mul /
// v_add but not into v_fmac, riccati_n4_quad.hpp).  Lane (i, j) of a row:
//   Quu  = Luu + sum_ij f[i] V[i][j] f[j]      Qu = Lu + sum_j f[j] V_z[j]
//   A    = F^T V:  sum_d F[(i+d)%4][i] V[(i+d)%4][j]        (row_ror by 16 - 4d)
//   Qzz  = Lzz[i][j] + sum_d A[i][(j+d)%4] F[(j+d)%4][j]    (quad rotations)
//   Qzc  = Lz[j] + sum_d F[(j+d)%4][j] V_z[(j+d)%4]         (column form)
//   Quzr = Luz[i] + sum_j A[i][j] f[j]                      (row form)
// The reductions are butterflies of ROUNDED products (separate v_mul), whose
// results are bit-identical in every lane that holds a copy (a + b == b + a).
template <typename T>
struct StepCoreT {
  T Quu, Qu, Qzz, Qzc, Quzr;
};
typedef StepCoreT<float> StepCore;
// The same products for any scalar type, statement by statement (the DPP
// helpers of riccati_n4.hpp; a 64-bit DPP move is two 32-bit ones): the
// float64 instantiation of the sweep (pddp_sweep_nominal_f64, cartpole).
template <typename T>
PDDP_DEV StepCoreT<T> step_core(T V, T vc, T fr, T fc,
                                typename Vec<T>::v4 Fs, typename Vec<T>::v4 Fq,
                                T Lzz, T Lzc, T Luzr, T Luu, T Lu) {
  using n4::dot_cols;
  using n4::dot_rows;
  using n4::from_col_plus;
  using n4::from_row_plus;
  StepCoreT<T> o;
  o.Quu = Luu + dot_cols(dot_rows(fr, V), fc);
  o.Qu = Lu + dot_cols(fc, vc);
  T A = V * Fs[0];
  A = fma_(from_row_plus<1>(V), Fs[1], A);
  A = fma_(from_row_plus<2>(V), Fs[2], A);
  A = fma_(from_row_plus<3>(V), Fs[3], A);
  T Qzz = fma_(A, Fq[0], Lzz);
  Qzz = fma_(from_col_plus<1>(A), Fq[1], Qzz);
  Qzz = fma_(from_col_plus<2>(A), Fq[2], Qzz);
  o.Qzz = fma_(from_col_plus<3>(A), Fq[3], Qzz);
  T Qzc = fma_(vc, Fq[0], Lzc);
  Qzc = fma_(from_col_plus<1>(vc), Fq[1], Qzc);
  Qzc = fma_(from_col_plus<2>(vc), Fq[2], Qzc);
  o.Qzc = fma_(from_col_plus<3>(vc), Fq[3], Qzc);
  o.Quzr = Luzr + dot_cols(A, fc);
  return o;
}
PDDP_DEV StepCore step_core(float V, float vc, float fr, float fc, f32x4 Fs,
                            f32x4 Fq, float Lzz, float Lzc, float Luzr,
                            float Luu, float Lu) {
  StepCore o;
  float A;
#define PDDP_RM " row_mask:0xf bank_mask:0xf\n\t"
  asm("v_mul_f32 %[p1], %[fr], %[V]\n\t"
      "v_mul_f32 %[p2], %[fc], %[vc]\n\t"
      "v_mul_f32 %[A], %[V], %[Fs0]\n\t"
      "v_add_f32_dpp %[p1], %[p1], %[p1] row_ror:8" PDDP_RM
      "v_add_f32_dpp %[p2], %[p2], %[p2] quad_perm:[2,3,0,1]" PDDP_RM
      "v_fmac_f32_dpp %[A], %[V], %[Fs1] row_ror:12" PDDP_RM
      "v_add_f32_dpp %[p1], %[p1], %[p1] row_ror:12" PDDP_RM
      "v_add_f32_dpp %[p2], %[p2], %[p2] quad_perm:[1,0,3,2]" PDDP_RM
      "v_fmac_f32_dpp %[A], %[V], %[Fs2] row_ror:8" PDDP_RM
      "v_mul_f32 %[p1], %[p1], %[fc]\n\t"
      "v_fmac_f32_dpp %[A], %[V], %[Fs3] row_ror:4" PDDP_RM
      "v_fma_f32 %[Qzc], %[vc], %[Fq0], %[Lzc]\n\t"
      "v_add_f32_dpp %[p1], %[p1], %[p1] quad_perm:[2,3,0,1]" PDDP_RM
      "v_fmac_f32_dpp %[Qzc], %[vc], %[Fq1] quad_perm:[1,2,3,0]" PDDP_RM
      "v_fma_f32 %[Qzz], %[A], %[Fq0], %[Lzz]\n\t"
      "v_add_f32_dpp %[p1], %[p1], %[p1] quad_perm:[1,0,3,2]" PDDP_RM
      "v_fmac_f32_dpp %[Qzz], %[A], %[Fq1] quad_perm:[1,2,3,0]" PDDP_RM
      "v_fmac_f32_dpp %[Qzc], %[vc], %[Fq2] quad_perm:[2,3,0,1]" PDDP_RM
      "v_mul_f32 %[q1], %[A], %[fc]\n\t"
      "v_fmac_f32_dpp %[Qzz], %[A], %[Fq2] quad_perm:[2,3,0,1]" PDDP_RM
      "v_fmac_f32_dpp %[Qzc], %[vc], %[Fq3] quad_perm:[3,0,1,2]" PDDP_RM
      "v_add_f32_dpp %[q1], %[q1], %[q1] quad_perm:[2,3,0,1]" PDDP_RM
      "v_fmac_f32_dpp %[Qzz], %[A], %[Fq3] quad_perm:[3,0,1,2]" PDDP_RM
      "v_add_f32 %[p1], %[Luu], %[p1]\n\t"
      "v_add_f32_dpp %[q1], %[q1], %[q1] quad_perm:[1,0,3,2]" PDDP_RM
      "v_add_f32 %[p2], %[Lu], %[p2]\n\t"
      "v_add_f32 %[q1], %[Luzr], %[q1]\n\t"
      : [p1] "=&v"(o.Quu), [p2] "=&v"(o.Qu), [q1] "=&v"(o.Quzr),
        [Qzz] "=&v"(o.Qzz), [Qzc] "=&v"(o.Qzc), [A] "=&v"(A)
      : [V] "v"(V), [vc] "v"(vc), [fr] "v"(fr), [fc] "v"(fc),
        [Fs0] "v"(Fs[0]), [Fs1] "v"(Fs[1]), [Fs2] "v"(Fs[2]), [Fs3] "v"(Fs[3]),
        [Fq0] "v"(Fq[0]), [Fq1] "v"(Fq[1]), [Fq2] "v"(Fq[2]), [Fq3] "v"(Fq[3]),
        [Lzz] "v"(Lzz), [Lzc] "v"(Lzc), [Luzr] "v"(Luzr), [Luu] "v"(Luu),
        [Lu] "v"(Lu));
#undef PDDP_RM
  return o;
}

// The gains of one step for the four trajectories of a wavefront (row = lane
// >> 4, the same values in the 16 lanes of a row): the scalar BoxQP of
// ilqr.py:645-656 / constraint.py:150-266 on e = (Quu < 0 ? 1e-12 : Quu) + reg
// (ilqr.py:633-634) warm-started at the previous step's k, then c and w of the
// rank-one value update.  float: QpLean1, and - behind ONE class test - the
// closed form of riccati_n4.hpp and the reference's loop for the rows the lean
// form does not cover; float64: the closed form (IEEE division) and the loop.
// `status` / `alive_m`: a row that fails gets its PDDP_BWD_* code and leaves
// the mask of live rows.  Shared by the sweep (elem_sweep_body) and by
// pddp_boxqp_m1_lean_f32, the unit-test entry of exactly this routine.
template <typename T>
struct ElemGains {
  T kt, sK, c, wv;
};
template <typename T>
PDDP_DEV ElemGains<T> elem_gains(T kprev, T Quu, T Qu, T reg, T lo_b, T hi_b,
                                 int lane, int& status,
                                 unsigned long long& alive_m) {
  constexpr bool F32 = std::is_same<T, float>::value;
  const int l = lane & 15;
  const unsigned long long lane_bit = 1ull << lane;
  T qp_Q, kt = T(0), sK = T(0), c = T(0), wv = T(0);
  unsigned long long oddm;
  if constexpr (F32) {
    qp_Q = bsel(splat(sgn(Quu)), 1e-12f, Quu) + reg;
    QpLean1 ql;
    ql.solve(kprev, qp_Q, Qu, lo_b, hi_b);
    kt = ql.x;
    sK = __int_as_float(splat(ql.free_w) & __float_as_int(ql.inv));
    n4q::rank_one_coeffs(kt, sK, Quu, Qu, c, wv);
    // anything the lean form does not cover - a non-finite Quu (0 Quu is
    // NaN then), a Q that is not positive and finite - in ONE class test:
    // QpClosed, the reference's loop behind it, for those rows only
    const T chk = fma_(Quu, T(0), qp_Q);
    unsigned long long regular;  // (the mask straight into a scalar pair)
    asm("v_cmp_class_f32 %0, %1, %2" : "=s"(regular) : "v"(chk), "v"(0x180));
    oddm = ~regular & alive_m;
  } else {
    // float64: every live row through the closed form of riccati_n4.hpp
    // (IEEE division; the reference's loop behind it) - the block below
    qp_Q = (Quu < T(0) ? T(1e-12) : Quu) + reg;
    oddm = alive_m;
  }
  if (__builtin_expect(oddm != 0, F32 ? 0 : 1)) {
    const bool take = (oddm & lane_bit) != 0;
    int st = PDDP_BWD_OK;
    if (!is_finite(Quu)) st = PDDP_BWD_NAN;      // eig raises (ilqr.py:631)
    n4::QpClosed<T, true> qc;
    qc.solve(kprev, qp_Q, Qu, lo_b, hi_b);
    T kx = qc.x;
    bool Kzero = !qc.free_, fail = qc.fail;
    const bool slow = qc.slow & take;
    if (__any(slow)) {
      // rare: the reference's loop, one slow trajectory at a time on the
      // whole wavefront
      unsigned long long todo = __ballot(slow && l == 0);
      while (todo != 0) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const n4::SlowQpOut<T> o = n4q::boxqp1_wave<T, true>(
            __shfl(kprev, src), __shfl(qp_Q, src), __shfl(Qu, src),
            __shfl(lo_b, src), __shfl(hi_b, src), ls_table<T>(), lane);
        const bool mine = (lane >> 4) == (src >> 4);
        kx = mine ? o.x : kx;
        Kzero = mine ? ((o.result_free & 1) == 0) : Kzero;
        fail = mine ? (o.result_free < 2) : fail;
      }
    }
    const T sx = Kzero ? T(0) : qc.inv;
    const int stt = st != PDDP_BWD_OK ? st : (fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
    T cx, wx;
    n4q::rank_one_coeffs(kx, sx, Quu, Qu, cx, wx);
    kt = take ? kx : kt; sK = take ? sx : sK;
    c = take ? cx : c; wv = take ? wx : wv;
    const bool bad = take & (stt != PDDP_BWD_OK);
    status = bad ? stt : status;
    alive_m &= ~__ballot(bad);
  }
  return ElemGains<T>{kt, sK, c, wv};
}

#ifdef PDDP_ELEM_MARKS
// time marks of wavefront 0 of workgroup 0: begin, first step, last step done,
// end (tools/elem_sweep_marks.py)
// [4] cycles in the generator passes, [5] cycles in the steps
__device__ long long g_elem_marks[8];
#define PDDP_EM_MARK(I) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_elem_marks[I] = clock64(); } while (0)
#define PDDP_EM_ACC(I, T0) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_elem_marks[I] += clock64() - (T0); } while (0)
#define PDDP_EM_NOW() clock64()
#else
#define PDDP_EM_MARK(I)
#define PDDP_EM_ACC(I, T0)
#define PDDP_EM_NOW() 0
#endif

#ifdef PDDP_WG_TIMELINE
// when every workgroup starts, reaches its first step, ends (wall clock, 100
// MHz, the same counter for the whole chip) and where it ran: wavefront 0
// (sweep) in slots 0-3, wavefront 4 (its generator) in 4-7, 8 HW_ID, 9 XCC_ID,
// 10 / 11 the shader clock at begin / end (tools/wg_timeline.py)
__device__ long long g_elem_timeline[1024][12];
#define PDDP_TL(I)                                                            \
  do {                                                                        \
    if ((threadIdx.x & 255) == 0 && blockIdx.x < 1024)                        \
      g_elem_timeline[blockIdx.x][(I) + 4 * (threadIdx.x >> 8)] =             \
          wall_clock64();                                                     \
  } while (0)
#define PDDP_TL_HW(I)                                                         \
  do {                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                              \
      g_elem_timeline[blockIdx.x][8] =                                        \
          __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));               \
      g_elem_timeline[blockIdx.x][9] =                                        \
          __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));               \
      g_elem_timeline[blockIdx.x][I] = clock64();                             \
    }                                                                         \
  } while (0)
#else
#define PDDP_TL(I)
#define PDDP_TL_HW(I)
#endif

// OVL = false: four independent wavefronts per workgroup, each evaluates its
// own records between its blocks of steps (LDS: 17.6 KB per wavefront - up to
// two workgroups per CU).
// OVL = true: eight wavefronts - four (sweep, generator) pairs, the two of a
// pair on one SIMD.  The generator evaluates block jb + 1 into the second
// image buffer while the sweep runs the steps of block jb; one s_barrier per
// block (the generator sleeps there most of the time: a wavefront that has
// its SIMD to itself uses a fifth of its issue slots, the partner's 250
// instructions per block fit into the rest).  LDS: 31 KB per pair - one
// workgroup per CU, for batches of up to 16 trajectories per CU.
//
// ROUND (round_n4.hip; implies OVL): the sweep is the first phase of a launch
// that goes on with the line search of the same four trajectories in the same
// wavefronts (line_search_lds.hpp).  The gains then ALSO land in a region of
// LDS in the layout the search reads ([t][k, K0..K3] per trajectory), the
// generator stages the nominal's states and actions into the image buffer the
// last block does not use, the nominal's cost is handed over in LDS, and both
// wavefronts of a pair leave through one more barrier instead of returning.
// Returns false when the pair has nothing to do (both wavefronts alike).
struct RoundOut {       // (ROUND) what the search phase needs, per lane
  const float* Zs;      // this lane's trajectory: states [N + 1][4] in LDS
  const float* Gs;      //   [N][6]: k, K[0..3], the nominal action
  const float* Us;      //   = Gs + 5 (stride 6)
  int status;           // sweep wavefront: PDDP_BWD_* of the trajectory
  float J_opt;          // cost of the nominal (summed here when it was new)
  float* carry_rows;    // [17][5] rows carried between rounds, or NULL
};
constexpr int round_zu_stride(int N) { return 4 * (N + 1); }
// (ROUND, several rounds per launch) the nominal's last seventeen rows {z, u}
// of the pair's four trajectories, kept in LDS from round to round: what the
// next round's first block of records and its terminal value are made from
// (index N - t: 0 the terminal state) - the winner's rows go there from the
// search's tail as well as to global memory, and the next round does not wait
// out a trip to L2 for rows its own workgroup has just produced
constexpr int kCarryRows = kBlk + 1, kCarryW = 5;
constexpr int kCarryF = kTrajW * kCarryRows * kCarryW;  // 340 floats
// (ROUND) a step's row in the search's LDS table: k, K[0..3] and the nominal
// action - 24 bytes, three 8-byte reads per rollout step
constexpr int kGainL = kGain + 1;
constexpr int round_gains_floats(int N) { return (kTrajW * N * kGainL + 3) & ~3; }

template <typename T, unsigned QM, bool OVL, bool ROUND>
PDDP_DEV bool elem_sweep_body(const RiccatiArgs<T>& a, const GenArgs<T>& gen,
                              const ProblemT<T>& prob, unsigned char* smem_raw,
                              RoundOut& ro,
                              const unsigned tid = threadIdx.x,
                              const int carry = -1) {
  // carry (ROUND): -1 none; 1 this is the launch's first round - the rows
  // above are filled from global memory; 0 a later round - they are read
  static_assert(OVL || !ROUND, "");
  // (float64: the inline form only - two image buffers of doubles for four
  // pairs are 220 KB - with the closed-form BoxQP of riccati_n4.hpp and IEEE
  // division: the design held to the oracle at 1e-9, tests/test_gpu_parity.py)
  constexpr bool F32 = std::is_same<T, float>::value;
  static_assert(F32 || (!OVL && !ROUND), "");
  using V4 = typename Vec<T>::v4;
  using V2 = typename Vec<T>::v2;
  constexpr int MODEL = PDDP_MODEL_CARTPOLE;
  constexpr RecLayout lay(4, 1);
  const int kPairLds = OVL ? kPairLdsOvl + (ROUND ? round_gains_floats(a.N) +
                                                        (carry >= 0 ? kCarryF : 0)
                                                  : 0)
                           : kPairLdsInl;
  // (`tid`: threadIdx.x - through an opaque move in the loop over rounds of
  // round_n4.hip, so that nothing derived from it is hoisted out of a round)
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)tid >> 6);
  const int pair = OVL ? (wave & (kWaves - 1)) : wave;
  const bool is_gen = OVL && wave >= kWaves;
  T* const img0 = reinterpret_cast<T*>(smem_raw) + pair * kPairLds;
  // (inline: the first rows of the image buffer, before pass 0 writes it)
  T* const term_w = OVL ? img0 + 2 * kImgBuf : img0;

  const int row = lane >> 4, l = lane & 15, i = l >> 2, j = l & 3;
  const int N = a.N;
  const int b0 = (blockIdx.x * kWaves + pair) * kTrajW;
  const int b = b0 + row;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const T* Zg = gen.Z + (size_t)bc * (size_t)(N + 1) * 4;
  const T* Ug = gen.U + (size_t)bc * (size_t)N;
  const int nblk = (N + kBlk - 1) / kBlk;
  // (both wavefronts of a pair decide alike: they own the same trajectories;
  // s_barrier does not wait for wavefronts that have ended)
  if (b0 >= a.B) return false;
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return false;
  PDDP_EM_MARK(0);
  PDDP_TL(0);
  PDDP_TL_HW(10);
  const T umin = a.u_min[0], umax = a.u_max[0];
  const int rbase = row * kRowStride;

  // =============================================================== generator
  // lane (row, l) evaluates the record of step N - 1 - 16 jb - l of
  // trajectory `row` (models.hpp record_of: the code of derivs_kernel) and
  // writes its image; the operands of a block are requested one block ahead
  T zq[4], uq, Jacc = T(0), l_term = T(0);
  // (ROUND) rows carried from round to round, this lane's trajectory's
  [[maybe_unused]] T* const carry_w =
      img0 + kPairLdsOvl + (ROUND ? round_gains_floats(N) : 0) +
      row * kCarryRows * kCarryW;
  if constexpr (ROUND) ro.carry_rows = carry >= 0 ? carry_w : nullptr;
  auto request = [&](int jb) {
    int tau = N - 1 - kBlk * jb - l;
    tau = tau < 0 ? 0 : tau;
    if constexpr (ROUND) {
      if (jb == 0 && carry == 0) {  // from the rows the last round left
        const T* cr = carry_w + (N - tau) * kCarryW;
        zq[0] = cr[0]; zq[1] = cr[1]; zq[2] = cr[2]; zq[3] = cr[3];
        uq = cr[4];
        return;
      }
    }
    const V4 v = *reinterpret_cast<const V4*>(Zg + 4 * tau);
    zq[0] = v[0]; zq[1] = v[1]; zq[2] = v[2]; zq[3] = v[3];
    uq = Ug[tau];
    if constexpr (ROUND) {
      if (jb == 0 && carry == 1 && N - tau <= kBlk) {
        T* cr = carry_w + (N - tau) * kCarryW;
        cr[0] = zq[0]; cr[1] = zq[1]; cr[2] = zq[2]; cr[3] = zq[3];
        cr[4] = uq;
      }
    }
  };
  // terminal value function V = L_zz[N], V_z = L_z[N]: evaluated by every lane
  // of the row (the wavefront pays the instructions once either way), handed
  // over through LDS
  auto terminal = [&]() {
    T zN[4];
    if (ROUND && carry == 0) {
      zN[0] = carry_w[0]; zN[1] = carry_w[1]; zN[2] = carry_w[2];
      zN[3] = carry_w[3];
    } else {
      const V4 zNv = *reinterpret_cast<const V4*>(Zg + 4 * N);
      zN[0] = zNv[0]; zN[1] = zNv[1]; zN[2] = zNv[2]; zN[3] = zNv[3];
      if (ROUND && carry == 1 && l == 0) {
        carry_w[0] = zN[0]; carry_w[1] = zN[1]; carry_w[2] = zN[2];
        carry_w[3] = zN[3];
      }
    }
    T lz[4], lzz[16], lu[1], luu[1];
    l_term = cost_derivs<T, MODEL>(prob, zN, nullptr, trig_of<T, MODEL>(zN),
                                   true, lz, lzz, lu, luu);
    T* scr = term_w + row * kTermRow;
    if (l == 0) {
#pragma unroll
      for (int k = 0; k < 16; k += 4)
        *reinterpret_cast<V4*>(scr + k) =
            V4{lzz[k], lzz[k + 1], lzz[k + 2], lzz[k + 3]};
      *reinterpret_cast<V4*>(scr + 16) = V4{lz[0], lz[1], lz[2], lz[3]};
      scr[20] = l_term;
      if (counted) gen.L[(size_t)bc * (size_t)(N + 1) + N] = l_term;
    }
  };
  auto pass = [&](int jb, T* buf) {
    const T z[4] = {zq[0], zq[1], zq[2], zq[3]};
    const T u = uq;
    if (jb + 1 < nblk) request(jb + 1);
    const int tau = N - 1 - kBlk * jb - l;
    T w[lay.stride];
    const T lc = record_of<T, MODEL, QM>(prob, z, &u, false, true, a.u_min,
                                         a.u_max, w);
    T* dst = buf + rbase + l * kImg;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<V4*>(dst + 4 * r) =
          V4{w[lay.oFz + ((r + 0) & 3) * 4 + r],
                w[lay.oFz + ((r + 1) & 3) * 4 + r],
                w[lay.oFz + ((r + 2) & 3) * 4 + r],
                w[lay.oFz + ((r + 3) & 3) * 4 + r]};
#pragma unroll
    for (int x = 0; x < 4; x += 2) {
      *reinterpret_cast<V4*>(dst + 16 + 2 * x) =
          V4{w[lay.oFu + x], w[lay.oLuz + x], w[lay.oFu + x + 1],
                w[lay.oLuz + x + 1]};
      *reinterpret_cast<V4*>(dst + 24 + 2 * x) =
          V4{w[lay.oFu + x], w[lay.oLz + x], w[lay.oFu + x + 1],
                w[lay.oLz + x + 1]};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<V4*>(dst + 32 + 4 * r) =
          V4{w[lay.oLzz + 4 * r], w[lay.oLzz + 4 * r + 1],
                w[lay.oLzz + 4 * r + 2], w[lay.oLzz + 4 * r + 3]};
    *reinterpret_cast<V4*>(dst + 48) =
        V4{w[lay.oLuu], w[lay.oLu], umin - w[lay.oU], umax - w[lay.oU]};
    if (tau >= 0) {
      if (counted) gen.L[(size_t)bc * (size_t)(N + 1) + tau] = lc;
      Jacc += lc;
    }
  };
  // J_opt = sum of the stage costs and the terminal one (ilqr.py:289), for
  // the trajectories whose nominal changed
  auto finish_costs = [&]() {
    const bool sums = counted && (gen.fresh == nullptr || gen.fresh[bc] != 0);
    // (OVL: the terminal cost is the sweep wavefront's; inline: our own)
    const T Jrow = group_sum(Jacc) + (OVL ? term_w[row * kTermRow + 20] : l_term);
    if constexpr (ROUND) {
      // the cost the accept test compares with (ilqr.py:166): the new sum, or
      // what the nominal already had
      const T Jold = gen.J_opt[bc];
      if (l == 0) term_w[row * kTermRow + 21] = sums ? Jrow : Jold;
    }
    if (sums && l == 0) {
      gen.J_opt[bc] = Jrow;
      if (gen.fresh != nullptr) gen.fresh[bc] = 0;
    }
  };
  // the gains of a block's first `cnt` steps (t_top, t_top - 1, ...) from the
  // dead image words they were left in (see `step`) to HBM - lane (row, l)
  // stores step l's five words - and, ROUND, to the search's rows in LDS
  [[maybe_unused]] T* const gains_w = img0 + kPairLdsOvl;  // (ROUND) [row][N][5]
  auto flush_gains = [&](const T* ib, int t_top, int cnt) {
    const T* sg = ib + l * kImg + rbase;
    const V4 Kv = *reinterpret_cast<const V4*>(sg);
    const T kv = sg[4];
    if (exists && l < cnt) {
      T* g = a.gains + ((size_t)bc * (size_t)N + (size_t)(t_top - l)) * kGain;
      g[0] = kv; g[1] = Kv[0]; g[2] = Kv[1]; g[3] = Kv[2]; g[4] = Kv[3];
      if constexpr (ROUND) {
        T* gl = gains_w + (row * N + (t_top - l)) * kGainL;
        gl[0] = kv; gl[1] = Kv[0]; gl[2] = Kv[1]; gl[3] = Kv[2]; gl[4] = Kv[3];
      }
    }
  };
  [[maybe_unused]] T* const zu_w =
      img0 + (nblk & 1) * kImgBuf + row * round_zu_stride(N);
  // (ROUND) what the search reads, and the way out of the sweep phase
  [[maybe_unused]] auto round_out = [&](int status_) {
    if constexpr (ROUND) {
      ro.Zs = zu_w;
      ro.Gs = gains_w + row * N * kGainL;
      ro.Us = ro.Gs + kGain;  // (the sixth word of a step's row)
      ro.status = status_;
      n4::lds_publish_barrier();  // the last barrier: gains, nominal, J_opt
      ro.J_opt = term_w[row * kTermRow + 21];
    }
  };
  if constexpr (OVL) {
    if (is_gen) {
      // (the terminal state is the sweep wavefront's, which has nothing else
      // to do before its first block)
      request(0);
      pass(0, img0);
      PDDP_TL(1);
      n4::lds_publish_barrier();  // barrier 0: block 0, the terminal state
      for (int jb = 1; jb < nblk; ++jb) {
        // the gains of block jb - 2 leave from the buffer block jb is about
        // to be written to (the sweep wavefront only flushes its last block:
        // ~700 cycles a block off its chain)
        if (jb >= 2)
          flush_gains(img0 + (jb & 1) * kImgBuf, N - 1 - kBlk * (jb - 2), kBlk);
        pass(jb, img0 + (jb & 1) * kImgBuf);
        n4::lds_publish_barrier();  // barrier jb (LDS only: no vmcnt)
      }
      if (nblk >= 2)
        flush_gains(img0 + (nblk & 1) * kImgBuf, N - 1 - kBlk * (nblk - 2), kBlk);
      if constexpr (ROUND) {
        // the nominal's rows for the search, into the image buffer the last
        // block does not use (its gains were flushed just above); every load
        // requested before the first write: one memory latency
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int kCh = 8;  // N + 1 <= 128 (checked by the launcher)
        V4 zc[kCh];
        T uc[kCh];
#pragma unroll
        for (int c = 0; c < kCh; ++c) {
          const int tz = l + 16 * c;
          zc[c] = *reinterpret_cast<const V4*>(Zg + 4 * (tz <= N ? tz : N));
          uc[c] = Ug[tz < N ? tz : 0];
        }
#pragma unroll
        for (int c = 0; c < kCh; ++c) {
          const int tz = l + 16 * c;
          if (tz <= N) *reinterpret_cast<V4*>(zu_w + 4 * tz) = zc[c];
          if (tz < N) gains_w[(row * N + tz) * kGainL + kGain] = uc[c];
        }
      }
      finish_costs();
      PDDP_TL(3);
      if constexpr (ROUND) round_out(PDDP_BWD_OK);
      return true;
    }
  }

  // =================================================================== sweep
  const T reg = (T)a.reg[bc];
  // LDS offsets of this lane (floats, inside a step's image)
  const int oA = rbase + 4 * i, oB = rbase + 4 * j, oL = rbase + 32 + l;
  const int oTi = rbase + 16 + 2 * i, oTj = rbase + 24 + 2 * j;
  const int tr_addr = ((lane & 48) | (j * 4 + i)) * 4;  // lane (j, i)
  if constexpr (OVL) {
    // the sweep wins every issue slot both wavefronts of the SIMD ask for
    __builtin_amdgcn_s_setprio(3);
    terminal();
    n4::lds_publish_barrier();  // barrier 0
  } else {
    request(0);
    terminal();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  T V = term_w[row * kTermRow + l];
  T vc = term_w[row * kTermRow + 16 + j];

  struct Words {
    V4 Fs, Fq;
    V2 Ti, Tj;  // {f[i], Luz[i]}, {f[j], Lz[j]}
    V4 Sc;      // Luu, Lu, lo, hi
    T Lzz;
  };
  auto gather = [&](const T* ib, const int s) {
    const T* p = ib + s * kImg;
    Words w;
    w.Fs = *reinterpret_cast<const V4*>(p + oA);       // F_z[(i+d)%4][i]
    w.Fq = *reinterpret_cast<const V4*>(p + oB);       // F_z[(j+d)%4][j]
    w.Ti = *reinterpret_cast<const V2*>(p + oTi);
    w.Tj = *reinterpret_cast<const V2*>(p + oTj);
    w.Lzz = p[oL];
    w.Sc = *reinterpret_cast<const V4*>(p + 48 + rbase);
    return w;
  };
  T kprev = T(0);
  int status = PDDP_BWD_OK;
  // counted and status still OK (changes in the odd path only), as a lane mask
  unsigned long long alive_m = __ballot(counted);
  // gains of a block: word l of the (dead) image of (row, step)
  const int ostage = rbase + l;

  auto step = [&](const Words& w, T* ib, const int s) {
    const StepCoreT<T> q = step_core(V, vc, w.Ti[0], w.Tj[0], w.Fs, w.Fq,
                                     w.Lzz, w.Tj[1], w.Ti[1], w.Sc[0], w.Sc[1]);
    const T Quu = q.Quu, Qu = q.Qu;
    // transposes (lane (i, j) <- lane (j, i)) issued HERE: their latency under
    // the BoxQP (left to itself the scheduler sinks them to their use)
    const T QzzT = bperm(tr_addr, q.Qzz);
    const T Quzc = bperm(tr_addr, q.Quzr);
    __builtin_amdgcn_sched_barrier(0);
    // ---- gains: the scalar BoxQP of the step (elem_gains above)
    const ElemGains<T> g_ = elem_gains<T>(kprev, Quu, Qu, reg, w.Sc[2], w.Sc[3],
                                          lane, status, alive_m);
    const T kt = g_.kt, sK = g_.sK, c = g_.c, wv = g_.wv;
    // ---- gains out: K = -s Quz (column form in lanes (0, j)), k elsewhere
    ib[s * kImg + ostage] = (l < 4) ? -(sK * Quzc) : kt;
    kprev = kt;
    // ---- value update (ilqr.py:664-672 with K = -s Quz):
    // V' = sym(Qzz) + c Quz Quz^T,  V_z' = Qz + w Quz
    // (every term symmetric in (i, j) bit for bit: a + b == b + a)
    V = fma_(T(0.5), n4::opaque(q.Qzz + QzzT),
             mul_nc(c, mul_nc(q.Quzr, Quzc)));
    vc = fma_(wv, Quzc, q.Qzc);
  };

  PDDP_EM_MARK(1);
  PDDP_TL(1);
  int t = N - 1;
  for (int jb = 0; jb < nblk; ++jb) {
    T* ib = img0;
    if constexpr (OVL) {
      ib = img0 + (jb & 1) * kImgBuf;
    } else {
      [[maybe_unused]] const long long tp0 = PDDP_EM_NOW();
      pass(jb, img0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PDDP_EM_ACC(4, tp0);
    }
    [[maybe_unused]] const long long ts0 = PDDP_EM_NOW();
    // two word sets alternate: one in use, one being read for the next step
    Words wa = gather(ib, 0), wb = wa;
    if (t >= kBlk - 1) {
#pragma unroll
      for (int s = 0; s < kBlk; s += 2) {
        wb = gather(ib, s + 1);
        step(wa, ib, s);
        if (s + 2 < kBlk) wa = gather(ib, s + 2);
        step(wb, ib, s + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // (OVL: the generator flushes every block but the last)
      if (!OVL || jb + 1 == nblk) flush_gains(ib, t, kBlk);
      t -= kBlk;
    } else {
      // the last, partial block
      const int cnt = t + 1;
#pragma unroll 1
      for (int s = 0; s < cnt; ++s) {
        const Words w = gather(ib, s);
        step(w, ib, s);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      flush_gains(ib, t, cnt);
      t = -1;
    }
    // (the next pass overwrites the images: every read above has returned -
    // its value was consumed)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PDDP_EM_ACC(5, ts0);
    if constexpr (OVL) {
      if (jb + 1 < nblk) {
        [[maybe_unused]] const long long tb0 = PDDP_EM_NOW();
        n4::lds_publish_barrier();  // barrier jb + 1
        PDDP_EM_ACC(6, tb0);
      }
    }
  }
  PDDP_EM_MARK(2);
  PDDP_TL(2);
  if (counted && l == 0) a.status[bc] = status;
  if constexpr (!OVL) finish_costs();
  PDDP_EM_MARK(3);
  PDDP_TL(3);
  PDDP_TL_HW(11);
  if constexpr (ROUND) round_out(status);
  return true;
}

template <unsigned QM, bool OVL>
__global__ __launch_bounds__((OVL ? 2 : 1) * kWaves * kWave) void
riccati_n4_elem_kernel(RiccatiArgs<float> a, GenArgs<float> gen,
                       ProblemT<float> prob) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RoundOut ro;
  elem_sweep_body<float, QM, OVL, false>(a, gen, prob, smem_raw, ro);
}
// float64 (cartpole, bounded eig-clamp branch): the inline form
template <unsigned QM>
__global__ __launch_bounds__(kWaves * kWave) void riccati_n4_elem_f64_kernel(
    RiccatiArgs<double> a, GenArgs<double> gen, ProblemT<double> prob) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RoundOut ro;
  elem_sweep_body<double, QM, false, false>(a, gen, prob, smem_raw, ro);
}

}  // namespace n4e

// 0 auto (by batch), 3 / 4: the record generator inline / on wavefronts of
// its own (the numbers of round 4's A/B knob: 1 was the four-role kernel of
// round 3, removed)
inline int& nominal_kernel_choice() {
  static int choice = 0;
  return choice;
}

static int launch_n4_elem(const pddp_problem& p, const RiccatiArgs<float>& a,
                          const n4d::GenArgs<float>& gen, hipStream_t st,
                          int overlap = -1) {
  if (p.model != PDDP_MODEL_CARTPOLE ||
      p.encoding != PDDP_ENC_IGNORE_UNCERTAINTY || a.u_min == nullptr ||
      a.u_max == nullptr || a.branch != PDDP_BRANCH_EIG || a.N < 1)
    return PDDP_E_UNSUPPORTED;
  const ProblemT<float> P = convert_problem<float>(p);
  constexpr int kPer = n4e::kWaves * n4e::kTrajW;  // trajectories / workgroup
  const dim3 grid((a.B + kPer - 1) / kPer);
  // the generator on a wavefront of its own while one workgroup per CU holds
  // the batch; beyond that the SIMDs have other wavefronts to issue from and
  // the smaller LDS footprint (more workgroups per CU) counts
  const bool ovl = overlap < 0 ? grid.x <= 256u : overlap != 0;
  constexpr unsigned kSparse = 0b11001u;  // CartpoleCost: {x, sin, cos}
  const bool sparse =
      (live_mask(p.Q, ModelDims<PDDP_MODEL_CARTPOLE>::na) & ~kSparse) == 0;
#define PDDP_ELEM_GO(QMV, OV)                                                 \
  do {                                                                        \
    auto kern = n4e::riccati_n4_elem_kernel<QMV, OV>;                         \
    const size_t lds = (size_t)n4e::kWaves * sizeof(float) *                  \
                       (OV ? n4e::kPairLdsOvl : n4e::kPairLdsInl);            \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, dim3((OV ? 2 : 1) * n4e::kWaves * kWave), lds,    \
                st, a, gen, P);                                               \
  } while (0)
  constexpr unsigned kFull = kFullMask<PDDP_MODEL_CARTPOLE>;
  if (sparse) { if (ovl) PDDP_ELEM_GO(kSparse, true); else PDDP_ELEM_GO(kSparse, false); }
  else { if (ovl) PDDP_ELEM_GO(kFull, true); else PDDP_ELEM_GO(kFull, false); }
#undef PDDP_ELEM_GO
  return launch_status();
}

static int launch_n4_elem_f64(const pddp_problem& p,
                              const RiccatiArgs<double>& a,
                              const n4d::GenArgs<double>& gen, hipStream_t st) {
  if (p.model != PDDP_MODEL_CARTPOLE ||
      p.encoding != PDDP_ENC_IGNORE_UNCERTAINTY || a.u_min == nullptr ||
      a.u_max == nullptr || a.branch != PDDP_BRANCH_EIG || a.N < 1)
    return PDDP_E_UNSUPPORTED;
  const ProblemT<double> P = convert_problem<double>(p);
  constexpr int kPer = n4e::kWaves * n4e::kTrajW;
  const dim3 grid((a.B + kPer - 1) / kPer);
  constexpr unsigned kSparse = 0b11001u;
  constexpr unsigned kFull = kFullMask<PDDP_MODEL_CARTPOLE>;
  const bool sparse =
      (live_mask(p.Q, ModelDims<PDDP_MODEL_CARTPOLE>::na) & ~kSparse) == 0;
  const size_t lds =
      (size_t)n4e::kWaves * sizeof(double) * n4e::kPairLdsInl;  // 106 KB
#define PDDP_ELEM64_GO(QMV)                                                   \
  do {                                                                        \
    auto kern = n4e::riccati_n4_elem_f64_kernel<QMV>;                         \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, dim3(n4e::kWaves * kWave), lds, st, a, gen, P);   \
  } while (0)
  if (sparse) PDDP_ELEM64_GO(kSparse); else PDDP_ELEM64_GO(kFull);
#undef PDDP_ELEM64_GO
  return launch_status();
}

}  // namespace pddp
