// riccati_n4_qpipe.hpp - the n = 4, m = 1 bounded eig-clamp sweep (branch B,
// ilqr.py:629-672) on the quad mapping (riccati_n4_quad.hpp: four lanes per
// trajectory, sixteen trajectories per wavefront) with the step spread over
// THREE wavefronts of a workgroup, each on a SIMD of its own:
//
//   P  producer: the record stream.  Four full-wave LDS-DMA instructions per
//      step into the ring, R steps ahead; an LDS-DMA parks the issuing wave
//      for ~90 cycles - 28 % of the one-wave quad kernel's cycles - and this
//      wave has nothing else to do.
//   M  matrices: with the BoxQP result of step t+1 it forms V_{t+1}, V_z,t+1
//      (rank-one update), stores the gains of step t+1, computes the 4x4
//      products of step t and the three coefficients (A0, g, B0) through
//      which step t-1's action scalars see those products.
//   Q  scalars: Quu_t = A0 + c_{t+1} g^2, Qu_t = B0 + g w_{t+1} from its own
//      previous result, the closed-form BoxQP (riccati_n4.hpp QpClosed; the
//      reference's loop wave-cooperatively, riccati_n4_quad.hpp boxqp1_wave),
//      (k, s, Quu, Qu) back through LDS.  It never touches V.
//
// The algebra is riccati_n4_pipe.hpp's: K_t = -s_t Quz_t (s_t = 0 on a clamped
// step), V_t = sym(Qzz_t) + c_t Quz_t^T Quz_t with c_t = s_t (s_t Quu_t - 2),
// V_z,t = Qz_t + Quz_t w_t with w_t = k_t - s_t (Qu_t + Quu_t k_t).  One
// s_barrier per step.  Why three waves: at B = 4096 there are 256 workgroups,
// one per CU, and a wavefront that has a SIMD to itself pays ~4.1 cycles of
// issue per instruction of any kind - the step is as long as its busiest
// wave's instruction stream: ~241 + 4 DMA in one wave, ~130 /
// ~120 / 4 DMA here.
#pragma once

#include "riccati_n4_quad.hpp"
#include "riccati_n4_split.hpp"

namespace pddp {

namespace n4q {

constexpr int kQpThreads = 3 * kWave;

#ifdef PDDP_QP_STATS
// debug builds: cycles each role spends waiting at the step barrier
// (g_quad_stats[role]) and in total ([3 + role]); tools/qpipe_wait.py
#define PDDP_QPW_DECL unsigned long long wait_acc = 0; const long long t_begin = clock64();
#define PDDP_QPW_PLAIN() do { const long long t0_ = clock64(); asm volatile("s_barrier" ::: "memory"); wait_acc += (unsigned long long)(clock64() - t0_); } while (0)
#define PDDP_QPW_PUBLISH() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t0_ = clock64(); asm volatile("s_barrier" ::: "memory"); wait_acc += (unsigned long long)(clock64() - t0_); } while (0)
#define PDDP_QPW_END(ROLE) do { if (lane == 0) { atomicAdd(&g_quad_stats[ROLE], wait_acc); atomicAdd(&g_quad_stats[3 + ROLE], (unsigned long long)(clock64() - t_begin)); } } while (0)
#else
#define PDDP_QPW_DECL
#define PDDP_QPW_PLAIN() plain_barrier()
#define PDDP_QPW_PUBLISH() n4::lds_publish_barrier()
#define PDDP_QPW_END(ROLE)
#endif
PDDP_DEV void plain_barrier() { asm volatile("s_barrier" ::: "memory"); }

// c and w of the rank-one value update (role Q; role M takes them from LDS)
template <typename T>
PDDP_DEV void rank_one_coeffs(T k, T s, T Quu, T Qu, T& c, T& w) {
  c = mul_nc(s, fma_(s, Quu, T(-2)));
  w = fma_(-s, fma_(Quu, k, Qu), k);
}

// Qzz[q, :] is computed next to Qzz[:, q] (the same products in the same order
// as the partner lane), so that 0.5 (Q + Q^T) - and with it V - is symmetric
// to the last bit.  (A twin without the mirror products - 20 instructions less
// on role M - measured the same, 40.2 against 41.3 us: a step is as long as
// the dependent chain through role Q's BoxQP; removed in round 3.)
template <typename T, bool FAST, int R>
__global__ __launch_bounds__(kQpThreads) void riccati_n4_qpipe_kernel(
    RiccatiArgs<T> a) {
  using G = QuadGeom<T>;
  constexpr int NI = G::NI, RPI = G::RPI, CH = G::CH, CB = G::CB;
  constexpr int kTraj = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ __attribute__((aligned(16))) T xq[2][kTraj][4];  // Q -> M
  __shared__ __attribute__((aligned(16))) T xm[2][kTraj][4];  // M -> Q
  __shared__ T ls_tail[n4::kLsSteps];
  T* ring = reinterpret_cast<T*>(smem_raw);

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  for (int i = threadIdx.x; i < n4::kLsSteps; i += kQpThreads)
    ls_tail[i] = (T)n4::kLs.v[i];

  const int q = lane & 3, tr = lane >> 2;
  const int N = a.N;
  const int b0 = blockIdx.x * kTraj;
  const int b = b0 + tr;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  // (identical in the three waves: they own the same sixteen trajectories)
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  if (!__any(counted)) return;
  const int rbase = (tr / RPI) * G::GS + (tr % RPI) * kRec;
  PDDP_QPW_DECL

  if (role == 2) {
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
#pragma unroll
    for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
    n4::wait_vmcnt<0>();
    __syncthreads();  // (1) ring filled; Q published "step N"
    plain_barrier();  // (2) M published step N - 1's coefficients
    int t = N - 1;
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (t < 0) break;
        // phase t has begun: the slot of step t + 1 is dead (M used it for
        // its products in phase t + 1 and for the action side in phase t + 2;
        // Q read its U in phase t + 1)
        if (t != N - 1) dma((s + R - 1) % R, t + 1 - R);
        // before the barrier that ends this phase: records down to t - 3
        // (phase t - 1 works on t - 1 / t - 2 and gathers t - 2 / t - 3 for
        // the phase after it) have landed once at most R - 4 younger groups
        // of NI DMAs are outstanding
        n4::wait_vmcnt<(R - 4) * NI>();
        PDDP_QPW_PLAIN();  // (no vmcnt(0): the younger DMAs stay in flight)
        --t;
      }
    }
    n4::wait_vmcnt<0>();
    PDDP_QPW_END(2);
    return;
  }

  const T reg = (T)a.reg[bc];

  if (role == 1) {
    // =================================================================== Q
    const T umin = a.u_min[0], umax = a.u_max[0];
    T kprev = T(0), cp = T(0), wp = T(0);  // "step N": K = 0, c = w = 0
    int status = PDDP_BWD_OK;
    if (q == 0) {
      T* pq = &xq[1][tr][0];  // read by role M in the first phase (t = N - 1)
      pq[0] = T(0); pq[1] = T(0); pq[2] = T(0); pq[3] = T(0);
    }
    // (the first phase reads parity (N - 1 + 1) & 1 = N & 1; publish both)
    if (q == 1) {
      T* pq = &xq[0][tr][0];
      pq[0] = T(0); pq[1] = T(0); pq[2] = T(0); pq[3] = T(0);
    }
    __syncthreads();  // (1)
    plain_barrier();  // (2) M published step N - 1's coefficients
    int t = N - 1;
    auto phase = [&](const int s) {
      const bool alive = counted & (status == PDDP_BWD_OK);
      const T* pm = &xm[t & 1][tr][0];
      const T A0 = pm[0], g = pm[1], B0 = pm[2];
      const T Un = ring[s * G::SLOT + rbase + 46];
      const T Quu = fma_(cp, mul_nc(g, g), A0);
      const T Qu = fma_(g, wp, B0);
      int st = PDDP_BWD_OK;
      if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
      const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
      const T qp_Q = e + reg;                     // ilqr.py:634
      n4::QpClosed<T, FAST> qc;
      const T lo_b = umin - Un, hi_b = umax - Un;
      qc.solve(kprev, qp_Q, Qu, lo_b, hi_b);
      T kt = qc.x;
      bool Kzero = !qc.free_, fail = qc.fail;
      const bool slow = qc.slow && alive;
      if (__builtin_expect(__any(slow), 0)) {
        // rare: the reference's loop, one slow trajectory at a time on the
        // whole wavefront
        unsigned long long todo = __ballot(slow && q == 0);
        while (todo != 0) {
          const int src = __builtin_ctzll(todo);
          todo &= todo - 1;
          const n4::SlowQpOut<T> o = boxqp1_wave<T, FAST>(
              __shfl(kprev, src), __shfl(qp_Q, src), __shfl(Qu, src),
              __shfl(lo_b, src), __shfl(hi_b, src), ls_tail, lane);
          const bool mine = (lane >> 2) == (src >> 2);
          kt = mine ? o.x : kt;
          Kzero = mine ? ((o.result_free & 1) == 0) : Kzero;
          fail = mine ? (o.result_free < 2) : fail;
        }
      }
      // K = -s Quz: 1 / Q through v_rcp (FAST) or an IEEE division
      T sK;
      if constexpr (FAST) sK = qc.inv;
      else sK = T(1) / qp_Q;
      sK = Kzero ? T(0) : sK;
      const int stt = st != PDDP_BWD_OK ? st : (fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
      // the rank-one value update of this step, for role M and for the next
      // phase's affine update
      T c, w;
      rank_one_coeffs(kt, sK, Quu, Qu, c, w);
      if (q == 0) {
        T* pq = &xq[t & 1][tr][0];
        pq[0] = kt; pq[1] = sK; pq[2] = c; pq[3] = w;
      }
      status = (alive & (stt != PDDP_BWD_OK)) ? stt : status;
      kprev = kt; cp = c; wp = w;
      PDDP_QPW_PUBLISH();
    };
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (t < 0) break;
        phase(s);
        --t;
      }
    }
    if (counted && q == 0) a.status[bc] = status;
    PDDP_QPW_END(1);
    return;
  }

  // ===================================================================== M
  const int oq = rbase + q, or4 = rbase + 4 * q;
  struct Words {            // record t: the products' operands
    T F0, F1, F2, F3;       // F[k][q]
    T Lc0, Lc1, Lc2, Lc3;   // Lzz[i][q]
    T Lr0, Lr1, Lr2, Lr3;   // Lzz[q][i]
    T f0, f1, f2, f3;       // F_u
    T Luz, Lz;
  };
  struct Act {              // record t - 1: the action side
    T f0, f1, f2, f3, fq, Luu, Lu;
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
    w.Luz = rc[oq + 36];
    w.Lz = rc[oq + 40];
    return w;
  };
  auto gather_act = [&](int slot) {
    const T* rc = ring + slot * G::SLOT;
    Act x;
    x.f0 = rc[rbase + 32]; x.f1 = rc[rbase + 33]; x.f2 = rc[rbase + 34];
    x.f3 = rc[rbase + 35];
    x.fq = rc[oq + 32];
    x.Luu = rc[rbase + 44]; x.Lu = rc[rbase + 45];
    return x;
  };
  // products of the step after the current one: S = Qzz[:, q] + Qzz[q, :]
  // (twice the symmetrised column), Quz[q], Qz[q]
  T S0, S1, S2, S3, Quz = T(0), Qz;
  {
    // "products of step N": the terminal value function (ilqr.py:581-583)
    const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
    S0 = term[16 + 0 + q] + term[16 + 4 * q + 0];
    S1 = term[16 + 4 + q] + term[16 + 4 * q + 1];
    S2 = term[16 + 8 + q] + term[16 + 4 * q + 2];
    S3 = term[16 + 12 + q] + term[16 + 4 * q + 3];
    Qz = term[40 + q];
  }
  // (A0, g, B0) of the step whose action side is `x`, from (S, Quz, Qz)
  auto coeffs = [&](const Act& x, int parity) {
    T h = mul_nc(S0, x.f0);  // (2 sym(Qzz) f)[q]
    h = fma_(S1, x.f1, h);
    h = fma_(S2, x.f2, h);
    h = fma_(S3, x.f3, h);
    const T A0 = x.Luu + quad_sum(mul_nc(T(0.5), mul_nc(x.fq, h)));
    const T g = quad_sum(mul_nc(x.fq, Quz));
    const T B0 = x.Lu + quad_sum(mul_nc(x.fq, Qz));
    if (q == 0) {
      T* pm = &xm[parity][tr][0];
      pm[0] = A0; pm[1] = g; pm[2] = B0;
    }
  };
  __syncthreads();  // (1) P filled the ring; Q published "step N"
  {
    const Act x0 = gather_act(0);  // record N - 1
    coeffs(x0, (N - 1) & 1);
  }
  n4::lds_publish_barrier();  // (2)
  char* gains_w =
      reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
  // byte offset of K[q] of step t + 1 (the step whose BoxQP result arrives in
  // phase t); k sits one scalar before K[0]
  uint32_t gout_off = (uint32_t)(
      ((bc - b0) * N * kGain + N * kGain + 1 + q) * (int)sizeof(T));
  T Vc0 = T(0), Vc1 = T(0), Vc2 = T(0), Vc3 = T(0), vz = T(0);
  int t = N - 1;
  // finish step t + 1 with its BoxQP result: gains, value function
  auto finish_prev = [&](T kt, T sK, T c, T w, bool first) {
    if (!first) {
      T* dst = reinterpret_cast<T*>(gains_w + gout_off);
      if (exists) {
        *dst = -(sK * Quz);
        if (q == 0) dst[-1] = kt;
      }
    }
    gout_off -= (uint32_t)(kGain * sizeof(T));
    // V = 0.5 S + c Quz_i Quz_q (the product commutes: exactly symmetric)
    Vc0 = fma_(T(0.5), S0, mul_nc(c, mul_nc(qb<0>(Quz), Quz)));
    Vc1 = fma_(T(0.5), S1, mul_nc(c, mul_nc(qb<1>(Quz), Quz)));
    Vc2 = fma_(T(0.5), S2, mul_nc(c, mul_nc(qb<2>(Quz), Quz)));
    Vc3 = fma_(T(0.5), S3, mul_nc(c, mul_nc(qb<3>(Quz), Quz)));
    vz = fma_(Quz, w, Qz);
  };
  // the words of a phase are gathered one phase ahead (records t - 1 and
  // t - 2 have landed when phase t starts: role P), so their LDS latency
  // overlaps this phase's products
  auto phase = [&](const Words& w, const Act& x, Words& wn, Act& xn,
                   const int s) {
    const bool first = (t == N - 1);
    // role Q's result first: LDS returns in order, and the chain starts here
    const T* pq = &xq[(t + 1) & 1][tr][0];
    const T kt = pq[0], sK = pq[1], c = pq[2], w_ = pq[3];
    asm volatile("" ::: "memory");  // (keep the gathers behind it)
    wn = gather((s + 1) % R);                        // record t - 1
    xn = gather_act((s + 2) % R);                    // record t - 2
    finish_prev(kt, sK, c, w_, first);
    // ---- T[:, q] = V F[:, q]
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
    // Q_uz[q] = L_uz[q] + sum_k f[k] T[k][q]
    T Quz_n = fma_(w.f0, T0, w.Luz);
    Quz_n = fma_(w.f1, T1, Quz_n);
    Quz_n = fma_(w.f2, T2, Quz_n);
    Quz_n = fma_(w.f3, T3, Quz_n);
    // Qzz[:, q] and its mirror Qzz[q, :]
    T C0 = w.Lc0, C1 = w.Lc1, C2 = w.Lc2, C3 = w.Lc3;
    T R0 = w.Lr0, R1 = w.Lr1, R2 = w.Lr2, R3 = w.Lr3;
    T Qz_n = w.Lz;
    if constexpr (sizeof(T) == 4) {
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F0, T0);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F1, T1);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F2, T2);
      dpp_fmac4_lanes<false>(C0, C1, C2, C3, w.F3, T3);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T0, w.F0);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T1, w.F1);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T2, w.F2);
      dpp_fmac4_lanes<true>(R0, R1, R2, R3, T3, w.F3);
      dpp_dot4(Qz_n, vz, w.F0, w.F1, w.F2, w.F3);
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
      Qz_n = fma_(w.F0, qb<0>(vz), Qz_n);
      Qz_n = fma_(w.F1, qb<1>(vz), Qz_n);
      Qz_n = fma_(w.F2, qb<2>(vz), Qz_n);
      Qz_n = fma_(w.F3, qb<3>(vz), Qz_n);
    }
    S0 = C0 + R0; S1 = C1 + R1; S2 = C2 + R2; S3 = C3 + R3;
    Quz = Quz_n;
    Qz = Qz_n;
    // ---- what role Q needs for step t - 1
    coeffs(x, (t - 1) & 1);
    PDDP_QPW_PUBLISH();
  };
  Words wa = gather(0), wb = wa;                     // record N - 1
  Act xa = gather_act(1 % R), xb = xa;               // record N - 2
  static_assert(R % 2 == 0, "two word sets alternate over the ring");
  while (t >= 0) {
#pragma unroll
    for (int s = 0; s < R; s += 2) {
      if (t < 0) break;
      phase(wa, xa, wb, xb, s);
      --t;
      if (t < 0) break;
      phase(wb, xb, wa, xa, s + 1);
      --t;
    }
  }
  // step 0's gains: its BoxQP result was published by the last barrier
  {
    const T* pq = &xq[0][tr][0];
    finish_prev(pq[0], pq[1], pq[2], pq[3], false);
  }
  PDDP_QPW_END(0);
}

}  // namespace n4q

// bounded eig-clamp branch only; 16 trajectories per workgroup of three waves
template <typename T>
static int launch_n4_qpipe(const RiccatiArgs<T>& a, hipStream_t st,
                           bool fast_math) {
  constexpr int R = 8;
  using G = n4q::QuadGeom<T>;
  if (a.u_min == nullptr || a.branch != PDDP_BRANCH_EIG)
    return PDDP_E_UNSUPPORTED;
  const size_t lds = (size_t)R * G::SLOT * sizeof(T);
  const dim3 grid((a.B + 15) / 16), block(n4q::kQpThreads);
#define PDDP_QP_GO(F)                                                         \
  do {                                                                        \
    auto kern = n4q::riccati_n4_qpipe_kernel<T, F, R>;                        \
    const hipError_t e = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,        \
        (int)lds);                                                            \
    if (e != hipSuccess) return (int)e;                                       \
    PDDP_LAUNCH(kern, grid, block, lds, st, a);                               \
  } while (0)
  if (fast_math && sizeof(T) == 4) PDDP_QP_GO(true);
  else PDDP_QP_GO(false);
#undef PDDP_QP_GO
  return launch_status();
}

}  // namespace pddp
