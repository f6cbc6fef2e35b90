// riccati_n4_pipe.hpp - the n = 4, m = 1 bounded eig-clamp sweep (branch B,
// ilqr.py:629-672) with the BoxQP chain DECOUPLED from the value update.
//
// riccati_n4_split.hpp showed what bounds the sweep at one or two waves per
// SIMD: the dependent chain V -> f^T V f -> BoxQP -> K -> V' of every step,
// at ~9 cycles per dependent operation.  The value update is rank one,
//   V_t   = Qzz_t + c_t Quz_t^T Quz_t,        c_t = sK_t^2 Quu_t - 2 sK_t
//   V_z,t = Qz_t  + Quz_t^T w_t,               w_t = k_t - sK_t (Qu_t + Quu_t k_t)
// (K_t = -sK_t Quz_t, or c_t = 0, w_t = k_t for a clamped step: K_t = 0), so
// the two scalars the NEXT BoxQP needs are affine in (c_t, w_t):
//   Quu_{t-1} = A0 + c_t g^2,   A0 = Luu_{t-1} + f^T Qzz_t f,  g = f . Quz_t
//   Qu_{t-1}  = B0 + g w_t,     B0 = Lu_{t-1}  + f . Qz_t       (f = F_u,t-1)
// and A0, g, B0 only need the PRODUCTS of step t, not its BoxQP.  Two roles:
//   Q  a purely scalar recurrence: (A0, g, B0) of step t from LDS, the affine
//      update with its own previous result, the closed-form BoxQP
//      (riccati_n4.hpp QpClosed, the reference's loop out of line), result to
//      LDS.  It never touches V.
//   M  with the BoxQP result of step t+1: gains of step t+1, V_{t+1}, the 4x4
//      products of step t, and (A0, g, B0) for step t-1.
// One s_barrier per step; both chains are ~half the old one and run
// concurrently.  The association of the sums differs from the other variants
// (f^T V f is assembled from f^T Qzz f and g): results agree to rounding, not
// bit for bit.
#pragma once

#include "riccati_n4_split.hpp"

namespace pddp {
namespace n4 {

#ifdef PDDP_PIPE_TIMING
// cycles spent waiting at the step barrier, per role (debug builds only)
__device__ unsigned long long g_pipe_wait[4];
PDDP_DEV void timed_barrier(unsigned long long& acc) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t0 = clock64();
  asm volatile("s_barrier" ::: "memory");
  acc += (unsigned long long)(clock64() - t0);
}
#define PDDP_PIPE_BARRIER() timed_barrier(wait_acc)
#else
#define PDDP_PIPE_BARRIER() lds_publish_barrier()
#endif

template <typename T, bool FAST>
__global__ __launch_bounds__(kSplitThreads) void riccati_n4_pipe_kernel(
    RiccatiArgs<T> a) {
  constexpr int CB = 16;
  constexpr int CH = kRec * (int)sizeof(T) / CB;
  constexpr int NI = (4 * CH + kWave - 1) / kWave;
  constexpr int kSlot = NI * kWave * CB / (int)sizeof(T);
  constexpr int R = kRing;
  __shared__ __attribute__((aligned(16))) T ring[R][kSlot];
  __shared__ __attribute__((aligned(16))) T xq[2][kWave][4];  // Q -> M
  __shared__ __attribute__((aligned(16))) T xm[2][kWave][4];  // M -> Q
  __shared__ T ls_tail[kLsSteps];

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  for (int q = threadIdx.x; q < kLsSteps; q += kSplitThreads)
    ls_tail[q] = (T)kLs.v[q];
  const T lstep0 = (T)kLs.v[lane & 15];

  const int grp = lane >> 4, l = lane & 15, i = l >> 2, j = l & 3;
  const int N = a.N;
  const int b0 = blockIdx.x * 4;
  const int b = b0 + grp;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  const T reg = (T)a.reg[bc];
  const T umin = a.u_min[0], umax = a.u_max[0];
  const int gb = grp * kRec;
  const T kNaN = (T)__builtin_nan("");
#ifdef PDDP_PIPE_TIMING
  unsigned long long wait_acc = 0;
  const long long t_begin = clock64();
#endif

  if (role == 0) {
    // =================================================================== Q
    const int oU = gb + 46;
    T kprev = T(0), sKp = kNaN, Quup = T(0), Qup = T(0);  // "step N": K = 0
    int status = PDDP_BWD_OK;
    {
      T* pq = &xq[1][lane][0];  // read by role M in the first phase
      pq[0] = kprev; pq[1] = sKp; pq[2] = Quup; pq[3] = Qup;
    }
    __syncthreads();  // ring, step-size table, first coefficients
    int t = N - 1;
    auto phase = [&](const int s) {
      const bool alive = counted & (status == PDDP_BWD_OK);
      const T* pm = &xm[s & 1][lane][0];
      const T A0 = pm[0], g = pm[1], B0 = pm[2];
      const T Un = ring[s][oU];
      // the rank-one value update of the previous step, seen through f
      const bool Kz = (sKp != sKp);
      T sE;  // K = -sE Quz
      if constexpr (FAST) sE = Kz ? T(0) : sKp;
      else sE = Kz ? T(0) : div_<false>(div_<false>(T(1), sKp), sKp);
      const T c = sE * (sE * Quup - T(2));
      const T w = kprev - sE * (Qup + Quup * kprev);
      const T Quu = A0 + c * (g * g);
      const T Qu = B0 + g * w;
      int st = PDDP_BWD_OK;
      if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
      const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
      const T qp_Q = e + reg;                     // ilqr.py:634
      QpClosed<T, FAST> qc;
      qc.solve(kprev, qp_Q, Qu, umin - Un, umax - Un);
      T kt = qc.x;
      T sK = qc.free_ ? (FAST ? qc.inv : qc.U) : kNaN;
      int stt = qc.fail ? (int)PDDP_BWD_BOXQP_FAILED : st;
      if (__builtin_amdgcn_ballot_w64(qc.slow & alive) != 0) {
        // rare: the reference's loop as written
        const SlowQpOut<T> o = boxqp1_outlined<T, FAST>(
            alive ? kprev : T(0), alive ? qp_Q : T(1), alive ? Qu : T(0),
            umin - (alive ? Un : T(0)), umax - (alive ? Un : T(0)), lstep0,
            ls_tail, lane);
        kt = o.x;
        sK = (o.result_free & 1) ? (FAST ? qc.inv : o.U) : kNaN;
        stt = (o.result_free < 2) ? (int)PDDP_BWD_BOXQP_FAILED : st;
      }
      {
        T* pq = &xq[s & 1][lane][0];
        pq[0] = kt; pq[1] = sK; pq[2] = Quu; pq[3] = Qu;
      }
      status = (alive & (stt != PDDP_BWD_OK)) ? stt : status;
      kprev = kt; sKp = sK; Quup = Quu; Qup = Qu;
      PDDP_PIPE_BARRIER();
    };
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (t < 0) break;
        phase(s);
        --t;
      }
    }
    if (counted && l == 0) a.status[bc] = status;
#ifdef PDDP_PIPE_TIMING
    if (lane == 0) {
      atomicAdd(&g_pipe_wait[0], wait_acc);
      atomicAdd(&g_pipe_wait[2], (unsigned long long)(clock64() - t_begin));
    }
#endif
  } else {
    // =================================================================== M
    const char* rec_w = reinterpret_cast<const char*>(
        a.rec + (size_t)b0 * (size_t)(N + 1) * kRec);
    uint32_t src_off[NI];
#pragma unroll
    for (int r = 0; r < NI; ++r) {
      int q = lane + kWave * r;
      q = q < 4 * CH ? q : q - 4 * CH;  // padding lanes: any valid chunk
      const int tg = q / CH, c = q - tg * CH;
      int tb = b0 + tg;
      tb = tb < a.B ? tb : a.B - 1;
      src_off[r] =
          (uint32_t)((tb - b0) * (N + 1) * kRec * (int)sizeof(T) + c * CB);
    }
    auto dma = [&](int slot, int t) {
      const int tt = t < 0 ? 0 : t;
      const uint32_t toff = (uint32_t)tt * (uint32_t)(kRec * sizeof(T));
#pragma unroll
      for (int r = 0; r < NI; ++r)
        lds_dma16(rec_w, src_off[r] + toff,
                  __builtin_amdgcn_readfirstlane(lds_addr(&ring[slot][0])) +
                      r * kWave * CB);
    };
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
    const int oLuu = gb + 44, oLu = gb + 45;
    const int tr_addr = ((lane & 48) | (j * 4 + i)) * 4;  // lane (j, i)
    struct Words {
      T Fs0, Fs1, Fs2, Fs3, Fq0, Fq1, Fq2, Fq3, Ft, Lzz, fr, fc, Luzr, Lzr, Luu, Lu;
    };
    auto gather = [&](int slot) {
      const T* rc = &ring[slot][0];
      Words w;
      w.Fs0 = rc[oFs[0]]; w.Fs1 = rc[oFs[1]]; w.Fs2 = rc[oFs[2]]; w.Fs3 = rc[oFs[3]];
      w.Fq0 = rc[oFq[0]]; w.Fq1 = rc[oFq[1]]; w.Fq2 = rc[oFq[2]]; w.Fq3 = rc[oFq[3]];
      w.Ft = rc[oFt]; w.Lzz = rc[oLzz]; w.fr = rc[oFur]; w.fc = rc[oFuc];
      w.Luzr = rc[oLuzr]; w.Lzr = rc[oLzr]; w.Luu = rc[oLuu]; w.Lu = rc[oLu];
      return w;
    };
    // coefficients of the step whose record words are `wn`, from the products
    // (Qzzs, Quzc, Qzc) of the step after it
    auto coeffs = [&](const Words& wn, T Qzzs, T Quzc, T Qzc, int parity) {
      const T fQ = dot_rows(wn.fr, Qzzs);  // (f^T Qzz)[j], column form
      T* pm = &xm[parity][lane][0];
      pm[0] = wn.Luu + dot_cols(fQ, wn.fc);
      pm[1] = dot_cols(Quzc, wn.fc);
      pm[2] = wn.Lu + dot_cols(Qzc, wn.fc);
    };
#pragma unroll
    for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
    wait_vmcnt<0>();
    // "products of step N": the terminal value function (ilqr.py:581-583)
    const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
    MHalf<T, false> mh;
    mh.Qzzs = term[16 + i * 4 + j];
    mh.Quzr = T(0);
    mh.Quzc = T(0);
    mh.Qzc = term[40 + j];
    mh.Quzgr = T(0);
    mh.Quzgc = T(0);
    Words wa = gather(0), wb = wa;
    coeffs(wa, mh.Qzzs, mh.Quzc, mh.Qzc, 0);
    __syncthreads();
    char* gains_w =
        reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
    // byte offset of this lane's k / K word of step t + 1 (the step whose
    // BoxQP result arrives in phase t)
    uint32_t gout_off = (uint32_t)(
        ((bc - b0) * N * kGain + N * kGain + ((l < 4) ? 1 + l : 0)) *
        (int)sizeof(T));
    T V = T(0), Vzc = T(0);
    int t = N - 1;
    // finish step t + 1 with its BoxQP result: gains, value function
    auto finish_prev = [&](int parity, bool first) {
      const T* pq = &xq[parity][lane][0];
      QHalf<T> q{pq[0], pq[1], pq[2], pq[3]};
      T Kc;
      split_tail<T, false, FAST>(q, mh, V, Vzc, Kc);
      if (!first) {
        const T val = (l < 4) ? Kc : q.kt;
        T* dst = reinterpret_cast<T*>(gains_w + gout_off);
        if (exists && l < 5) *dst = val;
      }
      gout_off -= (uint32_t)(kGain * sizeof(T));
    };
    auto phase = [&](const Words& w, Words& wn, const int s) {
      const bool first = (t == N - 1);
      finish_prev((s + 1) & 1, first);
      // the slot of step t + 1 is free now (role Q read its U before the
      // barrier that let us in): refill it, R steps further down the sweep
      if (!first) dma((s + R - 1) % R, t + 1 - R);
      // DMA(t-1) has landed once at most (R-2) younger {store, DMA} pairs are
      // outstanding
      wait_vmcnt<(R - 2) * (1 + NI)>();
      wn = gather((s + 1) % R);
      // ---- the 4x4 products of step t (ilqr.py:489-526)
      T A = w.Fs0 * V;
      A += w.Fs1 * from_row_plus<1>(V);
      A += w.Fs2 * from_row_plus<2>(V);
      A += w.Fs3 * from_row_plus<3>(V);
      T Qzz = w.Lzz + A * w.Fq0;
      Qzz += from_col_plus<1>(A) * w.Fq1;
      Qzz += from_col_plus<2>(A) * w.Fq2;
      Qzz += from_col_plus<3>(A) * w.Fq3;
      const T Quzr = w.Luzr + dot_cols(A, w.fc);
      const T Qzr = w.Lzr + dot_cols(w.Ft, Vzc);
      const T QzzT = bperm(tr_addr, Qzz);
      mh.Quzr = Quzr;
      mh.Quzc = bperm(tr_addr, Quzr);
      mh.Qzc = bperm(tr_addr, Qzr);
      mh.Qzzs = mul_nc(T(0.5), Qzz + QzzT);
      // ---- what role Q needs for step t - 1
      coeffs(wn, mh.Qzzs, mh.Quzc, mh.Qzc, (s + 1) & 1);
      PDDP_PIPE_BARRIER();
    };
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; s += 2) {
        if (t < 0) break;
        phase(wa, wb, s);
        --t;
        if (t < 0) break;
        phase(wb, wa, s + 1);
        --t;
      }
    }
    // step 0's gains: its BoxQP result was published by the last barrier
    finish_prev((N - 1) & 1, false);
    wait_vmcnt<0>();
#ifdef PDDP_PIPE_TIMING
    if (lane == 0) {
      atomicAdd(&g_pipe_wait[1], wait_acc);
      atomicAdd(&g_pipe_wait[3], (unsigned long long)(clock64() - t_begin));
    }
#endif
  }
}

}  // namespace n4

template <typename T>
static int launch_n4_pipe(const RiccatiArgs<T>& a, hipStream_t st,
                          bool fast_math) {
  const dim3 grid((a.B + 3) / 4), block(n4::kSplitThreads);
  if (fast_math)
    PDDP_LAUNCH((n4::riccati_n4_pipe_kernel<T, true>), grid, block, 0, st, a);
  else
    PDDP_LAUNCH((n4::riccati_n4_pipe_kernel<T, false>), grid, block, 0, st, a);
  return launch_status();
}

}  // namespace pddp
