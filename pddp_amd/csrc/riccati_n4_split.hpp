// riccati_n4_split.hpp - the n = 4, m = 1 bounded sweep with the step split
// over TWO wavefronts (role kernel of riccati_n4.hpp; same mapping: 16 lanes
// per trajectory, four trajectories per wavefront).
//
// Why: at B = 4096 the one-wavefront kernel runs one wave per SIMD, and a lone
// wave issues one instruction of ANY kind per ~4 cycles, so the ~180
// instructions of a step cost ~1200 cycles although they form two nearly
// independent chains.  Here a workgroup is two wavefronts that both carry the
// value function (V_zz, V_z: bit-identical copies) of the same four
// trajectories:
//   * role Q  : f^T V f, Q_u, the regularised Q_uu and the scalar BoxQP in
//               closed form (QpClosed; the loop out of line) - the chain of
//               ilqr.py:590-612 / 629-657 and utils/constraint.py:150-266;
//   * role M  : the 4x4 products A = F^T V, Q_zz = L_zz + A F, Q_uz, Q_z and
//               their transposes (ilqr.py:489-526), the record DMA ring and the
//               gain stores.
// Each role writes its results (4 .. 6 words per lane) to LDS, ONE s_barrier
// per step, both read both halves and run the same value update
// (ilqr.py:664-672) redundantly - written with explicit fma / unfused products
// so that the two copies of V stay bit-identical whatever the compiler fuses
// elsewhere.  Exchange buffers alternate by step parity, so a wave can never
// overwrite words its partner has not consumed (the next barrier lies in
// between).
#pragma once

#include "riccati_n4.hpp"

namespace pddp {
namespace n4 {

constexpr int kSplitThreads = 2 * kWave;

typedef uint32_t Vec16 __attribute__((ext_vector_type(4)));
// Exchange reads are hand-issued: a compiler-visible LDS load makes the
// waitcnt pass drain vmcnt(0) first (it cannot tell the exchange buffers from
// the ring the record DMAs write), which would serialise every step behind the
// DMA issued one step earlier.  The values are defined once the matching
// lds_wait*() has run.
PDDP_DEV Vec16 lds_read16(uint32_t addr) {
  Vec16 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
PDDP_DEV uint32_t lds_read4(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
PDDP_DEV void lds_write4(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int NV>
PDDP_DEV void touch(Vec16 (&v)[NV]) {
#pragma unroll
  for (int c = 0; c < NV; ++c) asm volatile("" : "+v"(v[c]));
}
// all LDS traffic of this wave done (its exchange words are visible), then
// meet the partner.  No vmcnt wait: the record DMAs stay in flight.
PDDP_DEV void lds_publish_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
PDDP_DEV void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// what role Q hands over (identical in the 16 lanes of a group)
template <typename T>
struct QHalf {
  T kt, sK, Quu, Qu;  // sK: 1 / Q (FAST) or sqrt(Q); NaN = "K row is zero"
};
// what role M hands over (lane (i, j) of the group)
template <typename T, bool CHOL>
struct MHalf {
  T Qzzs, Quzr, Quzc, Qzc;
  T Quzgr, Quzgc;  // CHOL only: Q_uz with V + reg I (operand of the K solve)
};

// Value update of one step given both halves (ilqr.py:613-625 / 658-672).
// No implicit contraction: both roles must produce the same bits.
template <typename T, bool CHOL, bool FAST>
PDDP_DEV void split_tail(const QHalf<T>& q, const MHalf<T, CHOL>& mh, T& V,
                         T& Vzc, T& Kc_out) {
#pragma clang fp contract(off)
  const bool Kzero = (q.sK != q.sK);
  const T gr = CHOL ? mh.Quzgr : mh.Quzr;
  const T gc = CHOL ? mh.Quzgc : mh.Quzc;
  T Kr, Kc;
  if constexpr (FAST) {
    Kr = Kzero ? T(0) : -(gr * q.sK);
    Kc = Kzero ? T(0) : -(gc * q.sK);
  } else {
    Kr = Kzero ? T(0) : -div_<false>(div_<false>(gr, q.sK), q.sK);
    Kc = Kzero ? T(0) : -div_<false>(div_<false>(gc, q.sK), q.sK);
  }
  Kc_out = Kc;
  T v = fma_(Kc, q.Qu, mh.Qzc);
  v = fma_(mul_nc(Kc, q.Quu), q.kt, v);
  Vzc = fma_(mh.Quzc, q.kt, v);
  const T va = fma_(mul_nc(Kr, q.Quu), Kc, mh.Qzzs) +
               fma_(Kr, mh.Quzc, mul_nc(mh.Quzr, Kc));
  const T vb = fma_(mul_nc(Kc, q.Quu), Kr, mh.Qzzs) +
               fma_(Kc, mh.Quzr, mul_nc(mh.Quzc, Kr));
  V = T(0.5) * (va + vb);
}

template <typename T, bool CHOL, bool FAST>
__global__ __launch_bounds__(kSplitThreads) void riccati_n4_split_kernel(
    RiccatiArgs<T> a) {
  // Record DMAs are FULL-wave 16-byte instructions: the four records of a
  // wave are 48 (f32) / 96 (f64) chunks; the lanes past them re-load an earlier
  // chunk into the slot's padding (a slot is NI KiB).  No lane-dependent
  // branch around a DMA: the compiler merges such divergent calls into one
  // whose LDS base is a per-lane value, i.e. wrong data (seen with f64).
  // (global_load_lds_dwordx3 is no way out: its 12 bytes land at lane * 16.)
  constexpr int CB = 16;                            // bytes per chunk
  constexpr int CH = kRec * (int)sizeof(T) / CB;    // chunks per record
  constexpr int NI = (4 * CH + kWave - 1) / kWave;  // DMA instructions / step
  constexpr int kSlot = NI * kWave * CB / (int)sizeof(T);  // scalars per slot
  constexpr int R = kRing;
  constexpr int XQ = 4, XM = CHOL ? 6 : 4;          // words per lane and half
  __shared__ __attribute__((aligned(16))) T ring[R][kSlot];
  __shared__ __attribute__((aligned(16))) T xq[2][kWave][XQ];
  constexpr int XMP = XM + (XM & 2);                // padded to 16 B
  __shared__ __attribute__((aligned(16))) T xm[2][kWave][XMP];
  __shared__ uint32_t xf[2][kWave];  // step tag of the words in xm
  __shared__ T ls_tail[kLsSteps];    // T(0.6^n), read only past n = 31

  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  for (int q = threadIdx.x; q < kLsSteps; q += kSplitThreads)
    ls_tail[q] = (T)kLs.v[q];
  (&xf[0][0])[threadIdx.x] = 0xffffffffu;
  const T lstep0 = (T)kLs.v[lane & 15];

  const int grp = lane >> 4, l = lane & 15, i = l >> 2, j = l & 3;
  const int N = a.N;
  const int b0 = blockIdx.x * 4;
  const int b = b0 + grp;
  const bool exists = b < a.B;
  const int bc = exists ? b : a.B - 1;
  const bool counted = exists && (a.active == nullptr || a.active[bc] != 0);
  // (no early exit: both waves must meet at every barrier; a workgroup whose
  // trajectories are all inactive is rare and merely streams)

  const T reg = (T)a.reg[bc];
  const T umin = a.u_min[0], umax = a.u_max[0];

  // ---- terminal value function, in both roles
  const T* term = a.rec + ((size_t)bc * (size_t)(N + 1) + N) * kRec;
  T V = term[16 + i * 4 + j];
  T Vzc = term[40 + j];

  const int gb = grp * kRec;
  constexpr int QV = XQ * (int)sizeof(T) / 16, MV = XMP * (int)sizeof(T) / 16;
  auto unpack_q = [&](const Vec16 (&vq)[QV]) {
    T w[XQ];
    __builtin_memcpy(w, vq, sizeof(w));
    return QHalf<T>{w[0], w[1], w[2], w[3]};
  };
  auto unpack_m = [&](const Vec16 (&vm)[MV]) {
    T w[XMP];
    __builtin_memcpy(w, vm, sizeof(w));
    MHalf<T, CHOL> mh{w[0], w[1], w[2], w[3], T(0), T(0)};
    if constexpr (CHOL) { mh.Quzgr = w[4]; mh.Quzgc = w[5]; }
    return mh;
  };

  if (role == 0) {
    // =================================================================== Q
    const int oFur = gb + 32 + i, oFuc = gb + 32 + j;
    const int oLuu = gb + 44, oLu = gb + 45, oU = gb + 46;
    struct Words { T fr, fc, Luu, Lu, Un; };
    auto gather = [&](int slot) {
      const T* rc = &ring[slot][0];
      Words w;
      w.fr = rc[oFur]; w.fc = rc[oFuc];
      w.Luu = rc[oLuu]; w.Lu = rc[oLu]; w.Un = rc[oU];
      return w;
    };
    T kprev = T(0);
    int status = PDDP_BWD_OK;
    __syncthreads();  // ring filled by role M, ls_tail by everyone
    int t = N - 1;
    auto step = [&](const Words& w, Words& wn, const int s) {
      const bool alive = counted & (status == PDDP_BWD_OK);
      const T bTc = dot_rows(w.fr, V);
      const T Quu = w.Luu + dot_cols(bTc, w.fc);
      const T Qu = w.Lu + dot_cols(w.fc, Vzc);
      int st = PDDP_BWD_OK;
      T qp_Q;
      if constexpr (CHOL) {
        // second Q() with V + reg I                           (ilqr.py:590-592)
        const T Vr = (i == j) ? V + reg : V;
        const T bTrc = dot_rows(w.fr, Vr);
        qp_Q = w.Luu + dot_cols(bTrc, w.fc);
      } else {
        if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
        const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
        qp_Q = e + reg;                             // ilqr.py:634
      }
      QpClosed<T, FAST> qc;
      qc.solve(kprev, qp_Q, Qu, umin - w.Un, umax - w.Un);
      // results as register values (not flags): the rare loop call below then
      // merges through plain moves on its own path
      QHalf<T> q;
      q.kt = qc.x;
      q.sK = qc.free_ ? (FAST ? qc.inv : qc.U) : (T)__builtin_nan("");
      q.Quu = Quu;
      q.Qu = Qu;
      int stt = st != PDDP_BWD_OK ? st
                                   : (qc.fail ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
      if (__builtin_amdgcn_ballot_w64(qc.slow & alive) != 0) {
        // rare: the reference's loop as written
        const SlowQpOut<T> o = boxqp1_outlined<T, FAST>(
            alive ? kprev : T(0), alive ? qp_Q : T(1), alive ? Qu : T(0),
            umin - (alive ? w.Un : T(0)), umax - (alive ? w.Un : T(0)), lstep0,
            ls_tail, lane);
        q.kt = o.x;
        q.sK = (o.result_free & 1) ? (FAST ? qc.inv : o.U)
                                   : (T)__builtin_nan("");
        stt = st != PDDP_BWD_OK ? st
                               : ((o.result_free < 2) ? (int)PDDP_BWD_BOXQP_FAILED : (int)PDDP_BWD_OK);
      }
      {
        T* pq = &xq[s & 1][lane][0];
        pq[0] = q.kt; pq[1] = q.sK; pq[2] = q.Quu; pq[3] = q.Qu;
      }
      status = (alive & (stt != PDDP_BWD_OK)) ? stt : status;
      kprev = q.kt;
      // Role M reaches the barrier first (its chain is shorter), so its half
      // is normally in LDS already: read it BEFORE the barrier, tag first (M
      // writes its words, then the step tag; LDS executes a wave's
      // instructions in order), and only re-read after the barrier when the
      // tag was not there yet.  Takes one LDS round trip off the chain.
      uint32_t tag = lds_read4(lds_addr(&xf[s & 1][lane]));
      Vec16 vm[MV];
      const uint32_t am = lds_addr(&xm[s & 1][lane][0]);
#pragma unroll
      for (int c = 0; c < MV; ++c) vm[c] = lds_read16(am + 16 * c);
      lds_publish_barrier();
      asm volatile("" : "+v"(tag));
      touch(vm);
      wn = gather((s + 1) % R);  // record t-1: landed before role M's barrier
      if (__builtin_amdgcn_ballot_w64(tag != (uint32_t)t) != 0) {
#pragma unroll
        for (int c = 0; c < MV; ++c) vm[c] = lds_read16(am + 16 * c);
        lds_wait();
        touch(vm);
      }
      const MHalf<T, CHOL> mh = unpack_m(vm);
      T Kc;
      split_tail<T, CHOL, FAST>(q, mh, V, Vzc, Kc);
    };
    Words wa = gather(0), wb = wa;
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; s += 2) {
        if (t < 0) break;
        step(wa, wb, s);
        --t;
        if (t < 0) break;
        step(wb, wa, s + 1);
        --t;
      }
    }
    if (counted && l == 0) a.status[bc] = status;
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
      const int tt = t < 0 ? 0 : t;  // tail: harmless reload keeps vmcnt exact
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
    const int oFuc = gb + 32 + j;
    const int oLuzr = gb + 36 + i;
    const int oLzr = gb + 40 + i;
    const int tr_addr = ((lane & 48) | (j * 4 + i)) * 4;  // lane (j, i)
    struct Words { T Fs0, Fs1, Fs2, Fs3, Fq0, Fq1, Fq2, Fq3, Ft, Lzz, fc, Luzr, Lzr; };
    auto gather = [&](int slot) {
      const T* rc = &ring[slot][0];
      Words w;
      w.Fs0 = rc[oFs[0]]; w.Fs1 = rc[oFs[1]]; w.Fs2 = rc[oFs[2]]; w.Fs3 = rc[oFs[3]];
      w.Fq0 = rc[oFq[0]]; w.Fq1 = rc[oFq[1]]; w.Fq2 = rc[oFq[2]]; w.Fq3 = rc[oFq[3]];
      w.Ft = rc[oFt]; w.Lzz = rc[oLzz]; w.fc = rc[oFuc];
      w.Luzr = rc[oLuzr]; w.Lzr = rc[oLzr];
      return w;
    };
#pragma unroll
    for (int s = 0; s < R; ++s) dma(s, N - 1 - s);
    wait_vmcnt<0>();
    __syncthreads();
    char* gains_w =
        reinterpret_cast<char*>(a.gains + (size_t)b0 * (size_t)N * kGain);
    uint32_t gout_off = (uint32_t)(
        ((bc - b0) * N * kGain + (N - 1) * kGain + ((l < 4) ? 1 + l : 0)) *
        (int)sizeof(T));
    int t = N - 1;
    auto step = [&](const Words& w, Words& wn, const int s) {
      // DMA(t-1) has landed once at most (R-2) younger {store, DMA} pairs are
      // outstanding; the barrier below publishes it to role Q
      wait_vmcnt<(R - 2) * (1 + NI)>();
      wn = gather((s + 1) % R);
      // A = F^T V : A[i][j] = sum_k F[k][i] V[k][j], k = (i + d) % 4
      T A = w.Fs0 * V;
      A += w.Fs1 * from_row_plus<1>(V);
      A += w.Fs2 * from_row_plus<2>(V);
      A += w.Fs3 * from_row_plus<3>(V);
      // Q_zz (raw) = L_zz + A F : sum_k A[i][k] F[k][j], k = (j + d) % 4
      T Qzz = w.Lzz + A * w.Fq0;
      Qzz += from_col_plus<1>(A) * w.Fq1;
      Qzz += from_col_plus<2>(A) * w.Fq2;
      Qzz += from_col_plus<3>(A) * w.Fq3;
      const T Quzr = w.Luzr + dot_cols(A, w.fc);
      const T Qzr = w.Lzr + dot_cols(w.Ft, Vzc);
      T Quzgr = Quzr;
      if constexpr (CHOL) {
        const T Vr = (i == j) ? V + reg : V;
        T Ar = w.Fs0 * Vr;
        Ar += w.Fs1 * from_row_plus<1>(Vr);
        Ar += w.Fs2 * from_row_plus<2>(Vr);
        Ar += w.Fs3 * from_row_plus<3>(Vr);
        Quzgr = w.Luzr + dot_cols(Ar, w.fc);
      }
      const T QzzT = bperm(tr_addr, Qzz);
      MHalf<T, CHOL> mh;
      mh.Quzr = Quzr;
      mh.Quzc = bperm(tr_addr, Quzr);
      mh.Qzc = bperm(tr_addr, Qzr);
      mh.Qzzs = mul_nc(T(0.5), Qzz + QzzT);
      mh.Quzgr = T(0);
      mh.Quzgc = T(0);
      {
        T* pm = &xm[s & 1][lane][0];
        pm[0] = mh.Qzzs; pm[1] = mh.Quzr; pm[2] = mh.Quzc; pm[3] = mh.Qzc;
        if constexpr (CHOL) {
          mh.Quzgr = Quzgr;
          mh.Quzgc = bperm(tr_addr, Quzgr);
          pm[4] = mh.Quzgr;
          pm[5] = mh.Quzgc;
        }
      }
      lds_write4(lds_addr(&xf[s & 1][lane]), (uint32_t)t);  // after the words
      lds_publish_barrier();
      Vec16 vq[QV];
      const uint32_t aq = lds_addr(&xq[s & 1][lane][0]);
#pragma unroll
      for (int c = 0; c < QV; ++c) vq[c] = lds_read16(aq + 16 * c);
      lds_wait();
      touch(vq);
      const QHalf<T> q = unpack_q(vq);
      T Kc;
      split_tail<T, CHOL, FAST>(q, mh, V, Vzc, Kc);
      // ---- store k, K (lanes l < 5 of each group; dead groups write junk)
      {
        const T val = (l < 4) ? Kc : q.kt;
        T* dst = reinterpret_cast<T*>(gains_w + gout_off);
        if (exists && l < 5) *dst = val;
      }
      gout_off -= (uint32_t)(kGain * sizeof(T));
      dma(s, t - R);  // refill this slot, R steps further down the sweep
    };
    Words wa = gather(0), wb = wa;
    while (t >= 0) {
#pragma unroll
      for (int s = 0; s < R; s += 2) {
        if (t < 0) break;
        step(wa, wb, s);
        --t;
        if (t < 0) break;
        step(wb, wa, s + 1);
        --t;
      }
    }
    wait_vmcnt<0>();
  }
}

}  // namespace n4

template <typename T>
static int launch_n4_split(const RiccatiArgs<T>& a, hipStream_t st,
                           bool fast_math) {
  const bool chol = a.branch == PDDP_BRANCH_CHOLESKY;
  const dim3 grid((a.B + 3) / 4), block(n4::kSplitThreads);
#define PDDP_N4_SPLIT(C, F)                                                  \
  PDDP_LAUNCH((n4::riccati_n4_split_kernel<T, C, F>), grid, block, 0, st, a)
  if (fast_math) { if (chol) PDDP_N4_SPLIT(true, true); else PDDP_N4_SPLIT(false, true); }
  else { if (chol) PDDP_N4_SPLIT(true, false); else PDDP_N4_SPLIT(false, false); }
#undef PDDP_N4_SPLIT
  return launch_status();
}

}  // namespace pddp
