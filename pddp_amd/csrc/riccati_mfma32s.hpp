// riccati_mfma32s.hpp - riccati_mfma32.hpp's sweep (15 <= n <= 30, m = 1, fp32)
// with ONE TRAJECTORY'S STEP SPLIT OVER TWO WAVEFRONTS on different SIMDs
// (task 5).  At configs[3]'s 1024 trajectories per GPU the
// one-wave form leaves every SIMD with one wavefront whose 2048 matrix-core
// cycles and ~460 vector instructions per step serialise (the vector work
// depends on the products).  Here wavefront w of a pair owns 16 of the 32
// columns; both products are done on v_mfma_f32_16x16x4_f32 (at most 16
// instructions of 32 cycles per product and wave instead of 16 of 64; the
// k-slices that hold only zero rows of F~ are skipped, see the permutation
// below), the vector work halves, and a SIMD holds halves of two trajectories.
//
//   X = V F~ (32 x 32, X[:, 31] = V_z), Q~ = L~ + F~^T X as in
//   riccati_mfma32.hpp.  Lane (c, g) = (lane & 15, lane >> 4); rho(rb, r, g) =
//   16 rb + 4 g + r is the row that register (rb, r) of an accumulator block
//   holds in lane group g (the C/D layout of the 16x16 form).
//
//   wave w keeps  Vw[(rb, r)]  = V[rho][16 w + c]   (= V[16 w + c][rho]: symmetric)
//                 Vzw[r]       = V_z[16 w + 4 g + r]
//   product 1, ROWS 16 w ..: X[16 w + i][:] = sum_rho V[16 w + i][rho] F~[rho][:]
//              A of k-slice (rb, r) = Vw[(rb, r)] as it stands; B = F~[rho][16 cb + c]
//              gathered from the record in LDS (16 words: Fg[cb][(rb, r)])
//   exchange:  the 16 x 16 block of X the partner's columns need, through LDS
//   product 2, COLUMNS 16 w ..: Q~[:, 16 w + c] = L~ + sum_rho F~[rho][:]^T X[rho][16 w + c]
//              B of k-slice (rb, r) = the accumulator registers of X (own rows:
//              own registers; the partner's rows: the exchanged block);
//              A = F~[rho][16 ob + c] = the SAME 16 gathered words
//   Q~'s block lands in the layout of Vw: V' = sym(Q~) + c Quz^T Quz needs the
//   transposed entries and row n, both through a 32 x 32 tile in LDS that the
//   two waves fill together.
//
// Three workgroup barriers per step (record landed - X exchanged - tile
// filled).  One trajectory per workgroup: two sweeping waves + the producer
// wavefront of riccati_mfma32.hpp; the SIMD's second sweeping wave is another
// workgroup's.  (Measured and removed: a two-barrier form with the next
// record's operands gathered behind the last barrier - 280 against 297 us at
// 1024 trajectories, slower from 2048 on; four trajectories per workgroup in
// two groups held ONE PHASE APART by the shared barriers, so that a SIMD's two
// waves never do the same thing - 340 us: the f32 matrix instruction runs on
// the vector unit's own multipliers, its rate IS the packed vector rate, and a
// SIMD's second wave does not advance under it; what a SIMD has to issue per
// step pair - 2 x (896 matrix + ~800 other) cycles at n = 27 - is what the
// sweep takes.)
// Eig-clamp branches (B, D: the controller's default); the Cholesky branches
// stay on the one-wave kernel.
#pragma once

#include "riccati_mfma32.hpp"

namespace pddp {
namespace m32s {

constexpr int kRing = 2;
constexpr int kTileLd = 36;
constexpr int kTile = 32 * kTileLd;
constexpr int kXch = 2 * kWave * 4;      // two 16 x 16 blocks, lane-linear

constexpr int kTraj = 1;                 // trajectories per workgroup
constexpr int kSweep = 2 * kTraj;        // sweeping wavefronts
constexpr int kThreads = kWave * 3 * kTraj;  // + a producer per trajectory

#ifdef PDDP_WG_TIMELINE
__device__ long long g_mfma32s_clock[2];
#endif
template <bool BOUNDED, bool FAST, int NDMA>
__global__ __launch_bounds__(kThreads) void riccati_mfma32s_kernel(
    RiccatiArgs<float> a) {
  using T = float;
  using m16::f32x4;
  constexpr int kSlotWords = NDMA * kWave * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  constexpr int per_traj = kRing * kSlotWords + kTile + kXch;
  float* ls_tail = smem + kTraj * per_traj;
  if constexpr (BOUNDED) {
    for (int q = threadIdx.x; q < n4::kLsSteps; q += kThreads)
      ls_tail[q] = (T)n4::kLs.v[q];
  }
  const T lstep0 = (T)n4::kLs.v[lane & 15];
  const int n = a.n, N = a.N;
  const RecLayout lay(n, 1);
  const int S = lay.stride;
  const int chunks = S / 4;
  const int nd_live = (chunks + kWave - 1) / kWave;
  auto live_of = [&](int bb) {
    return bb < a.B && (a.active == nullptr || a.active[bb] != 0);
  };
  auto step_barrier = [] { asm volatile("s_barrier" ::: "memory"); };
  auto publish_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  if (wv >= kSweep) {
    const int pw = wv - kSweep;  // this producer's trajectory
    // ===================================================================
    // producer (riccati_mfma32.hpp): record t of both trajectories has landed
    // before the first barrier of step t; the two other barriers of a step
    // are the sweeping waves' own
    // ===================================================================
    uint32_t qoff[NDMA];
#pragma unroll
    for (int r = 0; r < NDMA; ++r) qoff[r] = (uint32_t)(lane + kWave * r) * 16u;
    const uint32_t qoff_last = (uint32_t)(lane + kWave * (nd_live - 1)) * 16u;
    const bool in_tail = lane + kWave * (nd_live - 1) < chunks;
    auto dma = [&](int slot, int t) {
      const int w = pw;
      const int bb = blockIdx.x * kTraj + w;
      if (!live_of(bb) || t < 0) return;  // (wave-uniform)
      const char* rec_b = reinterpret_cast<const char*>(
          a.rec + (size_t)bb * (size_t)(N + 1) * S);
      const uint32_t base = (uint32_t)t * (uint32_t)(S * sizeof(T));
      const uint32_t lbase = __builtin_amdgcn_readfirstlane(
          n4::lds_addr(smem + w * per_traj + slot * kSlotWords));
#pragma unroll
      for (int r = 0; r < NDMA - 1; ++r) {
        if (r < nd_live - 1)
          n4::lds_dma16(rec_b, base + qoff[r], lbase + r * kWave * 16);
      }
      if (in_tail)
        n4::lds_dma16(rec_b, base + qoff_last,
                      lbase + (uint32_t)(nd_live - 1) * kWave * 16);
    };
    __syncthreads();  // the sweeping waves have zeroed their slots' padding
    dma(0, N - 1);
    for (int t = N - 1; t >= 0; --t) {
      n4::wait_vmcnt<0>();
      step_barrier();  // (1) record t landed
      const int slot_next = (N - t) & 1;
      dma(slot_next, t - 1);
      step_barrier();  // (2)
      step_barrier();  // (3)
    }
    return;
  }

  const int tw = wv >> 1;  // trajectory of the workgroup
  const int w = wv & 1;    // column half
  float* ring = smem + tw * per_traj;
  float* tile = ring + kRing * kSlotWords;
  float* xch = tile + kTile;
  // zero the padding of both slots once (the pair shares the ring: half each)
  for (int sl = 0; sl < kRing; ++sl)
    for (int wd = S + lane + kWave * w; wd < kSlotWords; wd += 2 * kWave)
      ring[sl * kSlotWords + wd] = T(0);
  __syncthreads();

  const int b = blockIdx.x * kTraj + tw;
  if (!live_of(b)) {  // keep step with the workgroup's barriers
    for (int t = N - 1; t >= 0; --t) {
      step_barrier();
      step_barrier();
      step_barrier();
    }
    return;
  }
  const int c = lane & 15, g = lane >> 4;
  const int jcol = 16 * w + c;  // this lane's column of X (product 2) / Q~ / V
  const T reg = (T)a.reg[b];
  T umin = T(0), umax = T(0);
  if constexpr (BOUNDED) { umin = a.u_min[0]; umax = a.u_max[0]; }

  // ---- word offsets of this lane's operands inside a record; entries
  // outside the matrices read the slot's zeroed padding (word S)
  auto rho = [&](int s) { return 16 * (s >> 2) + 4 * g + (s & 3); };
  // Rows and columns live in PERMUTED positions, so that the indices that are
  // zero rows of F~ - n .. 31: the action's row n is one - fill whole k-slices:
  // q = (32 - n) / 4 of them, (1, 3), (1, 2), .. = the positions p >= 16 with
  // (p & 3) >= 4 - q, drop out of both products (n = 27: 28 matrix
  // instructions per step and wave instead of 32; n = 20: 20).  Indices go to
  // positions in ascending order, the live ones (0 .. 31 - 4 q) and the dead
  // ones (32 - 4 q .. 31) each among themselves: the carrier of V_z / Q_z stays
  // at 31.  lg: the index a position holds; pos: its inverse.
  const int q4 = (32 - n) >> 2;  // (n >= 15: at most 4)
  auto lg = [&](int p) {
    if (p < 16) return p;
    const int gg = (p - 16) >> 2, rr = p & 3;
    return rr >= 4 - q4 ? 32 - 4 * q4 + q4 * gg + (rr - (4 - q4))
                        : 16 + (4 - q4) * gg + rr;
  };
  auto pos = [&](int idx) {
    if (idx < 16) return idx;
    if (idx >= 32 - 4 * q4) {
      const int k = idx - (32 - 4 * q4);
      return 16 + 4 * (k / q4) + (4 - q4) + k % q4;
    }
    const int k = idx - 16;
    return 16 + 4 * (k / (4 - q4)) + k % (4 - q4);
  };
  // (wave-uniform) k-slice s = 4 + r is dead for r >= 4 - q
  const int first_dead = 8 - q4;
  const int pn = pos(n);    // position of the action's row / column
  const int jl = lg(jcol);  // index of this lane's column
  int oF[2][8], oL[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = lg(rho(s));
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int j = lg(16 * cb + c);
      const int f = (k < n) ? (j < n ? lay.oFz + k * n + j
                                     : (j == n ? lay.oFu + k : -1))
                            : -1;
      oF[cb][s] = f < 0 ? S : f;
    }
    int o = -1;
    if (k < n) {
      if (jl < n) o = lay.oLzz + k * n + jl;
      else if (jl == n) o = lay.oLuz + k;  // L_uz^T
      else if (jl == 31) o = lay.oLz + k;
    } else if (k == n) {
      if (jl < n) o = lay.oLuz + jl;
      else if (jl == n) o = lay.oLuu;
      else if (jl == 31) o = lay.oLu;
    }
    oL[s] = o < 0 ? S : o;
  }

  // ---- terminal value function (ilqr.py:581-583)
  T Vw[8], Vzw[4];
  {
    const T* term = a.rec + ((size_t)b * (size_t)(N + 1) + N) * S;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = lg(rho(s));
      Vw[s] = (k < n && jl < n) ? term[lay.oLzz + k * n + jl] : T(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = lg(16 * w + 4 * g + r);
      Vzw[r] = (k < n) ? term[lay.oLz + k] : T(0);
    }
  }

  T* gains_b = a.gains + (size_t)b * (size_t)N * lay.gstride;
  T kprev = T(0);
  int status = PDDP_BWD_OK;
  // row n of Q~ (position pn) sits in the registers of lane group (pn & 15) >> 2
  const int gn = (pn & 15) >> 2;
  static_assert(kRing == 2, "the step is instantiated once per ring slot");
  // this lane's operands of a step, gathered from the record's slot: 16 words
  // of F~ (both products), 8 of L~ (this wave's columns), the nominal action
  T Fg[2][8], Li[8], Un = T(0);
  f32x4 X0, X1, Q0, Q1;
  // ---- product 1: rows 16 w .. of X = V F~, both column blocks, on record R
  // (W: the wave's column half as a compile-time constant - the choices
  // between own and exchanged registers cost no instruction)
  auto phase_p1 = [&](auto half, const T* R) {
    constexpr int W = decltype(half)::value;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      Fg[0][s] = R[oF[0][s]];
      Fg[1][s] = R[oF[1][s]];
      Li[s] = R[oL[s]];
    }
    if constexpr (BOUNDED) Un = R[lay.oU];
    // (a wave that feeds the matrix pipe goes first: it needs one issue slot
    // in 32 cycles, and the other wavefront of the SIMD - another trajectory -
    // fills the rest with its vector work)
    X0 = f32x4{0, 0, 0, 0};
    X1 = f32x4{0, 0, 0, 0};
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s >= first_dead) break;
      X0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Vw[s], Fg[0][s], X0, 0, 0, 0);
      X1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Vw[s], Fg[1][s], X1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // X[:, 31] = V_z (lanes c = 15 of column block 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) X1[r] = (c == 15) ? Vzw[r] : X1[r];
    // the partner's columns of my rows go to it; mine of its rows come back
    const f32x4 give = W == 0 ? X1 : X0;
    *reinterpret_cast<f32x4*>(xch + (W * kWave + lane) * 4) = give;
  };
  // ---- product 2: columns 16 w .. of Q~ = L~ + F~^T X, both row blocks.
  // k-slices (rb, r): rows 16 rb + 4 g + r of X[:, 16 w + c]; rb = 0 are wave
  // 0's rows (w is wave-uniform: selects, no indexing)
  auto phase_p2 = [&](auto half) {
    constexpr int W = decltype(half)::value;
    const f32x4 Xo =
        *reinterpret_cast<const f32x4*>(xch + ((1 - W) * kWave + lane) * 4);
    const f32x4 Xm = W == 0 ? X0 : X1;
    Q0 = f32x4{Li[0], Li[1], Li[2], Li[3]};
    Q1 = f32x4{Li[4], Li[5], Li[6], Li[7]};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const T xk = s < 4 ? (W == 0 ? Xm[s & 3] : Xo[s & 3])
                         : (W == 0 ? Xo[s & 3] : Xm[s & 3]);
      if (s == 0) __builtin_amdgcn_s_setprio(3);
      if (s >= first_dead) break;
      Q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Fg[0][s], xk, Q0, 0, 0, 0);
      Q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Fg[1][s], xk, Q1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // transpose tile: T[col][row] = Q~[row][col]
    *reinterpret_cast<f32x4*>(tile + jcol * kTileLd + 4 * g) = Q0;
    *reinterpret_cast<f32x4*>(tile + jcol * kTileLd + 16 + 4 * g) = Q1;
  };
  // ---- BoxQP, gains of step t, V' and V_z' (the tile is filled)
  auto phase_v = [&](auto half, int t) {
    constexpr int W = decltype(half)::value;
    const T Quu = tile[pn * kTileLd + pn];
    const T Qu = tile[31 * kTileLd + pn];
    const T Quz_j = tile[jcol * kTileLd + pn];  // Q~[n][this lane's column]
    T QT[8], Qk[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      QT[s] = tile[rho(s) * kTileLd + jcol];  // Q~[16 w + c][rho]
      Qk[s] = tile[rho(s) * kTileLd + pn];    // Q~[n][rho]
    }
    const f32x4 Qz = *reinterpret_cast<const f32x4*>(
        tile + 31 * kTileLd + 16 * W + 4 * g);  // Q~[16 w + 4 g + r][31]
    // gains (every lane the same scalars)                      (ilqr.py:629-657)
    int st = PDDP_BWD_OK;
    if (!is_finite(Quu)) st = PDDP_BWD_NAN;     // eig raises (ilqr.py:631)
    const T e = (Quu < T(0)) ? T(1e-12) : Quu;  // ilqr.py:633
    const T qp_Q = e + reg;                     // ilqr.py:634
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
      // NaN in k or K raises (ilqr.py:639-640)
      const bool nanK = (g == gn) && (jl < n) && (sE * Quz_j != sE * Quz_j);
      // (the partner sees its own columns only: the statuses are merged below)
      if (kt != kt || __builtin_amdgcn_ballot_w64(nanK) != 0) stt = PDDP_BWD_NAN;
    }
    if (status == PDDP_BWD_OK && stt != PDDP_BWD_OK) status = stt;
    kprev = kt;
    const T cc = sE * (sE * Quu - T(2));
    const T wc = kt - sE * (Qu + Quu * kt);
    // k, K of step t: this wave's columns of row n
    {
      const T val = (jl < n) ? -(sE * Quz_j) : kt;
      T* dst = gains_b + (size_t)t * lay.gstride + (jl < n ? 1 + jl : 0);
      if (g == gn && jl <= n) *dst = val;
    }
    // V' = sym(Q_zz) + c Q_uz^T Q_uz,  V_z' = Q_z + Q_uz^T w
    // (no masks: entries outside the n x n block meet zero rows / columns of
    // F~ in both products of the next step - riccati_mfma32.hpp)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const T q = s < 4 ? Q0[s & 3] : Q1[s & 3];
      const T sym = T(0.5) * (q + QT[s]);
      Vw[s] = __builtin_fmaf(cc * Qk[s], Quz_j, sym);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const T qk = W == 0 ? Qk[r] : Qk[4 + r];  // rows 16 w + 4 g + r
      Vzw[r] = __builtin_fmaf(qk, wc, Qz[r]);
    }
  };
  // one step on the record in ring slot SLOT (a compile-time constant: the
  // gathers are ds_read with an immediate slot offset on per-lane addresses)
  auto step = [&](auto half, auto slot_c, int t) {
    constexpr int slot = decltype(slot_c)::value;
    const T* R = ring + slot * kSlotWords;
    step_barrier();  // (1) record t has landed
    phase_p1(half, R);
    publish_barrier();  // (2) X exchanged
    phase_p2(half);
    publish_barrier();  // (3) tile filled
    phase_v(half, t);
  };
  auto sweep = [&](auto half) {
    for (int t = N - 1; t >= 0; t -= 2) {
      step(half, std::integral_constant<int, 0>{}, t);
      if (t >= 1) step(half, std::integral_constant<int, 1>{}, t - 1);
    }
  };
#ifdef PDDP_WG_TIMELINE
  // (tools/dbg/clock_after_kernels.py: the shader clock DURING the sweep -
  // cycles and 100 MHz ticks of workgroup 0's first sweeping wavefront)
  const bool tl_ = blockIdx.x == 0 && tw == 0 && w == 0 && lane == 0;
  const long long tl_c0 = clock64(), tl_t0 = wall_clock64();
#endif
  if (w == 0) sweep(std::integral_constant<int, 0>{});
  else sweep(std::integral_constant<int, 1>{});
  n4::wait_vmcnt<0>();
#ifdef PDDP_WG_TIMELINE
  if (tl_) {
    g_mfma32s_clock[0] = clock64() - tl_c0;
    g_mfma32s_clock[1] = wall_clock64() - tl_t0;
  }
#endif
  // both halves hold the same status except for the unbounded branch's NaN
  // test of K, which each makes on its own columns: the first non-zero wins,
  // column half 0 first (one writer per trajectory: wave 0 after the exchange)
  if (lane == 0) xch[w] = __int_as_float(status);
  publish_barrier();
  if (w == 0 && lane == 0) {
    const int other = __float_as_int(xch[1]);
    a.status[b] = status != PDDP_BWD_OK ? status : other;
  }
}

}  // namespace m32s

// 15 <= n <= 30, m = 1, fp32, eig-clamp branches; PDDP_E_UNSUPPORTED otherwise
static int launch_mfma32s(const RiccatiArgs<float>& a, hipStream_t st,
                          bool fast_math) {
  if (a.n < 15 || a.n > 30) return PDDP_E_UNSUPPORTED;
  if (a.branch == PDDP_BRANCH_CHOLESKY) return PDDP_E_UNSUPPORTED;
  const RecLayout lay(a.n, 1);
  const int chunks = lay.stride / 4;
  const int ndma = chunks / kWave + 1 <= 4 ? 4 : 8;
  if (chunks >= ndma * kWave) return PDDP_E_UNSUPPORTED;
  constexpr int tj = m32s::kTraj;
  const size_t lds =
      sizeof(float) * ((size_t)tj * (m32s::kRing * ndma * kWave * 4 +
                                     m32s::kTile + m32s::kXch) +
                       n4::kLsSteps);
  const dim3 grid((a.B + tj - 1) / tj), block(m32s::kThreads);
  const bool bounded = a.u_min != nullptr;
#define PDDP_M32S(Bd, F, ND)                                                   \
  do {                                                                         \
    auto kern = m32s::riccati_mfma32s_kernel<Bd, F, ND>;                       \
    const hipError_t e_ = hipFuncSetAttribute(                                 \
        (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,         \
        (int)lds);                                                             \
    if (e_ != hipSuccess) return (int)e_;                                      \
    PDDP_LAUNCH(kern, grid, block, lds, st, a);                                \
  } while (0)
#define PDDP_M32S_ND(Bd, F)                                                    \
  do {                                                                         \
    if (ndma == 4) PDDP_M32S(Bd, F, 4); else PDDP_M32S(Bd, F, 8);              \
  } while (0)
  if (bounded) { if (fast_math) PDDP_M32S_ND(true, true); else PDDP_M32S_ND(true, false); }
  else { if (fast_math) PDDP_M32S_ND(false, true); else PDDP_M32S_ND(false, false); }
#undef PDDP_M32S_ND
#undef PDDP_M32S
  return launch_status();
}

}  // namespace pddp
