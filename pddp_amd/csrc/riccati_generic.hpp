// riccati_generic.hpp - backward Riccati sweep, any n, m == M <= 4.
//
// One workgroup owns one trajectory and walks its records t = N-1 .. 0.
// V_zz, V_z and the step's intermediates live in LDS.  Two instantiations of
// the same body:
//   * n <= NMAX in {8, 16, 32}: one 64-lane wavefront, statically sized LDS,
//     the next record prefetched into registers while the current step is
//     computed, so the HBM stream is one step ahead of the dependent chain;
//   * larger n (the FULL_COVARIANCE_MATRIX encoding of the 6- and 8-state
//     problems has n = 42 and 72): four wavefronts, dynamically sized LDS
//     (three n x n matrices, up to the CU's 160 KB), records read in place
//     through L2 because a record no longer fits beside the matrices.
// The action-space algebra (eig clamp / Cholesky / BoxQP) runs redundantly in
// every lane's registers (gains.hpp).
//
// Restates pddp/controllers/ilqr.py:489-526 (Q) and :529-674 (backward), all
// four gain branches.  The summation order follows the
// reference's association F^T V first, then (F^T V) F.
#pragma once

#include "gains.hpp"

namespace pddp {

template <typename T>
struct RiccatiArgs {
  int B, N, n;
  const T* rec;
  const T* u_min;
  const T* u_max;
  const double* reg;
  int branch;
  const uint8_t* active;
  T* gains;
  int32_t* status;
};

// LDS working set of one trajectory; Qzz may alias W (ALIAS: symmetrised in
// place), rec is null when records are read from global memory.
template <typename T>
struct RiccatiLds {
  T* rec;
  T* V;
  T* W;    // scratch: raw Q_zz, then raw V_zz
  T* A;    // F_z^T V
  T* Qzz;
  T* Vz;
  T* Qz;
  T* Bm;   // F_u^T V
  T* Bmr;  // F_u^T (V + reg I)
  T* Quz;
  T* Quzr;
  T* K;
  T* Qu;
  T* Quu;   // raw until phase 4, then symmetrised
  T* Quur;  // regularised (Cholesky branch), raw
};

template <int NMAX, int M>
constexpr int rec_max() {
  return ((2 * NMAX * NMAX + 2 * NMAX * M + NMAX + M * M + 2 * M) + 3) & ~3;
}

// Elements of LDS the body needs for state size n (ALIAS layout, no record).
__host__ __device__ inline size_t riccati_lds_elems(int n, int M) {
  return (size_t)3 * n * n + 2 * n + (size_t)5 * M * n + M + 2 * M * M;
}

template <typename T>
__device__ inline RiccatiLds<T> carve_lds(T* p, int n, int M, int rec_elems,
                                          bool alias) {
  RiccatiLds<T> s;
  s.rec = rec_elems ? p : nullptr;
  p += rec_elems;
  s.V = p; p += n * n;
  s.W = p; p += n * n;
  s.A = p; p += n * n;
  if (alias) {
    s.Qzz = s.W;
  } else {
    s.Qzz = p; p += n * n;
  }
  s.Vz = p; p += n;
  s.Qz = p; p += n;
  s.Bm = p; p += M * n;
  s.Bmr = p; p += M * n;
  s.Quz = p; p += M * n;
  s.Quzr = p; p += M * n;
  s.K = p; p += M * n;
  s.Qu = p; p += M;
  s.Quu = p; p += M * M;
  s.Quur = p;
  return s;
}

// NT threads; NREG > 0: records staged through registers into s.rec (needs
// NT * NREG >= stride); NREG == 0: records read in place from global memory.
template <typename T, int M, int NT, int NREG, bool ALIAS>
__device__ __forceinline__ void riccati_body(const RiccatiArgs<T>& a,
                                             const RiccatiLds<T>& s) {
  constexpr int kWave = NT;  // loop stride of this body
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (a.active != nullptr && a.active[b] == 0) return;

  const int n = a.n, N = a.N;
  const RecLayout lay(n, M);
  const int S = lay.stride;
  const T* rec_b = a.rec + (size_t)b * (size_t)(N + 1) * S;
  T* g_b = a.gains + (size_t)b * (size_t)N * lay.gstride;
  const T reg = (T)a.reg[b];
  const bool bounded = (a.u_min != nullptr) && (a.u_max != nullptr);
  const bool chol = (a.branch == PDDP_BRANCH_CHOLESKY);

  T umin[M], umax[M];
#pragma unroll
  for (int r = 0; r < M; ++r) {
    umin[r] = bounded ? a.u_min[r] : T(0);
    umax[r] = bounded ? a.u_max[r] : T(0);
  }

  // V = L_zz[N], V_z = L_z[N]                                   (ilqr.py:574-575)
  {
    const T* term = rec_b + (size_t)N * S;
    for (int i = lane; i < n * n; i += kWave) s.V[i] = term[lay.oLzz + i];
    for (int i = lane; i < n; i += kWave) s.Vz[i] = term[lay.oLz + i];
  }

  T pre[NREG > 0 ? NREG : 1];
  auto prefetch = [&](int t) {
    const T* src = rec_b + (size_t)t * S;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const int idx = lane + kWave * r;
      pre[r] = (idx < S) ? src[idx] : T(0);
    }
  };
  if (NREG > 0) prefetch(N - 1);

  T kprev[M];  // warm start k[t+1]; zeros at t = N-1        (ilqr.py:604,649)
#pragma unroll
  for (int r = 0; r < M; ++r) kprev[r] = T(0);
  int status = PDDP_BWD_OK;

  for (int t = N - 1; t >= 0; --t) {
    const T* R = rec_b + (size_t)t * S;
    if (NREG > 0) {
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const int idx = lane + kWave * r;
        if (idx < S) s.rec[idx] = pre[r];
      }
      __syncthreads();
      if (t > 0) prefetch(t - 1);
      R = s.rec;
    }
    const T* Fz = R + lay.oFz;
    const T* Fu = R + lay.oFu;
    const T* Lzz = R + lay.oLzz;
    const T* Luz = R + lay.oLuz;
    const T* Lz = R + lay.oLz;
    const T* Luu = R + lay.oLuu;
    const T* Lu = R + lay.oLu;
    const T* Un = R + lay.oU;

    // ---- phase 1: F^T V products, Q_z, Q_u                    (ilqr.py:519-524)
    for (int w = lane; w < n * n; w += kWave) {
      const int i = w / n, j = w - i * n;
      T acc = T(0);
      for (int k = 0; k < n; ++k) acc += Fz[k * n + i] * s.V[k * n + j];
      s.A[w] = acc;
    }
    for (int w = lane; w < M * n; w += kWave) {
      const int r = w / n, j = w - r * n;
      T acc = T(0), accr = T(0);
      for (int k = 0; k < n; ++k) {
        const T v = s.V[k * n + j];
        acc += Fu[k * M + r] * v;
        accr += Fu[k * M + r] * ((k == j) ? (v + reg) : v);
      }
      s.Bm[w] = acc;
      s.Bmr[w] = accr;
    }
    for (int i = lane; i < n; i += kWave) {
      T acc = T(0);
      for (int k = 0; k < n; ++k) acc += Fz[k * n + i] * s.Vz[k];
      s.Qz[i] = Lz[i] + acc;
    }
    if (lane < M) {
      T acc = T(0);
      for (int k = 0; k < n; ++k) acc += Fu[k * M + lane] * s.Vz[k];
      s.Qu[lane] = Lu[lane] + acc;
    }
    __syncthreads();

    // ---- phase 2: Q_zz (raw), Q_uz, Q_uu (raw)
    for (int w = lane; w < n * n; w += kWave) {
      const int i = w / n, j = w - i * n;
      T acc = T(0);
      for (int k = 0; k < n; ++k) acc += s.A[i * n + k] * Fz[k * n + j];
      s.W[w] = Lzz[w] + acc;
    }
    for (int w = lane; w < M * n; w += kWave) {
      const int r = w / n, j = w - r * n;
      T acc = T(0), accr = T(0);
      for (int k = 0; k < n; ++k) {
        acc += s.Bm[r * n + k] * Fz[k * n + j];
        accr += s.Bmr[r * n + k] * Fz[k * n + j];
      }
      s.Quz[w] = Luz[w] + acc;
      s.Quzr[w] = Luz[w] + accr;
    }
    if (lane < M * M) {
      const int r = lane / M, c = lane - r * M;
      T acc = T(0), accr = T(0);
      for (int k = 0; k < n; ++k) {
        acc += s.Bm[r * n + k] * Fu[k * M + c];
        accr += s.Bmr[r * n + k] * Fu[k * M + c];
      }
      s.Quu[lane] = Luu[lane] + acc;
      s.Quur[lane] = Luu[lane] + accr;
    }
    __syncthreads();

    // ---- phase 3: symmetrise Q_zz                                (ilqr.py:522)
    if (ALIAS) {  // Qzz is W: each pair (i < j) rewritten by one thread
      for (int w = lane; w < n * n; w += kWave) {
        const int i = w / n, j = w - i * n;
        if (i < j) {
          const T h = T(0.5) * (s.W[i * n + j] + s.W[j * n + i]);
          s.W[i * n + j] = h;
          s.W[j * n + i] = h;
        }
      }
      __syncthreads();
    } else {
      for (int w = lane; w < n * n; w += kWave) {
        const int i = w / n, j = w - i * n;
        s.Qzz[w] =
            (i == j) ? s.W[w] : T(0.5) * (s.W[i * n + j] + s.W[j * n + i]);
      }
    }

    // ---- phase 4: gains (registers, every lane)             (ilqr.py:587-662)
    T Quu[M * M], Qg[M * M], Qu[M], Qug[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      Qu[i] = s.Qu[i];
      Qug[i] = Qu[i];  // Q_u_reg == Q_u: it does not involve V_zz
#pragma unroll
      for (int j = 0; j < M; ++j) {
        // 0.5 (Q + Q^T), diagonal untouched                        (ilqr.py:525)
        Quu[i * M + j] = (i == j) ? s.Quu[i * M + i]
                                  : T(0.5) * (s.Quu[i * M + j] + s.Quu[j * M + i]);
        Qg[i * M + j] = (i == j) ? s.Quur[i * M + i]
                                 : T(0.5) * (s.Quur[i * M + j] + s.Quur[j * M + i]);
      }
    }
    const T* Quz_g = chol ? s.Quzr : s.Quz;  // operand of the K solve
    T kt[M];
    T inv[M * M];  // branch A: Q_uu^-1
    T Uf[M * M];   // Cholesky factor (branches B, C, D)
    unsigned free_bits = (1u << M) - 1u;
    int mode;  // 0: K = -inv Quz ; 1: K = -potrs(Quz[free], Uf)

    if (!chol) {
      bool finite = true;
#pragma unroll
      for (int i = 0; i < M * M; ++i) finite = finite && is_finite(Quu[i]);
      if (!finite) {  // torch's eig raises on non-finite input -> RuntimeError
        status = PDDP_BWD_NAN;
        break;
      }
      T e[M], E[M * M];
      jacobi_eig<T, M>(Quu, e, E);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        e[i] = (e[i] < T(0)) ? T(1e-12) : e[i];  // ilqr.py:633
        e[i] += reg;                              // ilqr.py:634
      }
      if (!bounded) {
        // Q_uu_inv = (E / e) E^T                                   (ilqr.py:636)
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            T acc = T(0);
#pragma unroll
            for (int q = 0; q < M; ++q) acc += (E[i * M + q] / e[q]) * E[j * M + q];
            inv[i * M + j] = acc;
          }
        bool bad = false;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          T acc = T(0);
#pragma unroll
          for (int j = 0; j < M; ++j) acc += inv[i * M + j] * Qu[j];
          kt[i] = -acc;
          bad = bad || (kt[i] != kt[i]);
        }
        // NaN anywhere in K also raises (ilqr.py:639): checked below, after K.
        mode = 0;
        if (bad) {
          status = PDDP_BWD_NAN;
          break;
        }
      } else {
        // Q_uu_reg = (E * e) E^T                                    (ilqr.py:645)
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
          for (int j = 0; j < M; ++j) {
            T acc = T(0);
#pragma unroll
            for (int q = 0; q < M; ++q) acc += (E[i * M + q] * e[q]) * E[j * M + q];
            Qg[i * M + j] = acc;
          }
        mode = 1;
      }
    } else {
      mode = 1;
      if (!bounded) {
        if (chol_upper_masked<T, M>(Qg, free_bits, Uf)) {  // ilqr.py:595
          status = PDDP_BWD_NOT_PD;
          break;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) kt[i] = Qug[i];
        chol_solve<T, M>(Uf, kt);
#pragma unroll
        for (int i = 0; i < M; ++i) kt[i] = -kt[i];
      }
    }
    if (bounded) {
      // BoxQP for k, warm-started at k[t+1]                  (ilqr.py:602-610)
      T lower[M], upper[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        lower[i] = umin[i] - Un[i];
        upper[i] = umax[i] - Un[i];
      }
      const int result =
          boxqp<T, M>(kprev, Qg, Qug, lower, upper, kt, Uf, free_bits);
      if (result < 1) {
        status = PDDP_BWD_BOXQP_FAILED;
        break;
      }
    }

    // K, one column per lane
    bool badK = false;
    for (int c = lane; c < n; c += kWave) {
      T col[M];
      if (mode == 0) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
          T acc = T(0);
#pragma unroll
          for (int j = 0; j < M; ++j) acc += inv[i * M + j] * Quz_g[j * n + c];
          col[i] = -acc;
          badK = badK || (col[i] != col[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < M; ++i)
          col[i] = ((free_bits >> i) & 1u) ? Quz_g[i * n + c] : T(0);
        if (free_bits != 0u) chol_solve<T, M>(Uf, col);
#pragma unroll
        for (int i = 0; i < M; ++i)
          col[i] = ((free_bits >> i) & 1u) ? -col[i] : T(0);
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        s.K[i * n + c] = col[i];
        g_b[(size_t)t * lay.gstride + M + i * n + c] = col[i];
      }
    }
    if (mode == 0 && __syncthreads_or(badK)) {  // ilqr.py:639-640
      status = PDDP_BWD_NAN;
      break;
    }
    if (lane < M) {
      T v = T(0);
#pragma unroll
      for (int i = 0; i < M; ++i) v = (lane == i) ? kt[i] : v;
      g_b[(size_t)t * lay.gstride + lane] = v;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) kprev[i] = kt[i];
    __syncthreads();

    // ---- phase 5: value update with the UN-regularised Q_uu, Q_uz
    //                                                (ilqr.py:619-625,664-672)
    for (int i = lane; i < n; i += kWave) {
      T s1 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll
      for (int r = 0; r < M; ++r) s1 += s.K[r * n + i] * Qu[r];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        T kq = T(0);  // (K^T Q_uu)[i][j]
#pragma unroll
        for (int r = 0; r < M; ++r) kq += s.K[r * n + i] * Quu[r * M + j];
        s2 += kq * kt[j];
      }
#pragma unroll
      for (int r = 0; r < M; ++r) s3 += s.Quz[r * n + i] * kt[r];
      T v = s.Qz[i] + s1;
      v += s2;
      v += s3;
      s.Qz[i] = v;  // new V_z, published in phase 6
    }
    for (int w = lane; w < n * n; w += kWave) {
      const int i = w / n, j = w - i * n;
      T s1 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll
      for (int q = 0; q < M; ++q) {
        T kq = T(0);
#pragma unroll
        for (int r = 0; r < M; ++r) kq += s.K[r * n + i] * Quu[r * M + q];
        s1 += kq * s.K[q * n + j];
      }
#pragma unroll
      for (int r = 0; r < M; ++r) s2 += s.K[r * n + i] * s.Quz[r * n + j];
#pragma unroll
      for (int r = 0; r < M; ++r) s3 += s.Quz[r * n + i] * s.K[r * n + j];
      T v = s.Qzz[w] + s1;
      v += s2 + s3;
      s.W[w] = v;
    }
    __syncthreads();

    // ---- phase 6: V_zz = 0.5 (V + V^T), V_z                      (ilqr.py:625)
    for (int w = lane; w < n * n; w += kWave) {
      const int i = w / n, j = w - i * n;
      s.V[w] = T(0.5) * (s.W[i * n + j] + s.W[j * n + i]);
    }
    for (int i = lane; i < n; i += kWave) s.Vz[i] = s.Qz[i];
    __syncthreads();
  }

  if (lane == 0) a.status[b] = status;
}

template <typename T, int NMAX, int M>
__global__ __launch_bounds__(64) void riccati_generic_kernel(
    RiccatiArgs<T> a) {
  constexpr int kRec = rec_max<NMAX, M>();
  constexpr int kElems = kRec + 4 * NMAX * NMAX + 2 * NMAX + 5 * M * NMAX + M +
                         2 * M * M;
  __shared__ T lds[kElems];
  riccati_body<T, M, 64, (kRec + 63) / 64, false>(
      a, carve_lds<T>(lds, a.n, M, kRec, false));
}

constexpr int kLargeThreads = 256;

template <typename T, int M>
__global__ __launch_bounds__(kLargeThreads) void riccati_large_kernel(
    RiccatiArgs<T> a) {
  extern __shared__ double lds_dyn[];
  riccati_body<T, M, kLargeThreads, 0, true>(
      a, carve_lds<T>(reinterpret_cast<T*>(lds_dyn), a.n, M, 0, true));
}

}  // namespace pddp
