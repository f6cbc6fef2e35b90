// controller.hip - per-trajectory controller state machine on the device,
// record packing and the host-side probes / timing helpers of the C ABI.
//
//   accept / reject + mu schedule   pddp/controllers/ilqr.py:102-181,364-390
//   fit() loop bookkeeping          pddp/controllers/ilqr.py:298-314
#include "accept.hpp"

namespace pddp {

template <typename T>
__global__ __launch_bounds__(kAcceptThreads) void accept_kernel(AcceptArgs<T> a) {
  __shared__ int sh_amin;  // >= 0: accepted candidate, -1: nothing to copy
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const bool attempted = a.active[b] != 0;
  const AcceptIn<T> in = accept_load(a, b);
  T J[kMaxAlphas];
  {
    const T* Jg = a.Jc + (size_t)b * a.A;
#pragma unroll
    for (int i = 0; i < kMaxAlphas; ++i) J[i] = Jg[i < a.A ? i : 0];
  }
  if (!attempted) return;  // masks of finished trajectories stay 0

  if (tid == 0) {
    T J_new;
    const int amin = argmin_first(J, a.A, J_new);
    bool fresh;
    sh_amin = accept_decide(a, b, in, amin, J_new, fresh);
  }
  __syncthreads();
  const int amin = sh_amin;
  if (amin < 0) return;
  // nominal <- winning candidate; self._K <- K                (ilqr.py:167-169)
  const int n = a.n, m = a.m, N = a.N;
  // candidates are time-major: Zc [B][N+1][A][n], Uc [B][N][A][m]
  T* Zb = a.Z + (size_t)b * (N + 1) * n;
  T* Ub = a.U + (size_t)b * N * m;
  const T* srcz = a.Zc + ((size_t)b * (N + 1) * a.A + amin) * n;
  const T* srcu = a.Uc + ((size_t)b * N * a.A + amin) * m;
  const size_t zstep = (size_t)a.A * n, ustep = (size_t)a.A * m;
  // lane -> (row t, word j) once; then whole rows per pass (no per-element
  // division) whenever the row length divides the wavefront
  auto gather = [&](T* dst, const T* src, int rows, int w, size_t step) {
    if (kAcceptThreads % w == 0) {
      const int per = kAcceptThreads / w;
      const int j = tid % w;
      for (int t = tid / w; t < rows; t += per) dst[t * w + j] = src[t * step + j];
    } else {
      for (int o = tid; o < rows * w; o += kAcceptThreads) {
        const int t = o / w, j = o - t * w;
        dst[o] = src[t * step + j];
      }
    }
  };
  gather(Zb, srcz, N + 1, n, zstep);
  gather(Ub, srcu, N, m, ustep);
  const int gs = m + m * n;
  const T* G = a.gains + (size_t)b * N * gs;
  T* Ga = a.gains_acc + (size_t)b * N * gs;
  for (int o = tid; o < N * gs; o += kAcceptThreads) Ga[o] = G[o];
}

template <typename T>
static int accept_impl(int B, int N, int n, int m, int A, const T* Zc,
                       const T* Uc, const T* Jc, const T* gains,
                       const int32_t* bwd_status, double tol, double max_reg,
                       int n_iterations, T* Z, T* U, T* gains_acc, T* J_opt,
                       double* mu, double* delta, int32_t* state,
                       int32_t* iter, uint8_t* active, uint8_t* fresh,
                       int32_t* n_live, void* stream) {
  if (A > kMaxAlphas) return PDDP_E_UNSUPPORTED;
  if (B <= 0 || N <= 0 || n <= 0 || m <= 0 || A <= 0 || !Zc || !Uc || !Jc ||
      !gains || !bwd_status || !Z || !U || !gains_acc || !J_opt || !mu ||
      !delta || !state || !iter || !active || !fresh)
    return PDDP_E_BADARG;
  AcceptArgs<T> a{B, N, n, m, A, Zc, Uc, Jc, gains, bwd_status, tol, max_reg,
                  n_iterations, Z, U, gains_acc, J_opt, mu, delta, state, iter,
                  active, fresh, n_live};
  PDDP_LAUNCH((accept_kernel<T>), dim3(B), dim3(kAcceptThreads), 0,
                     (hipStream_t)stream, a);
  return launch_status();
}

// --------------------------------------------------------------------------
// reference layout -> records
// --------------------------------------------------------------------------
template <typename T>
struct PackArgs {
  int B, N, n, m;
  const T *F_z, *F_u, *L_z, *L_u, *L_zz, *L_uz, *L_uu, *U;
  T* rec;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(PackArgs<T> a) {
  const RecLayout lay(a.n, a.m);
  const int n = a.n, m = a.m, N = a.N, S = lay.stride;
  const size_t total = (size_t)a.B * (N + 1) * S;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (size_t)gridDim.x * blockDim.x) {
    const size_t bt = o / S;
    const int w = (int)(o - bt * S);
    const size_t b = bt / (N + 1);
    const int t = (int)(bt - b * (N + 1));
    const bool term = (t == N);
    T v = T(0);
    if (w < lay.oLzz) {
      if (!term) v = a.F_z[(b * N + t) * n * n + (w - lay.oFz)];
    } else if (w < lay.oFu) {
      v = a.L_zz[(b * (N + 1) + t) * n * n + (w - lay.oLzz)];
    } else if (w < lay.oLuz) {
      if (!term) v = a.F_u[(b * N + t) * n * m + (w - lay.oFu)];
    } else if (w < lay.oLz) {
      if (!term) v = a.L_uz[(b * N + t) * m * n + (w - lay.oLuz)];
    } else if (w < lay.oLuu) {
      v = a.L_z[(b * (N + 1) + t) * n + (w - lay.oLz)];
    } else if (w < lay.oLu) {
      if (!term) v = a.L_uu[(b * N + t) * m * m + (w - lay.oLuu)];
    } else if (w < lay.oU) {
      if (!term) v = a.L_u[(b * N + t) * m + (w - lay.oLu)];
    } else if (w < lay.oU + m) {
      if (!term && a.U != nullptr) v = a.U[(b * N + t) * m + (w - lay.oU)];
    }
    a.rec[o] = v;
  }
}

template <typename T>
static int pack_impl(int B, int N, int n, int m, const T* F_z, const T* F_u,
                     const T* L_z, const T* L_u, const T* L_zz, const T* L_uz,
                     const T* L_uu, const T* U, T* rec, void* stream) {
  if (B <= 0 || N <= 0 || n <= 0 || m <= 0 || !F_z || !F_u || !L_z || !L_u ||
      !L_zz || !L_uz || !L_uu || !rec)
    return PDDP_E_BADARG;
  PackArgs<T> a{B, N, n, m, F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu, U, rec};
  const RecLayout lay(n, m);
  const size_t total = (size_t)B * (N + 1) * lay.stride;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  PDDP_LAUNCH((pack_kernel<T>), dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, a);
  return launch_status();
}

// J[b] = L[b][0] + ... + L[b][count-1] in index order: the trajectory cost of
// ilqr.py:484 (`L.sum()`) with a summation order that does not depend on the
// trajectory's position in the batch (a library row reduction does not
// promise that; duplicated trajectories must stay bit-identical)
template <typename T>
__global__ __launch_bounds__(256) void row_sum_kernel(int B, int count,
                                                      const T* L, T* J) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* row = L + (size_t)b * count;
  T acc = T(0);
  for (int t = 0; t < count; ++t) acc += row[t];
  J[b] = acc;
}

template <typename T>
static int row_sum_impl(int B, int count, const T* L, T* J, void* stream) {
  if (B <= 0 || count <= 0 || !L || !J) return PDDP_E_BADARG;
  PDDP_LAUNCH((row_sum_kernel<T>), dim3((B + 255) / 256), dim3(256), 0,
              (hipStream_t)stream, B, count, L, J);
  return launch_status();
}


// The record a rank contributes to the exchange of the best rollout
// (pddp_amd/parallel.py): out = {J_best, offset + index,
// Z[index][nz], U[index][nu]} with index = the first trajectory of least
// finite cost (torch.argmin over J with non-finite entries read as +inf).
// One launch instead of the eight small torch kernels of the same selection:
// with a round at 83 us the exchange has to cost a few microseconds.
constexpr int kPackBestThreads = 1024;
template <typename T>
__global__ __launch_bounds__(kPackBestThreads) void pack_best_kernel(
    int B, int nz, int nu, const T* J, const T* Z, const T* U, long long offset,
    T* out) {
  __shared__ T sv[kPackBestThreads];
  __shared__ int si[kPackBestThreads];
  const int tid = threadIdx.x;
  const T kInf = (T)__builtin_inff();
  T best = kInf;
  int bi = 0x7fffffff;
  for (int b = tid; b < B; b += kPackBestThreads) {
    const T v = J[b];
    const T key = is_finite(v) ? v : kInf;
    if (key < best || (key == best && b < bi)) {
      best = key;
      bi = b;
    }
  }
  sv[tid] = best;
  si[tid] = bi;
  __syncthreads();
  for (int w = kPackBestThreads / 2; w > 0; w >>= 1) {
    if (tid < w) {
      const T v = sv[tid + w];
      const int i = si[tid + w];
      if (v < sv[tid] || (v == sv[tid] && i < si[tid])) {
        sv[tid] = v;
        si[tid] = i;
      }
    }
    __syncthreads();
  }
  const int idx = si[0] < B ? si[0] : 0;
  if (tid == 0) {
    out[0] = sv[0];
    out[1] = (T)(offset + (long long)idx);
  }
  const T* zs = Z + (size_t)idx * nz;
  const T* us = U + (size_t)idx * nu;
  for (int o = tid; o < nz; o += kPackBestThreads) out[2 + o] = zs[o];
  for (int o = tid; o < nu; o += kPackBestThreads) out[2 + nz + o] = us[o];
}
template <typename T>
static int pack_best_impl(int B, int nz, int nu, const T* J, const T* Z,
                          const T* U, long long offset, T* out, void* stream) {
  if (B <= 0 || nz <= 0 || nu < 0 || !J || !Z || (nu > 0 && !U) || !out)
    return PDDP_E_BADARG;
  PDDP_LAUNCH((pack_best_kernel<T>), dim3(1), dim3(kPackBestThreads), 0,
              (hipStream_t)stream, B, nz, nu, J, Z, U, offset, out);
  return launch_status();
}

}  // namespace pddp

extern "C" {

int pddp_hip_abi_version(void) { return 1; }

int pddp_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* pddp_hip_arch(void) { return "gfx950"; }

int pddp_record_layout_of(int n, int m, pddp_record_layout* out) {
  if (n <= 0 || m <= 0 || out == nullptr) return PDDP_E_BADARG;
  const pddp::RecLayout l(n, m);
  out->n = n;
  out->m = m;
  out->o_Fz = l.oFz;
  out->o_Lzz = l.oLzz;
  out->o_Fu = l.oFu;
  out->o_Luz = l.oLuz;
  out->o_Lz = l.oLz;
  out->o_Luu = l.oLuu;
  out->o_Lu = l.oLu;
  out->o_U = l.oU;
  out->stride = l.stride;
  out->gain_stride = l.gstride;
  return 0;
}

int pddp_accept_f32(int B, int N, int n, int m, int A, const float* Zc,
                    const float* Uc, const float* Jc, const float* gains,
                    const int32_t* bwd_status, double tol, double max_reg,
                    int n_iterations, float* Z, float* U, float* gains_acc,
                    float* J_opt, double* mu, double* delta, int32_t* state,
                    int32_t* iter, uint8_t* active, uint8_t* fresh,
                    int32_t* n_live, void* stream) {
  return pddp::accept_impl<float>(B, N, n, m, A, Zc, Uc, Jc, gains, bwd_status,
                                  tol, max_reg, n_iterations, Z, U, gains_acc,
                                  J_opt, mu, delta, state, iter, active, fresh,
                                  n_live, stream);
}
int pddp_accept_f64(int B, int N, int n, int m, int A, const double* Zc,
                    const double* Uc, const double* Jc, const double* gains,
                    const int32_t* bwd_status, double tol, double max_reg,
                    int n_iterations, double* Z, double* U, double* gains_acc,
                    double* J_opt, double* mu, double* delta, int32_t* state,
                    int32_t* iter, uint8_t* active, uint8_t* fresh,
                    int32_t* n_live, void* stream) {
  return pddp::accept_impl<double>(B, N, n, m, A, Zc, Uc, Jc, gains,
                                   bwd_status, tol, max_reg, n_iterations, Z,
                                   U, gains_acc, J_opt, mu, delta, state, iter,
                                   active, fresh, n_live, stream);
}

int pddp_pack_records_f32(int B, int N, int n, int m, const float* F_z,
                          const float* F_u, const float* L_z, const float* L_u,
                          const float* L_zz, const float* L_uz,
                          const float* L_uu, const float* U, float* rec,
                          void* stream) {
  return pddp::pack_impl<float>(B, N, n, m, F_z, F_u, L_z, L_u, L_zz, L_uz,
                                L_uu, U, rec, stream);
}
int pddp_pack_records_f64(int B, int N, int n, int m, const double* F_z,
                          const double* F_u, const double* L_z,
                          const double* L_u, const double* L_zz,
                          const double* L_uz, const double* L_uu,
                          const double* U, double* rec, void* stream) {
  return pddp::pack_impl<double>(B, N, n, m, F_z, F_u, L_z, L_u, L_zz, L_uz,
                                 L_uu, U, rec, stream);
}

int pddp_sum_stage_costs_f32(int B, int count, const float* L, float* J,
                             void* stream) {
  return pddp::row_sum_impl<float>(B, count, L, J, stream);
}
int pddp_sum_stage_costs_f64(int B, int count, const double* L, double* J,
                             void* stream) {
  return pddp::row_sum_impl<double>(B, count, L, J, stream);
}

int pddp_pack_best_f32(int B, int nz, int nu, const float* J, const float* Z,
                       const float* U, long long offset, float* out,
                       void* stream) {
  return pddp::pack_best_impl<float>(B, nz, nu, J, Z, U, offset, out, stream);
}
int pddp_pack_best_f64(int B, int nz, int nu, const double* J, const double* Z,
                       const double* U, long long offset, double* out,
                       void* stream) {
  return pddp::pack_best_impl<double>(B, nz, nu, J, Z, U, offset, out, stream);
}

int pddp_event_create(void** ev) {
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) return (int)rc;
  *ev = (void*)e;
  return 0;
}
int pddp_event_record(void* ev, void* stream) {
  return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}
int pddp_event_elapsed_ms(void* start, void* stop, float* ms) {
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
int pddp_event_destroy(void* ev) {
  return (int)hipEventDestroy((hipEvent_t)ev);
}
int pddp_attach_events(void* start, void* stop) {
  pddp::launch_events() = {(hipEvent_t)start, (hipEvent_t)stop};
  return 0;
}

}  // extern "C"

// ---- a clock probe for tools/dbg/clock_after_kernels.py: shader cycles and
// ticks of the chip's constant 100 MHz clock over ~15 us of one sleeping
// wavefront - the shader clock the NEXT launch on the stream starts at
namespace pddp {
__global__ void clock_probe_kernel(long long* out) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);
  const long long c1 = clock64(), t1 = wall_clock64();
  out[0] = c1 - c0;
  out[1] = t1 - t0;
}
}  // namespace pddp
extern "C" int pddp_debug_clock_probe(long long* out, void* stream) {
  hipLaunchKernelGGL(pddp::clock_probe_kernel, dim3(1), dim3(64), 0,
                     (hipStream_t)stream, out);
  return pddp::launch_status();
}
