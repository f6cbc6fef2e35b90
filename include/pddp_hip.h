/*
 * pddp_hip.h - C ABI of libpddp_hip.so, the MI355X (gfx950) implementation of
 * the PDDP / iLQR data-parallel hot path.
 *
 * The reference (anassinator/pddp) is pure Python on torch 0.4.1 and has NO
 * native boundary; this ABI is what a native extension
 * of the reference would bind for its hot path.  Each entry point names the
 * reference function it replaces.  The Python plugin API on top
 * (pddp_amd.controllers / costs / models) mirrors the reference's classes and
 * calls these through ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host";
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered,
 *     nothing allocates, nothing synchronises the host, nothing throws;
 *   - return value: 0 on success, <0 PDDP_E_* for argument errors, >0 a
 *     hipError_t from the launch;
 *   - numerical failures the reference signals with RuntimeError
 *     (ilqr.py:608,639,653 and potrf at :595) are reported per trajectory in
 *     `status[b]` (PDDP_BWD_* of pddp_problem.h);
 *   - B independent trajectories (the batch axis the reference does not have:
 *     every trajectory is one reference controller call), N horizon,
 *     n encoded state size, m action size;
 *   - _f32 / _f64 suffix = arithmetic type of the whole path.
 *
 * Record layout (the HBM format the backward sweep streams; one record per
 * trajectory and time step, trajectory-major, N+1 records per trajectory, the
 * last one holding the terminal L_z, L_zz of ilqr.py:471-473):
 *
 *   rec[b][t][ F_z n*n | L_zz n*n | F_u n*m | L_uz m*n | L_z n | L_uu m*m |
 *              L_u m | U m | pad ]          stride = roundup4(total) scalars
 *
 * (cartpole n=4,m=1: 47 -> 48 floats = 192 B.)  `U` is the un-clamped nominal
 * action (ilqr.py:602,647 read it for the BoxQP bounds).  Gains come back as
 *
 *   gains[b][t][ k m | K m*n ]              (ilqr.py:581-582 k, K)
 */
#ifndef PDDP_HIP_H
#define PDDP_HIP_H

#include <stdint.h>

#include "pddp_problem.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  PDDP_E_BADARG = -1,      /* null pointer / non-positive size */
  PDDP_E_UNSUPPORTED = -2, /* (n, m, model, encoding) not built */
  PDDP_E_NODEVICE = -3
};

typedef struct pddp_record_layout {
  int n, m;
  int o_Fz, o_Lzz, o_Fu, o_Luz, o_Lz, o_Luu, o_Lu, o_U;
  int stride;      /* scalars per record */
  int gain_stride; /* m + m*n */
} pddp_record_layout;

/* Library / device probes (host). */
int pddp_hip_abi_version(void);
int pddp_hip_device_count(void);
const char* pddp_hip_arch(void); /* "gfx950" */
int pddp_record_layout_of(int n, int m, pddp_record_layout* out /* host */);

/* ---- ilqr.py:529-674 backward(): the Riccati sweep --------------------- */
/* rec    [B][N+1][stride]   records (see above)
 * u_min, u_max [m] or both NULL -> branch A/C (no BoxQP), else B/D
 * reg    [B] per-trajectory mu (ilqr.py:135 passes self._mu), host-side type
 *        double like the reference's python float
 * branch PDDP_BRANCH_EIG (controller default, V_zz_reg=False) or _CHOLESKY
 * active [B] nullable; trajectories with active[b]==0 are skipped: their
 *        status is left untouched, their gains rows are unspecified
 * gains  [B][N][m + m*n] out (unspecified from the failing step on when
 *        status[b] != 0, like the reference which raises)
 * status [B] out (PDDP_BWD_*) */
int pddp_riccati_backward_f32(int B, int N, int n, int m, const float* rec,
                              const float* u_min, const float* u_max,
                              const double* reg, int branch,
                              const uint8_t* active, float* gains,
                              int32_t* status, void* stream);
int pddp_riccati_backward_f64(int B, int N, int n, int m, const double* rec,
                              const double* u_min, const double* u_max,
                              const double* reg, int branch,
                              const uint8_t* active, double* gains,
                              int32_t* status, void* stream);

/* The same sweep through a chosen kernel variant (A/B measurements):
 * 0 auto (what the entry points above use), 1 generic kernel (one wavefront
 * per trajectory, any n <= 32, m <= 4; larger n: four wavefronts), and for
 * n = 4, m = 1:
 *   6 / 7   sixteen lanes per trajectory, BoxQP in closed form with the
 *           reference's loop as fall-back (IEEE division / v_rcp + v_sqrt,
 *           f32 only for the odd numbers throughout);
 *   16 / 17 four lanes per trajectory, sixteen trajectories per wavefront
 *           (all four branches); 18 = 16 with every BoxQP through the
 *           reference's loop (bounded branches);
 * 14 / 15: the matrix-core kernels for n <= 30, m = 1 (IEEE / approximate
 * division; auto for those shapes other than n = 4); f64: n <= 14 on the f64
 * matrix cores, variant 14 only (15 and n > 14: PDDP_E_UNSUPPORTED; auto then
 * takes the generic kernel).
 * 26 / 27 (IEEE / approximate division): 15 <= n <= 30, m = 1, f32, eig-clamp
 * branches with one trajectory's step split over two wavefronts; auto for
 * those shapes and branches: 27.
 * Auto for n = 4, m = 1: f32 from 12288 trajectories on -> 17, otherwise 7
 * (f32) / 6 (f64).  Any other number: PDDP_E_BADARG (rounds 1-4 carried 2 / 3,
 * 8 - 13, 20 - 25: other formulations of the n = 4 sweep ON RECORDS, retired
 * once the cartpole's rounds took their sweep from the nominal,
 * pddp_sweep_nominal_* / pddp_round_nominal_f32). */
int pddp_riccati_backward_variant_f32(int B, int N, int n, int m,
                                      const float* rec, const float* u_min,
                                      const float* u_max, const double* reg,
                                      int branch, const uint8_t* active,
                                      float* gains, int32_t* status,
                                      void* stream, int variant);
int pddp_riccati_backward_variant_f64(int B, int N, int n, int m,
                                      const double* rec, const double* u_min,
                                      const double* u_max, const double* reg,
                                      int branch, const uint8_t* active,
                                      double* gains, int32_t* status,
                                      void* stream, int variant);

/* ---- utils/constraint.py:150-266 boxqp() for one action dimension -------- */
/* `count` independent scalar QPs  min 0.5 Q x^2 + c x  s.t. lower <= x <= upper,
 * warm-started at x0: the projected-Newton loop of the reference with its
 * exit codes (`result`, constraint.py:23-32) and its `free` flag, exactly the
 * device routine the sweep calls.  All arrays [count]. */
int pddp_boxqp_m1_f32(int count, const float* x0, const float* Q,
                      const float* c, const float* lower, const float* upper,
                      float* x, int32_t* result, uint8_t* free_mask,
                      void* stream);
int pddp_boxqp_m1_f64(int count, const double* x0, const double* Q,
                      const double* c, const double* lower,
                      const double* upper, double* x, int32_t* result,
                      uint8_t* free_mask, void* stream);

/* The scalar BoxQP AS THE BENCHED SWEEP RUNS IT (csrc/riccati_n4_elem.hpp
 * elem_gains: ilqr.py:633-634 e = (Quu < 0 ? 1e-12 : Quu) + reg, then
 * constraint.py:150-266 on (x0, e, Qu, lower, upper) in the lean closed form
 * with v_rcp_f32: every exit test of the loop's two passes and its `free`
 * flag; not its Armijo back-tracking, which for one action returns the same
 * clamped Newton point in exact arithmetic - riccati_n4_elem.hpp QpLean1 has
 * the argument and the measured agreement; a curvature that is not positive
 * and finite goes to the closed form of pddp_boxqp_m1_* and the reference's
 * loop behind it): x = the feed-forward gain,
 * free_mask = 1 where the feedback row is not zeroed (the reference's possibly
 * stale `free`), status = PDDP_BWD_OK / _NAN / _BOXQP_FAILED; coeffs, nullable,
 * [count][3] = {s, c, w}: 1 / e or 0, and the two coefficients of the rank-one
 * value update V' = sym(Qzz) + c Quz Quz^T, V_z' = Qz + w Quz (ilqr.py:664-672
 * with K = -s Quz).  The unit-test entry of that routine. */
int pddp_boxqp_m1_lean_f32(int count, const float* x0, const float* Quu,
                           const float* Qu, const float* reg,
                           const float* lower, const float* upper, float* x,
                           uint8_t* free_mask, int32_t* status, float* coeffs,
                           void* stream);

/* ---- utils/constraint.py:150-266 boxqp() for m <= 4 action dimensions ------ */
/* `count` independent QPs  min 0.5 x^T Q x + c^T x  s.t. lower <= x <= upper,
 * warm-started at x0; the routine the generic sweep calls (csrc/gains.hpp),
 * one lane per problem.  x0, c, lower, upper, x [count][m]; Q [count][m][m];
 * result [count] (constraint.py:23-32); free_mask [count][m] (1 = free, the
 * possibly stale set of constraint.py:191-204); Ufree [count][m][m] = upper
 * Cholesky factor of the free block with identity rows / columns in place of
 * the clamped dimensions (the reference returns the compacted block).
 * PDDP_E_UNSUPPORTED for m > 4. */
int pddp_boxqp_f32(int count, int m, const float* x0, const float* Q,
                   const float* c, const float* lower, const float* upper,
                   float* x, int32_t* result, float* Ufree,
                   uint8_t* free_mask, void* stream);
int pddp_boxqp_f64(int count, int m, const double* x0, const double* Q,
                   const double* c, const double* lower, const double* upper,
                   double* x, int32_t* result, double* Ufree,
                   uint8_t* free_mask, void* stream);

/* Packs reference-layout tensors (what ilqr.py:393-486 forward() returns,
 * with a leading batch axis) into records, for plugin models whose
 * derivatives come from elsewhere.  F_z [B][N][n][n], F_u [B][N][n][m],
 * L_z [B][N+1][n], L_u [B][N][m], L_zz [B][N+1][n][n], L_uz [B][N][m][n],
 * L_uu [B][N][m][m], U [B][N][m] (nullable -> zeros). */
int pddp_pack_records_f32(int B, int N, int n, int m, const float* F_z,
                          const float* F_u, const float* L_z, const float* L_u,
                          const float* L_zz, const float* L_uz,
                          const float* L_uu, const float* U, float* rec,
                          void* stream);
int pddp_pack_records_f64(int B, int N, int n, int m, const double* F_z,
                          const double* F_u, const double* L_z,
                          const double* L_u, const double* L_zz,
                          const double* L_uz, const double* L_uu,
                          const double* U, double* rec, void* stream);

/* Trajectory cost J[b] = sum_t L[b][t] (ilqr.py:484 `L.sum()`), summed in t
 * order by one lane per trajectory: the result does not depend on the
 * trajectory's position in the batch.  L [B][count], J [B]. */
int pddp_sum_stage_costs_f32(int B, int count, const float* L, float* J,
                             void* stream);
int pddp_sum_stage_costs_f64(int B, int count, const double* L, double* J,
                             void* stream);

/* ---- ilqr.py:393-486 forward(): derivative rollout for the sample
 * problems (analytic; replaces utils/evaluation.py:134-288 autograd) ------ */
/* Nominal rollout Z[b][0] = z0[b], Z[b][t+1] = model(Z[b][t], clamp(U[b][t]))
 * (ilqr.py:457-468).  problem: host pointer, copied by value into the launch.
 * z0 [B][n]; U [B][N][m]; u_min/u_max [m] nullable (ilqr.py:461-462);
 * mask [B] nullable (0 -> trajectory skipped); Z [B][N+1][n] out. */
int pddp_nominal_rollout_f32(const pddp_problem* problem, int B, int N,
                             const float* z0, const float* U,
                             const float* u_min, const float* u_max,
                             const uint8_t* mask, float* Z, void* stream);
int pddp_nominal_rollout_f64(const pddp_problem* problem, int B, int N,
                             const double* z0, const double* U,
                             const double* u_min, const double* u_max,
                             const uint8_t* mask, double* Z, void* stream);

/* Derivative records along a given nominal (Z, U): F_z, F_u, L_z .. L_uu at
 * every (b, t) (ilqr.py:464-473), parallel over b AND t.
 * rec [B][N+1][stride] out; L [B][N+1] out; J [B] out = L.sum() (ilqr.py:209);
 * state [B] nullable: set to UNDEFINED for computed trajectories
 * (ilqr.py:212). */
int pddp_derivs_f32(const pddp_problem* problem, int B, int N, const float* Z,
                    const float* U, const float* u_min, const float* u_max,
                    const uint8_t* mask, float* rec, float* L, float* J,
                    int32_t* state, void* stream);
int pddp_derivs_f64(const pddp_problem* problem, int B, int N, const double* Z,
                    const double* U, const double* u_min, const double* u_max,
                    const uint8_t* mask, double* rec, double* L, double* J,
                    int32_t* state, void* stream);

/* ---- ilqr.py:677-723 _control_law() + :764-791 _trajectory_cost() ------- */
/* A candidate step sizes per trajectory.  Z [B][N+1][n], U [B][N][m] nominal;
 * gains [B][N][m+m*n]; alphas [A]; bwd_status [B] nullable (non-zero ->
 * skipped, the reference never reaches the line search then, ilqr.py:140-145);
 * Zc [B][N+1][A][n], Uc [B][N][A][m] candidates out (time-major, the
 * reference's Z_new / U_new per trajectory: the A rollouts of a trajectory
 * write one contiguous segment per step), Jc [B][A] out. */
int pddp_line_search_f32(const pddp_problem* problem, int B, int N, int A,
                         const float* Z, const float* U, const float* gains,
                         const float* alphas, const float* u_min,
                         const float* u_max, const uint8_t* active,
                         const int32_t* bwd_status, float* Zc, float* Uc,
                         float* Jc, void* stream);
int pddp_line_search_f64(const pddp_problem* problem, int B, int N, int A,
                         const double* Z, const double* U, const double* gains,
                         const double* alphas, const double* u_min,
                         const double* u_max, const uint8_t* active,
                         const int32_t* bwd_status, double* Zc, double* Uc,
                         double* Jc, void* stream);

/* ---- ilqr.py:102-181 _step() accept / reject, :364-390 mu schedule and the
 * fit() loop bookkeeping (:298-314), per trajectory, device resident -------- */
/* Controller state arrays (all [B]):
 *   J_opt (T), mu, delta (double, the reference's python floats), state
 *   (iLQRState), iter (number of step() calls started, ilqr.py:298),
 *   active (attempted this round), fresh (needs new derivatives next round).
 * For each trajectory with active[b] != 0:
 *   bwd_status != 0 -> _increase_reg, NOT_PD / MAX_REG          (:140-145)
 *   else amin = argmin Jc[b], accept iff J_new < J_opt           (:161-166)
 *     accept: Z, U <- candidate amin; gains_acc <- gains (self._K);
 *             _decrease_reg; CONVERGED iff |dJ|/J_opt < tol      (:167-176)
 *     reject: _increase_reg, REJECTED / MAX_REG                  (:178-181)
 * then the masks of the NEXT round: retry states keep active=1, fresh=0;
 * ACCEPTED starts the next step() (iter+1, fresh=1) unless iter == n_iterations;
 * CONVERGED / MAX_REG leave the loop (:313).  n_live (device
 * int32[PDDP_LIVE_SHARDS], nullable) accumulates, sharded by b mod
 * PDDP_LIVE_SHARDS, the number of trajectories still active after this round
 * (sum the shards; one word would serialise 4096 atomics). */
#define PDDP_LIVE_SHARDS 256
int pddp_accept_f32(int B, int N, int n, int m, int A, const float* Zc,
                    const float* Uc, const float* Jc, const float* gains,
                    const int32_t* bwd_status, double tol, double max_reg,
                    int n_iterations, float* Z, float* U, float* gains_acc,
                    float* J_opt, double* mu, double* delta, int32_t* state,
                    int32_t* iter, uint8_t* active, uint8_t* fresh,
                    int32_t* n_live, void* stream);
int pddp_accept_f64(int B, int N, int n, int m, int A, const double* Zc,
                    const double* Uc, const double* Jc, const double* gains,
                    const int32_t* bwd_status, double tol, double max_reg,
                    int n_iterations, double* Z, double* U, double* gains_acc,
                    double* J_opt, double* mu, double* delta, int32_t* state,
                    int32_t* iter, uint8_t* active, uint8_t* fresh,
                    int32_t* n_live, void* stream);

/* ---- multi-GPU exchange: the record a rank contributes to the
 * all-gather of the best rollout, out[2 + nz + nu] = {J_best, offset + index,
 * Z[index][nz], U[index][nu]}, index = the first trajectory of least finite
 * J (non-finite costs count as +inf; all non-finite: index 0, J_best = inf).
 * J [B], Z [B][nz], U [B][nu]; one launch, no host synchronisation. */
int pddp_pack_best_f32(int B, int nz, int nu, const float* J, const float* Z,
                       const float* U, long long offset, float* out,
                       void* stream);
int pddp_pack_best_f64(int B, int nz, int nu, const double* J, const double* Z,
                       const double* U, long long offset, double* out,
                       void* stream);

/* ---- the backward sweep of a sample problem FROM ITS NOMINAL TRAJECTORY: the
 * derivative records (ilqr.py:464-473: F_z, F_u, L_z ... of every step) are
 * evaluated inside the sweep's workgroups, in LDS, and never written - the
 * 79 MB per launch that pddp_derivs_* / pddp_search_accept_* write and
 * pddp_riccati_backward_* reads at B = 4096, N = 100 stay on the chip.
 * Results as pddp_derivs_* followed by pddp_riccati_backward_* (auto kernel):
 * gains [B][N][5], status [B]; L [B][N+1] the stage / terminal costs of the
 * nominal (rows of ACTIVE trajectories);
 * for trajectories with fresh[b] != 0 (all, when fresh is NULL)
 * J_opt[b] = sum_t L[b][t] in t order (ilqr.py:289 L.sum()) and fresh[b] is
 * cleared.  Z [B][N+1][n], U [B][N] un-clamped nominal actions.  Under
 * IGNORE_UNCERTAINTY:
 *   cartpole          f32, bounded (u_min, u_max non-NULL), PDDP_BRANCH_EIG
 *                     (csrc/riccati_n4_elem.hpp);
 *   pendulum, double cartpole   f32 and f64, both branches, bounded or not
 *                     (csrc/riccati_mfma16_nominal.hpp: the 16 x 16
 *                     matrix-core sweep, its records generated block by block
 *                     in the wavefront);
 * PDDP_E_UNSUPPORTED otherwise (make the two calls then).  `L` of
 * pddp_search_accept_* may be NULL with this sweep. */
int pddp_sweep_nominal_f32(const pddp_problem* problem, int B, int N,
                           const float* Z, const float* U, const float* u_min,
                           const float* u_max, const double* reg, int branch,
                           const uint8_t* active, uint8_t* fresh, float* gains,
                           int32_t* status, float* L, float* J_opt,
                           void* stream);
int pddp_sweep_nominal_f64(const pddp_problem* problem, int B, int N,
                           const double* Z, const double* U, const double* u_min,
                           const double* u_max, const double* reg, int branch,
                           const uint8_t* active, uint8_t* fresh, double* gains,
                           int32_t* status, double* L, double* J_opt,
                           void* stream);

/* How pddp_sweep_nominal_f32 (csrc/riccati_n4_elem.hpp) generates its records:
 * 0 = auto (on wavefronts of their own while one workgroup per CU holds the
 * batch, inline beyond), 3 = inline, 4 = on wavefronts of their own; results
 * are bit-identical.  Other values are ignored.  Process-wide (an A/B and
 * test knob); -1 only queries.  Returns the previous choice. */
int pddp_sweep_nominal_kernel(int which);

/* ---- one launch for the rest of a round: the line search, the accept step
 * and the derivative records of the trajectories whose nominal changed and
 * whose fit goes on.  Same arguments and semantics as the three calls; Z, U, active are
 * in/out; `fresh` is cleared for the trajectories whose records were written
 * here.  Sample problems with at most 16 step sizes; returns
 * PDDP_E_UNSUPPORTED otherwise (make the three calls then).  L == NULL: no
 * records are written and `fresh` stays set - the caller's next sweep is
 * pddp_sweep_nominal_*, which needs none; `rec`, when not NULL, is then
 * scratch of at least B (N+1) n scalars: the states of every trajectory's
 * full-step candidate go there, rows next to one another, INSTEAD of
 * Zc[b][.][0][.] (the usual winner: its copy into the nominal reads whole
 * sectors there, 16 bytes out of every A n 4-byte step in Zc).  In that form
 * (L == NULL, rec given, N + 1 within the tail's short form: 128 steps for the
 * paired launch) the candidates' ACTIONS `Uc` are not written either: the
 * winner's are its control law at its states, re-evaluated bit for bit. */
/* Candidates of pddp_search_accept_* in its form without records (L == NULL,
 * rec given as scratch): 0 = auto - kept while B A (N (n + m) + n) scalars fit
 * the Infinity Cache next to the round's other traffic (200 MB), dropped
 * beyond; 1 = always kept; 2 = always dropped.  Dropped: Zc / Uc are scratch
 * (contents unspecified after the call), only the costs Jc are formed; the
 * full step's states go to `rec` as above, a winner other than the full step
 * is rolled out a second time.  Every other output is the same to rounding.
 * Process-wide (an A/B and test knob); -1 only queries.  Returns the previous
 * mode. */
int pddp_search_candidates(int mode);
/* Which form of pddp_search_accept_* runs (f32, n <= 4): 0 = auto - the paired
 * form (a helper wavefront per rollout wavefront, the nominal's 4 KB per
 * trajectory in LDS, two workgroups per CU), from 8193 to 49152 trajectories
 * the dense form (gains only in LDS, no helpers, four workgroups per CU); 1 =
 * always the paired form; 2 = the dense form wherever it is built.  Results are
 * bit-identical.  Process-wide (an A/B and test knob); -1 only queries.
 * Returns the previous mode. */
int pddp_search_form(int mode);

int pddp_search_accept_f32(const pddp_problem* problem, int B, int N, int A,
                           float* Z, float* U, const float* gains,
                           const float* alphas, const float* u_min,
                           const float* u_max, uint8_t* active,
                           const int32_t* bwd_status, float* Zc, float* Uc,
                           float* Jc, double tol, double max_reg,
                           int n_iterations, float* gains_acc, float* J_opt,
                           double* mu, double* delta, int32_t* state,
                           int32_t* iter, uint8_t* fresh, int32_t* n_live,
                           float* rec, float* L, void* stream);
int pddp_search_accept_f64(const pddp_problem* problem, int B, int N, int A,
                           double* Z, double* U, const double* gains,
                           const double* alphas, const double* u_min,
                           const double* u_max, uint8_t* active,
                           const int32_t* bwd_status, double* Zc, double* Uc,
                           double* Jc, double tol, double max_reg,
                           int n_iterations, double* gains_acc, double* J_opt,
                           double* mu, double* delta, int32_t* state,
                           int32_t* iter, uint8_t* fresh, int32_t* n_live,
                           double* rec, double* L, void* stream);

/* ---- a whole round in ONE launch: pddp_sweep_nominal_f32 followed by
 * pddp_search_accept_f32(L = NULL) for the same arguments, the workgroups
 * going straight from their sweep to their line search (csrc/round_n4.hip:
 * the gains, the nominal's rows, its cost and the sweep's status stay in LDS /
 * registers between the phases; no second launch, no second prologue).  Results
 * are the two calls' bit for bit (replaces ilqr.py:125-181 of one attempt:
 * backward, _control_law, _trajectory_cost, accept / reject, mu schedule).
 * (of the candidates, `Zc` and the costs `Jc` are written, `Uc` is NOT: the
 * winner's actions are its control law at its states, re-evaluated.)
 * `mu` is both the sweep's `reg` and the schedule's state; `scratch` as `rec`
 * of pddp_search_accept_* with L == NULL (B (N+1) n scalars).  Cartpole under
 * IGNORE_UNCERTAINTY, f32, bounded, PDDP_BRANCH_EIG, A <= 16, N <= 127 and at
 * most 4096 trajectories (one workgroup of 16 per CU); PDDP_E_UNSUPPORTED
 * otherwise (make the two calls then).
 * `rounds` >= 1: that many attempts of every trajectory in the one launch,
 * exactly as `rounds` calls with rounds = 1 (trajectories are independent and
 * a workgroup owns its sixteen for the whole launch; one that has left the fit
 * - active[b] == 0 - is skipped, as by a later call).
 * `phase_ticks`, nullable: [ceil(B / 16)][2] counters to which every workgroup
 * ADDS the ticks of the chip's 100 MHz clock it spent in its sweeps and in its
 * searches (rocprofv3 sees one kernel; bench.py's roofline leg wants the
 * sweep's share). */
int pddp_round_nominal_f32(const pddp_problem* problem, int B, int N, int A,
                           float* Z, float* U, const float* alphas,
                           const float* u_min, const float* u_max, int branch,
                           uint8_t* active, uint8_t* fresh, float* gains,
                           int32_t* bwd_status, float* L, float* J_opt,
                           float* Zc, float* Uc, float* Jc, double tol,
                           double max_reg, int n_iterations, float* gains_acc,
                           double* mu, double* delta, int32_t* state,
                           int32_t* iter, int32_t* n_live, float* scratch,
                           int rounds, long long* phase_ticks, void* stream);

/* The variant entry with two HIP events (pddp_event_create) attached to the
 * sweep's own dispatch: elapsed(start, stop) is the kernel's duration as
 * rocprofv3 --kernel-trace reports it (bench.py's roofline leg). */
int pddp_riccati_backward_timed_f32(int B, int N, int n, int m,
                                    const float* rec, const float* u_min,
                                    const float* u_max, const double* reg,
                                    int branch, const uint8_t* active,
                                    float* gains, int32_t* status,
                                    void* stream, int variant, void* start,
                                    void* stop);
int pddp_riccati_backward_timed_f64(int B, int N, int n, int m,
                                    const double* rec, const double* u_min,
                                    const double* u_max, const double* reg,
                                    int branch, const uint8_t* active,
                                    double* gains, int32_t* status,
                                    void* stream, int variant, void* start,
                                    void* stop);

/* Arithmetic of the hidden-to-hidden contraction (layer 2) of pddp_bnn_mlp_* /
 * pddp_bnn_mlp_jvp_*: 0 = exact f32 on v_mfma_f32_32x32x2_f32 (default; bitwise
 * an fmaf chain), 3 = the bf16-split twin - weights and activations as three
 * bf16 parts, six v_mfma_f32_32x32x16_bf16 per product: f32 accuracy to
 * rounding (not bit-exact), 2.5 times fewer matrix cycles; H = 200 only, other
 * widths stay exact.  Process-wide; -1 only queries.  Returns the previous
 * mode (PDDP_MLP_BF16X3=1 in the environment makes 3 the initial one). */
int pddp_bnn_mlp_precision(int mode);

/* How the exact-f32 network kernel deals its layer-2 contraction out over a
 * workgroup's wavefronts at H = 200 (csrc/bnn_mlp.hip): 0 = every
 * block its own, 1 = round 2's balanced roles, 2 = round 5's (the last block's
 * 8 units on 16 x 16 x 4 tiles, three chunks handed over), -1 = the default:
 * 2 for inference, 1 for forward mode.  Another deal is another summation
 * order - the same numbers to rounding.  Process-wide; -2 only queries.
 * Returns the previous value (PDDP_MLP_BALANCED in the environment sets the
 * initial one). */
int pddp_bnn_mlp_deal(int deal);

/* ---- pddp/models/bnn/modules.py:774-864: the Bayesian network of the learned
 * dynamics model, fused (fc -> dropout mask -> ReLU, twice, fc_out), float:
 *   Y[r] = W3 relu(M2[p] * (W2 relu(M1[p] * (W1 X[r] + b1)) + b2)) + b3,
 * p = r % P the particle of row r (rows = states x particles, particle
 * fastest).  X [R][in_dim], W1 [H][in_dim], W2 [H][H], W3 [out_dim][H]
 * (torch.nn.Linear layout); M1, M2 [P][H] the dropout masks of the two hidden
 * layers as the framework holds them (all ones: no dropout), rows 16-byte
 * aligned (H a multiple of 4); Y [R][out_dim].  in_dim <= 15, out_dim <= 16, H in {64, 128,
 * 200}; PDDP_E_UNSUPPORTED otherwise (use the library GEMMs then). */
int pddp_bnn_mlp_f32(int R, int P, int in_dim, int H, int out_dim,
                     const float* X, const float* W1, const float* b1,
                     const float* M1, const float* W2, const float* b2,
                     const float* M2, const float* W3, const float* b3,
                     float* Y, void* stream);
/* The same on the first min(R, *live_rows) rows only (live_rows: one int32 on
 * the device, read by the kernel - no host synchronisation; nullable = R): the
 * launch is sized for R, workgroups without a tile leave at once.  The rows
 * beyond are neither read nor written.  For a line search whose live
 * candidates are packed to the front (pddp_bnn_step.slot): a round of retries
 * with three of 256 restarts alive runs the network on 3 / 256 of the rows. */
int pddp_bnn_mlp_rows_f32(int R, int P, int in_dim, int H, int out_dim,
                          const float* X, const float* W1, const float* b1,
                          const float* M1, const float* W2, const float* b2,
                          const float* M2, const float* W3, const float* b3,
                          float* Y, const int32_t* live_rows, void* stream);
/* The same network in double precision (modules.py runs in the dtype of its
 * inputs) on v_mfma_f64_16x16x4_f64, weights-stationary, four wavefronts per
 * CU (csrc/bnn_mlp_f64.hip); arguments as above with double data.  M1, M2:
 * 32-byte aligned (their rows then are: H is a multiple of 4). */
int pddp_bnn_mlp_f64(int R, int P, int in_dim, int H, int out_dim,
                     const double* X, const double* W1, const double* b1,
                     const double* M1, const double* W2, const double* b2,
                     const double* M2, const double* W3, const double* b3,
                     double* Y, void* stream);
int pddp_bnn_mlp_rows_f64(int R, int P, int in_dim, int H, int out_dim,
                          const double* X, const double* W1, const double* b1,
                          const double* M1, const double* W2, const double* b2,
                          const double* M2, const double* W3, const double* b3,
                          double* Y, const int32_t* live_rows, void* stream);

/* ---- one time step of the moment-matched line-search rollout under a BNN
 * dynamics model, everything but the network: ilqr.py:677-723 (_control_law),
 * :764-791 (_trajectory_cost), modules.py:287-386 (BNNDynamicsModel.forward:
 * particles in, particle moments out), encoding.py:99-141 (encode, DEFAULT) and
 * angular.py:47-84,161-248 (angle augmentation of the moments for the cost).
 * Called N + 1 times (t = 0 .. N) with the network (above) on F in between:
 *   t > 0:  Xp += net_out[:, :D] * dX_std + dX_mean   (the particle cloud is
 *           carried: `infer_noise_variables` re-whitens and re-colours with the
 *           same factor), z_t = encode(mean, covariance of Xp); t = 0: z_0 =
 *           Z[b][0], Xp as given (sampled by the caller, modules.py:312-330);
 *   Zc[b][t][a] = z_t;  t < N: Uc[b][t][a] = u_t = clamp(U + alpha k + K (z_t -
 *           Z)), J += l(z_t, u_t), F = network input rows of the particles;
 *   t = N:  J += l_f(z_N), Jc = J.
 * Candidate c = b * A + a; particle rows c * P + p.  DEFAULT encoding
 * (n = D + D (D + 1) / 2), D <= 8, at most two angular states, m <= 2,
 * P <= 128; the cost is a QR cost on the augmented moments. */
typedef struct pddp_bnn_step {
  int32_t B, A, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8]; /* angular / non-angular state indices */
  int32_t in_dim, out_dim;              /* network: n_non + 2 n_ang + m, >= D */
  const float* Z;        /* [B][N+1][n] nominal */
  const float* U;        /* [B][N][m] */
  const float* gains;    /* [B][N][m + m n] */
  const float* alphas;   /* [A] */
  const float* u_min;    /* [m], nullable with u_max */
  const float* u_max;
  const uint8_t* active;       /* [B] nullable */
  const int32_t* bwd_status;   /* [B] nullable: non-zero -> skipped */
  const float* Q;        /* [na][na], na = n_non + 2 n_ang */
  const float* Q_term;
  const float* R;        /* [m][m] */
  const float* x_goal;   /* [na] */
  const float* u_goal;   /* [m] */
  const float* X_mean;   /* [in_dim] input normalisation (modules.py:181-186) */
  const float* X_std_inv;
  const float* dX_mean;  /* [D] output de-normalisation (modules.py:262) */
  const float* dX_std;
  const float* net_out;  /* [B A P][out_dim]: the network's output of step t-1 */
  float* Xp;             /* [B A][P][D] particles, in / out */
  float* F;              /* [B A P][in_dim] network input, out (t < N) */
  float* Zc;             /* [B][N+1][A][n] out */
  float* Uc;             /* [B][N][A][m] out */
  float* J;              /* [B A] running cost, in / out */
  float* Jc;             /* [B A] out at t = N */
  /* use_predicted_std (modules.py:242-262): the standardised normals of step
   * t - 1, [P][D]; then X_t += exp(net_out[:, D + d] + log dX_std[d]) *
   * eps_out[p][d] and out_dim >= 2 D.  NULL: the predicted std is not used. */
  const float* eps_out;
  /* [B] nullable: the rank of trajectory b among the live ones (active and
   * bwd_status == 0; anything for the others).  Given, the network-facing rows
   * - F and net_out - of candidate (b, a) are (slot[b] A + a) P + p instead of
   * (b A + a) P + p: the network then runs on the first (live count) A P rows
   * only (pddp_bnn_mlp_rows_f32).  The other arrays keep their indexing. */
  const int32_t* slot;
} pddp_bnn_step;
int pddp_bnn_moment_step_f32(const pddp_bnn_step* step, void* stream);
/* double precision: the same fields, double data */
typedef struct pddp_bnn_step_f64 {
  int32_t B, A, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8];
  int32_t in_dim, out_dim;
  const double* Z;
  const double* U;
  const double* gains;
  const double* alphas;
  const double* u_min;
  const double* u_max;
  const uint8_t* active;
  const int32_t* bwd_status;
  const double* Q;
  const double* Q_term;
  const double* R;
  const double* x_goal;
  const double* u_goal;
  const double* X_mean;
  const double* X_std_inv;
  const double* dX_mean;
  const double* dX_std;
  const double* net_out;
  double* Xp;
  double* F;
  double* Zc;
  double* Uc;
  double* J;
  double* Jc;
  const double* eps_out;
  const int32_t* slot;
} pddp_bnn_step_f64;
int pddp_bnn_moment_step_f64(const pddp_bnn_step_f64* step, void* stream);

/* ---- the same network in forward mode (JVP), for the derivative rollout
 * (ilqr.py:457-468 -> utils/evaluation.py:203-235 batch_eval_dynamics, which
 * replicates the input n times and back-propagates an identity): rows come in
 * groups of `group` = 8, 16 or 32 = one (state, particle) input row followed
 * by group - 1 tangent rows d X / d direction; a tangent row passes through the
 * weights without biases and through the ReLUs linearised at its group's first
 * row.  p = (r / group) % P, R a multiple of group; everything else as
 * pddp_bnn_mlp_f32. */
int pddp_bnn_mlp_jvp_f32(int R, int P, int group, int in_dim, int H,
                         int out_dim, const float* X, const float* W1,
                         const float* b1,
                         const float* M1, const float* W2, const float* b2,
                         const float* M2, const float* W3, const float* b3,
                         float* Y, void* stream);
/* The same with only the first `live` <= group rows of every group in use (the
 * input row and the 1 + D + m - 1 tangent rows that exist): the other rows are
 * neither read nor written, and the kernel packs 32 / live whole groups into a
 * tile instead of 32 / group (group = 8, live <= 4: eight, live <= 6: five). */
int pddp_bnn_mlp_jvp_live_f32(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const float* X,
                              const float* W1, const float* b1, const float* M1,
                              const float* W2, const float* b2, const float* M2,
                              const float* W3, const float* b3, float* Y,
                              void* stream);
/* The same on the first min(R, *live_rows) rows (a device scalar, nullable =
 * R; whole groups): see pddp_bnn_mlp_rows_f32 and pddp_bnn_jvp.slot. */
int pddp_bnn_mlp_jvp_rows_f32(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const float* X,
                              const float* W1, const float* b1, const float* M1,
                              const float* W2, const float* b2, const float* M2,
                              const float* W3, const float* b3, float* Y,
                              const int32_t* live_rows, void* stream);
/* double precision (group = 8 or 16; 32: PDDP_E_UNSUPPORTED) */
int pddp_bnn_mlp_jvp_rows_f64(int R, int P, int group, int live, int in_dim,
                              int H, int out_dim, const double* X,
                              const double* W1, const double* b1,
                              const double* M1, const double* W2,
                              const double* b2, const double* M2,
                              const double* W3, const double* b3, double* Y,
                              const int32_t* live_rows, void* stream);

/* ---- Jacobians F_z, F_u of one moment-matched BNN step (modules.py:287-386
 * under DEFAULT encoding) in forward mode, around pddp_bnn_mlp_jvp_f32:
 *   features(t): eps = (Xp - mean_t) U_t^-1 (the detached re-whitening of
 *                modules.py:333-348), F = [B P][8][in_dim]: the input row, the
 *                D + m tangent rows for the directions (mean_d | u) at the
 *                clamped action, zero rows.  The Cholesky directions U_ab need
 *                no rows of their own: d X / d U_ab = eps[a] d X / d mean_b per
 *                particle, and the network pass is linear in the tangent;
 *   moments(t):  net_out [B P][8][out_dim] -> Xp_next (output particles =
 *                the cloud of step t + 1), Z_next = encode(mean, covariance),
 *                F_z[b][t], F_u[b][t] through the differential of the Cholesky
 *                factor.
 * pddp_bnn_jvp_group(D, m) = lanes per trajectory of the moments launch: 16
 * for D <= 4 with at most 15 directions of (z | u) (cartpole, pendulum), 32
 * for D <= 6 with at most 31 (double cartpole), 0 = PDDP_E_UNSUPPORTED (D + m >
 * 7 or more directions: the autograd path then).  The network launch in
 * between is pddp_bnn_mlp_jvp_f32 with group = 8. */
int pddp_bnn_jvp_group(int D, int m);
typedef struct pddp_bnn_jvp {
  int32_t B, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8];
  int32_t in_dim, out_dim;
  const float* Z;        /* [B][N+1][n] nominal */
  const float* U;        /* [B][N][m] */
  const float* u_min;    /* [m], nullable with u_max */
  const float* u_max;
  const float* X_mean;   /* [in_dim] */
  const float* X_std_inv;
  const float* dX_mean;  /* [D] */
  const float* dX_std;
  const float* net_out;  /* [B P][8][out_dim] (moments) */
  const float* Xp;       /* [B][P][D] particles of step t */
  float* Xp_next;        /* [B][P][D] out (moments), nullable */
  float* eps;            /* [B][P][D] out (features), in (moments) */
  float* F;              /* [B P][8][in_dim] out (features) */
  float* Z_next;         /* [B][n] out (moments), nullable */
  float* F_z;            /* [B][N][n][n] out (moments) */
  float* F_u;            /* [B][N][n][m] out (moments) */
  /* use_predicted_std: standardised normals of step t, [P][D] (moments); the
   * network rows then carry 2 D outputs (mean | log std).  NULL: unused.
   * independent_noise != 0: the std is a constant of the differentiation
   * (modules.py:256-258). */
  const float* eps_out;
  int32_t independent_noise;
  /* [B] nullable: trajectories with slot[b] < 0 are skipped (nothing of theirs
   * is read or written); the others' network-facing rows - F and net_out - are
   * those of trajectory slot[b] (their rank among the ones that run), so that
   * the network launch in between covers the first (count) P 8 rows only
   * (pddp_bnn_mlp_jvp_rows_f32).  The derivative rollout of a round passes the
   * trajectories whose nominal is new. */
  const int32_t* slot;
} pddp_bnn_jvp;
int pddp_bnn_jvp_features_f32(const pddp_bnn_jvp* step, void* stream);
int pddp_bnn_jvp_moments_f32(const pddp_bnn_jvp* step, void* stream);
/* double precision: the same fields, double data */
typedef struct pddp_bnn_jvp_f64 {
  int32_t B, P, D, m, N, t;
  int32_t n_ang, ang[2], n_non, non[8];
  int32_t in_dim, out_dim;
  const double* Z;
  const double* U;
  const double* u_min;
  const double* u_max;
  const double* X_mean;
  const double* X_std_inv;
  const double* dX_mean;
  const double* dX_std;
  const double* net_out;
  const double* Xp;
  double* Xp_next;
  double* eps;
  double* F;
  double* Z_next;
  double* F_z;
  double* F_u;
  const double* eps_out;
  int32_t independent_noise;
  const int32_t* slot;
} pddp_bnn_jvp_f64;
int pddp_bnn_jvp_features_f64(const pddp_bnn_jvp_f64* step, void* stream);
int pddp_bnn_jvp_moments_f64(const pddp_bnn_jvp_f64* step, void* stream);

/* ---- value, gradient and Hessian of the QR cost on the angle-augmented
 * Gaussian state under DEFAULT encoding, every (trajectory, time step) in one
 * launch: costs/quadratic.py:60-99 on examples/cartpole/cost.py:60-87's
 * augmented state (utils/angular.py:161-248), differentiated as
 * utils/evaluation.py:238-288 batch_eval_cost does (L_z, L_u, L_zz, L_uz, L_uu
 * at the clamped action), by hyper-dual evaluation instead of autograd's double
 * backward.  Step N is the terminal state (Q_term, no action; its L_u / L_uz /
 * L_uu rows do not exist).  D in {2, 4, 6}, m <= 2, at most two angular
 * states; PDDP_E_UNSUPPORTED otherwise. */
typedef struct pddp_qr_cost {
  int32_t B, N, D, m;
  int32_t n_ang, ang[2], n_non, non[8];
  const float* Z;       /* [B][N+1][n], n = D + D (D + 1) / 2 */
  const float* U;       /* [B][N][m] */
  const float* u_min;   /* [m], nullable with u_max */
  const float* u_max;
  const float* Q;       /* [na][na], na = n_non + 2 n_ang */
  const float* Q_term;
  const float* R;       /* [m][m] */
  const float* x_goal;  /* [na] */
  const float* u_goal;  /* [m] */
  float* L;             /* [B][N+1] */
  float* L_z;           /* [B][N+1][n] */
  float* L_u;           /* [B][N][m] */
  float* L_zz;          /* [B][N+1][n][n] */
  float* L_uz;          /* [B][N][m][n] */
  float* L_uu;          /* [B][N][m][m] */
} pddp_qr_cost;
int pddp_qr_cost_derivs_f32(const pddp_qr_cost* cost, void* stream);
/* double precision: the same fields, double data */
typedef struct pddp_qr_cost_f64 {
  int32_t B, N, D, m;
  int32_t n_ang, ang[2], n_non, non[8];
  const double* Z;
  const double* U;
  const double* u_min;
  const double* u_max;
  const double* Q;
  const double* Q_term;
  const double* R;
  const double* x_goal;
  const double* u_goal;
  double* L;
  double* L_z;
  double* L_u;
  double* L_zz;
  double* L_uz;
  double* L_uu;
} pddp_qr_cost_f64;
int pddp_qr_cost_derivs_f64(const pddp_qr_cost_f64* cost, void* stream);

/* ---- pddp_amd/models/gp.py (the build's own plugin behind models/base.py:24-83
 * `DynamicsModel.forward(z, u, i, encoding)`; the reference has no GP - PARITY
 * UNPINNED): one moment-matched step of squared-exponential ARD GPs, one per
 * state increment, on the features [x_non-angular, sin a, cos a, u]:
 *   z [R][n] encoded state distributions, u [R][m] (already clamped) actions ->
 *   z_next [R][n]; with Fz [R][n][n] and Fu [R][n][m] non-NULL (both or
 *   neither) also d z_next / d z and d z_next / d u (the dynamics half of the
 *   derivative records, ilqr.py:443-470).
 * encoding: StateEncoding 1 (upper-triangular Cholesky, n = E + E(E+1)/2), 2
 * (variance), 3 (standard deviation), 4 (mean only); 0 (full covariance):
 * PDDP_E_UNSUPPORTED.  Built for (state_size, features + actions) = (2, 4),
 * (4, 6), (6, 9) - pendulum, cartpole, double cartpole; n + m <= 64; the
 * training set must fit the workgroup's 160 KB of LDS next to the inverses
 * (double cartpole: M <= 1278 in f32 / 596 in f64 without the Jacobian, 318 /
 * 74 with; the smaller systems several times that; PDDP_E_UNSUPPORTED beyond).
 * All arrays on the device. */
typedef struct pddp_gp_model {
  int state_size;        /* E: one GP per state increment */
  int action_size;       /* m */
  int M;                 /* training points */
  int encoding;
  int n_ang, n_non;
  int ang[4], non[8];    /* angular / non-angular state indices */
  const void* Xt;        /* [M][d] training inputs, d = n_non + 2 n_ang + m */
  /* the same inputs in pairs of points, for the M^2 loop's scalar loads:
   * [MQ/2][PS], MQ = M rounded up to a multiple of 4, PS = 2 d rounded up to a
   * multiple of 4, element 2 p + h of row j = Xt[2 j + h][p]; zero beyond
   * (points M .. MQ-1 and the row padding); 16-byte aligned */
  const void* Xt_pairs;
  const void* beta;      /* [E][M]  (K_a + sn2_a I)^-1 y_a */
  const void* beta_pairs; /* [E][MQ] = beta, rows zero-padded to MQ points */
  const void* Kinv;      /* [E][M][M] (K_a + sn2_a I)^-1, symmetric */
  const void* inv_ell2;  /* [E][d]  1 / lengthscale^2 */
  const void* sf2;       /* [E] signal variances */
  const void* sn2;       /* [E] noise variances */
} pddp_gp_model;
/* Bytes of LDS a workgroup of pddp_gp_step_* needs for `M` training points
 * (inputs = n + m, element_size 4 or 8); the launch needs <= 160 KB.  -1: the
 * (state_size, d) pair is not built.  Host function. */
long long pddp_gp_step_lds_bytes(int state_size, int d, int M, int inputs, int jacobian,
                                 int element_size);
int pddp_gp_step_f32(const pddp_gp_model* gp, int R, const float* z, const float* u,
                     float* z_next, float* Fz, float* Fu, void* stream);
int pddp_gp_step_f64(const pddp_gp_model* gp, int R, const double* z, const double* u,
                     double* z_next, double* Fz, double* Fu, void* stream);
/* The same with a mask over groups of `rows_per_mask` consecutive rows
 * (row_mask [ceil(R / rows_per_mask)], nullable = every row): rows of a group
 * with row_mask[group] == 0 are skipped - nothing of theirs is read or
 * written.  The derivative rollout of a round passes the trajectories whose
 * nominal is new (rows_per_mask = N): the records of the others stand
 * (ilqr.py:125-139: a rejected step retries with a larger mu on the same
 * nominal). */
int pddp_gp_step_masked_f32(const pddp_gp_model* gp, int R, const float* z, const float* u,
                            float* z_next, float* Fz, float* Fu, const uint8_t* row_mask,
                            int rows_per_mask, void* stream);
int pddp_gp_step_masked_f64(const pddp_gp_model* gp, int R, const double* z, const double* u,
                            double* z_next, double* Fz, double* Fu, const uint8_t* row_mask,
                            int rows_per_mask, void* stream);

/* ---- the GP workload's line search as a device rollout (the plugin path's
 * counterpart of pddp_line_search_*; ilqr.py:677-723 control law + rollout,
 * :764-791 trajectory cost): for every trajectory b and step size a, from
 * Z_new[0] = Z[b][0],
 *   u_t = clamp(U[b][t] + alphas[a] k[b][t] + K[b][t] (z_t - Z[b][t]))
 *   Jc[b][a] += QR cost of (z_t, u_t);  z_{t+1} = GP step of (z_t, u_t)
 * and the terminal cost of z_N: N + 1 launches of the moment-matched step's
 * kernel with nothing between them (stream-ordered, capturable).  The cost is
 * the QR cost on the angle-augmented Gaussian state (costs/quadratic.py:60-99:
 * E[(x~ - x_goal)^T Q (x~ - x_goal)] + (u - u_goal)^T R (u - u_goal), Q_term
 * at t = N) with the model's own angular / non-angular indices - its moments
 * are the step's feature moments; the covariance term enters as the encoding
 * carries it (full for 1, diagonal for 2 / 3, none for 4).  Zc [B][N+1][A][n],
 * Uc [B][N][A][m] time-major like pddp_line_search_*; Jc [B][A].  Rows of
 * inactive trajectories (active[b] == 0) and of failed sweeps
 * (bwd_status[b] != 0) are skipped (both nullable).  Same model limits as
 * pddp_gp_step_*; na = n_non + 2 n_ang <= 8, m <= 4. */
typedef struct pddp_gp_rollout {
  int32_t B, N, A;
  const void* Z;        /* [B][N+1][n] nominal states */
  const void* U;        /* [B][N][m] nominal actions (un-clamped) */
  const void* gains;    /* [B][N][m + m n]: k | K */
  const void* alphas;   /* [A] */
  const void* u_min;    /* [m], nullable with u_max */
  const void* u_max;
  const uint8_t* active;       /* [B] nullable */
  const int32_t* bwd_status;   /* [B] nullable */
  void* Zc;
  void* Uc;
  void* Jc;
  const void* Q;        /* [na][na] */
  const void* Q_term;
  const void* R;        /* [m][m] */
  const void* x_goal;   /* [na] */
  const void* u_goal;   /* [m] */
} pddp_gp_rollout;
int pddp_gp_rollout_f32(const pddp_gp_model* model, const pddp_gp_rollout* r,
                        void* stream);
int pddp_gp_rollout_f64(const pddp_gp_model* model, const pddp_gp_rollout* r,
                        void* stream);

/* Timing helper for bench.py: HIP events on `stream` (torch.cuda.Event only
 * sees torch's current stream). Host functions. */
int pddp_event_create(void** ev);
int pddp_event_record(void* ev, void* stream);
int pddp_event_elapsed_ms(void* start, void* stop, float* ms); /* syncs stop */
int pddp_event_destroy(void* ev);
/* Attaches (start, stop) to the NEXT kernel this host thread launches through
 * any pddp_* entry point: the events then time that kernel from its own start
 * to its own end (what rocprofv3 --kernel-trace reports), without the
 * stream's dispatch gaps.  One-shot; (NULL, NULL) detaches. */
int pddp_attach_events(void* start, void* stop);

#ifdef __cplusplus
}
#endif
#endif /* PDDP_HIP_H */
