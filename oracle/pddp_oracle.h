/*
 * pddp_oracle.h - CPU restatement (plain C) of the reference's iLQR hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is the checker for the HIP kernels: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (pddp_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity pin: every function below is checked against golden vectors captured
 * from the reference itself (tools/make_golden.py -> tests/golden/ npz files; see
 * tests/test_oracle_golden.py).  The reference's arithmetic lives in the
 * un-vendored dependency torch==0.4.1 (Pipfile.lock:117-131); the goldens were
 * produced with torch 2.10 CPU through the shims of tools/ref_shims.py.
 *
 * Each entry point exists as <name>_f64 and <name>_f32 (all arithmetic in that
 * type, like a reference run with that tensor dtype).  Single trajectory, like
 * the reference: the caller loops over a batch.
 *
 * Supported here: StateEncoding.IGNORE_UNCERTAINTY for the model / cost
 * derivatives (BASELINE.json configs[0..1]); `backward`, `boxqp` and the
 * controller state machine are encoding-agnostic.
 */
#ifndef PDDP_ORACLE_H
#define PDDP_ORACLE_H

#include "../include/pddp_problem.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PDDP_ORACLE_DECL(T, S)                                                 \
  /* model(z,u) and its Jacobians: pddp/examples/<m>/model.py forward() +     \
   * utils/evaluation.py:242-288.  F_z [n][n], F_u [n][m] may be NULL. */     \
  int pddp_oracle_dynamics_##S(const pddp_problem* p, const T* z, const T* u, \
                               T* z_next, T* F_z, T* F_u);                     \
  /* cost(z,u) with gradient / Hessian: costs/quadratic.py:60-99 +            \
   * utils/evaluation.py:134-239.  u == NULL <=> terminal. */                 \
  int pddp_oracle_cost_##S(const pddp_problem* p, const T* z, const T* u,     \
                           int terminal, T* l, T* l_z, T* l_u, T* l_zz,       \
                           T* l_uz, T* l_uu);                                  \
  /* controllers/ilqr.py:393-486 forward(). u_min/u_max nullable. */          \
  int pddp_oracle_forward_##S(const pddp_problem* p, const T* z0, const T* U, \
                              int N, const T* u_min, const T* u_max, T* Z,    \
                              T* F_z, T* F_u, T* L, T* L_z, T* L_u, T* L_zz,  \
                              T* L_uz, T* L_uu);                               \
  /* utils/constraint.py:150-266 boxqp(). Returns `result`; Ufree is          \
   * [nfree][nfree] upper Cholesky packed at stride nfree. */                 \
  int pddp_oracle_boxqp_##S(int m, const T* x0, const T* Q, const T* c,       \
                            const T* lower, const T* upper, T* x, T* Ufree,   \
                            unsigned char* free_mask);                         \
  /* controllers/ilqr.py:529-674 backward(). Returns PDDP_BWD_* (the          \
   * reference raises RuntimeError for every non-zero status). */             \
  int pddp_oracle_backward_##S(int n, int m, int N, const T* F_z,             \
                               const T* F_u, const T* L_z, const T* L_u,      \
                               const T* L_zz, const T* L_uz, const T* L_uu,   \
                               double reg, int V_zz_reg, const T* u_min,      \
                               const T* u_max, const T* U, T* k, T* K);        \
  /* controllers/ilqr.py:677-723 _control_law() for A alphas:                 \
   * Z_new [N+1][A][n], U_new [N][A][m]. */                                   \
  int pddp_oracle_control_law_##S(const pddp_problem* p, int N, int A,        \
                                  const T* Z, const T* U, const T* k,         \
                                  const T* K, const T* alphas, const T* u_min,\
                                  const T* u_max, T* Z_new, T* U_new);         \
  /* controllers/ilqr.py:764-791 _trajectory_cost(): J [A]. */                \
  int pddp_oracle_trajectory_cost_##S(const pddp_problem* p, int N, int A,    \
                                      const T* Z_new, const T* U_new, T* J);   \
  /* controllers/ilqr.py:237-316 fit() incl. step()/_step() and the mu/delta  \
   * schedule (:364-390). U [N][m] in/out, Z [N+1][n] out, K [N][m][n] out.   \
   * trace rows = {iteration, state, J_opt, mu, delta} per attempt (what       \
   * on_iteration observes, :232-233). Returns the final iLQRState. */        \
  int pddp_oracle_fit_##S(const pddp_problem* p, const T* z0, T* U, int N,    \
                          int n_iterations, double tol, double max_reg,       \
                          const T* u_min, const T* u_max, const T* alphas,    \
                          int A, T* Z, T* K, double* trace, int max_trace,    \
                          int* n_trace);

PDDP_ORACLE_DECL(double, f64)
PDDP_ORACLE_DECL(float, f32)

#ifdef __cplusplus
}
#endif
#endif /* PDDP_ORACLE_H */
