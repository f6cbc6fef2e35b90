/*
 * pddp_problem.h - plain-C description of one of the reference's sample
 * problems (dynamics model + quadratic cost on the augmented state).
 *
 * Shared by the C-ABI of the HIP library (include/pddp_hip.h) and by the CPU
 * oracle (oracle/pddp_oracle.h).  It carries DATA only; each side has its own
 * implementation of the maths.
 *
 * Reference types this struct flattens:
 *   pddp/examples/problems.py:21-30        SampleProblems enum -> `model`
 *   pddp/utils/encoding.py:25-43           StateEncoding enum  -> `encoding`
 *   pddp/examples/cartpole/model.py:37-54  model constants (float32 tensors!)
 *   pddp/costs/quadratic.py:36-58          Q, R, Q_term, x_goal, u_goal
 *
 * All constants are stored as doubles holding the float32-rounded values the
 * reference ends up with: e.g. dt = 0.1f =
 * 0.100000001490116...
 */
#ifndef PDDP_PROBLEM_H
#define PDDP_PROBLEM_H

#ifdef __cplusplus
extern "C" {
#endif

/* pddp/examples/problems.py:25-28 */
enum {
  PDDP_MODEL_CARTPOLE = 1,
  PDDP_MODEL_DOUBLE_CARTPOLE = 2,
  PDDP_MODEL_PENDULUM = 3,
  PDDP_MODEL_RENDEZVOUS = 4
};

/* pddp/utils/encoding.py:29-43 */
enum {
  PDDP_ENC_FULL_COVARIANCE_MATRIX = 0,
  PDDP_ENC_UPPER_TRIANGULAR_CHOLESKY = 1,
  PDDP_ENC_VARIANCE_ONLY = 2,
  PDDP_ENC_STANDARD_DEVIATION_ONLY = 3,
  PDDP_ENC_IGNORE_UNCERTAINTY = 4
};

/* pddp/controllers/ilqr.py:35-55 */
enum {
  PDDP_STATE_UNDEFINED = 0,
  PDDP_STATE_ACCEPTED = 1,
  PDDP_STATE_REJECTED = 2,
  PDDP_STATE_NOT_PD = 3,
  PDDP_STATE_MAX_REG = 4,
  PDDP_STATE_CONVERGED = 5
};

/* Gain branches of pddp/controllers/ilqr.py:584-672. */
enum {
  PDDP_BRANCH_EIG = 0,     /* V_zz_reg=False: A (no bounds) / B (BoxQP) */
  PDDP_BRANCH_CHOLESKY = 1 /* V_zz_reg=True:  C (no bounds) / D (BoxQP) */
};

/* Status codes of the backward sweep (never thrown across the C ABI). */
enum {
  PDDP_BWD_OK = 0,
  PDDP_BWD_NAN = 1,         /* ilqr.py:639-640 non-finite gains */
  PDDP_BWD_NOT_PD = 2,      /* ilqr.py:595 potrf failure */
  PDDP_BWD_BOXQP_FAILED = 3 /* ilqr.py:608-610,653-655 result < 1 */
};

#define PDDP_MAX_STATE 8  /* un-encoded state size D (rendezvous) */
#define PDDP_MAX_AUG 8    /* augmented state size (double cartpole, rendezvous) */
#define PDDP_MAX_ACTION 4 /* rendezvous */
#define PDDP_MAX_PARAMS 8

typedef struct pddp_problem {
  int model;     /* PDDP_MODEL_* */
  int encoding;  /* PDDP_ENC_* */
  int state_size;  /* D */
  int action_size; /* m */
  int encoded_size; /* n (encoding.py:46-67) */
  int aug_size;     /* angular.py:329-340 */
  /* params[0] = dt, then in constructor order:
   *   cartpole        mc, mp, l, mu, g            (cartpole/model.py:37)
   *   double cartpole mc, mp1, mp2, l1, l2, mu, g (double_cartpole/model.py:36-44)
   *   pendulum        m, l, mu, g                 (pendulum/model.py:38)
   *   rendezvous      m, alpha                    (rendezvous/model.py:37) */
  double params[PDDP_MAX_PARAMS];
  /* QRCost on the augmented state, row-major (quadratic.py:36-58). */
  double Q[PDDP_MAX_AUG * PDDP_MAX_AUG];
  double Q_term[PDDP_MAX_AUG * PDDP_MAX_AUG];
  double R[PDDP_MAX_ACTION * PDDP_MAX_ACTION];
  double x_goal[PDDP_MAX_AUG];
  double u_goal[PDDP_MAX_ACTION];
} pddp_problem;

#ifdef __cplusplus
}
#endif
#endif /* PDDP_PROBLEM_H */
