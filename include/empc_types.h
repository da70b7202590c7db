/*
 * empc_types.h -- plain-old-data descriptors of one optimal-control problem.
 *
 * These structs are the flat, pointer-free (except for the two tables at the end of
 * EmpcProblemDesc) description of what the reference builds as a graph of shared_ptr objects:
 *   robot model        pinocchio::Model from urdf::buildModel(..., JointModelFreeFlyer)
 *                        (reference: src/trajectory.cpp:29-31)
 *   platform           MultiCopterBaseParams tau_f_/u_lb/u_ub (src/multicopter-base-params.cpp:67-101)
 *   stage cost tables  crocoddyl::CostModelSum per Stage (src/stage.cpp:52-70, src/factory/cost.cpp)
 *   contacts           crocoddyl::ContactModelMultiple per Stage (src/stage.cpp:38-50)
 *   knot table         ShootingProblem running models + terminal model (src/trajectory.cpp:110-140)
 *
 * They are consumed by (1) the HIP solver behind include/empc.h and (2) the CPU oracle under
 * oracle/ (test infrastructure).  All reals are FP64, all matrices row-major.
 *
 * Conventions (Pinocchio free-flyer):
 *   q = [p(3), quat xyzw(4), theta(nj)],  v = [v_lin(3) body frame, omega(3) body frame, theta_dot(nj)]
 *   x = [q; v] (nx = nq+nv), tangent dx = [dq(nv); dv(nv)] (ndx = 2 nv), spatial order [linear; angular].
 */
#ifndef EMPC_TYPES_H
#define EMPC_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMPC_MAX_BODIES 8   /* floating base + up to 7 revolute joints            */
#define EMPC_MAX_NV 14      /* 6 + (EMPC_MAX_BODIES-1) rounded up                 */
#define EMPC_MAX_NQ 15
#define EMPC_MAX_NX 29
#define EMPC_MAX_NDX 28
#define EMPC_MAX_ROTORS 8
#define EMPC_MAX_NU 16      /* rotors + arm joints                                */
#define EMPC_MAX_NR 28      /* longest residual (state residual = ndx)            */
#define EMPC_MAX_FRAMES 8
#define EMPC_MAX_COSTS 20   /* per cost set, including the solver's "barrier"     */
#define EMPC_MAX_CONTACTS 2
#define EMPC_NAME_LEN 40

/* Kinematic tree flattened from the URDF. Fixed joints are merged into their parent body
 * (as pinocchio::urdf::buildModel does); body 0 is the free-flyer root. */
typedef struct EmpcModelDesc {
  int32_t nbodies; /* moving bodies incl. the floating base */
  int32_t nq, nv;
  int32_t nframes;
  int32_t parent[EMPC_MAX_BODIES];       /* parent[0] = -1                                    */
  double jplace_R[EMPC_MAX_BODIES][9];   /* joint frame in the parent body frame (rotation)   */
  double jplace_p[EMPC_MAX_BODIES][3];   /* ... (translation)                                 */
  double axis[EMPC_MAX_BODIES][3];       /* revolute axis in the joint frame (unit)           */
  double mass[EMPC_MAX_BODIES];
  double com[EMPC_MAX_BODIES][3];        /* centre of mass in the body frame                  */
  double inertia[EMPC_MAX_BODIES][9];    /* rotational inertia about the COM, body frame      */
  double effort_limit[EMPC_MAX_BODIES];  /* URDF <limit effort>, entry 0 unused               */
  int32_t frame_body[EMPC_MAX_FRAMES];   /* body each operational frame is rigidly attached to */
  double frame_R[EMPC_MAX_FRAMES][9];    /* frame placement in that body frame                */
  double frame_p[EMPC_MAX_FRAMES][3];
  char frame_name[EMPC_MAX_FRAMES][EMPC_NAME_LEN];
  double gravity[3];                     /* world frame, (0,0,-9.81)                          */
} EmpcModelDesc;

/* reference: include/eagle_mpc/factory/cost.hpp:38-63 (CostModelTypes) */
enum EmpcCostType {
  EMPC_COST_STATE = 0,
  EMPC_COST_CONTROL = 1,
  EMPC_COST_FRAME_PLACEMENT = 2,
  EMPC_COST_FRAME_ROTATION = 3,
  EMPC_COST_FRAME_VELOCITY = 4,
  EMPC_COST_FRAME_TRANSLATION = 5,
  EMPC_COST_CONTACT_FRICTION_CONE = 6
};

/* reference: include/eagle_mpc/factory/activation.hpp:25-53 (the four that src/factory/activation.cpp implements) */
enum EmpcActivationType {
  EMPC_ACT_QUAD = 0,
  EMPC_ACT_WEIGHTED_QUAD = 1,
  EMPC_ACT_QUADRATIC_BARRIER = 2,
  EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER = 3
};

enum EmpcContactType { EMPC_CONTACT_3D = 0, EMPC_CONTACT_6D = 1 };

/* One entry of a CostModelSum: CostModelResidual(state, activation, residual) + weight + active. */
typedef struct EmpcCost {
  char name[EMPC_NAME_LEN]; /* CostModelSum iterates a std::map => alphabetical by name      */
  int32_t type;             /* EmpcCostType                                                  */
  int32_t activation;       /* EmpcActivationType                                            */
  int32_t active;
  int32_t frame;            /* index into EmpcModelDesc frames (frame costs / friction cone) */
  int32_t nr;               /* residual length                                               */
  int32_t is_barrier;       /* 1 for the cost SolverSbFDDP::barrierInit injects              */
  int32_t ref_share;        /* filled by the solver: index of an earlier State cost of the same set with an
                               identical reference (its residual is reused), or -1                          */
  int32_t reserved;
  double weight;
  /* reference payload, by type:
   *   STATE: xref[nx] | CONTROL: uref[nu] | FRAME_PLACEMENT: p[3], R[9] | FRAME_ROTATION: R[9]
   *   FRAME_VELOCITY: lin[3], ang[3] | FRAME_TRANSLATION: p[3] | FRICTION_CONE: n[3], mu      */
  double ref[EMPC_MAX_NX];
  double act_w[EMPC_MAX_NR]; /* activation weights (ones when unused)                        */
  double lb[EMPC_MAX_NR];    /* barrier bounds (+-inf allowed)                               */
  double ub[EMPC_MAX_NR];
} EmpcCost;

typedef struct EmpcContact {
  char name[EMPC_NAME_LEN];
  int32_t type;  /* EmpcContactType */
  int32_t frame;
  double ref_p[3];
  double ref_R[9];
  double gains[2];
} EmpcContact;

/* The cost/contact tables of one action model (= one Stage, or one private MPC knot). */
typedef struct EmpcCostSet {
  int32_t ncosts;
  int32_t ncontacts;
  EmpcCost costs[EMPC_MAX_COSTS]; /* sorted by name */
  EmpcContact contacts[EMPC_MAX_CONTACTS]; /* sorted by name: the row order of crocoddyl's ContactModelMultiple (src/stage.cpp:38-48);
                                              two entries: two ContactModel3D (six stacked rows), opt-in kernels */
} EmpcCostSet;

enum EmpcIntegrator { EMPC_INTEGRATOR_EULER = 0, EMPC_INTEGRATOR_RK4 = 1 };

typedef struct EmpcProblemDesc {
  EmpcModelDesc model;
  int32_t nx, ndx, nu;   /* nu = n_rotors + (nv-6)                                           */
  int32_t n_rotors;
  int32_t T;             /* running knots; node T is the terminal node                       */
  int32_t n_sets;
  int32_t has_contact;   /* Contact forward dynamics on every node (src/trajectory.cpp:84-86) */
  int32_t use_squash;    /* ActuationSquashingModel (src/factory/diff-action.cpp:24-28)      */
  int32_t integrator;    /* EmpcIntegrator                                                   */
  int32_t reserved;
  double dt;             /* seconds                                                          */
  double tau_f[6 * EMPC_MAX_ROTORS]; /* 6 x n_rotors row-major (multicopter-base-params.cpp:71-78) */
  double u_lb[EMPC_MAX_NU];
  double u_ub[EMPC_MAX_NU];
  double x0[EMPC_MAX_NX];
  const EmpcCostSet* sets;  /* n_sets entries                                                */
  const int32_t* knot_set;  /* T+1 entries: cost-set index of every node (terminal last)     */
} EmpcProblemDesc;

/* Solver constants. Defaults (empc_solver_params_default) are the values the reference hard-codes in
 * src/sbfddp.cpp:5-31 plus the Crocoddyl defaults listed in SURVEY.md A.1; the trailing option
 * block selects among the behaviours that the un-vendored Crocoddyl fork leaves unpinned (A.8). */
enum EmpcStopCriteria { EMPC_STOP_COST_REDUCTION = 0, EMPC_STOP_EXPECTED_REDUCTION = 1, EMPC_STOP_QU_NORM = 2 };
enum EmpcGapNorm { EMPC_GAP_L1 = 0, EMPC_GAP_LINF = 1 };
/* Which solver the handle is: the fork's SolverSbFDDP (src/sbfddp.cpp) or crocoddyl's SolverBoxFDDP / SolverBoxDDP, the
 * other two back ends MpcAbstract accepts (include/eagle_mpc/mpc-base.hpp:36-47, src/mpc-controllers/carrot-mpc.cpp:232-242)
 * and the `useSquash = False` branch of the reference's examples (examples/python/trajectory.py:20-23). */
enum EmpcSolverType { EMPC_SOLVER_SBFDDP = 0, EMPC_SOLVER_BOXFDDP = 1, EMPC_SOLVER_BOXDDP = 2 };

typedef struct EmpcSolverParams {
  double smooth_init, smooth_mult;             /* 0.1, 0.5  (sbfddp.cpp:9-10)   */
  double barrier_weight;                       /* 1e-3      (sbfddp.cpp:11)     */
  double convergence_init, convergence_stop, convergence_mult; /* 1e-2,1e-3,1e-1 (:12-14) */
  double reg_init;                             /* 1e-9      (sbfddp.cpp:16)     */
  double th_acceptnegstep;                     /* 2         (sbfddp.cpp:17)     */
  double th_stop_gaps;                         /* 1         (sbfddp.cpp:27)     */
  double th_grad, th_acceptstep, th_stepdec, th_stepinc; /* 1e-12, 0.1, 0.5, 0.01 */
  double reg_incfactor, reg_decfactor, reg_min, reg_max; /* 10, 10, 1e-9, 1e9     */
  double th_gaptol;                            /* 1e-16                         */
  int32_t n_alphas;                            /* 10: alpha_n = 2^-n            */
  int32_t stop_criteria;                       /* U1: EmpcStopCriteria          */
  int32_t gap_norm;                            /* U1: EmpcGapNorm               */
  int32_t terminal_dt_scaling;                 /* U2: 1 = terminal node is IAM.calc(x,u=0), scaled by dt */
  int32_t smoothsat_power;                     /* U3: 2 (d^2) or 4 (d^4)        */
  int32_t solver_type;                         /* EmpcSolverType                */
  /* crocoddyl SolverBoxFDDP / SolverBoxDDP (~1.8; SURVEY A.8 register, unpinned like the rest of A.1): convergence
   * threshold of the box solvers and the BoxQP(nu, maxiter, th_acceptstep, th_grad, reg) they construct */
  double box_th_stop;                          /* 5e-5                          */
  double boxqp_th_acceptstep, boxqp_th_grad, boxqp_reg; /* 0.1, 1e-5, 0        */
  int32_t boxqp_maxiter;                       /* 100                           */
  int32_t reserved;
} EmpcSolverParams;

/* per-trajectory status bits returned by the solver */
#define EMPC_STATUS_CONVERGED 1   /* last inner loop returned true                    */
#define EMPC_STATUS_REG_MAX 2     /* an inner loop gave up at reg_max                 */
#define EMPC_STATUS_MAXITER 4     /* an inner loop ran out of iterations              */
#define EMPC_STATUS_DDP_CLEANUP 8 /* solveDDP ran because the FDDP result was infeasible */

/* Solver scalars of ONE trajectory: the members of crocoddyl::SolverAbstract / SolverDDP / SolverFDDP and of the fork's
 * SolverSbFDDP that one pass through the loop body of solveFDDP / solveDDP (src/sbfddp.cpp:241-311, 329-389) reads and
 * writes, plus the continuation bookkeeping of solve() (:205-220).  The solver keeps one per trajectory on the device; the
 * step-wise entry points (empc_solver_get_states / set_states / empc_sweep_batch / empc_select_batch) expose it so that a
 * test can put the solver at any iterate of any pass and run exactly one iteration from there (teacher-forced parity). */
typedef struct EmpcTrajState {
  int32_t phase;         /* 0,1,.. = FDDP pass index; 100 = DDP clean-up (solveDDP); 255 = done                     */
  int32_t iter;          /* iter_ of the running pass; after the solve: total iterations - 1 (sbfddp.cpp:222)       */
  int32_t total_iters;   /* iterations of the passes already finished                                               */
  int32_t status;        /* EMPC_STATUS_* bits                                                                       */
  int32_t is_feasible, was_feasible;
  int32_t need_calc;     /* the pass starts: problem.calc at the candidate before calcDiff (iter_ == 0)              */
  int32_t need_lin;      /* recalc flag of the loop: calcDiff runs in the next computeDirection                      */
  int32_t maxiter;
  int32_t bwd_failed;    /* computeDirection gave up at reg_max in this iteration                                    */
  int32_t trace_count;   /* iteration records written in this solve                                                  */
  int32_t last_ok;       /* the last solveFDDP / solveDDP returned true                                              */
  int32_t accepted_alpha;/* index n of the step length 2^-n accepted by the last line search, -1 = none              */
  int32_t last_alpha;    /* index of the last step length the line search tried                                      */
  int32_t job;           /* streamed solves (empc_solver_solve_stream): the queue entry this slot works on, -1 = none */
  int32_t reserved;
  double smooth, smooth_next, convergence, th_stop;
  double xreg, ureg, cost, cost_prev, stop, steplength, dV, dVexp, d0, d1;
  double dg_u, dq_u;     /* sum Qu.k , -sum k.Quu k      (control part of dg / dq)                                    */
  double dg_f, dq_f;     /* -sum Vx.f , +sum f.Vxx f     (gap part, used while the trajectory is infeasible)         */
  double gapnorm, qu2;   /* norm of the gaps (EmpcGapNorm), sum |Qu|^2                                               */
} EmpcTrajState;

/* stages of one sweep, for empc_sweep_batch */
#define EMPC_STAGE_LINEARIZE 1 /* calc (pass starts) + calcDiff           */
#define EMPC_STAGE_BACKWARD 2  /* backwardPass + computeGains, with the regularisation retry of computeDirection */
#define EMPC_STAGE_ROLLOUT 4   /* forwardPass / forwardPassDDP for every step length */
#define EMPC_STAGE_SELECT 8    /* acceptance, regularisation update, stopping test, continuation */
#define EMPC_STAGE_ALL 15

/* One record of the per-iteration trace (empc_solver_enable_trace / empc_solver_get_trace): what a
 * crocoddyl::CallbackAbstract sees after stoppingCriteria() in solveFDDP / solveDDP
 * (reference src/sbfddp.cpp:303-307, 381-385).  EMPC_TRACE_WORDS doubles per record:
 *   0 phase (0,1,.. = FDDP pass; 100 = DDP clean-up)  1 iter  2 cost  3 stop  4 xreg  5 steplength
 *   6 is_feasible  7 dV  8 dVexp  9 gap norm  10 d0  11 d1                                          */
#define EMPC_TRACE_WORDS 12

#ifdef __cplusplus
}
#endif
#endif /* EMPC_TYPES_H */
