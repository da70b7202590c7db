/*
 * empc.h -- C ABI of the MI355X-native batched Squash-box FDDP solver (libempc.so).
 *
 * This is the drop-in boundary for the eagle-mpc hot path: plain pointers and sizes, no
 * C++ / torch / Eigen types.  All host buffers are caller-owned, row-major FP64; device memory is owned by the opaque
 * handles.  Every function returning int returns 0 on success and a negative EMPC_ERR_* code otherwise;
 * empc_last_error() gives the message (thread-local).  No C++ exception crosses this boundary.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   empc_trajectory_*         Trajectory::create/autoSetup/createProblem/set_initial_state + getters
 *                             include/eagle_mpc/trajectory.hpp:49-74, src/trajectory.cpp:21-143
 *   empc_solver_create        SolverSbFDDP::SolverSbFDDP(problem, squashing_model)   include/eagle_mpc/sbfddp.hpp:39-40,
 *                             src/sbfddp.cpp:5-38 (incl. barrierInit :169-190)
 *   empc_solver_solve         SolverSbFDDP::solve(init_xs, init_us, maxiter, is_feasible, regInit)
 *                             include/eagle_mpc/sbfddp.hpp:42-46, src/sbfddp.cpp:192-226
 *   empc_solver_set_x0        problem->set_x0(x0)   (examples/python/mpc.py:50)
 *   empc_solver_set_warmstart the init_xs/init_us arguments of solve() (crocoddyl setCandidate, src/sbfddp.cpp:199)
 *   empc_solver_get_xs/us     SolverAbstract::get_xs/get_us (bindings/python/eagle_mpc/sbfddp.hpp:46-79)
 *   empc_solver_get_us_squash SolverSbFDDP::getSquashControls   include/eagle_mpc/sbfddp.hpp:48, src/sbfddp.cpp:479-487
 *   empc_solver_get_cost/iters/stop   get_cost/get_iter/get_stop (crocoddyl::SolverAbstract)
 *   empc_solver_set_convergence_init  SolverSbFDDP::set_convergence_init   src/sbfddp.cpp:491
 *   empc_solver_update_problem  what MpcAbstract::updateProblem does by mutating shared cost models in place
 *                             (src/mpc-controllers/carrot-mpc.cpp:298-359)
 *   empc_carrot_mpc_*         CarrotMpc ctor/createProblem/updateProblem/computeStateReference
 *                             (src/mpc-controllers/carrot-mpc.cpp:15-50,178-248,298-403, src/mpc-base.cpp:5-60)
 *   empc_rail_mpc_* / empc_weighted_mpc_* / empc_mpc_*
 *                             RailMpc, WeightedMpc (src/mpc-controllers/rail-mpc.cpp, weighted-mpc.cpp)
 *   empc_plant_*              AerialSimulator.simulateStep (bindings/python/eagle_mpc/utils/simulator.py:24-29)
 *   empc_linearize_batch / empc_backward_batch / empc_rollout_batch
 *                             crocoddyl SolverDDP::calcDiff / backwardPass / SolverFDDP::forwardPass as called at
 *                             src/sbfddp.cpp:244,264 -- exposed per phase for parity tests and roofline measurements
 */
#ifndef EMPC_H
#define EMPC_H

#include "empc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EMPC_OK 0
#define EMPC_ERR_INVALID -1     /* bad argument / malformed problem          */
#define EMPC_ERR_RUNTIME -2     /* HIP runtime error, no device, out of memory */
#define EMPC_ERR_UNSUPPORTED -3 /* problem shape the kernels are not built for */
#define EMPC_ERR_IO -4          /* YAML / URDF could not be read              */

typedef struct EmpcSolver EmpcSolver;
typedef struct EmpcTrajectory EmpcTrajectory;
typedef struct EmpcProblem EmpcProblem;

const char* empc_last_error(void);
const char* empc_version(void);
/* number of HIP devices visible (0 when there is none); never throws */
int empc_device_count(void);

/* ---- data directories (reference: EAGLE_MPC_YAML_DIR / EAGLE_MPC_ROBOT_DATA_DIR, config/path.hpp.in:4-11) ---- */
int empc_set_data_dirs(const char* yaml_dir, const char* robot_data_dir);

/* ---- YAML problem factory ------------------------------------------------------------------------------- */
EmpcTrajectory* empc_trajectory_create(const char* yaml_path);           /* create() + autoSetup(); NULL on error */
void empc_trajectory_destroy(EmpcTrajectory* t);
int empc_trajectory_dims(const EmpcTrajectory* t, int* nx, int* ndx, int* nu, int* n_stages, int* has_contact,
                         int* duration_ms);
int empc_trajectory_stage_info(const EmpcTrajectory* t, int stage, char* name, int name_len, int* duration_ms,
                               int* is_transition, int* n_costs, int* n_contacts);
/* Stage::get_t_ini() (include/eagle_mpc/stage.hpp); -1 on error */
long long empc_trajectory_stage_t_ini(const EmpcTrajectory* t, int stage);
/* Stage::get_costs()->get_costs(): the cost-th entry in name order, with its weight and active flag */
int empc_trajectory_stage_cost(const EmpcTrajectory* t, int stage, int cost, char* name, int name_len, double* weight,
                               int* active);
/* Trajectory::removeStage(idx_stage) (src/trajectory.cpp:145-150; bindings/python/eagle_mpc/trajectory.hpp:62) */
int empc_trajectory_remove_stage(EmpcTrajectory* t, int stage);
/* Trajectory::get_robot_model_path() (src/trajectory.cpp:160; bindings .../trajectory.hpp:42-43); returns the length */
int empc_trajectory_robot_model_path(const EmpcTrajectory* t, char* path, int path_len);
/* Stage::get_cost_types() / get_contacts() / get_contact_types() (include/eagle_mpc/stage.hpp; bindings/python/eagle_mpc/stage.hpp:50-73):
   the factory type name of a cost ("CostModelState", ...), name and type ("ContactModel3D" | "ContactModel6D") of contact k */
int empc_trajectory_stage_cost_type(const EmpcTrajectory* t, int stage, const char* cost_name, char* type, int type_len);
int empc_trajectory_stage_contact(const EmpcTrajectory* t, int stage, int contact, char* name, int name_len, char* type, int type_len);
/* MultiCopterBaseParams (include/eagle_mpc/multicopter-base-params.hpp:40-60; bindings .../multicopter-base-params.hpp:47-88):
   scalars = { cf, cm, max_thrust, min_thrust, max_prop_speed, min_prop_speed }, base_link_name; pose of rotor i (R row-major
   3 x 3, p) and its spin direction */
int empc_trajectory_get_platform_params(const EmpcTrajectory* t, double* scalars /* 6 */, char* base_link_name, int name_len);
int empc_trajectory_get_rotor_pose(const EmpcTrajectory* t, int rotor, double* R /* 9 */, double* p /* 3 */, int* spin_direction);
int empc_trajectory_get_initial_state(const EmpcTrajectory* t, double* x0 /* nx */);
int empc_trajectory_set_initial_state(EmpcTrajectory* t, const double* x0 /* nx */);
int empc_trajectory_get_platform(const EmpcTrajectory* t, double* tau_f /* 6 x n_rotors */, double* u_lb, double* u_ub,
                                 int* n_rotors);
/* createProblem(dt_ms, squash, integration_method); dt_ms == 0 uses the YAML's problem_params */
EmpcProblem* empc_trajectory_create_problem(const EmpcTrajectory* t, int dt_ms, int squash, const char* integration_method);
void empc_problem_destroy(EmpcProblem* p);
/* flat descriptor of the problem; valid until the problem is modified or destroyed */
const EmpcProblemDesc* empc_problem_desc(EmpcProblem* p);
int empc_problem_set_x0(EmpcProblem* p, const double* x0);
/* flat-key parameter lookup (ParamsServer::getParam<std::string>); returns length or EMPC_ERR_INVALID */
int empc_trajectory_get_param(const EmpcTrajectory* t, const char* key, char* value, int value_len);

/* ---- solver ------------------------------------------------------------------------------------------------ */
void empc_solver_params_default(EmpcSolverParams* p);
/* batch trajectories of the same problem on HIP device `device`. params == NULL -> defaults. */
EmpcSolver* empc_solver_create(const EmpcProblemDesc* problem, const EmpcSolverParams* params, int batch, int device);
void empc_solver_destroy(EmpcSolver* s);
int empc_solver_update_problem(EmpcSolver* s, const EmpcProblemDesc* problem); /* same shapes; new cost tables / x0 */
/* which kernel instantiation serves this solver: "runtime model" (any serial-chain robot of the class; the robot is read from
 * the problem image) or "baked <robot>[, <contact>]" (the robot equals, bit for bit, one of the tables compiled into the
 * library, eagle-mpc_amd/csrc/baked/: its constants are literals of the kernels).  EMPC_BAKED=0 in the environment keeps
 * every solver on the runtime-model kernels.  The string lives as long as the library. */
const char* empc_solver_kernel_family(const EmpcSolver* s);
int empc_solver_set_x0(EmpcSolver* s, const double* x0s /* batch x nx, NULL = problem x0 for every trajectory */);
int empc_solver_set_warmstart(EmpcSolver* s, const double* xs /* batch x (T+1) x nx or NULL = zero state */,
                              const double* us /* batch x T x nu or NULL = zeros */);
int empc_solver_set_convergence_init(EmpcSolver* s, double convergence_init);
/* returns EMPC_OK when the call ran (like the reference's solve(), which always returns true);
 * per-trajectory outcomes are in empc_solver_get_status */
int empc_solver_solve(EmpcSolver* s, int maxiter, int is_feasible);
int empc_solver_get_xs(EmpcSolver* s, double* xs /* batch x (T+1) x nx */);
int empc_solver_get_us(EmpcSolver* s, double* us /* batch x T x nu */);
int empc_solver_get_us_squash(EmpcSolver* s, double* us_squash /* batch x T x nu */);
int empc_solver_get_cost(EmpcSolver* s, double* cost /* batch */);
int empc_solver_get_stop(EmpcSolver* s, double* stop /* batch */);
int empc_solver_get_iters(EmpcSolver* s, int* iters /* batch: iter_ = total iterations - 1 */);
int empc_solver_get_status(EmpcSolver* s, int* status /* batch: EMPC_STATUS_* bits */);
/* results packed on the device, one row per rollout: xs | us_squash | cost | iters (as double) -- the payload of the
 * multi-GPU gather; dst_device == NULL only returns the row length in doubles */
int empc_solver_pack_results_device(EmpcSolver* s, double* dst_device /* batch x row, device memory */, int* row_doubles);

/* Per-iteration trace: the record a crocoddyl callback would see each iteration (reference: setCallbacks +
 * src/sbfddp.cpp:303-307, 381-385; src/mpc-base.cpp:52-57 installs CallbackVerbose when the YAML asks for it).
 * enable_trace allocates a device ring of `capacity` records per trajectory (0 switches it off); it is filled by
 * every later solve.  get_trace copies trajectory b's records of the last solve, oldest first:
 * records x EMPC_TRACE_WORDS doubles (layout: include/empc_types.h), at most max_records; *n_records = records
 * written by the solve (may exceed the ring's capacity, then only the newest `capacity` are available). */
int empc_solver_enable_trace(EmpcSolver* s, int capacity);
int empc_solver_get_trace(EmpcSolver* s, int b, double* records, int max_records, int* n_records);

/* timing / accounting of the last solve (device time measured with HIP events on the solver's stream) */
typedef struct EmpcSolveStats {
  int sweeps;                 /* host-side sweeps (one sweep = one pass of the kernel sequence)             */
  int max_iters;              /* largest per-trajectory DDP iteration count                                 */
  long long total_iters;      /* sum over trajectories of DDP iterations (FDDP passes + DDP clean-up)       */
  long long linearize_units;  /* (trajectory, node) units linearized                                        */
  long long rollout_units;    /* (trajectory, alpha, node) units rolled out                                 */
  long long backward_units;   /* (trajectory, node) units of the backward pass                              */
  double ms_total;            /* wall time of the solve on the stream                                       */
  /* kernel times from HIP events on the solver's stream, summed over the TIMED launches: every sweep of the phase-level
   * calls, every EMPC_TIMING_EVERY-th sweep (default 4) of a solve -- an event record between two kernels costs ~10 us of
   * stream time.  n_* and *_units count the same timed launches, so ms / n and units / n are per-launch averages. */
  double ms_linearize, ms_backward, ms_rollout, ms_select, ms_calc;
  int n_linearize, n_backward, n_rollout, n_select, n_calc;
  int timing_every;           /* the sampling stride of the event-timed sweeps of this solve (1 for the phase-level calls) */
  /* units of EVERY sweep of the solve, timed or not: the total work (the *_units above cover the timed launches only) */
  long long linearize_units_all, rollout_units_all, backward_units_all;
} EmpcSolveStats;
int empc_solver_get_stats(EmpcSolver* s, EmpcSolveStats* stats);
/* diagnostic builds only (-DEMPC_STAMPS): in-kernel cycle stamps of the backward kernel, trajectory 0 */
int empc_solver_debug_counters(EmpcSolver* s, unsigned long long* out, int n);
int empc_solver_dims(const EmpcSolver* s, int* batch, int* T, int* nx, int* ndx, int* nu, int* rec_doubles);
/* 1 when empc_solver_create would accept this problem (a kernel instantiation exists for its (bodies, rotors, contact)
 * class and the problem passes the device-side limits), 0 otherwise with the reason in empc_last_error().  Needs no GPU.
 * Instantiated: (1,4) iris | (1,6) hexacopter370, hextilt | (3,6) hexacopter680_flying_arm_2 |
 * (4,6) hexacopter370_flying_arm_3: free, ContactModel3D, ContactModel6D and mixed-contact dynamics |
 * (6,6) hextilt_flying_arm_5: free, ContactModel3D and ContactModel6D dynamics.
 * Behind the environment switch EMPC_EXPERIMENTAL_CONTACT=1 (instantiations that have only run on the CPU lane emulator; refused
 * with that reason otherwise): contact dynamics on the (1,4) / (1,6) / (3,6) classes, stages of both contact types on (6,6), and
 * stages with TWO ContactModel3D contacts (the reference's ContactModelMultiple, src/stage.cpp:38-48) on (4,6) and (6,6). */
int empc_solver_supported(const EmpcProblemDesc* problem, const EmpcSolverParams* params);

/* ---- phase-level entry points (device kernels, one call = one launch over the whole batch) ------------------
 * xs: batch x (T+1) x nx, us: batch x T x nu. `smooth` is the squashing smoothness (0.1 in the first pass).
 * Outputs may be NULL. */
/* calc + calcDiff of every node: tape records batch x (T+1) x rec_doubles (layout: empc_tape_layout), cost per
 * trajectory, gaps are computed against the solver's x0s when is_feasible == 0 */
int empc_linearize_batch(EmpcSolver* s, const double* xs, const double* us, double smooth, int is_feasible,
                         double* tape, double* cost /* batch */, double* xnext /* batch x T x nx */);
/* backwardPass on the tape left by empc_linearize_batch; ok[b] = 0 where the LLT failed ("backward_error") */
int empc_backward_batch(EmpcSolver* s, double xreg, int is_feasible, double* K /* batch x T x nu x ndx */,
                        double* k /* batch x T x nu */, double* Vx /* batch x (T+1) x ndx */,
                        double* dgdq /* batch x 2 */, int* ok /* batch */);
/* forwardPass(alpha) from the current candidate and gains; ddp != 0 selects forwardPassDDP (no gap terms) */
int empc_rollout_batch(EmpcSolver* s, double alpha, int ddp, int is_feasible, double* xs_try, double* us_try,
                       double* cost_try /* batch */, int* ok /* batch */);
/* ---- step-wise entry points: one iteration from any iterate ---------------------------------------------------------
 * The loop body of solveFDDP / solveDDP (src/sbfddp.cpp:241-311, 329-389) as separately callable stages over the whole
 * batch, starting from per-trajectory solver scalars the caller supplies.  Purpose: teacher-forced parity -- put every
 * trajectory at an iterate recorded from a CPU solve (candidate via empc_solver_set_warmstart, x0 via
 * empc_solver_set_x0, scalars via empc_solver_set_states), run ONE iteration, and compare every intermediate (tape,
 * gains, the cost of every step length, accepted step, new regularisation, feasibility, stop decision) with the CPU
 * side.  Unlike empc_linearize_batch & co. these calls never reset the scalars. */
int empc_solver_get_states(EmpcSolver* s, EmpcTrajState* states /* batch */);
int empc_solver_set_states(EmpcSolver* s, const EmpcTrajState* states /* batch */);
/* runs the stages named in `stages` (EMPC_STAGE_* bits, in the order linearize, backward, rollout, select) once */
int empc_sweep_batch(EmpcSolver* s, int stages);
/* the line-search decision alone (select_decide_state: src/sbfddp.cpp:260-311, 348-389, 205-220): trial results are
 * taken from the arguments (batch x n_alphas each; NULL = what the last rollout left on the device) */
int empc_select_batch(EmpcSolver* s, const int* try_ok, const double* try_cost, const double* try_dv);
/* results of the last rollout for every step length: cost_try, the gap term dv of expectedImprovement, ok flag */
int empc_solver_get_trials(EmpcSolver* s, double* try_cost, double* try_dv, int* try_ok /* each batch x n_alphas */);
/* device buffers as they are: tape batch x (T+1) x rec (empc_tape_layout); K batch x T x nu x ndx; k batch x T x nu;
 * Vx batch x (T+1) x ndx.  Any pointer may be NULL. */
int empc_solver_get_tape(EmpcSolver* s, double* tape);
int empc_solver_get_gains(EmpcSolver* s, double* K, double* k, double* Vx);
/* overwrite the gains on the device (NULL = keep): the box solvers warm-start the QP of knot t at k_[t] of the previous
 * iteration, so an iterate of theirs includes k */
int empc_solver_set_gains(EmpcSolver* s, const double* K, const double* k);

/* ---- streamed solves ("continuous batching") ------------------------------------------------------------------------
 * n_jobs independent solves of the solver's problem -- solve([], [], maxiter) from the initial states x0s, what the
 * reference's benchmark loop does one after another (benchmark/utils/utils.hpp:15-27 + SolverSbFDDP::solve) -- pushed
 * through the solver's `batch` slots: a slot whose trajectory finishes writes its result row and takes the next job of the
 * queue inside the same sweep, so every sweep runs on a full batch until the queue is dry (a plain empc_solver_solve
 * spends most of its sweeps on the few trajectories that need many iterations).  Each job's arithmetic is that of a plain
 * solve (a trajectory never sees its batch neighbours), so its row is bitwise what empc_solver_solve gives for the same x0.
 *   stream_begin   copies the queue to the device and allocates the result rows there (inputs resident in HBM)
 *   stream_run     processes the whole queue
 *   stream_results copies rows out: xs | us | us_squash | cost | iters | status per job (row length via row_doubles;
 *                  rows == NULL only queries it) */
int empc_solver_stream_begin(EmpcSolver* s, int n_jobs, const double* x0s /* n_jobs x nx */);
int empc_solver_stream_run(EmpcSolver* s, int maxiter);
int empc_solver_stream_results(EmpcSolver* s, double* rows /* n_jobs x row */, int* row_doubles);
/* the same rows copied device to device (dst_device: n_jobs x row doubles of DEVICE memory, e.g. the send buffer of the
 * multi-GPU gather): nothing of the payload touches the host */
int empc_solver_stream_results_device(EmpcSolver* s, double* dst_device);
/* the HIP device the solver's memory lives on (queried from its problem image, not echoed from empc_solver_create) and that
 * device's PCI bus id "domain:bus:device.function" (pci_bus_id may be NULL; pci_len >= 16) -- the multi-GPU self-check of
 * bench.py compares it across ranks (one rank per GPU) */
int empc_solver_device_info(EmpcSolver* s, int* device_index, char* pci_bus_id, int pci_len);

/* offsets (in doubles) of the blocks inside one tape record */
typedef struct EmpcTapeLayout {
  int rec, off_fx, off_fu, off_lxx, off_lxu, off_luu, off_lx, off_lu, off_gap, off_cost;
  int ld_fx, ld_fu, ld_lxx, ld_lxu, ld_luu; /* leading dimensions (row strides) of the matrix blocks */
} EmpcTapeLayout;
int empc_tape_layout(const EmpcSolver* s, EmpcTapeLayout* layout);

/* ---- plant of closed-loop MPC runs --------------------------------------------------------------------------
 * Reference: AerialSimulator (bindings/python/eagle_mpc/utils/simulator.py:8-29): FreeFwdDynamics with the unsquashed
 * multicopter actuation, IntegratedActionModelRK4, no costs.  One plant per trajectory of the batch, states resident on
 * the device next to the solver so that a closed-loop step (plant -> x0 -> solve) moves no state through the host. */
int empc_plant_set_state(EmpcSolver* s, const double* x /* batch x nx */);
int empc_plant_get_state(EmpcSolver* s, double* x /* batch x nx */);
/* x <- RK4(x, u, dt_s), n_substeps times.  u: batch x nu rotor thrusts + arm torques, or NULL = the squashed first
 * control of the last solve (control = solver.us_squash[0], examples/python/mpc.py:60) */
int empc_plant_step(EmpcSolver* s, double dt_s, const double* u, int n_substeps);
/* problem.x0 = simulator.states[-1] for every trajectory (examples/python/mpc.py:50), device to device */
int empc_solver_set_x0_from_plant(EmpcSolver* s);

/* ---- Carrot MPC controller (include/eagle_mpc/mpc-controllers/carrot-mpc.hpp:23-88) ----------------------------
 * Host-side logic only: builds the receding-horizon problem (one private cost set per knot) and re-targets its
 * carrot references for a given time; pass empc_carrot_mpc_problem_desc() to empc_solver_create / _update_problem. */
typedef struct EmpcCarrotMpc EmpcCarrotMpc;
/* CarrotMpc(trajectory, state_ref, dt_ref, yaml_path)   src/mpc-controllers/carrot-mpc.cpp:15-50 */
EmpcCarrotMpc* empc_carrot_mpc_create(const EmpcTrajectory* t, const double* state_ref /* n_ref x nx */, int n_ref,
                                      int dt_ref_ms, const char* mpc_yaml_path);
void empc_carrot_mpc_destroy(EmpcCarrotMpc* m);
int empc_carrot_mpc_params(const EmpcCarrotMpc* m, int* knots, int* iters, int* dt_ms, int* nx, int* ndx, int* nu,
                           int* n_t_stages);
int empc_carrot_mpc_t_stages(const EmpcCarrotMpc* m, long long* t_stages /* n_t_stages */);
/* updateProblem(current_time)   src/mpc-controllers/carrot-mpc.cpp:298-313 */
int empc_carrot_mpc_update_problem(EmpcCarrotMpc* m, long long current_time_ms);
/* computeStateReference(time)   src/mpc-controllers/carrot-mpc.cpp:384-403 */
int empc_carrot_mpc_state_reference(EmpcCarrotMpc* m, long long time_ms, double* xref /* nx */);
/* problem.x0 = x0 */
int empc_carrot_mpc_set_x0(EmpcCarrotMpc* m, const double* x0 /* nx */);
const EmpcProblemDesc* empc_carrot_mpc_problem_desc(EmpcCarrotMpc* m);

/* ---- Rail / Weighted MPC controllers ----------------------------------------------------------------------------
 * (include/eagle_mpc/mpc-controllers/rail-mpc.hpp:24-60, weighted-mpc.hpp:24-70; both derive MpcAbstract,
 * include/eagle_mpc/mpc-base.hpp:59-119).  Host-side logic only, same use as EmpcCarrotMpc: update the cost tables for a
 * time, hand empc_mpc_problem_desc() to empc_solver_create / empc_solver_update_problem. */
typedef struct EmpcMpc EmpcMpc;
/* RailMpc(state_ref, dt_ref, yaml_path)   src/mpc-controllers/rail-mpc.cpp:14-60 */
EmpcMpc* empc_rail_mpc_create(const double* state_ref /* n_ref x nx */, int n_ref, int nx, int dt_ref_ms,
                              const char* mpc_yaml_path);
/* WeightedMpc(trajectory, dt_ref, yaml_path)   src/mpc-controllers/weighted-mpc.cpp:16-72.  Like the reference this
 * EDITS the trajectory: each transition stage is merged into the stage after it. */
EmpcMpc* empc_weighted_mpc_create(EmpcTrajectory* t, int dt_ref_ms, const char* mpc_yaml_path);
void empc_mpc_destroy(EmpcMpc* m);
/* get_knots / get_iters / get_dt (src/mpc-base.cpp:81-85) and the problem's dimensions */
int empc_mpc_params(const EmpcMpc* m, int* knots, int* iters, int* dt_ms, int* nx, int* ndx, int* nu);
/* One cost entry of a node's model, edited in place between two solves -- residual->set_reference(...), cost->weight,
 * cost->active as MpcAbstract::updateProblem applies them to the crocoddyl models (src/mpc-controllers/carrot-mpc.cpp:298-359,
 * rail-mpc.cpp:151-161, weighted-mpc.cpp:170-243).  `knot` in [0, T]; the cost is named as in the YAML / CostModelSum.
 * ref: new reference payload (layout of EmpcCost.ref; NULL = keep), active: 0 / 1 (-1 = keep), weight: NaN = keep.  Nodes
 * that share a cost set (the knots of one trajectory stage) change together, like the reference's shared models.  The edit is
 * uploaded before the next solve / phase call; empc_solver_update_problem replaces the whole table instead. */
int empc_solver_set_cost_refs(EmpcSolver* s, int knot, const char* cost_name, const double* ref, int nref, int active, double weight);
/* get_solver_type() (src/mpc-base.cpp:85): the EmpcSolverType the controller's YAML names (`solver: SolverSbFDDP |
 * SolverBoxFDDP | SolverBoxDDP`); pass the handle you have and NULL for the other.  Create the solver with that
 * EmpcSolverParams.solver_type (src/mpc-controllers/carrot-mpc.cpp:232-242). */
int empc_mpc_solver_type(const EmpcCarrotMpc* carrot, const EmpcMpc* other);
/* updateProblem(current_time)   rail-mpc.cpp:151-161, weighted-mpc.cpp:170-185 */
int empc_mpc_update_problem(EmpcMpc* m, long long current_time_ms);
/* problem.x0 = x0 */
int empc_mpc_set_x0(EmpcMpc* m, const double* x0 /* nx */);
const EmpcProblemDesc* empc_mpc_problem_desc(EmpcMpc* m);
/* RailMpc::computeStateReference(time)   rail-mpc.cpp:176-200 (error for a WeightedMpc handle) */
int empc_rail_mpc_state_reference(EmpcMpc* m, long long time_ms, double* xref /* nx */);
/* WeightedMpc::get_t_stages()   weighted-mpc.cpp:246; returns the count, fills t_stages when not NULL
 * (error for a RailMpc handle) */
int empc_weighted_mpc_t_stages(const EmpcMpc* m, long long* t_stages, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* EMPC_H */
