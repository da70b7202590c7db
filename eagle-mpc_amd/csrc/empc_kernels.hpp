// empc_kernels.hpp -- bodies of the HIP kernels of the batched Squash-box FDDP solver.
//
//   linearize_unit2 HOT-A  (empc_linearize2.hpp) one (trajectory, node) unit per half wavefront: IAM calcDiff
//   backward_traj4  HOT-B  (empc_backward4.hpp) one wavefront per trajectory: Riccati sweep on the matrix cores, gains
//   rollout_group6  HOT-C  (empc_rollout6.hpp) packed trajectories, four role wavefronts per workgroup
//   rollout_thread         the same forward pass with one lane per (trajectory, step length): RK4 nodes, > 16 step lengths
//   calc_thread            one lane per (trajectory, node): IAM calc at the current candidate (phase starts)
//   select_traj            line-search acceptance, regularisation, stopping tests, continuation schedule
//
// The bodies are written against an executor (Exec) so that tests can run them lane by lane on the CPU
// (tests/csrc/lane_emulator.cpp); on the GPU the executor is the identity and every "sync" is a wave-level fence --
// a unit never spans more than one wavefront, so no workgroup barrier is needed anywhere.
//
// Reference semantics: src/sbfddp.cpp (solve loop, acceptance rules, barrier/squash schedule) and SURVEY.md
// Appendix A.1-A.7 (Crocoddyl / Pinocchio behaviour).  Everything is FP64.
#pragma once
#include "empc_dev_model.hpp"

namespace empc {

constexpr int PHASE_DDP = 100;
constexpr int PHASE_DONE = 255;
constexpr int MAX_ALPHAS = 16;

// the per-trajectory solver scalars are part of the C ABI (step-wise entry points): include/empc_types.h
typedef EmpcTrajState TrajState;

// Per-cost-set work lists, built once on the host (prepare_problem) so that the kernels walk only the costs a stage needs
// instead of scanning the whole table with dependent scalar loads: indices into EmpcCostSet::costs, active costs only,
// each list in table (= alphabetical = CostModelSum) order.
struct SetInfo {
  int ncap, capf[NCAP], ccap;  // operational frames captured by the bias pass: frame costs first, then the contact frame
  int n_sc, n_state, n_ctrl, n_frame, n_cone;
  int sc_ci[EMPC_MAX_COSTS];     // State and Control costs, interleaved as they appear in the table (order of the cost sum)
  int state_ci[EMPC_MAX_COSTS];  // State costs
  int ctrl_ci[EMPC_MAX_COSTS];   // Control costs (incl. the solver's barrier)
  int frame_ci[EMPC_MAX_COSTS];  // FramePlacement / Rotation / Translation / Velocity costs
  int cone_ci[EMPC_MAX_COSTS];   // ContactFrictionCone costs
  int state_own[EMPC_MAX_COSTS]; // per entry of state_ci: position (in state_ci) of the first State cost with the same reference
};

struct DevBuffers {
  const DevProblem* P;
  const EmpcCostSet* sets;
  const SetInfo* set_info;  // [n_sets]
  const int* knot_set;
  const int* lin_knots;  // [T+1] knots sorted: first the n_lean knots whose cost set captures no operational frames, then the rest
  int n_lean;            // linearize runs its lean body over the first group and the full body over the second
  TrajState* st;
  double* x0;       // [B][NX]
  double* xs;       // [B][T+1][NX]
  double* us;       // [B][T][NU]
  double* acc;      // [B][T+1][NV]
  double* tape;     // [B][T+1][REC]
  double* K;        // [B][T][NU*NDX]
  double* kff;      // [B][T][NU]
  double* Vx;       // [B][T+1][NDX]
  double* Vf;       // [B][T+1][NDX]   Vxx[t] fs[t]
  double* xs_try;   // [B][NA][T+1][NX]
  double* us_try;   // [B][NA][T][NU]
  double* acc_try;  // [B][NA][T+1][NV]
  double* try_cost; // [B][NA]
  double* try_dv;   // [B][NA]
  int* try_ok;      // [B][NA]
  int* try_ncalc;   // [B][NA]  running nodes whose calc ran in this trial: T, or (failing knot + 1) after a "forward_error"
  double* us_last;  // [B][T][NU]  control of the last IAM.calc at every node (fillSquashedOutputs semantics)
  int* n_active;    // [1]
  // trajectories that linearize in this sweep, written by the previous sweep's select (compact, any order); nullptr =
  // every trajectory.  linearize is the one throughput-bound kernel: with the list its time follows the number of
  // trajectories still iterating instead of the number of workgroups that hold at least one of them.
  int lin_bound = 0;  // host-side upper bound on the length of lin_list (the active count two sweeps back; 0 = B): sizes the grid
  const int* lin_list = nullptr;
  const int* lin_count = nullptr;
  int* lin_list_out = nullptr;   // the list select builds for the next sweep (nullptr = none)
  int* lin_count_out = nullptr;
  // trajectories still iterating (phase != DONE), same hand-over: the packed rollout forms its wavefronts from this list,
  // so a sweep with few stragglers launches few workgroups.  act_count is the previous sweep's n_active.
  const int* act_list = nullptr;
  const int* act_count = nullptr;
  int* act_list_out = nullptr;
  // trajectories whose pass starts in the next sweep (need_calc): the calc kernel walks this list instead of scanning the batch
  const int* calc_list = nullptr;
  const int* calc_count = nullptr;
  int* calc_list_out = nullptr;
  int* calc_count_out = nullptr;
  // hand-over of the counters without host commands in the stream: select zeroes the counters of the NEXT sweep's slot
  // (nobody reads them any more) and the last workgroup to finish publishes the active count to pinned host memory
  int* counters_next = nullptr;   // {n_active, lin_count} of the other sweep slot
  int* done_ticket = nullptr;     // workgroups of this select that have finished
  int* host_active = nullptr;     // host-visible copy of n_active of this sweep
  unsigned long long* dbg;  // [128] cycle stamps of diagnostic builds (EMPC_STAMPS); unused otherwise
  // optional per-iteration record (the reference's callback hook, src/sbfddp.cpp:303-307,381-385): ring of trace_cap
  // records of EMPC_TRACE_WORDS doubles per trajectory, written by select; nullptr = off
  double* trace = nullptr;  // [B][trace_cap][EMPC_TRACE_WORDS]
  int trace_cap = 0;
  // streamed solves (empc_solver_stream_*): queue of initial states and the result rows, both resident on the device; a
  // trajectory that finishes hands its row over and its slot takes the next job inside select.  nullptr = plain solve.
  const double* q_x0 = nullptr;            // [q_njobs][NX]
  double* q_rows = nullptr;                // [q_njobs][(T+1) NX + 2 T NU + 3]: xs | us | us_squash | cost | iters | status
  int* q_head = nullptr;                   // next job to hand out
  unsigned long long* q_iters = nullptr;   // DDP iterations of the finished jobs, summed
  int q_njobs = 0, q_maxiter = 0;
  int B, T, NA;
  int integrator = 0;  // EmpcIntegrator of the problem (host copy: selects the kernel forms that support it)
  int solver_type = 0; // EmpcSolverType (host copy): the box solvers need the kernel forms that implement them
  int raw = 0;         // linearize: write the differential model's derivatives (da/dx, da/du, unscaled costs) instead of the
                       // Euler node's -- the stage records of IntegratedActionModelRK4 (empc_rk4.hpp)
  double gaptol;    // feasibility tolerance actually used: max(th_gaptol, 1e-13)
};

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#else
inline void wave_sync() {}
#endif

// GPU executor: one lane; the CPU emulator provides the same interface with a loop over lanes.
struct LaneExec {
  int lane;
  static constexpr int SLOTS = 1;
  template <class F>
  EMPC_HD void each(F&& f) {
    f(lane, 0);
  }
  EMPC_HD void sync() { wave_sync(); }
  // workgroup barrier (role split of linearize)
  EMPC_HD void block_sync() {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
  }
};

// =====================================================================================================================
// calc: IAM.calc at (xs[t], us[t]) -> acc (and the last-calc control).  One lane per (b, t).
// =====================================================================================================================
template <class DM, int CT>
EMPC_HD void calc_thread(const DevBuffers& D, int b, int t) {
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || !st.need_calc) return;
  const int T = D.T;
  const double* x = D.xs + ((size_t)b * (T + 1) + t) * DM::NX;
  const double* u = (t < T) ? D.us + ((size_t)b * T + t) * DM::NU : nullptr;
  double xnext[DM::NX], acc[DM::NV], usq[DM::NU], lam[6], cost;
  node_nominal<DM, CT>(EMPC_KREF(DevProblem, D.P), EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]], st.smooth, x, u, t == T, xnext, acc, cost, usq, lam);
  double* ao = D.acc + ((size_t)b * (T + 1) + t) * DM::NACC;
  for (int i = 0; i < DM::NV; ++i) ao[i] = acc[i];
  for (int i = 0; i < 6; ++i) ao[DM::NV + i] = lam[i];
  if (t < T) {
    double* ul = D.us_last + ((size_t)b * T + t) * DM::NU;
    for (int i = 0; i < DM::NU; ++i) ul[i] = u[i];
  }
}

// =====================================================================================================================
// rollout: SolverFDDP::forwardPass(alpha) / SolverSbFDDP::forwardPassDDP(alpha).  One lane per (b, alpha index).
// =====================================================================================================================
// Per-lane state of one (trajectory, step length) trial across the knots.
template <class DM>
struct RollLane {
  double xnext[DM::NX];
  double cost_try, dv;
  int ok;
};

// One knot of the forward pass for one trial.  The knot's nominal data comes through typed pointers so that the same
// body serves any placement of that data (k_rollout reads it per lane from global memory).  Returns false when the trial
// failed (NaN / overflow, crocoddyl's raiseIfNaN) and the caller must stop.
template <class DM, int CT, class SetT, class PX, class PU, class PK, class PG>
EMPC_HD bool rollout_knot(const EMPC_K DevProblem& P, const SetT& set, RollLane<DM>& L, int t, int T, bool plain, bool need_dv,
                          double alpha, double smooth, PX xc, PU uc, PU kk, PK KK, PG gap, PG vf, double* xs_o, double* us_o,
                          double* ac_o, unsigned long long* stp) {
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX;
  double xtry[NX], dx[NDX], utry[NU], acc[NV], usq[NU], lam[6];
  EMPC_STAMP(9);  // stores / loop tail of the previous knot
  if (plain) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xtry[i] = L.xnext[i];
  } else {
    double step[NDX];
#pragma unroll
    for (int i = 0; i < NDX; ++i) step[i] = gap[i] * (alpha - 1.0);
    state_integrate<DM>(L.xnext, step, xtry, nullptr);
  }
  state_diff<DM>(xc, xtry, dx, nullptr);
  if (need_dv) {
    double dv = L.dv;
#pragma unroll
    for (int i = 0; i < NDX; ++i) dv += vf[i] * dx[i];  // -f^T Vxx (xs (-) xs_try) = +(Vxx f).(xs_try (-) xs)
    L.dv = dv;
  }
  double cost;
  if (t < T) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double a_ = uc[i] - kk[i] * alpha;
#pragma unroll
      for (int j = 0; j < NDX; ++j) a_ -= KK[i * NDX + j] * dx[j];
      // SolverBox{DDP,FDDP}::forwardPass clamp the trial control to the limits of the model
      utry[i] = (P.prm.solver_type != EMPC_SOLVER_SBFDDP) ? fmin(fmax(a_, P.u_lb[i]), P.u_ub[i]) : a_;
    }
    EMPC_STAMP(0);  // x_try, state difference, feedback
    node_nominal<DM, CT>(P, set, smooth, xtry, utry, false, L.xnext, acc, cost, usq, lam, stp);
#pragma unroll
    for (int i = 0; i < NU; ++i) us_o[(size_t)t * NU + i] = utry[i];
  } else {
    double xn2[NX];
    node_nominal<DM, CT>(P, set, smooth, xtry, (const double*)nullptr, true, xn2, acc, cost, usq, lam, stp);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) xs_o[(size_t)t * NX + i] = xtry[i];
#pragma unroll
  for (int i = 0; i < NV; ++i) ac_o[(size_t)t * DM::NACC + i] = acc[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) ac_o[(size_t)t * DM::NACC + NV + i] = lam[i];
  L.cost_try += cost;
  if (bad_number(L.cost_try)) {
    L.ok = 0;
    return false;
  }
  if (t < T) {
    double mx = 0;
    bool isn = false;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      mx = fmax(mx, fabs(L.xnext[i]));
      isn = isn || is_nan(L.xnext[i]);
    }
    if (isn || bad_number(mx)) {
      L.ok = 0;
      return false;
    }
  }
  return true;
}

template <class DM, int CT>
EMPC_HD void rollout_thread(const DevBuffers& D, int b, int ai) {
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || st.bwd_failed) return;
  constexpr int NX = DM::NX, NU = DM::NU, NDX = DM::NDX, REC = DM::REC;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T, NA = D.NA;
  const bool ddp = (st.phase == PHASE_DDP);
  const bool feas = st.is_feasible != 0;
  const double alpha = ldexp(1.0, -ai);
  const bool plain = ddp || feas || (ai == 0);
  const double smooth = st.smooth;
  RollLane<DM> L;
  for (int i = 0; i < NX; ++i) L.xnext[i] = D.x0[(size_t)b * NX + i];
  L.cost_try = 0;
  L.dv = 0;
  L.ok = 1;
  const size_t slot = (size_t)b * NA + ai;
  double* xs_o = D.xs_try + slot * (T + 1) * NX;
  double* us_o = D.us_try + slot * T * NU;
  double* ac_o = D.acc_try + slot * (T + 1) * DM::NACC;
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long stamps[32];
  for (int i = 0; i < 32; ++i) stamps[i] = 0;
  unsigned long long* stp = (b == 0 && ai == 1) ? stamps : nullptr;
  if (stp) stp[31] = __builtin_readcyclecounter();
#else
  unsigned long long* stp = nullptr;
#endif
  int ncalc = 0;  // knots completed without a failure
  for (int t = 0; t <= T; ++t) {
    const double* rec = D.tape + ((size_t)b * (T + 1) + t) * REC;
    const double* xc = D.xs + ((size_t)b * (T + 1) + t) * NX;
    const double* vf = D.Vf + ((size_t)b * (T + 1) + t) * NDX;
    const int tt = (t < T) ? t : 0;  // unused at the terminal node
    const double* uc = D.us + ((size_t)b * T + tt) * NU;
    const double* kk = D.kff + ((size_t)b * T + tt) * NU;
    const double* KK = D.K + ((size_t)b * T + tt) * NU * NDX;
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
    if (!rollout_knot<DM, CT>(P, set, L, t, T, plain, !ddp && !feas, alpha, smooth, xc, uc, kk, KK, rec + DM::OFF_GAP, vf, xs_o,
                              us_o, ac_o, stp))
      break;
    ncalc = (t + 1 < T) ? t + 1 : T;
  }
  D.try_cost[slot] = L.cost_try;
  D.try_dv[slot] = L.dv;
  D.try_ok[slot] = L.ok;
  D.try_ncalc[slot] = L.ok ? T : ((ncalc + 1 < T) ? ncalc + 1 : T);  // the failing knot's calc ran too
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  if (stp)
    for (int i = 0; i < 16; ++i) D.dbg[i] = stamps[i];
#endif
}

// State of one trajectory at the start of SolverSbFDDP::solve (src/sbfddp.cpp:198-210).  `prev` carries the members that
// the reference keeps across solve() calls on one solver object (cost_, cost_prev_, stop_); nullptr = a fresh solver.
template <class PrmT>
EMPC_HD void traj_state_init(TrajState& s, const PrmT& prm, int maxiter, bool is_feasible_arg, const TrajState* prev) {
  TrajState z;
  z.phase = z.iter = z.total_iters = z.status = z.is_feasible = z.was_feasible = 0;
  z.bwd_failed = z.trace_count = z.last_ok = z.reserved = 0;
  z.accepted_alpha = z.last_alpha = -1;
  z.job = -1;
  z.cost = prev ? prev->cost : 0.0;
  z.cost_prev = prev ? prev->cost_prev : 0.0;
  z.stop = prev ? prev->stop : 0.0;
  z.gapnorm = prev ? prev->gapnorm : 0.0;
  z.dV = z.dVexp = z.d0 = z.d1 = z.dg_u = z.dq_u = z.dg_f = z.dq_f = z.qu2 = 0.0;
  z.maxiter = maxiter;
  z.smooth = z.smooth_next = prm.smooth_init;
  z.convergence = prm.convergence_init;
  z.th_stop = prm.convergence_init;
  z.xreg = z.ureg = prm.reg_init;
  z.steplength = 1.0;
  z.need_calc = 1;
  z.need_lin = 1;
  if (prm.solver_type != EMPC_SOLVER_SBFDDP) {
    // crocoddyl::SolverBoxFDDP / SolverBoxDDP::solve: one loop from the candidate's feasibility flag, th_stop_ = 5e-5
    z.phase = (prm.solver_type == EMPC_SOLVER_BOXDDP) ? PHASE_DDP : 0;
    z.is_feasible = is_feasible_arg ? 1 : 0;
    z.th_stop = prm.box_th_stop;
  } else if (prm.convergence_init >= prm.convergence_stop) {
    z.phase = 0;
    z.is_feasible = 0;  // solveFDDP(maxiter, false, reg_init_)
  } else if (!is_feasible_arg) {
    z.phase = PHASE_DDP;
    z.is_feasible = 0;
    z.status |= EMPC_STATUS_DDP_CLEANUP;
  } else {
    z.phase = PHASE_DONE;
    z.is_feasible = 1;
    z.iter = -1;
  }
  s = z;
}

// =====================================================================================================================
// select: one trajectory; the serial decision logic of solveFDDP / solveDDP / solve after the trial rollouts.
// `tid`/`nthreads` cooperate on the candidate copy; all threads must call it.
// =====================================================================================================================
template <int NAMAX>
EMPC_HD void select_decide_state(const DevBuffers& D, int b, TrajState& st, const int* try_ok_v, const double* try_cost_v,
                                 const double* try_dv_v, int& accepted_ai, int& last_ai) {
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcSolverParams& prm = P.prm;
  const int NA = D.NA;
  const bool ddp = (st.phase == PHASE_DDP);
  bool phase_end = false, returned = false;
  if (st.bwd_failed) {
    // computeDirection gave up at reg_max: solveFDDP/solveDDP return false without finishing the iteration
    st.status |= EMPC_STATUS_REG_MAX;
    phase_end = true;
  } else {
    st.need_lin = 0;
    st.need_calc = 0;
    if (ddp) {  // expectedImprovementDDP() runs once, before the line search (src/sbfddp.cpp:344): d_ is set even if every trial fails
      st.d0 = st.dg_u;
      st.d1 = st.dq_u;
    }
    // (no break / continue: the loop unrolls, so the register copies of the trial results are indexed at compile time)
    bool searching = true;
#pragma unroll
    for (int ai = 0; ai < NAMAX; ++ai) {
      if (!(searching && ai < NA)) continue;
      const double alpha = ldexp(1.0, -ai);
      st.steplength = alpha;
      last_ai = ai;
      if (try_ok_v[ai]) {  // else "forward_error": next step length
      st.dV = st.cost - try_cost_v[ai];
      if (!ddp) {
        const double dv = st.is_feasible ? 0.0 : try_dv_v[ai];
        const double dg = st.dg_u + (st.is_feasible ? 0.0 : st.dg_f);
        const double dq = st.dq_u + (st.is_feasible ? 0.0 : st.dq_f);
        st.d0 = dg + dv;
        st.d1 = dq - 2.0 * dv;
      } else {
        st.d0 = st.dg_u;
        st.d1 = st.dq_u;
      }
      st.dVexp = alpha * (st.d0 + 0.5 * alpha * st.d1);
      bool acc = false;
      if (st.dVexp >= 0) {
        if (!ddp)
          acc = (st.d0 < prm.th_grad) || (st.dV > prm.th_acceptstep * st.dVexp);
        else
          acc = (st.d0 < prm.th_grad) || (!st.is_feasible) || (st.dV > prm.th_acceptstep * st.dVexp);
      } else if (!ddp) {
        acc = st.dV > prm.th_acceptnegstep * st.dVexp;
      }
      if (acc) {
        st.was_feasible = st.is_feasible;
        st.is_feasible = ddp ? 1 : ((st.was_feasible || alpha == 1.0) ? 1 : 0);
        st.cost_prev = st.cost;
        st.cost = try_cost_v[ai];
        st.need_lin = 1;
        accepted_ai = ai;
        searching = false;
      }
      }
    }
    if (st.steplength > prm.th_stepdec) {
      st.xreg /= prm.reg_decfactor;
      if (st.xreg < prm.reg_min) st.xreg = prm.reg_min;
      st.ureg = st.xreg;
    }
    bool regmax = false;
    if (st.steplength <= prm.th_stepinc) {
      st.xreg *= prm.reg_incfactor;
      if (st.xreg > prm.reg_max) st.xreg = prm.reg_max;
      st.ureg = st.xreg;
      if (st.xreg == prm.reg_max) regmax = true;
    }
    if (regmax) {
      st.status |= EMPC_STATUS_REG_MAX;
      phase_end = true;
    } else {
      // stoppingCriteria (fork, U1); the crocoddyl box solvers use SolverDDP::stoppingCriteria = sum |Qu|^2
      const bool box = prm.solver_type != EMPC_SOLVER_SBFDDP;
      if (box)
        st.stop = st.qu2;
      else if (prm.stop_criteria == EMPC_STOP_COST_REDUCTION)
        st.stop = fabs(st.cost_prev - st.cost);
      else if (prm.stop_criteria == EMPC_STOP_EXPECTED_REDUCTION)
        st.stop = fabs(st.d0 + 0.5 * st.d1);
      else
        st.stop = st.qu2;
      if (D.trace) {
        // what a crocoddyl callback sees at this point of solveFDDP / solveDDP (same fields as oracle::IterRecord)
        double* r = D.trace + ((size_t)b * D.trace_cap + (st.trace_count % D.trace_cap)) * EMPC_TRACE_WORDS;
        r[0] = (double)st.phase;
        r[1] = (double)st.iter;
        r[2] = st.cost;
        r[3] = st.stop;
        r[4] = st.xreg;
        r[5] = st.steplength;
        r[6] = st.is_feasible ? 1.0 : 0.0;
        r[7] = st.dV;
        r[8] = st.dVexp;
        r[9] = st.gapnorm;
        r[10] = st.d0;
        r[11] = st.d1;
        st.trace_count += 1;
      }
      // SolverDDP / SolverFDDP::solve (and the fork's solveDDP) stop on was_feasible_ && stop_ < th_stop_; the fork's solveFDDP
      // on its gap test
      const bool stop_now = (ddp || box) ? (st.was_feasible && st.stop < st.th_stop)
                                         : (st.stop < st.th_stop && st.gapnorm < prm.th_stop_gaps);
      if (stop_now) {
        phase_end = true;
        returned = true;
      } else {
        st.iter += 1;
        if (st.iter >= st.maxiter) {
          st.iter = st.maxiter - 1;
          st.status |= EMPC_STATUS_MAXITER;
          phase_end = true;
        }
      }
    }
  }
  st.last_ok = returned ? 1 : 0;
  if (phase_end) {
    st.total_iters += st.iter + 1;
    bool next_fddp = false;
    if (prm.solver_type != EMPC_SOLVER_SBFDDP) {
      // SolverBoxFDDP / SolverBoxDDP: a single loop, no continuation and no clean-up pass
      st.phase = PHASE_DONE;
      st.iter = st.total_iters - 1;
      if (st.last_ok) st.status |= EMPC_STATUS_CONVERGED;
      st.bwd_failed = 0;
      return;
    }
    if (!ddp) {
      st.smooth_next *= prm.smooth_mult;
      st.convergence *= prm.convergence_mult;
      next_fddp = st.convergence >= prm.convergence_stop;
    }
    if (next_fddp) {
      st.phase += 1;
      st.smooth = st.smooth_next;  // squashingUpdate + barrierUpdate
      st.th_stop = st.convergence;
      st.is_feasible = 0;
      st.was_feasible = 0;
      st.xreg = st.ureg = prm.reg_init;
      st.iter = 0;
      st.need_calc = 1;
      st.need_lin = 1;
      st.status &= ~(EMPC_STATUS_MAXITER | EMPC_STATUS_REG_MAX);
    } else if (!ddp && !st.is_feasible) {
      st.phase = PHASE_DDP;
      st.was_feasible = 0;
      st.xreg = st.ureg = prm.reg_init;
      st.iter = 0;
      st.need_calc = 1;
      st.need_lin = 1;
      st.status &= ~(EMPC_STATUS_MAXITER | EMPC_STATUS_REG_MAX);
      st.status |= EMPC_STATUS_DDP_CLEANUP;
    } else {
      st.phase = PHASE_DONE;
      st.iter = st.total_iters - 1;
      if (st.last_ok) st.status |= EMPC_STATUS_CONVERGED;
    }
  }
  st.bwd_failed = 0;
}

template <class DM>
EMPC_HD void select_decide(const DevBuffers& D, int b, int& accepted_ai, int& last_ai) {
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcSolverParams& prm = P.prm;
  const int NA = D.NA;
  accepted_ai = -1;
  last_ai = -1;
  if (D.st[b].phase == PHASE_DONE) return;
  // The state machine runs on a register copy of the trajectory's state and on the trial results fetched in one batch:
  // one thread walking `D.st[b].field` and `D.try_*[slot]` in place pays a memory round trip per step of the walk (the
  // floor of this kernel when few trajectories are left).
  TrajState st = D.st[b];
  int try_ok_v[MAX_ALPHAS];
  double try_cost_v[MAX_ALPHAS], try_dv_v[MAX_ALPHAS];
#pragma unroll
  for (int ai = 0; ai < MAX_ALPHAS; ++ai) {
    const size_t slot = (size_t)b * NA + (ai < NA ? ai : 0);
    try_ok_v[ai] = D.try_ok[slot];
    try_cost_v[ai] = D.try_cost[slot];
    try_dv_v[ai] = D.try_dv[slot];
  }
  select_decide_state<MAX_ALPHAS>(D, b, st, try_ok_v, try_cost_v, try_dv_v, accepted_ai, last_ai);
  st.accepted_alpha = accepted_ai;
  st.last_alpha = last_ai;
  D.st[b] = st;
}

// copy helper used by select: candidate <- trial slot `ai` (xs, us, acc); threads cooperate.  Four independent loads per
// thread are issued before the first store (the copy is a chain of memory latencies otherwise).
EMPC_HD void copy_doubles(double* dst, const double* src, int n, int tid, int nthreads) {
  int i = tid;
  for (; i + 3 * nthreads < n; i += 4 * nthreads) {
    const double a = src[i], b = src[i + nthreads], c = src[i + 2 * nthreads], d = src[i + 3 * nthreads];
    dst[i] = a;
    dst[i + nthreads] = b;
    dst[i + 2 * nthreads] = c;
    dst[i + 3 * nthreads] = d;
  }
  for (; i < n; i += nthreads) dst[i] = src[i];
}
template <class DM>
EMPC_HD void select_copy(const DevBuffers& D, int b, int accepted_ai, int last_ai, int tid, int nthreads) {
  const int T = D.T, NA = D.NA;
  if (accepted_ai >= 0) {
    const size_t slot = (size_t)b * NA + accepted_ai;
    copy_doubles(D.xs + (size_t)b * (T + 1) * DM::NX, D.xs_try + slot * (T + 1) * DM::NX, (T + 1) * DM::NX, tid, nthreads);
    copy_doubles(D.us + (size_t)b * T * DM::NU, D.us_try + slot * T * DM::NU, T * DM::NU, tid, nthreads);
    copy_doubles(D.acc + (size_t)b * (T + 1) * DM::NACC, D.acc_try + slot * (T + 1) * DM::NACC, (T + 1) * DM::NACC, tid, nthreads);
  }
  if (last_ai >= 0) {
    // fillSquashedOutputs reads the data of the LAST calc at every node.  The step lengths are tried one after the other
    // (0 .. last_ai) and a trial that fails ("forward_error") stops at its failing knot: node t keeps the control of the last
    // trial whose rollout reached it -- or what the calc of the iterate left there when none did
    const size_t slot = (size_t)b * NA + last_ai;
    if (D.try_ncalc[slot] >= T) {
      copy_doubles(D.us_last + (size_t)b * T * DM::NU, D.us_try + slot * T * DM::NU, T * DM::NU, tid, nthreads);
    } else {
      for (int i = tid; i < T * DM::NU; i += nthreads) {
        const int t = i / DM::NU;
        for (int aj = last_ai; aj >= 0; --aj)
          if (D.try_ncalc[(size_t)b * NA + aj] > t) {
            D.us_last[(size_t)b * T * DM::NU + i] = D.us_try[((size_t)b * NA + aj) * T * DM::NU + i];
            break;
          }
      }
    }
  }
}

// ---- streamed solves: hand-over of a finished trajectory's slot (called by select; threads cooperate) -------------------
template <class DM>
EMPC_HD size_t stream_row_doubles(int T) {
  return (size_t)(T + 1) * DM::NX + 2 * (size_t)T * DM::NU + 3;
}
// result row of the job in slot b: xs | us | us_squash | cost | iters | status (us_squash: fillSquashedOutputs semantics,
// the squashed control of the LAST calc at every node, as empc_solver_get_us_squash reports it)
template <class DM>
EMPC_HD void stream_write_row(const DevBuffers& D, int b, int tid, int nthreads) {
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const TrajState& st = D.st[b];
  const int T = D.T;
  const size_t nxs = (size_t)(T + 1) * DM::NX, nus = (size_t)T * DM::NU;
  double* row = D.q_rows + (size_t)st.job * stream_row_doubles<DM>(T);
  const double* xs = D.xs + (size_t)b * nxs;
  const double* us = D.us + (size_t)b * nus;
  const double* ul = D.us_last + (size_t)b * nus;
  for (size_t i = tid; i < nxs; i += nthreads) row[i] = xs[i];
  for (size_t i = tid; i < nus; i += nthreads) {
    row[nxs + i] = us[i];
    double u = us[i], du;
    const int c = (int)(i % DM::NU);
    if (P.use_squash) squash1(ul[i], P.u_lb[c], P.u_ub[c], st.smooth, P.prm.smoothsat_power, u, du);
    row[nxs + nus + i] = u;
  }
  if (tid == 0) {
    row[nxs + 2 * nus] = st.cost;
    row[nxs + 2 * nus + 1] = (double)st.iter;
    row[nxs + 2 * nus + 2] = (double)st.status;
  }
}
// the slot takes job `job`: setCandidate([], []) (zero state at every node, zero controls), problem.x0 = the job's state
template <class DM>
EMPC_HD void stream_refill(const DevBuffers& D, int b, int job, int tid, int nthreads) {
  const int T = D.T;
  const size_t nxs = (size_t)(T + 1) * DM::NX, nus = (size_t)T * DM::NU;
  double* xs = D.xs + (size_t)b * nxs;
  double* us = D.us + (size_t)b * nus;
  double* kf = D.kff + (size_t)b * nus;
  for (size_t i = tid; i < nxs; i += nthreads) xs[i] = ((i % DM::NX) == 6) ? 1.0 : 0.0;
  for (size_t i = tid; i < nus; i += nthreads) {
    us[i] = 0.0;
    kf[i] = 0.0;  // warm start of the box QPs (a fresh solver's k_)
  }
  for (int i = tid; i < DM::NX; i += nthreads) D.x0[(size_t)b * DM::NX + i] = D.q_x0[(size_t)job * DM::NX + i];
}

}  // namespace empc
