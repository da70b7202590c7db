// empc_kernels.hpp -- bodies of the HIP kernels of the batched Squash-box FDDP solver.
//
//   linearize_unit2 HOT-A  (empc_linearize2.hpp) one (trajectory, node) unit per half wavefront: IAM calcDiff
//   backward_traj2  HOT-B  (empc_backward2.hpp) one wavefront per trajectory: Riccati sweep, gains, expected improvement
//   rollout_thread  HOT-C  one lane per (trajectory, step length): nonlinear forward pass
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

struct TrajState {
  int phase;  // 0.. : FDDP pass index; PHASE_DDP; PHASE_DONE
  int iter, total_iters, status;
  int is_feasible, was_feasible;
  int need_calc, need_lin;
  int maxiter, bwd_failed, reserved0, last_ok;
  double smooth, smooth_next, convergence, th_stop;
  double xreg, ureg, cost, cost_prev, stop, steplength, dV, dVexp, d0, d1;
  double dg_u, dq_u;      // sum Qu.k , -sum k.Quuk          (control part)
  double dg_f, dq_f;      // -sum Vx.f , +sum f.Vxx f         (gap part, valid when infeasible)
  double gapnorm, qu2;
};

struct DevBuffers {
  const DevProblem* P;
  const EmpcCostSet* sets;
  const int* knot_set;
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
  double* us_last;  // [B][T][NU]  control of the last IAM.calc at every node (fillSquashedOutputs semantics)
  int* n_active;    // [1]
  unsigned long long* dbg;  // [64] cycle stamps of diagnostic builds (EMPC_STAMPS); unused otherwise
  int B, T, NA;
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
};

// =====================================================================================================================
// calc: IAM.calc at (xs[t], us[t]) -> acc (and the last-calc control).  One lane per (b, t).
// =====================================================================================================================
template <class DM, bool CT>
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
template <class DM, bool CT>
EMPC_HD void rollout_thread(const DevBuffers& D, int b, int ai) {
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || st.bwd_failed) return;
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX, REC = DM::REC;
  const int T = D.T, NA = D.NA;
  const bool ddp = (st.phase == PHASE_DDP);
  const bool feas = st.is_feasible != 0;
  const double alpha = ldexp(1.0, -ai);
  const bool plain = ddp || feas || (ai == 0);
  const double smooth = st.smooth;
  double xnext[NX], xtry[NX], dx[NDX], utry[NU], acc[NV], usq[NU], lam[6];
  for (int i = 0; i < NX; ++i) xnext[i] = D.x0[(size_t)b * NX + i];
  double cost_try = 0, dv = 0;
  int ok = 1;
  const size_t slot = (size_t)b * NA + ai;
  double* xs_o = D.xs_try + slot * (T + 1) * NX;
  double* us_o = D.us_try + slot * T * NU;
  double* ac_o = D.acc_try + slot * (T + 1) * DM::NACC;
  for (int t = 0; t <= T; ++t) {
    const double* rec = D.tape + ((size_t)b * (T + 1) + t) * REC;
    if (plain) {
      for (int i = 0; i < NX; ++i) xtry[i] = xnext[i];
    } else {
      double step[NDX];
      for (int i = 0; i < NDX; ++i) step[i] = rec[DM::OFF_GAP + i] * (alpha - 1.0);
      state_integrate<DM>(xnext, step, xtry, nullptr);
    }
    const double* xc = D.xs + ((size_t)b * (T + 1) + t) * NX;
    state_diff<DM>(xc, xtry, dx, nullptr);
    if (!ddp && !feas) {
      const double* vf = D.Vf + ((size_t)b * (T + 1) + t) * NDX;
      for (int i = 0; i < NDX; ++i) dv += vf[i] * dx[i];  // -f^T Vxx (xs (-) xs_try) = +(Vxx f).(xs_try (-) xs)
    }
    double cost;
    if (t < T) {
      const double* uc = D.us + ((size_t)b * T + t) * NU;
      const double* kk = D.kff + ((size_t)b * T + t) * NU;
      const double* KK = D.K + ((size_t)b * T + t) * NU * NDX;
      for (int i = 0; i < NU; ++i) {
        double a_ = uc[i] - kk[i] * alpha;
        for (int j = 0; j < NDX; ++j) a_ -= KK[i * NDX + j] * dx[j];
        utry[i] = a_;
      }
      node_nominal<DM, CT>(EMPC_KREF(DevProblem, D.P), EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]], smooth, xtry, utry, false, xnext, acc, cost, usq, lam);
      for (int i = 0; i < NU; ++i) us_o[(size_t)t * NU + i] = utry[i];
    } else {
      double xn2[NX];
      node_nominal<DM, CT>(EMPC_KREF(DevProblem, D.P), EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]], smooth, xtry, nullptr, true, xn2, acc, cost, usq, lam);
    }
    for (int i = 0; i < NX; ++i) xs_o[(size_t)t * NX + i] = xtry[i];
    for (int i = 0; i < NV; ++i) ac_o[(size_t)t * DM::NACC + i] = acc[i];
    for (int i = 0; i < 6; ++i) ac_o[(size_t)t * DM::NACC + NV + i] = lam[i];
    cost_try += cost;
    if (bad_number(cost_try)) {
      ok = 0;
      break;
    }
    if (t < T) {
      double mx = 0;
      for (int i = 0; i < NX; ++i) mx = fmax(mx, fabs(xnext[i]));
      bool isn = false;
      for (int i = 0; i < NX; ++i) isn = isn || (xnext[i] != xnext[i]);
      if (isn || bad_number(mx)) {
        ok = 0;
        break;
      }
    }
  }
  D.try_cost[slot] = cost_try;
  D.try_dv[slot] = dv;
  D.try_ok[slot] = ok;
}

// =====================================================================================================================
// select: one trajectory; the serial decision logic of solveFDDP / solveDDP / solve after the trial rollouts.
// `tid`/`nthreads` cooperate on the candidate copy; all threads must call it.
// =====================================================================================================================
template <class DM>
EMPC_HD void select_decide(const DevBuffers& D, int b, int& accepted_ai, int& last_ai) {
  TrajState& st = D.st[b];
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcSolverParams& prm = P.prm;
  const int NA = D.NA;
  accepted_ai = -1;
  last_ai = -1;
  if (st.phase == PHASE_DONE) return;
  const bool ddp = (st.phase == PHASE_DDP);
  bool phase_end = false, returned = false;
  if (st.bwd_failed) {
    // computeDirection gave up at reg_max: solveFDDP/solveDDP return false without finishing the iteration
    st.status |= EMPC_STATUS_REG_MAX;
    phase_end = true;
  } else {
    st.need_lin = 0;
    st.need_calc = 0;
    for (int ai = 0; ai < NA; ++ai) {
      const double alpha = ldexp(1.0, -ai);
      st.steplength = alpha;
      last_ai = ai;
      const size_t slot = (size_t)b * NA + ai;
      if (!D.try_ok[slot]) continue;  // "forward_error"
      st.dV = st.cost - D.try_cost[slot];
      if (!ddp) {
        const double dv = st.is_feasible ? 0.0 : D.try_dv[slot];
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
        st.cost = D.try_cost[slot];
        st.need_lin = 1;
        accepted_ai = ai;
        break;
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
      // stoppingCriteria (fork, U1)
      if (prm.stop_criteria == EMPC_STOP_COST_REDUCTION)
        st.stop = fabs(st.cost_prev - st.cost);
      else if (prm.stop_criteria == EMPC_STOP_EXPECTED_REDUCTION)
        st.stop = fabs(st.d0 + 0.5 * st.d1);
      else
        st.stop = st.qu2;
      const bool stop_now = ddp ? (st.was_feasible && st.stop < st.th_stop)
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

// copy helper used by select: candidate <- trial slot `ai` (xs, us, acc); threads cooperate
template <class DM>
EMPC_HD void select_copy(const DevBuffers& D, int b, int accepted_ai, int last_ai, int tid, int nthreads) {
  const int T = D.T, NA = D.NA;
  if (accepted_ai >= 0) {
    const size_t slot = (size_t)b * NA + accepted_ai;
    const double* xs_i = D.xs_try + slot * (T + 1) * DM::NX;
    const double* us_i = D.us_try + slot * T * DM::NU;
    const double* ac_i = D.acc_try + slot * (T + 1) * DM::NACC;
    double* xs_o = D.xs + (size_t)b * (T + 1) * DM::NX;
    double* us_o = D.us + (size_t)b * T * DM::NU;
    double* ac_o = D.acc + (size_t)b * (T + 1) * DM::NACC;
    for (int i = tid; i < (T + 1) * DM::NX; i += nthreads) xs_o[i] = xs_i[i];
    for (int i = tid; i < T * DM::NU; i += nthreads) us_o[i] = us_i[i];
    for (int i = tid; i < (T + 1) * DM::NACC; i += nthreads) ac_o[i] = ac_i[i];
  }
  if (last_ai >= 0) {
    // fillSquashedOutputs reads the data of the LAST calc at every node: the last trial that was rolled out
    const size_t slot = (size_t)b * NA + last_ai;
    const double* us_i = D.us_try + slot * T * DM::NU;
    double* ul = D.us_last + (size_t)b * T * DM::NU;
    for (int i = tid; i < T * DM::NU; i += nthreads) ul[i] = us_i[i];
  }
}

}  // namespace empc
