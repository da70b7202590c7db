// empc_kernels.hpp -- bodies of the HIP kernels of the batched Squash-box FDDP solver.
//
//   linearize_unit  HOT-A  one (trajectory, node) unit per wavefront (or half wavefront): IAM calcDiff
//   backward_traj   HOT-B  one wavefront per trajectory: Riccati sweep, gains, expected-improvement sums
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
// linearize: IAM.calcDiff of one node, LPU lanes cooperating (LPU = 32 or 64, never crossing a wavefront).
//   lanes [0, NV)        tangent direction dq_j (right perturbation)
//   lanes [NV, 2NV)      tangent direction dv_j
//   lanes [2NV, 3NV)     direction da_j  -> column j of the joint-space inertia (RNEA is linear in a);
//                        afterwards reused as the control columns k = lane - 2NV < NU
// smem: per-unit scratch of LIN_SMEM doubles.
// =====================================================================================================================
template <class DM>
struct LinSmem {
  static constexpr int OFF_REC = 0;
  static constexpr int OFF_M = DM::REC;
  static constexpr int OFF_R = OFF_M + DM::NV * DM::NV;    // residual Jacobian staging 6 x (NDX + NU)
  static constexpr int OFF_X = OFF_R + 6 * (DM::NDX + DM::NU);  // nominal x, s, a of the unit
  static constexpr int OFF_F = OFF_X + DM::NX + DM::NU + DM::NV; // nominal frame data: NCAP x (R 9, p 3, v 6)
  static constexpr int SIZE = (OFF_F + NCAP * 18 + 1) / 2 * 2;
};

template <class DM>
struct LinLane {  // per-lane state that survives across syncs (kept small: it lives in registers)
  double dtau[DM::NV];
  double jc[NCAP][6], dvc[NCAP][6];     // LOCAL frame Jacobian column / frame velocity derivative column
  int capf[NCAP];
  int ncap;
  double lx;  // accumulated Lx[j] (x lanes) or Lu[k] (u lanes)
  double wcol[6];  // weighted residual-Jacobian column of the frame cost being processed
  double cost;
};

template <class DM, class Exec>
EMPC_HD void linearize_unit(Exec& ex, const DevBuffers& D, int b, int t, int lpu, double* smem) {
  constexpr int NB = DM::NB, NV = DM::NV, NQ = DM::NQ, NX = DM::NX, NDX = DM::NDX, NU = DM::NU, NROT = DM::NROT;
  constexpr int REC = DM::REC;
  static_assert(NU <= NV, "control columns reuse the NV inertia-column lanes");
  typedef LinSmem<DM> SM;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcModelDesc& m = P.model;
  const TrajState& st = D.st[b];
  const int T = D.T;
  const bool terminal = (t == T);
  const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
  const double dt = P.dt;
  const double smooth = st.smooth;
  double* rec = smem + SM::OFF_REC;
  double* Msh = smem + SM::OFF_M;
  double* Rsh = smem + SM::OFF_R;
  double* Xsh = smem + SM::OFF_X;  // x | s | a
  double* Fsh = smem + SM::OFF_F;
  LinLane<DM> LS[Exec::SLOTS];

  // ---- stage A: zero the record; dual RNEA ------------------------------------------------------------------
  ex.each([&](int lane, int sl) {
    for (int i = lane; i < REC; i += lpu) rec[i] = 0.0;
  });
  ex.sync();
  ex.each([&](int lane, int sl) {
    LinLane<DM>& L = LS[sl];
    const double* xg = D.xs + ((size_t)b * (T + 1) + t) * NX;
    const double* ag = D.acc + ((size_t)b * (T + 1) + t) * DM::NACC;
    double x[NX], a[NV];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xg[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = ag[i];
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) Xsh[i] = x[i];
#pragma unroll
      for (int i = 0; i < NV; ++i) Xsh[NX + NU + i] = a[i];
      const double* ug = D.us + ((size_t)b * T + (terminal ? 0 : t)) * NU;
#pragma unroll
      for (int i = 0; i < NU; ++i) Xsh[NX + i] = terminal ? 0.0 : ug[i];
    }
    // frames referenced by this node's costs
    L.ncap = 0;
#pragma unroll
    for (int k = 0; k < NCAP; ++k) L.capf[k] = 0;
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const EMPC_K EmpcCost& c = set.costs[ci];
      if (!c.active || c.frame < 0 || c.type == EMPC_COST_CONTACT_FRICTION_CONE) continue;
      bool seen = false;
#pragma unroll
      for (int k = 0; k < NCAP; ++k) seen = seen || (k < L.ncap && L.capf[k] == c.frame);
      if (!seen) {
#pragma unroll
        for (int k = 0; k < NCAP; ++k)
          if (k == L.ncap) L.capf[k] = c.frame;
        L.ncap = (L.ncap < NCAP) ? L.ncap + 1 : L.ncap;
      }
    }
    // seeds
    const bool jq = lane < NV, jv = lane >= NV && lane < 2 * NV, ja = lane >= 2 * NV && lane < 3 * NV;
    double R0[9];
    quat_to_R(x + 3, R0);
    D1 R0d[9], p0d[3], csd[NB], snd[NB], vd[NV], ad[NV];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0d[i] = D1(R0[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) p0d[i] = D1(x[i]);
    if (jq && lane < 3) {
#pragma unroll
      for (int i = 0; i < 3; ++i) p0d[i].d = R0[3 * i + lane];
    }
    if (jq && lane >= 3 && lane < 6) {
      double e[3] = {0, 0, 0}, E[9], RE[9];
      e[lane - 3] = 1.0;
      skew3(e, E);
      matmul3<double>(R0, E, RE);
#pragma unroll
      for (int i = 0; i < 9; ++i) R0d[i].d = RE[i];
    }
#pragma unroll
    for (int bb = 1; bb < NB; ++bb) {
      const double th = x[7 + bb - 1];
      const double sn_ = sin(th), cs_ = cos(th);
      const double seed = (jq && lane == 6 + bb - 1) ? 1.0 : 0.0;
      csd[bb - 1] = D1(cs_, -sn_ * seed);
      snd[bb - 1] = D1(sn_, cs_ * seed);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      vd[i] = D1(x[NQ + i], (jv && lane - NV == i) ? 1.0 : 0.0);
      ad[i] = D1(a[i], (ja && lane - 2 * NV == i) ? 1.0 : 0.0);
    }
    D1 taud[NV];
    FrameCap<D1> caps[NCAP];
    rnea_chain<NB, D1>(m, R0d, p0d, csd, snd, vd, ad, true, -1, nullptr, taud, L.ncap, L.capf, caps);
#pragma unroll
    for (int i = 0; i < NV; ++i) L.dtau[i] = taud[i].d;
    if (ja) {
#pragma unroll
      for (int i = 0; i < NV; ++i) Msh[i * NV + (lane - 2 * NV)] = taud[i].d;
    }
#pragma unroll
    for (int c = 0; c < NCAP; ++c) {
      if (c >= L.ncap) continue;
      double Rf[9], dR[9], dp[3], RtdR[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        Rf[i] = caps[c].R[i].v;
        dR[i] = caps[c].R[i].d;
        if (lane == 0) Fsh[c * 18 + i] = Rf[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        dp[i] = caps[c].p[i].d;
        if (lane == 0) Fsh[c * 18 + 9 + i] = caps[c].p[i].v;
      }
      matTvec3<double>(Rf, dp, L.jc[c]);
      matTmul3<double>(Rf, dR, RtdR);
      L.jc[c][3] = RtdR[7];
      L.jc[c][4] = RtdR[2];
      L.jc[c][5] = RtdR[3];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (lane == 0) Fsh[c * 18 + 12 + i] = caps[c].v[i].v;
        L.dvc[c][i] = caps[c].v[i].d;
      }
    }
  });
  ex.sync();

  // ---- stage B: M^-1 solves, Euler Jacobian columns, gaps ----------------------------------------------------
  ex.each([&](int lane, int sl) {
    LinLane<DM>& L = LS[sl];
    const bool xlane = lane < NDX;
    const int k = lane - 2 * NV;
    const bool ulane = k >= 0 && k < NU;
    double Lm[DM::NTRI];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Lm[i * (i + 1) / 2 + j] = Msh[i * NV + j];
    chol_packed<NV>(Lm);
    double da[NV];
    if (xlane) {
#pragma unroll
      for (int i = 0; i < NV; ++i) da[i] = -L.dtau[i];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double Bik = 0.0;
        if (ulane) {
          if (i < 6)
            Bik = (k < NROT) ? P.tau_f[i * NROT + k] : 0.0;
          else
            Bik = (k == NROT + i - 6) ? 1.0 : 0.0;
        }
        da[i] = Bik;
      }
      if (ulane) {
        double usq_k, dus_k = 1.0;
        if (P.use_squash) squash1(Xsh[NX + k], P.u_lb[k], P.u_ub[k], smooth, P.prm.smoothsat_power, usq_k, dus_k);
#pragma unroll
        for (int i = 0; i < NV; ++i) da[i] *= dus_k;
      }
    }
    chol_solve_packed<NV>(Lm, da);
    // Euler step of the nominal state (A.3)
    double x[NX], dxe[NDX], xnext[NX], pe[3];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Xsh[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double ai = Xsh[NX + NU + i];
      dxe[i] = x[NQ + i] * dt + ai * dt * dt;
      dxe[NV + i] = ai * dt;
    }
    state_integrate<DM>(x, dxe, xnext, pe);
    double J2[36];
    Jexp6(dxe, pe, J2);
    // column of [A dt^2 + [0 | I dt]; A dt]
    double G[NDX];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      G[i] = da[i] * dt * dt + ((xlane && lane >= NV && lane - NV == i) ? dt : 0.0);
      G[NV + i] = da[i] * dt;
    }
    double top[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double a_ = 0;
#pragma unroll
      for (int l = 0; l < 6; ++l) a_ += J2[r * 6 + l] * G[l];
      top[r] = a_;
    }
    if (xlane && lane < 6) {
      // + column of J1 = Ad(exp6(xi)^-1) = [[R^T, -R^T [p]x],[0, R^T]]
      double qe[4], pe2[3], Re[9];
      exp6_quat(dxe, qe, pe2);
      quat_to_R(qe, Re);
      if (lane < 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) top[r] += Re[3 * lane + r];
      } else {
        const int c = lane - 3;
        double Px[9], RtP[9];
        skew3(pe2, Px);
        matTmul3<double>(Re, Px, RtP);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          top[r] -= RtP[3 * r + c];
          top[3 + r] += Re[3 * c + r];
        }
      }
    }
    if (xlane) {
#pragma unroll
      for (int r = 0; r < 6; ++r) rec[DM::OFF_FX + r * NDX + lane] = top[r];
#pragma unroll
      for (int r = 6; r < NDX; ++r) rec[DM::OFF_FX + r * NDX + lane] = G[r] + ((r == lane) ? 1.0 : 0.0);
    } else if (ulane) {
#pragma unroll
      for (int r = 0; r < 6; ++r) rec[DM::OFF_FU + r * NU + k] = top[r];
#pragma unroll
      for (int r = 6; r < NDX; ++r) rec[DM::OFF_FU + r * NU + k] = G[r];
    }
    // gaps: fs[t+1] = xnext (-) xs[t+1] is stored in the NEXT node's record; fs[0] = x0 (-) xs[0]
    const bool feas = st.is_feasible != 0;
    if (!terminal) {
      double gap[NDX];
      if (!feas) {
        const double* xn = D.xs + ((size_t)b * (T + 1) + t + 1) * NX;
        state_diff<DM>(xn, xnext, gap, nullptr);
      }
      if (lane < NDX) D.tape[((size_t)b * (T + 1) + t + 1) * REC + DM::OFF_GAP + lane] = feas ? 0.0 : gap[lane];
    }
    if (t == 0) {
      double gap[NDX];
      if (!feas) state_diff<DM>(x, D.x0 + (size_t)b * NX, gap, nullptr);
      if (lane < NDX) D.tape[((size_t)b * (T + 1)) * REC + DM::OFF_GAP + lane] = feas ? 0.0 : gap[lane];
    }
    L.lx = 0.0;
    L.cost = 0.0;
  });
  ex.sync();

  // ---- stage C: costs (Gauss-Newton), CostModelSum order ------------------------------------------------------
  for (int ci = 0; ci < set.ncosts; ++ci) {
    const EMPC_K EmpcCost& c = set.costs[ci];
    if (!c.active) continue;
    const double w = c.weight;
    if (c.type == EMPC_COST_STATE) {
      ex.each([&](int lane, int sl) {
        LinLane<DM>& L = LS[sl];
        double r[NDX], dpl[3], J6[36], Ar[NDX], Arr[NDX];
        state_diff<DM>(c.ref, Xsh, r, dpl);
        Jlog6(r, dpl, J6);
        double cv = 0;
#pragma unroll
        for (int i = 0; i < NDX; ++i) {
          double av;
          activation1(c.activation, r[i], c.act_w[i], c.lb[i], c.ub[i], av, Ar[i], Arr[i]);
          cv += av;
        }
        L.cost += w * cv;
        if (lane < 6) {
          double g = 0;
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) g += J6[rr * 6 + lane] * Ar[rr];
          L.lx += w * g;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double h = 0;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) h += J6[rr * 6 + i] * Arr[rr] * J6[rr * 6 + lane];
            rec[DM::OFF_LXX + i * NDX + lane] += w * h;
          }
        } else if (lane < NDX) {
          L.lx += w * Ar[lane];
          rec[DM::OFF_LXX + lane * NDX + lane] += w * Arr[lane];
        }
      });
    } else if (c.type == EMPC_COST_CONTROL) {
      ex.each([&](int lane, int sl) {
        LinLane<DM>& L = LS[sl];
        const int k = lane - 2 * NV;
        double cv = 0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double av, Ar, Arr;
          activation1(c.activation, Xsh[NX + i] - c.ref[i], act_weight(c, i, smooth, P), c.lb[i], c.ub[i], av, Ar, Arr);
          cv += av;
          if (i == k) {
            L.lx += w * Ar;
            rec[DM::OFF_LUU + k * NU + k] += w * Arr;
          }
        }
        L.cost += w * cv;
      });
    } else if (c.type == EMPC_COST_CONTACT_FRICTION_CONE) {
      // contact problems are handled by the contact build of this kernel (see linearize_contact); no-op here
    } else {
      // frame costs: residual (nominal), residual Jacobian column per lane, exchange through Rsh
      int nr = 6;
      ex.each([&](int lane, int sl) {
        LinLane<DM>& L = LS[sl];
        int cc = 0;
#pragma unroll
        for (int kk = 1; kk < NCAP; ++kk)
          if (kk < L.ncap && L.capf[kk] == c.frame) cc = kk;
        double jcc[6], dvcc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          jcc[i] = L.jc[0][i];
          dvcc[i] = L.dvc[0][i];
        }
#pragma unroll
        for (int kk = 1; kk < NCAP; ++kk)
          if (kk == cc) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              jcc[i] = L.jc[kk][i];
              dvcc[i] = L.dvc[kk][i];
            }
          }
        double r[6], col[6] = {0, 0, 0, 0, 0, 0};
        const bool qlane = lane < NV;
        if (c.type == EMPC_COST_FRAME_PLACEMENT) {
          double rR[9], dp[3], rp[3], qq[4], J6[36];
          matTmul3<double>(c.ref + 3, (Fsh + cc * 18), rR);
#pragma unroll
          for (int i = 0; i < 3; ++i) dp[i] = Fsh[cc * 18 + 9 + i] - c.ref[i];
          matTvec3<double>(c.ref + 3, dp, rp);
          R_to_quat(rR, qq);
          log6_quat(qq, rp, r);
          Jlog6(r, rp, J6);
          if (qlane) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              double a_ = 0;
#pragma unroll
              for (int l = 0; l < 6; ++l) a_ += J6[i * 6 + l] * jcc[l];
              col[i] = a_;
            }
          }
          nr = 6;
        } else if (c.type == EMPC_COST_FRAME_ROTATION) {
          double rR[9], qq[4], J3[9];
          matTmul3<double>(c.ref, (Fsh + cc * 18), rR);
          R_to_quat(rR, qq);
          quat_log3(qq, r);
          SO3Coef kc;
          so3_coef(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], kc);
          Jlog3(r, kc, J3);
          if (qlane) matvec3<double>(J3, jcc + 3, col);
          nr = 3;
        } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
#pragma unroll
          for (int i = 0; i < 3; ++i) r[i] = Fsh[cc * 18 + 9 + i] - c.ref[i];
          if (qlane) matvec3<double>((Fsh + cc * 18), jcc, col);
          nr = 3;
        } else {  // FRAME_VELOCITY (LOCAL)
#pragma unroll
          for (int i = 0; i < 6; ++i) r[i] = Fsh[cc * 18 + 12 + i] - c.ref[i];
          if (lane < NDX) {
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = dvcc[i];
          }
          nr = 6;
        }
        double cv = 0, g = 0;
        for (int i = 0; i < nr; ++i) {
          double av, Ar, Arr;
          activation1(c.activation, r[i], c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
          cv += av;
          g += col[i] * Ar;
          L.wcol[i] = w * Arr * col[i];
          if (lane < NDX) Rsh[i * NDX + lane] = col[i];
        }
        L.cost += w * cv;
        if (lane < NDX) L.lx += w * g;
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane >= NDX) return;
        LinLane<DM>& L = LS[sl];
        // column `lane` of Lxx += Rx^T (w Arr) Rx[:, lane]
        const int ni = (c.type == EMPC_COST_FRAME_VELOCITY) ? NDX : NV;
        for (int i = 0; i < ni; ++i) {
          double h = 0;
          for (int rr = 0; rr < nr; ++rr) h += Rsh[rr * NDX + i] * L.wcol[rr];
          rec[DM::OFF_LXX + i * NDX + lane] += h;
        }
      });
      ex.sync();
    }
  }

  // ---- stage D: scale by dt, finish the record, copy out ---------------------------------------------------------
  const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 : dt;
  ex.each([&](int lane, int sl) {
    LinLane<DM>& L = LS[sl];
    if (lane < NDX) {
      rec[DM::OFF_LX + lane] = L.lx * cscale;
      for (int i = 0; i < NDX; ++i) rec[DM::OFF_LXX + i * NDX + lane] *= cscale;
    }
    const int k = lane - 2 * NV;
    if (k >= 0 && k < NU) {
      rec[DM::OFF_LU + k] = L.lx * cscale;
      for (int i = 0; i < NU; ++i) rec[DM::OFF_LUU + i * NU + k] *= cscale;
      for (int i = 0; i < NDX; ++i) rec[DM::OFF_LXU + i * NU + k] *= cscale;
    }
    if (lane == 0) rec[DM::OFF_COST] = L.cost * cscale;
  });
  ex.sync();
  ex.each([&](int lane, int sl) {
    double* out = D.tape + ((size_t)b * (T + 1) + t) * REC;
    for (int i = lane; i < REC; i += lpu)
      if (i < DM::OFF_GAP || i >= DM::OFF_GAP + NDX) out[i] = rec[i];  // the gap slot is written by node t-1
  });
}

// =====================================================================================================================
// backward: SolverDDP::backwardPass + computeGains + updateExpectedImprovement for one trajectory (one wavefront).
// =====================================================================================================================
template <class DM>
struct BwdSmem {
  static constexpr int n = DM::NDX, m = DM::NU, nm = n + m;
  static constexpr int OFF_REC = 0;
  static constexpr int OFF_V = DM::REC;          // Vxx' n x n
  static constexpr int OFF_VX = OFF_V + n * n;   // Vx' n
  static constexpr int OFF_W = OFF_VX + n;       // W = Vxx' [Fx Fu]  n x nm
  static constexpr int OFF_Q = OFF_W + n * nm;   // Q nm x nm (only xx, xu, uu blocks used)
  static constexpr int OFF_QV = OFF_Q + nm * nm; // [Qx; Qu] nm
  static constexpr int OFF_K = OFF_QV + nm;      // K m x n
  static constexpr int OFF_KF = OFF_K + m * n;   // k m, then Quuk m
  static constexpr int OFF_RED = OFF_KF + 2 * m; // 64 reduction slots
  static constexpr int OFF_FLAG = OFF_RED + 64;  // small flags
  static constexpr int SIZE = OFF_FLAG + 8;
};

template <class DM, class Exec>
EMPC_HD void backward_traj(Exec& ex, const DevBuffers& D, int b, double* smem) {
  typedef BwdSmem<DM> SM;
  constexpr int n = DM::NDX, m = DM::NU, nm = n + m, REC = DM::REC;
  constexpr int NL = 64;
  constexpr int CW = (nm <= 32) ? 32 : 64;   // lanes per column group
  constexpr int NH = NL / CW;                // row halves handled in parallel
  TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const int T = D.T;
  double* rec = smem + SM::OFF_REC;
  double* V = smem + SM::OFF_V;
  double* vx = smem + SM::OFF_VX;
  double* W = smem + SM::OFF_W;
  double* Q = smem + SM::OFF_Q;
  double* qv = smem + SM::OFF_QV;
  double* Ks = smem + SM::OFF_K;
  double* kf = smem + SM::OFF_KF;
  double* red = smem + SM::OFF_RED;
  double* flag = smem + SM::OFF_FLAG;
  const double* tape = D.tape + (size_t)b * (T + 1) * REC;

  // ---- prologue: cost, gap norms, feasibility (SolverDDP::calcDiff tail) --------------------------------------------
  double cost = st.cost, gapnorm = st.gapnorm;
  int is_feasible = st.is_feasible;
  if (st.need_lin) {
    // three reductions over nodes: sum cost, max |gap|, L1 (or Linf) gap norm
    double r_cost[Exec::SLOTS], r_max[Exec::SLOTS], r_l1[Exec::SLOTS];
    ex.each([&](int lane, int sl) {
      double c = 0, mx = 0, l1 = 0;
      for (int t = lane; t <= T; t += NL) {
        const double* r = tape + (size_t)t * REC;
        c += r[DM::OFF_COST];
        for (int i = 0; i < n; ++i) {
          const double g = fabs(r[DM::OFF_GAP + i]);
          mx = fmax(mx, g);
          l1 += g;
        }
      }
      r_cost[sl] = c;
      r_max[sl] = mx;
      r_l1[sl] = l1;
    });
    double tot_c = 0, tot_mx = 0, tot_l1 = 0;
    for (int pass = 0; pass < 3; ++pass) {
      ex.each([&](int lane, int sl) { red[lane] = pass == 0 ? r_cost[sl] : (pass == 1 ? r_max[sl] : r_l1[sl]); });
      ex.sync();
      double acc = 0;
      for (int i = 0; i < NL; ++i) acc = (pass == 1) ? fmax(acc, red[i]) : acc + red[i];
      if (pass == 0) tot_c = acc;
      if (pass == 1) tot_mx = acc;
      if (pass == 2) tot_l1 = acc;
      ex.sync();
    }
    cost = tot_c;
    if (!is_feasible) is_feasible = (tot_mx < D.gaptol) ? 1 : 0;
    gapnorm = (P.prm.gap_norm == EMPC_GAP_L1) ? tot_l1 : tot_mx;
  }
  const bool infeas = !is_feasible;

  double xreg = st.xreg, ureg = st.ureg;
  double dg_u = 0, dq_u = 0, dg_f = 0, dq_f = 0, qu2 = 0;
  bool failed_final = false;
  while (true) {
    bool fail = false;
    dg_u = dq_u = dg_f = dq_f = qu2 = 0;
    // terminal node
    {
      const double* r = tape + (size_t)T * REC;
      ex.each([&](int lane, int sl) {
        for (int i = lane; i < n * n; i += NL) V[i] = r[DM::OFF_LXX + i] + (((i / n) == (i % n)) ? xreg : 0.0);
        if (lane < n) {
          vx[lane] = r[DM::OFF_LX + lane];
          rec[DM::OFF_GAP + lane] = r[DM::OFF_GAP + lane];
        }
      });
      ex.sync();
      double vf_l[Exec::SLOTS];
      ex.each([&](int lane, int sl) {
        double a_ = 0;
        if (lane < n && infeas)
          for (int j = 0; j < n; ++j) a_ += V[lane * n + j] * rec[DM::OFF_GAP + j];
        vf_l[sl] = a_;
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane < n) {
          if (infeas) vx[lane] += vf_l[sl];
          D.Vf[((size_t)b * (T + 1) + T) * n + lane] = vf_l[sl];
          D.Vx[((size_t)b * (T + 1) + T) * n + lane] = vx[lane];
          red[lane] = infeas ? vx[lane] * rec[DM::OFF_GAP + lane] : 0.0;
          red[32 + lane] = infeas ? rec[DM::OFF_GAP + lane] * vf_l[sl] : 0.0;
        }
      });
      ex.sync();
      for (int i = 0; i < n; ++i) {
        dg_f -= red[i];
        dq_f += red[32 + i];
      }
      ex.sync();
    }
    for (int t = T - 1; t >= 0 && !fail; --t) {
      const double* r = tape + (size_t)t * REC;
      ex.each([&](int lane, int sl) {
        for (int i = lane; i < REC; i += NL) rec[i] = r[i];
      });
      ex.sync();
      // W = V' A, A = [Fx Fu]
      ex.each([&](int lane, int sl) {
        const int c = lane % CW, hh = lane / CW;
        if (c >= nm) return;
        double Acol[n];
#pragma unroll
        for (int k2 = 0; k2 < n; ++k2) Acol[k2] = (c < n) ? rec[DM::OFF_FX + k2 * n + c] : rec[DM::OFF_FU + k2 * m + (c - n)];
        const int r0 = hh * ((n + NH - 1) / NH), r1 = (r0 + (n + NH - 1) / NH < n) ? r0 + (n + NH - 1) / NH : n;
        for (int i = r0; i < r1; ++i) {
          double a_ = 0;
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) a_ += V[i * n + k2] * Acol[k2];
          W[i * nm + c] = a_;
        }
      });
      ex.sync();
      // Q = H + A^T W  (blocks xx, xu, uu), qv = [Lx; Lu] + A^T vx'
      ex.each([&](int lane, int sl) {
        const int c = lane % CW, hh = lane / CW;
        if (c >= nm) return;
        double Wcol[n];
#pragma unroll
        for (int k2 = 0; k2 < n; ++k2) Wcol[k2] = W[k2 * nm + c];
        const int rows = nm;
        const int r0 = hh * ((rows + NH - 1) / NH), r1 = (r0 + (rows + NH - 1) / NH < rows) ? r0 + (rows + NH - 1) / NH : rows;
        for (int rr = r0; rr < r1; ++rr) {
          if (rr >= n && c < n) continue;  // Qux = Qxu^T is not needed
          double a_;
          if (rr < n && c < n)
            a_ = rec[DM::OFF_LXX + rr * n + c];
          else if (rr < n)
            a_ = rec[DM::OFF_LXU + rr * m + (c - n)];
          else
            a_ = rec[DM::OFF_LUU + (rr - n) * m + (c - n)];
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) {
            const double Akr = (rr < n) ? rec[DM::OFF_FX + k2 * n + rr] : rec[DM::OFF_FU + k2 * m + (rr - n)];
            a_ += Akr * Wcol[k2];
          }
          Q[rr * nm + c] = a_;
        }
        if (hh == 0) {
          double a_ = (c < n) ? rec[DM::OFF_LX + c] : rec[DM::OFF_LU + (c - n)];
#pragma unroll
          for (int k2 = 0; k2 < n; ++k2) {
            const double Akc = (c < n) ? rec[DM::OFF_FX + k2 * n + c] : rec[DM::OFF_FU + k2 * m + (c - n)];
            a_ += Akc * vx[k2];
          }
          qv[c] = a_;
        }
      });
      ex.sync();
      // computeGains: LLT(Quu + ureg I); K = Quu^-1 Qxu^T ; k = Quu^-1 Qu
      ex.each([&](int lane, int sl) {
        double Lq[m * (m + 1) / 2];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) Lq[i * (i + 1) / 2 + j] = Q[(n + i) * nm + n + j] + ((i == j) ? ureg : 0.0);
        // Quu is symmetric up to rounding; the reference factorises the lower triangle of the full matrix
        const bool pd = chol_packed<m>(Lq);
        if (lane == 0) flag[0] = pd ? 0.0 : 1.0;
        if (lane <= n) {
          double rhs[m];
#pragma unroll
          for (int i = 0; i < m; ++i) rhs[i] = (lane < n) ? Q[lane * nm + n + i] : qv[n + i];
          chol_solve_packed<m>(Lq, rhs);
          if (lane < n) {
#pragma unroll
            for (int i = 0; i < m; ++i) Ks[i * n + lane] = rhs[i];
          } else {
#pragma unroll
            for (int i = 0; i < m; ++i) kf[i] = rhs[i];
            // Quuk = (Quu + ureg I) k
#pragma unroll
            for (int i = 0; i < m; ++i) {
              double a_ = 0;
#pragma unroll
              for (int j = 0; j < m; ++j) a_ += (Q[(n + (i > j ? i : j)) * nm + n + (i > j ? j : i)] ) * rhs[j];
              kf[m + i] = a_ + ureg * rhs[i];
            }
          }
        }
      });
      ex.sync();
      if (flag[0] != 0.0) {
        fail = true;
        break;
      }
      // expected-improvement sums and gain write-out
      for (int i = 0; i < m; ++i) {
        dg_u += qv[n + i] * kf[i];
        dq_u -= kf[i] * kf[m + i];
        qu2 += qv[n + i] * qv[n + i];
      }
      ex.each([&](int lane, int sl) {
        double* Kg = D.K + ((size_t)b * T + t) * m * n;
        for (int i = lane; i < m * n; i += NL) Kg[i] = Ks[i];
        if (lane < m) D.kff[((size_t)b * T + t) * m + lane] = kf[lane];
      });
      // Vxx = Qxx - Qxu K (into W, n x n), Vx = Qx + K^T Quuk - 2 K^T Qu
      ex.each([&](int lane, int sl) {
        const int c = lane % CW, hh = lane / CW;
        if (c < n) {
          double Kcol[m];
#pragma unroll
          for (int l = 0; l < m; ++l) Kcol[l] = Ks[l * n + c];
          const int r0 = hh * ((n + NH - 1) / NH), r1 = (r0 + (n + NH - 1) / NH < n) ? r0 + (n + NH - 1) / NH : n;
          for (int i = r0; i < r1; ++i) {
            double a_ = Q[i * nm + c];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ -= Q[i * nm + n + l] * Kcol[l];
            W[i * nm + c] = a_;
          }
          if (hh == 0) {
            double a_ = qv[c];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ += Kcol[l] * kf[m + l];
#pragma unroll
            for (int l = 0; l < m; ++l) a_ -= 2.0 * Kcol[l] * qv[n + l];
            red[c] = a_;  // new Vx (before the gap term)
          }
        }
      });
      ex.sync();
      // symmetrise, regularise -> V ; then gap contribution
      ex.each([&](int lane, int sl) {
        for (int i = lane; i < n * n; i += NL) {
          const int rr = i / n, cc = i % n;
          V[i] = 0.5 * (W[rr * nm + cc] + W[cc * nm + rr]) + ((rr == cc) ? xreg : 0.0);
        }
      });
      ex.sync();
      double vf_l[Exec::SLOTS];
      ex.each([&](int lane, int sl) {
        double a_ = 0;
        if (lane < n && infeas)
          for (int j = 0; j < n; ++j) a_ += V[lane * n + j] * rec[DM::OFF_GAP + j];
        vf_l[sl] = a_;
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane < n) {
          const double nv = red[lane] + (infeas ? vf_l[sl] : 0.0);
          vx[lane] = nv;
          D.Vf[((size_t)b * (T + 1) + t) * n + lane] = vf_l[sl];
          D.Vx[((size_t)b * (T + 1) + t) * n + lane] = nv;
          red[lane] = infeas ? nv * rec[DM::OFF_GAP + lane] : 0.0;
          red[32 + lane] = infeas ? rec[DM::OFF_GAP + lane] * vf_l[sl] : 0.0;
        }
      });
      ex.sync();
      // NaN / overflow guard + gap sums
      {
        double mxv = 0;
        bool nanv = false;
        for (int i = 0; i < n; ++i) {
          mxv = fmax(mxv, fabs(vx[i]));
          nanv = nanv || (vx[i] != vx[i]);
          dg_f -= red[i];
          dq_f += red[32 + i];
        }
        ex.sync();
        double r_mx[Exec::SLOTS];
        ex.each([&](int lane, int sl) {
          double mm = 0;
          for (int i = lane; i < n * n; i += NL) mm = (V[i] != V[i]) ? 1e300 : fmax(mm, fabs(V[i]));
          r_mx[sl] = mm;
        });
        ex.each([&](int lane, int sl) { red[lane] = r_mx[sl]; });
        ex.sync();
        double mxV = 0;
        for (int i = 0; i < NL; ++i) mxV = fmax(mxV, red[i]);
        ex.sync();
        if (nanv || bad_number(mxv) || bad_number(mxV)) fail = true;
      }
    }
    if (!fail) break;
    // "backward_error": increaseRegularization and retry (src/sbfddp.cpp:242-255)
    xreg *= P.prm.reg_incfactor;
    if (xreg > P.prm.reg_max) xreg = P.prm.reg_max;
    ureg = xreg;
    if (xreg == P.prm.reg_max) {
      failed_final = true;
      break;
    }
  }
  ex.each([&](int lane, int sl) {
    if (lane == 0) {
      st.cost = cost;
      st.gapnorm = gapnorm;
      st.is_feasible = is_feasible;
      st.xreg = xreg;
      st.ureg = ureg;
      st.dg_u = dg_u;
      st.dq_u = dq_u;
      st.dg_f = dg_f;
      st.dq_f = dq_f;
      st.qu2 = qu2;
      st.bwd_failed = failed_final ? 1 : 0;
    }
  });
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
