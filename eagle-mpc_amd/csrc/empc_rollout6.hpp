// empc_rollout6.hpp -- HOT-C kernel body, role-split form (SolverFDDP::forwardPass / SolverSbFDDP::forwardPassDDP for all
// step lengths at once; reference call sites src/sbfddp.cpp:264, 352, 416-460).
//
// Why this shape.  The forward pass is a chain over the knots; at batch 1024 the wave-per-trajectory form (rollout_wave5)
// runs one wavefront per SIMD with 10 of 64 lanes busy and ~4.6k dependent vector instructions per knot: its time is the
// length of that instruction chain, not memory (profiles/r01_rollout_ablation.txt).  Two changes shorten the chain:
//   * packing: one wavefront carries G = 64 / NA trajectories x NA step lengths (6 x 10 = 60 busy lanes), so a batch of
//     1024 needs 171 workgroups instead of 1024 wavefronts, and the SIMDs that frees run
//   * roles: the four wavefronts of a workgroup each execute ONE independent part of a node for the same 60 trials, side
//     by side on the four SIMDs of a CU, and hand their results over through LDS between two workgroup barriers per knot:
//       A  feedback + actuation   dx = x_try (-) xs[t], dv term, u = us - alpha k - K dx, squash, tau = B sigma(u)
//       B  bias + frames          RNEA bias forces h(q, v) with the operational-frame captures, frame costs, contact frame
//       C  inertia + integration  CRBA + Cholesky of M(q) beside A and B; after the barrier a = M^-1 (tau - h), contact KKT,
//                                 semi-implicit Euler step, gap contraction -> x_try of the next knot
//       D  costs + memory         State / Control / friction-cone cost values and the node's cost sum (one knot behind),
//                                 stores of xs_try / us_try / acc_try, staging of the next knot's nominal data
//     The chain of a knot becomes max(A, C) + the join instead of the sum of everything.
// The arithmetic of a trial is the arithmetic of node_nominal / rollout_wave5, operation for operation (same helper
// functions, same summation order of the costs), so the step lengths accepted by select do not depend on the form.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

constexpr int R6_A = 0, R6_B = 1, R6_C = 2, R6_D = 3, R6_WAVES = 4;
constexpr int R6_GMAX = 8;  // most trajectories packed into one wavefront (LDS staging is sized for it)

template <class DM>
struct Roll6Smem {
  static constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NDX = DM::NDX, NL = 64;
  // nominal data of one trajectory at one knot
  static constexpr int NOM_X = 0, NOM_GAP = NOM_X + NX, NOM_VF = NOM_GAP + NDX, NOM_US = NOM_VF + NDX, NOM_KF = NOM_US + NU,
                       NOM_K = NOM_KF + NU;
  static constexpr int NOMSZ = (NOM_K + NU * NDX) | 1;  // odd stride: the G broadcast addresses of a read fall in distinct banks
  // per-lane exchange slots, laid out [item][lane]
  static constexpr int OFF_XT = 0;                            // x_try of the current knot            (C -> A, B, C, D)
  static constexpr int OFF_UT = OFF_XT + NX * NL;             // control of the trial (unsquashed s)  (A -> D)
  static constexpr int OFF_TAU = OFF_UT + NU * NL;            // generalized force                    (A -> C)
  static constexpr int OFF_H = OFF_TAU + NV * NL;             // bias forces                          (B -> C)
  static constexpr int OFF_CAP = OFF_H + NV * NL;             // contact frame capture, 24 doubles    (B -> C)
  static constexpr int OFF_ACC = OFF_CAP + 24 * NL;           // acceleration | contact force         (C -> D)
  static constexpr int OFF_ELLF = OFF_ACC + DM::NACC * NL;    // frame-cost sum, by knot parity       (B -> D)
  static constexpr int OFF_VAL = OFF_ELLF + 2 * NL;           // activation value per cost            (D -> D)
  static constexpr int OFF_FLAG = OFF_VAL + EMPC_MAX_COSTS * NL;  // ok flag of role C
  static constexpr int OFF_TB = OFF_FLAG + NL;                // trajectory index of each packed slot (ints)
  static constexpr int OFF_NOM = OFF_TB + R6_GMAX;            // [2][GMAX][NOMSZ]
  static constexpr int SIZE = (OFF_NOM + 2 * R6_GMAX * NOMSZ + 1) / 2 * 2;
};

// trajectories per wavefront for NA step lengths
EMPC_HD int roll6_group_size(int NA) {
  const int g = 64 / NA;
  return g > R6_GMAX ? R6_GMAX : g;
}

// per-lane view of its trial: which trajectory / step length, and the trajectory-level switches of the forward pass
struct Roll6Lane {
  int b, g, ai, live;
  double alpha, smooth;
  bool plain, need_dv;
};
EMPC_HD Roll6Lane roll6_lane(const DevBuffers& D, const int* TB, int lane, int G) {
  Roll6Lane L;
  const int NA = D.NA;
  L.g = lane / NA;
  L.ai = lane % NA;
  L.b = (L.g < G) ? TB[L.g] : -1;
  L.live = 0;
  L.alpha = ldexp(1.0, -L.ai);
  L.smooth = 0.1;
  L.plain = true;
  L.need_dv = false;
  if (L.b >= 0) {
    const TrajState& st = D.st[L.b];
    const bool ddp = (st.phase == PHASE_DDP), feas = st.is_feasible != 0;
    L.live = (st.phase == PHASE_DONE || st.bwd_failed) ? 0 : 1;
    L.smooth = st.smooth;
    L.plain = ddp || feas || (L.ai == 0);
    L.need_dv = !ddp && !feas;
  }
  return L;
}

// Exec concept: ex.role(w, f) runs f(lane, slot) on wavefront w only, ex.all(f) on every wavefront; ex.sync() is a
// workgroup barrier.
template <class DM, bool CT, class Exec>
EMPC_HD void rollout_group6(Exec& ex, const DevBuffers& D, int group, double* N) {
  typedef Roll6Smem<DM> SM;
  constexpr int NX = DM::NX, NU = DM::NU, NV = DM::NV, NQ = DM::NQ, NDX = DM::NDX, REC = DM::REC, NB = DM::NB, NROT = DM::NROT;
  constexpr int NL = SM::NL;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcModelDesc& m = P.model;
  const int T = D.T, NA = D.NA;
  const int G = roll6_group_size(NA);
  const int nlist = D.act_list ? *D.act_count : D.B;
  if (group * G >= nlist) return;  // whole workgroup: nothing packed here
  int* TB = reinterpret_cast<int*>(N + SM::OFF_TB);
  double* XT = N + SM::OFF_XT;
  double* UT = N + SM::OFF_UT;
  double* TAU = N + SM::OFF_TAU;
  double* HB = N + SM::OFF_H;
  double* CAP = N + SM::OFF_CAP;
  double* ACC = N + SM::OFF_ACC;
  double* ELLF = N + SM::OFF_ELLF;
  double* VAL = N + SM::OFF_VAL;
  double* FLAG = N + SM::OFF_FLAG;
  double* NOM = N + SM::OFF_NOM;
  const double dt = P.dt;

  // ---- packed slots -> trajectories ---------------------------------------------------------------------------------
  ex.role(R6_A, [&](int lane, int sl) {
    if (lane < R6_GMAX) {
      const int i = group * G + lane;
      TB[lane] = (lane < G && i < nlist) ? (D.act_list ? D.act_list[i] : i) : -1;
    }
  });
  ex.sync();

  // nominal data of knot t for every packed trajectory -> staging buffer t & 1 (all lanes of the calling wavefront)
  auto stage_nominal = [&](int t, int lane) {
    double* dst = NOM + (size_t)(t & 1) * R6_GMAX * SM::NOMSZ;
    const int per = (t < T) ? SM::NOMSZ : SM::NOM_US;  // the terminal node has no control / gains
    for (int i = lane; i < G * SM::NOMSZ; i += NL) {
      const int g = i / SM::NOMSZ, j = i % SM::NOMSZ;
      const int b = TB[g];
      if (b < 0 || j >= per || j >= SM::NOM_K + NU * NDX) continue;
      double v;
      if (j < SM::NOM_GAP)
        v = D.xs[((size_t)b * (T + 1) + t) * NX + (j - SM::NOM_X)];
      else if (j < SM::NOM_VF)
        v = D.tape[((size_t)b * (T + 1) + t) * REC + DM::OFF_GAP + (j - SM::NOM_GAP)];
      else if (j < SM::NOM_US)
        v = D.Vf[((size_t)b * (T + 1) + t) * NDX + (j - SM::NOM_VF)];
      else if (j < SM::NOM_KF)
        v = D.us[((size_t)b * T + t) * NU + (j - SM::NOM_US)];
      else if (j < SM::NOM_K)
        v = D.kff[((size_t)b * T + t) * NU + (j - SM::NOM_KF)];
      else
        v = D.K[((size_t)b * T + t) * NU * NDX + (j - SM::NOM_K)];
      dst[(size_t)g * SM::NOMSZ + j] = v;
    }
  };
  ex.role(R6_D, [&](int lane, int sl) { stage_nominal(0, lane); });
  ex.sync();

  // every wavefront keeps the same lane -> (trajectory, step length) view in registers
  Roll6Lane LL[Exec::SLOTS];
  ex.all([&](int lane, int sl) { LL[sl] = roll6_lane(D, TB, lane, G); });

  // ---- per-role state that lives across the knots -----------------------------------------------------------------------
  double dvA[Exec::SLOTS];                    // A: -f^T Vxx (xs (-) xs_try) summed over the knots
  double Lc[Exec::SLOTS][DM::NTRI];           // C: Cholesky factor of M, from phase I to phase II of a knot
  int okC[Exec::SLOTS];                       // C: trial state stayed finite
  double costD[Exec::SLOTS];                  // D: cost of the trial
  int okD[Exec::SLOTS];

  // x_try of knot 0 (C): x0, contracted towards the nominal start by the gap when the pass keeps gaps
  ex.role(R6_C, [&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    okC[sl] = 1;
    if (!L.live) return;
    const double* nom = NOM + (size_t)L.g * SM::NOMSZ;
    double xn[NX], xt[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = D.x0[(size_t)L.b * NX + i];
    if (L.plain) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = xn[i];
    } else {
      double step[NDX];
#pragma unroll
      for (int i = 0; i < NDX; ++i) step[i] = nom[SM::NOM_GAP + i] * (L.alpha - 1.0);
      state_integrate<DM>(xn, step, xt, nullptr);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) XT[i * NL + lane] = xt[i];
  });
  ex.role(R6_A, [&](int lane, int sl) { dvA[sl] = 0.0; });
  ex.role(R6_D, [&](int lane, int sl) {
    costD[sl] = 0.0;
    okD[sl] = 1;
  });
  ex.sync();

  // cost of knot tp for D's lane, from the values D left in VAL, the contact force in ACC and B's frame-cost sum
  auto finish_cost = [&](int tp, int lane, int sl, const Roll6Lane& L) {
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[tp]];
    const bool terminal = (tp == T);
    const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
    double ell = 0;
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const auto& c = set.costs[ci];
      if (!c.active) continue;
      if (c.type == EMPC_COST_STATE || c.type == EMPC_COST_CONTROL) ell += c.weight * VAL[ci * NL + lane];
    }
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const auto& c = set.costs[ci];
      if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
      double lam[3], r[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 3; ++i) lam[i] = ACC[(NV + i) * NL + lane];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        r[i] = use_contact ? (c.ref[4 + 3 * i] * lam[0] + c.ref[5 + 3 * i] * lam[1] + c.ref[6 + 3 * i] * lam[2]) : 0.0;
      ell += c.weight * activation_value<6>(c, r, 5);
    }
    ell += ELLF[(tp & 1) * NL + lane];
    const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 : dt;
    costD[sl] += cscale * ell;
    if (bad_number(costD[sl])) okD[sl] = 0;
    // acceleration | contact force of the knot -> acc_try (linearize reuses them for the accepted trial)
    double* ac_o = D.acc_try + ((size_t)L.b * NA + L.ai) * (T + 1) * DM::NACC + (size_t)tp * DM::NACC;
#pragma unroll
    for (int i = 0; i < DM::NACC; ++i) ac_o[i] = ACC[i * NL + lane];
  };

  for (int t = 0; t <= T; ++t) {
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
    const bool terminal = (t == T);
    const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
    // =============================================== phase I ===========================================================
    // ---- A: state difference to the nominal trajectory, feedback, squashing, generalized force ----------------------------
    ex.role(R6_A, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      const double* nom = NOM + ((size_t)(t & 1) * R6_GMAX + L.g) * SM::NOMSZ;
      double x[NX], dx[NDX], s[NU], u[NU], tau[NV];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
      state_diff<DM>(nom + SM::NOM_X, x, dx, nullptr);
      if (L.need_dv) {
        double dv = dvA[sl];
#pragma unroll
        for (int i = 0; i < NDX; ++i) dv += nom[SM::NOM_VF + i] * dx[i];  // +(Vxx f).(xs_try (-) xs)
        dvA[sl] = dv;
      }
      if (!terminal) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double a_ = nom[SM::NOM_US + i] - nom[SM::NOM_KF + i] * L.alpha;
#pragma unroll
          for (int j = 0; j < NDX; ++j) a_ -= nom[SM::NOM_K + i * NDX + j] * dx[j];
          s[i] = a_;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) s[i] = 0.0;
      }
      if (P.use_squash) {
        double lbv[NU], ubv[NU];
        const int power = P.prm.smoothsat_power;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          lbv[i] = P.u_lb[i];
          ubv[i] = P.u_ub[i];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          double du;
          squash1(s[i], lbv[i], ubv[i], L.smooth, power, u[i], du);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = s[i];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double a_ = 0;
#pragma unroll
        for (int c = 0; c < NROT; ++c) a_ += P.tau_f[r * NROT + c] * u[c];
        tau[r] = a_;
      }
#pragma unroll
      for (int i = 6; i < NV; ++i) tau[i] = u[NROT + i - 6];
#pragma unroll
      for (int i = 0; i < NV; ++i) TAU[i * NL + lane] = tau[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) UT[i * NL + lane] = s[i];
    });
    // ---- B: bias forces with the frame captures; frame costs; contact frame for C ------------------------------------------
    ex.role(R6_B, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      double x[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
      const double* q = x;
      const double* v = x + NQ;
      double R0[9], cs[NB], sn[NB];
      quat_to_R(q + 3, R0);
#pragma unroll
      for (int b = 1; b < NB; ++b) fsincos(q[7 + b - 1], &sn[b - 1], &cs[b - 1]);
      // frames referenced by this node's costs / contacts (same scan as node_nominal)
      int capf[NCAP] = {0, 0};
      int ncap = 0;
      for (int ci = 0; ci < set.ncosts; ++ci) {
        const auto& c = set.costs[ci];
        if (!c.active || c.frame < 0 || c.type == EMPC_COST_CONTACT_FRICTION_CONE) continue;
        bool seen = false;
#pragma unroll
        for (int k = 0; k < NCAP; ++k) seen = seen || (k < ncap && capf[k] == c.frame);
        if (!seen) {
#pragma unroll
          for (int k = 0; k < NCAP; ++k)
            if (k == ncap) capf[k] = c.frame;
          ncap = (ncap < NCAP) ? ncap + 1 : ncap;
        }
      }
      int ccap = 0;
      if constexpr (CT) {
        if (use_contact) {
          const int cframe = set.contacts[0].frame;
          bool seen = false;
#pragma unroll
          for (int k = 0; k < NCAP; ++k)
            if (k < ncap && capf[k] == cframe) {
              seen = true;
              ccap = k;
            }
          if (!seen) {
#pragma unroll
            for (int k = 0; k < NCAP; ++k)
              if (k == ncap) capf[k] = cframe;
            ccap = ncap;
            ncap = (ncap < NCAP) ? ncap + 1 : ncap;
          }
        }
      }
      FrameCap<double> caps[NCAP];
      double zero[NV], h[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) zero[i] = 0.0;
      rnea_chain<NB, double>(m, R0, q, cs, sn, v, zero, true, -1, nullptr, h, ncap, capf, caps);
#pragma unroll
      for (int i = 0; i < NV; ++i) HB[i * NL + lane] = h[i];
      if constexpr (CT) {
        if (use_contact) {
          FrameCap<double> ck = caps[0];
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk == ccap) ck = caps[kk];
#pragma unroll
          for (int i = 0; i < 9; ++i) CAP[i * NL + lane] = ck.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) CAP[(9 + i) * NL + lane] = ck.p[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) CAP[(12 + i) * NL + lane] = ck.v[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) CAP[(18 + i) * NL + lane] = ck.a[i];
        }
      }
      // frame costs (value only), summed in cost order
      double ell_frames = 0;
      for (int ci = 0; ci < set.ncosts; ++ci) {
        const auto& c = set.costs[ci];
        if (!c.active || c.type == EMPC_COST_STATE || c.type == EMPC_COST_CONTROL || c.type == EMPC_COST_CONTACT_FRICTION_CONE)
          continue;
        double cval = 0;
        {
          FrameCap<double> fk = caps[0];
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk < ncap && capf[kk] == c.frame) fk = caps[kk];
          double r[6];
          int nr = 6;
          if (c.type == EMPC_COST_FRAME_PLACEMENT) {
            double rR[9], dp[3], rp[3], qq[4];
            matTmul3<double>(c.ref + 3, fk.R, rR);
#pragma unroll
            for (int i = 0; i < 3; ++i) dp[i] = fk.p[i] - c.ref[i];
            matTvec3<double>(c.ref + 3, dp, rp);
            R_to_quat(rR, qq);
            log6_quat(qq, rp, r);
          } else if (c.type == EMPC_COST_FRAME_ROTATION) {
            double rR[9], qq[4];
            matTmul3<double>(c.ref, fk.R, rR);
            R_to_quat(rR, qq);
            quat_log3(qq, r);
            nr = 3;
          } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
#pragma unroll
            for (int i = 0; i < 3; ++i) r[i] = fk.p[i] - c.ref[i];
            nr = 3;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = fk.v[i] - c.ref[i];
          }
          if (nr == 3) {
            r[3] = r[4] = r[5] = 0.0;
          }
          cval = activation_value<6>(c, r, nr);
        }
        ell_frames += c.weight * cval;
      }
      ELLF[(t & 1) * NL + lane] = ell_frames;
    });
    // ---- C: joint-space inertia and its Cholesky factor ------------------------------------------------------------------------
    ex.role(R6_C, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      double cs[NB], sn[NB];
#pragma unroll
      for (int b = 1; b < NB; ++b) fsincos(XT[(7 + b - 1) * NL + lane], &sn[b - 1], &cs[b - 1]);
      crba_chain<NB>(m, cs, sn, Lc[sl]);
      chol_packed<NV>(Lc[sl]);
    });
    // ---- D: cost of the previous knot, State costs of this one, stores, staging of the next knot's nominal data ----------------
    ex.role(R6_D, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (L.live) {
        if (t > 0) finish_cost(t - 1, lane, sl, L);
        double x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
        double* xs_o = D.xs_try + ((size_t)L.b * NA + L.ai) * (T + 1) * NX + (size_t)t * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) xs_o[i] = x[i];
        double rstate[NDX];
        int rstate_of = -1;
        for (int ci = 0; ci < set.ncosts; ++ci) {
          const auto& c = set.costs[ci];
          if (!c.active || c.type != EMPC_COST_STATE) continue;
          if (!(c.ref_share >= 0 && c.ref_share == rstate_of)) {
            state_diff<DM>(c.ref, x, rstate, nullptr);
            rstate_of = (c.ref_share >= 0) ? c.ref_share : ci;
          }
          VAL[ci * NL + lane] = activation_value<NDX>(c, rstate, NDX);
        }
      }
      if (t < T) stage_nominal(t + 1, lane);
    });
    ex.sync();
    // =============================================== phase II ==========================================================
    // ---- C: acceleration, contact, Euler step, next trial state -------------------------------------------------------------
    ex.role(R6_C, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      double x[NX], a[NV], lam[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = XT[i * NL + lane];
#pragma unroll
      for (int i = 0; i < NV; ++i) a[i] = TAU[i * NL + lane] - HB[i * NL + lane];
      chol_solve_packed<NV>(Lc[sl], a);
      if constexpr (CT) {
        if (use_contact) {
          FrameCap<double> ck;
#pragma unroll
          for (int i = 0; i < 9; ++i) ck.R[i] = CAP[i * NL + lane];
#pragma unroll
          for (int i = 0; i < 3; ++i) ck.p[i] = CAP[(9 + i) * NL + lane];
#pragma unroll
          for (int i = 0; i < 6; ++i) ck.v[i] = CAP[(12 + i) * NL + lane];
#pragma unroll
          for (int i = 0; i < 6; ++i) ck.a[i] = CAP[(18 + i) * NL + lane];
          double R0[9], cs[NB], sn[NB];
          quat_to_R(x + 3, R0);
#pragma unroll
          for (int b = 1; b < NB; ++b) fsincos(x[7 + b - 1], &sn[b - 1], &cs[b - 1]);
          contact_forward<DM>(m, set.contacts[0], ck, R0, x, cs, sn, Lc[sl], a, lam);
        }
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) ACC[i * NL + lane] = a[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) ACC[(NV + i) * NL + lane] = lam[i];
      if (!terminal) {
        const double* v = x + NQ;
        double dxe[NDX], xn[NX], xt[NX];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          dxe[i] = v[i] * dt + a[i] * dt * dt;
          dxe[NV + i] = a[i] * dt;
        }
        state_integrate<DM>(x, dxe, xn, nullptr);
        double mx = 0;
        bool isn = false;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          mx = fmax(mx, fabs(xn[i]));
          isn = isn || (xn[i] != xn[i]);
        }
        if (isn || bad_number(mx)) okC[sl] = 0;
        if (L.plain) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xt[i] = xn[i];
        } else {
          const double* nomn = NOM + ((size_t)((t + 1) & 1) * R6_GMAX + L.g) * SM::NOMSZ;
          double step[NDX];
#pragma unroll
          for (int i = 0; i < NDX; ++i) step[i] = nomn[SM::NOM_GAP + i] * (L.alpha - 1.0);
          state_integrate<DM>(xn, step, xt, nullptr);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) XT[i * NL + lane] = xt[i];
      }
    });
    // ---- D: Control costs of this knot, control of the trial to memory -----------------------------------------------------------
    ex.role(R6_D, [&](int lane, int sl) {
      const Roll6Lane& L = LL[sl];
      if (!L.live) return;
      double s[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i) s[i] = terminal ? 0.0 : UT[i * NL + lane];
      for (int ci = 0; ci < set.ncosts; ++ci) {
        const auto& c = set.costs[ci];
        if (!c.active || c.type != EMPC_COST_CONTROL) continue;
        VAL[ci * NL + lane] = control_cost_value<NU>(c, s, L.smooth, P);
      }
      if (!terminal) {
        double* us_o = D.us_try + ((size_t)L.b * NA + L.ai) * T * NU + (size_t)t * NU;
#pragma unroll
        for (int i = 0; i < NU; ++i) us_o[i] = s[i];
      }
    });
    ex.sync();
  }
  // ---- results of the trials -------------------------------------------------------------------------------------------------------
  ex.role(R6_C, [&](int lane, int sl) { FLAG[lane] = okC[sl] ? 1.0 : 0.0; });
  ex.role(R6_A, [&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    if (L.live) D.try_dv[(size_t)L.b * NA + L.ai] = dvA[sl];
  });
  ex.sync();
  ex.role(R6_D, [&](int lane, int sl) {
    const Roll6Lane& L = LL[sl];
    if (!L.live) return;
    finish_cost(T, lane, sl, L);
    const size_t slot = (size_t)L.b * NA + L.ai;
    D.try_cost[slot] = costD[sl];
    D.try_ok[slot] = (okD[sl] && FLAG[lane] != 0.0) ? 1 : 0;
  });
}

}  // namespace empc
