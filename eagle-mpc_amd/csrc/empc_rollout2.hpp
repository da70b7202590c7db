// empc_rollout2.hpp -- HOT-C, cooperative form: SolverFDDP::forwardPass(alpha) / SolverSbFDDP::forwardPassDDP(alpha)
// (reference call sites src/sbfddp.cpp:264,340, :416-460) with LPR = 16 lanes per (trajectory, step length) unit.
//
// The forward pass is a serial chain over the knots, so the only way to shorten it is to shorten the work of one knot.
// Lanes that run the SAME instruction stream on different data are the only real parallelism of a wavefront, hence the
// stage plan below gives every stage one uniform body:
//   A  lane 0: dx = x_try (-) xs[t]; lanes 1..: one State-cost residual class each (same state_diff code, other origin)
//      joint sin/cos one joint per lane
//   B  lane i < NU: feedback row i  u_i = us_i - alpha k_i - K_i dx, squash, control-cost terms of component i
//   C  lane j < NV: column j of the joint-space inertia = RNEA(q, 0, e_j) without gravity (RNEA-as-CRBA);
//      lane NV: bias forces h = RNEA(q, v, 0) with gravity and the operational-frame captures -- one rnea_chain body
//   D  lane 0: Cholesky solve of M a = tau - h (+ contact KKT), Euler step, next x_try; lanes 1..: one frame cost each
// Exchange goes through LDS (Roll2Smem, ~3 KB per unit); a unit never spans a wavefront so every sync is a wave fence.
// Numerics follow rollout_thread / node_nominal (empc_kernels.hpp / empc_dev_model.hpp): same operation order inside
// every residual, activation and cost sum, costs added in cost-table order, knots in time order.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

template <class DM>
struct Roll2Smem {
  static constexpr int NB = DM::NB, NV = DM::NV, NX = DM::NX, NU = DM::NU, NDX = DM::NDX, NJ = DM::NJ;
  static constexpr int OFF_X = 0;                      // x_try of the current knot
  static constexpr int OFF_XN = OFF_X + NX;            // xnext of the previous knot
  static constexpr int OFF_DX = OFF_XN + NX;
  static constexpr int OFF_S = OFF_DX + NDX;           // control before / after the squashing
  static constexpr int OFF_USQ = OFF_S + NU;
  static constexpr int OFF_R0 = OFF_USQ + NU;
  static constexpr int OFF_CS = OFF_R0 + 9;
  static constexpr int OFF_SN = OFF_CS + (NJ > 0 ? NJ : 1);
  static constexpr int OFF_M = OFF_SN + (NJ > 0 ? NJ : 1);  // NV x NV, column j written by lane j
  static constexpr int OFF_H = OFF_M + NV * NV;
  static constexpr int CAP = 24;                       // R 9 | p 3 | v 6 | a 6
  static constexpr int OFF_CAP = OFF_H + NV;
  static constexpr int OFF_JC = OFF_CAP + NCAP * CAP;  // contact Jacobian 6 x NV (rows = LOCAL spatial components)
  static constexpr int OFF_SLOT = OFF_JC + 6 * NV;     // weight * activation of every cost, double-buffered by knot parity
  static constexpr int OFF_PART = OFF_SLOT + 2 * EMPC_MAX_COSTS;  // control-cost terms [cost][component]
  static constexpr int OFF_FLAG = OFF_PART + EMPC_MAX_COSTS * NU;
  static constexpr int SIZE = (OFF_FLAG + 2 + 1) / 2 * 2;
};

template <class DM, bool CT, class Exec>
EMPC_HD void rollout_unit2(Exec& ex, const DevBuffers& D, int b, int ai, int lpr, double* N) {
  typedef Roll2Smem<DM> SM;
  constexpr int NB = DM::NB, NX = DM::NX, NU = DM::NU, NV = DM::NV, NQ = DM::NQ, NDX = DM::NDX, REC = DM::REC, NROT = DM::NROT,
                NJ = DM::NJ;
  const TrajState& st = D.st[b];
  if (st.phase == PHASE_DONE || st.bwd_failed) return;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const EMPC_K EmpcModelDesc& m = P.model;
  const int T = D.T, NA = D.NA;
  const bool ddp = (st.phase == PHASE_DDP);
  const bool feas = st.is_feasible != 0;
  const double alpha = ldexp(1.0, -ai);
  const bool plain = ddp || feas || (ai == 0);
  const double smooth = st.smooth;
  const double dt = P.dt;
  const size_t slot = (size_t)b * NA + ai;
  double* xs_o = D.xs_try + slot * (T + 1) * NX;
  double* us_o = D.us_try + slot * T * NU;
  double* ac_o = D.acc_try + slot * (T + 1) * DM::NACC;

  double cost_l[Exec::SLOTS], dv_l[Exec::SLOTS];  // lane 0 only
  ex.each([&](int lane, int sl) {
    cost_l[sl] = 0.0;
    dv_l[sl] = 0.0;
    for (int i = lane; i < NX; i += lpr) N[SM::OFF_XN + i] = D.x0[(size_t)b * NX + i];
    if (lane == 0) {
      N[SM::OFF_FLAG] = 0.0;      // 1 = failed
      N[SM::OFF_FLAG + 1] = 0.0;  // 1 = knot costs pending in the slot buffer
    }
  });
  ex.sync();

  // x_try[t] from xnext (SolverFDDP::forwardPass: xnext (+) (alpha - 1) fs[t]); run by lane 0
  auto advance = [&](int t) {
    const double* rec = D.tape + ((size_t)b * (T + 1) + t) * REC;
    double xn[NX], xt[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = N[SM::OFF_XN + i];
    if (plain) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xt[i] = xn[i];
    } else {
      double step[NDX];
#pragma unroll
      for (int i = 0; i < NDX; ++i) step[i] = rec[DM::OFF_GAP + i] * (alpha - 1.0);
      state_integrate<DM>(xn, step, xt, nullptr);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      N[SM::OFF_X + i] = xt[i];
      xs_o[(size_t)t * NX + i] = xt[i];
    }
  };
  // add the finished knot's costs (cost-table order) to the running total; run by lane 0
  auto close_knot = [&](int t, int sl) {
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
    const double* S = N + SM::OFF_SLOT + (t & 1) * EMPC_MAX_COSTS;
    double ell = 0;
    for (int ci = 0; ci < set.ncosts; ++ci)
      if (set.costs[ci].active) ell += S[ci];
    const double cscale = (t == T && !P.prm.terminal_dt_scaling) ? 1.0 : dt;
    cost_l[sl] += cscale * ell;
    if (bad_number(cost_l[sl])) N[SM::OFF_FLAG] = 1.0;
  };

  ex.each([&](int lane, int sl) {
    if (lane == 0) advance(0);
  });
  ex.sync();

  for (int t = 0; t <= T; ++t) {
    const bool terminal = (t == T);
    const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
    double* S = N + SM::OFF_SLOT + (t & 1) * EMPC_MAX_COSTS;
    if (N[SM::OFF_FLAG] != 0.0) break;  // uniform over the unit: written before the last sync

    // frames referenced by this node's costs / contact (uniform)
    int capf[NCAP] = {0, 0};
    int ncap = 0;
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const EMPC_K EmpcCost& c = set.costs[ci];
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
    const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
    int ccap = 0;
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

    // ---- A: state differences (feedback origin on lane 0, one State-cost reference class per further lane), trig ------
    ex.each([&](int lane, int sl) {
      if (lane == 0 && t > 0) close_knot(t - 1, sl);
      // which origin does this lane difference against?  lane 0: xs[t]; lane 1 + k: leader of the k-th active class of
      // State costs (costs sharing one reference share one residual; lpr - 1 >= EMPC_MAX_COSTS, so one class per lane)
      int leader = -1;
      if (lane > 0) {
        int k = 0;
        for (int ci = 0; ci < set.ncosts; ++ci) {
          const EMPC_K EmpcCost& c = set.costs[ci];
          if (!c.active || c.type != EMPC_COST_STATE) continue;
          const int cls = (c.ref_share >= 0) ? c.ref_share : ci;
          bool first = true;  // first active member of its class?
          for (int cj = 0; cj < ci; ++cj) {
            const EMPC_K EmpcCost& d = set.costs[cj];
            if (d.active && d.type == EMPC_COST_STATE && ((d.ref_share >= 0) ? d.ref_share : cj) == cls) first = false;
          }
          if (!first) continue;
          if (1 + k == lane) leader = ci;
          ++k;
        }
      }
      if (lane == 0 || leader >= 0) {
        double x0[NX], x1[NX], r[NDX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x1[i] = N[SM::OFF_X + i];
        if (lane == 0) {
          const double* xc = D.xs + ((size_t)b * (T + 1) + t) * NX;
#pragma unroll
          for (int i = 0; i < NX; ++i) x0[i] = xc[i];
        } else {
#pragma unroll
          for (int i = 0; i < NX; ++i) x0[i] = set.costs[leader].ref[i];
        }
        state_diff<DM>(x0, x1, r, nullptr);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < NDX; ++i) N[SM::OFF_DX + i] = r[i];
          if (!ddp && !feas) {
            const double* vf = D.Vf + ((size_t)b * (T + 1) + t) * NDX;
            double dv = dv_l[sl];
#pragma unroll
            for (int i = 0; i < NDX; ++i) dv += vf[i] * r[i];  // -f^T Vxx (xs (-) xs_try) = +(Vxx f).(xs_try (-) xs)
            dv_l[sl] = dv;
          }
        } else {
          const int cls = (set.costs[leader].ref_share >= 0) ? set.costs[leader].ref_share : leader;
          for (int ci = leader; ci < set.ncosts; ++ci) {
            const EMPC_K EmpcCost& c = set.costs[ci];
            if (!c.active || c.type != EMPC_COST_STATE || ((c.ref_share >= 0) ? c.ref_share : ci) != cls) continue;
            double cval = 0;
#pragma unroll
            for (int i = 0; i < NDX; ++i) {
              double av, Ar, Arr;
              activation1(c.activation, r[i], c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
              cval += av;
            }
            S[ci] = c.weight * cval;
          }
        }
      }
      // joint sin / cos: joint j on lane j; base rotation on the lane after them
      if (lane < NJ) {
        double s_, c_;
        fsincos(N[SM::OFF_X + 7 + lane], &s_, &c_);
        N[SM::OFF_SN + lane] = s_;
        N[SM::OFF_CS + lane] = c_;
      }
      if (lane == NJ) {
        double q[4] = {N[SM::OFF_X + 3], N[SM::OFF_X + 4], N[SM::OFF_X + 5], N[SM::OFF_X + 6]}, R0[9];
        quat_to_R(q, R0);
#pragma unroll
        for (int i = 0; i < 9; ++i) N[SM::OFF_R0 + i] = R0[i];
      }
    });
    ex.sync();

    // ---- B: feedback rows, squashing, control-cost terms ------------------------------------------------------------
    ex.each([&](int lane, int sl) {
      if (lane >= NU) return;
      double s = 0.0;
      if (!terminal) {
        const double* uc = D.us + ((size_t)b * T + t) * NU;
        const double* kk = D.kff + ((size_t)b * T + t) * NU;
        const double* KK = D.K + ((size_t)b * T + t) * NU * NDX;
        double a_ = uc[lane] - kk[lane] * alpha;
#pragma unroll
        for (int j = 0; j < NDX; ++j) a_ -= KK[lane * NDX + j] * N[SM::OFF_DX + j];
        s = a_;
        us_o[(size_t)t * NU + lane] = s;
      }
      double u = s, du;
      if (P.use_squash) squash1(s, P.u_lb[lane], P.u_ub[lane], smooth, P.prm.smoothsat_power, u, du);
      N[SM::OFF_S + lane] = s;
      N[SM::OFF_USQ + lane] = u;
      for (int ci = 0; ci < set.ncosts; ++ci) {
        const EMPC_K EmpcCost& c = set.costs[ci];
        if (!c.active || c.type != EMPC_COST_CONTROL) continue;
        double av, Ar, Arr;
        activation1(c.activation, s - c.ref[lane], act_weight(c, lane, smooth, P), c.lb[lane], c.ub[lane], av, Ar, Arr);
        N[SM::OFF_PART + ci * NU + lane] = av;
      }
    });
    ex.sync();

    // ---- C: inertia columns and bias forces, one rnea_chain body ----------------------------------------------------
    ex.each([&](int lane, int sl) {
      if (lane > NV) return;
      const bool bias = (lane == NV);
      double R0[9], cs[NB], sn[NB], q[NQ], v[NV], a[NV], tau[NV];
#pragma unroll
      for (int i = 0; i < 9; ++i) R0[i] = N[SM::OFF_R0 + i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        cs[j] = N[SM::OFF_CS + j];
        sn[j] = N[SM::OFF_SN + j];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] = N[SM::OFF_X + i];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        v[i] = bias ? N[SM::OFF_X + NQ + i] : 0.0;
        a[i] = (!bias && i == lane) ? 1.0 : 0.0;
      }
      FrameCap<double> caps[NCAP];
      rnea_chain<NB, double>(m, R0, q, cs, sn, v, a, bias, -1, nullptr, tau, ncap, capf, caps);
      if (bias) {
#pragma unroll
        for (int i = 0; i < NV; ++i) N[SM::OFF_H + i] = tau[i];
#pragma unroll
        for (int c = 0; c < NCAP; ++c) {
          if (c >= ncap) continue;
          double* F = N + SM::OFF_CAP + c * SM::CAP;
#pragma unroll
          for (int i = 0; i < 9; ++i) F[i] = caps[c].R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) F[9 + i] = caps[c].p[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            F[12 + i] = caps[c].v[i];
            F[18 + i] = caps[c].a[i];
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) N[SM::OFF_M + i * NV + lane] = tau[i];
        if constexpr (CT) {
          if (use_contact) {
            // spatial acceleration of the contact frame for qdd = e_j at zero velocity = column j of its LOCAL Jacobian
#pragma unroll
            for (int c = 0; c < NCAP; ++c)
              if (c == ccap) {
#pragma unroll
                for (int r = 0; r < 6; ++r) N[SM::OFF_JC + r * NV + lane] = caps[c].a[r];
              }
          }
        }
      }
    });
    ex.sync();

    // ---- D: lane 0 solves the dynamics and steps; the other lanes evaluate one frame cost each --------------------------
    ex.each([&](int lane, int sl) {
      if (lane == 0) {
        // control costs: component terms summed in component order
        for (int ci = 0; ci < set.ncosts; ++ci) {
          const EMPC_K EmpcCost& c = set.costs[ci];
          if (!c.active || c.type != EMPC_COST_CONTROL) continue;
          double cval = 0;
#pragma unroll
          for (int i = 0; i < NU; ++i) cval += N[SM::OFF_PART + ci * NU + i];
          S[ci] = c.weight * cval;
        }
        double a[NV], L[DM::NTRI];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double a_ = 0;
#pragma unroll
          for (int c = 0; c < NROT; ++c) a_ += P.tau_f[r * NROT + c] * N[SM::OFF_USQ + c];
          a[r] = a_ - N[SM::OFF_H + r];
        }
#pragma unroll
        for (int i = 6; i < NV; ++i) a[i] = N[SM::OFF_USQ + NROT + i - 6] - N[SM::OFF_H + i];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = N[SM::OFF_M + i * NV + j];
        chol_packed<NV>(L);
        chol_solve_packed<NV>(L, a);
        double lam[6] = {0, 0, 0, 0, 0, 0};
        if constexpr (CT) {
          if (use_contact) {
            // ContactModel3D: [M Jc^T; Jc 0][a; -lam] = [tau - h; -a0]
            const EMPC_K EmpcContact& ct = set.contacts[0];
            const int nc = (ct.type == EMPC_CONTACT_3D) ? 3 : 6;
            const double* F = N + SM::OFF_CAP + ccap * SM::CAP;
            // drift: frame acceleration at qdd = 0 without gravity = captured (with gravity) minus R_f^T (-g)
            double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]}, gf[3], a0[6];
            matTvec3<double>(F, ng, gf);
            if (nc == 3) {
              double wxv[3];
              cross3<double>(F + 15, F + 12, wxv);
              for (int r = 0; r < 3; ++r) a0[r] = (F[18 + r] - gf[r]) + wxv[r];
            } else {
              for (int r = 0; r < 3; ++r) a0[r] = F[18 + r] - gf[r];
              for (int r = 3; r < 6; ++r) a0[r] = F[18 + r];
            }
            if (ct.gains[0] != 0.0 && nc == 3) {
              double dp[3], dpl[3];
              for (int r = 0; r < 3; ++r) dp[r] = F[9 + r] - ct.ref_p[r];
              matTvec3<double>(F, dp, dpl);
              for (int r = 0; r < 3; ++r) a0[r] += ct.gains[0] * dpl[r];
            }
            if (ct.gains[1] != 0.0)
              for (int r = 0; r < nc; ++r) a0[r] += ct.gains[1] * F[12 + r];
            double MiJt[6][NV], G[21];
            for (int r = 0; r < nc; ++r) {
#pragma unroll
              for (int i = 0; i < NV; ++i) MiJt[r][i] = N[SM::OFF_JC + r * NV + i];
              chol_solve_packed<NV>(L, MiJt[r]);
            }
            for (int r = 0; r < nc; ++r)
              for (int c = 0; c <= r; ++c) {
                double g = 0;
#pragma unroll
                for (int i = 0; i < NV; ++i) g += N[SM::OFF_JC + r * NV + i] * MiJt[c][i];
                G[r * (r + 1) / 2 + c] = g;
              }
            if (nc == 3)
              chol_packed<3>(G);
            else
              chol_packed<6>(G);
            for (int r = 0; r < nc; ++r) {
              double g = a0[r];
#pragma unroll
              for (int i = 0; i < NV; ++i) g += N[SM::OFF_JC + r * NV + i] * a[i];
              lam[r] = -g;
            }
            if (nc == 3)
              chol_solve_packed<3>(G, lam);
            else
              chol_solve_packed<6>(G, lam);
            for (int r = 0; r < nc; ++r)
#pragma unroll
              for (int i = 0; i < NV; ++i) a[i] += MiJt[r][i] * lam[r];
          }
          // friction-cone costs need the contact force
          for (int ci = 0; ci < set.ncosts; ++ci) {
            const EMPC_K EmpcCost& c = set.costs[ci];
            if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
            double AR[5][3];
            double nsf[3] = {c.ref[0], c.ref[1], c.ref[2]};
            cone_rows(nsf, c.ref[3], AR);
            double cval = 0;
            for (int i = 0; i < 5; ++i) {
              double r = use_contact ? (AR[i][0] * lam[0] + AR[i][1] * lam[1] + AR[i][2] * lam[2]) : 0.0;
              double av, Ar, Arr;
              activation1(c.activation, r, c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
              cval += av;
            }
            S[ci] = c.weight * cval;
          }
        } else {
          for (int ci = 0; ci < set.ncosts; ++ci) {
            const EMPC_K EmpcCost& c = set.costs[ci];
            if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
            double cval = 0;
            for (int i = 0; i < 5; ++i) {
              double av, Ar, Arr;
              activation1(c.activation, 0.0, c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
              cval += av;
            }
            S[ci] = c.weight * cval;
          }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) ac_o[(size_t)t * DM::NACC + i] = a[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) ac_o[(size_t)t * DM::NACC + NV + i] = lam[i];
        if (!terminal) {
          // Euler step and the next trial state
          double x[NX], xn[NX], dxe[NDX];
#pragma unroll
          for (int i = 0; i < NX; ++i) x[i] = N[SM::OFF_X + i];
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            dxe[i] = x[NQ + i] * dt + a[i] * dt * dt;
            dxe[NV + i] = a[i] * dt;
          }
          state_integrate<DM>(x, dxe, xn, nullptr);
          double mx = 0;
          bool isn = false;
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            mx = fmax(mx, fabs(xn[i]));
            isn = isn || (xn[i] != xn[i]);
            N[SM::OFF_XN + i] = xn[i];
          }
          if (isn || bad_number(mx)) {
            N[SM::OFF_FLAG] = 1.0;
          } else {
            advance(t + 1);
          }
        }
      } else {
        // frame costs: the k-th active one on lane 1 + k % (lpr - 1)
        int k = 0;
        for (int ci = 0; ci < set.ncosts; ++ci) {
          const EMPC_K EmpcCost& c = set.costs[ci];
          if (!c.active || c.type == EMPC_COST_STATE || c.type == EMPC_COST_CONTROL || c.type == EMPC_COST_CONTACT_FRICTION_CONE)
            continue;
          const bool mine = (1 + (k % (lpr - 1)) == lane);
          ++k;
          if (!mine) continue;
          const double* F = N + SM::OFF_CAP;
#pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk < ncap && capf[kk] == c.frame) F = N + SM::OFF_CAP + kk * SM::CAP;
          double r[6];
          int nr = 6;
          if (c.type == EMPC_COST_FRAME_PLACEMENT) {
            double rR[9], dp[3], rp[3], qq[4];
            matTmul3<double>(c.ref + 3, F, rR);
#pragma unroll
            for (int i = 0; i < 3; ++i) dp[i] = F[9 + i] - c.ref[i];
            matTvec3<double>(c.ref + 3, dp, rp);
            R_to_quat(rR, qq);
            log6_quat(qq, rp, r);
          } else if (c.type == EMPC_COST_FRAME_ROTATION) {
            double rR[9], qq[4];
            matTmul3<double>(c.ref, F, rR);
            R_to_quat(rR, qq);
            quat_log3(qq, r);
            nr = 3;
          } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
#pragma unroll
            for (int i = 0; i < 3; ++i) r[i] = F[9 + i] - c.ref[i];
            nr = 3;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = F[12 + i] - c.ref[i];
          }
          double cval = 0;
          for (int i = 0; i < nr; ++i) {
            double av, Ar, Arr;
            activation1(c.activation, r[i], c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
            cval += av;
          }
          S[ci] = c.weight * cval;
        }
      }
    });
    ex.sync();
    if (t == T) {
      ex.each([&](int lane, int sl) {
        if (lane == 0 && N[SM::OFF_FLAG] == 0.0) close_knot(T, sl);
      });
      ex.sync();
    }
  }
  ex.each([&](int lane, int sl) {
    if (lane != 0) return;
    D.try_cost[slot] = cost_l[sl];
    D.try_dv[slot] = dv_l[sl];
    D.try_ok[slot] = (N[SM::OFF_FLAG] == 0.0) ? 1 : 0;
  });
}

}  // namespace empc
