// ORACLE (test infrastructure only -- never linked into or called from the product path).
//
// action.hpp: one node of the shooting problem = IntegratedActionModelEuler(DifferentialActionModel
// {Free,Contact}FwdDynamics(StateMultibody, ActuationSquashingModel(ActuationModelMultiCopterBase,
// SquashingModelSmoothSat), CostModelSum)).  Restates the Crocoddyl (~v1.8) semantics specified in
// SURVEY.md Appendix A.3-A.7 for the objects the reference constructs at
//   src/factory/int-action.cpp:26, src/factory/diff-action.cpp:31,34, src/trajectory.cpp:47-52,
//   src/factory/cost.cpp:38-168, src/factory/activation.cpp:35-96, src/factory/contacts.cpp:49-79,
// and the barrier cost of src/sbfddp.cpp:169-190,464-477.
// Parity status: UNPINNED (Crocoddyl fork not available).
#pragma once
#include <algorithm>
#include <cstring>
#include <limits>
#include <vector>

#include "dynamics.hpp"

namespace oracle {

constexpr int NX = EMPC_MAX_NX;
constexpr int NDX = EMPC_MAX_NDX;
constexpr int NU = EMPC_MAX_NU;
constexpr int NR = EMPC_MAX_NR;
constexpr int NCM = 6 * EMPC_MAX_CONTACTS;  // constraint rows of a stage's ContactModelMultiple (each contact 3 or 6)

struct Problem {
  EmpcProblemDesc d;
  std::vector<EmpcCostSet> sets;
  std::vector<int32_t> knot_set;
  EmpcSolverParams prm;
  double smooth;  // SquashingModelSmoothSat::smooth (set_smooth at src/sbfddp.cpp:462)

  int nq() const { return d.model.nq; }
  int nv() const { return d.model.nv; }
};

// ---- StateMultibody ---------------------------------------------------------------
inline void state_zero(const Problem& P, double* x) {
  for (int i = 0; i < P.d.nx; ++i) x[i] = 0;
  x[6] = 1.0;
}
// dx = x1 (-) x0   (crocoddyl StateMultibody::diff(x0, x1, dx))
inline void state_diff(const Problem& P, const double* x0, const double* x1, double* dx) {
  const int nq = P.nq(), nv = P.nv();
  double qc[4], qd[4], R[9], dp[3], dpl[3], R0[9];
  quat_conj(x0 + 3, qc);
  quat_mul(qc, x1 + 3, qd);
  quat_normalize(qd);
  quat_to_R(qd, R);
  for (int i = 0; i < 3; ++i) dp[i] = x1[i] - x0[i];
  quat_to_R(x0 + 3, R0);
  matTvec3<double>(R0, dp, dpl);
  // log6 through the quaternion (keeps exact zeros for identical inputs)
  double w[3];
  quat_log3(qd, w);
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = std::sqrt(t2);
  double al, be;
  if (t < 1e-3) {
    al = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    be = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0;
  } else {
    const double st = std::sin(t), ct = std::cos(t);
    al = t * st / (2.0 * (1.0 - ct));
    be = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3];
  cross3<double>(w, dpl, wxp);
  const double wp = dot3<double>(w, dpl);
  for (int i = 0; i < 3; ++i) {
    dx[i] = al * dpl[i] - 0.5 * wxp[i] + be * wp * w[i];
    dx[3 + i] = w[i];
  }
  for (int i = 7; i < nq; ++i) dx[i - 1] = x1[i] - x0[i];
  for (int i = 0; i < nv; ++i) dx[nv + i] = x1[nq + i] - x0[nq + i];
}
// xout = x (+) dx   (StateMultibody::integrate)
inline void state_integrate(const Problem& P, const double* x, const double* dx, double* xout) {
  const int nq = P.nq(), nv = P.nv();
  double Re[9], pe[3], R0[9], Rp[3], qe[4], qn[4];
  exp6(dx, Re, pe);
  quat_to_R(x + 3, R0);
  matvec3<double>(R0, pe, Rp);
  quat_exp3(dx + 3, qe);
  quat_mul(x + 3, qe, qn);
  quat_normalize(qn);
  for (int i = 0; i < 3; ++i) xout[i] = x[i] + Rp[i];
  for (int i = 0; i < 4; ++i) xout[3 + i] = qn[i];
  for (int i = 7; i < nq; ++i) xout[i] = x[i] + dx[i - 1];
  for (int i = 0; i < nv; ++i) xout[nq + i] = x[nq + i] + dx[nv + i];
}
// Jdiff w.r.t. the second argument: blkdiag(Jlog6(M0^-1 M1), I, I); returns only the 6x6 block.
inline void state_Jdiff_second_block(const Problem& P, const double* x0, const double* x1, double* J6) {
  double dx[NDX];
  state_diff(P, x0, x1, dx);
  Jlog6(dx, J6);
}

// ---- SquashingModelSmoothSat ------------------------------------------------------
inline void squash(const Problem& P, const double* s, double* u, double* du_ds) {
  for (int i = 0; i < P.d.nu; ++i) {
    const double lb = P.d.u_lb[i], ub = P.d.u_ub[i];
    const double dd = P.smooth * (ub - lb);
    const double a = (P.prm.smoothsat_power == 4) ? dd * dd * dd * dd : dd * dd;
    const double sl = std::sqrt((s[i] - lb) * (s[i] - lb) + a);
    const double su = std::sqrt((s[i] - ub) * (s[i] - ub) + a);
    u[i] = 0.5 * (sl - su + ub + lb);
    if (du_ds) du_ds[i] = 0.5 * ((s[i] - lb) / sl - (s[i] - ub) / su);
  }
}

// ---- per-node result ----------------------------------------------------------------------------
struct NodeData {
  double xnext[NX];
  double cost;
  double xout[NV];        // acceleration
  double u_squash[NU];    // sigma(s): what SolverSbFDDP::fillSquashedOutputs copies (src/sbfddp.cpp:479-486)
  double lambda[NCM];     // contact forces (LOCAL), stacked in the stage's contact order, when contacts are active
  double Fx[NDX * NDX], Fu[NDX * NU];
  double Lx[NDX], Lu[NU], Lxx[NDX * NDX], Lxu[NDX * NU], Luu[NU * NU];
};

// activation a(r), Ar, diag(Arr)   (crocoddyl Activation{Quad,WeightedQuad,QuadraticBarrier,WeightedQuadraticBarrier})
inline double activation(const EmpcCost& c, const double* r, double* Ar, double* Arr) {
  double a = 0;
  for (int i = 0; i < c.nr; ++i) {
    switch (c.activation) {
      case EMPC_ACT_QUAD:
        a += 0.5 * r[i] * r[i];
        Ar[i] = r[i];
        Arr[i] = 1.0;
        break;
      case EMPC_ACT_WEIGHTED_QUAD:
        a += 0.5 * c.act_w[i] * r[i] * r[i];
        Ar[i] = c.act_w[i] * r[i];
        Arr[i] = c.act_w[i];
        break;
      case EMPC_ACT_QUADRATIC_BARRIER:
      case EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER: {
        const double w = (c.activation == EMPC_ACT_QUADRATIC_BARRIER) ? 1.0 : c.act_w[i];
        const double lo = std::min(r[i] - c.lb[i], 0.0);
        const double hi = std::max(r[i] - c.ub[i], 0.0);
        a += 0.5 * w * lo * lo + 0.5 * w * hi * hi;
        Ar[i] = w * (lo + hi);
        const double ind = ((r[i] - c.lb[i] <= 0.0) ? 1.0 : 0.0) + ((r[i] - c.ub[i] >= 0.0) ? 1.0 : 0.0);
        Arr[i] = w * ind;
      } break;
    }
  }
  return a;
}

// Differential action model at (x, s): acceleration, contact force, cost sum and -- with diff -- their derivatives, all
// UNSCALED (the integrator applies dt).  u == nullptr means the terminal call calc(x) == calc(x, u = 0).
// Outputs: D.xout, D.lambda, D.u_squash, ell; with diff also D.Lx .. D.Luu (unscaled) and da_dx (nv x ndx), da_du (nv x nu).
inline void dam_eval(const Problem& P, int t, const double* x, const double* u_in, bool diff, NodeData& D, double& ell_out,
                     double* da_dx, double* da_du) {
  const EmpcModelDesc& m = P.d.model;
  const int nq = m.nq, nv = m.nv, ndx = P.d.ndx, nu = P.d.nu, nr_rot = P.d.n_rotors;
  const EmpcCostSet& set = P.sets[P.knot_set[t]];
  const double dt = P.d.dt;
  const bool terminal = (u_in == nullptr);
  double uz[NU] = {0};
  const double* s = terminal ? uz : u_in;

  // --- actuation: tau = B sigma(s)  (A.5)
  double u[NU], dus[NU];
  if (P.d.use_squash) {
    squash(P, s, u, dus);
  } else {
    for (int i = 0; i < nu; ++i) {
      u[i] = s[i];
      dus[i] = 1.0;
    }
  }
  for (int i = 0; i < nu; ++i) D.u_squash[i] = u[i];
  double tau[NV];
  for (int r = 0; r < 6; ++r) {
    double acc = 0;
    for (int c = 0; c < nr_rot; ++c) acc += P.d.tau_f[r * nr_rot + c] * u[c];
    tau[r] = acc;
  }
  for (int i = 6; i < nv; ++i) tau[i] = u[nr_rot + i - 6];

  // --- dynamics: a = M^-1 (tau - h)  (A.6); contact variant (A.7) handled below
  const double* q = x;
  const double* v = x + nq;
  double R0[9], cs[NB], sn[NB], zero[NV] = {0};
  plain_state(m, q, R0, cs, sn);
  Kin<double> kin;
  double h[NV];
  rnea<double>(m, R0, q, cs, sn, v, zero, nullptr, h, kin);
  double M[NV * NV], L[NV * NV];
  crba(m, kin, M);
  std::memcpy(L, M, sizeof(double) * nv * nv);
  // contacts of this node (ContactModelMultiple of the stage, src/stage.cpp:38-48: every name of the stage's list is
  // added; crocoddyl stacks their rows in the order of its name-sorted map, which is the order of set.contacts[]); nc rows
  // in total, contact k owns rows coff[k] .. coff[k] + cnr[k] - 1
  int nc = 0;
  double Jc[NCM * NV], a0[NCM];
  const bool use_contact = P.d.has_contact && set.ncontacts > 0;
  const int ncon = use_contact ? set.ncontacts : 0;
  int coff[EMPC_MAX_CONTACTS] = {0}, cnr[EMPC_MAX_CONTACTS] = {0};
  Kin<double> kin0;  // true accelerations with qdd = 0
  FrameKin<double> cfks[EMPC_MAX_CONTACTS];
  if (use_contact) forward_kin<double>(m, R0, q, cs, sn, v, zero, false, kin0);
  for (int k = 0; k < ncon; ++k) {
    // ContactModel3D/6D (A.7): LOCAL frame Jacobian rows and drift a0
    const EmpcContact& ct = set.contacts[k];
    FrameKin<double>& cfk = cfks[k];
    frame_kin<double>(m, kin0, ct.frame, cfk);
    const int nk = (ct.type == EMPC_CONTACT_3D) ? 3 : 6;
    const int o = nc;
    coff[k] = o;
    cnr[k] = nk;
    nc += nk;
    // frame Jacobian (LOCAL) via unit velocities: column j = local frame velocity for v = e_j
    for (int j = 0; j < nv; ++j) {
      double ej[NV] = {0};
      ej[j] = 1.0;
      Kin<double> kj;
      FrameKin<double> fj;
      forward_kin<double>(m, R0, q, cs, sn, ej, zero, false, kj);
      frame_kin<double>(m, kj, ct.frame, fj);
      for (int r = 0; r < nk; ++r) Jc[(o + r) * nv + j] = fj.v[r];
    }
    // drift: 3D = classical acceleration of the frame origin, 6D = spatial acceleration (+ Baumgarte terms)
    if (nk == 3) {
      double wxv[3];
      cross3<double>(cfk.v + 3, cfk.v, wxv);
      for (int r = 0; r < 3; ++r) a0[o + r] = cfk.a[r] + wxv[r];
    } else {
      for (int r = 0; r < 6; ++r) a0[o + r] = cfk.a[r];
    }
    if (ct.gains[0] != 0.0) {
      if (nk == 3) {
        // crocoddyl 1.8 ContactModel3D::calc: a0 += gains[0] * (oMf.translation() - xref)   (world-frame error)
        for (int r = 0; r < 3; ++r) a0[o + r] += ct.gains[0] * (cfk.p[r] - ct.ref_p[r]);
      } else {
        double rR[9], dp[3], rp[3], xi[6];
        matTmul3<double>(ct.ref_R, cfk.R, rR);
        for (int r = 0; r < 3; ++r) dp[r] = cfk.p[r] - ct.ref_p[r];
        matTvec3<double>(ct.ref_R, dp, rp);
        log6(rR, rp, xi);
        for (int r = 0; r < 6; ++r) a0[o + r] += ct.gains[0] * xi[r];
      }
    }
    if (ct.gains[1] != 0.0)
      for (int r = 0; r < nk; ++r) a0[o + r] += ct.gains[1] * cfk.v[r];
  }

  double a[NV];
  bool ok = cholesky(L, nv);
  (void)ok;
  double lam[NCM] = {0};
  double MinvJt[NV * NCM], Lc[NCM * NCM];  // M^-1 Jc^T and chol(Jc M^-1 Jc^T)
  if (!use_contact) {
    for (int i = 0; i < nv; ++i) a[i] = tau[i] - h[i];
    cholesky_solve(L, nv, a);
  } else {
    // [M Jc^T; Jc 0] [a; -lam] = [tau - h; -a0]   =>  lam = -(Jc M^-1 Jc^T)^-1 (Jc afree + a0),  a = afree + M^-1 Jc^T lam
    double afree[NV];
    for (int i = 0; i < nv; ++i) afree[i] = tau[i] - h[i];
    cholesky_solve(L, nv, afree);
    for (int r = 0; r < nc; ++r) {
      double col[NV];
      for (int i = 0; i < nv; ++i) col[i] = Jc[r * nv + i];
      cholesky_solve(L, nv, col);
      for (int i = 0; i < nv; ++i) MinvJt[i * NCM + r] = col[i];
    }
    for (int r = 0; r < nc; ++r)
      for (int c = 0; c < nc; ++c) {
        double acc = 0;
        for (int i = 0; i < nv; ++i) acc += Jc[r * nv + i] * MinvJt[i * NCM + c];
        Lc[r * nc + c] = acc;
      }
    if (ncon > 1)
      cholesky_rank_deficient(Lc, nc);  // (several contacts: see its comment -- the one stated deviation from Eigen's LLT)
    else
      cholesky(Lc, nc);
    for (int r = 0; r < nc; ++r) {
      double acc = a0[r];
      for (int i = 0; i < nv; ++i) acc += Jc[r * nv + i] * afree[i];
      lam[r] = -acc;
    }
    cholesky_solve(Lc, nc, lam);
    for (int i = 0; i < nv; ++i) {
      double acc = afree[i];
      for (int r = 0; r < nc; ++r) acc += MinvJt[i * NCM + r] * lam[r];
      a[i] = acc;
    }
  }
  for (int i = 0; i < nv; ++i) D.xout[i] = a[i];
  for (int i = 0; i < NCM; ++i) D.lambda[i] = lam[i];

  // --- derivatives of the dynamics
  //   dtau_dx: derivative of RNEA(q,v,a) - Jc^T lam at fixed (a, lam); da0_dx: derivative of the contact drift + Jc a
  double dtau_dx[NV * NDX];
  double dcon_dx[NCM * NDX];
  static thread_local Kin<Dual> kd_s, kd0_s;
  static thread_local Dual fext_s[NB][6];
  static thread_local DualState ds;
  Kin<Dual>* kd = &kd_s;
  if (diff) {
    seed_dual_state(m, q, v, ds);
    Dual ad[NV], taud[NV];
    for (int i = 0; i < nv; ++i) ad[i] = Dual(a[i]);
    Dual(*fext)[6] = nullptr;
    if (use_contact) {
      // external force on each contact body = X_f^* lam_k (lam_k in the LOCAL frame, body coordinates); contacts on one body add up
      fext = fext_s;
      for (int bb = 0; bb < NB; ++bb)
        for (int i = 0; i < 6; ++i) fext[bb][i] = Dual(0.0);
      for (int k = 0; k < ncon; ++k) {
        const EmpcContact& ct = set.contacts[k];
        const int b = m.frame_body[ct.frame];
        const double* lk = lam + coff[k];
        double fl[3] = {lk[0], lk[1], lk[2]}, fn[3] = {0, 0, 0};
        if (cnr[k] == 6) {
          fn[0] = lk[3];
          fn[1] = lk[4];
          fn[2] = lk[5];
        }
        double fb[3], nb_[3], nb2[3], rxf[3];
        matvec3<double>(m.frame_R[ct.frame], fl, fb);
        matvec3<double>(m.frame_R[ct.frame], fn, nb_);
        cross3<double>(m.frame_p[ct.frame], fb, rxf);
        for (int i = 0; i < 3; ++i) nb2[i] = nb_[i] + rxf[i];
        for (int i = 0; i < 3; ++i) {
          fext[b][i] = fext[b][i] + Dual(fb[i]);
          fext[b][3 + i] = fext[b][3 + i] + Dual(nb2[i]);
        }
      }
    }
    rnea<Dual>(m, ds.R0, ds.p0, ds.cs, ds.sn, ds.v, ad, fext, taud, *kd);
    for (int r = 0; r < nv; ++r)
      for (int c = 0; c < ndx; ++c) dtau_dx[r * ndx + c] = taud[r].d[c];
    if (use_contact) {
      Kin<Dual>* kd0 = &kd0_s;
      forward_kin<Dual>(m, ds.R0, ds.p0, ds.cs, ds.sn, ds.v, ad, false, *kd0);
      for (int k = 0; k < ncon; ++k) {
        const EmpcContact& ct = set.contacts[k];
        const FrameKin<double>& cfk = cfks[k];
        const int nk = cnr[k], o = coff[k];
        FrameKin<Dual> fk;
        frame_kin<Dual>(m, *kd0, ct.frame, fk);
        Dual con[6];
        if (nk == 3) {
          Dual wxv[3];
          cross3<Dual>(fk.v + 3, fk.v, wxv);
          for (int r = 0; r < 3; ++r) con[r] = fk.a[r] + wxv[r];
        } else {
          for (int r = 0; r < 6; ++r) con[r] = fk.a[r];
        }
        if (ct.gains[1] != 0.0)
          for (int r = 0; r < nk; ++r) con[r] += ct.gains[1] * fk.v[r];
        if (ct.gains[0] != 0.0 && nk == 3)  // derivative: gains[0] * oRf * fJf.topRows<3>() (ContactModel3D::calcDiff)
          for (int r = 0; r < 3; ++r) con[r] += ct.gains[0] * (fk.p[r] - ct.ref_p[r]);
        for (int r = 0; r < nk; ++r)
          for (int c = 0; c < ndx; ++c) dcon_dx[(o + r) * ndx + c] = con[r].d[c];
        if (ct.gains[0] != 0.0 && nk == 6) {
          // ContactModel6D::calcDiff: da0_dq += gains[0] * Jlog6(rMf) * fJf, rMf = Mref^-1 oMf, fJf the LOCAL frame Jacobian
          double rR[9], dp[3], rp[3], xi[6], J6[36];
          matTmul3<double>(ct.ref_R, cfk.R, rR);
          for (int r = 0; r < 3; ++r) dp[r] = cfk.p[r] - ct.ref_p[r];
          matTvec3<double>(ct.ref_R, dp, rp);
          log6(rR, rp, xi);
          Jlog6(xi, J6);
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < nv; ++c) {
              double acc = 0;
              for (int l = 0; l < 6; ++l) acc += J6[r * 6 + l] * Jc[(o + l) * nv + c];
              dcon_dx[(o + r) * ndx + c] += ct.gains[0] * acc;
            }
        }
      }
    }
  }

  // da/dx (nv x ndx), da/du (nv x nu) -> caller; dlam/dx, dlam/du
  double dl_dx[NCM * NDX], dl_du[NCM * NU];
  if (diff) {
    // actuation derivative dtau/ds = B diag(sigma')
    double dtau_du[NV * NU];
    for (int r = 0; r < nv; ++r)
      for (int c = 0; c < nu; ++c) {
        double Brc = 0;
        if (r < 6)
          Brc = (c < nr_rot) ? P.d.tau_f[r * nr_rot + c] : 0.0;
        else
          Brc = (c == nr_rot + r - 6) ? 1.0 : 0.0;
        dtau_du[r * nu + c] = Brc * dus[c];
      }
    auto solve_cols = [&](const double* rhs_tau /*nv x n*/, const double* rhs_con /*nc x n or null*/, int n,
                          double* dA, double* dLam) {
      for (int c = 0; c < n; ++c) {
        double col[NV];
        for (int i = 0; i < nv; ++i) col[i] = rhs_tau[i * n + c];
        cholesky_solve(L, nv, col);  // M^-1 rhs
        if (!use_contact) {
          for (int i = 0; i < nv; ++i) dA[i * n + c] = col[i];
        } else {
          // KKT: [M Jc^T; Jc 0][da; -dlam] = [rhs_tau; -rhs_con]
          double y[NCM];
          for (int r = 0; r < nc; ++r) {
            double acc = rhs_con ? rhs_con[r * n + c] : 0.0;
            for (int i = 0; i < nv; ++i) acc += Jc[r * nv + i] * col[i];
            y[r] = -acc;
          }
          cholesky_solve(Lc, nc, y);
          for (int r = 0; r < nc; ++r) dLam[r * n + c] = y[r];
          for (int i = 0; i < nv; ++i) {
            double acc = col[i];
            for (int r = 0; r < nc; ++r) acc += MinvJt[i * NCM + r] * y[r];
            dA[i * n + c] = acc;
          }
        }
      }
    };
    double rhs[NV * NDX];
    for (int i = 0; i < nv * ndx; ++i) rhs[i] = -dtau_dx[i];
    solve_cols(rhs, use_contact ? dcon_dx : nullptr, ndx, da_dx, dl_dx);
    solve_cols(dtau_du, nullptr, nu, da_du, dl_du);
  }

  // --- costs (A.6): CostModelSum over the stage's table, alphabetical order
  double ell = 0;
  if (diff) {
    std::memset(D.Lx, 0, sizeof(double) * ndx);
    std::memset(D.Lu, 0, sizeof(double) * nu);
    std::memset(D.Lxx, 0, sizeof(double) * ndx * ndx);
    std::memset(D.Lxu, 0, sizeof(double) * ndx * nu);
    std::memset(D.Luu, 0, sizeof(double) * nu * nu);
  }
  for (int ci = 0; ci < set.ncosts; ++ci) {
    const EmpcCost& c = set.costs[ci];
    if (!c.active) continue;
    double r[NR];
    static thread_local double Rx[NR * NDX], Ru[NR * NU];
    const int nr = c.nr;
    if (diff) {
      std::memset(Rx, 0, sizeof(double) * nr * ndx);
      std::memset(Ru, 0, sizeof(double) * nr * nu);
    }
    switch (c.type) {
      case EMPC_COST_STATE: {
        state_diff(P, c.ref, x, r);
        if (diff) {
          double J6[36];
          Jlog6(r, J6);
          for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) Rx[i * ndx + j] = J6[i * 6 + j];
          for (int i = 6; i < ndx; ++i) Rx[i * ndx + i] = 1.0;
        }
      } break;
      case EMPC_COST_CONTROL: {
        for (int i = 0; i < nu; ++i) r[i] = s[i] - c.ref[i];
        if (diff)
          for (int i = 0; i < nu; ++i) Ru[i * nu + i] = 1.0;
      } break;
      case EMPC_COST_FRAME_PLACEMENT:
      case EMPC_COST_FRAME_ROTATION:
      case EMPC_COST_FRAME_TRANSLATION:
      case EMPC_COST_FRAME_VELOCITY: {
        FrameKin<double> fk;
        frame_kin<double>(m, kin, c.frame, fk);
        double fJ[6 * NV], dv_dq[6 * NV];  // LOCAL frame Jacobian, d(local velocity)/dq
        if (diff) {
          FrameKin<Dual> fkd;
          frame_kin<Dual>(m, *kd, c.frame, fkd);
          for (int j = 0; j < nv; ++j) {
            double dp[3], dR[9], l[3], RtdR[9];
            for (int i = 0; i < 3; ++i) dp[i] = fkd.p[i].d[j];
            for (int i = 0; i < 9; ++i) dR[i] = fkd.R[i].d[j];
            matTvec3<double>(fk.R, dp, l);
            matTmul3<double>(fk.R, dR, RtdR);
            fJ[0 * nv + j] = l[0];
            fJ[1 * nv + j] = l[1];
            fJ[2 * nv + j] = l[2];
            fJ[3 * nv + j] = RtdR[7];  // vee of a skew matrix: (2,1), (0,2), (1,0)
            fJ[4 * nv + j] = RtdR[2];
            fJ[5 * nv + j] = RtdR[3];
            for (int i = 0; i < 6; ++i) dv_dq[i * nv + j] = fkd.v[i].d[j];
          }
        }
        if (c.type == EMPC_COST_FRAME_PLACEMENT) {
          const double* pref = c.ref;
          const double* Rref = c.ref + 3;
          double rR[9], dp[3], rp[3];
          matTmul3<double>(Rref, fk.R, rR);
          for (int i = 0; i < 3; ++i) dp[i] = fk.p[i] - pref[i];
          matTvec3<double>(Rref, dp, rp);
          log6(rR, rp, r);
          if (diff) {
            double J6[36];
            Jlog6(r, J6);
            for (int i = 0; i < 6; ++i)
              for (int j = 0; j < nv; ++j) {
                double acc = 0;
                for (int l = 0; l < 6; ++l) acc += J6[i * 6 + l] * fJ[l * nv + j];
                Rx[i * ndx + j] = acc;
              }
          }
        } else if (c.type == EMPC_COST_FRAME_ROTATION) {
          double rR[9];
          matTmul3<double>(c.ref, fk.R, rR);
          log3(rR, r);
          if (diff) {
            double J3[9];
            Jlog3(r, J3);
            for (int i = 0; i < 3; ++i)
              for (int j = 0; j < nv; ++j) {
                double acc = 0;
                for (int l = 0; l < 3; ++l) acc += J3[i * 3 + l] * fJ[(3 + l) * nv + j];
                Rx[i * ndx + j] = acc;
              }
          }
        } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
          for (int i = 0; i < 3; ++i) r[i] = fk.p[i] - c.ref[i];
          if (diff)
            for (int i = 0; i < 3; ++i)
              for (int j = 0; j < nv; ++j) {
                double acc = 0;
                for (int l = 0; l < 3; ++l) acc += fk.R[i * 3 + l] * fJ[l * nv + j];
                Rx[i * ndx + j] = acc;
              }
        } else {  // FRAME_VELOCITY, LOCAL
          for (int i = 0; i < 6; ++i) r[i] = fk.v[i] - c.ref[i];
          if (diff)
            for (int i = 0; i < 6; ++i)
              for (int j = 0; j < nv; ++j) {
                Rx[i * ndx + j] = dv_dq[i * nv + j];
                Rx[i * ndx + nv + j] = fJ[i * nv + j];
              }
        }
      } break;
      case EMPC_COST_CONTACT_FRICTION_CONE: {
        // FrictionCone(n, mu, nf=4, inner_appr=false): A rows [+-1,0,-mu],[0,+-1,-mu] (surface frame),[0,0,1]
        // Crocoddyl builds A with the cone rotated to the normal; for n = (0,0,1) the rotation is identity.
        const double mu = c.ref[3];
        double A[5][3] = {{1, 0, -mu}, {0, 1, -mu}, {-1, 0, -mu}, {0, -1, -mu}, {0, 0, 1}};
        // general normal: rotate rows by the minimal rotation taking e3 to n
        const double nn[3] = {c.ref[0], c.ref[1], c.ref[2]};
        double nrm = std::sqrt(dot3<double>(nn, nn));
        double nz[3] = {nn[0] / nrm, nn[1] / nrm, nn[2] / nrm};
        double Rn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        {
          const double e3[3] = {0, 0, 1};
          double ax[3];
          cross3<double>(e3, nz, ax);
          const double sn_ = std::sqrt(dot3<double>(ax, ax)), cs_ = nz[2];
          if (sn_ > 1e-12) {
            double w[3] = {ax[0] / sn_ * std::atan2(sn_, cs_), ax[1] / sn_ * std::atan2(sn_, cs_),
                           ax[2] / sn_ * std::atan2(sn_, cs_)};
            exp3(w, Rn);
          } else if (cs_ < 0) {
            Rn[4] = -1;
            Rn[8] = -1;
          }
        }
        double AR[5][3];
        for (int i = 0; i < 5; ++i)
          for (int j = 0; j < 3; ++j) {
            double acc = 0;
            for (int l = 0; l < 3; ++l) acc += A[i][l] * Rn[3 * j + l];  // A R^T
            AR[i][j] = acc;
          }
        const bool has = use_contact;
        // the force of the contact on the cost's frame (crocoddyl's residual data looks the contact up by frame id);
        // a stage with one contact keeps that contact whatever the cost's frame says (behaviour of rounds 1-5)
        int fo = 0;
        for (int k = 0; k < ncon; ++k)
          if (ncon > 1 && set.contacts[k].frame == c.frame) fo = coff[k];
        for (int i = 0; i < 5; ++i) {
          double acc = 0;
          if (has)
            for (int l = 0; l < 3; ++l) acc += AR[i][l] * lam[fo + l];
          r[i] = acc;
        }
        if (diff && has) {
          for (int i = 0; i < 5; ++i) {
            for (int j = 0; j < ndx; ++j) {
              double acc = 0;
              for (int l = 0; l < 3; ++l) acc += AR[i][l] * dl_dx[(fo + l) * ndx + j];
              Rx[i * ndx + j] = acc;
            }
            for (int j = 0; j < nu; ++j) {
              double acc = 0;
              for (int l = 0; l < 3; ++l) acc += AR[i][l] * dl_du[(fo + l) * nu + j];
              Ru[i * nu + j] = acc;
            }
          }
        }
      } break;
    }
    double Ar[NR], Arr[NR];
    const double aval = activation(c, r, Ar, Arr);
    ell += c.weight * aval;
    if (diff) {
      const double w = c.weight;
      for (int j = 0; j < ndx; ++j) {
        double acc = 0;
        for (int i = 0; i < nr; ++i) acc += Rx[i * ndx + j] * Ar[i];
        D.Lx[j] += w * acc;
      }
      for (int j = 0; j < nu; ++j) {
        double acc = 0;
        for (int i = 0; i < nr; ++i) acc += Ru[i * nu + j] * Ar[i];
        D.Lu[j] += w * acc;
      }
      for (int a_ = 0; a_ < ndx; ++a_)
        for (int b_ = 0; b_ < ndx; ++b_) {
          double acc = 0;
          for (int i = 0; i < nr; ++i) acc += Rx[i * ndx + a_] * Arr[i] * Rx[i * ndx + b_];
          D.Lxx[a_ * ndx + b_] += w * acc;
        }
      for (int a_ = 0; a_ < ndx; ++a_)
        for (int b_ = 0; b_ < nu; ++b_) {
          double acc = 0;
          for (int i = 0; i < nr; ++i) acc += Rx[i * ndx + a_] * Arr[i] * Ru[i * nu + b_];
          D.Lxu[a_ * nu + b_] += w * acc;
        }
      for (int a_ = 0; a_ < nu; ++a_)
        for (int b_ = 0; b_ < nu; ++b_) {
          double acc = 0;
          for (int i = 0; i < nr; ++i) acc += Ru[i * nu + a_] * Arr[i] * Ru[i * nu + b_];
          D.Luu[a_ * nu + b_] += w * acc;
        }
    }
  }
  ell_out = ell;
  (void)dt;
}

// blkdiag(J (6 x 6), I) applied from the left to an ndx x cols matrix (row-major, in place)
inline void apply_base_block(const double* J6, double* M, int ndx, int cols) {
  for (int j = 0; j < cols; ++j) {
    double col[6];
    for (int i = 0; i < 6; ++i) {
      double acc = 0;
      for (int l = 0; l < 6; ++l) acc += J6[i * 6 + l] * M[l * cols + j];
      col[i] = acc;
    }
    for (int i = 0; i < 6; ++i) M[i * cols + j] = col[i];
  }
}
// Jintegrate of StateMultibody at (x, d): first = d(x (+) d)/dx = blkdiag(Ad(exp6(d)^-1), I), second = d(x (+) d)/dd =
// blkdiag(Jexp6(d), I)   (SURVEY A.4); only the 6 x 6 base blocks are returned
inline void state_Jintegrate_blocks(const double* d, double* J1, double* J2) {
  double Re[9], pe[3], neg[6];
  Jexp6(d, J2);
  for (int i = 0; i < 6; ++i) neg[i] = -d[i];
  exp6(neg, Re, pe);
  adjoint6(Re, pe, J1);  // Ad(exp6(d)^-1) = Ad(exp6(-d))
}

// IntegratedActionModelEuler (A.3) on top of dam_eval.  If diff is false only xnext / cost / u_squash are produced.
inline void node_calc_euler(const Problem& P, int t, const double* x, const double* u_in, bool diff, NodeData& D) {
  const EmpcModelDesc& m = P.d.model;
  const int nq = m.nq, nv = m.nv, ndx = P.d.ndx, nu = P.d.nu;
  const double dt = P.d.dt;
  const bool terminal = (u_in == nullptr);
  static thread_local double da_dx[NV * NDX], da_du[NV * NU];
  double ell = 0;
  dam_eval(P, t, x, u_in, diff, D, ell, da_dx, da_du);
  const double* v = x + nq;
  const double* a = D.xout;
  // --- Euler step (A.3): dx = [v dt + a dt^2; a dt]
  double dxe[NDX];
  for (int i = 0; i < nv; ++i) {
    dxe[i] = v[i] * dt + a[i] * dt * dt;
    dxe[nv + i] = a[i] * dt;
  }
  state_integrate(P, x, dxe, D.xnext);
  // Euler scaling of the cost (A.3); the terminal node is a running IAM too (src/trajectory.cpp:135, U2)
  const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 : dt;
  D.cost = cscale * ell;

  if (diff) {
    for (int i = 0; i < ndx; ++i) D.Lx[i] *= cscale;
    for (int i = 0; i < nu; ++i) D.Lu[i] *= cscale;
    for (int i = 0; i < ndx * ndx; ++i) D.Lxx[i] *= cscale;
    for (int i = 0; i < ndx * nu; ++i) D.Lxu[i] *= cscale;
    for (int i = 0; i < nu * nu; ++i) D.Luu[i] *= cscale;
    // Fx = J2 [A dt^2 + [0 | I dt]; A dt] + J1,  Fu = J2 [B dt^2; B dt]   (A.3)
    double J2[36], J1[36], Re[9], pe[3], neg[6];
    Jexp6(dxe, J2);
    for (int i = 0; i < 6; ++i) neg[i] = -dxe[i];
    exp6(neg, Re, pe);
    adjoint6(Re, pe, J1);  // Ad(exp6(xi)^-1) = Ad(exp6(-xi))
    static thread_local double G[NDX * NDX], Gu[NDX * NU];
    for (int i = 0; i < nv; ++i) {
      for (int j = 0; j < ndx; ++j) {
        G[i * ndx + j] = da_dx[i * ndx + j] * dt * dt;
        G[(nv + i) * ndx + j] = da_dx[i * ndx + j] * dt;
      }
      G[i * ndx + nv + i] += dt;
      for (int j = 0; j < nu; ++j) {
        Gu[i * nu + j] = da_du[i * nu + j] * dt * dt;
        Gu[(nv + i) * nu + j] = da_du[i * nu + j] * dt;
      }
    }
    for (int i = 0; i < ndx; ++i) {
      for (int j = 0; j < ndx; ++j) {
        double acc;
        if (i < 6) {
          acc = 0;
          for (int l = 0; l < 6; ++l) acc += J2[i * 6 + l] * G[l * ndx + j];
          if (j < 6) acc += J1[i * 6 + j];
        } else {
          acc = G[i * ndx + j] + ((i == j) ? 1.0 : 0.0);
        }
        D.Fx[i * ndx + j] = acc;
      }
      for (int j = 0; j < nu; ++j) {
        double acc;
        if (i < 6) {
          acc = 0;
          for (int l = 0; l < 6; ++l) acc += J2[i * 6 + l] * Gu[l * nu + j];
        } else {
          acc = Gu[i * nu + j];
        }
        D.Fu[i * nu + j] = acc;
      }
    }
  }
}


// IntegratedActionModelRK4 (src/factory/int-action.cpp:29-31; crocoddyl ~1.8 integ-action/rk4.hxx): four evaluations of the
// differential model at y_i = x (+) c_i dt k_{i-1}, c = (0, 1/2, 1/2, 1), k_i = [v(y_i); a(y_i, u)];
// xnext = x (+) dt/6 (k0 + 2 k1 + 2 k2 + k3), cost = dt/6 (l0 + 2 l1 + 2 l2 + l3); derivatives by the chain rule through
// the stages, cost Hessians in Gauss-Newton form (second derivatives of y_i dropped, as crocoddyl does).  The squashing /
// contact data a caller reads afterwards are those of stage 0 (src/sbfddp.cpp:144-145 takes differential[0]).
// The terminal node is the same model called with u = 0 (U2).
inline void node_calc_rk4(const Problem& P, int t, const double* x, const double* u_in, bool diff, NodeData& D) {
  const EmpcModelDesc& m = P.d.model;
  const int nq = m.nq, nv = m.nv, n = P.d.ndx, nu = P.d.nu, nx = P.d.nx;
  const double dt = P.d.dt;
  const bool terminal = (u_in == nullptr);
  const double c[4] = {0.0, 0.5, 0.5, 1.0}, w[4] = {1.0, 2.0, 2.0, 1.0};
  static thread_local NodeData S[4];
  static thread_local double A[4][NV * NDX], B[4][NV * NU];
  double y[4][NX], k[4][NDX], dxr[4][NDX], ell[4];
  for (int i = 0; i < nx; ++i) y[0][i] = x[i];
  for (int i = 0; i < 4; ++i) {
    if (i > 0) {
      for (int j = 0; j < n; ++j) dxr[i][j] = c[i] * dt * k[i - 1][j];
      state_integrate(P, x, dxr[i], y[i]);
    }
    dam_eval(P, t, y[i], u_in, diff, S[i], ell[i], A[i], B[i]);
    for (int j = 0; j < nv; ++j) {
      k[i][j] = y[i][nq + j];
      k[i][nv + j] = S[i].xout[j];
    }
  }
  double dx[NDX];
  for (int j = 0; j < n; ++j) dx[j] = (k[0][j] + 2.0 * k[1][j] + 2.0 * k[2][j] + k[3][j]) * dt / 6.0;
  state_integrate(P, x, dx, D.xnext);
  const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 / 6.0 : dt / 6.0;
  D.cost = (ell[0] + 2.0 * ell[1] + 2.0 * ell[2] + ell[3]) * cscale;
  for (int j = 0; j < nv; ++j) D.xout[j] = S[0].xout[j];
  for (int j = 0; j < NCM; ++j) D.lambda[j] = S[0].lambda[j];
  for (int j = 0; j < nu; ++j) D.u_squash[j] = S[0].u_squash[j];
  if (!diff) return;
  // dk_i/dx (n x n), dk_i/du (n x nu), dy_i/dx, dy_i/du
  static thread_local double dkx[4][NDX * NDX], dku[4][NDX * NU], dyx[4][NDX * NDX], dyu[4][NDX * NU];
  for (int i = 0; i < 4; ++i) {
    if (i == 0) {
      std::memset(dyx[0], 0, sizeof(double) * n * n);
      for (int j = 0; j < n; ++j) dyx[0][j * n + j] = 1.0;
      std::memset(dyu[0], 0, sizeof(double) * n * nu);
    } else {
      double J1[36], J2[36];
      state_Jintegrate_blocks(dxr[i], J1, J2);
      for (int j = 0; j < n * n; ++j) dyx[i][j] = c[i] * dt * dkx[i - 1][j];
      for (int j = 0; j < n * nu; ++j) dyu[i][j] = c[i] * dt * dku[i - 1][j];
      apply_base_block(J2, dyx[i], n, n);
      apply_base_block(J2, dyu[i], n, nu);
      for (int r = 0; r < n; ++r)
        for (int q = 0; q < n; ++q) dyx[i][r * n + q] += (r < 6 && q < 6) ? J1[r * 6 + q] : ((r >= 6 && r == q) ? 1.0 : 0.0);
    }
    // k_i = [v(y_i); a(y_i, u)]: rows 0..nv-1 pick the velocity rows of dy_i, rows nv.. are A_i dy_i (+ B_i)
    for (int r = 0; r < nv; ++r) {
      for (int q = 0; q < n; ++q) dkx[i][r * n + q] = dyx[i][(nv + r) * n + q];
      for (int q = 0; q < nu; ++q) dku[i][r * nu + q] = dyu[i][(nv + r) * nu + q];
      for (int q = 0; q < n; ++q) {
        double acc = 0;
        for (int l = 0; l < n; ++l) acc += A[i][r * n + l] * dyx[i][l * n + q];
        dkx[i][(nv + r) * n + q] = acc;
      }
      for (int q = 0; q < nu; ++q) {
        double acc = B[i][r * nu + q];
        for (int l = 0; l < n; ++l) acc += A[i][r * n + l] * dyu[i][l * nu + q];
        dku[i][(nv + r) * nu + q] = acc;
      }
    }
  }
  {
    double J1[36], J2[36];
    state_Jintegrate_blocks(dx, J1, J2);
    for (int j = 0; j < n * n; ++j) D.Fx[j] = (dkx[0][j] + 2.0 * dkx[1][j] + 2.0 * dkx[2][j] + dkx[3][j]) * dt / 6.0;
    for (int j = 0; j < n * nu; ++j) D.Fu[j] = (dku[0][j] + 2.0 * dku[1][j] + 2.0 * dku[2][j] + dku[3][j]) * dt / 6.0;
    apply_base_block(J2, D.Fx, n, n);
    apply_base_block(J2, D.Fu, n, nu);
    for (int r = 0; r < n; ++r)
      for (int q = 0; q < n; ++q) D.Fx[r * n + q] += (r < 6 && q < 6) ? J1[r * 6 + q] : ((r >= 6 && r == q) ? 1.0 : 0.0);
  }
  // costs
  std::memset(D.Lx, 0, sizeof(double) * n);
  std::memset(D.Lu, 0, sizeof(double) * nu);
  std::memset(D.Lxx, 0, sizeof(double) * n * n);
  std::memset(D.Lxu, 0, sizeof(double) * n * nu);
  std::memset(D.Luu, 0, sizeof(double) * nu * nu);
  static thread_local double XX[NDX * NDX], XU[NDX * NU];  // lxx_i dy_i/dx, lxx_i dy_i/du
  for (int i = 0; i < 4; ++i) {
    const double wi = w[i] * cscale;
    const NodeData& Si = S[i];
    for (int q = 0; q < n; ++q) {
      double acc = 0;
      for (int l = 0; l < n; ++l) acc += dyx[i][l * n + q] * Si.Lx[l];
      D.Lx[q] += wi * acc;
    }
    for (int q = 0; q < nu; ++q) {
      double acc = Si.Lu[q];
      for (int l = 0; l < n; ++l) acc += dyu[i][l * nu + q] * Si.Lx[l];
      D.Lu[q] += wi * acc;
    }
    for (int r = 0; r < n; ++r) {
      for (int q = 0; q < n; ++q) {
        double acc = 0;
        for (int l = 0; l < n; ++l) acc += Si.Lxx[r * n + l] * dyx[i][l * n + q];
        XX[r * n + q] = acc;
      }
      for (int q = 0; q < nu; ++q) {
        double acc = 0;
        for (int l = 0; l < n; ++l) acc += Si.Lxx[r * n + l] * dyu[i][l * nu + q];
        XU[r * nu + q] = acc;
      }
    }
    for (int r = 0; r < n; ++r) {
      for (int q = 0; q < n; ++q) {
        double acc = 0;
        for (int l = 0; l < n; ++l) acc += dyx[i][l * n + r] * XX[l * n + q];
        D.Lxx[r * n + q] += wi * acc;
      }
      for (int q = 0; q < nu; ++q) {
        double acc = 0;
        for (int l = 0; l < n; ++l) acc += dyx[i][l * n + r] * (Si.Lxu[l * nu + q] + XU[l * nu + q]);
        D.Lxu[r * nu + q] += wi * acc;
      }
    }
    for (int r = 0; r < nu; ++r)
      for (int q = 0; q < nu; ++q) {
        double acc = Si.Luu[r * nu + q];
        for (int l = 0; l < n; ++l) acc += Si.Lxu[l * nu + r] * dyu[i][l * nu + q] + dyu[i][l * nu + r] * Si.Lxu[l * nu + q] +
                                           dyu[i][l * nu + r] * XU[l * nu + q];
        D.Luu[r * nu + q] += wi * acc;
      }
  }
}

// One node of the shooting problem with the problem's integrator (EmpcProblemDesc::integrator).
inline void node_calc(const Problem& P, int t, const double* x, const double* u_in, bool diff, NodeData& D) {
  if (P.d.integrator == EMPC_INTEGRATOR_RK4)
    node_calc_rk4(P, t, x, u_in, diff, D);
  else
    node_calc_euler(P, t, x, u_in, diff, D);
}

}  // namespace oracle
