// ORACLE (test infrastructure only -- never linked into or called from the product path).
//
// dynamics.hpp: rigid-body algorithms for a free-flyer + revolute tree, restating what the reference
// obtains from Pinocchio through Crocoddyl:
//   pinocchio::aba                   -> forward_dynamics()  (a = M^-1 (tau - h); CRBA + RNEA + LLT)
//   pinocchio::computeABADerivatives -> rnea_derivatives() + M^-1 (da/dx = -M^-1 dRNEA/dx at fixed a)
//   frame placements / Jacobians / velocities -> frame_kinematics<>
// Call sites in the reference: src/factory/diff-action.cpp:31,34 (DAM Free/Contact FwdDynamics).
//
// Formulation: classic body-frame Featherstone recursions (deliberately different from the
// world-frame tangent recursion used by the HIP kernels, so that agreement between the two is a
// meaningful check).  Derivatives come from mechanical forward-mode differentiation (oracle::Dual)
// of the very same templated recursions: exact derivatives, no finite differences.
// Parity status: UNPINNED (no Pinocchio here); checked by identities + finite differences in tests/.
#pragma once
#include "../include/empc_types.h"
#include "omath.hpp"

namespace oracle {

constexpr int NB = EMPC_MAX_BODIES;
constexpr int NV = EMPC_MAX_NV;

template <class S>
struct Kin {
  S R[NB][9], p[NB][3];  // world placement of every body
  S XR[NB][9];           // rotation child -> parent (joint placement * joint rotation)
  S v[NB][6], a[NB][6];  // body-frame spatial velocity / acceleration [lin; ang]
  S f[NB][6];            // body-frame spatial force
};

// Rodrigues rotation about a unit axis from cos/sin
template <class S>
inline void axis_rot(const double* ax, const S& c, const S& s, S* R) {
  S omc = 1.0 - c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = (ax[i] * ax[j]) * omc;
  R[0] += c;
  R[4] += c;
  R[8] += c;
  R[1] -= ax[2] * s;
  R[2] += ax[1] * s;
  R[3] += ax[2] * s;
  R[5] -= ax[0] * s;
  R[6] -= ax[1] * s;
  R[7] += ax[0] * s;
}

// Forward recursion: placements, velocities, accelerations.
//  R0,p0: base placement; cs,sn: cos/sin of joint angles (index = body-1); v,a: generalized velocity / acceleration.
//  with_gravity: fold -g into the base acceleration (RNEA trick) or not (true accelerations).
template <class S>
inline void forward_kin(const EmpcModelDesc& m, const S* R0, const S* p0, const S* cs, const S* sn, const S* v,
                        const S* a, bool with_gravity, Kin<S>& k) {
  for (int i = 0; i < 9; ++i) k.R[0][i] = R0[i];
  for (int i = 0; i < 3; ++i) k.p[0][i] = p0[i];
  for (int i = 0; i < 6; ++i) {
    k.v[0][i] = v[i];
    k.a[0][i] = a[i];
  }
  if (with_gravity) {
    double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
    S gl[3];
    matTvec3<S>(R0, ng, gl);
    for (int i = 0; i < 3; ++i) k.a[0][i] += gl[i];
  }
  for (int b = 1; b < m.nbodies; ++b) {
    const int par = m.parent[b];
    S Rj[9];
    axis_rot<S>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
    matmul3<S>(m.jplace_R[b], Rj, k.XR[b]);
    const double* r = m.jplace_p[b];
    // world placement
    matmul3<S>(k.R[par], k.XR[b], k.R[b]);
    S Rr[3];
    matvec3<S>(k.R[par], r, Rr);
    for (int i = 0; i < 3; ++i) k.p[b][i] = k.p[par][i] + Rr[i];
    // velocity: v_c = E (v_p + w_p x r), w_c = E w_p, E = XR^T; plus S qd
    const S* vp = k.v[par];
    S wxr[3], tmp[3];
    cross3<S>(vp + 3, r, wxr);
    for (int i = 0; i < 3; ++i) tmp[i] = vp[i] + wxr[i];
    matTvec3<S>(k.XR[b], tmp, k.v[b]);
    matTvec3<S>(k.XR[b], vp + 3, k.v[b] + 3);
    const S& qd = v[6 + b - 1];
    S sv[3];
    for (int i = 0; i < 3; ++i) sv[i] = m.axis[b][i] * qd;
    for (int i = 0; i < 3; ++i) k.v[b][3 + i] += sv[i];
    // acceleration: a_c = X a_p + S qdd + v_c x (S qd)
    const S* ap = k.a[par];
    cross3<S>(ap + 3, r, wxr);
    for (int i = 0; i < 3; ++i) tmp[i] = ap[i] + wxr[i];
    matTvec3<S>(k.XR[b], tmp, k.a[b]);
    matTvec3<S>(k.XR[b], ap + 3, k.a[b] + 3);
    const S& qdd = a[6 + b - 1];
    S c1[3], c2[3];
    cross3<S>(k.v[b], sv, c1);      // v x (axis qd)   (linear part)
    cross3<S>(k.v[b] + 3, sv, c2);  // w x (axis qd)   (angular part)
    for (int i = 0; i < 3; ++i) {
      k.a[b][i] += c1[i];
      k.a[b][3 + i] += m.axis[b][i] * qdd + c2[i];
    }
  }
}

// spatial inertia applied to a motion, body frame: [m (v + w x c); Ic w + c x m (v + w x c)]
template <class S>
inline void inertia_apply(const EmpcModelDesc& m, int b, const S* mot, S* out) {
  S wxc[3], lin[3], Iw[3], cxl[3];
  cross3<S>(mot + 3, m.com[b], wxc);
  for (int i = 0; i < 3; ++i) lin[i] = m.mass[b] * (mot[i] + wxc[i]);
  matvec3<S>(m.inertia[b], mot + 3, Iw);
  cross3<S>(m.com[b], lin, cxl);
  for (int i = 0; i < 3; ++i) {
    out[i] = lin[i];
    out[3 + i] = Iw[i] + cxl[i];
  }
}

// Backward recursion of RNEA. fext (may be null): external spatial force acting ON body b, body frame.
template <class S>
inline void rnea_backward(const EmpcModelDesc& m, Kin<S>& k, const S (*fext)[6], S* tau) {
  for (int b = 0; b < m.nbodies; ++b) {
    S Ia[6], Iv[6];
    inertia_apply<S>(m, b, k.a[b], Ia);
    inertia_apply<S>(m, b, k.v[b], Iv);
    // v x* (I v) = [w x f; w x n + v x f]
    S c1[3], c2[3], c3[3];
    cross3<S>(k.v[b] + 3, Iv, c1);
    cross3<S>(k.v[b] + 3, Iv + 3, c2);
    cross3<S>(k.v[b], Iv, c3);
    for (int i = 0; i < 3; ++i) {
      k.f[b][i] = Ia[i] + c1[i];
      k.f[b][3 + i] = Ia[3 + i] + c2[i] + c3[i];
    }
    if (fext)
      for (int i = 0; i < 6; ++i) k.f[b][i] -= fext[b][i];
  }
  for (int b = m.nbodies - 1; b >= 1; --b) {
    const int par = m.parent[b];
    tau[6 + b - 1] = dot3<S>(m.axis[b], k.f[b] + 3);
    // f_p += X^T f_c: lin = XR f, ang = XR n + r x (XR f)
    S fl[3], fn[3], rxf[3];
    matvec3<S>(k.XR[b], k.f[b], fl);
    matvec3<S>(k.XR[b], k.f[b] + 3, fn);
    cross3<S>(m.jplace_p[b], fl, rxf);
    for (int i = 0; i < 3; ++i) {
      k.f[par][i] += fl[i];
      k.f[par][3 + i] += fn[i] + rxf[i];
    }
  }
  for (int i = 0; i < 6; ++i) tau[i] = k.f[0][i];
}

template <class S>
inline void rnea(const EmpcModelDesc& m, const S* R0, const S* p0, const S* cs, const S* sn, const S* v, const S* a,
                 const S (*fext)[6], S* tau, Kin<S>& k) {
  forward_kin<S>(m, R0, p0, cs, sn, v, a, true, k);
  rnea_backward<S>(m, k, fext, tau);
}

// Joint-space inertia matrix by the composite-rigid-body algorithm (dense 6x6 composites, body frame).
inline void crba(const EmpcModelDesc& m, const Kin<double>& k, double* M /* nv x nv */) {
  const int nv = m.nv;
  double Ic[NB][36];
  for (int b = 0; b < m.nbodies; ++b) {
    double C[9], CC[9];
    skew3(m.com[b], C);
    matmul3<double>(C, C, CC);
    double* I6 = Ic[b];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        I6[6 * i + j] = (i == j) ? m.mass[b] : 0.0;
        I6[6 * i + 3 + j] = -m.mass[b] * C[3 * i + j];
        I6[6 * (3 + i) + j] = m.mass[b] * C[3 * i + j];
        I6[6 * (3 + i) + 3 + j] = m.inertia[b][3 * i + j] - m.mass[b] * CC[3 * i + j];
      }
  }
  for (int i = 0; i < nv * nv; ++i) M[i] = 0;
  // X (motion transform parent -> child) as a dense 6x6: [[E, -E [r]x],[0, E]], E = XR^T
  auto build_X = [&](int b, double* X) {
    double E[9], Rx[9], ERx[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) E[3 * i + j] = k.XR[b][3 * j + i];
    skew3(m.jplace_p[b], Rx);
    matmul3<double>(E, Rx, ERx);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        X[6 * i + j] = E[3 * i + j];
        X[6 * i + 3 + j] = -ERx[3 * i + j];
        X[6 * (3 + i) + j] = 0;
        X[6 * (3 + i) + 3 + j] = E[3 * i + j];
      }
  };
  for (int b = m.nbodies - 1; b >= 0; --b) {
    if (b == 0) {
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) M[i * nv + j] = Ic[0][6 * i + j];
    } else {
      const int col = 6 + b - 1;
      double F[6];
      for (int i = 0; i < 6; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s += Ic[b][6 * i + 3 + j] * m.axis[b][j];
        F[i] = s;
      }
      M[col * nv + col] = dot3<double>(m.axis[b], F + 3);
      int j = b;
      while (j > 0) {
        double X[36];
        build_X(j, X);
        double Fp[6];
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          for (int c = 0; c < 6; ++c) s += X[6 * c + r] * F[c];  // X^T F
          Fp[r] = s;
        }
        for (int r = 0; r < 6; ++r) F[r] = Fp[r];
        j = m.parent[j];
        if (j == 0) {
          for (int r = 0; r < 6; ++r) {
            M[r * nv + col] = F[r];
            M[col * nv + r] = F[r];
          }
        } else {
          const int row = 6 + j - 1;
          const double val = dot3<double>(m.axis[j], F + 3);
          M[row * nv + col] = val;
          M[col * nv + row] = val;
        }
      }
      // composite inertia to the parent: Ic_p += X^T Ic X
      double X[36], T[36];
      build_X(b, X);
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int l = 0; l < 6; ++l) s += Ic[b][6 * r + l] * X[6 * l + c];
          T[6 * r + c] = s;
        }
      const int par = m.parent[b];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int l = 0; l < 6; ++l) s += X[6 * l + r] * T[6 * l + c];
          Ic[par][6 * r + c] += s;
        }
    }
  }
}

// Frame placement and local (frame-axes) spatial velocity / acceleration of an operational frame.
template <class S>
struct FrameKin {
  S R[9], p[3];  // world placement
  S v[6];        // LOCAL spatial velocity
  S a[6];        // LOCAL spatial acceleration (of whatever Kin::a holds)
};
template <class S>
inline void frame_kin(const EmpcModelDesc& m, const Kin<S>& k, int f, FrameKin<S>& fk) {
  const int b = m.frame_body[f];
  matmul3<S>(k.R[b], m.frame_R[f], fk.R);
  S Rp[3];
  matvec3<S>(k.R[b], m.frame_p[f], Rp);
  for (int i = 0; i < 3; ++i) fk.p[i] = k.p[b][i] + Rp[i];
  S wxr[3], tmp[3];
  cross3<S>(k.v[b] + 3, m.frame_p[f], wxr);
  for (int i = 0; i < 3; ++i) tmp[i] = k.v[b][i] + wxr[i];
  matTvec3<S>(m.frame_R[f], tmp, fk.v);
  matTvec3<S>(m.frame_R[f], k.v[b] + 3, fk.v + 3);
  cross3<S>(k.a[b] + 3, m.frame_p[f], wxr);
  for (int i = 0; i < 3; ++i) tmp[i] = k.a[b][i] + wxr[i];
  matTvec3<S>(m.frame_R[f], tmp, fk.a);
  matTvec3<S>(m.frame_R[f], k.a[b] + 3, fk.a + 3);
}

// Build the dual-number seeds for the 2*nv directions [dq (right perturbation); dv] at state (q, v).
struct DualState {
  Dual R0[9], p0[3], cs[NB], sn[NB], v[NV];
};
inline void seed_dual_state(const EmpcModelDesc& m, const double* q, const double* v, DualState& ds) {
  const int nv = m.nv;
  double R[9];
  quat_to_R(q + 3, R);
  for (int i = 0; i < 9; ++i) ds.R0[i] = Dual(R[i]);
  for (int i = 0; i < 3; ++i) ds.p0[i] = Dual(q[i]);
  // linear base directions k=0..2: dp = R e_k ; angular k=3..5: dR = R [e_k]x
  for (int k = 0; k < 3; ++k) {
    for (int i = 0; i < 3; ++i) ds.p0[i].d[k] = R[3 * i + k];
    double e[3] = {0, 0, 0};
    e[k] = 1;
    double E[9], RE[9];
    skew3(e, E);
    matmul3<double>(R, E, RE);
    for (int i = 0; i < 9; ++i) ds.R0[i].d[3 + k] = RE[i];
  }
  for (int b = 1; b < m.nbodies; ++b) {
    const double th = q[7 + b - 1];
    ds.cs[b - 1] = Dual(std::cos(th));
    ds.sn[b - 1] = Dual(std::sin(th));
    ds.cs[b - 1].d[6 + b - 1] = -std::sin(th);
    ds.sn[b - 1].d[6 + b - 1] = std::cos(th);
  }
  for (int i = 0; i < nv; ++i) {
    ds.v[i] = Dual(v[i]);
    ds.v[i].d[nv + i] = 1.0;
  }
}
inline void plain_state(const EmpcModelDesc& m, const double* q, double* R0, double* cs, double* sn) {
  quat_to_R(q + 3, R0);
  for (int b = 1; b < m.nbodies; ++b) {
    cs[b - 1] = std::cos(q[7 + b - 1]);
    sn[b - 1] = std::sin(q[7 + b - 1]);
  }
}

}  // namespace oracle
