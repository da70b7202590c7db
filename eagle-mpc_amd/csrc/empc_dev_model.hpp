// empc_dev_model.hpp -- device-side problem description and the per-lane rigid-body / cost routines.
//
// Replaces, for the hot path, what the reference obtains from Crocoddyl + Pinocchio:
//   DifferentialActionModel{Free,Contact}FwdDynamics::calc   (src/factory/diff-action.cpp:31,34)
//   IntegratedActionModelEuler::calc                          (src/factory/int-action.cpp:26)
//   ActuationSquashingModel / SquashingModelSmoothSat         (src/trajectory.cpp:48-52)
//   CostModelSum of CostModelResidual(...)                    (src/factory/cost.cpp:38-168)
// Semantics: SURVEY.md Appendix A.3-A.7.  The kinematic tree must be a serial chain (body i's parent is body i-1),
// which holds for every multicopter + arm model the reference ships problems for.
#pragma once
#include "../../include/empc_types.h"
#include "empc_dev_math.hpp"
#include <type_traits>
#include "baked/empc_baked_models.hpp"

namespace empc {

// Device copies are the host structs themselves (plain data); only the cost-set table and knot table are separate
// device arrays.  Costs carry their names, unused on the device.
struct DevProblem {
  EmpcModelDesc model;
  int nx, ndx, nu, n_rotors, T, n_sets, has_contact, use_squash;
  int integrator, reserved_;  // EmpcIntegrator
  double dt;
  double tau_f[6 * EMPC_MAX_ROTORS];
  double u_lb[EMPC_MAX_NU];
  double u_ub[EMPC_MAX_NU];
  EmpcSolverParams prm;
};

// Model constants of a kernel instantiation: RuntimeModel reads the robot from the problem image (constant address space,
// scalar loads) -- any serial-chain robot of the class; a baked model (csrc/baked/) is a struct of constexpr tables with
// the same member names, so that the structural zeros / ones of a known robot fold at compile time.
struct RuntimeModel {};
template <int NB_, int NROT_, class MODEL_ = RuntimeModel>
struct Dims {
  typedef MODEL_ Model;
  static constexpr int NB = NB_;
  static constexpr int NROT = NROT_;
  static constexpr int NJ = NB_ - 1;
  static constexpr int NV = 6 + NJ;
  static constexpr int NQ = 7 + NJ;
  static constexpr int NX = NQ + NV;
  static constexpr int NDX = 2 * NV;
  static constexpr int NU = NROT_ + NJ;
  static constexpr int NTRI = NV * (NV + 1) / 2;
  static constexpr int NACC = NV + 6;  // stride of the per-node stash: generalized acceleration | contact force
  // tape record of one node (doubles); layout shared by linearize (writer) and backward (reader):
  //   A  = [Fx Fu]    n x (n+m) row-major  (so that row k of the dynamics Jacobian is one contiguous LDS row)
  //   HX = [Lxx Lxu]  n x (n+m) row-major
  //   LUU m x m | LX n | LU m | GAP n (fs[t]) | COST 1
  static constexpr int NM = NDX + NU;
  static constexpr int OFF_A = 0;
  static constexpr int OFF_FX = OFF_A;                   // leading dimension NM
  static constexpr int OFF_FU = OFF_A + NDX;             // leading dimension NM
  static constexpr int OFF_HX = OFF_A + NDX * NM;
  // the layout every record has outside the device (C ABI empc_tape_layout / empc_solver_get_tape, emulator API) and, with
  // EMPC_REC_TRI off, on the device as well: [Lxx Lxu] n x (n+m) row-major | Luu m x m
  static constexpr int FULL_LXX = OFF_HX, FULL_LXU = OFF_HX + NDX, FULL_LUU = OFF_HX + NDX * NM, FULL_LX = FULL_LUU + NU * NU;
  static constexpr int FULL_LU = FULL_LX + NDX, FULL_GAP = FULL_LU + NU, FULL_COST = FULL_GAP + NDX;
  static constexpr int REC_FULL = (FULL_COST + 1 + 15) / 16 * 16;
#if EMPC_REC_TRI
  // upper triangles: row i of the Hessian block is Lxx(i, i .. n-1) | Lxu(i, 0 .. m-1); Luu row i is Luu(i, i .. m-1)
  static constexpr int hxrow(int i) { return OFF_HX + i * NM - i * (i - 1) / 2; }
  static constexpr int lxx(int i, int j) { return (i <= j) ? hxrow(i) + (j - i) : hxrow(j) + (i - j); }
  static constexpr int lxu(int i, int k) { return hxrow(i) + (NDX - i) + k; }
  static constexpr int OFF_LUU = OFF_HX + NDX * (NDX + 1) / 2 + NDX * NU;
  static constexpr int luu(int i, int k) { return (i <= k) ? OFF_LUU + i * NU - i * (i - 1) / 2 + (k - i) : OFF_LUU + k * NU - k * (k - 1) / 2 + (i - k); }
  static constexpr int OFF_LX = OFF_LUU + NU * (NU + 1) / 2;  // ndx
  static constexpr bool stored_xx(int i, int j) { return i <= j; }  // the writer of column j stores rows i <= j only
#else
  static constexpr int OFF_LXX = OFF_HX;                 // leading dimension NM
  static constexpr int OFF_LXU = OFF_HX + NDX;           // leading dimension NM
  static constexpr int OFF_LUU = OFF_HX + NDX * NM;      // leading dimension NU
  static constexpr int lxx(int i, int j) { return OFF_LXX + i * NM + j; }
  static constexpr int lxu(int i, int k) { return OFF_LXU + i * NM + k; }
  static constexpr int luu(int i, int k) { return OFF_LUU + i * NU + k; }
  static constexpr int OFF_LX = OFF_LUU + NU * NU;       // ndx
  static constexpr bool stored_xx(int, int) { return true; }
#endif
  static constexpr int OFF_LU = OFF_LX + NDX;            // nu
  static constexpr int OFF_GAP = OFF_LU + NU;            // ndx   fs[t]
  static constexpr int OFF_COST = OFF_GAP + NDX;         // 1
  static constexpr int REC_RAW = OFF_COST + 1;
  static constexpr int REC = (REC_RAW + 15) / 16 * 16;   // padded to 128 B
};

// one record from the device layout to the full layout (what it is outside the device); the same copy when EMPC_REC_TRI is off
template <class DM>
inline void unpack_record(const double* dev, double* full) {
  for (int i = 0; i < DM::REC_FULL; ++i) full[i] = 0.0;
  for (int i = 0; i < DM::NDX * DM::NM; ++i) full[DM::OFF_A + i] = dev[DM::OFF_A + i];
  for (int i = 0; i < DM::NDX; ++i) {
    for (int j = 0; j < DM::NDX; ++j) full[DM::FULL_LXX + i * DM::NM + j] = dev[DM::lxx(i, j)];
    for (int k = 0; k < DM::NU; ++k) full[DM::FULL_LXU + i * DM::NM + k] = dev[DM::lxu(i, k)];
  }
  for (int i = 0; i < DM::NU; ++i)
    for (int k = 0; k < DM::NU; ++k) full[DM::FULL_LUU + i * DM::NU + k] = dev[DM::luu(i, k)];
  for (int i = 0; i < DM::NDX; ++i) full[DM::FULL_LX + i] = dev[DM::OFF_LX + i];
  for (int i = 0; i < DM::NU; ++i) full[DM::FULL_LU + i] = dev[DM::OFF_LU + i];
  for (int i = 0; i < DM::NDX; ++i) full[DM::FULL_GAP + i] = dev[DM::OFF_GAP + i];
  full[DM::FULL_COST] = dev[DM::OFF_COST];
}

// view of a baked model: the tree constants are the static members of B; the operational frames a problem selects (cost
// and contact frames, EmpcModelDesc::frame_*) stay in the problem image
struct BakedView {
  const double (&jplace_R)[8][9];
  const double (&jplace_p)[8][3];
  const double (&axis)[8][3];
  const double (&mass)[8];
  const double (&com)[8][3];
  const double (&inertia)[8][9];
  const double (&gravity)[3];
  const EMPC_K int32_t (&frame_body)[EMPC_MAX_FRAMES];
  const EMPC_K double (&frame_R)[EMPC_MAX_FRAMES][9];
  const EMPC_K double (&frame_p)[EMPC_MAX_FRAMES][3];
};
template <class DM>
EMPC_HD decltype(auto) model_of(const EMPC_K DevProblem& P) {
  if constexpr (std::is_same<typename DM::Model, RuntimeModel>::value)
    return (P.model);
  else
  {
    constexpr const BakedTree& t = DM::Model::tree();
    return BakedView{t.jplace_R, t.jplace_p, t.axis, t.mass, t.com, t.inertia, t.gravity, P.model.frame_body, P.model.frame_R, P.model.frame_p};
  }
}

// platform constants (MultiCopterBaseParams: actuation matrix, control limits) of a kernel instantiation: from the problem
// image, or literals of the baked robot
struct RuntimePlatform {
  const EMPC_K double (&tau_f)[6 * EMPC_MAX_ROTORS];
  const EMPC_K double (&u_lb)[EMPC_MAX_NU];
  const EMPC_K double (&u_ub)[EMPC_MAX_NU];
};
struct BakedPlatform {
  const double (&tau_f)[48];
  const double (&u_lb)[16];
  const double (&u_ub)[16];
};
static_assert(6 * EMPC_MAX_ROTORS == 48 && EMPC_MAX_NU == 16 && EMPC_MAX_BODIES == 8, "layout of BakedTree");
template <class DM>
EMPC_HD auto platform_of(const EMPC_K DevProblem& P) {
  if constexpr (std::is_same<typename DM::Model, RuntimeModel>::value)
    return RuntimePlatform{P.tau_f, P.u_lb, P.u_ub};
  else {
    constexpr const BakedTree& t = DM::Model::tree();
    return BakedPlatform{t.tau_f, t.u_lb, t.u_ub};
  }
}

// Rodrigues rotation about a unit axis from cos/sin
template <class S>
EMPC_HD void axis_rot(const double* ax, const S& c, const S& s, S* R) {
  S omc = 1.0 - c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
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

// spatial inertia of body b applied to a motion (body frame): [m (v + w x c); Ic w + c x m (v + w x c)]
template <class S, class MT>
EMPC_HD void inertia_apply(const MT& m, int b, const S* mot, S* out) {
  S wxc[3], lin[3], Iw[3], cxl[3];
  cross3<S>(mot + 3, m.com[b], wxc);
#pragma unroll
  for (int i = 0; i < 3; ++i) lin[i] = m.mass[b] * (mot[i] + wxc[i]);
  matvec3<S>(m.inertia[b], mot + 3, Iw);
  cross3<S>(m.com[b], lin, cxl);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[i] = lin[i];
    out[3 + i] = Iw[i] + cxl[i];
  }
}

// Operational-frame capture: placement and LOCAL velocity / acceleration of up to NCAP frames, filled during the
// forward recursion when the recursion reaches the frame's body.
// (2 in the product: every shipped stage names at most two distinct frames.  -DEMPC_NCAP=3 builds a library whose kernels capture
//  three -- e.g. two contact frames AND a frame cost on a third link -- at ~36 more registers per lane in the full linearize body;
//  emulator-verified, tests/test_two_contacts_emulator.py::test_three_frames_need_a_build_with_three_captures)
#ifndef EMPC_NCAP
#define EMPC_NCAP 2
#endif
constexpr int NCAP = EMPC_NCAP;
static_assert(NCAP >= 2 && NCAP <= 4, "capture slots: 2 (product) to 4");
// contact rows of a kernel instantiation (template parameter CT / NC): 0 free dynamics, 3 ContactModel3D, 6 ContactModel6D,
// CT_MIXED a problem whose stages use both (the bodies of 3 and 6 behind a uniform branch on the node's contact type)
constexpr int CT_MIXED = 9;
// CT_PAIR3 a problem with a stage whose ContactModelMultiple holds TWO ContactModel3D contacts (src/stage.cpp:38-48 adds every
// name of the stage's list; crocoddyl stacks their rows in the order of its name-sorted map = the order of set.contacts[]): six
// stacked rows, contact 0 in rows 0-2, contact 1 in rows 3-5, one KKT system.  Stages of the same problem with ONE ContactModel3D
// run the 3-row body behind a uniform branch on the node's contact count.  Opt-in (EMPC_EXPERIMENTAL_CONTACT), never run on a GPU.
constexpr int CT_PAIR3 = 33;
// constraint rows a kernel instantiation reserves room for
constexpr int ct_rows(int CT) { return (CT == CT_MIXED || CT == CT_PAIR3) ? 6 : CT; }
template <class S>
struct FrameCap {
  S R[9], p[3];  // world placement
  S v[6];        // LOCAL spatial velocity
  S a[6];        // LOCAL spatial acceleration (of the recursion's a, i.e. including the gravity offset if enabled)
};
template <class S, class MT>
EMPC_HD void frame_capture(const MT& m, int f, const S* Rb, const S* pb, const S* vb, const S* ab,
                           FrameCap<S>& fk) {
  matmul3<S>(Rb, m.frame_R[f], fk.R);
  S Rp[3];
  matvec3<S>(Rb, m.frame_p[f], Rp);
#pragma unroll
  for (int i = 0; i < 3; ++i) fk.p[i] = pb[i] + Rp[i];
  S wxr[3], tmp[3];
  cross3<S>(vb + 3, m.frame_p[f], wxr);
#pragma unroll
  for (int i = 0; i < 3; ++i) tmp[i] = vb[i] + wxr[i];
  matTvec3<S>(m.frame_R[f], tmp, fk.v);
  matTvec3<S>(m.frame_R[f], vb + 3, fk.v + 3);
  cross3<S>(ab + 3, m.frame_p[f], wxr);
#pragma unroll
  for (int i = 0; i < 3; ++i) tmp[i] = ab[i] + wxr[i];
  matTvec3<S>(m.frame_R[f], tmp, fk.a);
  matTvec3<S>(m.frame_R[f], ab + 3, fk.a + 3);
}

// Recursive Newton-Euler on a serial chain, body-frame Featherstone form, generic in the scalar (double or D1).
//   R0,p0     base placement; cs,sn cos/sin of the joint angles; v,a generalized velocity / acceleration
//   gravity   fold -g into the base acceleration
//   fext_b/fext: optional external spatial force (body frame) acting on body fext_b
//   cap_frames[ncap]: operational frames to capture (indices into the model's frame table)
// Register discipline: only the per-body forces survive the forward sweep; joint rotations are rebuilt from cs/sn
// in the backward sweep.
#if EMPC_ROLL_CAP_LDS
// (EMPC_ROLL_CAP_LDS: the recursion hands a captured frame to `capture(slot, frame, Rw, pw, vb, ab)`; the classic signature below
//  stores it in caps[slot])
template <int NB, class S, class MT, class CapF>
EMPC_HD void rnea_chain_f(const MT& m, const S* R0, const S* p0, const S* cs, const S* sn, const S* v,
                          const S* a, bool gravity, int fext_b, const S* fext, S* tau, int ncap, const int* cap_frames,
                          CapF&& capture) {
#else
template <int NB, class S, class MT>
EMPC_HD void rnea_chain(const MT& m, const S* R0, const S* p0, const S* cs, const S* sn, const S* v,
                        const S* a, bool gravity, int fext_b, const S* fext, S* tau, int ncap, const int* cap_frames,
                        FrameCap<S>* caps) {
#endif
  S f[NB][6];
  S Rw[9], pw[3], vb[6], ab[6];  // rolling: world placement, body-frame velocity / acceleration of the current body
#pragma unroll
  for (int i = 0; i < 9; ++i) Rw[i] = R0[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) pw[i] = p0[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    vb[i] = v[i];
    ab[i] = a[i];
  }
  if (gravity) {
    double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
    S gl[3];
    matTvec3<S>(R0, ng, gl);
#pragma unroll
    for (int i = 0; i < 3; ++i) ab[i] += gl[i];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b > 0) {
      S Rj[9], XR[9];
      axis_rot<S>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
      matmul3<S>(m.jplace_R[b], Rj, XR);
      const double* r = m.jplace_p[b];
      S Rr[3], Rn[9];
      matvec3<S>(Rw, r, Rr);
      matmul3<S>(Rw, XR, Rn);
#pragma unroll
      for (int i = 0; i < 9; ++i) Rw[i] = Rn[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) pw[i] += Rr[i];
      // v_c = E (v_p + w_p x r), w_c = E w_p, E = XR^T ; + S qd
      S wxr[3], tmp[3], vn[6], an[6];
      cross3<S>(vb + 3, r, wxr);
#pragma unroll
      for (int i = 0; i < 3; ++i) tmp[i] = vb[i] + wxr[i];
      matTvec3<S>(XR, tmp, vn);
      matTvec3<S>(XR, vb + 3, vn + 3);
      const S& qd = v[6 + b - 1];
      S sv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) sv[i] = m.axis[b][i] * qd;
#pragma unroll
      for (int i = 0; i < 3; ++i) vn[3 + i] += sv[i];
      cross3<S>(ab + 3, r, wxr);
#pragma unroll
      for (int i = 0; i < 3; ++i) tmp[i] = ab[i] + wxr[i];
      matTvec3<S>(XR, tmp, an);
      matTvec3<S>(XR, ab + 3, an + 3);
      const S& qdd = a[6 + b - 1];
      S c1[3], c2[3];
      cross3<S>(vn, sv, c1);
      cross3<S>(vn + 3, sv, c2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        an[i] += c1[i];
        an[3 + i] += m.axis[b][i] * qdd + c2[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        vb[i] = vn[i];
        ab[i] = an[i];
      }
    }
    // frame captures on this body
#pragma unroll
    for (int c = 0; c < NCAP; ++c)
#if EMPC_ROLL_CAP_LDS
      if (c < ncap && m.frame_body[cap_frames[c]] == b) capture(c, cap_frames[c], Rw, pw, vb, ab);
#else
      if (c < ncap && m.frame_body[cap_frames[c]] == b) frame_capture<S>(m, cap_frames[c], Rw, pw, vb, ab, caps[c]);
#endif
    // f_b = I a + v x* (I v) - fext
    S Ia[6], Iv[6], c1[3], c2[3], c3[3];
    inertia_apply<S>(m, b, ab, Ia);
    inertia_apply<S>(m, b, vb, Iv);
    cross3<S>(vb + 3, Iv, c1);
    cross3<S>(vb + 3, Iv + 3, c2);
    cross3<S>(vb, Iv, c3);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f[b][i] = Ia[i] + c1[i];
      f[b][3 + i] = Ia[3 + i] + c2[i] + c3[i];
    }
    if (fext != nullptr && fext_b == b) {
#pragma unroll
      for (int i = 0; i < 6; ++i) f[b][i] -= fext[i];
    }
  }
#pragma unroll
  for (int b = NB - 1; b >= 1; --b) {
    tau[6 + b - 1] = dot3<S>(m.axis[b], f[b] + 3);
    S Rj[9], XR[9];
    axis_rot<S>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
    matmul3<S>(m.jplace_R[b], Rj, XR);
    S fl[3], fn[3], rxf[3];
    matvec3<S>(XR, f[b], fl);
    matvec3<S>(XR, f[b] + 3, fn);
    cross3<S>(m.jplace_p[b], fl, rxf);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f[b - 1][i] += fl[i];
      f[b - 1][3 + i] += fn[i] + rxf[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) tau[i] = f[0][i];
}
#if EMPC_ROLL_CAP_LDS
template <int NB, class S, class MT>
EMPC_HD void rnea_chain(const MT& m, const S* R0, const S* p0, const S* cs, const S* sn, const S* v,
                        const S* a, bool gravity, int fext_b, const S* fext, S* tau, int ncap, const int* cap_frames,
                        FrameCap<S>* caps) {
  rnea_chain_f<NB, S>(m, R0, p0, cs, sn, v, a, gravity, fext_b, fext, tau, ncap, cap_frames,
                      [&](int c, int f, const S* Rw, const S* pw, const S* vb, const S* ab) { frame_capture<S>(m, f, Rw, pw, vb, ab, caps[c]); });
}
#endif

// Composite-rigid-body algorithm on a serial chain; output: packed lower triangle of M (idx(i,j) = i(i+1)/2 + j).
// Composite inertias are carried as (mass, COM, rotational inertia about the COM) in body axes.
template <int NB, class MT>
EMPC_HD void crba_chain(const MT& m, const double* cs, const double* sn, double* Mp) {
  constexpr int NV = 6 + NB - 1;
  double XR[NB][9];
  double cm[NB], cc[NB][3], cI[NB][9];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    cm[b] = m.mass[b];
#pragma unroll
    for (int i = 0; i < 3; ++i) cc[b][i] = m.com[b][i];
#pragma unroll
    for (int i = 0; i < 9; ++i) cI[b][i] = m.inertia[b][i];
    if (b > 0) {
      double Rj[9];
      axis_rot<double>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
      matmul3<double>(m.jplace_R[b], Rj, XR[b]);
    }
  }
#pragma unroll
  for (int b = NB - 1; b >= 1; --b) {
    // F = Ic_b * [0; axis_b]
    double F[6];
    {
      double wxc[3], Iw[3], cxl[3];
      cross3<double>(m.axis[b], cc[b], wxc);
      matvec3<double>(cI[b], m.axis[b], Iw);
#pragma unroll
      for (int i = 0; i < 3; ++i) F[i] = cm[b] * wxc[i];
      cross3<double>(cc[b], F, cxl);
#pragma unroll
      for (int i = 0; i < 3; ++i) F[3 + i] = Iw[i] + cxl[i];
    }
    const int col = 6 + b - 1;
    Mp[col * (col + 1) / 2 + col] = dot3<double>(m.axis[b], F + 3);
#pragma unroll
    for (int j = b; j >= 1; --j) {
      double fl[3], fn[3], rxf[3];
      matvec3<double>(XR[j], F, fl);
      matvec3<double>(XR[j], F + 3, fn);
      cross3<double>(m.jplace_p[j], fl, rxf);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        F[i] = fl[i];
        F[3 + i] = fn[i] + rxf[i];
      }
      if (j - 1 >= 1) {
        const int row = 6 + (j - 1) - 1;
        Mp[col * (col + 1) / 2 + row] = dot3<double>(m.axis[j - 1], F + 3);
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) Mp[col * (col + 1) / 2 + i] = F[i];
      }
    }
    // add the composite of body b to its parent (expressed in the parent's axes)
    double c2[3], RI[9], I2[9];
    matvec3<double>(XR[b], cc[b], c2);
#pragma unroll
    for (int i = 0; i < 3; ++i) c2[i] += m.jplace_p[b][i];
    matmul3<double>(XR[b], cI[b], RI);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        I2[3 * i + j] = RI[3 * i] * XR[b][3 * j] + RI[3 * i + 1] * XR[b][3 * j + 1] + RI[3 * i + 2] * XR[b][3 * j + 2];
    const double m1 = cm[b - 1], m2 = cm[b], Mt = m1 + m2, iMt = frcp(Mt);
    double cn[3], d1[3], d2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      cn[i] = (m1 * cc[b - 1][i] + m2 * c2[i]) * iMt;
      d1[i] = cc[b - 1][i] - cn[i];
      d2[i] = c2[i] - cn[i];
    }
    const double dd1 = dot3<double>(d1, d1), dd2 = dot3<double>(d2, d2);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        cI[b - 1][3 * i + j] += I2[3 * i + j] + m1 * ((i == j ? dd1 : 0.0) - d1[i] * d1[j]) +
                                m2 * ((i == j ? dd2 : 0.0) - d2[i] * d2[j]);
    cm[b - 1] = Mt;
#pragma unroll
    for (int i = 0; i < 3; ++i) cc[b - 1][i] = cn[i];
  }
  // base 6x6 block: [[m I, -m [c]x],[m [c]x, Ic - m [c]x^2]]
  {
    double C[9], CC[9];
    skew3(cc[0], C);
    matmul3<double>(C, C, CC);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Mp[i * (i + 1) / 2 + j] = (i == j) ? cm[0] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Mp[(3 + i) * (4 + i) / 2 + j] = cm[0] * C[3 * i + j];
#pragma unroll
      for (int j = 0; j <= i; ++j) Mp[(3 + i) * (4 + i) / 2 + 3 + j] = cI[0][3 * i + j] - cm[0] * CC[3 * i + j];
    }
  }
}

// In-register Cholesky of a packed lower-triangular NV x NV matrix (row-major packed: idx(i,j) = i(i+1)/2 + j).
// On exit L holds the factor with RECIPROCAL diagonal entries. Returns false if not positive definite.
template <int N>
EMPC_HD bool chol_packed(double* L) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double s = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    if (!(s > 0.0) || is_nan(s)) ok = false;
    const double inv = frsqrt(s);
    L[j * (j + 1) / 2 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      L[i * (i + 1) / 2 + j] = t * inv;
    }
  }
  return ok;
}
// Cholesky of the CONSTRAINT matrix G = Jc M^-1 Jc^T of a two-contact stage (CT_PAIR3), which can be rank deficient: any two
// points of a stretched chain sit on one line, their constraint rows along that line coincide -- and the solver's default initial
// guess (every knot at the zero state) is exactly that configuration.  pinocchio::forwardDynamics factors G with Eigen's LLT,
// which STOPS at the first non-positive pivot and leaves the rest of the matrix as it was (llt_inplace::unblocked: `if (x <= 0)
// return k`); the triangular solves that follow use that half-factored matrix -- the redundant row gets a multiplier of ~0 and the
// dynamics come out sane.  The oracle's cholesky() does the same.  This is that behaviour in straight-line code (selects, no
// branches): from the first failed pivot on, columns keep their entries (diagonal stored as reciprocal like everywhere else).
// A pivot below 1e-13 of its diagonal entry is rounding noise of a rank-deficient system (its sign is an accident of the
// summation order, which differs between the nominal pass, linearize and the oracle): treated like a non-positive one, so that
// every pass of the device makes the same choice.  chol_packed (frsqrt of whatever the pivot is) stays what every other
// factorisation uses.
template <int N>
EMPC_HD bool chol_packed_stop(double* L) {
  bool stopped = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double raw = L[j * (j + 1) / 2 + j];
    double s = raw;
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    const bool bad = stopped || !(s > 1e-13 * raw) || is_nan(s);
    stopped = bad;
    const double inv = bad ? frcp(raw) : frsqrt(s);
    L[j * (j + 1) / 2 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      L[i * (i + 1) / 2 + j] = bad ? L[i * (i + 1) / 2 + j] : t * inv;
    }
  }
  return !stopped;
}
template <int N>
EMPC_HD void chol_solve_packed(const double* L, double* b) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * (i + 1) / 2 + k] * b[k];
    b[i] = s * L[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k * (k + 1) / 2 + i] * b[k];
    b[i] = s * L[i * (i + 1) / 2 + i];
  }
}

// SquashingModelSmoothSat: u = sigma(s), du = sigma'(s)
EMPC_HD void squash1(double s, double lb, double ub, double smooth, int power, double& u, double& du) {
  const double dd = smooth * (ub - lb);
  const double a = (power == 4) ? dd * dd * dd * dd : dd * dd;
  const double al = (s - lb) * (s - lb) + a, au = (s - ub) * (s - ub) + a;
  const double rl = frsqrt(al), ru = frsqrt(au);
  const double sl = al * rl, su = au * ru;
  u = 0.5 * (sl - su + ub + lb);
  du = 0.5 * ((s - lb) * rl - (s - ub) * ru);
}

// activation value and derivatives of one residual component
EMPC_HD void activation1(int act, double r, double w, double lb, double ub, double& a, double& Ar, double& Arr) {
  if (act == EMPC_ACT_QUAD) {
    a = 0.5 * r * r;
    Ar = r;
    Arr = 1.0;
  } else if (act == EMPC_ACT_WEIGHTED_QUAD) {
    a = 0.5 * w * r * r;
    Ar = w * r;
    Arr = w;
  } else {
    const double ww = (act == EMPC_ACT_QUADRATIC_BARRIER) ? 1.0 : w;
    const double lo = fmin(r - lb, 0.0);
    const double hi = fmax(r - ub, 0.0);
    a = 0.5 * ww * lo * lo + 0.5 * ww * hi * hi;
    Ar = ww * (lo + hi);
    const double ind = ((r - lb <= 0.0) ? 1.0 : 0.0) + ((r - ub >= 0.0) ? 1.0 : 0.0);
    Arr = ww * ind;
  }
}
// activation1 without branches (selects only): for lanes that evaluate DIFFERENT costs side by side, where a branch on
// the activation type would serialise the lanes.  Same formulas, bit for bit (the neutral weight is an exact 1.0).
EMPC_HD void activation_sel(int act, double r, double w, double lb, double ub, double& a, double& Ar, double& Arr) {
  const bool barrier = (act == EMPC_ACT_QUADRATIC_BARRIER || act == EMPC_ACT_WEIGHTED_QUADRATIC_BARRIER);
  const double ww = (act == EMPC_ACT_QUAD || act == EMPC_ACT_QUADRATIC_BARRIER) ? 1.0 : w;
  const double lo = fmin(r - lb, 0.0);
  const double hi = fmax(r - ub, 0.0);
  const double ind = ((r - lb <= 0.0) ? 1.0 : 0.0) + ((r - ub >= 0.0) ? 1.0 : 0.0);
  const double aq = (act == EMPC_ACT_QUAD) ? 0.5 * r * r : 0.5 * ww * r * r;
  a = barrier ? 0.5 * ww * lo * lo + 0.5 * ww * hi * hi : aq;
  Ar = barrier ? ww * (lo + hi) : ((act == EMPC_ACT_QUAD) ? r : ww * r);
  Arr = barrier ? ww * ind : ww;
}
// weight of component i of cost c, with the barrier cost's weights derived from the trajectory's current smooth
// (SolverSbFDDP::barrierUpdate, src/sbfddp.cpp:464-477)
template <class CostT, class PlatT>
EMPC_HD double act_weight(const CostT& c, int i, double smooth, const PlatT& P) {
  if (c.is_barrier) {
    const double aux = smooth * (P.u_ub[i] - P.u_lb[i]);
    return frcp(aux * aux);
  }
  return c.act_w[i];
}

// prepare_problem marks the cost sets that capture operational frames (active frame costs or a contact) in bit 0 of
// costs[0].reserved; linearize runs a lean body for the others
template <class SetT>
EMPC_HD bool set_uses_frames(const SetT& set) {
  return set.ncosts > 0 && (set.costs[0].reserved & 1) != 0;
}

// Sum of the activation values of the first nr (<= NR) residual components -- the value-only path of the rollouts.
// The activation type is branched on ONCE and each case is straight-line code, so the parameter loads (weights, bounds)
// of all components are issued together instead of one dependent scalar load per component (measured: 1/3 of the
// rollout kernel, profiles/r01_rollout_ablation.txt).  Same formulas and summation order as activation1.
template <int NR, class CostT>
EMPC_HD double activation_value(const CostT& c, const double* r, int nr) {
  double cval = 0;
  if (c.activation == EMPC_ACT_QUAD) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i < nr) cval += 0.5 * r[i] * r[i];
  } else if (c.activation == EMPC_ACT_WEIGHTED_QUAD) {
    double w[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) w[i] = c.act_w[i];
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i < nr) cval += 0.5 * w[i] * r[i] * r[i];
  } else {
    const bool weighted = (c.activation != EMPC_ACT_QUADRATIC_BARRIER);
    double w[NR], lb[NR], ub[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      w[i] = weighted ? c.act_w[i] : 1.0;
      lb[i] = c.lb[i];
      ub[i] = c.ub[i];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i < nr) {
        const double lo = fmin(r[i] - lb[i], 0.0);
        const double hi = fmax(r[i] - ub[i], 0.0);
        cval += 0.5 * w[i] * lo * lo + 0.5 * w[i] * hi * hi;
      }
    }
  }
  return cval;
}
// Control cost value: residual s - ref with the barrier cost's weights derived from the current smoothness
template <int NU, class CostT, class PlatT>
EMPC_HD double control_cost_value(const CostT& c, const double* s, double smooth, const PlatT& P) {
  double r[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) r[i] = s[i] - c.ref[i];
  if (!c.is_barrier) return activation_value<NU>(c, r, NU);
  // SolverSbFDDP::barrierUpdate (src/sbfddp.cpp:464-477): weights 1 / (smooth (ub - lb))^2, bounds from the cost
  double cval = 0;
  double w[NU], lb[NU], ub[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const double aux = smooth * (P.u_ub[i] - P.u_lb[i]);
    w[i] = frcp(aux * aux);
    lb[i] = c.lb[i];
    ub[i] = c.ub[i];
  }
  if (c.activation == EMPC_ACT_QUAD || c.activation == EMPC_ACT_WEIGHTED_QUAD) {
    const bool weighted = (c.activation == EMPC_ACT_WEIGHTED_QUAD);
#pragma unroll
    for (int i = 0; i < NU; ++i) cval += weighted ? 0.5 * w[i] * r[i] * r[i] : 0.5 * r[i] * r[i];
  } else {
    const bool weighted = (c.activation != EMPC_ACT_QUADRATIC_BARRIER);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const double ww = weighted ? w[i] : 1.0;
      const double lo = fmin(r[i] - lb[i], 0.0);
      const double hi = fmax(r[i] - ub[i], 0.0);
      cval += 0.5 * ww * lo * lo + 0.5 * ww * hi * hi;
    }
  }
  return cval;
}

// StateMultibody::diff(x0, x1) for the free-flyer + joints layout; also returns the translation part of
// M0^-1 M1 (needed by Jlog6)
template <class DM, class X0, class X1>
EMPC_HD void state_diff(const X0* x0, const X1* x1, double* dx, double* dpl_out) {
  double qc[4] = {-x0[3], -x0[4], -x0[5], x0[6]};
  double q0[4] = {x0[3], x0[4], x0[5], x0[6]};
  double q1[4] = {x1[3], x1[4], x1[5], x1[6]};
  double qd[4], R0[9], dp[3], dpl[3];
  quat_mul(qc, q1, qd);
  quat_normalize(qd);
#pragma unroll
  for (int i = 0; i < 3; ++i) dp[i] = x1[i] - x0[i];
  quat_to_R(q0, R0);
  matTvec3<double>(R0, dp, dpl);
  log6_quat(qd, dpl, dx);
#pragma unroll
  for (int i = 7; i < DM::NQ; ++i) dx[i - 1] = x1[i] - x0[i];
#pragma unroll
  for (int i = 0; i < DM::NV; ++i) dx[DM::NV + i] = x1[DM::NQ + i] - x0[DM::NQ + i];
  if (dpl_out) {
    dpl_out[0] = dpl[0];
    dpl_out[1] = dpl[1];
    dpl_out[2] = dpl[2];
  }
}
// StateMultibody::integrate(x, dx); optionally returns the translation of exp6(dx[0:6]) (for Jexp6)
template <class DM, class X0>
EMPC_HD void state_integrate(const X0* x, const double* dx, double* xout, double* pe_out) {
  double qe[4], pe[3], R0[9], Rp[3], qn[4];
  double q0[4] = {x[3], x[4], x[5], x[6]};
  exp6_quat(dx, qe, pe);
  quat_to_R(q0, R0);
  matvec3<double>(R0, pe, Rp);
  quat_mul(q0, qe, qn);
  quat_normalize(qn);
#pragma unroll
  for (int i = 0; i < 3; ++i) xout[i] = x[i] + Rp[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) xout[3 + i] = qn[i];
#pragma unroll
  for (int i = 7; i < DM::NQ; ++i) xout[i] = x[i] + dx[i - 1];
#pragma unroll
  for (int i = 0; i < DM::NV; ++i) xout[DM::NQ + i] = x[DM::NQ + i] + dx[DM::NV + i];
  if (pe_out) {
    pe_out[0] = pe[0];
    pe_out[1] = pe[1];
    pe_out[2] = pe[2];
  }
}

// state_integrate with exp6 of the step's first six entries already taken (qe, pe = exp6_quat(dx)): the same operations in the same
// order, minus that call -- for a step that is known before the state is (the gap correction of a rollout, empc_rollout6.hpp)
template <class DM, class X0>
EMPC_HD void state_integrate_pre(const X0* x, const double* dx, const double* qe, const double* pe, double* xout) {
  double R0[9], Rp[3], qn[4];
  double q0[4] = {x[3], x[4], x[5], x[6]};
  quat_to_R(q0, R0);
  matvec3<double>(R0, pe, Rp);
  quat_mul(q0, qe, qn);
  quat_normalize(qn);
#pragma unroll
  for (int i = 0; i < 3; ++i) xout[i] = x[i] + Rp[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) xout[3 + i] = qn[i];
#pragma unroll
  for (int i = 7; i < DM::NQ; ++i) xout[i] = x[i] + dx[i - 1];
#pragma unroll
  for (int i = 0; i < DM::NV; ++i) xout[DM::NQ + i] = x[DM::NQ + i] + dx[DM::NV + i];
}

// Friction-cone matrix rows (FrictionCone(n, mu, 4, false)): row i of A R_n^T
EMPC_HD void cone_rows(const double* nsurf, double mu, double AR[5][3]) {
  const double A[5][3] = {{1, 0, -mu}, {0, 1, -mu}, {-1, 0, -mu}, {0, -1, -mu}, {0, 0, 1}};
  double nrm = sqrt(dot3<double>(nsurf, nsurf));
  double nz[3] = {nsurf[0] / nrm, nsurf[1] / nrm, nsurf[2] / nrm};
  double Rn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double e3[3] = {0, 0, 1};
  double ax[3];
  cross3<double>(e3, nz, ax);
  const double sn_ = sqrt(dot3<double>(ax, ax)), cs_ = nz[2];
  if (sn_ > 1e-12) {
    const double ang = atan2(sn_, cs_);
    double w[3] = {ax[0] / sn_, ax[1] / sn_, ax[2] / sn_};
    axis_rot<double>(w, cos(ang), sin(ang), Rn);
  } else if (cs_ < 0) {
    Rn[4] = -1;
    Rn[8] = -1;
  }
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 3; ++j) AR[i][j] = A[i][0] * Rn[3 * j] + A[i][1] * Rn[3 * j + 1] + A[i][2] * Rn[3 * j + 2];
}

// ContactModel3D / ContactModel6D forward dynamics on top of the free solution:
// [M Jc^T; Jc 0][a; -lam] = [tau - h; -a0].
// In: the Cholesky factor L of M (reciprocal diagonal), a = M^-1 (tau - h), the contact frame's capture ck (from the bias
// pass: placement, LOCAL velocity, LOCAL acceleration at qdd = 0 with the gravity offset).  Out: a corrected in place,
// lam[0..NC-1] = contact force (NC = 3: LOCAL linear force; NC = 6: LOCAL wrench [f; n]).  The number of constraint rows is
// fixed at compile time (one kernel instantiation per contact type): every loop unrolls and Jc / M^-1 Jc^T stay in registers.
//   3D: a0 = a_lin + w x v_lin + g0 (p_f - xref) + g1 v_lin          (classical acceleration of the contact point)
//   6D: a0 = a (spatial, LOCAL) + g0 log6(Mref^-1 oMf) + g1 v
template <class DM, int NC, class ContactT, class MT>
EMPC_HD void contact_forward(const MT& m, const ContactT& ct, const FrameCap<double>& ck, const double* R0,
                             const double* q, const double* cs, const double* sn, const double* L, double* a, double* lam) {
  if constexpr (NC == CT_MIXED) {
    // a problem with stages of both contact types: the type of this node's contact picks the body (uniform over the
    // lanes of a wavefront: they hold trajectories at the same knot)
    if (ct.type == EMPC_CONTACT_6D)
      contact_forward<DM, 6>(m, ct, ck, R0, q, cs, sn, L, a, lam);
    else
      contact_forward<DM, 3>(m, ct, ck, R0, q, cs, sn, L, a, lam);
    return;
  }
  constexpr int NV = DM::NV;
  constexpr int nc = (NC == CT_MIXED) ? 6 : NC;
  static_assert(NC == 3 || NC == 6 || NC == CT_MIXED, "ContactModel3D, ContactModel6D or both");
  // drift (frame acceleration at qdd = 0, no gravity): the bias pass carries gravity as a base acceleration -g, which
  // reaches every frame as the pure translation R_f^T (-g); take it out again
  double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]}, gf[3], a0[6];
  matTvec3<double>(ck.R, ng, gf);
  if constexpr (NC == 3) {
    double wxv[3];
    cross3<double>(ck.v + 3, ck.v, wxv);  // classical acceleration of the contact point
#pragma unroll
    for (int r = 0; r < 3; ++r) a0[r] = (ck.a[r] - gf[r]) + wxv[r];
    if (ct.gains[0] != 0.0)
#pragma unroll
      for (int r = 0; r < 3; ++r) a0[r] += ct.gains[0] * (ck.p[r] - ct.ref_p[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      a0[r] = ck.a[r] - gf[r];
      a0[3 + r] = ck.a[3 + r];
    }
    if (ct.gains[0] != 0.0) {
      double rR[9], dp[3], rp[3], qq[4], xi[6];
      matTmul3<double>(ct.ref_R, ck.R, rR);
#pragma unroll
      for (int r = 0; r < 3; ++r) dp[r] = ck.p[r] - ct.ref_p[r];
      matTvec3<double>(ct.ref_R, dp, rp);
      R_to_quat(rR, qq);
      log6_quat(qq, rp, xi);
#pragma unroll
      for (int r = 0; r < 6; ++r) a0[r] += ct.gains[0] * xi[r];
    }
  }
  if (ct.gains[1] != 0.0)
#pragma unroll
    for (int r = 0; r < nc; ++r) a0[r] += ct.gains[1] * ck.v[r];
  // Jc: LOCAL frame Jacobian from the kinematics (column j = [R_f^T (z_j x (p_f - o_j)); R_f^T z_j] for a rotation about
  // the world axis z_j through o_j, R_f^T e_j for the base translations), joints after the frame's body contribute nothing
  double Jc[nc][NV], MiJt[nc][NV];
  {
    const int bf = m.frame_body[ct.frame];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double w[3] = {R0[j], R0[3 + j], R0[6 + j]}, lin[3];
      matTvec3<double>(ck.R, w, lin);
#pragma unroll
      for (int r = 0; r < 3; ++r) Jc[r][j] = lin[r];
      if constexpr (NC == 6) Jc[3][j] = Jc[4][j] = Jc[5][j] = 0.0;
    }
    double Rw[9], pw[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rw[i] = R0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pw[i] = q[i];
#pragma unroll
    for (int j = 3; j < NV; ++j) {
      double z[3];
      bool on_path = true;
      if (j < 6) {
        z[0] = R0[j - 3];
        z[1] = R0[3 + j - 3];
        z[2] = R0[6 + j - 3];
      } else {
        const int b = j - 6 + 1;
        double Rj[9], XR[9], Rr[3], Rn[9];
        axis_rot<double>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
        matmul3<double>(m.jplace_R[b], Rj, XR);
        matvec3<double>(Rw, m.jplace_p[b], Rr);
        matmul3<double>(Rw, XR, Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rw[i] = Rn[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pw[i] += Rr[i];
        double ax[3] = {m.axis[b][0], m.axis[b][1], m.axis[b][2]};
        matvec3<double>(Rw, ax, z);
        on_path = (b <= bf);
      }
      double d[3] = {ck.p[0] - pw[0], ck.p[1] - pw[1], ck.p[2] - pw[2]}, zxd[3], lin[3];
      cross3<double>(z, d, zxd);
      matTvec3<double>(ck.R, zxd, lin);
#pragma unroll
      for (int r = 0; r < 3; ++r) Jc[r][j] = on_path ? lin[r] : 0.0;
      if constexpr (NC == 6) {
        double ang[3];
        matTvec3<double>(ck.R, z, ang);
#pragma unroll
        for (int r = 0; r < 3; ++r) Jc[3 + r][j] = on_path ? ang[r] : 0.0;
      }
    }
  }
  double G[nc * (nc + 1) / 2];  // packed nc x nc
#pragma unroll
  for (int r = 0; r < nc; ++r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) MiJt[r][i] = Jc[r][i];
    chol_solve_packed<NV>(L, MiJt[r]);
  }
#pragma unroll
  for (int r = 0; r < nc; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double g = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) g += Jc[r][i] * MiJt[c][i];
      G[r * (r + 1) / 2 + c] = g;
    }
  chol_packed<nc>(G);
#pragma unroll
  for (int r = 0; r < nc; ++r) {
    double g = a0[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) g += Jc[r][i] * a[i];
    lam[r] = -g;
  }
  chol_solve_packed<nc>(G, lam);
#pragma unroll
  for (int r = 0; r < nc; ++r)
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] += MiJt[r][i] * lam[r];
}

// Two ContactModel3D contacts of one stage (CT_PAIR3): [M Jc^T; Jc 0][a; -lam] = [tau - h; -a0] with Jc = [JcA; JcB] (rows of
// contact 0 first), a0 likewise; lam[0..2] = LOCAL force of contact 0, lam[3..5] of contact 1.  Same steps as contact_forward<DM, 3>
// per contact -- drift = classical acceleration of the contact point + Baumgarte terms, LOCAL linear Jacobian from the kinematics --
// with the walk along the chain (joint axes and origins in the world frame) shared by the two frames.
template <class DM, class ContactT, class MT>
EMPC_HD void contact_forward_pair3(const MT& m, const ContactT& ctA, const ContactT& ctB, const FrameCap<double>& ckA,
                                   const FrameCap<double>& ckB, const double* R0, const double* q, const double* cs,
                                   const double* sn, const double* L, double* a, double* lam) {
  constexpr int NV = DM::NV;
  constexpr int nc = 6;
  double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]}, a0[nc];
  {
    double gf[3], wxv[3];
    matTvec3<double>(ckA.R, ng, gf);
    cross3<double>(ckA.v + 3, ckA.v, wxv);
#pragma unroll
    for (int r = 0; r < 3; ++r) a0[r] = (ckA.a[r] - gf[r]) + wxv[r];
    if (ctA.gains[0] != 0.0)
#pragma unroll
      for (int r = 0; r < 3; ++r) a0[r] += ctA.gains[0] * (ckA.p[r] - ctA.ref_p[r]);
    if (ctA.gains[1] != 0.0)
#pragma unroll
      for (int r = 0; r < 3; ++r) a0[r] += ctA.gains[1] * ckA.v[r];
  }
  {
    double gf[3], wxv[3];
    matTvec3<double>(ckB.R, ng, gf);
    cross3<double>(ckB.v + 3, ckB.v, wxv);
#pragma unroll
    for (int r = 0; r < 3; ++r) a0[3 + r] = (ckB.a[r] - gf[r]) + wxv[r];
    if (ctB.gains[0] != 0.0)
#pragma unroll
      for (int r = 0; r < 3; ++r) a0[3 + r] += ctB.gains[0] * (ckB.p[r] - ctB.ref_p[r]);
    if (ctB.gains[1] != 0.0)
#pragma unroll
      for (int r = 0; r < 3; ++r) a0[3 + r] += ctB.gains[1] * ckB.v[r];
  }
  double Jc[nc][NV], MiJt[nc][NV];
  {
    const int bfA = m.frame_body[ctA.frame], bfB = m.frame_body[ctB.frame];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double w[3] = {R0[j], R0[3 + j], R0[6 + j]}, linA[3], linB[3];
      matTvec3<double>(ckA.R, w, linA);
      matTvec3<double>(ckB.R, w, linB);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Jc[r][j] = linA[r];
        Jc[3 + r][j] = linB[r];
      }
    }
    double Rw[9], pw[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rw[i] = R0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pw[i] = q[i];
#pragma unroll
    for (int j = 3; j < NV; ++j) {
      double z[3];
      bool onA = true, onB = true;
      if (j < 6) {
        z[0] = R0[j - 3];
        z[1] = R0[3 + j - 3];
        z[2] = R0[6 + j - 3];
      } else {
        const int b = j - 6 + 1;
        double Rj[9], XR[9], Rr[3], Rn[9];
        axis_rot<double>(m.axis[b], cs[b - 1], sn[b - 1], Rj);
        matmul3<double>(m.jplace_R[b], Rj, XR);
        matvec3<double>(Rw, m.jplace_p[b], Rr);
        matmul3<double>(Rw, XR, Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) Rw[i] = Rn[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pw[i] += Rr[i];
        double ax[3] = {m.axis[b][0], m.axis[b][1], m.axis[b][2]};
        matvec3<double>(Rw, ax, z);
        onA = (b <= bfA);
        onB = (b <= bfB);
      }
      double dA[3] = {ckA.p[0] - pw[0], ckA.p[1] - pw[1], ckA.p[2] - pw[2]}, dB[3] = {ckB.p[0] - pw[0], ckB.p[1] - pw[1], ckB.p[2] - pw[2]};
      double zxd[3], lin[3];
      cross3<double>(z, dA, zxd);
      matTvec3<double>(ckA.R, zxd, lin);
#pragma unroll
      for (int r = 0; r < 3; ++r) Jc[r][j] = onA ? lin[r] : 0.0;
      cross3<double>(z, dB, zxd);
      matTvec3<double>(ckB.R, zxd, lin);
#pragma unroll
      for (int r = 0; r < 3; ++r) Jc[3 + r][j] = onB ? lin[r] : 0.0;
    }
  }
  double G[nc * (nc + 1) / 2];  // packed nc x nc
#pragma unroll
  for (int r = 0; r < nc; ++r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) MiJt[r][i] = Jc[r][i];
    chol_solve_packed<NV>(L, MiJt[r]);
  }
#pragma unroll
  for (int r = 0; r < nc; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double g = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) g += Jc[r][i] * MiJt[c][i];
      G[r * (r + 1) / 2 + c] = g;
    }
  chol_packed_stop<nc>(G);
#pragma unroll
  for (int r = 0; r < nc; ++r) {
    double g = a0[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) g += Jc[r][i] * a[i];
    lam[r] = -g;
  }
  chol_solve_packed<nc>(G, lam);
#pragma unroll
  for (int r = 0; r < nc; ++r)
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] += MiJt[r][i] * lam[r];
}
// row offset of the contact whose force a ContactFrictionCone cost on frame `cframe` reads (crocoddyl's residual data looks the
// contact up by frame id): 3 when the stage has two contacts and the second one sits on that frame, 0 otherwise (a stage with
// one contact keeps that contact whatever the cost's frame says -- behaviour of rounds 1-5, and of the oracle)
template <int CT, class SetT>
EMPC_HD int cone_force_offset(const SetT& set, int cframe) {
  if constexpr (CT == CT_PAIR3) return (set.ncontacts > 1 && set.contacts[1].frame == cframe) ? 3 : 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Nominal evaluation of one node by ONE lane (used by the per-lane rollout and the calc kernels), in two layers:
//   dam_nominal   DifferentialActionModel{Free,Contact}FwdDynamics::calc(x, s): acceleration, contact force, cost sum
//   node_nominal  IntegratedActionModel{Euler,RK4}::calc on top of it (P.integrator)
//   terminal: the reference's IAM.calc(x) == calc(x, u = 0)   
// Outputs of node_nominal: xnext[NX], acc[NV] (generalized acceleration, reused by linearize), cost, usq[NU] (squashed
// control), lam[6] (contact force).
// ---------------------------------------------------------------------------------------------------------
// diagnostic builds (-DEMPC_STAMPS): cycle counter deltas per section, accumulated in a caller-provided array
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define EMPC_STAMP(i)                                               \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    if (stp) {                                                      \
      const unsigned long long now_ = __builtin_readcyclecounter(); \
      stp[i] += now_ - stp[31];                                     \
      stp[31] = now_;                                               \
    }                                                               \
  } while (0)
#else
#define EMPC_STAMP(i) \
  do {                \
  } while (0)
#endif
// SetT: the cost set either in the constant address space (scalar loads) or staged in LDS / generic memory
// dam_nominal: ell_out is the UNSCALED cost sum; with `euler_xnext` != nullptr the semi-implicit Euler step is taken in the
// middle of the function, where the r01 kernels had it (the compiler's schedule -- and with it the last bits of the
// iteration path of ill-conditioned problems -- stays what the golden vectors were recorded with)
template <class DM, int CT, class SetT>
EMPC_HD void dam_nominal(const EMPC_K DevProblem& P, const SetT& set, double smooth, const double* x, const double* s_in,
                         bool terminal, double* euler_xnext, double* acc, double& ell_out, double* usq, double* lam_out,
                         unsigned long long* stp = nullptr) {
  constexpr int NB = DM::NB, NV = DM::NV, NQ = DM::NQ, NU = DM::NU, NROT = DM::NROT;
  const auto& m = model_of<DM>(P);
  const auto PL = platform_of<DM>(P);
  const double dt = P.dt;
  double s[NU], u[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) s[i] = terminal ? 0.0 : s_in[i];
  if (P.use_squash) {  // one uniform branch, then straight-line code: the bound loads of all controls go out together
    double lbv[NU], ubv[NU];
    const int power = P.prm.smoothsat_power;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      lbv[i] = PL.u_lb[i];
      ubv[i] = PL.u_ub[i];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      double du;
      squash1(s[i], lbv[i], ubv[i], smooth, power, u[i], du);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = s[i];
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) usq[i] = u[i];
  double tau[NV];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double a_ = 0;
#pragma unroll
    for (int c = 0; c < NROT; ++c) a_ += PL.tau_f[r * NROT + c] * u[c];
    tau[r] = a_;
  }
#pragma unroll
  for (int i = 6; i < NV; ++i) tau[i] = u[NROT + i - 6];

  const double* q = x;
  const double* v = x + NQ;
  double R0[9], cs[NB], sn[NB];
  quat_to_R(q + 3, R0);
#pragma unroll
  for (int b = 1; b < NB; ++b) {
    fsincos(q[7 + b - 1], &sn[b - 1], &cs[b - 1]);
  }
  // frames referenced by this node's costs / contacts
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
  // State and control costs first: they need only x and s, so their residuals and parameter loads are out of the way
  // before the dynamics claim the registers (A.6).  Frame costs follow the bias pass, the friction cone the contact solve.
  double ell = 0;
  {
    double rstate[DM::NDX];  // residual of the most recent State cost (shared between costs with one reference)
    int rstate_of = -1;
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const auto& c = set.costs[ci];
      if (!c.active) continue;
      if (c.type == EMPC_COST_STATE) {
        if (!(c.ref_share >= 0 && c.ref_share == rstate_of)) {
          state_diff<DM>(c.ref, x, rstate, nullptr);
          rstate_of = (c.ref_share >= 0) ? c.ref_share : ci;
        }
        ell += c.weight * activation_value<DM::NDX>(c, rstate, DM::NDX);
      } else if (c.type == EMPC_COST_CONTROL) {
        ell += c.weight * control_cost_value<NU>(c, s, smooth, PL);
      }
    }
  }
  const bool use_contact = CT && P.has_contact && set.ncontacts > 0;
  int ccap = 0;
  if constexpr (CT) {
    if (use_contact) {  // the contact frame rides along in the bias pass (placement, velocity, drift acceleration)
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
  int ccap2 = 0;  // CT_PAIR3: capture slot of the stage's second contact frame
  if constexpr (CT == CT_PAIR3) {
    if (use_contact && set.ncontacts > 1) {
      const int cframe2 = set.contacts[1].frame;
      bool seen = false;
#pragma unroll
      for (int k = 0; k < NCAP; ++k)
        if (k < ncap && capf[k] == cframe2) {
          seen = true;
          ccap2 = k;
        }
      if (!seen) {
#pragma unroll
        for (int k = 0; k < NCAP; ++k)
          if (k == ncap) capf[k] = cframe2;
        ccap2 = ncap;
        ncap = (ncap < NCAP) ? ncap + 1 : ncap;
      }
    }
  }
  FrameCap<double> caps[NCAP];
  EMPC_STAMP(1);  // squash, tau, quaternion, joint sin/cos, frame scan
  // bias forces h = RNEA(q, v, 0)
  double zero[NV], h[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) zero[i] = 0.0;
  rnea_chain<NB, double>(m, R0, q, cs, sn, v, zero, true, -1, nullptr, h, ncap, capf, caps);
  EMPC_STAMP(2);  // RNEA bias
  // frame costs right after the pass that captured the frames, so the captures (24 doubles each) die before the inertia
  // matrix and its factor come alive; their sum joins the other costs at the end
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
  // joint-space inertia (packed lower triangle) by the composite-rigid-body algorithm
  double L[DM::NTRI];
  crba_chain<NB>(m, cs, sn, L);
  EMPC_STAMP(3);  // CRBA
  chol_packed<NV>(L);
  double a[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = tau[i] - h[i];
  chol_solve_packed<NV>(L, a);
  EMPC_STAMP(4);  // Cholesky + solve
  double lam[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (CT) if (use_contact) {
    FrameCap<double> ck = caps[0];
#pragma unroll
    for (int kk = 1; kk < NCAP; ++kk)
      if (kk == ccap) ck = caps[kk];
    if constexpr (CT == CT_PAIR3) {
      if (set.ncontacts > 1) {
        FrameCap<double> ck2 = caps[0];
#pragma unroll
        for (int kk = 1; kk < NCAP; ++kk)
          if (kk == ccap2) ck2 = caps[kk];
        contact_forward_pair3<DM>(m, set.contacts[0], set.contacts[1], ck, ck2, R0, q, cs, sn, L, a, lam);
      } else {
        contact_forward<DM, 3>(m, set.contacts[0], ck, R0, q, cs, sn, L, a, lam);
      }
    } else {
      contact_forward<DM, CT>(m, set.contacts[0], ck, R0, q, cs, sn, L, a, lam);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = a[i];
  if (lam_out)
    for (int i = 0; i < 6; ++i) lam_out[i] = lam[i];

  // Euler step (A.3)
  EMPC_STAMP(5);  // contact KKT
  if (euler_xnext) {
    double dxe[DM::NDX];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      dxe[i] = v[i] * dt + a[i] * dt * dt;
      dxe[NV + i] = a[i] * dt;
    }
    state_integrate<DM>(x, dxe, euler_xnext, nullptr);
  }
  EMPC_STAMP(6);  // Euler step

  // friction-cone costs need the contact force
  for (int ci = 0; ci < set.ncosts; ++ci) {
    const auto& c = set.costs[ci];
    if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE) continue;
    double r[6] = {0, 0, 0, 0, 0, 0};  // rows of A R_n^T precomputed by prepare_problem in ref[4..18]
    double lc[3] = {lam[0], lam[1], lam[2]};  // force of the contact the cost reads (selects, not a run-time index)
    if constexpr (CT == CT_PAIR3) {
      if (cone_force_offset<CT>(set, c.frame) != 0) {
        lc[0] = lam[3];
        lc[1] = lam[4];
        lc[2] = lam[5];
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
      r[i] = use_contact ? (c.ref[4 + 3 * i] * lc[0] + c.ref[5 + 3 * i] * lc[1] + c.ref[6 + 3 * i] * lc[2]) : 0.0;
    ell += c.weight * activation_value<6>(c, r, 5);
  }
  ell += ell_frames;
  ell_out = ell;
  EMPC_STAMP(7);  // costs
}

// IntegratedActionModelRK4::calc (src/factory/int-action.cpp:29-31; oracle/action.hpp node_calc_rk4): four evaluations
// of the differential model at y_i = x (+) c_i dt k_{i-1}, k_i = [v(y_i); a(y_i, s)]; the acceleration / squashing /
// contact outputs are those of stage 0 (src/sbfddp.cpp:144-145 reads differential[0]).
template <class DM, int CT, class SetT>
EMPC_HD void node_nominal_rk4(const EMPC_K DevProblem& P, const SetT& set, double smooth, const double* x, const double* s_in,
                              bool terminal, double* xnext, double* acc, double& cost_out, double* usq, double* lam_out) {
  constexpr int NV = DM::NV, NQ = DM::NQ, NX = DM::NX, NU = DM::NU, NDX = DM::NDX;
  const double dt = P.dt;
  const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
  double y[NX], kprev[NDX], ksum[NDX], ellsum = 0;
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = x[i];
  for (int st = 0; st < 4; ++st) {
    if (st > 0) {
      double dxr[NDX];
#pragma unroll
      for (int j = 0; j < NDX; ++j) dxr[j] = rk4_c[st] * dt * kprev[j];
      state_integrate<DM>(x, dxr, y, nullptr);
    }
    double a_[NV], us_[NU], lam_[6], ell_;
    dam_nominal<DM, CT>(P, set, smooth, y, s_in, terminal, (double*)nullptr, a_, ell_, us_, lam_);
    const double w = (st == 0 || st == 3) ? 1.0 : 2.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      kprev[j] = y[NQ + j];
      kprev[NV + j] = a_[j];
    }
    if (st == 0) {
#pragma unroll
      for (int j = 0; j < NDX; ++j) ksum[j] = kprev[j];
      ellsum = ell_;
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[j] = a_[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) usq[j] = us_[j];
      if (lam_out)
        for (int j = 0; j < 6; ++j) lam_out[j] = lam_[j];
    } else {
#pragma unroll
      for (int j = 0; j < NDX; ++j) ksum[j] = ksum[j] + w * kprev[j];
      ellsum = ellsum + w * ell_;
    }
  }
  double dx[NDX];
#pragma unroll
  for (int j = 0; j < NDX; ++j) dx[j] = ksum[j] * dt / 6.0;
  state_integrate<DM>(x, dx, xnext, nullptr);
  const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 / 6.0 : dt / 6.0;
  cost_out = ellsum * cscale;
}

template <class DM, int CT, class SetT>
EMPC_HD void node_nominal(const EMPC_K DevProblem& P, const SetT& set, double smooth, const double* x, const double* s_in,
                          bool terminal, double* xnext, double* acc, double& cost_out, double* usq, double* lam_out,
                          unsigned long long* stp = nullptr) {
  if (P.integrator == EMPC_INTEGRATOR_RK4) {
    node_nominal_rk4<DM, CT>(P, set, smooth, x, s_in, terminal, xnext, acc, cost_out, usq, lam_out);
    return;
  }
  double ell;
  dam_nominal<DM, CT>(P, set, smooth, x, s_in, terminal, xnext, acc, ell, usq, lam_out, stp);
  const double cscale = (terminal && !P.prm.terminal_dt_scaling) ? 1.0 : P.dt;
  cost_out = cscale * ell;
}

// ---------------------------------------------------------------------------------------------------------
// Plant model of the closed-loop MPC runs (reference: bindings/python/eagle_mpc/utils/simulator.py:8-29):
// DifferentialActionModelFreeFwdDynamics with the *unsquashed* ActuationModelMultiCopterBase, no costs, integrated
// with crocoddyl::IntegratedActionModelRK4.  `u` are rotor thrusts + arm torques.
// ---------------------------------------------------------------------------------------------------------
template <class DM>
EMPC_HD void free_fwd_acc(const EMPC_K DevProblem& P, const double* x, const double* u, double* a) {
  constexpr int NB = DM::NB, NV = DM::NV, NQ = DM::NQ, NROT = DM::NROT;
  const auto& m = model_of<DM>(P);
  const auto PL = platform_of<DM>(P);
  double tau[NV];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0;
#pragma unroll
    for (int c = 0; c < NROT; ++c) s += PL.tau_f[r * NROT + c] * u[c];
    tau[r] = s;
  }
#pragma unroll
  for (int i = 6; i < NV; ++i) tau[i] = u[NROT + i - 6];
  const double* q = x;
  const double* v = x + NQ;
  double R0[9], cs[NB], sn[NB];
  quat_to_R(q + 3, R0);
#pragma unroll
  for (int b = 1; b < NB; ++b) fsincos(q[7 + b - 1], &sn[b - 1], &cs[b - 1]);
  int capf[NCAP] = {0, 0};
  FrameCap<double> caps[NCAP];
  double zero[NV], h[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) zero[i] = 0.0;
  rnea_chain<NB, double>(m, R0, q, cs, sn, v, zero, true, -1, nullptr, h, 0, capf, caps);
  double L[DM::NTRI];
  crba_chain<NB>(m, cs, sn, L);
  chol_packed<NV>(L);
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = tau[i] - h[i];
  chol_solve_packed<NV>(L, a);
}

template <class DM>
EMPC_HD void plant_rk4_step(const EMPC_K DevProblem& P, const double* x, const double* u, double dt, double* xnext) {
  constexpr int NV = DM::NV, NQ = DM::NQ, NX = DM::NX, NDX = DM::NDX;
  const double rk4_c[4] = {0.0, 0.5, 0.5, 1.0};
  double ksum[NDX], ki[NDX], y[NX], dxi[NDX];
#pragma unroll
  for (int i = 0; i < NX; ++i) y[i] = x[i];
#pragma unroll
  for (int i = 0; i < NDX; ++i) ksum[i] = 0.0;
  for (int stage = 0; stage < 4; ++stage) {
    if (stage > 0) {
#pragma unroll
      for (int i = 0; i < NDX; ++i) dxi[i] = rk4_c[stage] * ki[i] * dt;
      state_integrate<DM>(x, dxi, y, (double*)nullptr);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) ki[i] = y[NQ + i];
    free_fwd_acc<DM>(P, y, u, ki + NV);
    const double w = (stage == 0 || stage == 3) ? 1.0 : 2.0;
#pragma unroll
    for (int i = 0; i < NDX; ++i) ksum[i] += w * ki[i];
  }
#pragma unroll
  for (int i = 0; i < NDX; ++i) dxi[i] = ksum[i] * dt / 6.0;
  state_integrate<DM>(x, dxi, xnext, (double*)nullptr);
}

}  // namespace empc
