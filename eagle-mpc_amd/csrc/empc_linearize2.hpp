// empc_linearize2.hpp -- HOT-A kernel body, second generation.
//
// One (trajectory, node) unit is handled by LPU lanes (32 or 64, never crossing a wavefront):
//   * every lane-invariant ("nominal") quantity is computed ONCE and kept in LDS: squashed controls, joint rotations,
//     body velocities / accelerations / momenta / subtree forces, the Cholesky factor of the joint-space inertia, the
//     Euler step, its Lie Jacobians, cost residuals and their activation derivatives;
//   * lane j carries only TANGENT quantities of direction j (dq_j, dv_j or da_j) through a hand-derived tangent
//     recursion of the body-frame Newton-Euler equations, reading nominals from LDS with broadcast reads;
//   * columns of Fx, Fu, Lxx, Lxu, Luu are produced in registers and stored straight to the tape record.
// Lanes [0,NV): dq_j   [NV,2NV): dv_j   [2NV,3NV): da_j (-> joint-space inertia columns), then reused as control columns.
//
// Semantics: identical to linearize_unit (v1) and to the oracle: crocoddyl IntegratedActionModelEuler::calcDiff over
// DifferentialActionModelFreeFwdDynamics with ActuationSquashingModel and CostModelSum.
#pragma once
#include "empc_kernels.hpp"

namespace empc {

template <class DM>
struct Lin2Smem {
  static constexpr int NB = DM::NB, NV = DM::NV, NX = DM::NX, NU = DM::NU, NDX = DM::NDX, NJ = DM::NJ;
  static constexpr int OFF_X = 0;                   // x | s | a
  static constexpr int OFF_S = OFF_X + NX;
  static constexpr int OFF_A = OFF_S + NU;
  static constexpr int OFF_USQ = OFF_A + NV;        // sigma(s), sigma'(s)
  static constexpr int OFF_DUS = OFF_USQ + NU;
  static constexpr int OFF_CS = OFF_DUS + NU;       // cos / sin of the joint angles
  static constexpr int OFF_SN = OFF_CS + (NJ > 0 ? NJ : 1);
  static constexpr int OFF_R0 = OFF_SN + (NJ > 0 ? NJ : 1);
  static constexpr int OFF_GL = OFF_R0 + 9;         // R0^T (-g)
  static constexpr int OFF_BODY = OFF_GL + 3;       // per body: XR 9 | Rw 9 | pw 3 | VX 6 | AX 6 | vb 6 | Iv 6 | fsub 6
  static constexpr int BODY = 51;
  static constexpr int B_XR = 0, B_RW = 9, B_PW = 18, B_VX = 21, B_AX = 27, B_VB = 33, B_IV = 39, B_FS = 45;
  static constexpr int OFF_M = OFF_BODY + NB * BODY;  // NV x NV: inertia columns, then the Cholesky factor (recip. diagonal)
  static constexpr int OFF_DXE = OFF_M + NV * NV;     // Euler step dx (NDX), J2 = Jexp6 (36), J1 (36), xnext (NX)
  static constexpr int OFF_J2 = OFF_DXE + NDX;
  static constexpr int OFF_J1 = OFF_J2 + 36;
  static constexpr int OFF_XN = OFF_J1 + 36;
  static constexpr int OFF_GAP = OFF_XN + NX;         // gap written to the next record (NDX) and, for t = 0, fs[0] (NDX)
  static constexpr int OFF_FR = OFF_GAP + 2 * NDX;    // per captured frame: R 9 | p 3 | v 6
  static constexpr int OFF_CST = OFF_FR + NCAP * 18;  // cost nominal slots
  static constexpr int NSLOT = 2;                     // State costs handled side by side (a stage of the shipped files has two)
  static constexpr int SLOT = 3 * NDX + 36 + 4;       // r | Ar | Arr | J6 | value
  // residual-Jacobian exchange 6 x (NDX + NU) of the cost rounds (S6): lives where the inertia factor, the Euler step and its
  // Lie Jacobians were -- all of them dead once S5 has written Fx / Fu (LDS per unit decides how many units a CU holds)
  static constexpr int OFF_RSH = OFF_M;
  static_assert(6 * (NDX + NU) <= OFF_GAP - OFF_M, "the exchange area must fit the block it aliases");
  static constexpr int OFF_RED = OFF_CST + NSLOT * SLOT;  // small reduction area: cost sum | control-cost partial sums
  // contact block (nc = 3 rows for ContactModel3D, 6 for ContactModel6D): lambda 6 | fext 6 | cone rows 15 + Ar 5 + Arr 5 |
  // Jc nc x NV | M^-1 Jc^T NV x nc | packed G nc (nc + 1) / 2.  The nc-dependent part comes last, so a unit of the 3D
  // instantiation is no larger than it has to be (units per CU are LDS-bound in the contact problem)
  static constexpr int OFF_LAM = OFF_RED + 16;
  static_assert(NU + 1 <= 12, "reduction area: cost sum | control-cost partial sums | [12..14] flags of the unit for the role lanes");
  static constexpr int OFF_FEXT = OFF_LAM + 6;
  static constexpr int OFF_CONE = OFF_FEXT + 6;
  static constexpr int OFF_JC = OFF_CONE + 26;
  static constexpr int off_mij(int nc) { return OFF_JC + nc * NV; }
  static constexpr int off_g(int nc) { return OFF_JC + 2 * nc * NV; }
  static constexpr int size_for(int nc) {
    return nc == 0 ? SIZE_NC : (nc == CT_MIXED ? size_for(6) : (nc == CT_PAIR3 ? size_for(6) + 6 : (off_g(nc) + nc * (nc + 1) / 2 + 1) / 2 * 2));
  }
  // CT_PAIR3 (two ContactModel3D of one stage): the spatial force of the second contact on ITS body, behind the six-row unit
  static constexpr int OFF_FEXT2 = (OFF_JC + 12 * NV + 21 + 1) / 2 * 2;
  static constexpr int SIZE = OFF_FEXT2 + 6;  // the largest unit (six rows + the second contact's force)
  static constexpr int SIZE_NC = (OFF_LAM + 1) / 2 * 2;  // problems without contacts never touch the contact block
};

// Lanes per (trajectory, node) unit.  3 NV tangent directions (dq | dv | da) and NDX + NU output columns: 27 / 27 for the 9-dof
// arm -> 32 lanes, two units per wavefront.  The 11-dof arm needs 33 / 33: one lane over a half wavefront -- a whole
// wavefront per unit with 31 idle lanes and half the units in flight.  FOLDED layout (free dynamics only): 32 lanes, the 33rd
// direction and column folded away --
//   * direction da of the LAST joint only supplies the last diagonal entry of the joint-space inertia (the rest of its column is
//     the upper triangle, which the Cholesky factorisation never reads): S^T I S of the last body, evaluated directly (lane 0);
//   * the control column of the last joint (k = NU - 1) is a second pass of lane 2 NV - 1 (the last dv lane) through the solve,
//     the Control-cost and the store stages, with its own three accumulators (the Control costs only touch the diagonal of Luu).
// Same operations per quantity as the 64-lane layout.
template <class DM, int CT>
constexpr bool lin_folded() {
  return CT == 0 && 3 * DM::NV == 33 && DM::NDX + DM::NU == 33;
}
template <class DM, int CT>
constexpr int lin_lanes_per_unit() {
  return (3 * DM::NV <= 32 || lin_folded<DM, CT>()) ? 32 : 64;
}

// forward kinematics + nominal Newton-Euler quantities of one unit, executed by ONE lane
template <class DM, class MT>
EMPC_HD void lin2_nominal_chain(const MT& m, double* N, int cbody = -1, int cbody2 = -1) {
  typedef Lin2Smem<DM> SM;
  constexpr int NB = DM::NB, NQ = DM::NQ, NX = DM::NX;
  const double* x = N + SM::OFF_X;
  const double* acc = N + SM::OFF_A;
  const double* R0 = N + SM::OFF_R0;
  double* gl = N + SM::OFF_GL;
  {
    double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
    matTvec3<double>(R0, ng, gl);
  }
  // body 0
  {
    double* B0 = N + SM::OFF_BODY;
#pragma unroll
    for (int i = 0; i < 9; ++i) B0[SM::B_RW + i] = R0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) B0[SM::B_PW + i] = x[i];
    double vb[6], ab[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      vb[i] = x[NQ + i];
      ab[i] = acc[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) ab[i] += gl[i];
    double Iv[6], Ia[6], c1[3], c2[3], c3[3];
    inertia_apply<double>(m, 0, vb, Iv);
    inertia_apply<double>(m, 0, ab, Ia);
    cross3<double>(vb + 3, Iv, c1);
    cross3<double>(vb + 3, Iv + 3, c2);
    cross3<double>(vb, Iv, c3);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      B0[SM::B_VB + i] = vb[i];
      B0[SM::B_IV + i] = Iv[i];
      B0[SM::B_VX + i] = 0.0;
      B0[SM::B_AX + i] = ab[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      B0[SM::B_FS + i] = Ia[i] + c1[i];
      B0[SM::B_FS + 3 + i] = Ia[3 + i] + c2[i] + c3[i];
    }
  }
#pragma unroll
  for (int b = 1; b < NB; ++b) {
    double* Bb = N + SM::OFF_BODY + b * SM::BODY;
    const double* Bp = N + SM::OFF_BODY + (b - 1) * SM::BODY;
    double Rj[9], XR[9];
    axis_rot<double>(m.axis[b], N[SM::OFF_CS + b - 1], N[SM::OFF_SN + b - 1], Rj);
    matmul3<double>(m.jplace_R[b], Rj, XR);
    double Rw[9], Rr[3];
    matmul3<double>(Bp + SM::B_RW, XR, Rw);
    matvec3<double>(Bp + SM::B_RW, m.jplace_p[b], Rr);
    const double* vp = Bp + SM::B_VB;
    // parent acceleration (full, including joint terms) is reconstructed below and kept in `ap`
    double ap[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ap[i] = Bp[SM::B_AX + i];  // for body 0 AX holds the full base acceleration
    if (b > 1) {
      // full acceleration of the parent = AX + S qdd + v x (S qd)
      const double qd = x[NQ + 6 + (b - 1) - 1], qdd = acc[6 + (b - 1) - 1];
      double sv[3], c1[3], c2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) sv[i] = m.axis[b - 1][i] * qd;
      cross3<double>(vp, sv, c1);
      cross3<double>(vp + 3, sv, c2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ap[i] += c1[i];
        ap[3 + i] += m.axis[b - 1][i] * qdd + c2[i];
      }
    }
    double wxr[3], tmp[3], VX[6], AX[6];
    cross3<double>(vp + 3, m.jplace_p[b], wxr);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = vp[i] + wxr[i];
    matTvec3<double>(XR, tmp, VX);
    matTvec3<double>(XR, vp + 3, VX + 3);
    cross3<double>(ap + 3, m.jplace_p[b], wxr);
#pragma unroll
    for (int i = 0; i < 3; ++i) tmp[i] = ap[i] + wxr[i];
    matTvec3<double>(XR, tmp, AX);
    matTvec3<double>(XR, ap + 3, AX + 3);
    const double qd = x[NQ + 6 + b - 1], qdd = acc[6 + b - 1];
    double vb[6], ab[6], sv[3], c1[3], c2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) sv[i] = m.axis[b][i] * qd;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      vb[i] = VX[i];
      vb[3 + i] = VX[3 + i] + sv[i];
    }
    cross3<double>(vb, sv, c1);
    cross3<double>(vb + 3, sv, c2);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ab[i] = AX[i] + c1[i];
      ab[3 + i] = AX[3 + i] + m.axis[b][i] * qdd + c2[i];
    }
    double Iv[6], Ia[6], d1[3], d2[3], d3[3];
    inertia_apply<double>(m, b, vb, Iv);
    inertia_apply<double>(m, b, ab, Ia);
    cross3<double>(vb + 3, Iv, d1);
    cross3<double>(vb + 3, Iv + 3, d2);
    cross3<double>(vb, Iv, d3);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      Bb[SM::B_XR + i] = XR[i];
      Bb[SM::B_RW + i] = Rw[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) Bb[SM::B_PW + i] = Bp[SM::B_PW + i] + Rr[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      Bb[SM::B_VX + i] = VX[i];
      Bb[SM::B_AX + i] = AX[i];
      Bb[SM::B_VB + i] = vb[i];
      Bb[SM::B_IV + i] = Iv[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Bb[SM::B_FS + i] = Ia[i] + d1[i];
      Bb[SM::B_FS + 3 + i] = Ia[3 + i] + d2[i] + d3[i];
    }
  }
  // external (contact) force acting on body `cbody`, body coordinates
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (b == cbody) {
      double* Bb = N + SM::OFF_BODY + b * SM::BODY;
#pragma unroll
      for (int i = 0; i < 6; ++i) Bb[SM::B_FS + i] -= N[SM::OFF_FEXT + i];
    }
  // (CT_PAIR3: the second contact of the stage, on body `cbody2` -- possibly the same body)
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (b == cbody2) {
      double* Bb = N + SM::OFF_BODY + b * SM::BODY;
#pragma unroll
      for (int i = 0; i < 6; ++i) Bb[SM::B_FS + i] -= N[SM::OFF_FEXT2 + i];
    }
  // subtree forces
#pragma unroll
  for (int b = NB - 1; b >= 1; --b) {
    double* Bb = N + SM::OFF_BODY + b * SM::BODY;
    double* Bp = N + SM::OFF_BODY + (b - 1) * SM::BODY;
    double fl[3], fn[3], rxf[3];
    matvec3<double>(Bb + SM::B_XR, Bb + SM::B_FS, fl);
    matvec3<double>(Bb + SM::B_XR, Bb + SM::B_FS + 3, fn);
    cross3<double>(m.jplace_p[b], fl, rxf);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Bp[SM::B_FS + i] += fl[i];
      Bp[SM::B_FS + 3 + i] += fn[i] + rxf[i];
    }
  }
}

// tangent of the inertia-weighted force: dF = I da + dv x* (I v) + v x* (I dv)
template <class DM, class MT>
EMPC_HD void lin2_dforce(const MT& m, int b, const double* Bb, const double* dv, const double* da,
                         double* df) {
  typedef Lin2Smem<DM> SM;
  double Ida[6], Idv[6], c1[3], c2[3], c3[3], e1[3], e2[3], e3[3];
  inertia_apply<double>(m, b, da, Ida);
  inertia_apply<double>(m, b, dv, Idv);
  const double* Iv = Bb + SM::B_IV;
  const double* vb = Bb + SM::B_VB;
  cross3<double>(dv + 3, Iv, c1);
  cross3<double>(dv + 3, Iv + 3, c2);
  cross3<double>(dv, Iv, c3);
  cross3<double>(vb + 3, Idv, e1);
  cross3<double>(vb + 3, Idv + 3, e2);
  cross3<double>(vb, Idv, e3);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    df[i] = Ida[i] + c1[i] + e1[i];
    df[3 + i] = Ida[3 + i] + c2[i] + c3[i] + e2[i] + e3[i];
  }
}

// Tangent recursion of RNEA(q, v, a) for the direction owned by `lane`. Outputs dtau[NV]; capdv[c] = d(body velocity)
// of the body carrying captured frame c.
template <class DM, class MT>
EMPC_HD void lin2_tangent(const MT& m, const double* N, int lane, double* dtau, int ncap,
                          const int* capf, double (*capdv)[6], int cbody = -1, double* capda = nullptr, int cbody2 = -1,
                          double* capda2 = nullptr) {
  typedef Lin2Smem<DM> SM;
  constexpr int NB = DM::NB, NV = DM::NV, NQ = DM::NQ, NX = DM::NX;
  const double* x = N + SM::OFF_X;
  const bool jq = lane < NV, jv = lane >= NV && lane < 2 * NV, ja = lane >= 2 * NV && lane < 3 * NV;
  const int kq = lane, kv = lane - NV, ka = lane - 2 * NV;
  double df[NB][6];
  double dv[6], da[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    dv[i] = (jv && kv == i) ? 1.0 : 0.0;
    da[i] = (ja && ka == i) ? 1.0 : 0.0;
  }
  if (jq && kq >= 3 && kq < 6) {
    // d(R0^T (-g)) for a rotation of the base about body axis e: -e x (R0^T (-g))
    const double* gl = N + SM::OFF_GL;
    double e[3] = {0, 0, 0}, c[3];
    e[kq - 3] = 1.0;
    cross3<double>(e, gl, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) da[i] = -c[i];
  }
#pragma unroll
  for (int c = 0; c < NCAP; ++c)
    if (c < ncap && m.frame_body[capf[c]] == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) capdv[c][i] = dv[i];
    }
  if (cbody == 0 && capda) {
#pragma unroll
    for (int i = 0; i < 6; ++i) capda[i] = da[i];
  }
  if (cbody2 == 0 && capda2) {
#pragma unroll
    for (int i = 0; i < 6; ++i) capda2[i] = da[i];
  }
  lin2_dforce<DM>(m, 0, N + SM::OFF_BODY, dv, da, df[0]);
#pragma unroll
  for (int b = 1; b < NB; ++b) {
    const double* Bb = N + SM::OFF_BODY + b * SM::BODY;
    const double* XR = Bb + SM::B_XR;
    const double dth = (jq && kq == 6 + b - 1) ? 1.0 : 0.0;
    const double dqd = (jv && kv == 6 + b - 1) ? 1.0 : 0.0;
    const double dqdd = (ja && ka == 6 + b - 1) ? 1.0 : 0.0;
    const double qd = x[NQ + 6 + b - 1];
    double ax[3] = {m.axis[b][0], m.axis[b][1], m.axis[b][2]};
    double t[3], wxr[3], nv[6], na[6], c[3];
    // velocity
    cross3<double>(dv + 3, m.jplace_p[b], wxr);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = dv[i] + wxr[i];
    matTvec3<double>(XR, t, nv);
    matTvec3<double>(XR, dv + 3, nv + 3);
    cross3<double>(ax, Bb + SM::B_VX, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) nv[i] -= dth * c[i];
    cross3<double>(ax, Bb + SM::B_VX + 3, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) nv[3 + i] += ax[i] * dqd - dth * c[i];
    // acceleration
    cross3<double>(da + 3, m.jplace_p[b], wxr);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = da[i] + wxr[i];
    matTvec3<double>(XR, t, na);
    matTvec3<double>(XR, da + 3, na + 3);
    cross3<double>(ax, Bb + SM::B_AX, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) na[i] -= dth * c[i];
    cross3<double>(ax, Bb + SM::B_AX + 3, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) na[3 + i] += ax[i] * dqdd - dth * c[i];
    // d( v x (S qd) ) = dv x (ax qd) + v x (ax dqd)
    {
      double c1[3], c2[3], c3[3], c4[3];
      cross3<double>(nv, ax, c1);
      cross3<double>(nv + 3, ax, c2);
      cross3<double>(Bb + SM::B_VB, ax, c3);
      cross3<double>(Bb + SM::B_VB + 3, ax, c4);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        na[i] += c1[i] * qd + c3[i] * dqd;
        na[3 + i] += c2[i] * qd + c4[i] * dqd;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      dv[i] = nv[i];
      da[i] = na[i];
    }
#pragma unroll
    for (int cc = 0; cc < NCAP; ++cc)
      if (cc < ncap && m.frame_body[capf[cc]] == b) {
#pragma unroll
        for (int i = 0; i < 6; ++i) capdv[cc][i] = dv[i];
      }
    if (cbody == b && capda) {
#pragma unroll
      for (int i = 0; i < 6; ++i) capda[i] = da[i];
    }
    if (cbody2 == b && capda2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) capda2[i] = da[i];
    }
    lin2_dforce<DM>(m, b, Bb, dv, da, df[b]);
  }
#pragma unroll
  for (int b = NB - 1; b >= 1; --b) {
    const double* Bb = N + SM::OFF_BODY + b * SM::BODY;
    const double dth = (jq && kq == 6 + b - 1) ? 1.0 : 0.0;
    double ax[3] = {m.axis[b][0], m.axis[b][1], m.axis[b][2]};
    dtau[6 + b - 1] = dot3<double>(ax, df[b] + 3);
    double c1[3], c2[3], gl_[3], ga[3], pl[3], pa[3], rxf[3];
    cross3<double>(ax, Bb + SM::B_FS, c1);
    cross3<double>(ax, Bb + SM::B_FS + 3, c2);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      gl_[i] = df[b][i] + dth * c1[i];
      ga[i] = df[b][3 + i] + dth * c2[i];
    }
    matvec3<double>(Bb + SM::B_XR, gl_, pl);
    matvec3<double>(Bb + SM::B_XR, ga, pa);
    cross3<double>(m.jplace_p[b], pl, rxf);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      df[b - 1][i] += pl[i];
      df[b - 1][3 + i] += pa[i] + rxf[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) dtau[i] = df[0][i];
}

// LOCAL frame Jacobian column of generalized velocity `j` for a frame with world placement (Rf, pf) on body bf
template <class DM, class MT>
EMPC_HD void lin2_frame_jcol(const MT& m, const double* N, int j, int bf, const double* Rf,
                             const double* pf, double* col) {
  typedef Lin2Smem<DM> SM;
  constexpr int NB = DM::NB, NV = DM::NV;
#pragma unroll
  for (int i = 0; i < 6; ++i) col[i] = 0.0;
  if (j >= NV) return;
  const double* R0 = N + SM::OFF_R0;
  if (j < 3) {
    double w[3] = {R0[j], R0[3 + j], R0[6 + j]};
    matTvec3<double>(Rf, w, col);
    return;
  }
  double z[3], org[3];
  if (j < 6) {
    const int k = j - 3;
    z[0] = R0[k];
    z[1] = R0[3 + k];
    z[2] = R0[6 + k];
    const double* p0 = N + SM::OFF_BODY + SM::B_PW;
    org[0] = p0[0];
    org[1] = p0[1];
    org[2] = p0[2];
  } else {
    const int b = j - 6 + 1;
    if (b > bf) return;
    bool found = false;
#pragma unroll
    for (int bb = 1; bb < NB; ++bb)
      if (bb == b) {
        const double* Bb = N + SM::OFF_BODY + bb * SM::BODY;
        double ax[3] = {m.axis[bb][0], m.axis[bb][1], m.axis[bb][2]};
        matvec3<double>(Bb + SM::B_RW, ax, z);
        org[0] = Bb[SM::B_PW];
        org[1] = Bb[SM::B_PW + 1];
        org[2] = Bb[SM::B_PW + 2];
        found = true;
      }
    if (!found) return;
  }
  double d[3] = {pf[0] - org[0], pf[1] - org[1], pf[2] - org[2]}, zxd[3];
  cross3<double>(z, d, zxd);
  matTvec3<double>(Rf, zxd, col);
  matTvec3<double>(Rf, z, col + 3);
}

#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define LIN_STAMP(i)                                              \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    lst[i] += now_ - lst[15];                                     \
    lst[15] = now_;                                               \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
#else
#define LIN_STAMP(i) \
  do {               \
  } while (0)
#endif
// scheduling fence: a block of LDS reads stays ahead of the arithmetic that consumes it (without it the compiler pairs every
// read with its own wait: one exposed LDS round trip per operand)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define LIN_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define LIN_FENCE() \
  do {              \
  } while (0)
#endif
// FR: this instantiation handles the units whose cost set captures operational frames (frame costs or a contact);
// FR = false is the lean body for all other units -- the frame Jacobian / velocity-derivative columns (72 registers per
// lane) do not exist in it.  The kernel is launched once per flavour; a unit returns at once from the wrong one
// (flag bit 0 of costs[0].reserved, set by prepare_problem).
// Role split (RW > 0, GPU only): the three single-lane sections of a unit -- nominal chain, Euler step with its Lie
// Jacobians, state differences of the State costs -- run the same code for every unit of a workgroup (all units of a
// block are trajectories of one knot: same cost set).  Instead of every wavefront running them one after the other on
// one lane per unit, wavefront 0 runs the chain for ALL units of the block (lane = unit), wavefront 1 the Euler step,
// wavefront 2 (or 1 when the block has two) the state differences, between two workgroup barriers.
struct LinRole {
  int tid;       // thread in the block
  int upb;       // units per block
  int usz;       // doubles of LDS per unit
  int b0;        // index of the block's unit 0 in the list of trajectories (DevBuffers::lin_list) or trajectory itself
  int n;         // entries of that list (D.B without a list)
  const int* list;
  double* base;  // LDS of unit 0
  bool active;   // this thread's unit has work (the threads of idle units still serve as role lanes)
};
template <class DM, int CT, bool FR, class Exec, int RW = 0>
EMPC_HD void linearize_unit2(Exec& ex, const DevBuffers& D, int b, int t, int lpu, double* N, const LinRole* RL = nullptr) {
  typedef Lin2Smem<DM> SM;
  constexpr int NB = DM::NB, NV = DM::NV, NQ = DM::NQ, NX = DM::NX, NDX = DM::NDX, NU = DM::NU, NROT = DM::NROT;
  constexpr int REC = DM::REC;
  static_assert(NU <= NV, "control columns reuse the NV inertia-column lanes");
  constexpr bool FOLD = lin_folded<DM, CT>();  // 32-lane layout of the 11-dof class (see lin_folded)
  constexpr int LXL = 2 * NV - 1;              // FOLD: the lane that also carries control column KX
  constexpr int KX = NU - 1;
  const EMPC_K DevProblem& P = EMPC_KREF(DevProblem, D.P);
  const auto& m = model_of<DM>(P);
  const TrajState& st = D.st[b];
  const int T = D.T;
  const bool terminal = (t == T);
  const EMPC_K EmpcCostSet& set = EMPC_KPTR(EmpcCostSet, D.sets)[EMPC_KPTR(int, D.knot_set)[t]];
  const double dt = P.dt;
  const double smooth = st.smooth;
  const bool raw = D.raw != 0;  // RK4 stage record: differential-model derivatives, no integrator, no gaps
  const bool feas = st.is_feasible != 0 || raw;
  double* out = D.tape + ((size_t)b * (T + 1) + t) * REC;

  if (set_uses_frames(set) != FR) return;
  // captured frames (uniform over the unit)
  int capf[NCAP] = {0, 0};
  int ncap = 0;
  for (int ci = 0; FR && ci < set.ncosts; ++ci) {
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

  // contact of this node: CT = 3 (ContactModel3D) or 6 (ContactModel6D) constraint rows, fixed per kernel instantiation
  constexpr int NCR = CT ? ct_rows(CT) : 3;
  constexpr int OFF_MIJ = SM::off_mij(NCR), OFF_G = SM::off_g(NCR);
  const bool use_contact = FR && CT && P.has_contact && set.ncontacts > 0;
  int cframe = -1, cbody = -1, ccap = 0;
  if (use_contact) {
    cframe = set.contacts[0].frame;
    cbody = m.frame_body[cframe];
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
  // CT_PAIR3: the stage's second ContactModel3D (rows 3-5); this body runs only on nodes that have two (the launcher sends the
  // others to the 3-row body)
  int cframe2 = -1, cbody2 = -1, ccap2 = 0;
  if constexpr (CT == CT_PAIR3) {
    if (use_contact) {
      cframe2 = set.contacts[1].frame;
      cbody2 = m.frame_body[cframe2];
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

#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long lst[16];
  for (int i = 0; i < 16; ++i) lst[i] = 0;
  lst[15] = __builtin_readcyclecounter();
#endif
  // the three single-lane sections of a unit, as functions of the unit's LDS block (see LinRole)
  auto chain_section = [&](double* Nu) {
    if constexpr (CT == CT_PAIR3)
      lin2_nominal_chain<DM>(m, Nu, use_contact ? cbody : -1, use_contact ? cbody2 : -1);
    else
      lin2_nominal_chain<DM>(m, Nu, use_contact ? cbody : -1);
    // nominal frame data
#pragma unroll
    for (int c = 0; c < NCAP; ++c) {
      if (c >= ncap) continue;
      const int f = capf[c];
      const int bf = m.frame_body[f];
      double Rb[9], pb[3], vb[6];
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
        if (bb == bf) {
          const double* Bb = Nu + SM::OFF_BODY + bb * SM::BODY;
#pragma unroll
          for (int i = 0; i < 9; ++i) Rb[i] = Bb[SM::B_RW + i];
#pragma unroll
          for (int i = 0; i < 3; ++i) pb[i] = Bb[SM::B_PW + i];
#pragma unroll
          for (int i = 0; i < 6; ++i) vb[i] = Bb[SM::B_VB + i];
        }
      double* F = Nu + SM::OFF_FR + c * 18;
      double Rf[9], Rp[3], wxr[3], tmp[3], fv[6];
      matmul3<double>(Rb, m.frame_R[f], Rf);
      matvec3<double>(Rb, m.frame_p[f], Rp);
      cross3<double>(vb + 3, m.frame_p[f], wxr);
#pragma unroll
      for (int i = 0; i < 3; ++i) tmp[i] = vb[i] + wxr[i];
      matTvec3<double>(m.frame_R[f], tmp, fv);
      matTvec3<double>(m.frame_R[f], vb + 3, fv + 3);
#pragma unroll
      for (int i = 0; i < 9; ++i) F[i] = Rf[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) F[9 + i] = pb[i] + Rp[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) F[12 + i] = fv[i];
    }
  };
  // Euler step and its Lie Jacobians (J1, J2), next state
  auto euler_section = [&](double* Nu) {
    double x[NX], dxe[NDX], xnext[NX], pe[3], J2[36];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Nu[SM::OFF_X + i];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double ai = Nu[SM::OFF_A + i];
      dxe[i] = x[NQ + i] * dt + ai * dt * dt;
      dxe[NV + i] = ai * dt;
    }
    state_integrate<DM>(x, dxe, xnext, pe);
    Jexp6(dxe, pe, J2);
    double qe[4], pe2[3], Re[9], Px[9], RtP[9];
    exp6_quat(dxe, qe, pe2);
    quat_to_R(qe, Re);
    skew3(pe2, Px);
    matTmul3<double>(Re, Px, RtP);
    // J1 = Ad(exp6(xi)^-1) = [[R^T, -R^T [p]x],[0, R^T]]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Nu[SM::OFF_J1 + r * 6 + c] = Re[3 * c + r];
        Nu[SM::OFF_J1 + r * 6 + 3 + c] = -RtP[3 * r + c];
        Nu[SM::OFF_J1 + (3 + r) * 6 + c] = 0.0;
        Nu[SM::OFF_J1 + (3 + r) * 6 + 3 + c] = Re[3 * c + r];
      }
#pragma unroll
    for (int i = 0; i < 36; ++i) Nu[SM::OFF_J2 + i] = J2[i];
#pragma unroll
    for (int i = 0; i < NDX; ++i) Nu[SM::OFF_DXE + i] = dxe[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) Nu[SM::OFF_XN + i] = xnext[i];
  };
  // gaps: fs[t+1] = xnext (-) xs[t+1];  fs[0] = x0 (-) xs[0].  Its own section (its own wavefront in the role phase): it takes
  // the Euler step again -- the same operations, the same xnext -- instead of waiting for the Jacobian section's
  auto gap_section = [&](double* Nu, int bu, bool feasu) {
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Nu[SM::OFF_X + i];
    if (!terminal) {
      double gap[NDX];
      if (!feasu) {
        double dxe[NDX], xnext[NX], xn[NX];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const double ai = Nu[SM::OFF_A + i];
          dxe[i] = x[NQ + i] * dt + ai * dt * dt;
          dxe[NV + i] = ai * dt;
        }
        state_integrate<DM>(x, dxe, xnext, nullptr);
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = Nu[SM::OFF_GAP + i];  // staged at S0
        state_diff<DM>(xn, xnext, gap, nullptr);
      }
#pragma unroll
      for (int i = 0; i < NDX; ++i) Nu[SM::OFF_GAP + i] = feasu ? 0.0 : gap[i];
    }
    if (t == 0 && !raw) {
      double gap[NDX];
      if (!feasu) state_diff<DM>(x, D.x0 + (size_t)bu * NX, gap, nullptr);
#pragma unroll
      for (int i = 0; i < NDX; ++i) Nu[SM::OFF_GAP + NDX + i] = feasu ? 0.0 : gap[i];
    }
  };
  // State cost at position k0 + q of the set's State-cost list: state difference and its log Jacobian, unless an earlier cost of
  // the same group has the same reference (SetInfo::state_own)
  const EMPC_K SetInfo& si = EMPC_KPTR(SetInfo, D.set_info)[EMPC_KPTR(int, D.knot_set)[t]];
  auto owner_section = [&](double* Nu, int k0, int q) {
    if (k0 + q >= si.n_state) return;
    if (si.state_own[k0 + q] != k0 + q && si.state_own[k0 + q] >= k0) return;
    const EMPC_K EmpcCost& c = set.costs[si.state_ci[k0 + q]];
    double* S = Nu + SM::OFF_CST + q * SM::SLOT;
    double xref[NX], dpl[3];
#pragma unroll
    for (int i = 0; i < NX; ++i) xref[i] = c.ref[i];
    // residual and log Jacobian go straight to their LDS slot (r | Ar | Arr | J6 | value)
    state_diff<DM>(xref, Nu + SM::OFF_X, S, dpl);
    Jlog6(S, dpl, S + 3 * NDX);
  };
  // this thread's share of the role phase (RW > 0)
  auto role_phase = [&]() {
    if constexpr (RW > 0) {
      const int wv = RL->tid / 64, wl = RL->tid % 64;
      // what a role lane needs to know about ITS unit (does it have work, is its trajectory feasible, which trajectory is
      // it) was left in the unit's LDS block by the unit's own lane 0 before the barrier: no dependent global loads
      // (list entry -> trajectory state) at the head of the phase
      auto flagp = [&](int u) { return RL->base + (size_t)u * RL->usz + SM::OFF_RED + 12; };
      auto traj = [&](int u) { return (int)flagp(u)[2]; };
      auto live = [&](int u) { return u < RL->upb && flagp(u)[0] != 0.0; };
      auto feasible = [&](int u) { return flagp(u)[1] != 0.0; };
      // wavefront 0: chain; 1: Euler step + Lie Jacobians; the last: gaps (with four wavefronts it has nothing else to do;
      // with fewer it is the Euler wavefront again); RW - 1: state differences
      constexpr int NWV = (RW >= 3) ? 4 : RW;
      if (wv == 0) {
        if (live(wl)) chain_section(RL->base + (size_t)wl * RL->usz);
      } else if (wv == 1) {
        if (live(wl)) euler_section(RL->base + (size_t)wl * RL->usz);
      }
      if (wv == (NWV == 4 ? 3 : 1)) {
        if (live(wl)) gap_section(RL->base + (size_t)wl * RL->usz, traj(wl), feasible(wl));
      }
      if (wv == RW - 1) {
        const int u = wl / SM::NSLOT, q = wl % SM::NSLOT;
        if (live(u)) owner_section(RL->base + (size_t)u * RL->usz, 0, q);
      }
    }
  };
  // A unit without work (finished trajectory, b >= B) still lends its threads as role lanes.  The workgroup barriers
  // must sit in wave-uniform control flow -- a wavefront holds two units, one of which may be idle -- so the idle unit
  // skips the stages, not the barriers.
  bool unit_on = true;
  if constexpr (RW > 0) {
    unit_on = RL->active;
    // flags for the role lanes (see role_phase): every unit of the block writes them, idle ones too
    ex.each([&](int lane, int sl) {
      if (lane == 0) {
        N[SM::OFF_RED + 12] = unit_on ? 1.0 : 0.0;
        N[SM::OFF_RED + 13] = (unit_on && feas) ? 1.0 : 0.0;
        N[SM::OFF_RED + 14] = (double)b;
      }
    });
  }
  // ---- S0: load x, s, a ------------------------------------------------------------------------------------
  if (unit_on) ex.each([&](int lane, int sl) {
    const double* xg = D.xs + ((size_t)b * (T + 1) + t) * NX;
    const double* ag = D.acc + ((size_t)b * (T + 1) + t) * DM::NACC;
    const double* ug = D.us + ((size_t)b * T + (terminal ? 0 : t)) * NU;
    for (int i = lane; i < NX; i += lpu) N[SM::OFF_X + i] = xg[i];
    for (int i = lane; i < NV; i += lpu) N[SM::OFF_A + i] = ag[i];
    for (int i = lane; i < NU; i += lpu) N[SM::OFF_S + i] = terminal ? 0.0 : ug[i];
    // the next nominal state (origin of the gap) travels with the same batch of loads; parked in the gap slot
    if (!terminal && !feas)
      for (int i = lane; i < NX; i += lpu) N[SM::OFF_GAP + i] = xg[NX + i];
    if (lane == 0) N[SM::OFF_RED] = 0.0;  // cost accumulator
    if (CT && lane < 6) {
      N[SM::OFF_LAM + lane] = use_contact ? ag[NV + lane] : 0.0;
    }
  });
  ex.sync();
  LIN_STAMP(0);
  // ---- S1: squash (lanes < NU), joint sin/cos (next NJ lanes), base rotation (last lane) -------------------------
  if (unit_on) ex.each([&](int lane, int sl) {
    if (lane < NU) {
      double u = N[SM::OFF_S + lane], du = 1.0;
      if (P.use_squash) squash1(N[SM::OFF_S + lane], P.u_lb[lane], P.u_ub[lane], smooth, P.prm.smoothsat_power, u, du);
      N[SM::OFF_USQ + lane] = u;
      N[SM::OFF_DUS + lane] = du;
    } else if (lane < NU + DM::NJ) {
      const int j = lane - NU;
      const double th = N[SM::OFF_X + 7 + j];
      double s_, c_;
      fsincos(th, &s_, &c_);
      N[SM::OFF_CS + j] = c_;
      N[SM::OFF_SN + j] = s_;
    } else if (CT && lane == lpu - 2) {
      // contact force as a spatial force on the contact body (body coordinates): X_f^* [f; n]  (n = 0 for the 3D contact)
      if (use_contact) {
        double fl[3] = {N[SM::OFF_LAM], N[SM::OFF_LAM + 1], N[SM::OFF_LAM + 2]}, fb[3], rxf[3];
        matvec3<double>(m.frame_R[cframe], fl, fb);
        cross3<double>(m.frame_p[cframe], fb, rxf);
        if constexpr (CT == 6) {
          double nl[3] = {N[SM::OFF_LAM + 3], N[SM::OFF_LAM + 4], N[SM::OFF_LAM + 5]}, nb[3];
          matvec3<double>(m.frame_R[cframe], nl, nb);
#pragma unroll
          for (int i = 0; i < 3; ++i) rxf[i] += nb[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          N[SM::OFF_FEXT + i] = fb[i];
          N[SM::OFF_FEXT + 3 + i] = rxf[i];
        }
        if constexpr (CT == CT_PAIR3) {
          double fl2[3] = {N[SM::OFF_LAM + 3], N[SM::OFF_LAM + 4], N[SM::OFF_LAM + 5]}, fb2[3], rxf2[3];
          matvec3<double>(m.frame_R[cframe2], fl2, fb2);
          cross3<double>(m.frame_p[cframe2], fb2, rxf2);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            N[SM::OFF_FEXT2 + i] = fb2[i];
            N[SM::OFF_FEXT2 + 3 + i] = rxf2[i];
          }
        }
      }
    } else if (lane == lpu - 1) {
      double q[4] = {N[SM::OFF_X + 3], N[SM::OFF_X + 4], N[SM::OFF_X + 5], N[SM::OFF_X + 6]};
      double R0[9];
      quat_to_R(q, R0);
#pragma unroll
      for (int i = 0; i < 9; ++i) N[SM::OFF_R0 + i] = R0[i];
    }
  });
  ex.sync();
  LIN_STAMP(1);
  // ---- S2: nominal chain || Euler step and its Lie Jacobians (|| state differences of the first State-cost group) ----
  if constexpr (RW > 0) {
    ex.block_sync();
    role_phase();
    ex.block_sync();
    if (!unit_on) return;
  } else {
    ex.each([&](int lane, int sl) {
      if (lane == 0) chain_section(N);
    });
    LIN_STAMP(12);
    ex.each([&](int lane, int sl) {
      if (lane == 1) euler_section(N);
      if (lane == 2) gap_section(N, b, feas);
    });
    ex.sync();
  }
  LIN_STAMP(2);
  // ---- S3: tangent recursion; inertia columns to LDS ---------------------------------------------------------
  double dtau_l[Exec::SLOTS][NV];
  double jc_l[Exec::SLOTS][NCAP][6], dvc_l[Exec::SLOTS][NCAP][6];
  double dcon_l[Exec::SLOTS][NCR], dlam_l[Exec::SLOTS][NCR];
  ex.each([&](int lane, int sl) {
    double capdv[NCAP][6], capda[6] = {0, 0, 0, 0, 0, 0};
    double capda2[CT == CT_PAIR3 ? 6 : 1] = {0};
#pragma unroll
    for (int c = 0; c < NCAP; ++c)
#pragma unroll
      for (int i = 0; i < 6; ++i) capdv[c][i] = 0.0;
    if constexpr (CT == CT_PAIR3)
      lin2_tangent<DM>(m, N, lane, dtau_l[sl], ncap, capf, capdv, use_contact ? cbody : -1, capda, use_contact ? cbody2 : -1, capda2);
    else
      lin2_tangent<DM>(m, N, lane, dtau_l[sl], ncap, capf, capdv, use_contact ? cbody : -1, capda);
    if (lane >= 2 * NV && lane < 3 * NV) {
#pragma unroll
      for (int i = 0; i < NV; ++i) N[SM::OFF_M + i * NV + (lane - 2 * NV)] = dtau_l[sl][i];
    }
    if constexpr (FOLD) {
      // direction da of the last joint: da = S = [0; axis] on the last body, nothing on the others, no velocity terms --
      // dtau[NV - 1] = axis . (I S).angular, the operations lin2_tangent / lin2_dforce perform for that direction
      if (lane == 0) {
        const double Sl[6] = {0.0, 0.0, 0.0, m.axis[NB - 1][0], m.axis[NB - 1][1], m.axis[NB - 1][2]};
        double IS[6];
        inertia_apply<double>(m, NB - 1, Sl, IS);
        const double ax[3] = {m.axis[NB - 1][0], m.axis[NB - 1][1], m.axis[NB - 1][2]};
        N[SM::OFF_M + (NV - 1) * NV + (NV - 1)] = dot3<double>(ax, IS + 3);
      }
    }
#pragma unroll
    for (int c = 0; c < NCAP; ++c) {
      if (c >= ncap) continue;
      const int f = capf[c];
      const double* F = N + SM::OFF_FR + c * 18;
      lin2_frame_jcol<DM>(m, N, lane, m.frame_body[f], F, F + 9, jc_l[sl][c]);
      // frame velocity derivative column: Ad(bMf^-1) d(v_body)
      double wxr[3], tmp[3];
      cross3<double>(capdv[c] + 3, m.frame_p[f], wxr);
#pragma unroll
      for (int i = 0; i < 3; ++i) tmp[i] = capdv[c][i] + wxr[i];
      matTvec3<double>(m.frame_R[f], tmp, dvc_l[sl][c]);
      matTvec3<double>(m.frame_R[f], capdv[c] + 3, dvc_l[sl][c] + 3);
    }
    if constexpr (CT) {
#pragma unroll
      for (int r = 0; r < NCR; ++r) dcon_l[sl][r] = dlam_l[sl][r] = 0.0;
      if (use_contact) {
        if constexpr (CT == CT_PAIR3) {
          // two ContactModel3D: the 3D form below once per contact, rows 3 k .. 3 k + 2 (contact k on frame cf, capture slot cc,
          // tangent acceleration of its body cda)
          auto drift3 = [&](const EmpcContact& ctk, const int cf, const int cc, const double* cda, double* out3) {
            const double* F = N + SM::OFF_FR + cc * 18;
            double daf[3], wxr[3], tmp[3], gf[3], c1[3], c2[3], c3[3];
            cross3<double>(cda + 3, m.frame_p[cf], wxr);
#pragma unroll
            for (int i = 0; i < 3; ++i) tmp[i] = cda[i] + wxr[i];
            matTvec3<double>(m.frame_R[cf], tmp, daf);
            double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
            matTvec3<double>(F, ng, gf);
            double jcc[6], dvcc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              jcc[i] = jc_l[sl][0][i];
              dvcc[i] = dvc_l[sl][0][i];
            }
#pragma unroll
            for (int kk = 1; kk < NCAP; ++kk)
              if (kk == cc) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                  jcc[i] = jc_l[sl][kk][i];
                  dvcc[i] = dvc_l[sl][kk][i];
                }
              }
            cross3<double>(jcc + 3, gf, c1);
            cross3<double>(dvcc + 3, F + 12, c2);
            cross3<double>(F + 15, dvcc, c3);
#pragma unroll
            for (int i = 0; i < 3; ++i) out3[i] = daf[i] + c1[i] + c2[i] + c3[i];
            if (ctk.gains[0] != 0.0) {
              double wl[3];
              matvec3<double>(F, jcc, wl);  // oRf * (LOCAL linear Jacobian column)
#pragma unroll
              for (int i = 0; i < 3; ++i) out3[i] += ctk.gains[0] * wl[i];
            }
            if (ctk.gains[1] != 0.0)
#pragma unroll
              for (int i = 0; i < 3; ++i) out3[i] += ctk.gains[1] * dvcc[i];
          };
          double d0[3], d1[3];
          drift3(set.contacts[0], cframe, ccap, capda, d0);
          drift3(set.contacts[1], cframe2, ccap2, capda2, d1);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            dcon_l[sl][i] = d0[i];
            dcon_l[sl][3 + i] = d1[i];
          }
        } else {
          // derivative of the contact drift at fixed generalized acceleration.  3D: d(classical acceleration of the contact
          // frame origin, LOCAL) = d a_f.lin + dphi x (Rf^T (-g)) + d w_f x v_f.lin + w_f x d v_f.lin;
          // 6D: d(spatial acceleration, LOCAL) = [d a_f.lin + dphi x (Rf^T (-g)); d a_f.ang].  Baumgarte terms
          // (ContactModel3D/6D::calcDiff): 3D g0 oRf fJf.lin, 6D g0 Jlog6(Mref^-1 oMf) fJf, both g1 d(v_f)
          const EmpcContact& ctc = set.contacts[0];
          const double* F = N + SM::OFF_FR + ccap * 18;
          double daf[3], wxr[3], tmp[3], gf[3], c1[3], c2[3], c3[3];
          cross3<double>(capda + 3, m.frame_p[cframe], wxr);
  #pragma unroll
          for (int i = 0; i < 3; ++i) tmp[i] = capda[i] + wxr[i];
          matTvec3<double>(m.frame_R[cframe], tmp, daf);
          double ng[3] = {-m.gravity[0], -m.gravity[1], -m.gravity[2]};
          matTvec3<double>(F, ng, gf);
          double jcc[6], dvcc[6];
  #pragma unroll
          for (int i = 0; i < 6; ++i) {
            jcc[i] = jc_l[sl][0][i];
            dvcc[i] = dvc_l[sl][0][i];
          }
  #pragma unroll
          for (int kk = 1; kk < NCAP; ++kk)
            if (kk == ccap) {
  #pragma unroll
              for (int i = 0; i < 6; ++i) {
                jcc[i] = jc_l[sl][kk][i];
                dvcc[i] = dvc_l[sl][kk][i];
              }
            }
          cross3<double>(jcc + 3, gf, c1);
          if constexpr (CT == 6) {
            double dang[3];
            matTvec3<double>(m.frame_R[cframe], capda + 3, dang);
  #pragma unroll
            for (int i = 0; i < 3; ++i) {
              dcon_l[sl][i] = daf[i] + c1[i];
              dcon_l[sl][3 + i] = dang[i];
            }
            if (ctc.gains[0] != 0.0) {
              double rR[9], dp[3], rp[3], qq[4], xi[6], J6[36];
              matTmul3<double>(ctc.ref_R, F, rR);
  #pragma unroll
              for (int i = 0; i < 3; ++i) dp[i] = F[9 + i] - ctc.ref_p[i];
              matTvec3<double>(ctc.ref_R, dp, rp);
              R_to_quat(rR, qq);
              log6_quat(qq, rp, xi);
              Jlog6(xi, rp, J6);
  #pragma unroll
              for (int r = 0; r < 6; ++r) {
                double a_ = 0;
  #pragma unroll
                for (int l = 0; l < 6; ++l) a_ += J6[r * 6 + l] * jcc[l];
                dcon_l[sl][r] += ctc.gains[0] * a_;
              }
            }
          } else {
            cross3<double>(dvcc + 3, F + 12, c2);
            cross3<double>(F + 15, dvcc, c3);
  #pragma unroll
            for (int i = 0; i < 3; ++i) dcon_l[sl][i] = daf[i] + c1[i] + c2[i] + c3[i];
            if (ctc.gains[0] != 0.0) {
              double wl[3];
              matvec3<double>(F, jcc, wl);  // oRf * (LOCAL linear Jacobian column)
  #pragma unroll
              for (int i = 0; i < 3; ++i) dcon_l[sl][i] += ctc.gains[0] * wl[i];
            }
          }
          if (ctc.gains[1] != 0.0)
  #pragma unroll
            for (int r = 0; r < NCR; ++r) dcon_l[sl][r] += ctc.gains[1] * dvcc[r];
        }
        if (lane >= 2 * NV && lane < 3 * NV) {
          // direction da_j: d(con)/d(a_j) is column j of the contact Jacobian
#pragma unroll
          for (int r = 0; r < NCR; ++r) N[SM::OFF_JC + r * NV + (lane - 2 * NV)] = dcon_l[sl][r];
        }
      }
    }
  });
  ex.sync();
  LIN_STAMP(3);
  // ---- S4: Cholesky of M, in the registers of every lane that solves with it ------------------------------------------------
  // (Round 3 had lane 0 factor M in place in LDS -- 3.7k cycles with 31 lanes idle -- and every solve of S5 read the factor
  //  back entry by entry: 90 dependent LDS round trips per lane, ~6k cycles.  Now each solving lane reads the lower triangle
  //  once, in one block, factors it redundantly in registers (165 multiply-adds) and solves from registers.  Same
  //  factorisation, same substitutions, same order: chol_packed / chol_solve_packed.)
  double Lq_l[Exec::SLOTS][DM::NTRI];
  ex.each([&](int lane, int sl) {
    const int k = lane - 2 * NV;
    if (!(lane < NDX) && !(k >= 0 && k < NU)) return;  // (FOLD: all 32 lanes)
    const double* M = N + SM::OFF_M;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Lq_l[sl][i * (i + 1) / 2 + j] = M[i * NV + j];
    LIN_FENCE();
    chol_packed<NV>(Lq_l[sl]);
  });
  if constexpr (CT) {
    if (use_contact) {
      // M^-1 Jc^T (lanes 0..nc-1), then G = Jc M^-1 Jc^T and its Cholesky factor (lane 0)
      ex.each([&](int lane, int sl) {
        if (lane >= NCR) return;
        double y[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) y[i] = N[SM::OFF_JC + lane * NV + i];
        LIN_FENCE();
        chol_solve_packed<NV>(Lq_l[sl], y);
#pragma unroll
        for (int i = 0; i < NV; ++i) N[OFF_MIJ + i * NCR + lane] = y[i];
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane != 0) return;
        constexpr int NG = NCR * (NCR + 1) / 2;
        double G[NG];
        // (compile-time indices throughout: an array indexed by a run-time loop counter lives in scratch memory)
#pragma unroll
        for (int r = 0; r < NCR; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) {
            double g = 0;
            for (int i = 0; i < NV; ++i) g += N[SM::OFF_JC + r * NV + i] * N[OFF_MIJ + i * NCR + c];
            G[r * (r + 1) / 2 + c] = g;
          }
        if constexpr (CT == CT_PAIR3)
          chol_packed_stop<NCR>(G);  // (rank-deficient pairs: Eigen's stop-at-the-failed-pivot behaviour, see its comment)
        else
          chol_packed<NCR>(G);
#pragma unroll
        for (int i = 0; i < NG; ++i) N[OFF_G + i] = G[i];
      });
      ex.sync();
    }
  }
  LIN_STAMP(4);
  // ---- S5: M^-1 solves, Euler Jacobian columns -> tape ------------------------------------------------------------
  auto s5_column = [&](int lane, int sl, const bool xlane, const int k) {
    double da[NV];
    if (xlane) {
#pragma unroll
      for (int i = 0; i < NV; ++i) da[i] = -dtau_l[sl][i];
    } else {
      const double dus = N[SM::OFF_DUS + k];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double Bik;
        if (i < 6)
          Bik = (k < NROT) ? P.tau_f[i * NROT + k] : 0.0;
        else
          Bik = (k == NROT + i - 6) ? 1.0 : 0.0;
        da[i] = Bik * dus;
      }
    }
    chol_solve_packed<NV>(Lq_l[sl], da);
    if constexpr (CT) {
      if (use_contact) {
        // [M Jc^T; Jc 0][da; -dlam] = [rhs; -dcon]:  dlam = -(Jc M^-1 Jc^T)^-1 (Jc M^-1 rhs + dcon), da += M^-1 Jc^T dlam
        constexpr int NG = NCR * (NCR + 1) / 2;
        double z[NCR];
#pragma unroll
        for (int r = 0; r < NCR; ++r) {
          double a_ = xlane ? dcon_l[sl][r] : 0.0;
#pragma unroll
          for (int i = 0; i < NV; ++i) a_ += N[SM::OFF_JC + r * NV + i] * da[i];
          z[r] = -a_;
        }
        double Gf[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) Gf[i] = N[OFF_G + i];
        chol_solve_packed<NCR>(Gf, z);
#pragma unroll
        for (int r = 0; r < NCR; ++r) dlam_l[sl][r] = z[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          double a_ = N[OFF_MIJ + i * NCR] * z[0] + N[OFF_MIJ + i * NCR + 1] * z[1] + N[OFF_MIJ + i * NCR + 2] * z[2];
          if constexpr (NCR == 6)
            a_ += N[OFF_MIJ + i * NCR + 3] * z[3] + N[OFF_MIJ + i * NCR + 4] * z[4] + N[OFF_MIJ + i * NCR + 5] * z[5];
          da[i] += a_;
        }
      }
    }
    if (raw) {
      // column of da/dx (x lanes) or da/du (u lanes) in rows NV.. of the block; rows 0..NV-1 (dv/d. = [0 I | 0]) are implied
      if (xlane) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
          out[DM::OFF_FX + r * DM::NM + lane] = 0.0;
          out[DM::OFF_FX + (NV + r) * DM::NM + lane] = da[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
          out[DM::OFF_FU + r * DM::NM + k] = 0.0;
          out[DM::OFF_FU + (NV + r) * DM::NM + k] = da[r];
        }
      }
      return;
    }
    double G[NDX];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      G[i] = da[i] * dt * dt + ((xlane && lane >= NV && lane - NV == i) ? dt : 0.0);
      G[NV + i] = da[i] * dt;
    }
    double top[6];
    {
      // the Lie Jacobian of the Euler step, read as one block (36 broadcast reads in flight, one wait) before the products
      double J2r[36], J1c[6];
#pragma unroll
      for (int i = 0; i < 36; ++i) J2r[i] = N[SM::OFF_J2 + i];
#pragma unroll
      for (int r = 0; r < 6; ++r) J1c[r] = N[SM::OFF_J1 + r * 6 + (lane < 6 ? lane : 0)];
      LIN_FENCE();
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double a_ = 0;
#pragma unroll
        for (int l = 0; l < 6; ++l) a_ += J2r[r * 6 + l] * G[l];
        top[r] = a_;
      }
      if (xlane && lane < 6) {
#pragma unroll
        for (int r = 0; r < 6; ++r) top[r] += J1c[r];
      }
    }
    if (xlane) {
#pragma unroll
      for (int r = 0; r < 6; ++r) out[DM::OFF_FX + r * DM::NM + lane] = top[r];
#pragma unroll
      for (int r = 6; r < NDX; ++r) out[DM::OFF_FX + r * DM::NM + lane] = G[r] + ((r == lane) ? 1.0 : 0.0);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) out[DM::OFF_FU + r * DM::NM + k] = top[r];
#pragma unroll
      for (int r = 6; r < NDX; ++r) out[DM::OFF_FU + r * DM::NM + k] = G[r];
    }
    // gaps
    if (xlane) {
      if (!terminal) D.tape[((size_t)b * (T + 1) + t + 1) * REC + DM::OFF_GAP + lane] = N[SM::OFF_GAP + lane];
      if (t == 0) D.tape[((size_t)b * (T + 1)) * REC + DM::OFF_GAP + lane] = N[SM::OFF_GAP + NDX + lane];
    }
  };
  ex.each([&](int lane, int sl) {
    const bool xlane = lane < NDX;
    const int k = lane - 2 * NV;
    const bool ulane = k >= 0 && k < NU && lane < lpu;
    if (xlane || ulane) s5_column(lane, sl, xlane, k);
    if constexpr (FOLD) {
      if (lane == LXL) s5_column(lane, sl, false, KX);  // the folded control column: a second pass of this lane
    }
  });

  LIN_STAMP(5);
  // ---- S6: costs. Column accumulators live in registers: x lanes hold column `lane` of Lxx, u lanes column k of Luu.
  double hx_l[Exec::SLOTS][NDX];  // column of Lxx (x lanes) -- or of Luu in the first NU entries (u lanes)
  double lx_l[Exec::SLOTS];
  double hxu_l[Exec::SLOTS][CT ? NDX : 1];  // column of Lxu (u lanes; only the friction-cone cost couples x and u)
  ex.each([&](int lane, int sl) {
#pragma unroll
    for (int i = 0; i < NDX; ++i) hx_l[sl][i] = 0.0;
    lx_l[sl] = 0.0;
    if constexpr (CT) {
#pragma unroll
      for (int i = 0; i < NDX; ++i) hxu_l[sl][i] = 0.0;
    }
  });
  // round 1: State costs, up to NSLOT at a time.  Phase A: one lane per cost that owns its reference (ref_share, set by
  // prepare_problem) takes the state difference and its log Jacobian -- the only serial part.  Phase B: the activation of
  // all components at once, lane i = component i, costs in a uniform loop: the cost set is the same for the whole unit,
  // so type and weight are scalar loads and the branch on the activation type is uniform; the component parameters
  // (act_w / lb / ub) are one coalesced vector load each.  (Before: one lane per cost ran the 18 activations in a
  // row, after a copy of all parameters into LDS; 19k of a unit's 87k cycles.)
  for (int base = 0; base < si.n_state; base += SM::NSLOT) {
    // uniform facts of the group, loaded once
    bool on[SM::NSLOT];
    int own[SM::NSLOT], act[SM::NSLOT], cidx[SM::NSLOT];
    double wg[SM::NSLOT];
#pragma unroll
    for (int q = 0; q < SM::NSLOT; ++q) {
      on[q] = base + q < si.n_state;
      cidx[q] = on[q] ? si.state_ci[base + q] : si.state_ci[base];
      const EMPC_K EmpcCost& c = set.costs[cidx[q]];
      const int ow = on[q] ? si.state_own[base + q] : base;
      own[q] = (ow >= base) ? ow - base : q;  // slot holding this cost's residual and log Jacobian
      act[q] = c.activation;
      wg[q] = c.weight;
    }
    ex.sync();
    LIN_STAMP(10);
    if (RW == 0 || base > 0) {  // the first group was done in the role phase
      ex.each([&](int lane, int sl) {
        if (lane < SM::NSLOT) owner_section(N, base, lane);
      });
    }
    ex.sync();
    LIN_STAMP(11);
    ex.each([&](int lane, int sl) {
      if (lane >= NDX) return;
      // all parameter loads of the group first, then the arithmetic: one memory latency for the group, not one per cost
      double pw[SM::NSLOT], plb[SM::NSLOT], pub[SM::NSLOT];
#pragma unroll
      for (int q = 0; q < SM::NSLOT; ++q) {
        const EMPC_K EmpcCost& c = set.costs[cidx[q]];
        pw[q] = c.act_w[lane];
        plb[q] = c.lb[lane];
        pub[q] = c.ub[lane];
      }
#pragma unroll
      for (int q = 0; q < SM::NSLOT; ++q) {
        if (!on[q]) continue;
        double* S = N + SM::OFF_CST + q * SM::SLOT;
        double av, Ar, Arr;
        activation1(act[q], N[SM::OFF_CST + own[q] * SM::SLOT + lane], pw[q], plb[q], pub[q], av, Ar, Arr);
        S[NDX + lane] = wg[q] * Ar;
        S[2 * NDX + lane] = wg[q] * Arr;
        N[SM::OFF_RSH + q * NDX + lane] = av;  // the exchange area of the frame-cost round is free here
      }
    });
    ex.sync();
    LIN_STAMP(13);
    ex.each([&](int lane, int sl) {
      if (lane == 0) {
        double a_ = 0;
#pragma unroll
        for (int q = 0; q < SM::NSLOT; ++q) {
          if (!on[q]) continue;
          double cv = 0;
#pragma unroll
          for (int i = 0; i < NDX; ++i) cv += N[SM::OFF_RSH + q * NDX + i];
          a_ += wg[q] * cv;
        }
        N[SM::OFF_RED] += a_;
      }
      if (lane >= NDX) return;
#pragma unroll
      for (int q = 0; q < SM::NSLOT; ++q) {
        if (!on[q]) continue;
        const double* S = N + SM::OFF_CST + q * SM::SLOT;
        const double* J6 = N + SM::OFF_CST + own[q] * SM::SLOT + 3 * NDX;
        // operands first, as one block of LDS reads (the base lanes need the whole log Jacobian and the six leading activation
        // derivatives: 48 reads in flight instead of 120 read-wait pairs), then the arithmetic in its original order
        double J6r[36], J6c[6], Ar6[6], Arr6[6];
        if (lane < 6) {
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) J6c[rr] = J6[rr * 6 + lane];  // column `lane` of the log Jacobian
#pragma unroll
          for (int i = 0; i < 36; ++i) J6r[i] = J6[i];
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) {
            Ar6[rr] = S[NDX + rr];
            Arr6[rr] = S[2 * NDX + rr];
          }
        }
        const double ar_own = S[NDX + lane], arr_own = S[2 * NDX + lane];
        LIN_FENCE();
        if (lane < 6) {
          double g = 0;
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) g += J6c[rr] * Ar6[rr];
          lx_l[sl] += g;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double h = 0;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) h += J6r[rr * 6 + i] * Arr6[rr] * J6c[rr];
            hx_l[sl][i] += h;
          }
        } else {
          lx_l[sl] += ar_own;
#pragma unroll
          for (int i = 6; i < NDX; ++i) hx_l[sl][i] += (i == lane) ? arr_own : 0.0;  // (a select, not a branch per entry)
        }
      }
    });
    ex.sync();
  }
  LIN_STAMP(6);
  // round 2: Control costs (including the barrier): component k on u lane k
  // (component k of every Control cost of the set: value sum returned, gradient entry added to `lxa`, Hessian diagonal entry
  //  to `diag` -- a Control cost is separable, it only touches the diagonal of Luu)
  auto ctrl_component = [&](const int k, double& lxa, double& diag) {
    double cv = 0;
    // The set's Control costs from the host-made list (same order as the table scan), in groups of CG: the lane's parameters
    // of the whole group (reference, weight, bounds of component k: vector loads from the problem image) are requested before
    // the first activation is evaluated -- one memory round trip per group instead of one per cost.
    constexpr int CG = 3;
    const double sk = N[SM::OFF_S + k];
    const auto PL = platform_of<DM>(P);
    for (int base = 0; base < si.n_ctrl; base += CG) {
      double pref[CG], pw[CG], plb[CG], pub[CG], wgt[CG];
      int act[CG];
      bool on[CG];
#pragma unroll
      for (int q = 0; q < CG; ++q) {
        on[q] = base + q < si.n_ctrl;
        const EMPC_K EmpcCost& c = set.costs[si.ctrl_ci[on[q] ? base + q : base]];
        pref[q] = c.ref[k];
        pw[q] = act_weight(c, k, smooth, PL);
        plb[q] = c.lb[k];
        pub[q] = c.ub[k];
        wgt[q] = c.weight;
        act[q] = c.activation;
      }
      LIN_FENCE();
#pragma unroll
      for (int q = 0; q < CG; ++q) {
        if (!on[q]) continue;
        double av, Ar, Arr;
        activation1(act[q], sk - pref[q], pw[q], plb[q], pub[q], av, Ar, Arr);
        cv += wgt[q] * av;
        lxa += wgt[q] * Ar;
        diag += wgt[q] * Arr;
      }
    }
    return cv;
  };
  double lux_l[Exec::SLOTS], luux_l[Exec::SLOTS];  // FOLD: gradient / Hessian-diagonal entry of the folded control column
  ex.each([&](int lane, int sl) {
    const int k = lane - 2 * NV;
    if (k >= 0 && k < NU && lane < lpu) {
      double diag = 0.0;
      const double cv = ctrl_component(k, lx_l[sl], diag);
#pragma unroll
      for (int i = 0; i < NU; ++i) hx_l[sl][i] += (i == k) ? diag : 0.0;  // (a select, not a branch per entry)
      N[SM::OFF_RED + 1 + k] = cv;
    }
    if constexpr (FOLD) {
      if (lane == LXL) {
        lux_l[sl] = 0.0;
        luux_l[sl] = 0.0;
        N[SM::OFF_RED + 1 + KX] = ctrl_component(KX, lux_l[sl], luux_l[sl]);
      }
    }
  });
  ex.sync();
  ex.each([&](int lane, int sl) {
    if (lane == 0) {
      double a_ = 0;
      for (int k = 0; k < NU; ++k) a_ += N[SM::OFF_RED + 1 + k];
      N[SM::OFF_RED] += a_;
    }
  });
  LIN_STAMP(7);
  // round 3: frame costs, one at a time
  for (int ci = 0; FR && ci < set.ncosts; ++ci) {
    const EMPC_K EmpcCost& c = set.costs[ci];
    if (!c.active || c.frame < 0 || c.type == EMPC_COST_CONTACT_FRICTION_CONE || c.type == EMPC_COST_STATE ||
        c.type == EMPC_COST_CONTROL)
      continue;
    int cc = 0;
#pragma unroll
    for (int kk = 1; kk < NCAP; ++kk)
      if (kk < ncap && capf[kk] == c.frame) cc = kk;
    const int nr = (c.type == EMPC_COST_FRAME_PLACEMENT || c.type == EMPC_COST_FRAME_VELOCITY) ? 6 : 3;
    ex.sync();
    // nominal residual, activation and (for log-map residuals) the Jacobian of the log, by lane 0 -> slot 0
    ex.each([&](int lane, int sl) {
      if (lane != 0) return;
      const double* F = N + SM::OFF_FR + cc * 18;
      double* S = N + SM::OFF_CST;
      double r[6];
      if (c.type == EMPC_COST_FRAME_PLACEMENT) {
        double rR[9], dp[3], rp[3], qq[4], J6[36];
        matTmul3<double>(c.ref + 3, F, rR);
#pragma unroll
        for (int i = 0; i < 3; ++i) dp[i] = F[9 + i] - c.ref[i];
        matTvec3<double>(c.ref + 3, dp, rp);
        R_to_quat(rR, qq);
        log6_quat(qq, rp, r);
        Jlog6(r, rp, J6);
#pragma unroll
        for (int i = 0; i < 36; ++i) S[3 * NDX + i] = J6[i];
      } else if (c.type == EMPC_COST_FRAME_ROTATION) {
        double rR[9], qq[4], J3[9];
        matTmul3<double>(c.ref, F, rR);
        R_to_quat(rR, qq);
        quat_log3(qq, r);
        SO3Coef kc;
        so3_coef(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], kc);
        Jlog3(r, kc, J3);
#pragma unroll
        for (int i = 0; i < 9; ++i) S[3 * NDX + i] = J3[i];
      } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = F[9 + i] - c.ref[i];
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = F[12 + i] - c.ref[i];
      }
      double cv = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // (compile-time indices: r stays in registers)
        if (i >= nr) continue;
        double av, Ar, Arr;
        activation1(c.activation, r[i], c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
        cv += av;
        S[NDX + i] = c.weight * Ar;
        S[2 * NDX + i] = c.weight * Arr;
      }
      N[SM::OFF_RED] += c.weight * cv;
    });
    ex.sync();
    double wcol_l[Exec::SLOTS][6];
    ex.each([&](int lane, int sl) {
      if (lane >= NDX) return;
      const double* F = N + SM::OFF_FR + cc * 18;
      const double* S = N + SM::OFF_CST;
      double jcc[6], dvcc[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        jcc[i] = jc_l[sl][0][i];
        dvcc[i] = dvc_l[sl][0][i];
      }
#pragma unroll
      for (int kk = 1; kk < NCAP; ++kk)
        if (kk == cc) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            jcc[i] = jc_l[sl][kk][i];
            dvcc[i] = dvc_l[sl][kk][i];
          }
        }
      double col[6] = {0, 0, 0, 0, 0, 0};
      if (c.type == EMPC_COST_FRAME_PLACEMENT) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a_ = 0;
#pragma unroll
          for (int l = 0; l < 6; ++l) a_ += S[3 * NDX + i * 6 + l] * jcc[l];
          col[i] = a_;
        }
      } else if (c.type == EMPC_COST_FRAME_ROTATION) {
        matvec3<double>(S + 3 * NDX, jcc + 3, col);
      } else if (c.type == EMPC_COST_FRAME_TRANSLATION) {
        matvec3<double>(F, jcc, col);
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = dvcc[i];
      }
      double g = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // (compile-time indices: col / wcol stay in registers)
        if (i >= nr) continue;
        g += col[i] * S[NDX + i];
        wcol_l[sl][i] = S[2 * NDX + i] * col[i];
        N[SM::OFF_RSH + i * NDX + lane] = col[i];
      }
      lx_l[sl] += g;
    });
    ex.sync();
    ex.each([&](int lane, int sl) {
      if (lane >= NDX) return;
      const int ni = (c.type == EMPC_COST_FRAME_VELOCITY) ? NDX : NV;
#pragma unroll
      for (int i = 0; i < NDX; ++i) {
        if (i >= ni) continue;
        double h = 0;
#pragma unroll
        for (int rr = 0; rr < 6; ++rr)
          if (rr < nr) h += N[SM::OFF_RSH + rr * NDX + i] * wcol_l[sl][rr];
        hx_l[sl][i] += h;
      }
    });
  }
  if constexpr (CT) {
    // round 4: ContactFrictionCone (r = A lambda; Rx = A dlambda/dx, Ru = A dlambda/du)
    for (int ci = 0; ci < set.ncosts; ++ci) {
      const EMPC_K EmpcCost& c = set.costs[ci];
      if (!c.active || c.type != EMPC_COST_CONTACT_FRICTION_CONE || !use_contact) continue;
      ex.sync();
      ex.each([&](int lane, int sl) {
        if (lane != 0) return;
        double AR[5][3];  // A R_n^T, precomputed by prepare_problem in ref[4..18]
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) AR[i][j] = c.ref[4 + 3 * i + j];
        double cv = 0;
        const int fo = cone_force_offset<CT>(set, c.frame);  // CT_PAIR3: the contact on the cost's frame; 0 otherwise
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const double r = AR[i][0] * N[SM::OFF_LAM + fo] + AR[i][1] * N[SM::OFF_LAM + fo + 1] + AR[i][2] * N[SM::OFF_LAM + fo + 2];
          double av, Ar, Arr;
          activation1(c.activation, r, c.act_w[i], c.lb[i], c.ub[i], av, Ar, Arr);
          cv += av;
          N[SM::OFF_CONE + 15 + i] = c.weight * Ar;
          N[SM::OFF_CONE + 20 + i] = c.weight * Arr;
          for (int j = 0; j < 3; ++j) N[SM::OFF_CONE + i * 3 + j] = AR[i][j];
        }
        N[SM::OFF_RED] += c.weight * cv;
      });
      ex.sync();
      double wc_l[Exec::SLOTS][5];
      ex.each([&](int lane, int sl) {
        const bool xlane = lane < NDX;
        const int k = lane - 2 * NV;
        const bool ulane = k >= 0 && k < NU;
        if (!xlane && !ulane) return;
        const int colidx = xlane ? lane : NDX + k;
        double g = 0;
        double dl3[3] = {dlam_l[sl][0], dlam_l[sl][1], dlam_l[sl][2]};
        if constexpr (CT == CT_PAIR3) {
          if (cone_force_offset<CT>(set, c.frame) != 0) {
            dl3[0] = dlam_l[sl][3];
            dl3[1] = dlam_l[sl][4];
            dl3[2] = dlam_l[sl][5];
          }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const double col = N[SM::OFF_CONE + i * 3] * dl3[0] + N[SM::OFF_CONE + i * 3 + 1] * dl3[1] +
                             N[SM::OFF_CONE + i * 3 + 2] * dl3[2];
          g += col * N[SM::OFF_CONE + 15 + i];
          wc_l[sl][i] = N[SM::OFF_CONE + 20 + i] * col;
          N[SM::OFF_RSH + i * (NDX + NU) + colidx] = col;
        }
        lx_l[sl] += g;
      });
      ex.sync();
      ex.each([&](int lane, int sl) {
        const bool xlane = lane < NDX;
        const int k = lane - 2 * NV;
        const bool ulane = k >= 0 && k < NU;
        if (!xlane && !ulane) return;
        if (xlane) {
#pragma unroll
          for (int i = 0; i < NDX; ++i) {
            double h = 0;
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) h += N[SM::OFF_RSH + rr * (NDX + NU) + i] * wc_l[sl][rr];
            hx_l[sl][i] += h;
          }
        } else {
#pragma unroll
          for (int i = 0; i < NDX; ++i) {
            double h = 0;
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) h += N[SM::OFF_RSH + rr * (NDX + NU) + i] * wc_l[sl][rr];
            hxu_l[sl][i] += h;
          }
#pragma unroll
          for (int l = 0; l < NU; ++l) {
            double h = 0;
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) h += N[SM::OFF_RSH + rr * (NDX + NU) + NDX + l] * wc_l[sl][rr];
            hx_l[sl][l] += h;
          }
        }
      });
    }
  }
  ex.sync();
  LIN_STAMP(8);
  // ---- S7: scale and store ---------------------------------------------------------------------------------------------
  const double cscale = raw ? 1.0 : ((terminal && !P.prm.terminal_dt_scaling) ? 1.0 : dt);
  ex.each([&](int lane, int sl) {
    if (lane < NDX) {
      out[DM::OFF_LX + lane] = lx_l[sl] * cscale;
#pragma unroll
      for (int i = 0; i < NDX; ++i)
        if (DM::stored_xx(i, lane)) out[DM::lxx(i, lane)] = hx_l[sl][i] * cscale;  // (EMPC_REC_TRI: rows i <= lane of the column)
    }
    const int k = lane - 2 * NV;
    if (k >= 0 && k < NU && lane < lpu) {
      out[DM::OFF_LU + k] = lx_l[sl] * cscale;
#pragma unroll
      for (int i = 0; i < NU; ++i)
        if (DM::stored_xx(i, k)) out[DM::luu(i, k)] = hx_l[sl][i] * cscale;
#pragma unroll
      for (int i = 0; i < NDX; ++i) {
        double v_ = 0.0;
        if constexpr (CT) v_ = hxu_l[sl][i] * cscale;
        out[DM::lxu(i, k)] = v_;
      }
    }
    if constexpr (FOLD) {
      if (lane == LXL) {  // the folded control column: Lu entry, column of Luu (diagonal entry only), column of Lxu (zeros)
        out[DM::OFF_LU + KX] = lux_l[sl] * cscale;
#pragma unroll
        for (int i = 0; i < NU; ++i)
          if (DM::stored_xx(i, KX)) out[DM::luu(i, KX)] = ((i == KX) ? luux_l[sl] : 0.0) * cscale;
#pragma unroll
        for (int i = 0; i < NDX; ++i) out[DM::lxu(i, KX)] = 0.0;
      }
    }
    if (lane == 0) out[DM::OFF_COST] = N[SM::OFF_RED] * cscale;
  });
  LIN_STAMP(9);
#if defined(EMPC_STAMPS) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
  if (b == 0 && t == 10)
    ex.each([&](int lane, int sl) {
      if (lane == 0)
        for (int i = 0; i < 14; ++i) D.dbg[32 + i] = lst[i];
    });
#endif
}

}  // namespace empc
